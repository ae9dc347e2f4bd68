// ubench_mfma_mix.hip -- which ingredient of a real f32 MFMA GEMM loop takes matrix-pipe time?  Two workgroups of 4 waves per CU
// (2 waves per SIMD, like wnv_fwd_layer_kernel), every wave runs steps of 64 v_mfma_f32_32x32x2_f32 (8 k-pairs x 8 tiles); variants add,
// one at a time: operands read from LDS one k-pair ahead, a barrier per step, LDS stores + global loads in the MFMAs' shadow.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_mix.bin scripts/ubench_mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int LDSOPS, int BARRIER, int STAGE, int NVALU>
__global__ void __launch_bounds__(256, 2) mix(float* out, const float* gsrc, int steps) {
    __shared__ __attribute__((aligned(16))) float xt[2][16 * 132];
    __shared__ __attribute__((aligned(16))) float wc[2][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 132; i += 256) (&xt[0][0])[i] = 1.0f + i * 1e-7f;
    for (int i = tid; i < 2 * 4096; i += 256) (&wc[0][0])[i] = 1.0f - i * 1e-7f;
    __syncthreads();
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float4 st[4] = {make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4)};
    float xv = 1.0f;
    const float* gp = gsrc + (size_t)(blockIdx.x & 63) * 4096 + 4 * tid;
    for (int s = 0; s < steps; ++s) {
        const float* Xb = &xt[s & 1][0] + (lane >> 5) * 132 + 32 * wave + (lane & 31);
        const float* Wa = &wc[s & 1][0] + (lane >> 5) * 256 + (lane & 31);
        float bv = LDSOPS ? Xb[0] : 1.0f, av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = LDSOPS ? Wa[32 * i] : 1.0f + i;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float bn = bv, an[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) an[i] = av[i];
            if (LDSOPS && ks + 1 < 8) {
                bn = Xb[(2 * ks + 2) * 132];
#pragma unroll
                for (int i = 0; i < 8; ++i) an[i] = Wa[(2 * ks + 2) * 256 + 32 * i];
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (STAGE && ks < 4) {
                reinterpret_cast<float4*>(&wc[(s + 1) & 1][0])[ks * 256 + tid] = st[ks];
                st[ks] = *reinterpret_cast<const float4*>(gp + (size_t)ks * 1024);
            }
#pragma unroll
            for (int q = 0; q < NVALU; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(xv));
            bv = bn;
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = an[i];
        }
        if (BARRIER) __syncthreads();
    }
    float r = xv + st[0].x + st[1].y + st[2].z + st[3].w;
    for (int i = 0; i < 8; ++i)
        for (int v = 0; v < 16; ++v) r += acc[i][v];
    out[blockIdx.x * 256 + tid] = r;
}

template <int LDSOPS, int BARRIER, int STAGE, int NVALU>
void run(const char* what, int ncu, float* out, const float* gsrc) {
    const int steps = 4000, grid = 2 * ncu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix<LDSOPS, BARRIER, STAGE, NVALU>), dim3(grid), dim3(256), 0, 0, out, gsrc, steps / 8);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix<LDSOPS, BARRIER, STAGE, NVALU>), dim3(grid), dim3(256), 0, 0, out, gsrc, steps);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 * steps * 64.0 * 4096;
    printf("%-78s %6.1f TFLOP/s = %5.1f %% of 157.3\n", what, flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 157.3);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    float *out, *gsrc;
    CK(hipMalloc(&out, (size_t)2 * ncu * 256 * 4)); CK(hipMalloc(&gsrc, (size_t)64 * 4096 * 4 + 65536)); CK(hipMemset(gsrc, 0, (size_t)64 * 4096 * 4 + 65536));
    run<0, 0, 0, 0>("MFMAs only (operands in registers)", ncu, out, gsrc);
    run<1, 0, 0, 0>("+ operands read from LDS one k-pair ahead (9 ds_read_b32 per 8 MFMAs)", ncu, out, gsrc);
    run<1, 1, 0, 0>("+ a barrier per 64 MFMAs", ncu, out, gsrc);
    run<1, 1, 1, 0>("+ staging: 4 ds_write_b128 + 4 global_load_dwordx4 per 64 MFMAs", ncu, out, gsrc);
    run<1, 1, 1, 1>("+ 1 VALU op per 8 MFMAs", ncu, out, gsrc);
    run<1, 1, 1, 4>("+ 4 VALU ops per 8 MFMAs", ncu, out, gsrc);
    run<1, 1, 1, 12>("+ 12 VALU ops per 8 MFMAs (= v12's measured 1.5 per MFMA)", ncu, out, gsrc);
    run<0, 1, 0, 0>("MFMAs from registers + a barrier per 64 MFMAs", ncu, out, gsrc);
    run<0, 0, 1, 0>("MFMAs from registers + staging", ncu, out, gsrc);
    return 0;
}
