// ubench_valu.hip -- issue cost (cycles per wave-instruction) of v_fma_f32 / v_pk_fma_f32 / DPP adds / ds_read_b128 on gfx950,
// with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_valu.bin scripts/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int MODE>
__global__ void k(float* out, u64* cyc, int n) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    f2 a[8]; f2 w[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{(float)tid, 1.f}; w[i] = f2{1.0001f + i, 0.9999f}; s[i] = (float)tid + i; }
    const f2 x = f2{out[tid & 7], out[(tid + 1) & 7]};
    u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {          // 64 independent-ish v_pk_fma_f32 (8 chains)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(w[i], x, a[i]);
        } else if (MODE == 1) {   // 64 v_fma_f32 (8 chains)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = fmaf(s[i], x.x, w[i].x);
        } else if (MODE == 2) {   // 64 DPP adds (8 chains)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s[i]), 0xB1, 0xF, 0xF, true));
        } else if (MODE == 3) {   // 16 ds_read_b128 + 64 v_pk_fma (the stage's z phase shape), same address per quad lane
            const float4* p = reinterpret_cast<const float4*>(lds) + (tid & 3) * 9 + (it & 1) * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 v = p[i];
                a[0] = __builtin_elementwise_fma(w[0], f2{v.x, v.y}, a[0]); a[1] = __builtin_elementwise_fma(w[1], f2{v.z, v.w}, a[1]);
                a[2] = __builtin_elementwise_fma(w[2], f2{v.x, v.y}, a[2]); a[3] = __builtin_elementwise_fma(w[3], f2{v.z, v.w}, a[3]);
                a[4] = __builtin_elementwise_fma(w[4], f2{v.x, v.y}, a[4]); a[5] = __builtin_elementwise_fma(w[5], f2{v.z, v.w}, a[5]);
                a[6] = __builtin_elementwise_fma(w[6], f2{v.x, v.y}, a[6]); a[7] = __builtin_elementwise_fma(w[7], f2{v.z, v.w}, a[7]);
            }
        } else if (MODE == 4) {   // 64 v_exp_f32
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
        }
    }
    u64 t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
    out[8 + tid] = r;
    if (tid == 0) cyc[0] = t1 - t0;
}

template <int MODE> void run(const char* name, int per_iter, float* d, u64* c) {
    for (int nw : {1, 4, 8, 16}) {
        const int n = 2000;
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * nw), 0, 0, d, c, n);
        CK(hipDeviceSynchronize());
        u64 cy; CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
        printf("%-34s waves/CU=%2d (%.1f per SIMD): %6.2f cycles per wave-instruction (wave 0 view), %6.2f SIMD-cycles per instr\n", name, nw,
               nw / 4.0, (double)cy / n / per_iter, (double)cy / n / per_iter / (nw < 4 ? 1.0 : nw / 4.0));
    }
}
int main() {
    float* d; u64* c;
    CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&c, 64));
    CK(hipMemset(d, 0, 4096 * 4));
    run<0>("v_pk_fma_f32 x64", 64, d, c);
    run<1>("v_fma_f32 x64", 64, d, c);
    run<2>("v_add_f32_dpp x64", 64, d, c);
    run<4>("v_exp_f32 x64", 64, d, c);
    run<3>("8 ds_read_b128 + 32 v_pk_fma_f32", 1, d, c);
    return 0;
}
