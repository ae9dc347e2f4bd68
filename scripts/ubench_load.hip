// ubench_load.hip -- does a stage's CHAIN PHASE (LDS reads, 64 packed FMAs per lane against register-resident weights, a DPP reduce,
// one granule store: no wait on memory) take longer when the rest of the chip is busy?  profiles/r04_throughput_bound_final_stages.txt
// left that open: at 48-64 utterances per GPU the same instruction sequence takes ~440 ns in the ring against ~290 ns at 8, the clock
// reads 2.4 GHz either way and the instruction cache hits.
// Block 0 (the probe) times REPS back-to-back phases (one barrier between them, as in the ring) while the other 255 workgroups
//   mode 0: exit at once                       mode 1: poll mailboxes with L1-bypassing loads (what waiting stages do)
//   mode 2: run packed FMAs on registers       mode 3: store granules (plain, same-XCD style) as fast as they issue
//   mode 4: store granules write-through       mode 5: FMAs + LDS traffic (a tap workgroup's mat-vec)
// and prints ns per phase for every mode.   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_load.bin scripts/ubench_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ u64 ld_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(512) k(int mode, int reps, u64* mail, unsigned* stop, u64* result, float* sink) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    f2 w[8][8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) w[s][c] = f2{0.001f * (float)(tid + s), 0.002f * (float)(lane - c)};
    for (int i = tid; i < 4096; i += 512) lds[i] = 0.01f * (float)i;
    __syncthreads();
    if (blockIdx.x == 0) {
        // ---- the probe: REPS chain phases ------------------------------------------------------------------------------------------
        float accum = 0.f;
        u64 t0 = 0, t1 = 0;
        for (int r = -8; r < reps; ++r) {
            if (r == 0) { __syncthreads(); t0 = __builtin_amdgcn_s_memrealtime(); }
            __syncthreads();
            if (tid < 256) {                                                  // the four chain waves
                float x[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(lds + 20 * (tid & 7) + 4 * q + ((r & 1) ? 256 : 0));
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
                f2 acc[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[s] = f2{accum * 1e-30f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc[s] = __builtin_elementwise_fma(w[s][c], f2{x[2 * c], x[2 * c + 1]}, acc[s]);
                float q8[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) q8[s] = acc[s].x + acc[s].y;
                float m = dpp_add<0x141>(q8[0] + q8[4]) + dpp_add<0x141>(q8[1] + q8[5]) + dpp_add<0x141>(q8[2] + q8[6]) + dpp_add<0x141>(q8[3] + q8[7]);
                m = dpp_add<0x4E>(m); m = dpp_add<0xB1>(m);
                const float u = __builtin_amdgcn_rcpf(1.0f + __expf(-m)) * m;      // a gate-sized tail
                accum = u;
                if ((tid & 7) == 0) {                                           // the granule store of the send
                    const u64 g = ((u64)(unsigned)(r + 9) << 32) | (u64)__float_as_uint(u);
                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(mail + 4096 + (tid >> 3)), "v"(g) : "memory");
                }
                lds[2048 + tid] = u;
            }
        }
        __syncthreads();
        t1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) { result[0] = t1 - t0; __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        sink[tid] = accum;
        return;
    }
    // ---- the background ----------------------------------------------------------------------------------------------------------------
    if (mode == 0) return;
    float bg = 0.f;
    unsigned it = 0;
    u64* mine = mail + (size_t)blockIdx.x * 16;
    for (;;) {
        if (mode == 1) {                                                      // waiting stages: wave 0 polls two granules per lane
            if (tid < 64) { const u64 a = ld_granule(mail + 8192 + ((blockIdx.x * 64 + lane) & 4095)); bg += (float)(unsigned)a; }
        } else if (mode == 2 || mode == 5) {
            f2 acc[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[s] = f2{bg * 1e-30f, 0.f};
            float x0 = 1.0f, x1 = 0.5f;
            if (mode == 5) { x0 = lds[(tid * 4 + it) & 2047]; x1 = lds[(tid * 4 + it + 1) & 2047]; }
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[s] = __builtin_elementwise_fma(w[s][c], f2{x0, x1}, acc[s]);
            bg = acc[0].x + acc[1].y + acc[2].x + acc[3].y + acc[4].x + acc[5].y + acc[6].x + acc[7].y;
        } else if (mode == 3 || mode == 4) {
            if (tid < 64) {
                const u64 g = ((u64)it << 32) | (u64)lane;
                if (mode == 3) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(mine + (lane & 15)), "v"(g) : "memory");
                else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(mine + (lane & 15)), "v"(g) : "memory");
            }
        }
        if ((++it & 63u) == 0u && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    }
    sink[512 + (blockIdx.x & 255) * 2 + (tid & 1)] = bg;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20000;
    const int nwg = argc > 2 ? atoi(argv[2]) : 256;                       // workgroups in all (probe + background)
    u64 *mail, *result;
    unsigned* stop;
    float* sink;
    CK(hipMalloc(&mail, (size_t)(1 << 16) * 8)); CK(hipMemset(mail, 0, (size_t)(1 << 16) * 8));
    CK(hipMalloc(&result, 64)); CK(hipMalloc(&stop, 64)); CK(hipMalloc(&sink, 4096 * 4));
    const char* names[] = {"nothing else runs", "255 workgroups poll (one wave each, L1-bypassing loads)", "255 workgroups run packed FMAs", "255 workgroups store granules (plain)",
                           "255 workgroups store granules (write-through)", "255 workgroups run FMAs fed from LDS"};
    for (int pass = 0; pass < 2; ++pass)
        for (int mode = 0; mode < 6; ++mode) {
            CK(hipMemset(stop, 0, 4));
            // (100 KB of LDS per workgroup: at most ONE workgroup per CU, as in the ring -- a background workgroup must not share the probe's SIMDs)
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 100 * 1024, 0, mode, reps, mail, stop, result, sink);
            CK(hipDeviceSynchronize());
            u64 ticks = 0;
            CK(hipMemcpy(&ticks, result, 8, hipMemcpyDeviceToHost));
            if (pass == 1) printf("%d workgroups, mode %d  %-62s %7.1f ns per chain phase (+ barrier)\n", nwg, mode, names[mode], (double)ticks * 10.0 / reps);
        }
    return 0;
}
