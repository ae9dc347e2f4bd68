// ubench_mfma_peak.hip -- what the f32 matrix pipe SUSTAINS chip-wide: every CU runs W waves per SIMD of back-to-back, independent
// v_mfma_f32_32x32x2_f32 out of registers (no memory traffic at all) for tens of milliseconds; reports TFLOP/s (hipEvents) and the
// shader clock under that load (s_memtime ticks / wall-clock ticks of the same wave).  The nominal 157.3 TFLOP/s assumes 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_peak.bin scripts/ubench_mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int NACC, int NV>
__global__ void __launch_bounds__(256) burn(float* out, int iters, u64* clk) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    float x[4] = {a, b, a + b, a - b};
    const u64 c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q & 3]) : "v"(a), "v"(b));     // NV plain VALU ops per MFMA, same wave
            }
    }
    const u64 c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC, int NV>
void run(int wg_per_cu, int ncu, int iters) {
    float* out; u64* clk;
    const int grid = ncu * wg_per_cu;
    CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((burn<NACC, NV>), dim3(grid), dim3(256), 0, 0, out, iters / 8, clk);     // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((burn<NACC, NV>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    u64 h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flop = (double)grid * 4 * (double)iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%d waves/SIMD, %d independent accumulators, %d VALU ops per MFMA: %.1f ms, %.1f TFLOP/s = %.1f %% of 157.3; shader clock under load %.0f MHz (%.1f cycles per MFMA per SIMD)\n",
           wg_per_cu, NACC, NV, ms, flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 157.3, (double)h[0] / ((double)h[1] / 100.0),
           (double)h[0] / ((double)iters * 8 * NACC * wg_per_cu));
    CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, ncu);
    run<4, 0>(1, ncu, 40000);
    run<4, 0>(2, ncu, 20000);
    run<8, 0>(2, ncu, 10000);
    run<4, 0>(2, ncu, 80000);       // a longer run: power management has settled
    // does ordinary VALU work of the same SIMD take matrix-pipe time?  (f32 MFMA and packed f32 FMA have the same peak rate)
    run<4, 1>(2, ncu, 20000);
    run<4, 2>(2, ncu, 20000);
    run<4, 4>(2, ncu, 20000);
    run<4, 8>(2, ncu, 20000);
    run<4, 4>(1, ncu, 40000);
    return 0;
}
