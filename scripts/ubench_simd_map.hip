// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id ...)
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_simd_map.hip -o scripts/ubench_simd_map.bin && scripts/ubench_simd_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out, int lds_bytes_dummy) {
    extern __shared__ float smem[];
    if (lds_bytes_dummy < 0) smem[threadIdx.x] = 1.f;
    if ((threadIdx.x & 63) == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x));
        out[blockIdx.x * 8 + (threadIdx.x >> 6)] = x;
    }
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4); hipMemset(d, 0, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k, dim3(224), dim3(512), 100 * 1024, 0, d, 0);      // one workgroup per CU (100 KB LDS each)
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < 224; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d[simd %u wave_id %u cu %u]", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
    return 0;
}
