// ubench_coresident.hip -- can THREE separately compiled persistent kernels, launched on three streams, be co-resident with the placement
// the ring layout needs?  (stage / tap / head as separate kernels, each with its own register budget.)
// Every workgroup (512 threads, 244+ VGPRs worth of registers via launch bounds + a big LDS carve: one per CU, as the ring roles) records
// the XCC it runs on and its arrival time, then spins until EVERY workgroup of EVERY kernel has arrived (bounded), i.e. the grid is only
// complete if all three kernels are resident at once.  Reported: did all arrive, how long after the first, and block -> XCD per kernel.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_coresident.bin scripts/ubench_coresident.hip && scripts/ubench_coresident.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t err__ = (e); if (err__ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(err__)); exit(1); } } while (0)

struct Rec { unsigned xcc, cu; unsigned long long t_arrive, t_all; };

template <int ROLE>
__global__ void __launch_bounds__(512) role_kernel(Rec* rec, unsigned* counter, int total, int base) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        Rec& r = rec[base + blockIdx.x];
        r.xcc = x & 0xf; r.cu = hw;
        r.t_arrive = __builtin_amdgcn_s_memrealtime();
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)total && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(8);
        r.t_all = spins < (1u << 24) ? __builtin_amdgcn_s_memrealtime() : 0ull;
        smem[0] = (float)ROLE;                                  // (keeps the LDS carve alive)
    }
}

int main() {
    const int n_stage = 8 * 24, n_head = 8, n_tap = 24;          // egs/mol at 8 utterances: 192 stages, 8 heads, 24 tap workgroups = 224 CUs
    const int total = n_stage + n_head + n_tap;
    Rec* d_rec; unsigned* d_cnt;
    CK(hipMalloc(&d_rec, total * sizeof(Rec)));
    CK(hipMalloc(&d_cnt, sizeof(unsigned)));
    hipStream_t s[3];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const size_t lds = 96 * 1024;                                // > half of a CU's 160 KB: one workgroup per CU whatever the registers
    CK(hipFuncSetAttribute((const void*)role_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)role_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)role_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int trial = 0; trial < 6; ++trial) {
        CK(hipMemset(d_rec, 0, total * sizeof(Rec)));
        CK(hipMemset(d_cnt, 0, sizeof(unsigned)));
        CK(hipDeviceSynchronize());
        // trials 0-2: stages first; 3-5: heads and taps first (does the launch order decide the placement?)
        if (trial < 3) {
            hipLaunchKernelGGL(role_kernel<0>, dim3(n_stage), dim3(512), lds, s[0], d_rec, d_cnt, total, 0);
            hipLaunchKernelGGL(role_kernel<1>, dim3(n_head), dim3(512), lds, s[1], d_rec, d_cnt, total, n_stage);
            hipLaunchKernelGGL(role_kernel<2>, dim3(n_tap), dim3(512), lds, s[2], d_rec, d_cnt, total, n_stage + n_head);
        } else {
            hipLaunchKernelGGL(role_kernel<1>, dim3(n_head), dim3(512), lds, s[1], d_rec, d_cnt, total, n_stage);
            hipLaunchKernelGGL(role_kernel<2>, dim3(n_tap), dim3(512), lds, s[2], d_rec, d_cnt, total, n_stage + n_head);
            hipLaunchKernelGGL(role_kernel<0>, dim3(n_stage), dim3(512), lds, s[0], d_rec, d_cnt, total, 0);
        }
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        std::vector<Rec> r(total);
        CK(hipMemcpy(r.data(), d_rec, total * sizeof(Rec), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; int complete = 0;
        for (auto& x : r) { if (x.t_arrive < t0) t0 = x.t_arrive; if (x.t_arrive > t1) t1 = x.t_arrive; complete += x.t_all != 0; }
        int stage_ok = 0, head_same = 0;
        for (int b = 0; b < n_stage; ++b) stage_ok += r[b].xcc == r[b % 8].xcc;
        for (int h = 0; h < n_head; ++h) head_same += r[n_stage + h].xcc == r[h].xcc;   // head of ring h on the XCD of ring h's stages (blocks h, h + 8, ...)?
        int per_xcd[3][16] = {};
        for (int b = 0; b < total; ++b) per_xcd[b < n_stage ? 0 : b < n_stage + n_head ? 1 : 2][r[b].xcc]++;
        printf("trial %d (%s first): %d of %d workgroups saw the whole grid resident; arrivals spread over %.1f us; stage block b on the XCD of block b %% 8: %d of %d; "
               "head h on ring h's XCD: %d of 8\n", trial, trial < 3 ? "stages" : "heads + taps", complete, total, (double)(t1 - t0) * 0.01, stage_ok, n_stage, head_same);
        for (int k = 0; k < 3; ++k) {
            printf("   %s per XCD:", k == 0 ? "stages" : k == 1 ? "heads " : "taps  ");
            for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[k][x]);
            printf("\n");
        }
        printf("   head XCDs:");
        for (int h = 0; h < n_head; ++h) printf(" %u", r[n_stage + h].xcc);
        printf("   | stage blocks 0-7 XCDs:");
        for (int b = 0; b < 8; ++b) printf(" %u", r[b].xcc);
        printf("\n");
    }
    return 0;
}
