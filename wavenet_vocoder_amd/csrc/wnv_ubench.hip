// wnv_ubench.hip -- the MEASURED on-chip peak the roofline of the sample-loop kernel is priced against ("use the
// measured LDS read peak from a microbenchmark on the box ... confirm 128 vs 256 B/clk empirically and state which was used").
//
// wnv_measure_lds_read_peak: every CU runs 16 waves that do nothing but conflict-free ds_read_b128 (64 lanes x 16 B = 1 KiB per
// instruction, eight in flight per wave, destinations never consumed: inline assembly the compiler can neither merge nor drop);
// bytes read / HIP-event time of the launch = the chip's LDS read bandwidth.  bench.py calls it once per run and reports it as
// roofline.peak_measured next to the nominal 256 CU x 256 B/clk x 2.4 GHz.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "wnv_hostutil.h"
#include "../../include/wnv_test.h"      // (the declaration carries WNV_API: the library is built with -fvisibility=hidden)

namespace {

constexpr int UT = 1024;           // threads per workgroup: 16 waves = 4 per SIMD
constexpr int READS = 8;           // ds_read_b128 in flight per wave

__global__ void __launch_bounds__(UT) wnv_lds_read_kernel(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[UT * 4 * 2];          // 32 KiB
    for (int i = threadIdx.x; i < UT * 4 * 2; i += UT) lds[i] = (float)i;
    __syncthreads();
    // lane address: 16 consecutive bytes per lane (the conflict-free pattern of a b128 read); the eight reads of a batch are 2 KiB apart
    const unsigned addr = (unsigned)(size_t)(lds) + (unsigned)threadIdx.x * 16u;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f4 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile(
            "ds_read_b128 %0, %8\n\t"
            "ds_read_b128 %1, %8 offset:2048\n\t"
            "ds_read_b128 %2, %8 offset:4096\n\t"
            "ds_read_b128 %3, %8 offset:6144\n\t"
            "ds_read_b128 %4, %8 offset:8192\n\t"
            "ds_read_b128 %5, %8 offset:10240\n\t"
            "ds_read_b128 %6, %8 offset:12288\n\t"
            "ds_read_b128 %7, %8 offset:14336\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
            : "v"(addr)
            : "memory");
        acc.x += v0.x;                                                       // one VALU op per 8 KiB read: the reads are the work
    }
    if (acc.x == -1.2345f) sink[blockIdx.x] = acc.x + acc.y;                // (never true: keeps the loop alive)
}

}  // namespace

// include/wnv.h
extern "C" wnv_status wnv_measure_lds_read_peak(int32_t device, double* gb_per_s, int32_t* n_cu) {
    if (!gb_per_s) return fail(WNV_ERR_INVALID_ARG, "wnv_measure_lds_read_peak: NULL result pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(WNV_ERR_INVALID_ARG, "wnv_measure_lds_read_peak: no such device %d", device);
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(WNV_ERR_HIP, "hipGetDeviceProperties failed");
    const int ncu = prop.multiProcessorCount;
    if (n_cu) *n_cu = ncu;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s = nullptr;
    wnv_status rc = WNV_OK;
    double best = 0.0;
#define UB_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { rc = fail(WNV_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__)); goto done; } } while (0)
    UB_TRY(hipMalloc((void**)&sink, (size_t)ncu * 4 * sizeof(float)));
    UB_TRY(hipStreamCreate(&s));
    UB_TRY(hipEventCreate(&e0));
    UB_TRY(hipEventCreate(&e1));
    {
        const int grid = ncu * 2, iters = 4000;                             // two rounds of workgroups per CU, ~0.5 ms per launch
        hipLaunchKernelGGL(wnv_lds_read_kernel, dim3(grid), dim3(UT), 0, s, sink, 64);      // warm-up (code object, clocks)
        UB_TRY(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            UB_TRY(hipEventRecord(e0, s));
            hipLaunchKernelGGL(wnv_lds_read_kernel, dim3(grid), dim3(UT), 0, s, sink, iters);
            UB_TRY(hipEventRecord(e1, s));
            UB_TRY(hipEventSynchronize(e1));
            float ms = 0.f;
            UB_TRY(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)grid * UT * 16.0 * READS * iters;
            best = std::max(best, bytes / (ms * 1e-3) / 1e9);
        }
    }
    *gb_per_s = best;
done:
#undef UB_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (sink) (void)hipFree(sink);
    return rc;
}
