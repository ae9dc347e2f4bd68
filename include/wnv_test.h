/* wnv_test.h -- TEST / BENCH hooks.  NOT part of the product ABI (include/wnv.h) and not exported by the product library:
 * these two entry points exist only in wavenet_vocoder_amd/libwnv_test.so -- the same sources compiled with -DWNV_TEST_HOOKS -DWNV_KNOBS
 * (python -m wavenet_vocoder_amd.build builds both libraries) --, which is also the only build that reads the WNV_* measurement
 * knobs from the environment (csrc/wnv_knobs.h).  Users: tests/test_gpu_zz_boundary.py (time-out policy), the variant tests of
 * tests/test_gpu_ring.py, bench.py (roofline.peak_measured), scripts/. */
#ifndef WNV_TEST_H_
#define WNV_TEST_H_
#include "wnv.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The next n persistent (ring) launches this handle would make in auto mode (kernel = 0) report WNV_ERR_TIMEOUT without being
 * launched -- drives the retry policy described at wnv_reset without a device that loses its CUs.  n = 0 clears. */
WNV_API wnv_status wnv_debug_inject_timeouts(wnv_handle h, int32_t n);

/* The MEASURED on-chip peak the sample loop's roofline is priced against: LDS read bandwidth of the whole device in
 * GB/s from a microbenchmark launch (every CU: 16 waves of conflict-free ds_read_b128; csrc/wnv_ubench.hip), ~5 ms.  *n_cu (optional)
 * receives the CU count the figure covers.  Synchronous; needs a GPU. */
WNV_API wnv_status wnv_measure_lds_read_peak(int32_t device, double* gb_per_s, int32_t* n_cu);

#ifdef __cplusplus
}
#endif
#endif /* WNV_TEST_H_ */
