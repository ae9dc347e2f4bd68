// wnv_post.hip -- the post-chain of synthesis.batch_wavegen / evaluate.py on the device.
//
//   y_hat (B, C, T) from wnv_generate
//     -> input_type "mulaw-quantize": argmax over C, inv_mulaw_quantize   (synthesis.py:68-70)
//        input_type "mulaw"         : inv_mulaw                            (synthesis.py:72-74)
//        input_type "raw"           : as is                                (synthesis.py:76)
//     -> optional postprocess = audio.inv_preemphasis(x, coef)             (synthesis.py:78-80, audio.py:57-58)
//     -> optional / global_gain_scale                                      (synthesis.py:82-84)
//     -> optional clip to [-1, 1] and int16 = (x * 32767) truncated        (evaluate.py:238, :43-48)
//
// inv_mulaw / inv_mulaw_quantize / inv_preemphasis live in the reference's unpinned, un-vendored dependency nnmnkwii
// (>= 0.0.11, setup.py:23); what is implemented is its published definition:
//   inv_mulaw(y, mu)          = sign(y) / mu * ((1 + mu)^|y| - 1)
//   inv_mulaw_quantize(i, mu) = inv_mulaw(2 i / mu - 1, mu)
//   inv_preemphasis(x, c)     = lfilter([1], [1, -c], x):   y[n] = x[n] + c y[n-1]
// One workgroup per utterance; the first-order recurrence is evaluated as 256 local segments + a carry pass
// (y = y_local + carry_in * c^(i+1)), which reassociates the sums (|difference| ~ 1e-7 relative).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/wnv.h"
#include "wnv_devguard.h"

namespace {
constexpr int PT = 256;

__device__ __forceinline__ float inv_mulaw(float y, float mu) {
    const float m = (powf(1.0f + mu, fabsf(y)) - 1.0f) / mu;
    return copysignf(m, y);
}

__global__ void __launch_bounds__(PT) wnv_post_kernel(const float* __restrict__ y, int C, long long T, int input_type, float mu,
                                                      float coef, float gain, int clip, float* __restrict__ wav,
                                                      short* __restrict__ pcm) {
    __shared__ float seg_end[PT];
    __shared__ float carry[PT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long seg = (T + PT - 1) / PT;
    const long long n0 = (long long)tid * seg, n1 = n0 + seg < T ? n0 + seg : T;
    const float* yb = y + (size_t)b * C * T;
    float* wb = wav + (size_t)b * T;
    // pass 1: decode + local recurrence with zero state
    float acc = 0.f;
    for (long long n = n0; n < n1; ++n) {
        float x;
        if (input_type == 2 && C == 1) {             // (ABI 5) y holds the sampled class itself (wnv_generate_args.index_out as floats)
            x = inv_mulaw(2.0f * yb[n] / mu - 1.0f, mu);
        } else if (input_type == 2) {                // argmax over the one-hot / probability axis, first maximum wins
            int best = 0;
            float bv = yb[n];
            for (int c = 1; c < C; ++c) {
                const float v = yb[(size_t)c * T + n];
                if (v > bv) { bv = v; best = c; }
            }
            x = inv_mulaw(2.0f * (float)best / mu - 1.0f, mu);
        } else if (input_type == 1) {
            x = inv_mulaw(yb[n], mu);
        } else {
            x = yb[n];
        }
        acc = coef > 0.f ? x + coef * acc : x;
        wb[n] = acc;
    }
    if (coef > 0.f) {
        seg_end[tid] = n1 > n0 ? acc : 0.f;
        __syncthreads();
        if (tid == 0) {                              // carry into segment j = true y at the end of segment j-1
            float cl = 1.f;
            for (long long i = 0; i < seg; ++i) cl *= coef;         // coef^seg (every segment but the last is full)
            float cr = 0.f;
            for (int j = 0; j < PT; ++j) {
                carry[j] = cr;
                const long long a = (long long)j * seg, e = a + seg < T ? a + seg : T;
                if (e > a) {
                    float cj = cl;
                    if (e - a != seg) { cj = 1.f; for (long long i = 0; i < e - a; ++i) cj *= coef; }
                    cr = seg_end[j] + cj * cr;
                }
            }
        }
        __syncthreads();
        float w = carry[tid];
        for (long long n = n0; n < n1; ++n) { w *= coef; wb[n] += w; }
    }
    // pass 2: gain, clip, int16
    for (long long n = n0; n < n1; ++n) {
        float v = wb[n];
        if (gain > 0.f) v /= gain;
        if (clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
        wb[n] = v;
        if (pcm) pcm[(size_t)b * T + n] = (short)(v * 32767.0f);       // numpy astype(int16): truncation toward zero
    }
}
}  // namespace

extern thread_local std::string wnv_g_err;

extern "C" wnv_status wnv_postprocess(int32_t device, const wnv_post_args* a) {
    if (!a || !a->y || !a->wav || a->B <= 0 || a->T <= 0 || a->C <= 0) { wnv_g_err = "wnv_postprocess: bad arguments"; return WNV_ERR_INVALID_ARG; }
    if (a->input_type < 0 || a->input_type > 2 || (a->input_type != 0 && a->mu <= 0)) { wnv_g_err = "wnv_postprocess: bad input_type / mu"; return WNV_ERR_INVALID_ARG; }
    if (a->input_type != 2 && a->C != 1) { wnv_g_err = "wnv_postprocess: scalar input types take C == 1"; return WNV_ERR_INVALID_ARG; }
    if (device < 0) { wnv_g_err = "wnv_postprocess: device must be >= 0"; return WNV_ERR_INVALID_ARG; }
    DeviceGuard guard(device);                         // the caller's current device is restored on return
    if (!guard.ok) { wnv_g_err = "wnv_postprocess: cannot select the device"; return WNV_ERR_HIP; }
    hipError_t e;
    hipLaunchKernelGGL(wnv_post_kernel, dim3(a->B), dim3(PT), 0, (hipStream_t)a->stream, a->y, a->C, (long long)a->T, a->input_type,
                       (float)a->mu, a->preemphasis, a->gain_scale, a->clip, a->wav, (short*)a->pcm);
    e = hipGetLastError();
    if (e != hipSuccess) { wnv_g_err = std::string("wnv_post_kernel launch failed: ") + hipGetErrorString(e); return WNV_ERR_HIP; }
    return WNV_OK;
}
