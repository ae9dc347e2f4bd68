// wnv_sample.h -- wave-level reductions and the three samplers, shared by the generic and the ring kernels.
// Reference: mixture.py:118-156 (MoL), mixture.py:221-270 (Gaussian), wavenet.py:332-335 (categorical).
#pragma once
#include <hip/hip_runtime.h>

namespace {
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// argmax with first-index tie break (torch.max / argmax semantics on CPU)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// ---- sampling (wave 0 only) ----------------------------------------------------------------------
// scalar outputs: mixture of logistics / Gaussians.  obuf = head output [O]; nzv = this step's noise.
__device__ __forceinline__ float sample_scalar(int dist, int O, const float* obuf, const float* nzv,
                                               int lane) {
    float mean, ls;
    int nmix = 0;
    if (dist == 2 && O == 2) { mean = obuf[0]; ls = obuf[1]; }                // mixture.py:258-259
    else if (dist == 2 && O == 3) { mean = obuf[1]; ls = obuf[2]; }           // mixture.py:260-261
    else {
        nmix = O / 3;
        // Gumbel-max over the mixture logits (mixture.py:138-140 / :247-249)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < nmix; i += 64) {
            const float v = obuf[i] - logf(-logf(nzv[i]));
            if (v > best) { best = v; bi = i; }
        }
        wave_argmax(best, bi);
        mean = obuf[nmix + bi];                                                 // mixture.py:143-146
        ls = obuf[2 * nmix + bi];
    }
    const float r = nzv[nmix];
    float x;
    if (dist == 1) x = mean + expf(ls) * (logf(r) - logf(1.0f - r));          // mixture.py:151-152
    else x = r * expf(ls) + mean;                                               // mixture.py:265-267
    return fminf(fmaxf(x, -1.0f), 1.0f);                                        // mixture.py:154 / :269
}

// categorical outputs.  Turns obuf into probabilities in place (when softmax) and returns the sampled
// class (when quantize), else -1.
__device__ __forceinline__ int sample_categorical(int O, float* obuf, const float* nzv,
                                                  int softmax, int quantize, int lane) {
    if (softmax) {                                                              // wavenet.py:332
        float mx = -INFINITY;
        for (int n = lane; n < O; n += 64) mx = fmaxf(mx, obuf[n]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int n = lane; n < O; n += 64) { const float e = expf(obuf[n] - mx); obuf[n] = e; s += e; }
        s = wave_sum(s);
        for (int n = lane; n < O; n += 64) obuf[n] = obuf[n] / s;
    }
    if (!quantize) return -1;
    // OneHotCategorical(p).sample(): Categorical renormalises, multinomial takes argmax(p_hat / e)
    float s2 = 0.f;
    for (int n = lane; n < O; n += 64) s2 += obuf[n];
    s2 = wave_sum(s2);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = lane; n < O; n += 64) {
        const float q = (obuf[n] / s2) / nzv[n];
        if (q > best) { best = q; bi = n; }
    }
    wave_argmax(best, bi);
    return bi;
}

}  // namespace
