// wnv_sample.h -- wave-level reductions and the three samplers, shared by the generic and the ring kernels.
// Reference: mixture.py:118-156 (MoL), mixture.py:221-270 (Gaussian), wavenet.py:332-335 (categorical).
#pragma once
#include <hip/hip_runtime.h>

namespace {
// Wave-wide all-reduces without the LDS crossbar (round 3): four DPP steps inside a row of 16 lanes (quad_perm x 2, row_half_mirror,
// row_mirror), then v_permlane16_swap / v_permlane32_swap across the rows -- six VALU instructions instead of six ds_bpermute round
// trips (~0.1 us each): the 256-way softmax + multinomial of a mu-law step does four of these in a row (wavenet.py:332-335).
template <int CTRL> __device__ __forceinline__ float wnv_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int wnv_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// the value of the lane 16 / 32 away (rows 0 <-> 1, 2 <-> 3;  halves 0 <-> 1)
__device__ __forceinline__ float wnv_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);        // (r[0]: odd rows hold the even rows' values; r[1]: even rows hold the odd rows')
}
__device__ __forceinline__ float wnv_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += wnv_dpp<0xB1>(v); v += wnv_dpp<0x4E>(v); v += wnv_dpp<0x141>(v); v += wnv_dpp<0x140>(v);
    v += wnv_xor16(v);
    return v + wnv_xor32(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, wnv_dpp<0xB1>(v)); v = fmaxf(v, wnv_dpp<0x4E>(v)); v = fmaxf(v, wnv_dpp<0x141>(v)); v = fmaxf(v, wnv_dpp<0x140>(v));
    v = fmaxf(v, wnv_xor16(v));
    return fmaxf(v, wnv_xor32(v));
}
// argmax with first-index tie break (torch.max / argmax semantics on CPU): the wave maximum, then the smallest index that attains it
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
    const float m = wave_max(v);
    int c = v == m ? i : 0x7fffffff;
    c = min(c, wnv_dpp_i<0xB1>(c)); c = min(c, wnv_dpp_i<0x4E>(c)); c = min(c, wnv_dpp_i<0x141>(c)); c = min(c, wnv_dpp_i<0x140>(c));
    c = min(c, __float_as_int(wnv_xor16(__int_as_float(c))));
    c = min(c, __float_as_int(wnv_xor32(__int_as_float(c))));
    v = m; i = c;
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

// categorical outputs.  Turns obuf into probabilities in place (when softmax and not quantize) and returns the sampled class (when
// quantize), else -1.
// THE CHOICE (wavenet.py:332-335): F.softmax, then OneHotCategorical(probs).sample() -- Categorical renormalises, torch.multinomial takes
// argmax(p_hat / e), e ~ Exp(1).  With x_k = exp(logit_k - max) that is argmax_k (x_k / s / s2) / e_k, where s and s2 are the
// two normalising sums: the SAME positive factors for every class, which an argmax does not see.  The kernels therefore take
// argmax_k x_k / e_k (round 4; until then they carried out both normalisations: two wave-wide sums and eight more divisions per lane, 1.35 us
// of a mu-law step in one wave).  Dropping the common factors moves the pick only where the top-2 margin is below the rounding of the
// quotients (~2e-7 in the log domain; tests/_margins.py admits a flip below 1e-6 + twice the head-output difference).  Every kernel
// (generic, ring, wide) uses this arithmetic -- the ring's categorical head with one class per lane --, so they agree on a class
// whenever they agree on the logits.  Ties: the smallest index (torch.argmax on CPU).
__device__ __forceinline__ int sample_categorical(int O, float* obuf, const float* nzv,
                                                  int softmax, int quantize, int lane) {
    // up to 256 classes: a lane keeps its (at most four) classes n = lane + 64 k in registers -- one LDS read of the logits and the noise
    if (O <= 256) {
        float x[4], e[4];
        bool on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            on[k] = lane + 64 * k < O;
            x[k] = on[k] ? obuf[lane + 64 * k] : -INFINITY;
            e[k] = (on[k] && quantize) ? nzv[lane + 64 * k] : 1.f;
        }
        if (softmax) {                                                          // wavenet.py:332
            float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
            mx = wave_max(mx);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = on[k] ? expf(x[k] - mx) : 0.f;
            if (!quantize) {                                                    // (tests only: the probabilities themselves are the output)
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s += x[k];
                s = wave_sum(s);
#pragma unroll
                for (int k = 0; k < 4; ++k) if (on[k]) obuf[lane + 64 * k] = x[k] / s;
            }
        }
        if (!quantize) return -1;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float q = x[k] / e[k];
            if (on[k] && q > best) { best = q; bi = lane + 64 * k; }
        }
        wave_argmax(best, bi);
        return bi;
    }
    if (softmax) {                                                              // wavenet.py:332
        float mx = -INFINITY;
        for (int n = lane; n < O; n += 64) mx = fmaxf(mx, obuf[n]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int n = lane; n < O; n += 64) { const float e = expf(obuf[n] - mx); obuf[n] = e; s += e; }
        if (!quantize) {
            s = wave_sum(s);
            for (int n = lane; n < O; n += 64) obuf[n] = obuf[n] / s;
        }
    }
    if (!quantize) return -1;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = lane; n < O; n += 64) {
        const float q = obuf[n] / nzv[n];
        if (q > best) { best = q; bi = n; }
    }
    wave_argmax(best, bi);
    return bi;
}

}  // namespace
