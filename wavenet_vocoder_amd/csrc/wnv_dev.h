// wnv_dev.h -- structures shared by the host packer and the gfx950 kernels, plus small device helpers.
//
// Weight blob layout ("K-major"): every 1x1 / dilated convolution is stored transposed, [K inputs][N outputs]
// with N padded to a multiple of 4 floats, so that one wave reads a 1 KiB contiguous row slice per
// global_load_dwordx4 and a lane owns 4 consecutive outputs.  The reference stores (out, in, kw) and
// linearises it to (out, kw*in) (conv.py:51-62); row k*Cin + i of our layout holds W[:, i, k].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WNV_MAX_LAYERS 128

struct WnvLayerDev {
    int dilation;
    int ring_rows;        // (kw - 1) * dilation history rows kept for this layer
    long long ring_off;   // float offset of this layer's ring inside one utterance's ring block
    long long w_in;       // [kw*R + cin][Gp]   dilated conv taps (oldest first) then local-conditioning 1x1
    long long b_in;       // [Gp]
    long long w_g;        // [gin][Gp] global-conditioning 1x1 (bias-free) or -1
    long long w_os;       // [G/2][NOSp]  columns [0,R) = conv1x1_out, [R,R+K) = conv1x1_skip
    long long b_os;       // [NOSp]
};

struct WnvModelDev {
    int L, R, G, K, O, kw, cin, gin, cin1;   // cin1: first_conv input channels (1 or O)
    int scalar_input, dist;
    int Rp, Gp, NOSp, Kp, Op;                // padded widths (multiples of 4)
    int nr_mix;
    long long w_first, b_first;              // [cin1][Rp], [Rp]
    long long w_h1, b_h1;                    // [K][Kp]
    long long w_h2, b_h2;                    // [K][Op]
    long long ring_floats;                   // per-utterance ring block size
    float skip_scale;                        // sqrt(1 / L)  (wavenet.py:313)
    // LDS carve (floats)
    int lds_xin, lds_u, lds_o, lds_vin, lds_taps, lds_nz, lds_part_stride;
    int taps_in_lds;
    int n_embed;                             // rows of embed_speakers.weight (0 = none): speaker ids are clamped into it
};

struct WnvGenArgs {
    int B;
    long long T, Tt;
    const float* c_up;       // (B, T, cin)
    const float* initial;    // (B, cin1) or null
    const float* teacher;    // (B, Tt, cin1) or null
    const float* noise;      // (T, B, nz) or null
    const float* zbias;      // effective conv bias: b_in (+ Wg.g) ; (B or 1, L, Gp)
    long long zbias_bstride;
    float* ring;             // (B, ring_floats) zeroed before launch
    unsigned long long seed;
    int softmax, quantize, nz;
    int async;               // ring kernel: do not synchronise after the launch (WNV_GEN_ASYNC)
    float* out;              // (B, C, T)
    float* params_out;       // (B, O, T) or null
    int* index_out;          // (B, T) or null
    const unsigned* noise_ready = nullptr;   // streamed tape: steps [0, *noise_ready) are valid (coherent host memory); ring kernel only
    const int *seg_start = nullptr, *seg_uid = nullptr;   // packed slots (wnv_generate_args): (B, T) each; ring kernel only
    const int* seg_gid = nullptr;                         // packed slots + global conditioning: (B, T) row of zbias (which then has n_g rows, not B)
    int b0 = 0, noise_B = 0; // a slice [b0, b0 + B) of a larger call (host-side chunking): the utterance index the noise tape /
                             // Philox stream is addressed with is b0 + b, the tape's batch stride is noise_B (0: B)
};

// ---------------------------------------------------------------------------------------------
// counter-based RNG for the tape-less mode (Philox4x32-10)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wnv_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform in (0, 1), EXACT in float32 (round 6): (x >> 9) + 0.5 <= 2^23 - 0.5 needs 24 bits, the scale is a power of two -- no rounding
// anywhere, never 0, never 1 (largest value 1 - 2^-24).  Exp(1) = -log u is therefore strictly positive and every form of the categorical
// pick (argmax x_k / e_k, argmax logit_k - log e_k) is defined for every draw.  (Until round 5 the map was ((x >> 8) + 0.5) / 2^24, whose sum
// rounds to 2^24 once in 2^24 draws: u = 1.0, e = -0.0, and the two pick forms parted there -- a one-hot waveform depended on the batch size.)
__device__ __forceinline__ float wnv_u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// noise value j of (utterance b, step t), same semantics as the tape (wavenet_vocoder_amd/noise.py)
// kind: 0 = U(1e-5, 1-1e-5), 1 = N(0,1), 2 = Exp(1)
__device__ __forceinline__ float wnv_noise_gen(unsigned long long seed, long long t, int b, int j, int kind) {
    uint32_t r[4];
    wnv_philox((uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)b, (uint32_t)j,
               (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u = wnv_u01(r[0]);
    if (kind == 0) return 1e-5f + u * (1.0f - 2e-5f);
    if (kind == 2) return -logf(u);
    const float v = wnv_u01(r[1]);
    return sqrtf(-2.0f * logf(u)) * cosf(6.28318530717958647692f * v);
}

__device__ __forceinline__ float wnv_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
