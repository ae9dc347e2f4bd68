// wnv_wide.hip -- the sample loop for WIDE models (residual_channels <= 512, gate_channels <= 512, skip_out_channels <= 256):
// the reference's own default constructor geometry (wavenet.py:98-101) and its published 512 / 512 / 256 models, for which one
// layer (4.1 MB of weights) is eight times what one CU can hold, so the one-layer-per-CU ring kernel (wnv_ring.hip) does not
// apply and the generic kernel has to stream 99 MB of weights per sample.
//
// GROUP RING.  Every layer gets a GROUP of 8 workgroups on one XCD, one per CU, each owning a slice of the layer's OUTPUT
// channels with its slice of the weights resident in registers:
//
//     head -> group 0 -> group 1 -> ... -> group L-1 -> head        (3-4 groups per XCD, consecutive layers share an XCD)
//
// ONE hop and ONE mat-vec phase per layer.  Group l-1 publishes the PAIR (u_{l-1}, h_{l-1}) -- its gate outputs and its own layer
// input; from that pair alone CU j of group l computes, in one pass over its registers (modules.py:127-163 with the residual
// recurrence substituted into the next pre-activation, folded on the host in double as in wnv_ring.hip):
//     z_l   = M_l u_{l-1} + N_l h_{l-1} + c_l + pre_l        64 gate rows;  M_l = sqrt(.5) W_cur,l W_out,l-1, N_l = sqrt(.5) W_cur,l,
//     u_l   = tanh . sigmoid (z_l)                           c_l = N_l b_out,l-1                        -> publish 32 values
//     h_l   = sqrt(.5) (W_out,l-1 u_{l-1} + b + h_{l-1})     64 rows: layer l's input, the reference's own recurrence -> publish
//     s_{l-1} = W_skip,l-1 u_{l-1} + b                       32 skip channels, summed from group to group (CU j -> CU j)
//   (group 0 reads h_0 from the head: z_0 = W_cur,0 h_0 + pre_0, and passes h_0 on; the last group gathers its own u for the last
//   skip term, one more hop once per step).  Behind the chain every CU gathers the full h_l its group just produced, keeps its OWN
//   copy of the layer's input history (no cross-CU ordering needed) and streams the older taps + the local-conditioning 1x1 of the
//   NEXT step -- pre_j[t+1], 282 KB of weights per CU -- from L2 / Infinity Cache, once per step for all utterances.
//   Hand-off = the ring kernel's data-tagged 8-byte granules; plain stores inside an XCD (the host's placement census verified that
//   blocks b and b % 8 share an XCD), write-through stores across XCDs.  Mat-vec mapping: lane = output row, the 8 waves split K,
//   partial sums meet in LDS.  (v1 of this kernel evaluated the layer unfolded, two hops + two phases per layer: 57 us per step.)
//
// Models narrower than 512 / 512 / 256 are zero-padded (exact).  Scalar-input models (MoL / Gaussian, out_channels <= 64);
// utterances beyond the first share the groups like a systolic array (B <= 8).  Every wait is bounded (WNV_ERR_TIMEOUT).
#include "wnv_wide.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wnv_sample.h"

namespace {

constexpr int WT = 512;            // threads per workgroup
constexpr int RWD = 512;           // padded residual channels
constexpr int GHD = 256;           // padded gate channels (gate rows = 2 GHD)
constexpr int KWD = 256;           // padded skip channels
constexpr int PG = 8;              // workgroups per layer group
constexpr int GS = GHD / PG;       // 32 gate channels per workgroup (64 gate rows = one per lane)
constexpr int RS = RWD / PG;       // 64 residual rows per workgroup
constexpr int KS = KWD / PG;       // 32 skip rows per workgroup
constexpr int BMAX = 8;
constexpr int XW = GHD + RWD;      // one mailbox slot: 256 gate outputs then 512 layer inputs
constexpr unsigned SPIN_LIMIT = 1u << 22;

using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

struct WideParams {
    int L, nL, B, T, Tt, O, cin, cinp, kw, nz, dist, kpre, nkb;
    int head_x, head_li, fast;
    int cin1, softmax, quantize;                          // first_conv input channels (1 = scalar input); categorical switches (wavenet.py:332-335)
    int* index_out;
    u64* hidmail;                                         // one-hot models: hidden layer of the head, HID[b][256] (head part A -> part B)
    unsigned tag_base;
    float skip_scale;
    const float *wm, *wn, *wo, *ws, *wsl, *bo, *bs, *cvec, *wpre;   // per (layer, slice) register / stream images (wsl: last layer's skip)
    const float *wh1, *bh1, *wh2, *bh2, *wfirst, *bfirst;
    const float* zbias;                                   // generic pack: [B or 1][L][zb_ld], rows = the model's gate rows
    long long zbias_bstride;
    int zb_ld, gh_model;
    const int *lay_dil, *lay_histoff;                     // dilation; float offset of the layer's history inside one copy set
    long long hist_b_floats;                              // floats of history per utterance (all layers, all 8 copies)
    u64 *xmail, *smail;                                   // X[b][L+1][768] = (u_{l-1} | h_{l-1}) for group l (slot L: the last group's own
                                                          // outputs); SK[b][L+2][256] running skip sums
    float* hist;
    const float *c_up, *initial, *teacher, *noise;
    u64 seed;
    float *out, *params_out;
    unsigned int* status;
};

__device__ __forceinline__ void st_granule(u64* p, unsigned tag, float v, bool fast) {
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
// one wave receives 128 consecutive granules (two per lane, one 16-byte L1-bypassing load) into dst[0 .. 128)
__device__ __forceinline__ bool recv128(const u64* g, unsigned tag, float* dst, unsigned int* status, unsigned code, int lane) {
    const u64* g2 = g + 2 * lane;
    for (unsigned spins = 0;;) {
        u4v x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(g2) : "memory");
        if (__all(x.y == tag && x.w == tag)) {
            *reinterpret_cast<float2*>(dst + 2 * lane) = make_float2(__uint_as_float(x.x), __uint_as_float(x.z));
            return true;
        }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}
// the first `n` lanes of a wave each wait for one granule
__device__ __forceinline__ bool recv_lanes(const u64* g, int n, unsigned tag, float& v, unsigned int* status, unsigned code, int lane) {
    for (unsigned spins = 0;;) {
        bool ok = true;
        if (lane < n) {
            const u64 x = __hip_atomic_load(g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __uint_as_float((unsigned)x);
            ok = (unsigned)(x >> 32) == tag;
        }
        if (__all(ok)) return true;
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}
__device__ __forceinline__ float wide_gate(float a, float g) {                 // tanh(a) sigmoid(g), hardware exp2 / rcp (as wnv_ring.hip)
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -2.8853900817779268f);
    const float f = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}
// dot product of NF4 float4s of weights (registers) with the same span of an LDS vector that every lane reads alike (broadcast)
template <int NF4>
__device__ __forceinline__ float dot_bcast(const float4 (&w)[NF4], const float* x) {
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NF4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(x)[c];
        a0 = __builtin_elementwise_fma(f2{w[c].x, w[c].y}, f2{v.x, v.y}, a0);
        a1 = __builtin_elementwise_fma(f2{w[c].z, w[c].w}, f2{v.z, v.w}, a1);
    }
    a0 += a1;
    return a0.x + a0.y;
}
template <int NF4>
__device__ __forceinline__ void load_img(float4 (&w)[NF4], const float* img, int wave, int lane) {     // [wave][c][lane][4]
    const float4* src = reinterpret_cast<const float4*>(img) + (size_t)wave * NF4 * 64 + lane;
#pragma unroll
    for (int c = 0; c < NF4; ++c) w[c] = src[(size_t)c * 64];
}

struct StageLds {
    float *hx, *ux, *pz, *po, *ps, *pre, *xin, *pt;
    int* flags;
};
__device__ __forceinline__ StageLds carve_stage(float* smem, int kpre) {
    StageLds s;
    s.hx = smem;                           // [512] h_{l-1}[t], then (behind the chain) the full h_l[t]
    s.ux = s.hx + RWD;                     // [256] gate outputs u_{l-1}[t] (the last group: then its own u)
    s.pz = s.ux + GHD;                     // [8][64] partial z
    s.po = s.pz + 8 * 64;                  // [8][64] partial conv1x1_out
    s.ps = s.po + 8 * 64;                  // [16][32] partial conv1x1_skip
    s.pre = s.ps + 16 * 32;                // [BMAX][64] pre_j of the step being computed
    s.pt = s.pre + BMAX * 64;              // [8][BMAX][64] partial taps
    s.flags = reinterpret_cast<int*>(s.pt + 8 * BMAX * 64);
    s.xin = reinterpret_cast<float*>(s.flags + 16);      // [BMAX][kpre] tap inputs of the next step
    (void)kpre;
    return s;
}
__host__ __device__ inline size_t stage_lds_floats(int kpre) { return (size_t)RWD + GHD + 3 * 512 + BMAX * 64 + 8 * BMAX * 64 + 16 + (size_t)BMAX * kpre + 8 * 4 * 64 * 4; }

// pre_j[tp] for every utterance: older taps of step tp out of this workgroup's own history copy (zeros before t = 0), the
// conditioning row c[tp], then the streamed [kpre][64] matrix (one pass over the weights for all utterances)
__device__ void compute_pre(const WideParams& p, const StageLds& s, int l, int j, int tp, int tid, int lane, int wave) {
    const int d = p.lay_dil[l], rows = (p.kw - 1) * d, hoff = (p.kw - 1) * RWD;
    for (int b = 0; b < p.B; ++b) {
        const float* hist = p.hist + (size_t)b * p.hist_b_floats + (size_t)PG * p.lay_histoff[l] + (size_t)j * rows * RWD;
        float* xb = s.xin + (size_t)b * p.kpre;
        for (int e = tid; e < p.kpre; e += WT) {
            float v = 0.f;
            if (e < hoff) {
                const int k = e >> 9, r = e & (RWD - 1);
                const int tt = tp - (p.kw - 1 - k) * d;                     // conv.py:43-44: tap k (oldest first) looks d (kw-1-k) back
                if (tt >= 0) v = hist[(size_t)(tt % rows) * RWD + r];       // rows this very workgroup stored (row tp - 1 a barrier ago)
            } else if (e - hoff < p.cin) {
                v = p.c_up[((size_t)b * p.T + tp) * p.cin + (e - hoff)];
            }
            xb[e] = v;
        }
    }
    __syncthreads();
    // stream: wave w takes the float4 K-blocks [kb0, kb1); lane = gate row; one accumulator per utterance
    const int per = (p.nkb + 7) >> 3, kb0 = wave * per, kb1 = min(p.nkb, kb0 + per);
    const float4* W = reinterpret_cast<const float4*>(p.wpre) + ((size_t)(l * PG + j) * p.nkb) * 64 + lane;
    float acc[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) acc[b] = 0.f;
    for (int kb = kb0; kb < kb1; kb += 4) {
        float4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = kb + q < kb1 ? W[(size_t)(kb + q) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kb + q >= kb1) break;
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (b >= p.B) break;
                const float4 x = *reinterpret_cast<const float4*>(s.xin + (size_t)b * p.kpre + 4 * (kb + q));
                acc[b] = fmaf(w[q].x, x.x, acc[b]); acc[b] = fmaf(w[q].y, x.y, acc[b]);
                acc[b] = fmaf(w[q].z, x.z, acc[b]); acc[b] = fmaf(w[q].w, x.w, acc[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
        if (b < p.B) s.pt[((size_t)wave * BMAX + b) * 64 + lane] = acc[b];
    __syncthreads();
    for (int e = tid; e < p.B * 64; e += WT) {
        const int b = e >> 6, r = e & 63;
        // bias (+ W_g g, hoisted by the host): the model's gate row of padded row r of this slice
        const int half = r >> 5, ch = GS * j + (r & 31);
        float v = ch < p.gh_model ? p.zbias[(size_t)b * p.zbias_bstride + (size_t)l * p.zb_ld + (size_t)half * p.gh_model + ch] : 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s.pt[((size_t)w * BMAX + b) * 64 + r];
        s.pre[b * 64 + r] = v;
    }
    __syncthreads();
}

__device__ void run_wide_stage(const WideParams& p, int l, int j, bool fast_next, float* smem) {
    const StageLds s = carve_stage(smem, p.kpre);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = l == 0, last = l == p.L - 1;
    float4 wn[16], wm[8], wo[8], ws[4];
    load_img<16>(wn, p.wn + (size_t)(l * PG + j) * 8 * 16 * 64 * 4, wave, lane);      // N_l (group 0: W_cur,0) x h_{l-1}, K chunk 64 wave
    load_img<8>(wm, p.wm + (size_t)(l * PG + j) * 8 * 8 * 64 * 4, wave, lane);        // M_l x u_{l-1}, K chunk 32 wave (group 0: zeros)
    load_img<8>(wo, p.wo + (size_t)(l * PG + j) * 8 * 8 * 64 * 4, wave, lane);        // W_out,l-1 rows 64 j + lane
    load_img<4>(ws, p.ws + (size_t)(l * PG + j) * 8 * 4 * 64 * 4, wave, lane);        // W_skip,l-1 rows 32 j + (lane & 31), K chunk 16 (2 wave + lane / 32)
    // the last group also holds W_skip,L-1 (same image layout) -- in LDS: the register file is full
    float4* wsl = reinterpret_cast<float4*>(s.xin + (size_t)BMAX * p.kpre);
    if (last)
        for (int c = 0; c < 4; ++c) wsl[(wave * 4 + c) * 64 + lane] = reinterpret_cast<const float4*>(p.wsl + (size_t)j * 8 * 4 * 64 * 4)[(wave * 4 + c) * 64 + lane];
    const float bo_r = p.bo[(size_t)l * RWD + RS * j + lane];                         // b_out,l-1
    const float bs_r = p.bs[(size_t)l * KWD + KS * j + (lane & 31)];                  // b_skip,l-1
    const float bsl_r = p.bs[(size_t)p.L * KWD + KS * j + (lane & 31)];               // b_skip,L-1
    const float cv_a = p.cvec[(size_t)l * 2 * GHD + GS * j + (lane & 31)], cv_g = p.cvec[(size_t)l * 2 * GHD + GHD + GS * j + (lane & 31)];
    const int d = p.lay_dil[l], rows = (p.kw - 1) * d;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    compute_pre(p, s, l, j, 0, tid, lane, wave);                  // pre_j[0]: history is zero, conditioning row c[0]

    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            u64* x_in = p.xmail + ((size_t)b * (p.L + 1) + l) * XW;
            u64* x_out = x_in + XW;
            // ---- gather (u_{l-1}, h_{l-1}) of step t: the chain ------------------------------------------------------------------
            if (wave < 2) {
                if (!first && !recv128(x_in + 128 * wave, tag, s.ux + 128 * wave, p.status, 0x200u + (unsigned)l, lane)) s.flags[0] = 1;
            } else if (wave < 6) {
                if (!recv128(x_in + GHD + 128 * (wave - 2), tag, s.hx + 128 * (wave - 2), p.status, 0x100u + (unsigned)l, lane)) s.flags[0] = 1;
            }
            __syncthreads();
            if (s.flags[0]) return;
            // ---- one pass: z_l, conv1x1_out of layer l-1, conv1x1_skip of layer l-1 -----------------------------------------------
            {
                float z = dot_bcast<16>(wn, s.hx + 64 * wave);
                if (!first) {
                    z += dot_bcast<8>(wm, s.ux + 32 * wave);
                    s.po[wave * 64 + lane] = dot_bcast<8>(wo, s.ux + 32 * wave);
                    s.ps[(2 * wave + (lane >> 5)) * 32 + (lane & 31)] = dot_bcast<4>(ws, s.ux + 16 * (2 * wave + (lane >> 5)));
                }
                s.pz[wave * 64 + lane] = z;
            }
            __syncthreads();
            float sk_run = 0.f;                                    // wave 2: skip sum up to layer l-1 (kept for the last group)
            if (wave == 0) {                                       // u_l: lanes c and 32 + c hold the tanh / sigmoid rows of channel 32 j + c
                float v = s.pre[b * 64 + lane] + ((lane >> 5) ? cv_g : cv_a);
#pragma unroll
                for (int w = 0; w < 8; ++w) v += s.pz[w * 64 + lane];
                const float g = __shfl_xor(v, 32, 64);
                if (lane < GS) st_granule(x_out + GS * j + lane, tag, wide_gate(v, g), fast_next);                     // modules.py:152-154
            } else if (wave == 1) {                                // h_l = layer l's input (group 0 passes h_0 on)
                float o = bo_r;
#pragma unroll
                for (int w = 0; w < 8; ++w) o += s.po[w * 64 + lane];
                const float hp = s.hx[RS * j + lane];
                st_granule(x_out + GHD + RS * j + lane, tag, first ? hp : (o + hp) * 0.70710678118654752440f, fast_next);  // modules.py:157-162
            } else if (wave == 2 && !first) {                      // skip sum over layers 0 .. l-1 (wavenet.py:312)
                float sk = bs_r, acc = 0.f;
                bool ok = true;
                if (lane < KS) {
#pragma unroll
                    for (int h = 0; h < 16; ++h) sk += s.ps[h * 32 + lane];
                }
                if (l > 1) ok = recv_lanes(p.smail + ((size_t)b * (p.L + 2) + l) * KWD + KS * j, KS, tag, acc, p.status, 0x300u + (unsigned)l, lane);
                sk_run = acc + sk;
                if (!ok) s.flags[0] = 1;
                else if (lane < KS && !last) st_granule(p.smail + ((size_t)b * (p.L + 2) + l + 1) * KWD + KS * j + lane, tag, sk_run, fast_next);
            }
            // ---- behind the chain: the full h_l this group just produced (history, next step's taps); the last group also needs its
            //      own u for the last layer's skip term, which the head is waiting for ---------------------------------------------
            if (last) {
                if (wave < 2) {
                    if (!recv128(x_out + 128 * wave, tag, s.ux + 128 * wave, p.status, 0x500u, lane)) s.flags[0] = 1;
                }
                __syncthreads();
                {
                    float4 wl[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) wl[c] = wsl[(wave * 4 + c) * 64 + lane];
                    s.ps[(2 * wave + (lane >> 5)) * 32 + (lane & 31)] = dot_bcast<4>(wl, s.ux + 16 * (2 * wave + (lane >> 5)));
                }
                __syncthreads();
                if (wave == 2 && lane < KS && !s.flags[0]) {
                    float sk = bsl_r + sk_run;
#pragma unroll
                    for (int h = 0; h < 16; ++h) sk += s.ps[h * 32 + lane];
                    st_granule(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * KWD + KS * j + lane, tag, sk, fast_next);
                }
            }
            if (!first) {                                          // (group 0: h_0 is already in hx)
                __syncthreads();                                   // every wave is done with hx
                if (wave >= 4) {
                    if (!recv128(x_out + GHD + 128 * (wave - 4), tag, s.hx + 128 * (wave - 4), p.status, 0x600u + (unsigned)l, lane)) s.flags[0] = 1;
                }
                __syncthreads();
            }
            if (s.flags[0]) return;
            if (rows > 0) {                                        // this workgroup's own copy of history row t
                float* hist = p.hist + (size_t)b * p.hist_b_floats + (size_t)PG * p.lay_histoff[l] + (size_t)j * rows * RWD;
                hist[(size_t)(t % rows) * RWD + tid] = s.hx[tid];
            }
            __syncthreads();
        }
        // ---- pre_j[t + 1] for every utterance (the rows written above are this CU's own stores) ------------------------------------
        if (t + 1 < p.T) compute_pre(p, s, l, j, t + 1, tid, lane, wave);
    }
}

struct HeadLds {
    float *vs, *hid, *ph, *pout, *obuf, *nz;
    int* flags;
};
__device__ __forceinline__ HeadLds carve_head(float* smem) {
    HeadLds s;
    s.vs = smem; s.hid = s.vs + KWD; s.ph = s.hid + KWD; s.pout = s.ph + 2 * KWD; s.obuf = s.pout + 8 * 64; s.nz = s.obuf + 64;
    s.flags = reinterpret_cast<int*>(s.nz + 64);
    return s;
}
constexpr size_t HEAD_LDS_FLOATS = 4 * KWD + 8 * 64 + 64 + 64 + 16;

__device__ void run_wide_head(const WideParams& p, bool fast_first, float* smem) {
    const HeadLds s = carve_head(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 w1[32], w2[8];
    load_img<32>(w1, p.wh1, wave, lane);                 // rows (wave & 3) 64 + lane, K half (wave >> 2)
    load_img<8>(w2, p.wh2, wave, lane);                  // row lane (< O), K chunk 32 wave
    const float wf = p.wfirst[tid], bf = p.bfirst[tid];
    const float b1 = tid < KWD ? p.bh1[tid] : 0.f;
    const float b2 = lane < p.O ? p.bh2[lane] : 0.f;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    // the input of step 0 (wavenet.py:283-289, :297-308)
    for (int b = 0; b < p.B; ++b) {
        const float xs = p.Tt > 0 ? p.teacher[(size_t)b * p.Tt] : (p.initial ? p.initial[b] : 0.f);
        st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, p.tag_base + 1u, fmaf(wf, xs, bf), fast_first);
    }
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (tid < p.nz) {
                const int kind = (p.dist == 2 && tid == p.nz - 1) ? 1 : 0;
                s.nz[tid] = p.noise ? p.noise[((size_t)t * p.B + b) * p.nz + tid] : wnv_noise_gen(p.seed, t, b, tid, kind);
            }
            if (wave < 2) {
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * KWD + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
            }
            __syncthreads();
            if (s.flags[0]) return;
            if (tid < KWD) s.vs[tid] = fmaxf(s.vs[tid] * p.skip_scale, 0.f);                        // wavenet.py:313-316
            __syncthreads();
            s.ph[(wave >> 2) * KWD + (wave & 3) * 64 + lane] = dot_bcast<32>(w1, s.vs + 128 * (wave >> 2));
            __syncthreads();
            if (tid < KWD) s.hid[tid] = fmaxf(s.ph[tid] + s.ph[KWD + tid] + b1, 0.f);               // wavenet.py:317-318
            __syncthreads();
            s.pout[wave * 64 + lane] = dot_bcast<8>(w2, s.hid + 32 * wave);
            __syncthreads();
            if (wave == 0) {
                float o = b2;
#pragma unroll
                for (int w = 0; w < 8; ++w) o += s.pout[w * 64 + lane];
                if (lane < p.O) {
                    s.obuf[lane] = o;                                                                // wavenet.py:319
                    if (p.params_out) p.params_out[((size_t)b * p.O + lane) * p.T + t] = o;
                }
                const float x = sample_scalar(p.dist, p.O, s.obuf, s.nz, lane);                      // mixture.py:118-156 / :221-270
                if (lane == 0) { p.out[(size_t)b * p.T + t] = x; s.nz[63] = x; }
            }
            __syncthreads();
            if (t + 1 < p.T) {
                const float xs = t + 1 < p.Tt ? p.teacher[(size_t)b * p.Tt + t + 1] : s.nz[63];   // wavenet.py:297-305
                st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, tag + 1u, fmaf(wf, xs, bf), fast_first);
            }
            __syncthreads();
        }
    }
}

// ---- head of one-hot (mu-law categorical) models: TWO workgroups, because the hidden layer (256 x 256) and the output layer
// (out_channels x 256, 256 x 256 for mu-law 256) fill a register file each.  Part A: skip sum -> ReLU -> 1x1 -> ReLU -> publish the
// hidden vector.  Part B: 1x1 -> softmax -> OneHotCategorical (sample_categorical of wnv_sample.h: argmax(p_hat / e)) -> first_conv
// of the next input (wavenet.py:315-319, :332-335, :297-308) -> publish h_0.  first_conv's matrix is K-major in memory: a sampled
// class is one 2-KB row gather (bit-identical to F.linear with a one-hot input); teacher-forced inputs, an explicit initial input and
// fed-back probabilities (quantize = False) take the dense mat-vec.
__device__ void run_wide_head_a(const WideParams& p, float* smem) {
    const HeadLds s = carve_head(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 w1[32];
    load_img<32>(w1, p.wh1, wave, lane);
    const float b1 = tid < KWD ? p.bh1[tid] : 0.f;
    const bool fast = p.fast != 0;                         // part B sits on the same XCD
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (wave < 2) {
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * KWD + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
            }
            __syncthreads();
            if (s.flags[0]) return;
            if (tid < KWD) s.vs[tid] = fmaxf(s.vs[tid] * p.skip_scale, 0.f);                        // wavenet.py:313-316
            __syncthreads();
            s.ph[(wave >> 2) * KWD + (wave & 3) * 64 + lane] = dot_bcast<32>(w1, s.vs + 128 * (wave >> 2));
            __syncthreads();
            if (tid < KWD) st_granule(p.hidmail + (size_t)b * KWD + tid, tag, fmaxf(s.ph[tid] + s.ph[KWD + tid] + b1, 0.f), fast);   // wavenet.py:317-318
            __syncthreads();
        }
    }
}

struct CatLds {
    float *hid, *pout, *obuf, *nz, *vin;
    int* ints;
};
__device__ __forceinline__ CatLds carve_cat(float* smem) {
    CatLds s;
    s.hid = smem; s.pout = s.hid + KWD; s.obuf = s.pout + 8 * 256; s.nz = s.obuf + 256; s.vin = s.nz + 256;
    s.ints = reinterpret_cast<int*>(s.vin + 256);
    return s;
}
constexpr size_t CAT_LDS_FLOATS = KWD + 8 * 256 + 3 * 256 + 16;

__device__ void run_wide_head_b(const WideParams& p, bool fast_first, float* smem) {
    const CatLds s = carve_cat(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int O = p.O;
    float4 w2[32];                                         // rows lane + 64 q (q = 0 .. 3), K chunk 32 wave: [wave][q * 8 + c][lane][4]
    load_img<32>(w2, p.wh2, wave, lane);
    const float b2 = tid < O ? p.bh2[tid] : 0.f;
    const float bf = p.bfirst[tid];
    if (tid == 0) { s.ints[0] = 0; s.ints[1] = 127; }     // abort flag; sampled class
    __syncthreads();
    // first_conv of a dense O-vector (global memory or LDS) or of the one-hot class idx -> h_0 of step `tag_next`
    auto send_input = [&](int b, const float* dense, int idx, unsigned tag_next) {
        float h;
        if (dense == nullptr) {
            h = p.wfirst[(size_t)idx * RWD + tid] + bf;                                              // one row of the K-major matrix
        } else {
            for (int k = tid; k < O; k += WT) s.vin[k] = dense[k];
            __syncthreads();
            float acc = 0.f;
            for (int k = 0; k < O; ++k) acc = fmaf(p.wfirst[(size_t)k * RWD + tid], s.vin[k], acc);
            h = acc + bf;
            __syncthreads();
        }
        st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, tag_next, h, fast_first);
    };
    for (int b = 0; b < p.B; ++b) {                        // wavenet.py:283-289: one-hot of class 127 unless given
        const float* dense = p.Tt > 0 ? p.teacher + (size_t)b * p.Tt * O : (p.initial ? p.initial + (size_t)b * O : nullptr);
        send_input(b, dense, 127, p.tag_base + 1u);
    }
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (tid < O) s.nz[tid] = p.noise ? p.noise[((size_t)t * p.B + b) * p.nz + tid] : wnv_noise_gen(p.seed, t, b, tid, 2);   // e ~ Exp(1)
            if (wave < 2) {
                if (!recv128(p.hidmail + (size_t)b * KWD + 128 * wave, tag, s.hid + 128 * wave, p.status, 0x480u, lane)) s.ints[0] = 1;
            }
            __syncthreads();
            if (s.ints[0]) return;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 wq[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) wq[c] = w2[q * 8 + c];
                s.pout[wave * 256 + 64 * q + lane] = dot_bcast<8>(wq, s.hid + 32 * wave);
            }
            __syncthreads();
            if (tid < O) {
                float o = b2;
#pragma unroll
                for (int w = 0; w < 8; ++w) o += s.pout[w * 256 + tid];
                s.obuf[tid] = o;                                                                     // wavenet.py:319
                if (p.params_out) p.params_out[((size_t)b * O + tid) * p.T + t] = o;
            }
            __syncthreads();
            if (wave == 0) {                                                                         // wavenet.py:332-335
                const int idx = sample_categorical(O, s.obuf, s.nz, p.softmax, p.quantize, lane);
                if (p.quantize) {
                    if (lane == 0) {
                        p.out[((size_t)b * O + idx) * p.T + t] = 1.0f;                               // out is pre-zeroed by the host
                        if (p.index_out) p.index_out[(size_t)b * p.T + t] = idx;
                        s.ints[1] = idx;
                    }
                } else {
                    for (int n = lane; n < O; n += 64) p.out[((size_t)b * O + n) * p.T + t] = s.obuf[n];
                }
            }
            __syncthreads();
            if (t + 1 < p.T) {                                                                       // wavenet.py:297-308 for step t + 1
                const float* dense = nullptr;
                if (t + 1 < p.Tt) dense = p.teacher + ((size_t)b * p.Tt + t + 1) * O;
                else if (!p.quantize) dense = s.obuf;                                                // fed-back probabilities
                send_input(b, dense, s.ints[1], tag + 1u);
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(WT) wnv_wide_kernel(const WideParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int x = blockIdx.x & 7, li = blockIdx.x >> 3;
    if (x == p.head_x && li == p.head_li) {
        if (p.cin1 == 1) run_wide_head(p, p.fast && p.head_x == 0, smem);      // group 0 lives on XCD 0
        else run_wide_head_a(p, smem);
        return;
    }
    if (p.cin1 > 1 && x == p.head_x && li == p.head_li + 1) {
        run_wide_head_b(p, p.fast && p.head_x == 0, smem);
        return;
    }
    const int sl = li / PG, j = li % PG, l = x * p.nL + sl;
    if (sl >= p.nL || l >= p.L) return;
    // who reads what this group publishes for the NEXT layer: group l + 1 (same XCD unless this is the last group of its XCD) or the head
    const int next_x = l + 1 < p.L ? (l + 1) / p.nL : p.head_x;
    run_wide_stage(p, l, j, p.fast && next_x == x, smem);
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================
struct WnvWideState {
    int device = 0;
    int L = 0, O = 0, cin = 0, cinp = 0, kw = 0, kpre = 0, nkb = 0, cin1 = 1;
    float* d_w = nullptr;
    size_t o_wn = 0, o_wm = 0, o_wo = 0, o_ws = 0, o_wsl = 0, o_bo = 0, o_bs = 0, o_cvec = 0, o_wpre = 0, o_wh1 = 0, o_bh1 = 0, o_wh2 = 0, o_bh2 = 0, o_wf = 0, o_bf = 0;
    int* d_dil = nullptr;
    int* d_histoff = nullptr;
    long long hist_layer_floats = 0;          // one copy set: sum over layers of rows * 512
    void* d_state = nullptr;
    size_t state_cap = 0;
    unsigned tag_next = 0;
    size_t mail_bytes = 0;
    unsigned int* h_status = nullptr;
    int ncu = 0, n_xcd = 0;
    bool map_ok = false;
};

static const char* wide_why_not(const wnv_config& c, int B) {
    if (c.scalar_input && c.out_channels > 64) return "scalar-input models need out_channels <= 64";
    if (!c.scalar_input && c.out_channels > 256) return "one-hot models need out_channels <= 256";
    if (c.residual_channels > RWD || c.gate_channels > 2 * GHD) return "needs residual_channels <= 512 and gate_channels <= 512";
    if (c.skip_out_channels > KWD) return "needs skip_out_channels <= 256";
    if (c.kernel_size < 2 || c.kernel_size > 4) return "needs 2 <= kernel_size <= 4";
    if (c.cin_channels > 128) return "needs cin_channels <= 128";
    if (c.layers > 31) return "needs layers <= 31 (8 workgroups per layer, 32 CUs per XCD, one or two more for the head)";
    if (B > BMAX) return "more than 8 utterances per call";
    return nullptr;
}
bool wnv_wide_supported(const wnv_config& c, int B) { return wide_why_not(c, B) == nullptr; }
const char* wnv_wide_why_not(const wnv_config& c, int B) { const char* w = wide_why_not(c, B); return w ? w : "supported"; }

void wnv_wide_destroy(WnvWideState* st) {
    if (!st) return;
    if (st->d_w) (void)hipFree(st->d_w);
    if (st->d_dil) (void)hipFree(st->d_dil);
    if (st->d_histoff) (void)hipFree(st->d_histoff);
    if (st->d_state) (void)hipFree(st->d_state);
    if (st->h_status) (void)hipHostFree(st->h_status);
    delete st;
}

#define WIDE_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) { err = std::string(#expr) + " failed: " + hipGetErrorString(e__); return WNV_ERR_HIP; } \
    } while (0)

static wnv_status wide_build(WnvWideState** out, int device, const wnv_config& c, const TensorStore& store, std::string& err) {
    WnvWideState* st = new WnvWideState();
    *out = st;
    st->device = device;
    const int L = c.layers, kw = c.kernel_size, cin = c.cin_channels > 0 ? c.cin_channels : 0, cinp = (cin + 3) & ~3;
    const int Ra = c.residual_channels, Ga = c.gate_channels, Gha = Ga / 2, Ka = c.skip_out_channels, O = c.out_channels;
    st->L = L; st->O = O; st->cin = cin; st->cinp = cinp; st->kw = kw;
    st->kpre = (kw - 1) * RWD + cinp; st->nkb = st->kpre / 4;
    std::vector<float> blob;
    auto alloc = [&](size_t n) { size_t o = (blob.size() + 3) & ~(size_t)3; blob.resize(o + n, 0.f); return o; };
    auto T = [&](const std::string& n) -> const HostTensor& { return *store.get(n); };
    // padded gate row o in [0, 512): tanh channels 0..255 then sigmoid channels 0..255 -> the model's gate row, or -1
    auto gate_row = [&](int o) { const int ch = o & (GHD - 1); return ch < Gha ? (o >> 8) * Gha + ch : -1; };
    // gate row of lane r of slice j: 32 tanh rows then the 32 sigmoid rows of the same channels
    auto slice_row = [&](int j, int r) { return (r >> 5) * GHD + GS * j + (r & 31); };
    const size_t n_wn = (size_t)8 * 16 * 64 * 4, n_wm = (size_t)8 * 8 * 64 * 4, n_wo = n_wm, n_ws = (size_t)8 * 4 * 64 * 4, n_pre = (size_t)st->nkb * 64 * 4;
    st->o_wn = alloc((size_t)L * PG * n_wn);
    st->o_wm = alloc((size_t)L * PG * n_wm);
    st->o_wo = alloc((size_t)L * PG * n_wo);
    st->o_ws = alloc((size_t)L * PG * n_ws);
    st->o_wsl = alloc((size_t)PG * n_ws);
    st->o_bo = alloc((size_t)L * RWD);
    st->o_bs = alloc((size_t)(L + 1) * KWD);
    st->o_cvec = alloc((size_t)L * 2 * GHD);
    st->o_wpre = alloc((size_t)L * PG * n_pre);
    std::vector<int> dil(L), hoff(L);
    long long hist = 0;
    const int per = L / c.stacks;
    const double rs = std::sqrt(0.5);
    std::vector<float> cur((size_t)2 * GHD * RWD), mmat((size_t)2 * GHD * GHD);
    std::vector<double> accd(GHD);
    // skip image of layer `ls` for slice j
    auto put_skip = [&](float* is, const HostTensor& wsk, int j) {
        for (int w = 0; w < 8; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                const int so = KS * j + (lane & 31), hw = 2 * w + (lane >> 5);      // skip row, K chunk [16 hw, 16 hw + 16)
                for (int cq = 0; cq < 4; ++cq)
                    for (int e = 0; e < 4; ++e) {
                        const int k = 16 * hw + 4 * cq + e;
                        is[(((size_t)w * 4 + cq) * 64 + lane) * 4 + e] = (so < Ka && k < Gha) ? wsk.data[(size_t)so * Gha + k] : 0.f;
                    }
            }
    };
    for (int l = 0; l < L; ++l) {
        const std::string pfx = "conv_layers." + std::to_string(l) + ".", ppx = "conv_layers." + std::to_string(l - 1) + ".";
        const HostTensor& wc = T(pfx + "conv.weight");                 // (G, R, kw)
        const HostTensor* wcc = cin > 0 ? &T(pfx + "conv1x1c.weight") : nullptr;   // (G, cin, 1)
        const HostTensor* wout = l > 0 ? &T(ppx + "conv1x1_out.weight") : nullptr; // (R, G/2, 1) of layer l-1
        const HostTensor* bout = l > 0 ? &T(ppx + "conv1x1_out.bias") : nullptr;
        const HostTensor* wsk = l > 0 ? &T(ppx + "conv1x1_skip.weight") : nullptr; // (K, G/2, 1) of layer l-1
        // newest tap as a padded (512 x 512) matrix; folded chain matrices (double accumulation, rounded once):
        //   M_l = sqrt(.5) W_cur,l W_out,l-1 (512 x 256), N_l = sqrt(.5) W_cur,l, c_l = N_l b_out,l-1;   layer 0: N_0 = W_cur,0, M_0 = 0
        std::fill(cur.begin(), cur.end(), 0.f);
        for (int o = 0; o < 2 * GHD; ++o) {
            const int go = gate_row(o);
            if (go < 0) continue;
            for (int k = 0; k < Ra; ++k) cur[(size_t)o * RWD + k] = wc.data[((size_t)go * Ra + k) * kw + (kw - 1)];
        }
        std::fill(mmat.begin(), mmat.end(), 0.f);
        if (l > 0) {
            for (int o = 0; o < 2 * GHD; ++o) {
                if (gate_row(o) < 0) continue;
                std::fill(accd.begin(), accd.end(), 0.0);
                double cb = 0.0;
                for (int m = 0; m < Ra; ++m) {
                    const double cm = (double)cur[(size_t)o * RWD + m];
                    const float* wrow = wout->data.data() + (size_t)m * Gha;
                    for (int k = 0; k < Gha; ++k) accd[k] += cm * (double)wrow[k];
                    cb += cm * (double)bout->data[m];
                }
                for (int k = 0; k < Gha; ++k) mmat[(size_t)o * GHD + k] = (float)(rs * accd[k]);
                blob[st->o_cvec + (size_t)l * 2 * GHD + o] = (float)(rs * cb);
            }
        }
        for (int j = 0; j < PG; ++j) {
            float* in_ = blob.data() + st->o_wn + (size_t)(l * PG + j) * n_wn;
            float* im = blob.data() + st->o_wm + (size_t)(l * PG + j) * n_wm;
            float* io = blob.data() + st->o_wo + (size_t)(l * PG + j) * n_wo;
            float* ip = blob.data() + st->o_wpre + (size_t)(l * PG + j) * n_pre;
            for (int w = 0; w < 8; ++w)
                for (int lane = 0; lane < 64; ++lane) {
                    const int o = slice_row(j, lane);
                    for (int cq = 0; cq < 16; ++cq)                     // N_l, K chunk [64 w, 64 w + 64) of h_{l-1}
                        for (int e = 0; e < 4; ++e) {
                            const int k = 64 * w + 4 * cq + e;
                            const float v = cur[(size_t)o * RWD + k];
                            in_[(((size_t)w * 16 + cq) * 64 + lane) * 4 + e] = l == 0 ? v : (float)(rs * (double)v);
                        }
                    const int ro = RS * j + lane;                       // M_l and W_out,l-1, K chunk [32 w, 32 w + 32) of u_{l-1}
                    for (int cq = 0; cq < 8; ++cq)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 32 * w + 4 * cq + e;
                            im[(((size_t)w * 8 + cq) * 64 + lane) * 4 + e] = mmat[(size_t)o * GHD + k];
                            io[(((size_t)w * 8 + cq) * 64 + lane) * 4 + e] = (l > 0 && ro < Ra && k < Gha) ? wout->data[(size_t)ro * Gha + k] : 0.f;
                        }
                }
            if (l > 0) put_skip(blob.data() + st->o_ws + (size_t)(l * PG + j) * n_ws, *wsk, j);
            if (l == L - 1) put_skip(blob.data() + st->o_wsl + (size_t)j * n_ws, T(pfx + "conv1x1_skip.weight"), j);
            for (int kb = 0; kb < st->nkb; ++kb)                        // older taps (oldest first) then local conditioning, [kb][lane][4]
                for (int lane = 0; lane < 64; ++lane) {
                    const int go = gate_row(slice_row(j, lane));
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * kb + e;
                        float v = 0.f;
                        if (go >= 0) {
                            if (k < (kw - 1) * RWD) {
                                const int tap = k >> 9, ch = k & (RWD - 1);
                                if (ch < Ra) v = wc.data[((size_t)go * Ra + ch) * kw + tap];
                            } else if (k - (kw - 1) * RWD < cin) {
                                v = wcc->data[(size_t)go * cin + (k - (kw - 1) * RWD)];
                            }
                        }
                        ip[((size_t)kb * 64 + lane) * 4 + e] = v;
                    }
                }
        }
        if (l > 0) {
            std::copy(bout->data.begin(), bout->data.end(), blob.begin() + st->o_bo + (size_t)l * RWD);
            const HostTensor& bs = T(ppx + "conv1x1_skip.bias");
            std::copy(bs.data.begin(), bs.data.end(), blob.begin() + st->o_bs + (size_t)l * KWD);
        }
        if (l == L - 1) {
            const HostTensor& bs = T(pfx + "conv1x1_skip.bias");
            std::copy(bs.data.begin(), bs.data.end(), blob.begin() + st->o_bs + (size_t)L * KWD);
        }
        dil[l] = 1 << (l % per);
        hoff[l] = (int)hist;
        hist += (long long)(kw - 1) * dil[l] * RWD;
    }
    st->hist_layer_floats = hist;
    // head: W1 rows (w & 3) 64 + lane, K half (w >> 2) -> [w][32][lane][4];  W2: scalar models row lane, K chunk 32 w -> [w][8][lane][4];
    // one-hot models rows lane + 64 q (q = 0 .. 3), K chunk 32 w -> [w][8 q + c][lane][4]
    const int cin1 = c.scalar_input ? 1 : O;
    st->cin1 = cin1;
    st->o_wh1 = alloc((size_t)8 * 32 * 64 * 4);
    st->o_wh2 = alloc((size_t)8 * (cin1 > 1 ? 32 : 8) * 64 * 4);
    {
        const HostTensor& w1 = T("last_conv_layers.1.weight");         // (K, K, 1)
        const HostTensor& w2 = T("last_conv_layers.3.weight");         // (O, K, 1)
        for (int w = 0; w < 8; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = (w & 3) * 64 + lane;
                for (int cq = 0; cq < 32; ++cq)
                    for (int e = 0; e < 4; ++e) {
                        const int k = 128 * (w >> 2) + 4 * cq + e;
                        blob[st->o_wh1 + (((size_t)w * 32 + cq) * 64 + lane) * 4 + e] = (row < Ka && k < Ka) ? w1.data[(size_t)row * Ka + k] : 0.f;
                    }
                for (int q = 0; q < (cin1 > 1 ? 4 : 1); ++q)
                    for (int cq = 0; cq < 8; ++cq)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 32 * w + 4 * cq + e, orow = lane + 64 * q;
                            blob[st->o_wh2 + (((size_t)w * (cin1 > 1 ? 32 : 8) + 8 * q + cq) * 64 + lane) * 4 + e] =
                                (orow < O && k < Ka) ? w2.data[(size_t)orow * Ka + k] : 0.f;
                        }
            }
    }
    st->o_bh1 = alloc(KWD);
    std::copy(T("last_conv_layers.1.bias").data.begin(), T("last_conv_layers.1.bias").data.end(), blob.begin() + st->o_bh1);
    st->o_bh2 = alloc(256);
    std::copy(T("last_conv_layers.3.bias").data.begin(), T("last_conv_layers.3.bias").data.end(), blob.begin() + st->o_bh2);
    // first_conv: (R, 1, 1) for scalar input; one-hot models: (R, O, 1) stored K-major [O][512] (row k = column k of the matrix)
    st->o_wf = alloc((size_t)cin1 * RWD);
    st->o_bf = alloc(RWD);
    for (int r = 0; r < Ra; ++r) {
        for (int k = 0; k < cin1; ++k) blob[st->o_wf + (size_t)k * RWD + r] = T("first_conv.weight").data[(size_t)r * cin1 + k];
        blob[st->o_bf + r] = T("first_conv.bias").data[r];
    }
    WIDE_HIP(hipMalloc((void**)&st->d_w, blob.size() * sizeof(float)));
    WIDE_HIP(hipMemcpy(st->d_w, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    WIDE_HIP(hipMalloc((void**)&st->d_dil, L * sizeof(int)));
    WIDE_HIP(hipMemcpy(st->d_dil, dil.data(), L * sizeof(int), hipMemcpyHostToDevice));
    WIDE_HIP(hipMalloc((void**)&st->d_histoff, L * sizeof(int)));
    WIDE_HIP(hipMemcpy(st->d_histoff, hoff.data(), L * sizeof(int), hipMemcpyHostToDevice));
    WIDE_HIP(hipHostMalloc((void**)&st->h_status, 64, hipHostMallocDefault));
    *st->h_status = 0;
    wnv_status cs = wnv_placement_census(device, &st->ncu, &st->n_xcd, &st->map_ok, err);
    return cs;
}

wnv_status wnv_wide_generate(WnvWideState** pst, int device, const wnv_config& c, const TensorStore& store, const WnvGenArgs& ga,
                             hipStream_t stream, std::string& err) {
    if (!*pst) {
        wnv_status st0 = wide_build(pst, device, c, store, err);
        if (st0 != WNV_OK) { wnv_wide_destroy(*pst); *pst = nullptr; return st0; }
    }
    WnvWideState* st = *pst;
    const int B = ga.B, L = st->L;
    if (!st->map_ok || st->n_xcd != 8) {
        char buf[160];
        snprintf(buf, sizeof buf, "wide kernel: placement census found %d XCDs over %d CUs with the block -> XCD mapping %s (needs 8 XCDs, b %% 8)",
                 st->n_xcd, st->ncu, st->map_ok ? "as assumed" : "NOT as assumed");
        err = buf;
        return WNV_ERR_UNSUPPORTED;
    }
    const int cus_per_xcd = st->ncu / 8;
    const int nL = (L + 7) / 8;                                      // layer groups per XCD
    // the head goes to the XCD of the last layer when that XCD has a free slot, else to the first XCD that has one
    int head_x = -1, head_li = -1;
    const int n_head = st->cin1 > 1 ? 2 : 1;                        // one-hot models: the head is two workgroups
    auto groups_on = [&](int x) { return std::max(0, std::min(nL, L - x * nL)); };
    const int last_x = (L - 1) / nL;
    for (int k = 0; k < 8 && head_x < 0; ++k) {
        const int x = (last_x + k) % 8;
        if (groups_on(x) * PG + n_head <= cus_per_xcd) { head_x = x; head_li = groups_on(x) * PG; }
    }
    if (head_x < 0 || nL * PG > cus_per_xcd) { err = "wide kernel: not enough CUs per XCD for the layer groups + the head"; return WNV_ERR_UNSUPPORTED; }
    WideParams p{};
    p.L = L; p.nL = nL; p.B = B; p.T = (int)ga.T; p.Tt = (int)ga.Tt; p.O = st->O; p.cin = st->cin; p.cinp = st->cinp; p.kw = st->kw; p.nz = ga.nz;
    p.dist = c.output_distribution; p.kpre = st->kpre; p.nkb = st->nkb;
    p.head_x = head_x; p.head_li = head_li;
    p.cin1 = st->cin1; p.softmax = ga.softmax; p.quantize = ga.quantize; p.index_out = ga.index_out;
    { const char* e = getenv("WNV_RING_FAST"); p.fast = !(e && e[0] == '0'); }
    p.skip_scale = (float)std::sqrt(1.0 / L);
    const float* w = st->d_w;
    p.wn = w + st->o_wn; p.wm = w + st->o_wm; p.wo = w + st->o_wo; p.ws = w + st->o_ws; p.wsl = w + st->o_wsl; p.bo = w + st->o_bo; p.bs = w + st->o_bs;
    p.cvec = w + st->o_cvec; p.wpre = w + st->o_wpre;
    p.wh1 = w + st->o_wh1; p.bh1 = w + st->o_bh1; p.wh2 = w + st->o_wh2; p.bh2 = w + st->o_bh2; p.wfirst = w + st->o_wf; p.bfirst = w + st->o_bf;
    p.zbias = ga.zbias; p.zbias_bstride = ga.zbias_bstride; p.zb_ld = (c.gate_channels + 3) & ~3; p.gh_model = c.gate_channels / 2;
    p.lay_dil = st->d_dil; p.lay_histoff = st->d_histoff;
    p.hist_b_floats = (long long)PG * st->hist_layer_floats;
    // state: [status 64 B][X B (L+1) 768 u64][SK B (L+2) 256 u64][history B x 8 copies x layers]
    const size_t head_bytes = 64;
    const size_t n_x = (size_t)B * (L + 1) * XW, n_s = (size_t)B * (L + 2) * KWD;
    const size_t n_hid = (size_t)B * KWD;
    const size_t mail_bytes = (n_x + n_s + n_hid) * sizeof(u64);
    const size_t hist_bytes = (size_t)B * p.hist_b_floats * sizeof(float);
    const size_t bytes = head_bytes + mail_bytes + hist_bytes;
    bool fresh = false;
    if (bytes > st->state_cap) {
        if (st->d_state) { WIDE_HIP(hipFree(st->d_state)); st->d_state = nullptr; st->state_cap = 0; }
        WIDE_HIP(hipMalloc(&st->d_state, bytes));
        st->state_cap = bytes;
        fresh = true;
    }
    char* base = (char*)st->d_state;
    if (fresh || mail_bytes != st->mail_bytes || (unsigned long long)st->tag_next + (unsigned long long)ga.T + 2ull > 0xFFFFFFF0ull) {
        WIDE_HIP(hipMemsetAsync(base, 0, head_bytes + mail_bytes, stream));
        st->tag_next = 0;
        st->mail_bytes = mail_bytes;
    } else {
        WIDE_HIP(hipMemsetAsync(base, 0, head_bytes, stream));
    }
    WIDE_HIP(hipMemsetAsync(base + head_bytes + mail_bytes, 0, hist_bytes, stream));        // clear_buffer (wavenet.py:241)
    p.tag_base = st->tag_next;
    st->tag_next += (unsigned)ga.T + 1u;
    p.status = (unsigned int*)base;
    p.xmail = (u64*)(base + head_bytes);
    p.smail = p.xmail + n_x;
    p.hidmail = p.smail + n_s;
    p.hist = (float*)(p.hidmail + n_hid);
    p.c_up = ga.c_up; p.initial = ga.initial; p.teacher = ga.teacher; p.noise = ga.noise; p.seed = ga.seed;
    p.out = ga.out; p.params_out = ga.params_out;
    const size_t lds = std::max(std::max(stage_lds_floats(p.kpre), HEAD_LDS_FLOATS), CAT_LDS_FLOATS) * sizeof(float);
    if (lds > 160 * 1024) { err = "wide kernel needs too much LDS"; return WNV_ERR_UNSUPPORTED; }
    WIDE_HIP(hipFuncSetAttribute((const void*)wnv_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        int per_cu = 0;
        WIDE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)wnv_wide_kernel, WT, lds));
        const int live = L * PG + n_head;
        if (per_cu < 1 || live > st->ncu * per_cu) {
            char buf[160];
            snprintf(buf, sizeof buf, "wide kernel: %d workgroups must be co-resident but the device holds %d", live, st->ncu * std::max(per_cu, 0));
            err = buf;
            return WNV_ERR_UNSUPPORTED;
        }
    }
    const int max_li = std::max(nL * PG - 1, head_li + n_head - 1);
    const int grid = 8 * (max_li + 1);
    hipLaunchKernelGGL(wnv_wide_kernel, dim3(grid), dim3(WT), lds, stream, p);
    WIDE_HIP(hipGetLastError());
    WIDE_HIP(hipMemcpyAsync(st->h_status, p.status, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    WIDE_HIP(hipStreamSynchronize(stream));
    if (*st->h_status != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "wide kernel gave up waiting (code 0x%x: 0x1ll / 0x2ll = h / u into group ll, 0x3ll = skip, 0x400 = head, 0x5.. / 0x6.. = own outputs)", *st->h_status);
        err = buf;
        return WNV_ERR_TIMEOUT;
    }
    return WNV_OK;
}
