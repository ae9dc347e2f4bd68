// wnv_wide.hip -- the sample loop for WIDE models (residual_channels <= 512, gate_channels <= 512, skip_out_channels <= 256):
// the reference's own default constructor geometry (wavenet.py:98-101) and its published 512 / 512 / 256 models, for which one
// layer (4.1 MB of weights) is eight times what one CU can hold, so the one-layer-per-CU ring kernel (wnv_ring.hip) does not
// apply and the generic kernel has to stream 99 MB of weights per sample.
//
// GROUP RING.  Every layer gets a GROUP of 8 workgroups on one XCD, one per CU, each owning a slice of the layer's OUTPUT
// channels with its slice of the weights resident in registers:
//
//     head -> group 0 -> group 1 -> ... -> group L-1 -> tail group -> head        (4 groups per XCD, consecutive layers share an XCD)
//
// ONE hop and ONE mat-vec phase per layer.  Group l-1 publishes the PAIR (u_{l-1}, h_{l-1}) -- its gate outputs and its own layer
// input; from that pair alone CU j of group l computes, in one pass over its registers (modules.py:127-163 with the residual
// recurrence substituted into the next pre-activation, folded on the host in double as in wnv_ring.hip):
//     z_l   = M_l u_{l-1} + N_l h_{l-1} + c_l + pre_l        64 gate rows;  M_l = sqrt(.5) W_cur,l W_out,l-1, N_l = sqrt(.5) W_cur,l,
//     u_l   = tanh . sigmoid (z_l)                           c_l = N_l b_out,l-1                        -> publish 32 values
//     h_l   = sqrt(.5) (W_out,l-1 u_{l-1} + b + h_{l-1})     64 rows: layer l's input, the reference's own recurrence -> publish
//     s_{l-1} = W_skip,l-1 u_{l-1} + b                       32 skip channels, summed from group to group (CU j -> CU j)
//   (group 0 reads h_0 from the head: z_0 = W_cur,0 h_0 + pre_0, and passes h_0 on; the TAIL group -- 8 more workgroups -- turns
//   u_{L-1} into the last skip term and hands the finished skip sum to the head).  Behind the chain every CU copies the full h_l its
//   group produced into its OWN copy of the layer's input history (no cross-CU ordering needed; deferred by two utterances in a
//   batch, see run_wide_stage) and streams the older taps + the local-conditioning 1x1 of the NEXT step -- pre_j[t+1], 282 KB of
//   weights per CU -- from L2 / Infinity Cache, once per step for up to 8 utterances (twice for 9 .. 16).
//   Hand-off = the ring kernel's data-tagged 8-byte granules; plain stores inside an XCD (the host's placement census verified that
//   blocks b and b % 8 share an XCD), write-through stores across XCDs.  Mat-vec mapping: the 8 waves split K; a lane holds four rows
//   x a K quarter (dot_quad / reduce_quads: v_permlane32_swap + v_permlane16_swap reduce-scatter), partial sums of the waves meet in
//   LDS.  (v1 of this kernel evaluated the layer unfolded, two hops + two phases per layer: 57 us per step; one row per lane with the
//   vector broadcast from LDS was LDS-bandwidth-bound: 39 us; now 36 us.)
//
// Models narrower than 512 / 512 / 256 are zero-padded (exact).  Scalar-input models (MoL / Gaussian, out_channels <= 64);
// utterances beyond the first share the groups like a systolic array (B <= 16).  Every wait is bounded (WNV_ERR_TIMEOUT).
#include "wnv_wide.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wnv_knobs.h"
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
constexpr int BMAX = 16;                              // utterances per call (the tap stream takes them 8 at a time)
constexpr int XW = GHD + RWD;      // one mailbox slot: 256 gate outputs then 512 layer inputs
constexpr unsigned SPIN_LIMIT = 1u << 22;

using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

struct WideParams {
    int L, nL, B, T, Tt, O, cin, cinp, kw, nz, dist, kpre, nkb;
    int b0, noise_B;                                      // this launch is utterances [b0, b0 + B) of a call of noise_B (noise addressing)
    int head_x, head_li, fast;
    int NB, KW;                                           // skip banks of 256 channels (1; 2 for 257 .. 512 skip channels) and the padded skip width 256 NB
    u64* omail;                                           // K = 512: partial head outputs of head parts 1 .. 3 -> part 0: O[b][4][64]
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
    u64 *xmail, *smail;                                   // X[b][L+1][768] = (u_{l-1} | h_{l-1}) for group l (slot L: the tail group's input and the last layer group's own
                                                          // outputs); SK[b][L+2][256] running skip sums
    float* hist;
    const float *c_up, *initial, *teacher, *noise;
    u64 seed;
    float *out, *params_out;
    unsigned int* status;
    unsigned long long* trace;                            // optional [trace_n][L + 1][16] wall-clock stamps of utterance 0, slice 0 (debug)
    int trace_t0, trace_n, trace_b;
};

// (timeline stamps exist in -DWNV_FINE_TRACE builds only: the run-time test sat on the chain of every group)
constexpr int WTW = 16;            // stamp slots per (step, group)
__device__ __forceinline__ void wstamp(const WideParams& p, int b, int t, int pos, int k, int who) {
#ifdef WNV_FINE_TRACE
    if (p.trace && b == p.trace_b && (int)threadIdx.x == who && t >= p.trace_t0 && t < p.trace_t0 + p.trace_n)
        p.trace[((size_t)(t - p.trace_t0) * (p.L + 1) + pos) * WTW + k] = wall_clock64();
#else
    (void)p; (void)b; (void)t; (void)pos; (void)k; (void)who;
#endif
}

// The stamps ON the chain are deferred (as in wnv_ring.hip): noted in registers where something happens -- one s_memrealtime, no store
// -- and written once per step behind everything that is timed; the immediate wstamp() above stays for the off-chain phases.
#ifdef WNV_FINE_TRACE
#define WTS_DECL unsigned long long wts[WTW] = {0}
#define WTS(k) (wts[k] = __builtin_amdgcn_s_memrealtime())
__device__ __forceinline__ void wts_flush(const WideParams& p, int b, int t, int pos, const unsigned long long (&wts)[WTW], unsigned mask) {
    if ((threadIdx.x & 63) != 0 || !p.trace || b != p.trace_b || t < p.trace_t0 || t >= p.trace_t0 + p.trace_n) return;
    unsigned long long* row = p.trace + ((size_t)(t - p.trace_t0) * (p.L + 1) + pos) * WTW;
#pragma unroll
    for (int k = 0; k < WTW; ++k)
        if ((mask >> k) & 1u) row[k] = wts[k];
}
#define WTS_FLUSH(b, t, pos, mask) wts_flush(p, b, t, pos, wts, mask)
#else
#define WTS_DECL
#define WTS(k) ((void)0)
#define WTS_FLUSH(b, t, pos, mask) ((void)0)
#endif

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
            *reinterpret_cast<float2*>(dst + 2 * lane) = make_float2(__uint_as_float(x.x), __uint_as_float(x.z));     // (LDS or global)
            return true;
        }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}
// one wave receives 256 consecutive granules (both loads in flight together)
__device__ __forceinline__ bool recv256(const u64* g, unsigned tag, float* dst, unsigned int* status, unsigned code, int lane) {
    const u64* g2 = g + 2 * lane;
    for (unsigned spins = 0;;) {
        u4v x, y;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x), "=&v"(y) : "v"(g2) : "memory");
        if (__all(x.y == tag && x.w == tag && y.y == tag && y.w == tag)) {
            *reinterpret_cast<float2*>(dst + 2 * lane) = make_float2(__uint_as_float(x.x), __uint_as_float(x.z));
            *reinterpret_cast<float2*>(dst + 128 + 2 * lane) = make_float2(__uint_as_float(y.x), __uint_as_float(y.z));
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
// init + x[0] + x[1] + ... (that order), x[i] = base[i * stride] in LDS.  Written as a loop the compiler waits for every load (pair)
// before it issues the next one -- five LDS round trips for eight values, ~0.18 us behind the barrier of EVERY group on the chain
// (profiles/r03_wide_timeline.txt: "sum_partials") --, so the loads are issued together and pinned before the first add.
__device__ __forceinline__ float lds_sum8(float init, const float* base, int stride) {
    float x0 = base[0], x1 = base[stride], x2 = base[2 * stride], x3 = base[3 * stride];
    float x4 = base[4 * stride], x5 = base[5 * stride], x6 = base[6 * stride], x7 = base[7 * stride];
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    return (((((((init + x0) + x1) + x2) + x3) + x4) + x5) + x6) + x7;
}
__device__ __forceinline__ float lds_sum16(float init, const float* base, int stride) {
    return lds_sum8(lds_sum8(init, base, stride), base + 8 * stride, stride);
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

// ---- lane-quad mat-vec: a wave owns 64 rows x one K span.  Lane L = (column i = L & 15, K quarter q = L >> 4) holds, for every k of
// its quarter, the weights of FOUR rows ("slots": rows i, 32 + i, 16 + i, 48 + i of the wave's 64) -- so a lane reads a quarter of
// the LDS vector (4x less LDS traffic than one row per lane, which was LDS-bandwidth-bound) and the four quarters are summed with a
// two-step reduce-scatter on v_permlane32_swap / v_permlane16_swap (VALU, no LDS), after which lane L holds row L.
template <int NK>
__device__ __forceinline__ void dot_quad(const float4 (&w)[NK], const float* xq, f2 (&a)[4]) {          // a: slots (0,1),(2,3) x even / odd k
#pragma unroll
    for (int c = 0; c < NK / 4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(xq)[c];
        a[0] = __builtin_elementwise_fma(f2{w[4 * c].x, w[4 * c].y}, f2{v.x, v.x}, a[0]);
        a[1] = __builtin_elementwise_fma(f2{w[4 * c].z, w[4 * c].w}, f2{v.x, v.x}, a[1]);
        a[2] = __builtin_elementwise_fma(f2{w[4 * c + 1].x, w[4 * c + 1].y}, f2{v.y, v.y}, a[2]);
        a[3] = __builtin_elementwise_fma(f2{w[4 * c + 1].z, w[4 * c + 1].w}, f2{v.y, v.y}, a[3]);
        a[0] = __builtin_elementwise_fma(f2{w[4 * c + 2].x, w[4 * c + 2].y}, f2{v.z, v.z}, a[0]);
        a[1] = __builtin_elementwise_fma(f2{w[4 * c + 2].z, w[4 * c + 2].w}, f2{v.z, v.z}, a[1]);
        a[2] = __builtin_elementwise_fma(f2{w[4 * c + 3].x, w[4 * c + 3].y}, f2{v.w, v.w}, a[2]);
        a[3] = __builtin_elementwise_fma(f2{w[4 * c + 3].z, w[4 * c + 3].w}, f2{v.w, v.w}, a[3]);
    }
}
// sum over lanes L and L ^ 32: lanes < 32 get a[L] + a[L + 32], lanes >= 32 get b[L - 32] + b[L]
__device__ __forceinline__ float swap32_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over lanes L and L ^ 16: even 16-lane rows get a[L] + a[L + 16], odd rows get b[L - 16] + b[L]
__device__ __forceinline__ float swap16_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float reduce_quads(const f2 (&a)[4]) {                                        // -> lane L holds row L
    const f2 s01 = a[0] + a[2], s23 = a[1] + a[3];
    return swap16_sum(swap32_sum(s01.x, s01.y), swap32_sum(s23.x, s23.y));
}
// skip rows: 32 rows x 32 k per wave; a lane holds rows i and 16 + i for the 8 k of its quarter: w[c] = {ra[2c], rb[2c], ra[2c+1], rb[2c+1]}.
// Returns the partial sum (over the K quarters q and q ^ 2) of row (L & 15) + 16 (L >> 5).
__device__ __forceinline__ float dot_skip(const float4 (&w)[4], const float* xq) {
    const float4 v0 = reinterpret_cast<const float4*>(xq)[0], v1 = reinterpret_cast<const float4*>(xq)[1];
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
    a0 = __builtin_elementwise_fma(f2{w[0].x, w[0].y}, f2{v0.x, v0.x}, a0);
    a1 = __builtin_elementwise_fma(f2{w[0].z, w[0].w}, f2{v0.y, v0.y}, a1);
    a0 = __builtin_elementwise_fma(f2{w[1].x, w[1].y}, f2{v0.z, v0.z}, a0);
    a1 = __builtin_elementwise_fma(f2{w[1].z, w[1].w}, f2{v0.w, v0.w}, a1);
    a0 = __builtin_elementwise_fma(f2{w[2].x, w[2].y}, f2{v1.x, v1.x}, a0);
    a1 = __builtin_elementwise_fma(f2{w[2].z, w[2].w}, f2{v1.y, v1.y}, a1);
    a0 = __builtin_elementwise_fma(f2{w[3].x, w[3].y}, f2{v1.z, v1.z}, a0);
    a1 = __builtin_elementwise_fma(f2{w[3].z, w[3].w}, f2{v1.w, v1.w}, a1);
    a0 += a1;
    return swap32_sum(a0.x, a0.y);
}

struct StageLds {
    float *hx, *ux, *pz, *po, *ps, *ps2, *pre, *xin, *pt;
    int* flags;
};
__device__ __forceinline__ StageLds carve_stage(float* smem, int kpre) {
    StageLds s;
    s.hx = smem;                           // [512] h_{l-1}[t]
    s.ux = s.hx + RWD;                     // [256] gate outputs u_{l-1}[t]
    s.pz = s.ux + GHD;                     // [8][64] partial z
    s.po = s.pz + 8 * 64;                  // [8][64] partial conv1x1_out
    s.ps = s.po + 8 * 64;                  // [16][32] partial conv1x1_skip
    s.ps2 = s.ps + 16 * 32;                // [16][32] the same for skip bank 1 (K > 256)
    s.pre = s.ps2 + 16 * 32;               // [BMAX][64] pre_j of the step being computed
    s.pt = s.pre + BMAX * 64;              // [8][BMAX][64] partial taps
    s.flags = reinterpret_cast<int*>(s.pt + 8 * BMAX * 64);
    s.xin = reinterpret_cast<float*>(s.flags + 16);      // [xin_slots(B)][kpre] tap inputs of the next step
    (void)kpre;
    return s;
}
__host__ __device__ inline int xin_slots(int B) { return B <= 1 ? 1 : B <= 2 ? 2 : B <= 4 ? 4 : B <= 8 ? 8 : 16; }     // what stream_pre<NB> reads
__host__ __device__ inline size_t stage_lds_floats(int kpre, int B) { return (size_t)RWD + GHD + 4 * 512 + BMAX * 64 + 8 * BMAX * 64 + 16 + (size_t)xin_slots(B) * kpre; }

__device__ __forceinline__ float reduce_quads2(f2 s01, f2 s23) { return swap16_sum(swap32_sum(s01.x, s01.y), swap32_sum(s23.x, s23.y)); }

// the streamed part of compute_pre for NB utterances (utterances >= p.B, if any, are computed on stale LDS and dropped): wave w takes
// the 16-k blocks [kb0, kb1); two blocks of weights in flight
template <int NB>
__device__ __forceinline__ void stream_pre(const WideParams& p, const StageLds& s, int l, int j, int lane, int wave, int b0 = 0) {
    const int nb = p.kpre >> 4, per = (nb + 7) >> 3, kb0 = wave * per, kb1 = min(nb, kb0 + per);
    const float4* W = reinterpret_cast<const float4*>(p.wpre) + ((size_t)(l * PG + j) * nb * 4) * 64 + lane;
    const float* xq = s.xin + (size_t)b0 * p.kpre + 4 * (lane >> 4);
    f2 a01[NB], a23[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { a01[b] = f2{0.f, 0.f}; a23[b] = f2{0.f, 0.f}; }
    float4 w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = W[(size_t)(min(kb0, nb - 1) * 4 + c) * 64];
    for (int kb = kb0; kb < kb1; ++kb) {
        float4 wnx[4];
        const int kn = kb + 1 < kb1 ? kb + 1 : kb;
#pragma unroll
        for (int c = 0; c < 4; ++c) wnx[c] = W[(size_t)(kn * 4 + c) * 64];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 x = *reinterpret_cast<const float4*>(xq + (size_t)b * p.kpre + 16 * kb);
            a01[b] = __builtin_elementwise_fma(f2{w[0].x, w[0].y}, f2{x.x, x.x}, a01[b]);
            a23[b] = __builtin_elementwise_fma(f2{w[0].z, w[0].w}, f2{x.x, x.x}, a23[b]);
            a01[b] = __builtin_elementwise_fma(f2{w[1].x, w[1].y}, f2{x.y, x.y}, a01[b]);
            a23[b] = __builtin_elementwise_fma(f2{w[1].z, w[1].w}, f2{x.y, x.y}, a23[b]);
            a01[b] = __builtin_elementwise_fma(f2{w[2].x, w[2].y}, f2{x.z, x.z}, a01[b]);
            a23[b] = __builtin_elementwise_fma(f2{w[2].z, w[2].w}, f2{x.z, x.z}, a23[b]);
            a01[b] = __builtin_elementwise_fma(f2{w[3].x, w[3].y}, f2{x.w, x.w}, a01[b]);
            a23[b] = __builtin_elementwise_fma(f2{w[3].z, w[3].w}, f2{x.w, x.w}, a23[b]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = wnx[c];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float v = reduce_quads2(a01[b], a23[b]);
        if (b0 + b < p.B) s.pt[((size_t)wave * BMAX + b0 + b) * 64 + lane] = v;
    }
}

// pre_j[tp] for every utterance: older taps of step tp out of this workgroup's own history copy (zeros before t = 0), the
// conditioning row c[tp], then the streamed [kpre][64] matrix (one pass over the weights for all utterances; lane-quad image in
// blocks of 16 k: a lane holds the 4 row slots x the 4 k of its quarter)
__device__ void compute_pre(const WideParams& p, const StageLds& s, int l, int j, int tp, int tid, int lane, int wave, int t_stamp = -1) {
    const int d = p.lay_dil[l], rows = (p.kw - 1) * d, hoff = (p.kw - 1) * RWD;
    const float* hist0 = p.hist + (size_t)PG * p.lay_histoff[l] + (size_t)j * rows * RWD;
    // tap inputs, four utterances' loads in flight together
    for (int b0 = 0; b0 < p.B; b0 += 4) {
        float v[4][3];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int tt = tp - (p.kw - 1 - k) * d;                     // conv.py:43-44: tap k (oldest first) looks d (kw-1-k) back
                const bool on = b0 + bi < p.B && k < p.kw - 1 && tt >= 0;
                v[bi][k] = on ? hist0[(size_t)(b0 + bi) * p.hist_b_floats + (size_t)(tt % rows) * RWD + tid] : 0.f;          // rows this very workgroup stored
            }
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (b0 + bi < p.B && k < p.kw - 1) s.xin[(size_t)(b0 + bi) * p.kpre + k * RWD + tid] = v[bi][k];
    }
    {   // conditioning rows c[tp] (zero-padded to the 16-k block)
        const int nc = p.kpre - hoff, total = p.B * nc;
        for (int i0 = tid; i0 < total; i0 += 4 * WT) {
            float v[4];
            int at[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * WT, b = i / nc, e = i - b * nc;
                at[q] = i < total ? b * p.kpre + hoff + e : -1;
                v[q] = (i < total && e < p.cin) ? p.c_up[((size_t)b * p.T + tp) * p.cin + e] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (at[q] >= 0) s.xin[at[q]] = v[q];
        }
    }
    __syncthreads();
    if (j == 0 && t_stamp >= 0) wstamp(p, p.trace_b, t_stamp, l, 5, 0);     // tap inputs in LDS
    if (p.B == 1) stream_pre<1>(p, s, l, j, lane, wave);
    else if (p.B == 2) stream_pre<2>(p, s, l, j, lane, wave);
    else if (p.B <= 4) stream_pre<4>(p, s, l, j, lane, wave);
    else {
        stream_pre<8>(p, s, l, j, lane, wave);
        if (p.B > 8) stream_pre<8>(p, s, l, j, lane, wave, 8);
    }
    if (j == 0 && t_stamp >= 0) wstamp(p, p.trace_b, t_stamp, l, 6, 0);     // this wave's share of the stream done
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
    WTS_DECL;
    const StageLds s = carve_stage(smem, p.kpre);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = l == 0;
    float4 wn[16], wm[8], wo[8], ws[4], ws2[4];
    load_img<16>(wn, p.wn + (size_t)(l * PG + j) * 8 * 16 * 64 * 4, wave, lane);      // N_l (group 0: W_cur,0) x h_{l-1}, K span 64 wave (lane-quad images)
    load_img<8>(wm, p.wm + (size_t)(l * PG + j) * 8 * 8 * 64 * 4, wave, lane);        // M_l x u_{l-1}, K span 32 wave (group 0: zeros)
    load_img<8>(wo, p.wo + (size_t)(l * PG + j) * 8 * 8 * 64 * 4, wave, lane);        // W_out,l-1 rows 64 j + ..
    load_img<4>(ws, p.ws + (size_t)((l * 2 + 0) * PG + j) * 8 * 4 * 64 * 4, wave, lane);        // W_skip,l-1 rows 32 j + .. (bank 0: skip channels < 256), K span 32 wave
    load_img<4>(ws2, p.ws + (size_t)((l * 2 + 1) * PG + j) * 8 * 4 * 64 * 4, wave, lane);       // ... bank 1: skip channels 256 + 32 j + .. (zeros unless K > 256)
    const float bo_r = p.bo[(size_t)l * RWD + RS * j + lane];                         // b_out,l-1
    const float bs_r = p.bs[(size_t)l * p.KW + (wave == 3 ? KWD : 0) + KS * j + (lane & 31)];        // b_skip,l-1 (wave 2: bank 0, wave 3: bank 1)
    const float cv_a = p.cvec[(size_t)l * 2 * GHD + GS * j + (lane & 31)], cv_g = p.cvec[(size_t)l * 2 * GHD + GHD + GS * j + (lane & 31)];
    const int d = p.lay_dil[l], rows = (p.kw - 1) * d;
    // this workgroup's own copy of layer l's input history, utterance 0
    float* const hist0 = p.hist + (size_t)PG * p.lay_histoff[l] + (size_t)j * rows * RWD;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    compute_pre(p, s, l, j, 0, tid, lane, wave);                  // pre_j[0]: history is zero, conditioning row c[0]

    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        const size_t hrow = rows > 0 ? (size_t)(t % rows) * RWD : 0;
        for (int b = 0; b < p.B; ++b) {
            u64* x_in = p.xmail + ((size_t)b * (p.L + 1) + l) * XW;
            u64* x_out = x_in + XW;
            // ---- gather (u_{l-1}, h_{l-1}) of step t: the chain.  Meanwhile waves 6 and 7 copy the full h_l of the utterance two back
            //      (published two passes ago by the eight workgroups of this group) into this workgroup's history -------------------
            //      (Tried: waves 0 .. 2 publish only and three waves gather 256 granules each, so that the next gather overlaps the
            //      publish -- +3 % at B = 16, -3 % at B <= 8: the two-load poll is slower than the one-load poll.)
            if (wave < 2) {
                if (!first && !recv128(x_in + 128 * wave, tag, s.ux + 128 * wave, p.status, 0x200u + (unsigned)l, lane)) s.flags[0] = 1;
                WTS(7);                                            // this wave's half of u has arrived
            } else if (wave < 6) {
                if (!recv128(x_in + GHD + 128 * (wave - 2), tag, s.hx + 128 * (wave - 2), p.status, 0x100u + (unsigned)l, lane)) s.flags[0] = 1;
                WTS(8);                                            // this wave's quarter of h has arrived
            } else if (!first && b >= 2 && rows > 0) {
                const u64* prev = x_out - (size_t)2 * (p.L + 1) * XW + GHD + 256 * (wave - 6);
                if (!recv256(prev, tag, hist0 + (size_t)(b - 2) * p.hist_b_floats + hrow + 256 * (wave - 6), p.status, 0x600u + (unsigned)l, lane)) s.flags[0] = 1;
            }
            __syncthreads();                                       // (a timed-out gather is noticed at the end of the step: off the chain)
            WTS(0);                                                // inputs gathered
            // ---- one pass: z_l, conv1x1_out of layer l-1, conv1x1_skip of layer l-1 -----------------------------------------------
            const float hp = s.hx[RS * j + lane];                  // (wave 1 needs it after the barrier, when hx may be refilled)
            {
                const int kq = lane >> 4, ps_at = (2 * wave + (kq & 1)) * 32 + (lane & 15) + 16 * (lane >> 5);
                f2 az[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<16>(wn, s.hx + 64 * wave + 16 * kq, az);
                if (!first) {
                    const float* uq = s.ux + 32 * wave + 8 * kq;
                    dot_quad<8>(wm, uq, az);
                    f2 ao[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                    dot_quad<8>(wo, uq, ao);
                    s.po[wave * 64 + lane] = reduce_quads(ao);
                    s.ps[ps_at] = dot_skip(ws, uq);
                    if (p.NB > 1) s.ps2[ps_at] = dot_skip(ws2, uq);
                }
                s.pz[wave * 64 + lane] = reduce_quads(az);
                WTS(10);                                           // this wave's partial sums issued to LDS
            }
            if (first && rows > 0) hist0[(size_t)b * p.hist_b_floats + hrow + tid] = s.hx[tid];        // group 0: the input is its history row
            __syncthreads();
            WTS(1);                                                // partial sums in LDS
            if (wave == 0) {                                       // u_l: lanes c and 32 + c hold the tanh / sigmoid rows of channel 32 j + c
                float v = lds_sum8(s.pre[b * 64 + lane] + (lane < GS ? cv_a : cv_g), s.pz + lane, 64);
#ifdef WNV_FINE_TRACE
                asm volatile("" : "+v"(v));
#endif
                WTS(12);                                           // pre-activation summed
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                if (lane < GS) st_granule(x_out + GS * j + lane, tag, wide_gate(__uint_as_float(r[0]), __uint_as_float(r[1])), fast_next);   // modules.py:152-154
                WTS(2);                                            // u published
            } else if (wave == 1) {                                // h_l = layer l's input (group 0 passes h_0 on)
                const float o = lds_sum8(bo_r, s.po + lane, 64);
                const float hv = first ? hp : (o + hp) * 0.70710678118654752440f;                                        // modules.py:157-162
                st_granule(x_out + GHD + RS * j + lane, tag, hv, fast_next);
                WTS(11);                                           // h published
            } else if ((wave == 2 || (wave == 3 && p.NB > 1)) && !first) {       // skip sum over layers 0 .. l-1 (wavenet.py:312) -> group l + 1 / the tail; one wave per bank
                const float* psb = wave == 2 ? s.ps : s.ps2;
                const size_t bank = wave == 2 ? 0 : KWD;
                float sk = bs_r, acc = 0.f;
                bool ok = true;
                if (lane < KS) sk = lds_sum16(sk, psb + lane, 32);
                if (l > 1) ok = recv_lanes(p.smail + ((size_t)b * (p.L + 2) + l) * p.KW + bank + KS * j, KS, tag, acc, p.status, 0x300u + (unsigned)l, lane);
                if (!ok) s.flags[0] = 1;
                else if (lane < KS) st_granule(p.smail + ((size_t)b * (p.L + 2) + l + 1) * p.KW + bank + KS * j + lane, tag, acc + sk, fast_next);
            }
            // slots: 0 inputs gathered | 1 partial sums in LDS | 2 u published | 7 u arrived (wave 0's half) | 8 h arrived (wave 2's quarter) |
            // 10 wave 0's partial sums issued | 11 h published | 12 pre-activation summed
            if (j == 0) {
                if (wave == 0) WTS_FLUSH(b, t, l, 0x1487u);
                else if (wave == 1) WTS_FLUSH(b, t, l, 0x0800u);
                else if (wave == 2) WTS_FLUSH(b, t, l, 0x0100u);
            }
        }
        // ---- behind the chain: the full h_l of the last two utterances -> history (waves 6, 7), then pre_j[t + 1] for every utterance
        //      (the history rows are this CU's own stores).  Memory traffic of this CU right after the publish -- polling for the own
        //      h_l, or the tap stream -- slows the hop to the next group (measured: +4 us per step over 24 groups), hence the pause
        //      where the deferred copies of the batch loop do not provide one.
        if (!first && rows > 0 && wave >= 6) {
            if (p.B <= 2) { __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(64); }
            for (int bb = max(0, p.B - 2); bb < p.B; ++bb) {
                const u64* own = p.xmail + ((size_t)bb * (p.L + 1) + l + 1) * XW + GHD + 256 * (wave - 6);
                if (!recv256(own, tag, hist0 + (size_t)bb * p.hist_b_floats + hrow + 256 * (wave - 6), p.status, 0x600u + (unsigned)l, lane)) s.flags[0] = 1;
            }
        }
        __syncthreads();
        if (s.flags[0]) return;
        if (j == 0) wstamp(p, p.trace_b, t, l, 3, 0);              // own h stored
        if (t + 1 < p.T) compute_pre(p, s, l, j, t + 1, tid, lane, wave, t);
        if (j == 0) wstamp(p, p.trace_b, t, l, 4, 0);                      // next step's pre-activations ready
    }
}

// ---- the tail group: conv1x1_skip of the LAST layer (8 workgroups, 32 skip rows each) + the running skip sum -> the head ------------
__device__ void run_wide_tail(const WideParams& p, int j, bool fast_next, float* smem) {
    float* ux = smem;                                              // [256] u_{L-1}
    float* ps = ux + GHD;                                          // [2 banks][16][32] partial sums
    int* flags = reinterpret_cast<int*>(ps + 2 * 16 * 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 ws[4], ws2[4];
    load_img<4>(ws, p.wsl + (size_t)j * 8 * 4 * 64 * 4, wave, lane);
    load_img<4>(ws2, p.wsl + (size_t)(PG + j) * 8 * 4 * 64 * 4, wave, lane);            // bank 1 (zeros unless K > 256)
    const float bs_r = p.bs[(size_t)p.L * p.KW + (wave == 1 ? KWD : 0) + KS * j + (lane & 31)];
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (wave < 2) {
                if (!recv128(p.xmail + ((size_t)b * (p.L + 1) + p.L) * XW + 128 * wave, tag, ux + 128 * wave, p.status, 0x500u, lane)) flags[0] = 1;
            }
            __syncthreads();
            if (flags[0]) return;
            {
                const int at = (2 * wave + ((lane >> 4) & 1)) * 32 + (lane & 15) + 16 * (lane >> 5);
                ps[at] = dot_skip(ws, ux + 32 * wave + 8 * (lane >> 4));
                if (p.NB > 1) ps[512 + at] = dot_skip(ws2, ux + 32 * wave + 8 * (lane >> 4));
            }
            __syncthreads();
            if (wave == 0 || (wave == 1 && p.NB > 1)) {            // one wave per bank
                const float* psb = ps + 512 * wave;
                const size_t bank = (size_t)KWD * wave;
                float sk = bs_r, acc = 0.f;
                bool ok = true;
                if (lane < KS) sk = lds_sum16(sk, psb + lane, 32);
                if (p.L > 1) ok = recv_lanes(p.smail + ((size_t)b * (p.L + 2) + p.L) * p.KW + bank + KS * j, KS, tag, acc, p.status, 0x300u + (unsigned)p.L, lane);
                if (!ok) flags[0] = 1;
                else if (lane < KS) st_granule(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * p.KW + bank + KS * j + lane, tag, acc + sk, fast_next);
            }
        }
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
    load_img<32>(w1, p.wh1, wave, lane);                 // lane-quad image: rows (wave & 3) 64 + .., K half (wave >> 2)
    load_img<8>(w2, p.wh2, wave, lane);                  // lane-quad image: rows 0 .. 63 (< O), K span 32 wave
    const float wf = p.wfirst[tid], bf = p.bfirst[tid];
    const float b1 = tid < KWD ? p.bh1[tid] : 0.f;
    const float b2 = lane < p.O ? p.bh2[lane] : 0.f;
    // output distribution (mixture.py:118-156 / :221-270): which head outputs are mixture logits / mean / log-scale
    const bool single = p.dist == 2 && p.O <= 3;
    const int nmix = single ? 0 : p.O / 3;
    const int o_mean = single ? (p.O == 2 ? 0 : 1) : nmix, o_ls = single ? (p.O == 2 ? 1 : 2) : 2 * nmix;
    const int nchunk = (nmix + 3) >> 2;
    float* vbuf = s.nz + 32;                             // [32] mixture logit + Gumbel noise, padded with -inf
    if (tid < 32) vbuf[tid] = -INFINITY;
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
            // ---- everything that does not depend on the network, while the groups work: the noise terms of the sampler ----------
            if (tid < p.nz) {
                const int kind = (p.dist == 2 && tid == p.nz - 1) ? 1 : 0;
                const float r = p.noise ? p.noise[((size_t)t * p.noise_B + p.b0 + b) * p.nz + tid] : wnv_noise_gen(p.seed, t, p.b0 + b, tid, kind);
                if (tid < nmix) s.nz[tid] = -logf(-logf(r));                                         // Gumbel noise (mixture.py:138-140)
                else s.nz[tid] = p.dist == 1 ? logf(r) - logf(1.0f - r) : r;                         // mixture.py:151-152 / :265-267
            }
            if (wave < 2) {
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * p.KW + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
                // (a lane rewrites the two values it has just stored)
                float2* v2 = reinterpret_cast<float2*>(s.vs + 128 * wave + 2 * lane);
                *v2 = make_float2(fmaxf(v2->x * p.skip_scale, 0.f), fmaxf(v2->y * p.skip_scale, 0.f));     // wavenet.py:313-316
            }
            __syncthreads();
            if (s.flags[0]) return;
            wstamp(p, b, t, p.L, 0, 0);                                                             // skip sum gathered
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<32>(w1, s.vs + 128 * (wave >> 2) + 32 * (lane >> 4), a);
                s.ph[(wave >> 2) * KWD + (wave & 3) * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (tid < KWD) s.hid[tid] = fmaxf(s.ph[tid] + s.ph[KWD + tid] + b1, 0.f);               // wavenet.py:317-318
            __syncthreads();
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<8>(w2, s.hid + 32 * wave + 8 * (lane >> 4), a);
                s.pout[wave * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (wave == 0 && lane < p.O) {
                const float o = lds_sum8(b2, s.pout + lane, 64);
                s.obuf[lane] = o;                                                                    // wavenet.py:319
                if (lane < nmix) vbuf[lane] = o + s.nz[lane];
                if (p.params_out) p.params_out[((size_t)b * p.O + lane) * p.T + t] = o;
            }
            __syncthreads();
            // ---- sample in every wave on its own (all 512 threads feed first_conv), then first_conv of step t + 1.  Up to 16 mixture
            //      components: lane c evaluates component c (independent LDS reads, one round trip), the Gumbel-max is a 16-lane DPP max
            //      butterfly + ballot (first index wins ties), v_readlane fetches the winner's sample (as wnv_ring.hip's head) ----------------
            {
                float xo;
                const float lr = s.nz[nmix];
                if (nmix <= 16) {
                    float key = -INFINITY, mean = 0.f, ls = 0.f;
                    if (nmix == 0) { mean = s.obuf[o_mean]; ls = s.obuf[o_ls]; }                     // mixture.py:258-261
                    else if (lane < nmix) { key = vbuf[lane]; mean = s.obuf[o_mean + lane]; ls = s.obuf[o_ls + lane]; }       // mixture.py:143-146
                    float xc = p.dist == 1 ? mean + __expf(ls) * lr : lr * __expf(ls) + mean;
                    xc = fminf(fmaxf(xc, -1.0f), 1.0f);                                              // mixture.py:154 / :269
                    if (nmix > 0) {
                        float m = key;
                        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xF, 0xF, true)));
                        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xF, 0xF, true)));
                        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xF, 0xF, true)));
                        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x140, 0xF, 0xF, true)));
                        const unsigned long long win = __ballot(lane < nmix && key == m);
                        const int wl = win ? __ffsll((long long)win) - 1 : 0;
                        xo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xc), wl));
                    } else {
                        xo = xc;
                    }
                } else {
                    int bi = 0;
                    float best = -INFINITY;
                    for (int c = 0; c < nchunk; ++c) {
                        const float4 v = reinterpret_cast<const float4*>(vbuf)[c];
                        if (v.x > best) { best = v.x; bi = 4 * c; }
                        if (v.y > best) { best = v.y; bi = 4 * c + 1; }
                        if (v.z > best) { best = v.z; bi = 4 * c + 2; }
                        if (v.w > best) { best = v.w; bi = 4 * c + 3; }
                    }
                    const float mean = s.obuf[o_mean + bi], ls = s.obuf[o_ls + bi];
                    xo = p.dist == 1 ? mean + expf(ls) * lr : lr * expf(ls) + mean;
                    xo = fminf(fmaxf(xo, -1.0f), 1.0f);
                }
                if (t + 1 < p.T) {
                    const float xs = t + 1 < p.Tt ? p.teacher[(size_t)b * p.Tt + t + 1] : xo;        // wavenet.py:297-305
                    st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, tag + 1u, fmaf(wf, xs, bf), fast_first);
                }
                if (tid == 0) p.out[(size_t)b * p.T + t] = xo;
            }
            wstamp(p, b, t, p.L, 1, 0);                                                             // next input sent
            __syncthreads();                                                                        // obuf / vbuf / nz are free again
        }
    }
}

// ---- head of scalar-input models with 257 .. 512 skip channels (the reference constructor's default geometry, wavenet.py:98-101): the
// hidden layer (512 x 512 = 1 MB) is four register files, so the head is FOUR workgroups ("parts", as in wnv_ring.hip): part q owns
// hidden units [128 q, 128 q + 128) -- it reads the whole skip sum, computes its hidden slice (W1 rows in VGPRs, 128 floats per thread)
// and the partial head outputs W2[:, 128 q ..] . hidden_q; parts 1 .. 3 send their partials to part 0, which adds them in part order,
// samples and feeds group 0.
struct Head5Lds {
    float *vs, *ph, *hid, *pout, *obuf, *nz;
    int* flags;
};
__device__ __forceinline__ Head5Lds carve_head5(float* smem) {
    Head5Lds s;
    s.vs = smem; s.ph = s.vs + 512; s.hid = s.ph + 4 * 128; s.pout = s.hid + 128; s.obuf = s.pout + 8 * 64; s.nz = s.obuf + 64;
    s.flags = reinterpret_cast<int*>(s.nz + 64);
    return s;
}
constexpr size_t HEAD5_LDS_FLOATS = 512 + 4 * 128 + 128 + 8 * 64 + 64 + 64 + 16;

__device__ void run_wide_head512(const WideParams& p, int q, bool fast_first, bool fast_parts, float* smem) {
    const Head5Lds s = carve_head5(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 w1[32], w2[4];
    load_img<32>(w1, p.wh1 + (size_t)q * 8 * 32 * 64 * 4, wave, lane);      // lane-quad image: hidden rows 128 q + (wave & 1) 64 + .., K quarter (wave >> 1)
    load_img<4>(w2, p.wh2 + (size_t)q * 8 * 4 * 64 * 4, wave, lane);        // lane-quad image: output rows 0 .. 63 (< O), hidden span 128 q + 16 wave + ..
    const float wf = p.wfirst[tid], bf = p.bfirst[tid];
    const float b1 = tid < 128 ? p.bh1[128 * q + tid] : 0.f;
    const float b2 = lane < p.O ? p.bh2[lane] : 0.f;
    const bool single = p.dist == 2 && p.O <= 3;
    const int nmix = single ? 0 : p.O / 3;
    const int o_mean = single ? (p.O == 2 ? 0 : 1) : nmix, o_ls = single ? (p.O == 2 ? 1 : 2) : 2 * nmix;
    float* vbuf = s.nz + 32;                             // [32] mixture logit + Gumbel noise, padded with -inf
    if (tid < 32) vbuf[tid] = -INFINITY;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    if (q == 0)                                          // the input of step 0 (wavenet.py:283-289, :297-308)
        for (int b = 0; b < p.B; ++b) {
            const float xs = p.Tt > 0 ? p.teacher[(size_t)b * p.Tt] : (p.initial ? p.initial[b] : 0.f);
            st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, p.tag_base + 1u, fmaf(wf, xs, bf), fast_first);
        }
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (q == 0 && tid < p.nz) {                  // the noise terms of the sampler, while the groups work
                const int kind = (p.dist == 2 && tid == p.nz - 1) ? 1 : 0;
                const float r = p.noise ? p.noise[((size_t)t * p.noise_B + p.b0 + b) * p.nz + tid] : wnv_noise_gen(p.seed, t, p.b0 + b, tid, kind);
                if (tid < nmix) s.nz[tid] = -logf(-logf(r));                                         // Gumbel noise (mixture.py:138-140)
                else s.nz[tid] = p.dist == 1 ? logf(r) - logf(1.0f - r) : r;                         // mixture.py:151-152 / :265-267
            }
            if (wave < 4) {                              // the whole skip sum (512 channels), every part
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * p.KW + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
                float2* v2 = reinterpret_cast<float2*>(s.vs + 128 * wave + 2 * lane);
                *v2 = make_float2(fmaxf(v2->x * p.skip_scale, 0.f), fmaxf(v2->y * p.skip_scale, 0.f));     // wavenet.py:313-316
            }
            __syncthreads();
            if (s.flags[0]) return;
            if (q == 0) wstamp(p, b, t, p.L, 0, 0);                                                 // skip sum gathered
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<32>(w1, s.vs + 128 * (wave >> 1) + 32 * (lane >> 4), a);
                s.ph[(wave >> 1) * 128 + (wave & 1) * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (tid < 128) s.hid[tid] = fmaxf(((s.ph[tid] + s.ph[128 + tid]) + (s.ph[256 + tid] + s.ph[384 + tid])) + b1, 0.f);      // wavenet.py:317-318
            __syncthreads();
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<4>(w2, s.hid + 16 * wave + 4 * (lane >> 4), a);
                s.pout[wave * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (wave == 0) {
                float o = 0.f;
                if (lane < p.O) o = lds_sum8(o, s.pout + lane, 64);
                if (q > 0) {
                    if (lane < p.O) st_granule(p.omail + ((size_t)b * 4 + q) * 64 + lane, tag, o, fast_parts);
                } else {
                    bool ok = true;
                    for (int part = 1; part < 4 && ok; ++part) {                                     // partial outputs, in part order
                        float v = 0.f;
                        ok = recv_lanes(p.omail + ((size_t)b * 4 + part) * 64, p.O, tag, v, p.status, 0x480u + (unsigned)part, lane);
                        o += v;
                    }
                    if (!ok) s.flags[0] = 1;
                    o += b2;
                    if (lane < p.O) {
                        s.obuf[lane] = o;                                                            // wavenet.py:319
                        if (lane < nmix) vbuf[lane] = o + s.nz[lane];
                        if (p.params_out) p.params_out[((size_t)b * p.O + lane) * p.T + t] = o;
                    }
                }
            }
            if (q > 0) continue;                         // (parts 1 .. 3: back to the next skip sum; their LDS is fenced by the barriers above)
            __syncthreads();
            if (s.flags[0]) return;
            {   // sample in every wave on its own (all 512 threads feed first_conv), then first_conv of step t + 1 (as run_wide_head)
                float xo;
                const float lr = s.nz[nmix];
                float key = -INFINITY, mean = 0.f, ls = 0.f;
                if (nmix == 0) { mean = s.obuf[o_mean]; ls = s.obuf[o_ls]; }                         // mixture.py:258-261
                else if (lane < nmix && lane < 16) { key = vbuf[lane]; mean = s.obuf[o_mean + lane]; ls = s.obuf[o_ls + lane]; }     // mixture.py:143-146
                float xc = p.dist == 1 ? mean + __expf(ls) * lr : lr * __expf(ls) + mean;
                xc = fminf(fmaxf(xc, -1.0f), 1.0f);                                                  // mixture.py:154 / :269
                if (nmix > 0) {
                    float m = key;
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xF, 0xF, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xF, 0xF, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xF, 0xF, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x140, 0xF, 0xF, true)));
                    const unsigned long long win = __ballot(lane < nmix && lane < 16 && key == m);
                    const int wl = win ? __ffsll((long long)win) - 1 : 0;
                    xo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xc), wl));
                } else {
                    xo = xc;
                }
                if (t + 1 < p.T) {
                    const float xs = t + 1 < p.Tt ? p.teacher[(size_t)b * p.Tt + t + 1] : xo;        // wavenet.py:297-305
                    st_granule(p.xmail + ((size_t)b * (p.L + 1)) * XW + GHD + tid, tag + 1u, fmaf(wf, xs, bf), fast_first);
                }
                if (tid == 0) p.out[(size_t)b * p.T + t] = xo;
            }
            wstamp(p, b, t, p.L, 1, 0);                                                             // next input sent
            __syncthreads();                                                                        // obuf / vbuf / nz are free again
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
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * p.KW + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
                // (a lane rewrites the two values it has just stored)
                float2* v2 = reinterpret_cast<float2*>(s.vs + 128 * wave + 2 * lane);
                *v2 = make_float2(fmaxf(v2->x * p.skip_scale, 0.f), fmaxf(v2->y * p.skip_scale, 0.f));     // wavenet.py:313-316
            }
            __syncthreads();
            if (s.flags[0]) return;
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<32>(w1, s.vs + 128 * (wave >> 2) + 32 * (lane >> 4), a);
                s.ph[(wave >> 2) * KWD + (wave & 3) * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (tid < KWD) st_granule(p.hidmail + (size_t)b * KWD + tid, tag, fmaxf(s.ph[tid] + s.ph[KWD + tid] + b1, 0.f), fast);   // wavenet.py:317-318
            __syncthreads();
        }
    }
}

// one-hot models with 257 .. 512 skip channels: part A is FOUR workgroups (hidden units [128 q, 128 q + 128) each, as run_wide_head512),
// part B two (the output layer's K halves; B1 sends its partial logits to B0)
__device__ void run_wide_head512a(const WideParams& p, int q, float* smem) {
    const Head5Lds s = carve_head5(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 w1[32];
    load_img<32>(w1, p.wh1 + (size_t)q * 8 * 32 * 64 * 4, wave, lane);
    const float b1 = tid < 128 ? p.bh1[128 * q + tid] : 0.f;
    const bool fast = p.fast != 0;                         // the B parts sit on the same XCD
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (wave < 4) {
                if (!recv128(p.smail + ((size_t)b * (p.L + 2) + p.L + 1) * p.KW + 128 * wave, tag, s.vs + 128 * wave, p.status, 0x400u, lane)) s.flags[0] = 1;
                float2* v2 = reinterpret_cast<float2*>(s.vs + 128 * wave + 2 * lane);
                *v2 = make_float2(fmaxf(v2->x * p.skip_scale, 0.f), fmaxf(v2->y * p.skip_scale, 0.f));     // wavenet.py:313-316
            }
            __syncthreads();
            if (s.flags[0]) return;
            {
                f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                dot_quad<32>(w1, s.vs + 128 * (wave >> 1) + 32 * (lane >> 4), a);
                s.ph[(wave >> 1) * 128 + (wave & 1) * 64 + lane] = reduce_quads(a);
            }
            __syncthreads();
            if (tid < 128)                                                                              // wavenet.py:317-318
                st_granule(p.hidmail + (size_t)b * p.KW + 128 * q + tid, tag, fmaxf(((s.ph[tid] + s.ph[128 + tid]) + (s.ph[256 + tid] + s.ph[384 + tid])) + b1, 0.f), fast);
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

__device__ void run_wide_head_b(const WideParams& p, bool fast_first, int part, int nparts, float* smem) {
    const CatLds s = carve_cat(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int O = p.O;
    float4 w2[32];                                         // rows lane + 64 q (q = 0 .. 3), K chunk 32 wave: [wave][q * 8 + c][lane][4]
    load_img<32>(w2, p.wh2 + (size_t)part * 8 * 32 * 64 * 4, wave, lane);      // (part: the hidden half [256 part, 256 part + 256) of a 512-wide hidden layer)
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
    for (int b = 0; b < p.B && part == 0; ++b) {           // wavenet.py:283-289: one-hot of class 127 unless given
        const float* dense = p.Tt > 0 ? p.teacher + (size_t)b * p.Tt * O : (p.initial ? p.initial + (size_t)b * O : nullptr);
        send_input(b, dense, 127, p.tag_base + 1u);
    }
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int b = 0; b < p.B; ++b) {
            if (part == 0 && tid < O) s.nz[tid] = p.noise ? p.noise[((size_t)t * p.noise_B + p.b0 + b) * p.nz + tid] : wnv_noise_gen(p.seed, t, p.b0 + b, tid, 2);   // e ~ Exp(1)
            if (wave < 2) {
                if (!recv128(p.hidmail + (size_t)b * p.KW + 256 * part + 128 * wave, tag, s.hid + 128 * wave, p.status, 0x480u, lane)) s.ints[0] = 1;
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
            if (part > 0) {                                        // the other K half: partial logits -> part 0
                if (tid < O) {
                    const float o = lds_sum8(0.f, s.pout + tid, 256);
                    st_granule(p.omail + (size_t)b * 256 + tid, tag, o, p.fast != 0);
                }
                __syncthreads();
                continue;
            }
            {
                float o = b2, other = 0.f;
                if (tid < O) o = lds_sum8(o, s.pout + tid, 256);
                if (nparts > 1 && wave < 4 && 64 * wave < O) {
                    if (!recv_lanes(p.omail + (size_t)b * 256 + 64 * wave, min(64, O - 64 * wave), tag, other, p.status, 0x490u, lane)) s.ints[0] = 1;
                    o += other;
                }
                if (tid < O) {
                    s.obuf[tid] = o;                                                                 // wavenet.py:319
                    if (p.params_out) p.params_out[((size_t)b * O + tid) * p.T + t] = o;
                }
            }
            __syncthreads();
            if (s.ints[0]) return;
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
    if (p.NB > 1 && x == p.head_x && li >= p.head_li && li < p.head_li + 4) {     // 257 .. 512 skip channels: four head parts on one XCD
        if (p.cin1 == 1) run_wide_head512(p, li - p.head_li, p.fast && p.head_x == 0, p.fast != 0, smem);
        else run_wide_head512a(p, li - p.head_li, smem);
        return;
    }
    if (p.NB > 1 && p.cin1 > 1 && x == p.head_x && (li == p.head_li + 4 || li == p.head_li + 5)) {     // ... and, one-hot, the two output-layer parts
        run_wide_head_b(p, p.fast && p.head_x == 0, li - p.head_li - 4, 2, smem);
        return;
    }
    if (x == p.head_x && li == p.head_li) {
        if (p.cin1 == 1) run_wide_head(p, p.fast && p.head_x == 0, smem);      // group 0 lives on XCD 0
        else run_wide_head_a(p, smem);
        return;
    }
    if (p.cin1 > 1 && x == p.head_x && li == p.head_li + 1) {
        run_wide_head_b(p, p.fast && p.head_x == 0, 0, 1, smem);
        return;
    }
    const int sl = li / PG, j = li % PG, l = x * p.nL + sl;         // groups 0 .. L-1: the layers; group L: the tail
    if (sl >= p.nL || l > p.L) return;
    // who reads what this group publishes: group l + 1 (same XCD unless this is the last group of its XCD); the tail feeds the head
    if (l == p.L) run_wide_tail(p, j, p.fast && p.head_x == x, smem);
    else run_wide_stage(p, l, j, p.fast && (l + 1) / p.nL == x, smem);
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================
struct WnvWideState {
    int device = 0;
    int L = 0, O = 0, cin = 0, cinp = 0, kw = 0, kpre = 0, nkb = 0, cin1 = 1, NB = 1;
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
    if (c.skip_out_channels > 2 * KWD) return "needs skip_out_channels <= 512";
    if (c.skip_out_channels > KWD && c.scalar_input && c.out_channels > 48) return "scalar-input models with more than 256 skip channels need at most 16 mixture components";
    if (c.kernel_size < 2 || c.kernel_size > 4) return "needs 2 <= kernel_size <= 4";
    if (c.cin_channels > 128) return "needs cin_channels <= 128";
    if (c.layers > 30) return "needs layers <= 30 (8 workgroups per layer and 8 for the tail, 32 CUs per XCD, one or two more for the head)";
    (void)B;                                           // any batch: the host runs it in slices of 16 utterances
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
    const int L = c.layers, kw = c.kernel_size, cin = c.cin_channels > 0 ? c.cin_channels : 0, cinp = (cin + 15) & ~15;
    const int Ra = c.residual_channels, Ga = c.gate_channels, Gha = Ga / 2, Ka = c.skip_out_channels, O = c.out_channels;
    st->L = L; st->O = O; st->cin = cin; st->cinp = cinp; st->kw = kw;
    const int NB = Ka > KWD ? 2 : 1, KW = KWD * NB;                  // skip banks of 256 channels
    st->NB = NB;
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
    st->o_ws = alloc((size_t)L * 2 * PG * n_ws);                    // [layer][bank][slice]
    st->o_wsl = alloc((size_t)2 * PG * n_ws);                       // [bank][slice]
    st->o_bo = alloc((size_t)L * RWD);
    st->o_bs = alloc((size_t)(L + 1) * KW);
    st->o_cvec = alloc((size_t)L * 2 * GHD);
    st->o_wpre = alloc((size_t)L * PG * n_pre);
    std::vector<int> dil(L), hoff(L);
    long long hist = 0;
    const int per = L / c.stacks;
    const double rs = std::sqrt(0.5);
    std::vector<float> cur((size_t)2 * GHD * RWD), mmat((size_t)2 * GHD * GHD);
    std::vector<double> accd(GHD);
    // skip image of layer `ls` for slice j
    // lane-quad images (dot_quad / dot_skip): lane = (column i = lane & 15, K quarter q = lane >> 4); row slot e of a lane
    auto quad_row = [&](int lane, int e) { static const int g[4] = {0, 2, 1, 3}; return 16 * g[e] + (lane & 15); };
    auto put_skip = [&](float* is, const HostTensor& wsk, int j, int bank) {
        for (int w = 0; w < 8; ++w)
            for (int lane = 0; lane < 64; ++lane)
                for (int cq = 0; cq < 4; ++cq)
                    for (int e = 0; e < 4; ++e) {
                        const int so = KWD * bank + KS * j + 16 * (e & 1) + (lane & 15);           // skip row
                        const int k = 32 * w + 8 * (lane >> 4) + 2 * cq + (e >> 1);
                        is[(((size_t)w * 4 + cq) * 64 + lane) * 4 + e] = (so < Ka && k < Gha) ? wsk.data[(size_t)so * Gha + k] : 0.f;
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
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int r = quad_row(lane, e), o = slice_row(j, r), ro = RS * j + r, q = lane >> 4;
                        for (int cq = 0; cq < 16; ++cq) {               // N_l, K span [64 w, 64 w + 64) of h_{l-1}
                            const int k = 64 * w + 16 * q + cq;
                            const float v = cur[(size_t)o * RWD + k];
                            in_[(((size_t)w * 16 + cq) * 64 + lane) * 4 + e] = l == 0 ? v : (float)(rs * (double)v);
                        }
                        for (int cq = 0; cq < 8; ++cq) {                // M_l and W_out,l-1, K span [32 w, 32 w + 32) of u_{l-1}
                            const int k = 32 * w + 8 * q + cq;
                            im[(((size_t)w * 8 + cq) * 64 + lane) * 4 + e] = mmat[(size_t)o * GHD + k];
                            io[(((size_t)w * 8 + cq) * 64 + lane) * 4 + e] = (l > 0 && ro < Ra && k < Gha) ? wout->data[(size_t)ro * Gha + k] : 0.f;
                        }
                    }
            for (int bank = 0; bank < NB; ++bank) {
                if (l > 0) put_skip(blob.data() + st->o_ws + (size_t)((l * 2 + bank) * PG + j) * n_ws, *wsk, j, bank);
                if (l == L - 1) put_skip(blob.data() + st->o_wsl + (size_t)(bank * PG + j) * n_ws, T(pfx + "conv1x1_skip.weight"), j, bank);
            }
            for (int kb = 0; kb < st->kpre / 16; ++kb)                  // older taps (oldest first) then local conditioning: lane-quad blocks of 16 k, [kb][4][lane][4]
                for (int cq = 0; cq < 4; ++cq)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int go = gate_row(slice_row(j, quad_row(lane, e)));
                            const int k = 16 * kb + 4 * (lane >> 4) + cq;
                            float v = 0.f;
                            if (go >= 0) {
                                if (k < (kw - 1) * RWD) {
                                    const int tap = k >> 9, ch = k & (RWD - 1);
                                    if (ch < Ra) v = wc.data[((size_t)go * Ra + ch) * kw + tap];
                                } else if (k - (kw - 1) * RWD < cin) {
                                    v = wcc->data[(size_t)go * cin + (k - (kw - 1) * RWD)];
                                }
                            }
                            ip[(((size_t)kb * 4 + cq) * 64 + lane) * 4 + e] = v;
                        }
        }
        if (l > 0) {
            std::copy(bout->data.begin(), bout->data.end(), blob.begin() + st->o_bo + (size_t)l * RWD);
            const HostTensor& bs = T(ppx + "conv1x1_skip.bias");
            std::copy(bs.data.begin(), bs.data.end(), blob.begin() + st->o_bs + (size_t)l * KW);
        }
        if (l == L - 1) {
            const HostTensor& bs = T(pfx + "conv1x1_skip.bias");
            std::copy(bs.data.begin(), bs.data.end(), blob.begin() + st->o_bs + (size_t)L * KW);
        }
        dil[l] = 1 << (l % per);
        hoff[l] = (int)hist;
        hist += (long long)(kw - 1) * dil[l] * RWD;
    }
    st->hist_layer_floats = hist;
    // head: W1 rows (w & 3) 64 + .., K half (w >> 2), lane-quad -> [w][32][lane][4];  W2: scalar models rows 0 .. 63, K span 32 w, lane-quad -> [w][8][lane][4];
    // one-hot models rows lane + 64 q (q = 0 .. 3), K chunk 32 w -> [w][8 q + c][lane][4]
    const int cin1 = c.scalar_input ? 1 : O;
    st->cin1 = cin1;
    st->o_wh1 = alloc((size_t)(NB > 1 ? 4 : 1) * 8 * 32 * 64 * 4);
    st->o_wh2 = alloc(NB > 1 ? (cin1 > 1 ? (size_t)2 * 8 * 32 * 64 * 4 : (size_t)4 * 8 * 4 * 64 * 4) : (size_t)8 * (cin1 > 1 ? 32 : 8) * 64 * 4);
    if (NB > 1) {
        // four head parts (run_wide_head512): part q: W1 rows 128 q + (w & 1) 64 + .., K quarter (w >> 1), lane-quad -> [q][w][32][lane][4];
        // W2 rows 0 .. 63, hidden span 128 q + 16 w + .., lane-quad -> [q][w][4][lane][4]
        const HostTensor& w1 = T("last_conv_layers.1.weight");         // (K, K, 1)
        const HostTensor& w2 = T("last_conv_layers.3.weight");         // (O, K, 1)
        for (int q = 0; q < 4; ++q)
            for (int w = 0; w < 8; ++w)
                for (int lane = 0; lane < 64; ++lane) {
                    for (int cq = 0; cq < 32; ++cq)
                        for (int e = 0; e < 4; ++e) {
                            const int row = 128 * q + (w & 1) * 64 + quad_row(lane, e), k = 128 * (w >> 1) + 32 * (lane >> 4) + cq;
                            blob[st->o_wh1 + ((((size_t)q * 8 + w) * 32 + cq) * 64 + lane) * 4 + e] = (row < Ka && k < Ka) ? w1.data[(size_t)row * Ka + k] : 0.f;
                        }
                    if (cin1 == 1) {
                        for (int cq = 0; cq < 4; ++cq)
                            for (int e = 0; e < 4; ++e) {
                                const int orow = quad_row(lane, e), k = 128 * q + 16 * w + 4 * (lane >> 4) + cq;
                                blob[st->o_wh2 + ((((size_t)q * 8 + w) * 4 + cq) * 64 + lane) * 4 + e] = (orow < O && k < Ka) ? w2.data[(size_t)orow * Ka + k] : 0.f;
                            }
                    } else if (q < 2) {                                 // one-hot: output-layer part q = hidden half [256 q, 256 q + 256): rows lane + 64 r, K chunk 32 w
                        for (int r = 0; r < 4; ++r)
                            for (int cq = 0; cq < 8; ++cq)
                                for (int e = 0; e < 4; ++e) {
                                    const int k = 256 * q + 32 * w + 4 * cq + e, orow = lane + 64 * r;
                                    blob[st->o_wh2 + ((((size_t)q * 8 + w) * 32 + 8 * r + cq) * 64 + lane) * 4 + e] = (orow < O && k < Ka) ? w2.data[(size_t)orow * Ka + k] : 0.f;
                                }
                    }
                }
    } else {
        const HostTensor& w1 = T("last_conv_layers.1.weight");         // (K, K, 1)
        const HostTensor& w2 = T("last_conv_layers.3.weight");         // (O, K, 1)
        for (int w = 0; w < 8; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                for (int cq = 0; cq < 32; ++cq)                         // W1, lane-quad: rows (w & 3) 64 + .., K half (w >> 2)
                    for (int e = 0; e < 4; ++e) {
                        const int row = (w & 3) * 64 + quad_row(lane, e), k = 128 * (w >> 2) + 32 * (lane >> 4) + cq;
                        blob[st->o_wh1 + (((size_t)w * 32 + cq) * 64 + lane) * 4 + e] = (row < Ka && k < Ka) ? w1.data[(size_t)row * Ka + k] : 0.f;
                    }
                if (cin1 == 1) {                                        // W2 of scalar models, lane-quad: rows 0 .. 63, K span 32 w
                    for (int cq = 0; cq < 8; ++cq)
                        for (int e = 0; e < 4; ++e) {
                            const int orow = quad_row(lane, e), k = 32 * w + 8 * (lane >> 4) + cq;
                            blob[st->o_wh2 + (((size_t)w * 8 + cq) * 64 + lane) * 4 + e] = (orow < O && k < Ka) ? w2.data[(size_t)orow * Ka + k] : 0.f;
                        }
                } else {
                    for (int q = 0; q < 4; ++q)
                        for (int cq = 0; cq < 8; ++cq)
                            for (int e = 0; e < 4; ++e) {
                                const int k = 32 * w + 4 * cq + e, orow = lane + 64 * q;
                                blob[st->o_wh2 + (((size_t)w * 32 + 8 * q + cq) * 64 + lane) * 4 + e] = (orow < O && k < Ka) ? w2.data[(size_t)orow * Ka + k] : 0.f;
                            }
                }
            }
    }
    st->o_bh1 = alloc(KW);
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
    if (ga.B > BMAX) {
        // more utterances than the groups pipeline at once: slices of BMAX, one launch each (utterances are independent; the noise tape
        // and the Philox stream are addressed with the utterance's index in the whole call, so the samples do not depend on the slicing)
        const int cin1 = st->cin1, cin = st->cin, O = st->O;
        for (int b0 = 0; b0 < ga.B; b0 += BMAX) {
            WnvGenArgs g = ga;
            g.B = std::min(BMAX, ga.B - b0);
            g.b0 = ga.b0 + b0; g.noise_B = ga.noise_B > 0 ? ga.noise_B : ga.B;
            if (ga.c_up) g.c_up = ga.c_up + (size_t)b0 * ga.T * cin;
            if (ga.initial) g.initial = ga.initial + (size_t)b0 * cin1;
            if (ga.teacher) g.teacher = ga.teacher + (size_t)b0 * ga.Tt * cin1;
            if (ga.zbias_bstride != 0) g.zbias = ga.zbias + (size_t)b0 * ga.zbias_bstride;
            if (ga.out) g.out = ga.out + (size_t)b0 * cin1 * ga.T;
            if (ga.params_out) g.params_out = ga.params_out + (size_t)b0 * O * ga.T;
            if (ga.index_out) g.index_out = ga.index_out + (size_t)b0 * ga.T;
            const wnv_status s0 = wnv_wide_generate(pst, device, c, store, g, stream, err);
            if (s0 != WNV_OK) return s0;
        }
        return WNV_OK;
    }
    const int B = ga.B, L = st->L;
    if (!st->map_ok || st->n_xcd != 8) {
        char buf[160];
        snprintf(buf, sizeof buf, "wide kernel: placement census found %d XCDs over %d CUs with the block -> XCD mapping %s (needs 8 XCDs, b %% 8)",
                 st->n_xcd, st->ncu, st->map_ok ? "as assumed" : "NOT as assumed");
        err = buf;
        return WNV_ERR_UNSUPPORTED;
    }
    const int cus_per_xcd = st->ncu / 8;
    const int NG = L + 1;                                            // groups of 8 workgroups: one per layer + the tail (last skip conv)
    const int nL = (NG + 7) / 8;                                     // groups per XCD
    // the head goes to the XCD of the tail group when that XCD has a free slot, else to the first XCD that has one
    int head_x = -1, head_li = -1;
    const int n_head = st->NB > 1 ? (st->cin1 > 1 ? 6 : 4) : st->cin1 > 1 ? 2 : 1;      // one-hot models: two workgroups; more than 256 skip channels: four parts (+ two for a one-hot output layer)
    auto groups_on = [&](int x) { return std::max(0, std::min(nL, NG - x * nL)); };
    const int last_x = (NG - 1) / nL;
    for (int k = 0; k < 8 && head_x < 0; ++k) {
        const int x = (last_x + k) % 8;
        if (groups_on(x) * PG + n_head <= cus_per_xcd) { head_x = x; head_li = groups_on(x) * PG; }
    }
    if (head_x < 0 || nL * PG > cus_per_xcd) { err = "wide kernel: not enough CUs per XCD for the layer groups + the head"; return WNV_ERR_UNSUPPORTED; }
    WideParams p{};
    p.L = L; p.nL = nL; p.B = B; p.T = (int)ga.T; p.Tt = (int)ga.Tt; p.O = st->O; p.cin = st->cin; p.cinp = st->cinp; p.kw = st->kw; p.nz = ga.nz;
    p.dist = c.output_distribution; p.kpre = st->kpre; p.nkb = st->nkb;
    p.b0 = ga.b0; p.noise_B = ga.noise_B > 0 ? ga.noise_B : B;
    p.head_x = head_x; p.head_li = head_li;
    p.NB = st->NB; p.KW = KWD * st->NB;
    p.cin1 = st->cin1; p.softmax = ga.softmax; p.quantize = ga.quantize; p.index_out = ga.index_out;
    { const char* e = wnv_knob("WNV_RING_FAST"); p.fast = !(e && e[0] == '0'); }
    p.skip_scale = (float)std::sqrt(1.0 / L);
    const float* w = st->d_w;
    p.wn = w + st->o_wn; p.wm = w + st->o_wm; p.wo = w + st->o_wo; p.ws = w + st->o_ws; p.wsl = w + st->o_wsl; p.bo = w + st->o_bo; p.bs = w + st->o_bs;
    p.cvec = w + st->o_cvec; p.wpre = w + st->o_wpre;
    p.wh1 = w + st->o_wh1; p.bh1 = w + st->o_bh1; p.wh2 = w + st->o_wh2; p.bh2 = w + st->o_bh2; p.wfirst = w + st->o_wf; p.bfirst = w + st->o_bf;
    p.zbias = ga.zbias; p.zbias_bstride = ga.zbias_bstride; p.zb_ld = (c.gate_channels + 3) & ~3; p.gh_model = c.gate_channels / 2;
    p.lay_dil = st->d_dil; p.lay_histoff = st->d_histoff;
    p.hist_b_floats = (long long)PG * st->hist_layer_floats;
    // state: [status 64 B][X B (L+1) 768 u64][SK B (L+2) 256 u64][hidden B 256 u64][history B x 8 copies x layers]
    const size_t head_bytes = 64;
    const size_t n_x = (size_t)B * (L + 1) * XW, n_s = (size_t)B * (L + 2) * KWD * st->NB;
    const size_t n_hid = (size_t)B * KWD * st->NB;
    const size_t n_om = (size_t)B * 4 * 64;                         // partial head outputs of head parts 1 .. 3 (more than 256 skip channels)
    const size_t mail_bytes = (n_x + n_s + n_hid + n_om) * sizeof(u64);
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
    p.omail = p.hidmail + n_hid;
    p.hist = (float*)(p.omail + n_om);
    p.c_up = ga.c_up; p.initial = ga.initial; p.teacher = ga.teacher; p.noise = ga.noise; p.seed = ga.seed;
    p.out = ga.out; p.params_out = ga.params_out;
    const size_t lds = std::max(std::max(std::max(stage_lds_floats(p.kpre, B), HEAD_LDS_FLOATS), CAT_LDS_FLOATS), HEAD5_LDS_FLOATS) * sizeof(float);
    if (lds > 160 * 1024) { err = "wide kernel needs too much LDS"; return WNV_ERR_UNSUPPORTED; }
    WIDE_HIP(hipFuncSetAttribute((const void*)wnv_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        int per_cu = 0;
        WIDE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)wnv_wide_kernel, WT, lds));
        const int live = NG * PG + n_head;
        if (per_cu < 1 || live > st->ncu * per_cu) {
            char buf[160];
            snprintf(buf, sizeof buf, "wide kernel: %d workgroups must be co-resident but the device holds %d", live, st->ncu * std::max(per_cu, 0));
            err = buf;
            return WNV_ERR_UNSUPPORTED;
        }
    }
    // optional timeline (WNV_WIDE_TRACE=<file>): wall-clock stamps of utterance 0 at slice 0 of every group, 8 steps in mid-run
    const char* trace_path = wnv_knob("WNV_WIDE_TRACE");
    unsigned long long* d_trace = nullptr;
    const int trace_n = 8;
    size_t trace_words = 0;
    if (trace_path && *trace_path && p.T > 64) {
        trace_words = (size_t)trace_n * (L + 1) * WTW;
        WIDE_HIP(hipMalloc((void**)&d_trace, trace_words * sizeof(unsigned long long)));
        WIDE_HIP(hipMemsetAsync(d_trace, 0, trace_words * sizeof(unsigned long long), stream));
        p.trace = d_trace; p.trace_t0 = std::min(p.T / 2, 1000); p.trace_n = trace_n;
        { const char* e = wnv_knob("WNV_WIDE_TRACE_B"); p.trace_b = e ? std::min(std::max(atoi(e), 0), B - 1) : 0; }
    }
    const int max_li = std::max(nL * PG - 1, head_li + n_head - 1);
    const int grid = 8 * (max_li + 1);
    hipLaunchKernelGGL(wnv_wide_kernel, dim3(grid), dim3(WT), lds, stream, p);
    WIDE_HIP(hipGetLastError());
    WIDE_HIP(hipMemcpyAsync(st->h_status, p.status, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    WIDE_HIP(hipStreamSynchronize(stream));
    if (d_trace) {
        std::vector<unsigned long long> tr(trace_words);
        WIDE_HIP(hipMemcpy(tr.data(), d_trace, trace_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        if (FILE* f = fopen(trace_path, "w")) {
            fprintf(f, "# step group(L = head) stamps in ns (100 MHz wall clock) relative to the head's send before the first traced step: group: inputs gathered | "
                       "partials in LDS | u published | own h gathered | next pre ready;  head: skip gathered | next input sent\n");
            const unsigned long long t00 = tr[((size_t)0 * (L + 1) + L) * WTW + 1];
            for (int tt = 0; tt < trace_n; ++tt)
                for (int pos = 0; pos <= L; ++pos) {
                    fprintf(f, "%d %d", p.trace_t0 + tt, pos);
                    for (int k = 0; k < WTW; ++k) {
                        const unsigned long long v = tr[((size_t)tt * (L + 1) + pos) * WTW + k];
                        fprintf(f, " %lld", v ? (long long)(v - t00) * 10 : -1LL);
                    }
                    fprintf(f, "\n");
                }
            fclose(f);
        }
    }
    if (*st->h_status != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "wide kernel gave up waiting (code 0x%x: 0x1ll / 0x2ll = h / u into group ll, 0x3ll = skip, 0x400 = head, 0x5.. / 0x6.. = own outputs)", *st->h_status);
        err = buf;
        return WNV_ERR_TIMEOUT;
    }
    return WNV_OK;
}
