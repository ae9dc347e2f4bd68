// wnv_forward.hip -- teacher-forced batch evaluation, WaveNet.forward (wavenet.py:164-213), SURVEY.md 8f row f3.
//
// Unlike the sample loop this IS a GEMM workload: per layer  Z[256 x T] = W_in^T [taps | c]  with the kw dilated taps of the
// layer input and the conditioning row (conv.py / modules.py:127-150 evaluated for all t at once), the gate, then
// [(128 + K) x 128] x [128 x T] for conv1x1_out | conv1x1_skip (modules.py:157-162).  f32 in, f32 out (the parity bar is 1e-4
// against an f32 reference), so the matrix cores run v_mfma_f32_32x32x2_f32: exact f32 FMAs at the f32 vector rate (157 TFLOP/s
// peak at 2.4 GHz) -- the bound of this kernel is that MFMA issue rate, not HBM (58 GFLOP against ~0.4 GB per layer at the bench
// size).
//
// v11 (round 2): CHANNEL-MAJOR activations and TRANSPOSED GEMMs.  Activations live as (B, 128, T) -- the layout the reference's
// conv1d tensors have -- so every global access runs along time, and the matrices are multiplied the other way round:
//     D[channel][time] = sum_k  A = W[k][channel]  x  B = X[k][time]
//   * a wave owns 32 time steps and ALL output channels: GEMM1 is 8 accumulator tiles (256 gate rows), 64 MFMAs per K chunk of 16;
//   * the accumulator layout D[8 (v / 4) + 4 (lane / 32) + v % 4][lane % 32] has the wave's time step in the lane, i.e. a gated
//     accumulator register IS the B operand (U[k][time], two k per MFMA) of the second GEMM: no LDS round trip, no transpose, no
//     barrier between the GEMMs -- the K order of GEMM2 is simply the order in which the accumulator rows come (the weight rows are
//     read from LDS in that order);
//   * one workgroup = 4 waves = 128 time steps of one utterance for one layer (half the weight bytes per time step of the 64-row
//     tile of v1-v10), two workgroups per CU;
//   * staging is software-pipelined under the MFMAs (as v10): double LDS chunk buffers, the commit of chunk g + 1 and the global
//     loads of chunk g + 2 are issued between the MFMAs of chunk g, ONE barrier per 64 MFMAs.
// History, each step measured (profiles/r01_forward_*, r02_forward_*): v1 59.9 TFLOP/s -> v8 93.8 (prefetch in registers, LDS bank
// conflicts, scheduling barriers, straight-line epilogue) -> v9 104.1 (epilogue addressing, no spills) -> v10 109.2 (software
// pipelining, time-major 64-row tile) -> v11.
// Lane layout of the 32x32x2 MFMA (checked on the device by scripts/ubench_mfma.hip): A[i = lane % 32][k = lane / 32],
// B[k = lane / 32][j = lane % 32], D[8 (v / 4) + 4 (lane / 32) + v % 4][lane % 32] for accumulator register v.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "wnv_dev.h"
#include "wnv_forward.h"

namespace {

constexpr int FT = 256;            // threads per workgroup (4 waves)
constexpr int TN = 128;            // time steps per tile (32 per wave)
constexpr int KT = 16;             // K rows per activation chunk
constexpr int XT = TN + 4;         // row stride of the K-major activation chunk (16-byte aligned rows)
constexpr int HC = 128;            // residual channels = gate half width this kernel is specialised for
constexpr int WCH = 4096;          // floats per weight chunk buffer: [16 k][256 channels] (GEMM1) or [32 k][128 channels] (GEMM2)

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int v, int lane) { return 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3); }

// tanh(a) * sigmoid(g) with the hardware exp2 / rcp (absolute error ~1e-7, as in the sample-loop kernel)
__device__ __forceinline__ float fwd_gate(float a, float g) {
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -2.8853900817779268f);
    const float f = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}

#ifdef WNV_FWD_TRACE
// debug build: per-phase cycles of wave 0 of every workgroup, summed (s_memtime), read back by the host after the last layer
__device__ unsigned long long g_fwd_phase[16];
#define FWD_STAMP(k) do { const unsigned long long now__ = __builtin_readcyclecounter(); ph__[k] += now__ - t_prev__; t_prev__ = now__; } while (0)
#else
#define FWD_STAMP(k) do { } while (0)
#endif

struct LayerArgs {
    const float* Hin; float* Hout; float* Skip;        // (B, 128, T), (B, 128, T), (B, K, T): channel-major
    const float* c_up;                                  // (B, T, cin) time-major (what the engine's upsampler writes), or null
    const float* zbias; long long zb_bstride;           // per utterance [256]: conv bias (+ Wg g), this layer
    const float *w_in, *w_os, *b_os;                    // K-major [kw*128 + cin][256], [128][nosp], [nosp]
    long long T; int tiles_per_utt, d, kw, cin, K, nosp;
};

// ---- the chunk sequence of one tile -------------------------------------------------------------------------------------------
// g < n1p: GEMM1 chunk g = rows [16 g, 16 g + 16) of W_in (all 256 columns) + the matching activation rows; n1p = the chunk count
// padded to even (a padding step runs no MFMAs).  Then GEMM2: for each block of 128 output channels, 4 chunks = rows [32 c, 32 c + 32)
// of [W_out | W_skip] x the block's 128 columns.  A chunk is 1024 float4 in both cases, row-major: thread tid stages the float4s
// q * 256 + tid (q = 0 .. 3), a wave reads a contiguous KiB per load and writes 64 consecutive 16-byte LDS slots per store.
//
// NO VECTOR ALU WORK IN THE STEPS.  On this chip an ordinary VALU instruction takes matrix-pipe time (scripts/ubench_mfma_peak.hip:
// every VALU op issued between f32 MFMAs costs 3 - 5 cycles of the 64 an MFMA takes; f32 MFMA and packed f32 FMA share their peak rate),
// and v11's steps carried ~200 of them per 64 MFMAs (per-element addresses, clamps, masks): 22 % of the pipe.  Every global address
// of a step is therefore  WAVE-UNIFORM base (scalar ALU) + a per-thread byte offset computed ONCE per tile;  masks, clamps and
// unaligned windows are resolved by uniform branches into a fast kind (nothing but loads and LDS stores) and a generic slow kind
// (edge tiles, partial chunks, time steps before the utterance starts).
struct Thr {                                             // per-thread constants of the staging maps
    unsigned w1;                                         // GEMM1 weight chunk: float4 tid of a contiguous 16 KB chunk
    unsigned w2;                                         // GEMM2 weight chunk: row tid / 32 (+ 8 q), float4 tid % 32 of [32][128] in a matrix of row stride nosp
    unsigned xtap;                                       // tap rows: row tid / 16, time 8 (tid % 16) of channel-major (., T)
    unsigned xcond;                                      // conditioning: time tid / 2, channels 8 (tid % 2) of time-major (., cin)
    int l_tap, l_cond;                                   // where those land in the K-major LDS chunk (floats)
};

struct WSel { const float* base; unsigned qstride; bool fast, one; int k0, c0; };        // fast: base + Thr offset + q * qstride (bytes); else generic
__device__ __forceinline__ WSel w_sel(const LayerArgs& a, int g, int n1p, int gtot, int Kin) {
    g = min(g, gtot - 1);                                          // past the end: a harmless re-fetch
    WSel w;
    w.one = g < n1p;
    const int r = g - n1p;
    w.k0 = w.one ? g * KT : (r & 3) * 32; w.c0 = w.one ? 0 : 128 * (r >> 2);
    w.fast = !w.one || w.k0 + KT <= Kin;
    w.base = w.one ? a.w_in + (size_t)w.k0 * 256 : a.w_os + (size_t)w.k0 * a.nosp + w.c0;
    w.qstride = w.one ? 4096u : 32u * (unsigned)a.nosp;            // 4 rows of 256 floats | 8 rows of the [W_out | W_skip] matrix
    return w;
}
// No masks on the weights: a row past the matrix (the partial last GEMM1 chunk, the idle padding step) meets activations that are
// zeroed, a column past it feeds accumulator rows that are never stored; the clamps of the generic kind keep every address inside.
__device__ __forceinline__ float4 fetch_w_piece(const WSel& w, const LayerArgs& a, const Thr& th, int q, int tid, int Kin) {
    if (w.fast) return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w.base) + (size_t)(w.one ? th.w1 : th.w2) + (size_t)q * w.qstride);
    const int f = q * FT + tid, k = w.k0 + (f >> 6), col = 4 * (f & 63);          // (only GEMM1 chunks can be partial)
    return *reinterpret_cast<const float4*>(a.w_in + (size_t)min(k, Kin - 1) * 256 + col);
}

// One chunk in registers: the weights and the activations -- 8 floats per thread; an unaligned tap window is loaded as three aligned
// float4 and cut at commit time; the generic kind carries a validity mask (bit e of xmask: rows before t = 0, rows past T, channels past
// cin read as zeros), applied at commit: nothing in the step that issues a load waits for it.
struct Stage { float4 w[4]; float4 x[3]; unsigned xmask; };

// activation chunk g of GEMM1: tap chunks (g < 8 kw: channels 16 (g % 8) .. + 16 of tap g / 8, oldest first, conv.py:55-61) come
// from the channel-major layer input -- thread (row tid / 16, 8 consecutive time steps 8 (tid % 16)); conditioning chunks
// (modules.py:141-144) from the time-major c -- thread (time tid / 2, 8 channels).
// kind 0: tap, 16-byte aligned window; 1 .. 3: tap, the window starts r floats into the first of three aligned float4; 4: full
// conditioning chunk; 5: generic (element loads, clamped addresses, mask).
struct XSel { int kind; const float* base; int g; };
__device__ __forceinline__ XSel x_sel(const LayerArgs& a, int b, long long t0, int g, int n1, bool aligned_T) {
    XSel x;
    x.g = g = min(g, n1 - 1);
    if (g < 8 * a.kw) {
        const long long shift = (long long)(a.kw - 1 - (g >> 3)) * a.d;
        const int r = (int)((4 - (shift & 3)) & 3);
        const long long start = t0 - shift - r;                    // (t0 + 8 (tid % 16) - shift) rounded down to a multiple of 4
        x.base = a.Hin + ((size_t)b * HC + KT * (g & 7)) * a.T + start;
        x.kind = (aligned_T && start >= 0 && t0 + TN <= a.T) ? r : 5;
    } else {
        const int c0 = KT * (g - 8 * a.kw);
        x.base = a.c_up + ((size_t)b * a.T + t0) * a.cin + c0;
        x.kind = (t0 + TN <= a.T && c0 + KT <= a.cin) ? 4 : 5;
    }
    return x;
}
__device__ __forceinline__ void fetch_x(Stage& R, const XSel& x, const LayerArgs& a, const Thr& th, int b, long long t0, int tid) {
    if (x.kind < 4) {
        const char* p = reinterpret_cast<const char*>(x.base) + (size_t)th.xtap;
        R.x[0] = *reinterpret_cast<const float4*>(p);
        R.x[1] = *reinterpret_cast<const float4*>(p + 16);
        if (x.kind > 0) R.x[2] = *reinterpret_cast<const float4*>(p + 32);
    } else if (x.kind == 4) {
        const char* p = reinterpret_cast<const char*>(x.base) + (size_t)th.xcond;
        R.x[0] = *reinterpret_cast<const float4*>(p);
        R.x[1] = *reinterpret_cast<const float4*>(p + 16);
    } else if (x.g < 8 * a.kw) {
        const int ch = KT * (x.g & 7) + (tid >> 4);
        const long long ts = t0 + 8 * (tid & 15) - (long long)(a.kw - 1 - (x.g >> 3)) * a.d;
        const float* row = a.Hin + ((size_t)b * HC + ch) * a.T;
        float v[8];
        unsigned mk = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long te = ts + e;
            v[e] = row[min(max(te, 0ll), a.T - 1)];
            mk |= (te >= 0 && te < a.T) ? (1u << e) : 0u;
        }
        R.x[0] = make_float4(v[0], v[1], v[2], v[3]);
        R.x[1] = make_float4(v[4], v[5], v[6], v[7]);
        R.xmask = mk;
    } else {
        const int c = KT * (x.g - 8 * a.kw) + 8 * (tid & 1);                          // cin % 4 == 0 on this path (checked by the host)
        const long long t = t0 + (tid >> 1);
        const float* p = a.c_up + ((size_t)b * a.T + min(t, a.T - 1)) * a.cin;
        R.x[0] = *reinterpret_cast<const float4*>(p + min(c, a.cin - 4));
        R.x[1] = *reinterpret_cast<const float4*>(p + min(c + 4, a.cin - 4));
        R.xmask = (t < a.T && c < a.cin ? 0x0Fu : 0u) | (t < a.T && c + 4 < a.cin ? 0xF0u : 0u);
    }
}
__device__ __forceinline__ float4 masked(const float4& v, unsigned m4) {
    return make_float4((m4 & 1u) ? v.x : 0.f, (m4 & 2u) ? v.y : 0.f, (m4 & 4u) ? v.z : 0.f, (m4 & 8u) ? v.w : 0.f);
}
// commit the activations of a chunk into the K-major LDS chunk xt [16][XT]
__device__ __forceinline__ void commit_x(float* xt, const Stage& R, const XSel& x, const LayerArgs& a, const Thr& th) {
    float4 lo = R.x[0], hi = R.x[1];
    const bool tap = x.g < 8 * a.kw;
    if (x.kind == 1) { lo = make_float4(R.x[0].y, R.x[0].z, R.x[0].w, R.x[1].x); hi = make_float4(R.x[1].y, R.x[1].z, R.x[1].w, R.x[2].x); }
    else if (x.kind == 2) { lo = make_float4(R.x[0].z, R.x[0].w, R.x[1].x, R.x[1].y); hi = make_float4(R.x[1].z, R.x[1].w, R.x[2].x, R.x[2].y); }
    else if (x.kind == 3) { lo = make_float4(R.x[0].w, R.x[1].x, R.x[1].y, R.x[1].z); hi = make_float4(R.x[1].w, R.x[2].x, R.x[2].y, R.x[2].z); }
    else if (x.kind == 5) { lo = masked(lo, R.xmask); hi = masked(hi, R.xmask >> 4); }
    if (tap) {
        *reinterpret_cast<float4*>(xt + th.l_tap) = lo;
        *reinterpret_cast<float4*>(xt + th.l_tap + 4) = hi;
    } else {
        float* dst = xt + th.l_cond;
        dst[0] = lo.x; dst[XT] = lo.y; dst[2 * XT] = lo.z; dst[3 * XT] = lo.w;
        dst[4 * XT] = hi.x; dst[5 * XT] = hi.y; dst[6 * XT] = hi.z; dst[7 * XT] = hi.w;
    }
}

// ---- one GEMM1 step: 64 MFMAs (8 k-pairs x 8 channel tiles) out of chunk buffers (xt, wc); in their shadow the commit of chunk
// g + 1 (registers Rc, loaded a step ago; selector xc) into the other buffers and the loads of chunk g + 2 (selectors wf, xf) into Rf.
// Ends in the one barrier.
template <bool MFMA>
__device__ __forceinline__ void gemm1_step(f16v (&acc)[8], const float* xt, const float* wc, const Stage& Rc, const XSel& xc, float* xt_n, float* wc_n, Stage& Rf,
                                           const WSel& wf, const XSel& xf, const LayerArgs& a, const Thr& th, int b, long long t0, int Kin, int tid, int lane, int wave) {
    if constexpr (MFMA) {
        const float* Xb = xt + (lane >> 5) * XT + 32 * wave + (lane & 31);
        const float* Wa = wc + (lane >> 5) * 256 + (lane & 31);
        float bv = Xb[0];
        float av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = Wa[32 * i];
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            float bn = 0.f, an[8];
            if (ks + 1 < KT / 2) {
                bn = Xb[(2 * ks + 2) * XT];
#pragma unroll
                for (int i = 0; i < 8; ++i) an[i] = Wa[(2 * ks + 2) * 256 + 32 * i];
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (ks < 4) {
                Rf.w[ks] = fetch_w_piece(wf, a, th, ks, tid, Kin);
                reinterpret_cast<float4*>(wc_n)[ks * FT + tid] = Rc.w[ks];
            } else if (ks == 4) {
                fetch_x(Rf, xf, a, th, b, t0, tid);
            } else if (ks == 5) {
                commit_x(xt_n, Rc, xc, a, th);
            }
            if (ks + 1 < KT / 2) {
                bv = bn;
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = an[i];
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Rf.w[q] = fetch_w_piece(wf, a, th, q, tid, Kin);
        fetch_x(Rf, xf, a, th, b, t0, tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(wc_n)[q * FT + tid] = Rc.w[q];
        commit_x(xt_n, Rc, xc, a, th);
    }
    __syncthreads();
}

// ---- one GEMM2 step: 64 MFMAs (16 k-pairs x 4 channel tiles).  B operand = the gated accumulator registers u[0 .. 15] of one
// 32-channel tile (register s holds channels 8 (s / 4) + s % 4 and + 4 of the tile: the k pair of MFMA s); A = the matching rows
// of the weight chunk [32 k][128 channels].  Weight staging as in gemm1_step.
__device__ __forceinline__ void gemm2_step(f16v (&acc)[4], const f16v& u, const float* wc, const Stage& Rc, float* wc_n, Stage& Rf, const WSel& wf,
                                           const LayerArgs& a, const Thr& th, int Kin, int tid, int lane) {
    const float* Wa = wc + (4 * (lane >> 5)) * 128 + (lane & 31);
    float av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = Wa[32 * i];
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float an[4];
        if (s + 1 < 16) {
            const int rown = 8 * ((s + 1) >> 2) + ((s + 1) & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = Wa[rown * 128 + 32 * i];
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], u[s], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (s < 4) {
            Rf.w[s] = fetch_w_piece(wf, a, th, s, tid, Kin);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        } else if (s >= 8 && s < 12) {
            reinterpret_cast<float4*>(wc_n)[(s - 8) * FT + tid] = Rc.w[s - 8];
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (s + 1 < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = an[i];
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(FT, 2) wnv_fwd_layer_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xt0 = smem;                                       // [2][KT][XT] activation chunks, K-major
    float* const wc0 = smem + 2 * KT * XT;                         // [2][WCH]    weight chunks
    float* const bz = wc0 + 2 * WCH;                               // [256] gate bias (+ Wg g), then [128 + K] b_out | b_skip
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TN;
    const int Kin = a.kw * HC + a.cin;
    const int n1 = (Kin + KT - 1) / KT, n1p = (n1 + 1) & ~1;       // GEMM1 steps, padded to an even count (the extra step runs no MFMAs)
    const int ntot = HC + a.K, nblk = ntot / HC, gtot = n1p + 4 * nblk;
    const bool aligned_T = (a.T & 3) == 0;
    const bool interior = t0 + TN <= a.T;
    Thr th;
    th.w1 = 16u * (unsigned)tid;
    th.w2 = 4u * ((unsigned)(tid >> 5) * (unsigned)a.nosp + 4u * (unsigned)(tid & 31));
    th.xtap = 4u * ((unsigned)(tid >> 4) * (unsigned)a.T + 8u * (unsigned)(tid & 15));          // (T <= 2^26: the host checks)
    th.xcond = 4u * ((unsigned)(tid >> 1) * (unsigned)a.cin + 8u * (unsigned)(tid & 1));
    th.l_tap = (tid >> 4) * XT + 8 * (tid & 15);
    th.l_cond = 8 * (tid & 1) * XT + (tid >> 1);

    Stage RA, RB;
#ifdef WNV_FWD_TRACE
    unsigned long long ph__[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev__ = __builtin_readcyclecounter();
    const unsigned long long wall0__ = wall_clock64();
#endif
    // ---- prologue: biases, chunk 0 into buffer 0, chunk 1 into RA -------------------------------------------------------------
    {
        bz[tid] = a.zbias[(size_t)b * a.zb_bstride + tid];
        for (int i = tid; i < ntot; i += FT) bz[256 + i] = a.b_os[i];
        const WSel w0 = w_sel(a, 0, n1p, gtot, Kin), w1 = w_sel(a, 1, n1p, gtot, Kin);
        const XSel x0 = x_sel(a, b, t0, 0, n1, aligned_T), x1 = x_sel(a, b, t0, 1, n1, aligned_T);
#pragma unroll
        for (int q = 0; q < 4; ++q) RB.w[q] = fetch_w_piece(w0, a, th, q, tid, Kin);
        fetch_x(RB, x0, a, th, b, t0, tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) RA.w[q] = fetch_w_piece(w1, a, th, q, tid, Kin);
        fetch_x(RA, x1, a, th, b, t0, tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(wc0)[q * FT + tid] = RB.w[q];
        commit_x(xt0, RB, x0, a, th);
        __syncthreads();
    }
    // the gate bias (+ global conditioning) is the accumulators' initial value (LDS reads, no vector ALU work)
    f16v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = bz[32 * i + acc_row(v, lane)];
    FWD_STAMP(0);                                                 // 0: prologue
    // ---- GEMM1: Z^T = W_in^T [taps | c]^T, two steps per iteration (the register sets swap roles) -------------------------------
    for (int g = 0; g < n1p; g += 2) {
        {
            const WSel wf = w_sel(a, g + 2, n1p, gtot, Kin);
            const XSel xc = x_sel(a, b, t0, g + 1, n1, aligned_T), xf = x_sel(a, b, t0, g + 2, n1, aligned_T);
            gemm1_step<true>(acc, xt0, wc0, RA, xc, xt0 + KT * XT, wc0 + WCH, RB, wf, xf, a, th, b, t0, Kin, tid, lane, wave);
        }
        {
            const WSel wf = w_sel(a, g + 3, n1p, gtot, Kin);
            const XSel xc = x_sel(a, b, t0, g + 2, n1, aligned_T), xf = x_sel(a, b, t0, g + 3, n1, aligned_T);
            if (g + 1 < n1) gemm1_step<true>(acc, xt0 + KT * XT, wc0 + WCH, RB, xc, xt0, wc0, RA, wf, xf, a, th, b, t0, Kin, tid, lane, wave);
            else gemm1_step<false>(acc, xt0 + KT * XT, wc0 + WCH, RB, xc, xt0, wc0, RA, wf, xf, a, th, b, t0, Kin, tid, lane, wave);
        }
    }
    FWD_STAMP(5);                                                 // 5: GEMM1 steps
    // ---- tanh . sigmoid, in registers: u[i] = gate channels 32 i .. 32 i + 31 of this wave's 32 time steps, already in the layout
    //      GEMM2 wants for its B operand (modules.py:152-154) ---------------------------------------------------------------------
    f16v u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) u[i][v] = fwd_gate(acc[i][v], acc[4 + i][v]);
    FWD_STAMP(6);                                                 // 6: gate
    // ---- GEMM2: [out | skip]^T = [W_out | W_skip]^T U^T in blocks of 128 output channels; epilogue per block --------------------------
    const long long tl = t0 + 32 * wave + (lane & 31);              // this lane's time step
    // addresses of the epilogue = WAVE-UNIFORM row pointer (scalar registers: channel 32 i + 8 (v / 4) + v % 4 of the block) + ONE
    // 32-bit per-lane offset (4 (lane / 32) rows + the lane's time step; the host checks T <= 2^26)
    const int loff = (int)(4 * (lane >> 5) * a.T + min(tl, a.T - 1));
    const bool live = interior || tl < a.T;
    for (int blk = 0; blk < nblk; ++blk) {
        // block 0 is the residual output (modules.py:157-162: (out + x) sqrt(.5)), the others accumulate into the skip sum
        // (wavenet.py:196-198).  Row = channel, lane = time: every access is a 128-byte run along time.
        const bool res = blk == 0;
        const float* src = res ? a.Hin + (size_t)b * HC * a.T : a.Skip + ((size_t)b * a.K + (size_t)(blk - 1) * HC) * a.T;
        float* dst = res ? a.Hout + (size_t)b * HC * a.T : a.Skip + ((size_t)b * a.K + (size_t)(blk - 1) * HC) * a.T;
        const float* bias = bz + 256 + HC * blk;
        // the accumulators start from  bias + (what the result is added to): requested here, consumed by the first MFMAs four steps
        // of staging later -- the read-modify-write of the epilogue costs no waiting and no vector ALU work
        f16v o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[i][v] = (src + (size_t)(32 * i + 8 * (v >> 2) + (v & 3)) * a.T)[loff];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[i][v] += bias[32 * i + acc_row(v, lane)];
        const int g = n1p + 4 * blk;
        {
            const WSel wf = w_sel(a, g + 2, n1p, gtot, Kin);
            gemm2_step(o, u[0], wc0, RA, wc0 + WCH, RB, wf, a, th, Kin, tid, lane);
        }
        {
            const WSel wf = w_sel(a, g + 3, n1p, gtot, Kin);
            gemm2_step(o, u[1], wc0 + WCH, RB, wc0, RA, wf, a, th, Kin, tid, lane);
        }
        {
            const WSel wf = w_sel(a, g + 4, n1p, gtot, Kin);
            gemm2_step(o, u[2], wc0, RA, wc0 + WCH, RB, wf, a, th, Kin, tid, lane);
        }
        {
            const WSel wf = w_sel(a, g + 5, n1p, gtot, Kin);
            gemm2_step(o, u[3], wc0 + WCH, RB, wc0, RA, wf, a, th, Kin, tid, lane);
        }
        FWD_STAMP(7);                                             // 7: GEMM2 steps
        const float scale = res ? 0.70710678118654752440f : 1.0f;
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) (dst + (size_t)(32 * i + 8 * (v >> 2) + (v & 3)) * a.T)[loff] = res ? o[i][v] * scale : o[i][v];
        }
        FWD_STAMP(8);                                             // 8: epilogue
    }
#ifdef WNV_FWD_TRACE
    if (tid == 0) {
        for (int q = 0; q < 9; ++q) atomicAdd(&g_fwd_phase[q], ph__[q]);
        atomicAdd(&g_fwd_phase[14], wall_clock64() - wall0__);                    // 100 MHz ticks of the same interval: the shader clock under load
        atomicAdd(&g_fwd_phase[15], 1ull);
    }
#endif
}

// ---- head: relu -> 1x1 -> relu -> 1x1 (wavenet.py:200-207) on the same tile shape, transposed the same way; 4 % of the work, plain
// load -> commit -> barrier -> MFMA steps.  NOT = 32-channel tiles of the output (out_channels <= 32 NOT).
struct HeadArgs {
    const float* Skip; float* out;                      // (B, K, T) -> (B, O, T)
    const float *w_h1, *b_h1, *w_h2, *b_h2;             // K-major [K][kp], [kp], [K][op], [op]
    long long T; int tiles_per_utt, K, kp, O, op;
    float scale;
};

template <int NOT>
__global__ void __launch_bounds__(FT, 2) wnv_fwd_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xt = smem;                                        // [KT][XT]
    float* const wc = smem + 2 * KT * XT;                          // up to [32][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TN;
    const int kl = lane >> 5, jl = lane & 31;
    f16v oacc[NOT];
#pragma unroll
    for (int i = 0; i < NOT; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) oacc[i][v] = 0.f;
    for (int hb = 0; hb < a.K / HC; ++hb) {                      // hidden channels [128 hb, 128 hb + 128)
        f16v hacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) hacc[i][v] = 0.f;
        for (int kc = 0; kc < a.K / KT; ++kc) {
            {   // relu(skips * sqrt(1/L)) chunk (wavenet.py:200-203): 16 skip channels x 128 time steps
                const int r = tid >> 4, m8 = 8 * (tid & 15);
                const float* row = a.Skip + ((size_t)b * a.K + kc * KT + r) * a.T;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const long long t = t0 + m8 + e;
                    v[e] = t < a.T ? fmaxf(row[t] * a.scale, 0.f) : 0.f;
                }
                *reinterpret_cast<float4*>(xt + r * XT + m8) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(xt + r * XT + m8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {                        // W_h1 rows [16 kc, +16) x hidden columns [128 hb, +128): 512 float4
                const int f = q * FT + tid, k = kc * KT + (f >> 5), col = HC * hb + 4 * (f & 31);
                reinterpret_cast<float4*>(wc)[f] = *reinterpret_cast<const float4*>(a.w_h1 + (size_t)k * a.kp + col);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < KT / 2; ++ks) {
                const float bv = xt[(2 * ks + kl) * XT + 32 * wave + jl];
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[(2 * ks + kl) * HC + 32 * i + jl], bv, hacc[i], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)                              // + bias, ReLU: the hidden block, in GEMM2's B layout
#pragma unroll
            for (int v = 0; v < 16; ++v) hacc[i][v] = fmaxf(hacc[i][v] + a.b_h1[HC * hb + 32 * i + acc_row(v, lane)], 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                            // out += W_h2[rows of hidden tile c]^T . hidden tile c
            for (int f = tid; f < 32 * 8 * NOT; f += FT) {       // [32 k][32 NOT columns] (columns past the padded width: clamped, never stored)
                const int k = HC * hb + 32 * c + f / (8 * NOT), col = 4 * (f % (8 * NOT));
                reinterpret_cast<float4*>(wc)[f] = *reinterpret_cast<const float4*>(a.w_h2 + (size_t)k * a.op + min(col, a.op - 4));
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int row = 8 * (s >> 2) + 4 * kl + (s & 3);
#pragma unroll
                for (int i = 0; i < NOT; ++i) oacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[row * (32 * NOT) + 32 * i + jl], hacc[c][s], oacc[i], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    const long long tl = t0 + 32 * wave + jl;
    if (tl < a.T) {
#pragma unroll
        for (int i = 0; i < NOT; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int o = 32 * i + acc_row(v, lane);
                if (o < a.O) a.out[((size_t)b * a.O + o) * a.T + tl] = oacc[i][v] + a.b_h2[o];
            }
    }
}

// first_conv (wavenet.py:192): x (B, cin1, T) -> H (B, 128, T)
__global__ void wnv_fwd_first_kernel(const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bf,
                                     float* __restrict__ H, int cin1, long long T, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long long t = idx % T, br = idx / T;
    const int r = (int)(br % HC);
    const long long b = br / HC;
    float acc = bf[r];
    const float* xb = x + (size_t)b * cin1 * T + t;
    for (int k = 0; k < cin1; ++k) acc = fmaf(wf[(size_t)k * HC + r], xb[(size_t)k * T], acc);
    H[idx] = acc;
}

// F.softmax(x, dim=1) in place on (B, O, T) (wavenet.py:211)
__global__ void wnv_fwd_softmax_kernel(float* __restrict__ out, int O, long long T, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long long b = idx / T, t = idx % T;
    float* p = out + (size_t)b * O * T + t;
    float mx = -INFINITY;
    for (int o = 0; o < O; ++o) mx = fmaxf(mx, p[(size_t)o * T]);
    float sum = 0.f;
    for (int o = 0; o < O; ++o) sum += expf(p[(size_t)o * T] - mx);
    for (int o = 0; o < O; ++o) p[(size_t)o * T] = expf(p[(size_t)o * T] - mx) / sum;
}

}  // namespace

const char* wnv_forward_why_not(const WnvModelDev& m) {
    if (m.R != HC || m.G != 2 * HC) return "needs residual_channels == 128 and gate_channels == 256";
    if (m.K % HC != 0) return "needs skip_out_channels to be a multiple of 128";
    if (m.O > 256) return "needs out_channels <= 256";
    if (m.Rp != m.R) return "padded residual width";
    if (m.cin > 0 && (m.cin & 3) != 0) return "needs cin_channels to be a multiple of 4";
    return nullptr;
}

size_t wnv_forward_scratch_floats(const WnvModelDev& m, int B, long long T) {
    return (size_t)B * T * (2 * HC + m.K);
}

hipError_t wnv_launch_forward(const WnvModelDev& m, const WnvLayerDev* layers_host, const float* d_W, const WnvForwardArgs& a,
                              hipStream_t s) {
    const long long T = a.T;
    const int B = a.B;
    float* H0 = a.scratch;
    float* H1 = H0 + (size_t)B * T * HC;
    float* Skip = H1 + (size_t)B * T * HC;
    hipError_t e = hipMemsetAsync(Skip, 0, (size_t)B * T * m.K * sizeof(float), s);
    if (e != hipSuccess) return e;
    {
        const long long n = (long long)B * T * HC;
        hipLaunchKernelGGL(wnv_fwd_first_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.x, d_W + m.w_first, d_W + m.b_first,
                           H0, m.cin1, T, n);
    }
    const int tiles = (int)((T + TN - 1) / TN);
    const size_t lds_l = ((size_t)2 * KT * XT + 2 * WCH + 256 + HC + m.K) * sizeof(float);
    const size_t lds_h = ((size_t)2 * KT * XT + 2 * WCH) * sizeof(float);
    e = hipFuncSetAttribute((const void*)wnv_fwd_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l);
    if (e != hipSuccess) return e;
    float *in = H0, *out = H1;
    for (int l = 0; l < m.L; ++l) {
        const WnvLayerDev& Ld = layers_host[l];
        LayerArgs la{};
        la.Hin = in; la.Hout = out; la.Skip = Skip; la.c_up = m.cin > 0 ? a.c_up : nullptr;
        la.zbias = a.zbias + (size_t)l * m.Gp; la.zb_bstride = a.zbias_bstride;
        la.w_in = d_W + Ld.w_in; la.w_os = d_W + Ld.w_os; la.b_os = d_W + Ld.b_os;
        la.T = T; la.tiles_per_utt = tiles; la.d = Ld.dilation; la.kw = m.kw; la.cin = m.cin; la.K = m.K; la.nosp = m.NOSp;
        hipLaunchKernelGGL(wnv_fwd_layer_kernel, dim3((unsigned)(B * tiles)), dim3(FT), lds_l, s, la);
        std::swap(in, out);
    }
#ifdef WNV_FWD_TRACE
    {
        unsigned long long ph[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_fwd_phase), sizeof ph);
        const double n = (double)ph[15];
        static const char* names[9] = {"prologue (biases, chunks 0 and 1)", "-", "-", "-", "-", "GEMM1 steps", "gate", "GEMM2 steps", "epilogue"};
        double tot = 0;
        for (int k = 0; k < 9; ++k) tot += (double)ph[k];
        fprintf(stderr, "[wnv_forward trace] %.0f workgroup-layers, cycles per workgroup (wave 0), total %.0f:\n", n, tot / n);
        for (int k = 0; k < 9; ++k)
            if (ph[k]) fprintf(stderr, "   %-34s %10.0f  %5.1f %%\n", names[k], (double)ph[k] / n, 100.0 * (double)ph[k] / tot);
        fprintf(stderr, "   shader clock while the workgroups ran: %.0f MHz\n", tot / ((double)ph[14] / 100.0));
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_phase), z, sizeof z);
    }
#endif
    HeadArgs ha{};
    ha.Skip = Skip; ha.out = a.out; ha.w_h1 = d_W + m.w_h1; ha.b_h1 = d_W + m.b_h1; ha.w_h2 = d_W + m.w_h2; ha.b_h2 = d_W + m.b_h2;
    ha.T = T; ha.tiles_per_utt = tiles; ha.K = m.K; ha.kp = m.Kp; ha.O = m.O; ha.op = m.Op; ha.scale = m.skip_scale;
    const dim3 grid((unsigned)(B * tiles));
#define WNV_HEAD_LAUNCH(N)                                                                                                             \
    do {                                                                                                                               \
        e = hipFuncSetAttribute((const void*)wnv_fwd_head_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);          \
        if (e != hipSuccess) return e;                                                                                                 \
        hipLaunchKernelGGL(wnv_fwd_head_kernel<N>, grid, dim3(FT), lds_h, s, ha);                                                      \
    } while (0)
    if (m.O <= 32) WNV_HEAD_LAUNCH(1);
    else if (m.O <= 64) WNV_HEAD_LAUNCH(2);
    else if (m.O <= 128) WNV_HEAD_LAUNCH(4);
    else WNV_HEAD_LAUNCH(8);
#undef WNV_HEAD_LAUNCH
    if (a.softmax) {
        const long long n = (long long)B * T;
        hipLaunchKernelGGL(wnv_fwd_softmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.out, m.O, T, n);
    }
    return hipGetLastError();
}
