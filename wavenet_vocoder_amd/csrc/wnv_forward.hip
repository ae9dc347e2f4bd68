// wnv_forward.hip -- teacher-forced batch evaluation, WaveNet.forward (wavenet.py:164-213), SURVEY.md 8f row f3.
//
// Unlike the sample loop this IS a GEMM workload: per layer  Z[T x 256] = X[T x (kw*128 + cin)] W  with X the kw dilated
// taps of the layer input and the conditioning row (conv.py / modules.py:127-150 evaluated for all t at once), the gate,
// then [T x 128] x [128 x (128 + K)] for conv1x1_out | conv1x1_skip (modules.py:157-162).  f32 in, f32 out (the parity bar is
// 1e-4 against an f32 reference), so the matrix cores run v_mfma_f32_32x32x2_f32: exact f32 FMAs at the f32 vector rate
// (157 TFLOP/s peak) -- the bound of this kernel is that MFMA issue rate, not HBM (58 GFLOP against ~0.4 GB per layer at
// the bench size).
//
// One workgroup (4 waves) owns a tile of 64 time steps of one utterance for one layer and runs the whole layer on it:
// GEMM1 (K chunks of 32 staged K-major through LDS, taps gathered straight from the time-major activations) -> bias
// (+ global conditioning, hoisted) -> tanh . sigmoid in registers -> U tile K-major in LDS -> GEMM2 in column blocks of
// 256 -> residual / skip epilogue.  Activations ping-pong between two (B, T, 128) buffers (a layer reads rows t - k d of its
// input), the skip sum accumulates in a (B, T, K) buffer, the head is a third kernel of the same shape.
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
constexpr int TM = 64;             // time steps per tile
constexpr int KC = 32;             // K rows per LDS chunk
constexpr int XS = TM + 1;         // padded row stride of the K-major activation tiles
constexpr int WS = 256;            // row stride of a weight chunk
constexpr int HC = 128;            // residual channels = gate half width this kernel is specialised for

typedef float f16v __attribute__((ext_vector_type(16)));

struct Lds {
    float* xt;                     // [KC][XS]   activation chunk, K-major
    float* wc;                     // [KC][WS]   weight chunk
    float* ut;                     // [128][XS]  gate output / hidden block, K-major
};
__device__ __forceinline__ Lds carve(float* smem) { return {smem, smem + KC * XS, smem + KC * XS + KC * WS}; }
constexpr size_t LDS_FLOATS = (size_t)KC * XS + (size_t)KC * WS + (size_t)HC * XS;

// acc[i] += X[m0 .. m0+32)[chunk] * W[chunk][ncol[i] .. ncol[i]+32)  for the whole KC chunk.  The LDS operands of step ks + 1 are
// requested before the MFMAs of step ks are issued; the sched_group_barrier sequence pins that order ([DS reads of the next
// step][MFMAs of this step] ...) against the machine scheduler, which otherwise sinks every read next to its use and lets the
// matrix pipe wait for an LDS round trip after every second MFMA.
template <int NT>
__device__ __forceinline__ void mfma_chunk(f16v (&acc)[NT], const float* xt, int m0, const float* wc, const int (&ncol)[NT], int lane) {
    const int kl = lane >> 5, jl = lane & 31;
    const float* xa = xt + kl * XS + m0 + jl;
    const float* wb = wc + kl * WS + jl;
    float a = xa[0];
    float b[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) b[i] = wb[ncol[i]];
    __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
#pragma unroll
    for (int ks = 0; ks < KC / 2; ++ks) {
        float an = 0.f, bn[NT];
        if (ks + 1 < KC / 2) {
            an = xa[(2 * ks + 2) * XS];
#pragma unroll
            for (int i = 0; i < NT; ++i) bn[i] = wb[(2 * ks + 2) * WS + ncol[i]];
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one global load of the caller's next-chunk fetch per step: their
                                                                // issue (it blocks while the L1 is busy) hides behind the MFMAs
        if (ks + 1 < KC / 2) {
            a = an;
#pragma unroll
            for (int i = 0; i < NT; ++i) b[i] = bn[i];
        }
    }
}

// A chunk travels global -> registers -> LDS in two steps so that the global loads of chunk k + 1 are in flight while the
// matrix cores work on chunk k (the LDS copy is single; the registers are the second buffer).
// weight chunk: rows [k0, k0 + KC) x columns [c0, c0 + 256) of a K-major matrix [nrows][ld] (zeros outside)
// thread tid owns the float4s f = q * 256 + tid (q = 0 .. 7) of the 32 x 64-float4 chunk: a wave reads one contiguous 1 KiB row
// segment per load and writes 64 consecutive 16-byte LDS slots per store (conflict-free; the earlier row-per-8-lanes mapping put
// all 64 lanes of a ds_write_b128 on the same four banks: 65 % of the LDS cycles were bank conflicts, profiles/r01_forward_v1_pmc.txt)
struct WChunk { float4 v[8]; };
__device__ __forceinline__ void fetch_w_chunk(WChunk& r, const float* __restrict__ W, int ld, int nrows, int ncols, int k0, int c0, int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int f = q * FT + tid;
        const int k = k0 + (f >> 6), c = c0 + 4 * (f & 63);
        // branch-free (a clamped, always valid address + a select): every load of the chunk sits in one basic block, so the
        // compiler issues them back to back and waits once; ncols % 4 == 0 (padded widths)
        // (the mask is a MULTIPLICATION: a select would let the compiler sink the load back under a branch; the data are finite)
        const float ok = k < nrows && c < ncols ? 1.0f : 0.0f;
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)min(k, nrows - 1) * ld + min(c, ncols - 4));
        r.v[q] = make_float4(v.x * ok, v.y * ok, v.z * ok, v.w * ok);
    }
}
__device__ __forceinline__ void commit_w_chunk(float* wc, const WChunk& r, int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) reinterpret_cast<float4*>(wc)[q * FT + tid] = r.v[q];
}
__device__ __forceinline__ void load_w_chunk(float* wc, const float* __restrict__ W, int ld, int nrows, int ncols, int k0, int c0, int tid) {
    WChunk r;
    fetch_w_chunk(r, W, ld, nrows, ncols, k0, c0, tid);
    commit_w_chunk(wc, r, tid);
}

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
    const float* Hin; float* Hout; float* Skip;        // (B, T, 128), (B, T, 128), (B, T, K)
    const float* c_up;                                  // (B, T, cin) or null
    const float* zbias; long long zb_bstride;           // per utterance [256]: conv bias (+ Wg g), this layer
    const float *w_in, *w_os, *b_os;                    // K-major [kw*128 + cin][256], [128][nosp], [nosp]
    long long T; int tiles_per_utt, d, kw, cin, K, nosp;
};

// activation chunk kc of GEMM1 for time row m (thread = (m, eight K values)): tap j of the dilated conv (oldest first,
// conv.py:55-61) or the local conditioning row c[t] (modules.py:141-144)
__device__ __forceinline__ void fetch_x_chunk(float (&v)[8], const LayerArgs& a, const float* Hin, int b, long long t, int kc, int sub) {
    // branch-free (selected pointers, clamped addresses, zeros selected afterwards: rows before t = 0, rows past T, channels past
    // cin), so that the two loads share a basic block with the MFMAs they are scheduled between
    const long long tc = min(t, a.T - 1);
    const bool tap = kc < 4 * a.kw;                               // uniform across the workgroup
    const int j = kc >> 2, ch0 = 32 * (kc & 3) + 8 * sub;
    const long long tt = tc - (long long)(a.kw - 1 - j) * a.d;
    const int c0 = 32 * (kc - 4 * a.kw) + 8 * sub;                // cin % 4 == 0 on this path (checked by the host)
    const float* cr = a.c_up ? a.c_up + ((size_t)b * a.T + tc) * a.cin : Hin;
    const int cmax = a.cin > 4 ? a.cin - 4 : 0;
    const float* tsrc = Hin + (size_t)max(tt, 0ll) * HC + ch0;
    const float* src_p = tap ? tsrc : cr + min(max(c0, 0), cmax);
    const float* src_q = tap ? tsrc + 4 : cr + min(max(c0 + 4, 0), cmax);
    const float4 p = *reinterpret_cast<const float4*>(src_p);
    const float4 q = *reinterpret_cast<const float4*>(src_q);
    const float ok_p = t < a.T && (tap ? tt >= 0 : c0 < a.cin) ? 1.0f : 0.0f;          // masks by multiplication (see fetch_w_chunk)
    const float ok_q = t < a.T && (tap ? tt >= 0 : c0 + 4 < a.cin) ? 1.0f : 0.0f;
    v[0] = p.x * ok_p; v[1] = p.y * ok_p; v[2] = p.z * ok_p; v[3] = p.w * ok_p;
    v[4] = q.x * ok_q; v[5] = q.y * ok_q; v[6] = q.z * ok_q; v[7] = q.w * ok_q;
}

__global__ void __launch_bounds__(FT, 2) wnv_fwd_layer_kernel_v9(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds s = carve(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TM;
    const int rb = wave & 1, cb = wave >> 1;
    const int m0 = 32 * rb;
    const float* Hin = a.Hin + (size_t)b * a.T * HC;
    const int Kin = a.kw * HC + a.cin;
    const int nchunk = (Kin + KC - 1) / KC;
    const int xm = tid >> 2, xsub = tid & 3;                    // this thread's slice of an activation chunk
    const int ntot = HC + a.K;

    // ---- GEMM1: Z = [taps | c] W_in -----------------------------------------------------------------------------------
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const int ncol1[4] = {64 * cb, 64 * cb + 32, HC + 64 * cb, HC + 64 * cb + 32};      // tanh tiles, sigmoid tiles of the same channels
    float xr[8];
    WChunk wr;
#ifdef WNV_FWD_TRACE
    unsigned long long ph__[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // accumulated locally, published once at the end
    unsigned long long t_prev__ = __builtin_readcyclecounter();
#endif
    fetch_x_chunk(xr, a, Hin, b, t0 + xm, 0, xsub);
    fetch_w_chunk(wr, a.w_in, 256, Kin, 256, 0, 0, tid);
    for (int kc = 0; kc < nchunk; ++kc) {
        if (kc > 0) __syncthreads();                              // the matrix cores are done with the previous chunk
        FWD_STAMP(kc == 0 ? 0 : 1);                               // 0: prologue  1: wait for the slowest wave's MFMAs
#pragma unroll
        for (int e = 0; e < 8; ++e) s.xt[(8 * xsub + e) * XS + xm] = xr[e];
        commit_w_chunk(s.wc, wr, tid);
        FWD_STAMP(2);                                             // 2: vmcnt wait + LDS commit
        __syncthreads();
        FWD_STAMP(3);                                             // 3: commit barrier
        {   // next chunk (after the last one: the first chunk of GEMM2), selected without a branch so that its loads can be
            // scheduled between this chunk's MFMAs
            const bool more = kc + 1 < nchunk;
            fetch_x_chunk(xr, a, Hin, b, t0 + xm, more ? kc + 1 : kc, xsub);           // (after the last chunk: re-read, unused)
            fetch_w_chunk(wr, more ? a.w_in : a.w_os, more ? 256 : a.nosp, more ? Kin : HC, more ? 256 : ntot, more ? (kc + 1) * KC : 0, 0, tid);
        }
        FWD_STAMP(4);                                             // 4: issue of the next fetch
        mfma_chunk<4>(acc, s.xt, m0, s.wc, ncol1, lane);
        FWD_STAMP(5);                                             // 5: GEMM1 MFMA phase
    }
    // ---- bias (+ global conditioning), tanh . sigmoid -> U tile, K-major ---------------------------------------------------
    {
        const float* zb = a.zbias + (size_t)b * a.zb_bstride;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = 64 * cb + 32 * j + (lane & 31);
            const float za = zb[ch], zg = zb[HC + ch];
#pragma unroll
            for (int v = 0; v < 16; ++v)
                s.ut[ch * XS + m0 + acc_row(v, lane)] = fwd_gate(acc[j][v] + za, acc[2 + j][v] + zg);      // modules.py:152-154
        }
    }
    FWD_STAMP(6);                                                 // 6: gate
    // ---- GEMM2: [out | skip] = U [W_out | W_skip], in column blocks of 256 ---------------------------------------------------
    for (int c0 = 0; c0 < ntot; c0 += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        const int ncol2[4] = {128 * cb, 128 * cb + 32, 128 * cb + 64, 128 * cb + 96};
        for (int kc = 0; kc < HC / KC; ++kc) {
            __syncthreads();                                      // previous chunk consumed (first pass: the U tile is complete)
            commit_w_chunk(s.wc, wr, tid);
            __syncthreads();
            if (kc + 1 < HC / KC) fetch_w_chunk(wr, a.w_os, a.nosp, HC, ntot, (kc + 1) * KC, c0, tid);
            else if (c0 + 256 < ntot) fetch_w_chunk(wr, a.w_os, a.nosp, HC, ntot, 0, c0 + 256, tid);
            mfma_chunk<4>(acc, s.ut + kc * KC * XS, m0, s.wc, ncol2, lane);
        }
        FWD_STAMP(7);                                             // 7: GEMM2 (barriers, commits, MFMAs)
        // epilogue in two passes: every residual / skip value this thread needs is requested first (64 loads in flight), then the
        // results are combined and stored.  (Load -> add -> store per element serialises on the memory latency: the compiler
        // cannot prove that Hout / Skip do not alias Hin; the phase trace showed 43 % of a workgroup's time here.)
        const bool interior = t0 + TM <= a.T && c0 + 256 <= ntot;        // uniform: no per-element conditions, no branches
#pragma unroll
        for (int half = 0; half < 2; ++half) {                          // two tiles at a time: 32 loads in flight, 32 registers
            float prev[2][16];
            if (interior) {
                // Addresses as  WAVE-UNIFORM row pointer (scalar registers) + ONE 32-bit per-lane offset: row v of the accumulator is
                // time step t0 + m0 + 8 (v / 4) + v % 4 (+ 4 for the upper half-wave), column gc; a 32-column tile lies entirely in
                // the residual block or entirely in the skip block.  (Per-element 64-bit addresses cost 64 VGPRs here and pushed the
                // kernel into 87 spilled registers.)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * (2 * half + j));      // first column of the tile
                    const bool res = col0 < HC;
                    const int ld = res ? HC : a.K;
                    const float* base = (res ? Hin + col0 : a.Skip + (size_t)b * a.T * a.K + (col0 - HC)) + (size_t)(t0 + m0) * ld;
                    const int loff = 4 * (lane >> 5) * ld + (lane & 31);
#pragma unroll
                    for (int v = 0; v < 16; ++v) prev[j][v] = (base + (size_t)(8 * (v >> 2) + (v & 3)) * ld)[loff];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = 2 * half + j;
                    const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * i);
                    const bool res = col0 < HC;
                    const int ld = res ? HC : a.K;
                    const float bias = a.b_os[col0 + (lane & 31)];
                    float* base = (res ? a.Hout + (size_t)b * a.T * HC + col0 : a.Skip + (size_t)b * a.T * a.K + (col0 - HC)) + (size_t)(t0 + m0) * ld;
                    const int loff = 4 * (lane >> 5) * ld + (lane & 31);
                    const float scale = res ? 0.70710678118654752440f : 1.0f;        // (out + residual) * sqrt(0.5) | skips += s
#pragma unroll
                    for (int v = 0; v < 16; ++v) (base + (size_t)(8 * (v >> 2) + (v & 3)) * ld)[loff] = (prev[j][v] + (acc[i][v] + bias)) * scale;
                }
                continue;
            }
            // edge tiles (the last time tile of an utterance, a partial column block): the same addressing with per-element
            // predicates; rows are 32-bit offsets from wave-uniform bases, so nothing 64-bit is kept per element
            const int rows_left = (int)min((long long)TM, a.T - t0) - m0;         // valid rows of this wave's 32-row block (may be <= 0)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * (2 * half + j));
                const bool res = col0 < HC;
                const int ld = res ? HC : a.K;
                const float* base = (res ? Hin : a.Skip + (size_t)b * a.T * a.K - HC) + (size_t)(t0 + m0) * ld;
                const int gc = col0 + (lane & 31);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = acc_row(v, lane);
                    prev[j][v] = (gc < ntot && row < rows_left) ? base[row * ld + gc] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = 2 * half + j;
                const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * i);
                const bool res = col0 < HC;
                const int ld = res ? HC : a.K;
                const int gc = col0 + (lane & 31);
                if (gc >= ntot) continue;
                const float bias = a.b_os[gc];
                float* base = (res ? a.Hout + (size_t)b * a.T * HC : a.Skip + (size_t)b * a.T * a.K - HC) + (size_t)(t0 + m0) * ld;
                const float scale = res ? 0.70710678118654752440f : 1.0f;            // modules.py:157-162 | wavenet.py:196-198
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = acc_row(v, lane);
                    if (row < rows_left) base[row * ld + gc] = (prev[j][v] + (acc[i][v] + bias)) * scale;
                }
            }
        }
        FWD_STAMP(8);                                             // 8: epilogue (residual / skip read-modify-write)
    }
#ifdef WNV_FWD_TRACE
    if (tid == 0) {
        for (int k = 0; k < 9; ++k) atomicAdd(&g_fwd_phase[k], ph__[k]);
        atomicAdd(&g_fwd_phase[15], 1ull);
    }
#endif
}

// =================================================================================================================================
// v10 of the layer kernel: the same tile (64 time steps x one layer, 4 waves, two workgroups per CU) with the staging SOFTWARE-
// PIPELINED under the MFMAs.  v9 spent more cycles per K chunk outside the MFMA phase (wait for the prefetch, commit to LDS, two
// barriers, issue the next fetch: ~4.5 k cycles) than inside it (4.1 k), so even two workgroups per CU left the matrix pipe idle
// 30 % of the time.  Here the LDS chunk buffers are double (chunks of 16 K rows, 74 KB per workgroup), a step is
//     [MFMAs of chunk g out of buffer g % 2]  interleaved in program order with
//     [the LDS commit of chunk g + 1 (registers loaded during step g - 1) into buffer (g + 1) % 2]  and
//     [the global loads of chunk g + 2 into the other register set]
// and ends in ONE barrier.  The matrix pipe executes a 32x32x2 MFMA for 64 cycles during which the wave issues the next LDS reads,
// one ds_write and one global load: nothing but the barrier skew is left outside the MFMA shadow.  The chunk sequence runs
// through GEMM1 into GEMM2 (and across its column blocks) without draining.
constexpr int KL = 16;             // K rows per LDS chunk
constexpr size_t LDS_FLOATS_L = (size_t)2 * KL * XS + (size_t)2 * KL * WS + (size_t)HC * XS;
struct Stage { float4 w[4]; float4 x; float xok; };  // one chunk in registers: 16 x 256 weights (4 float4 per thread), 64 x 16 activations (one float4 + its zero mask)

struct ChunkSel { const float* W; int ld, nrows, ncols, k0, c0; };
// chunk g of the sequence: g < n1p GEMM1 (w_in rows 16 g ..; rows past Kin read as zeros), then 8 chunks per column block of [W_out | W_skip]
__device__ __forceinline__ ChunkSel chunk_sel(const LayerArgs& a, int g, int n1p, int gtot, int Kin, int ntot) {
    g = min(g, gtot - 1);                                          // past the end: a harmless re-fetch
    const bool one = g < n1p;
    const int r = g - n1p;
    ChunkSel c;
    c.W = one ? a.w_in : a.w_os; c.ld = one ? 256 : a.nosp; c.nrows = one ? Kin : HC; c.ncols = one ? 256 : ntot;
    c.k0 = one ? g * KL : (r & 7) * KL; c.c0 = one ? 0 : 256 * (r >> 3);
    return c;
}
// No masks on the weights: a row past the matrix (the partial last GEMM1 chunk, the idle padding step) meets activations that
// are zeroed, a column past it feeds accumulator columns that are never stored; the clamps keep every address inside the matrix.
__device__ __forceinline__ float4 fetch_w_piece(const ChunkSel& c, int q, int tid) {
    const int f = q * FT + tid;
    const int k = c.k0 + (f >> 6), col = c.c0 + 4 * (f & 63);
    return *reinterpret_cast<const float4*>(c.W + (size_t)min(k, c.nrows - 1) * c.ld + min(col, c.ncols - 4));
}
// activation chunk g of GEMM1 (16 K values: channels 16 (g % 8) .. of tap g / 8, or conditioning channels), time row m, K values 4 sub ..
// The zero mask (rows before t = 0, rows past T, channels past cin) is returned separately and applied when the chunk is committed,
// a step later: nothing in the step that issues a load waits for it.  Branch-free: selected pointers, clamped addresses.
__device__ __forceinline__ float4 fetch_x_piece(const LayerArgs& a, const float* Hin, const float* cr, long long t, int g, int n1, int sub, float& ok) {
    g = min(g, n1 - 1);
    const long long tc = min(t, a.T - 1);
    const int tap = g < 8 * a.kw;                                  // uniform across the workgroup
    const int j = g >> 3, ch0 = KL * (g & 7) + 4 * sub;
    const long long tt = tc - (long long)(a.kw - 1 - j) * a.d;
    const int c0 = KL * (g - 8 * a.kw) + 4 * sub;                  // cin % 4 == 0 on this path (checked by the host)
    const int cmax = a.cin > 4 ? a.cin - 4 : 0;
    const float* p_tap = Hin + (size_t)(tt > 0 ? tt : 0) * HC + ch0;
    const float* p_c = cr + (size_t)tc * a.cin + min(max(c0, 0), cmax);
    const float* src = tap ? p_tap : p_c;
    const int valid = (int)(t < a.T) & (tap ? (int)(tt >= 0) : (int)(c0 < a.cin));
    ok = valid ? 1.0f : 0.0f;
    return *reinterpret_cast<const float4*>(src);
}

// One step of the pipeline (see above).  A = K-major A tile of this chunk (+ kl * XS + m0 + jl applied by the caller), Bw = this
// chunk's weight buffer (+ kl * WS + jl), Rc = the registers of chunk g + 1 (committed to wc_n / xt_n), Rf = where chunk g + 2 lands.
template <bool XSTAGE, bool MFMA>
__device__ __forceinline__ void pipe_step(f16v (&acc)[4], const float* A, const float* Bw, const int (&ncol)[4], const Stage& Rc, float* wc_n, float* xt_n,
                                          Stage& Rf, const ChunkSel& cs, const LayerArgs& a, const float* Hin, const float* cr, long long t, int gx, int n1, int tid) {
    const int xm = tid >> 2, xsub = tid & 3;
    if constexpr (MFMA) {
        float av = A[0];
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = Bw[ncol[i]];
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
        for (int ks = 0; ks < KL / 2; ++ks) {
            float an = 0.f, bn[4];
            if (ks + 1 < KL / 2) {
                an = A[(2 * ks + 2) * XS];
#pragma unroll
                for (int i = 0; i < 4; ++i) bn[i] = Bw[(2 * ks + 2) * WS + ncol[i]];
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            // under these MFMAs: one piece of the fetch of chunk g + 2 and one piece of the commit of chunk g + 1
            if (ks < 4) {
                Rf.w[ks] = fetch_w_piece(cs, ks, tid);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                reinterpret_cast<float4*>(wc_n)[ks * FT + tid] = Rc.w[ks];
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            } else if (XSTAGE) {
                if (ks == 4) { Rf.x = fetch_x_piece(a, Hin, cr, t, gx, n1, xsub, Rf.xok); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                const float xv = ks == 4 ? Rc.x.x : ks == 5 ? Rc.x.y : ks == 6 ? Rc.x.z : Rc.x.w;
                xt_n[(4 * xsub + (ks - 4)) * XS + xm] = xv * Rc.xok;
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            if (ks + 1 < KL / 2) {
                av = an;
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = bn[i];
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Rf.w[q] = fetch_w_piece(cs, q, tid);
        if (XSTAGE) Rf.x = fetch_x_piece(a, Hin, cr, t, gx, n1, xsub, Rf.xok);
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(wc_n)[q * FT + tid] = Rc.w[q];
        if (XSTAGE) {
            xt_n[(4 * xsub + 0) * XS + xm] = Rc.x.x * Rc.xok; xt_n[(4 * xsub + 1) * XS + xm] = Rc.x.y * Rc.xok;
            xt_n[(4 * xsub + 2) * XS + xm] = Rc.x.z * Rc.xok; xt_n[(4 * xsub + 3) * XS + xm] = Rc.x.w * Rc.xok;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(FT, 2) wnv_fwd_layer_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xt0 = smem;                                       // [2][KL][XS] activation chunks, K-major
    float* const wc0 = smem + 2 * KL * XS;                         // [2][KL][WS] weight chunks
    float* const ut = wc0 + 2 * KL * WS;                           // [128][XS]   gate outputs, K-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TM;
    const int rb = wave & 1, cb = wave >> 1;
    const int m0 = 32 * rb;
    const int kl = lane >> 5, jl = lane & 31;
    const float* Hin = a.Hin + (size_t)b * a.T * HC;
    const int Kin = a.kw * HC + a.cin;
    const int n1 = (Kin + KL - 1) / KL, n1p = (n1 + 1) & ~1;       // GEMM1 steps, padded to an even count (the extra step runs no MFMAs)
    const int ntot = HC + a.K, nblk = (ntot + 255) / 256, gtot = n1p + 8 * nblk;
    const int xm = tid >> 2, xsub = tid & 3;
    const long long tx = t0 + xm;
    const float* cr = a.c_up ? a.c_up + (size_t)b * a.T * a.cin : Hin;      // conditioning rows of this utterance (no c: any valid address)

    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const int ncol1[4] = {64 * cb, 64 * cb + 32, HC + 64 * cb, HC + 64 * cb + 32};      // tanh tiles, sigmoid tiles of the same channels
    const int ncol2[4] = {128 * cb, 128 * cb + 32, 128 * cb + 64, 128 * cb + 96};
    Stage RA, RB;
#ifdef WNV_FWD_TRACE
    unsigned long long ph__[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev__ = __builtin_readcyclecounter();
#endif
    // ---- prologue: chunk 0 into buffer 0, chunk 1 into RA ---------------------------------------------------------------------
    {
        const ChunkSel c0s = chunk_sel(a, 0, n1p, gtot, Kin, ntot), c1s = chunk_sel(a, 1, n1p, gtot, Kin, ntot);
#pragma unroll
        for (int q = 0; q < 4; ++q) RB.w[q] = fetch_w_piece(c0s, q, tid);
        RB.x = fetch_x_piece(a, Hin, cr, tx, 0, n1, xsub, RB.xok);
#pragma unroll
        for (int q = 0; q < 4; ++q) RA.w[q] = fetch_w_piece(c1s, q, tid);
        RA.x = fetch_x_piece(a, Hin, cr, tx, 1, n1, xsub, RA.xok);
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(wc0)[q * FT + tid] = RB.w[q];
        xt0[(4 * xsub + 0) * XS + xm] = RB.x.x * RB.xok; xt0[(4 * xsub + 1) * XS + xm] = RB.x.y * RB.xok;
        xt0[(4 * xsub + 2) * XS + xm] = RB.x.z * RB.xok; xt0[(4 * xsub + 3) * XS + xm] = RB.x.w * RB.xok;
        __syncthreads();
    }
    FWD_STAMP(0);                                                 // 0: prologue
    // ---- GEMM1: Z = [taps | c] W_in, two steps per iteration (the register sets swap roles) -------------------------------------
    for (int g = 0; g < n1p; g += 2) {
        {
            const ChunkSel cs = chunk_sel(a, g + 2, n1p, gtot, Kin, ntot);
            pipe_step<true, true>(acc, xt0 + kl * XS + m0 + jl, wc0 + kl * WS + jl, ncol1, RA, wc0 + KL * WS, xt0 + KL * XS, RB, cs, a, Hin, cr, tx, g + 2, n1, tid);
        }
        {
            const ChunkSel cs = chunk_sel(a, g + 3, n1p, gtot, Kin, ntot);
            if (g + 1 < n1)
                pipe_step<true, true>(acc, xt0 + KL * XS + kl * XS + m0 + jl, wc0 + KL * WS + kl * WS + jl, ncol1, RB, wc0, xt0, RA, cs, a, Hin, cr, tx, g + 3, n1, tid);
            else
                pipe_step<true, false>(acc, xt0 + KL * XS + kl * XS + m0 + jl, wc0 + KL * WS + kl * WS + jl, ncol1, RB, wc0, xt0, RA, cs, a, Hin, cr, tx, g + 3, n1, tid);
        }
    }
    FWD_STAMP(5);                                                 // 5: GEMM1 steps
    // ---- bias (+ global conditioning), tanh . sigmoid -> U tile, K-major ---------------------------------------------------
    {
        const float* zb = a.zbias + (size_t)b * a.zb_bstride;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = 64 * cb + 32 * j + (lane & 31);
            const float za = zb[ch], zg = zb[HC + ch];
#pragma unroll
            for (int v = 0; v < 16; ++v)
                ut[ch * XS + m0 + acc_row(v, lane)] = fwd_gate(acc[j][v] + za, acc[2 + j][v] + zg);      // modules.py:152-154
        }
    }
    __syncthreads();
    FWD_STAMP(6);                                                 // 6: gate
    // ---- GEMM2: [out | skip] = U [W_out | W_skip], in column blocks of 256 ---------------------------------------------------
    for (int blk = 0; blk < nblk; ++blk) {
        const int c0 = 256 * blk;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        for (int kc = 0; kc < HC / KL; kc += 2) {
            const int g = n1p + 8 * blk + kc;
            {
                const ChunkSel cs = chunk_sel(a, g + 2, n1p, gtot, Kin, ntot);
                pipe_step<false, true>(acc, ut + kc * KL * XS + kl * XS + m0 + jl, wc0 + kl * WS + jl, ncol2, RA, wc0 + KL * WS, xt0, RB, cs, a, Hin, cr, tx, 0, n1, tid);
            }
            {
                const ChunkSel cs = chunk_sel(a, g + 3, n1p, gtot, Kin, ntot);
                pipe_step<false, true>(acc, ut + (kc + 1) * KL * XS + kl * XS + m0 + jl, wc0 + KL * WS + kl * WS + jl, ncol2, RB, wc0, xt0, RA, cs, a, Hin, cr, tx, 0, n1, tid);
            }
        }
        FWD_STAMP(7);                                             // 7: GEMM2 steps
        // epilogue in two passes: every residual / skip value this thread needs is requested first (64 loads in flight), then the
        // results are combined and stored.  (Load -> add -> store per element serialises on the memory latency: the compiler
        // cannot prove that Hout / Skip do not alias Hin; the phase trace showed 43 % of a workgroup's time here.)
        const bool interior = t0 + TM <= a.T && c0 + 256 <= ntot;        // uniform: no per-element conditions, no branches
#pragma unroll
        for (int half = 0; half < 2; ++half) {                          // two tiles at a time: 32 loads in flight, 32 registers
            float prev[2][16];
            if (interior) {
                // Addresses as  WAVE-UNIFORM row pointer (scalar registers) + ONE 32-bit per-lane offset: row v of the accumulator is
                // time step t0 + m0 + 8 (v / 4) + v % 4 (+ 4 for the upper half-wave), column gc; a 32-column tile lies entirely in
                // the residual block or entirely in the skip block.  (Per-element 64-bit addresses cost 64 VGPRs here and pushed the
                // kernel into 87 spilled registers.)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * (2 * half + j));      // first column of the tile
                    const bool res = col0 < HC;
                    const int ld = res ? HC : a.K;
                    const float* base = (res ? Hin + col0 : a.Skip + (size_t)b * a.T * a.K + (col0 - HC)) + (size_t)(t0 + m0) * ld;
                    const int loff = 4 * (lane >> 5) * ld + (lane & 31);
#pragma unroll
                    for (int v = 0; v < 16; ++v) prev[j][v] = (base + (size_t)(8 * (v >> 2) + (v & 3)) * ld)[loff];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = 2 * half + j;
                    const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * i);
                    const bool res = col0 < HC;
                    const int ld = res ? HC : a.K;
                    const float bias = a.b_os[col0 + (lane & 31)];
                    float* base = (res ? a.Hout + (size_t)b * a.T * HC + col0 : a.Skip + (size_t)b * a.T * a.K + (col0 - HC)) + (size_t)(t0 + m0) * ld;
                    const int loff = 4 * (lane >> 5) * ld + (lane & 31);
                    const float scale = res ? 0.70710678118654752440f : 1.0f;        // (out + residual) * sqrt(0.5) | skips += s
#pragma unroll
                    for (int v = 0; v < 16; ++v) (base + (size_t)(8 * (v >> 2) + (v & 3)) * ld)[loff] = (prev[j][v] + (acc[i][v] + bias)) * scale;
                }
                continue;
            }
            // edge tiles (the last time tile of an utterance, a partial column block): the same addressing with per-element
            // predicates; rows are 32-bit offsets from wave-uniform bases, so nothing 64-bit is kept per element
            const int rows_left = (int)min((long long)TM, a.T - t0) - m0;         // valid rows of this wave's 32-row block (may be <= 0)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * (2 * half + j));
                const bool res = col0 < HC;
                const int ld = res ? HC : a.K;
                const float* base = (res ? Hin : a.Skip + (size_t)b * a.T * a.K - HC) + (size_t)(t0 + m0) * ld;
                const int gc = col0 + (lane & 31);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = acc_row(v, lane);
                    prev[j][v] = (gc < ntot && row < rows_left) ? base[row * ld + gc] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = 2 * half + j;
                const int col0 = __builtin_amdgcn_readfirstlane(c0 + 128 * cb + 32 * i);
                const bool res = col0 < HC;
                const int ld = res ? HC : a.K;
                const int gc = col0 + (lane & 31);
                if (gc >= ntot) continue;
                const float bias = a.b_os[gc];
                float* base = (res ? a.Hout + (size_t)b * a.T * HC : a.Skip + (size_t)b * a.T * a.K - HC) + (size_t)(t0 + m0) * ld;
                const float scale = res ? 0.70710678118654752440f : 1.0f;            // modules.py:157-162 | wavenet.py:196-198
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = acc_row(v, lane);
                    if (row < rows_left) base[row * ld + gc] = (prev[j][v] + (acc[i][v] + bias)) * scale;
                }
            }
        }
        FWD_STAMP(8);                                             // 8: epilogue
    }
#ifdef WNV_FWD_TRACE
    if (tid == 0) {
        for (int q = 0; q < 9; ++q) atomicAdd(&g_fwd_phase[q], ph__[q]);
        atomicAdd(&g_fwd_phase[15], 1ull);
    }
#endif
}

struct HeadArgs {
    const float* Skip; float* out;                      // (B, T, K) -> (B, O, T)
    const float *w_h1, *b_h1, *w_h2, *b_h2;             // K-major [K][kp], [kp], [K][op], [op]
    long long T; int tiles_per_utt, K, kp, O, op;
    float scale;
};

__global__ void __launch_bounds__(FT, 2) wnv_fwd_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds s = carve(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TM;
    const int rb = wave & 1, cb = wave >> 1, m0 = 32 * rb;
    f16v oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) oacc[i][v] = 0.f;
    const int ncolo[4] = {128 * cb, 128 * cb + 32, 128 * cb + 64, 128 * cb + 96};
    for (int hb = 0; hb < a.K / HC; ++hb) {                  // hidden columns [128 hb, 128 hb + 128)
        f16v hacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) hacc[i][v] = 0.f;
        const int ncolh[2] = {64 * cb, 64 * cb + 32};
        for (int kc = 0; kc < a.K / KC; ++kc) {
            {   // relu(skips * sqrt(1/L)) chunk (wavenet.py:200-203), K-major
                const int m = tid >> 2, sub = tid & 3;
                const long long t = t0 + m;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (t < a.T) {
                    const float* sp = a.Skip + ((size_t)b * a.T + t) * a.K + kc * KC + 8 * sub;
                    const float4 p = *reinterpret_cast<const float4*>(sp), q = *reinterpret_cast<const float4*>(sp + 4);
                    v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s.xt[(8 * sub + e) * XS + m] = fmaxf(v[e] * a.scale, 0.f);
            }
            load_w_chunk(s.wc, a.w_h1, a.kp, a.K, a.K, kc * KC, HC * hb, tid);
            __syncthreads();
            mfma_chunk<2>(hacc, s.xt, m0, s.wc, ncolh, lane);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                        // + bias, ReLU -> hidden block, K-major
            const int hc = 64 * cb + 32 * j + (lane & 31);
            const float bias = a.b_h1[HC * hb + hc];
#pragma unroll
            for (int v = 0; v < 16; ++v) s.ut[hc * XS + m0 + acc_row(v, lane)] = fmaxf(hacc[j][v] + bias, 0.f);
        }
        __syncthreads();
        for (int kc = 0; kc < HC / KC; ++kc) {               // out += hidden block . W_h2[rows of the block]
            load_w_chunk(s.wc, a.w_h2 + (size_t)HC * hb * a.op, a.op, HC, a.op, kc * KC, 0, tid);      // padded columns are zero
            __syncthreads();
            mfma_chunk<4>(oacc, s.ut + kc * KC * XS, m0, s.wc, ncolo, lane);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 128 * cb + 32 * i + (lane & 31);
        if (o >= a.O) continue;
        const float bias = a.b_h2[o];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const long long t = t0 + m0 + acc_row(v, lane);
            if (t < a.T) a.out[((size_t)b * a.O + o) * a.T + t] = oacc[i][v] + bias;
        }
    }
}

// first_conv (wavenet.py:192): x (B, cin1, T) -> H (B, T, 128)
__global__ void wnv_fwd_first_kernel(const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bf,
                                     float* __restrict__ H, int cin1, long long T, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int r = (int)(idx % HC);
    const long long bt = idx / HC, b = bt / T, t = bt % T;
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
    const int tiles = (int)((T + TM - 1) / TM);
    const size_t lds = LDS_FLOATS * sizeof(float), lds_l = LDS_FLOATS_L * sizeof(float);
    static const bool use_v9 = [] { const char* e = getenv("WNV_FWD_V9"); return e && e[0] == '1'; }();       // the previous layer kernel, for A/B runs
    e = hipFuncSetAttribute((const void*)wnv_fwd_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)wnv_fwd_layer_kernel_v9, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)wnv_fwd_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    float *in = H0, *out = H1;
    for (int l = 0; l < m.L; ++l) {
        const WnvLayerDev& Ld = layers_host[l];
        LayerArgs la{};
        la.Hin = in; la.Hout = out; la.Skip = Skip; la.c_up = m.cin > 0 ? a.c_up : nullptr;
        la.zbias = a.zbias + (size_t)l * m.Gp; la.zb_bstride = a.zbias_bstride;
        la.w_in = d_W + Ld.w_in; la.w_os = d_W + Ld.w_os; la.b_os = d_W + Ld.b_os;
        la.T = T; la.tiles_per_utt = tiles; la.d = Ld.dilation; la.kw = m.kw; la.cin = m.cin; la.K = m.K; la.nosp = m.NOSp;
        if (use_v9) hipLaunchKernelGGL(wnv_fwd_layer_kernel_v9, dim3((unsigned)(B * tiles)), dim3(FT), lds, s, la);
        else hipLaunchKernelGGL(wnv_fwd_layer_kernel, dim3((unsigned)(B * tiles)), dim3(FT), lds_l, s, la);
        std::swap(in, out);
    }
#ifdef WNV_FWD_TRACE
    {
        unsigned long long ph[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_fwd_phase), sizeof ph);
        const double n = (double)ph[15];
        static const char* names[9] = {"prologue (first fetch)", "barrier after MFMAs", "vmcnt wait + LDS commit", "commit barrier", "issue next fetch",
                                       "GEMM1 MFMA phase", "gate", "GEMM2", "epilogue"};
        double tot = 0;
        for (int k = 0; k < 9; ++k) tot += (double)ph[k];
        fprintf(stderr, "[wnv_forward trace] %.0f workgroup-layers, cycles per workgroup (wave 0), total %.0f:\n", n, tot / n);
        for (int k = 0; k < 9; ++k) fprintf(stderr, "   %-26s %10.0f  %5.1f %%\n", names[k], (double)ph[k] / n, 100.0 * (double)ph[k] / tot);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_phase), z, sizeof z);
    }
#endif
    HeadArgs ha{};
    ha.Skip = Skip; ha.out = a.out; ha.w_h1 = d_W + m.w_h1; ha.b_h1 = d_W + m.b_h1; ha.w_h2 = d_W + m.w_h2; ha.b_h2 = d_W + m.b_h2;
    ha.T = T; ha.tiles_per_utt = tiles; ha.K = m.K; ha.kp = m.Kp; ha.O = m.O; ha.op = m.Op; ha.scale = m.skip_scale;
    hipLaunchKernelGGL(wnv_fwd_head_kernel, dim3((unsigned)(B * tiles)), dim3(FT), lds, s, ha);
    if (a.softmax) {
        const long long n = (long long)B * T;
        hipLaunchKernelGGL(wnv_fwd_softmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.out, m.O, T, n);
    }
    return hipGetLastError();
}
