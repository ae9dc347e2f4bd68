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
//   * a step is 64 MFMAs + LDS reads + loads and nothing else (see "v13" below): the weights of chunk g + 1 are DMA-ed into the
//     other LDS buffer and the B operands of chunk g + 1 are loaded from global memory during the MFMAs of chunk g, ONE barrier per
//     64 MFMAs.
// History, each step measured (profiles/r01_forward_*, r02_forward_*): v1 59.9 TFLOP/s -> v8 93.8 (prefetch in registers, LDS bank
// conflicts, scheduling barriers, straight-line epilogue) -> v9 104.1 (epilogue addressing, no spills) -> v10 109.2 (software
// pipelining, time-major 64-row tile) -> v11 109.3 (this layout) -> v13 115.1 (steps without VALU work, DMA) -> v14 121.8 (buffer
// addressing) -> v17 124.5 (biases in accumulator order, parity template) -> v18 125.6 (packed gate math).
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

// two elements at a time: the multiplies and adds as packed f32 operations (every VALU instruction between MFMAs costs matrix-pipe time)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v fwd_gate2(f2v a, f2v g) {
    const f2v ea = f2v{fabsf(a.x), fabsf(a.y)} * f2v{-2.8853900817779268f, -2.8853900817779268f};
    const f2v eg = g * f2v{-1.4426950408889634f, -1.4426950408889634f};
    const f2v e = f2v{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)}, f = f2v{__builtin_amdgcn_exp2f(eg.x), __builtin_amdgcn_exp2f(eg.y)};
    const f2v one = f2v{1.0f, 1.0f};
    const f2v den = (one + e) * (one + f);
    const f2v t = (one - e) * f2v{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    return f2v{copysignf(t.x, a.x), copysignf(t.y, a.y)};
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
    const float* c_cm;                                  // (B, cp, T): the conditioning, transposed once per call (cp = cin padded to 16, zero rows)
    const float* zbias; long long zb_bstride;           // per utterance [256]: conv bias (+ Wg g), this layer
    const float *w_in, *w_os, *b_os;                    // K-major [kw*128 + cin][256], [128][nosp], [nosp]
    long long T; int tiles_per_utt, d, kw, cin, cp, K, nosp;
};

// =================================================================================================================================
// v13: NOTHING BUT MFMAs, LDS reads AND LOADS IN THE STEPS.
// Measured on this chip (scripts/ubench_mfma_peak.hip, ubench_mfma_mix.hip): the f32 matrix pipe sustains 155 TFLOP/s from registers,
// but every ordinary VALU instruction issued between MFMAs costs 3 - 5 of the 64 cycles an MFMA takes, and so does every staging
// instruction that moves data through the vector registers (global load -> VGPR -> ds_write).  So a step of 64 MFMAs consists of
//   * the B operand (activations X[k][time]: lane = (k parity, time)) loaded STRAIGHT from the channel-major activations, one
//     global_load_dword per k pair, prefetched a step ahead: no LDS copy of the activations at all, any dilation / alignment, and the
//     conditioning is one more channel-major matrix (transposed once per call);
//   * the weights DMA-ed global -> LDS (global_load_lds_dwordx4: no registers, no ds_write), chunk g + 1 during step g;
//   * the A operand (weights W[k][channel]: lane = (k parity, row)) as ds_read_b128: tile i, row r of a wave's output is channel
//     128 (i / 4) + 4 r + i % 4, so the 8 (4) tiles of a lane's row are consecutive floats of the chunk row;
//   * addresses = wave-uniform base (scalar ALU) + a per-lane offset computed once per tile.
// The channel <-> (tile, row) map is free: the tanh and sigmoid halves still meet in the same lane and register, GEMM2's K order and
// the epilogue's row pointers follow it.
// =================================================================================================================================
constexpr int WCHL = 4096;         // floats per weight chunk buffer: [16 k][256 channels] (GEMM1) or [32 k][128 channels] (GEMM2)

// (Inline assembly on purpose: behind the builtin the compiler treats the DMA as an LDS store that may alias every later ds_read and
// puts s_waitcnt vmcnt(0) in front of the step's first operand read -- the whole load latency exposed, every step.  The wait this kernel
// needs is the one before the barrier that hands the buffer over, and it is written out there.)
__device__ __forceinline__ void dma16(const void* g, float* lds_wave_base) {        // 64 lanes x 16 bytes -> 1 KB of LDS at lds_wave_base (+ 16 lane)
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(l) : "memory");
}
// the same with the address as  wave-uniform base (scalar registers) + 32-bit per-lane byte offset: no vector ALU work at all
__device__ __forceinline__ void dma16s(const void* ubase, unsigned voff, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(ubase), "s"(l) : "memory");
}

// raw buffer access:  wave-uniform descriptor (base) + 32-bit per-lane byte offset + wave-uniform byte offset -- an address with no vector
// ALU instruction behind it (a flat / global access would build a 64-bit address per lane)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

struct Lane {                                            // per-lane constants
    unsigned w1;                                         // GEMM1 weight chunk: float4 tid of a contiguous 16 KB chunk (bytes)
    unsigned w2;                                         // GEMM2 weight chunk: rows 4 (tid / 32 + 8 q) + c, float4 tid % 32 (bytes, q = 0)
    unsigned xo;                                         // B operand: row (lane / 32), time 32 wave + lane % 32 of a (., T) matrix (bytes)
    int a1, a2;                                          // A operand reads: float offsets of this lane in a GEMM1 / GEMM2 chunk
};

// DMA of weight chunk g into LDS buffer `dst` (this wave's quarter of each of the 4 KB-quarters).  GEMM1 chunk g = rows [16 g, +16) of
// W_in (rows past the matrix: clamped -- they meet zero activations); GEMM2 chunk (blk, c) = rows {4 rr + c} x columns [128 blk, +128).
__device__ __forceinline__ void dma_chunk(const LayerArgs& a, const Lane& ln, int g, int n1, int Kin, float* dst, int tid, int wave) {
    if (g < n1) {
        const int k0 = g * KT;
        if (k0 + KT <= Kin) {
            const char* base = reinterpret_cast<const char*>(a.w_in + (size_t)k0 * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16s(base + q * 4096, ln.w1, dst + (q * FT + 64 * wave) * 4);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = q * FT + tid, k = min(k0 + (f >> 6), Kin - 1);
                dma16(a.w_in + (size_t)k * 256 + 4 * (f & 63), dst + (q * FT + 64 * wave) * 4);
            }
        }
    } else {
        const int r = g - n1, c = r & 3, blk = r >> 2;
        const char* base = reinterpret_cast<const char*>(a.w_os + (size_t)c * a.nosp + 128 * blk);
        const size_t qs = (size_t)32 * a.nosp * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16s(base + q * qs, ln.w2, dst + (q * FT + 64 * wave) * 4);
    }
}

// B operands of GEMM1 chunk g for this lane: 8 k pairs -> 8 registers.  Rows [16 (g % 8), +16) of tap g / 8 of the layer input, shifted
// by the tap's dilation (conv.py:55-61: oldest tap first), or rows of the conditioning.  Time steps before the utterance read as zero
// (uniform slow variant); time steps past T are loaded from wherever the row runs on to (inside the scratch) and never stored.
__device__ __forceinline__ void load_b(float (&xb)[8], const LayerArgs& a, const Lane& ln, int b, long long t0, int g, int wave, int lane) {
    const bool tap = g < 8 * a.kw;
    const long long shift = tap ? (long long)(a.kw - 1 - (g >> 3)) * a.d : 0;
    const float* rows = tap ? a.Hin + ((size_t)b * HC + KT * (g & 7)) * a.T : a.c_cm + ((size_t)b * a.cp + KT * (g - 8 * a.kw)) * a.T;
    const char* ubase = reinterpret_cast<const char*>(rows + (t0 - shift));       // wave-uniform; the lane offset is added LAST (scalar base +
    const size_t ks2 = (size_t)2 * a.T * 4;                                        // 32-bit vector offset: no address arithmetic in vector registers)
    if (t0 >= shift) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(ubase);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xb[ks] = buf_load(r, ln.xo, (unsigned)(ks * ks2));
    } else {
        const char* base = ubase + ln.xo;
        const long long tl = t0 + 32 * wave + (lane & 31) - shift;                  // this lane's (possibly negative) time step
        const char* basec = base + (tl < 0 ? -tl * 4 : 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float v = *reinterpret_cast<const float*>(basec + ks * ks2);
            xb[ks] = tl < 0 ? 0.f : v;
        }
    }
}

// ---- one GEMM1 step: 64 MFMAs out of weight buffer wc and the B registers xb; in their shadow the DMA of chunk g + 1 into wc_n and
// the B loads of chunk g + 1 into xn.  Ends in the one barrier (after the DMA has landed).
__device__ __forceinline__ void gemm1_step(f16v (&acc)[8], const float (&xb)[8], float (&xn)[8], const float* wc, float* wc_n, const LayerArgs& a, const Lane& ln,
                                           int b, long long t0, int g, int n1, int Kin, int tid, int lane, int wave) {
    const float4* Wa = reinterpret_cast<const float4*>(wc + ln.a1);
    float4 av0 = Wa[0], av1 = Wa[32];
    dma_chunk(a, ln, g + 1, n1, Kin, wc_n, tid, wave);
    load_b(xn, a, ln, b, t0, min(g + 1, n1 - 1), wave, lane);               // (after the last chunk: a harmless re-load)
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) {
        float4 an0 = av0, an1 = av1;
        if (ks + 1 < KT / 2) {
            an0 = Wa[(2 * ks + 2) * 64];
            an1 = Wa[(2 * ks + 2) * 64 + 32];
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, xb[ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, xb[ks], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.z, xb[ks], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.w, xb[ks], acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, xb[ks], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, xb[ks], acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.z, xb[ks], acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.w, xb[ks], acc[7], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        av0 = an0; av1 = an1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): the DMA of the next chunk has landed
    __syncthreads();
}

// ---- one GEMM2 step: 64 MFMAs (16 k-pairs x 4 channel tiles).  B operand = the gated accumulator registers u[0 .. 15] of gate tile c
// (register s of a lane holds gate channel 4 (8 (s / 4) + 4 (lane / 32) + s % 4) + c: the k pair of MFMA s); A = row
// 8 (s / 4) + 4 (lane / 32) + s % 4 of the weight chunk [32 k][128 channels], one ds_read_b128 for the four tiles.
__device__ __forceinline__ void gemm2_step(f16v (&acc)[4], const f16v& u, const float* wc, float* wc_n, const LayerArgs& a, const Lane& ln, int g, int gtot, int n1,
                                           int Kin, int tid, int lane, int wave) {
    const float4* Wa = reinterpret_cast<const float4*>(wc + ln.a2);
    float4 av = Wa[0];
    if (g + 1 < gtot) dma_chunk(a, ln, g + 1, n1, Kin, wc_n, tid, wave);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float4 an = av;
        if (s + 1 < 16) {
            an = Wa[(8 * ((s + 1) >> 2) + ((s + 1) & 3)) * 32];
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, u[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, u[s], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, u[s], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, u[s], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        av = an;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
}

template <bool ODD>                                  // ODD: the number of GEMM1 chunks, 8 kw + cp / 16, is odd
__global__ void __launch_bounds__(FT, 2) wnv_fwd_layer_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wc0 = smem;                                       // [2][WCHL] weight chunks
    float* const bz = wc0 + 2 * WCHL;                              // [256] gate bias (+ Wg g), then [128 + K] b_out | b_skip
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.tiles_per_utt;
    const long long t0 = (long long)(blockIdx.x % a.tiles_per_utt) * TN;
    const int Kin = a.kw * HC + a.cin;                             // rows of W_in
    const int n1 = 8 * a.kw + a.cp / KT;                           // GEMM1 steps (the conditioning is padded to whole chunks)
    const int ntot = HC + a.K, nblk = ntot / HC, gtot = n1 + 4 * nblk;
    const int jl = lane & 31, kl = lane >> 5;
    Lane ln;
    ln.w1 = 16u * (unsigned)tid;
    ln.w2 = 4u * (4u * (unsigned)(tid >> 5) * (unsigned)a.nosp + 4u * (unsigned)(tid & 31));
    ln.xo = 4u * ((unsigned)kl * (unsigned)a.T + 32u * (unsigned)wave + (unsigned)jl);          // (T <= 2^23: the host checks)
    ln.a1 = kl * 256 + 4 * jl;
    ln.a2 = 4 * kl * 128 + 4 * jl;
#ifdef WNV_FWD_TRACE
    unsigned long long ph__[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev__ = __builtin_readcyclecounter();
    const unsigned long long wall0__ = wall_clock64();
#endif
    // ---- prologue: biases, weight chunk 0, B operands of chunk 0 ----------------------------------------------------------------
    float xa[8], xb[8];
    dma_chunk(a, ln, 0, n1, Kin, wc0, tid, wave);
    load_b(xa, a, ln, b, t0, 0, wave, lane);
    // biases in ACCUMULATOR order: [tile][lane / 32][v / 4][v % 4], so that one ds_read_b128 fills four consecutive accumulator registers
    // (channel of (tile i, row r): 128 (i / 4) + 4 r + i % 4 in GEMM1, 4 r + i in a GEMM2 block; row = 8 (v / 4) + 4 (lane / 32) + v % 4)
    {
        const int i = tid >> 5, r = tid & 31;                       // tid = 32 i + r  ->  slot [i][(r >> 2) & 1][r >> 3][r & 3]
        const int slot = i * 32 + ((r >> 2) & 1) * 16 + (r >> 3) * 4 + (r & 3);
        bz[slot] = a.zbias[(size_t)b * a.zb_bstride + 128 * (i >> 2) + 4 * r + (i & 3)];
        for (int j = tid; j < ntot; j += FT) {
            const int blk = j >> 7, ii = (j >> 5) & 3, rr = j & 31;
            bz[256 + blk * 128 + ii * 32 + ((rr >> 2) & 1) * 16 + (rr >> 3) * 4 + (rr & 3)] = a.b_os[128 * blk + 4 * rr + ii];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    // the gate bias (+ global conditioning) is the accumulators' initial value (LDS reads, no vector ALU work); tile i, row r = gate
    // row 128 (i / 4) + 4 r + i % 4
    f16v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = bz[i * 32 + kl * 16 + v];
    FWD_STAMP(0);                                                 // 0: prologue
    // ---- GEMM1: Z^T = W_in^T [taps | c]^T, two steps per iteration (the B register sets swap roles) ------------------------------
    // Two steps per iteration, the B register sets swapping roles; an odd chunk count ends in one more step -- a compile-time property
    // of the instantiation (ODD), because alternatives that MERGE at run time (a last step without prefetch, role-swapped sets) make the
    // register allocator copy or spill the 128 accumulator registers at the merge.
    {
        int g = 0;
        for (; g + 1 < n1; g += 2) {
            gemm1_step(acc, xa, xb, wc0, wc0 + WCHL, a, ln, b, t0, g, n1, Kin, tid, lane, wave);
            gemm1_step(acc, xb, xa, wc0 + WCHL, wc0, a, ln, b, t0, g + 1, n1, Kin, tid, lane, wave);
        }
        if constexpr (ODD) gemm1_step(acc, xa, xb, wc0, wc0 + WCHL, a, ln, b, t0, g, n1, Kin, tid, lane, wave);
    }
    FWD_STAMP(5);                                                 // 5: GEMM1 steps
    // ---- tanh . sigmoid, in registers: u[i][v] = gate channel 4 row + i of this lane's time step (modules.py:152-154) ---------------
    //      In place: the gated values take the tanh tiles' registers, the sigmoid tiles' registers become GEMM2's accumulators.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const f2v r = fwd_gate2(f2v{acc[i][v], acc[i][v + 1]}, f2v{acc[4 + i][v], acc[4 + i][v + 1]});
            acc[i][v] = r.x; acc[i][v + 1] = r.y;
        }
    f16v (&u)[4] = *reinterpret_cast<f16v(*)[4]>(&acc[0]);
    f16v (&o)[4] = *reinterpret_cast<f16v(*)[4]>(&acc[4]);
    FWD_STAMP(6);                                                 // 6: gate
    // ---- GEMM2: [out | skip]^T = [W_out | W_skip]^T U^T in blocks of 128 output channels; epilogue per block --------------------------
    const long long tl = t0 + 32 * wave + jl;                       // this lane's time step
    const unsigned eoff = 4u * (unsigned)(16 * kl * a.T + min(tl, a.T - 1));      // (the host checks T <= 2^23: offset + scalar offset < 2^31)
    const bool live = tl < a.T;
    for (int blk = 0; blk < nblk; ++blk) {
        // block 0 is the residual output (modules.py:157-162: (out + x) sqrt(.5)), the others accumulate into the skip sum
        // (wavenet.py:196-198).  Row = channel, lane = time: every access is a 128-byte run along time.
        const bool res = blk == 0;
        const float* src = res ? a.Hin + (size_t)b * HC * a.T : a.Skip + ((size_t)b * a.K + (size_t)(blk - 1) * HC) * a.T;
        float* dst = res ? a.Hout + (size_t)b * HC * a.T : a.Skip + ((size_t)b * a.K + (size_t)(blk - 1) * HC) * a.T;
        const float* bias = bz + 256 + HC * blk;
        // the accumulators start from the bias (LDS reads: no vector ALU work)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[i][v] = bias[i * 32 + kl * 16 + v];
        // what the block's result is added to (the layer input | the running skip sum) is requested BEFORE the block's four GEMM2 steps and
        // consumed after them: the read of the read-modify-write costs no waiting.  Channel 32 (v / 4) + 4 (v % 4) + i of the block =
        // buffer descriptor of the 32-channel group v / 4 + wave-uniform row offset + the lane's offset (16 (lane / 32) rows + time)
        float prev[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (size_t)32 * q * a.T);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) prev[i][4 * q + e] = buf_load(rs, eoff, (unsigned)((4 * e + i) * a.T * 4));
        }
        const int g2 = n1 + 4 * blk;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int par = (g2 + c) & 1;
            gemm2_step(o, u[c], wc0 + par * WCHL, wc0 + (par ^ 1) * WCHL, a, ln, g2 + c, gtot, n1, Kin, tid, lane, wave);
        }
        FWD_STAMP(7);                                             // 7: GEMM2 steps
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(dst + (size_t)32 * q * a.T);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = 4 * q + e;
                    const float r = res ? (prev[i][v] + o[i][v]) * 0.70710678118654752440f : prev[i][v] + o[i][v];
                    if (live) buf_store(r, rd, eoff, (unsigned)((4 * e + i) * a.T * 4));
                }
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

// the conditioning, time-major (B, T, cin) as the upsampler writes it -> channel-major (B, cp, T), rows cin .. cp - 1 zero
__global__ void wnv_fwd_ctrans_kernel(const float* __restrict__ c, float* __restrict__ out, int cin, int cp, long long T) {
    __shared__ float tile[64][129];                                // [time][channel]
    const int b = blockIdx.y;
    const long long t0 = (long long)blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int nt = (int)min((long long)64, T - t0);
    const float* src = c + ((size_t)b * T + t0) * cin;
    for (int i = tid; i < nt * cin; i += 256) tile[i / cin][i % cin] = src[i];          // contiguous rows
    __syncthreads();
    for (int i = tid; i < cp * 64; i += 256) {
        const int ch = i >> 6, t = i & 63;
        if (t < nt) out[((size_t)b * cp + ch) * T + t0 + t] = ch < cin ? tile[t][ch] : 0.f;
    }
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
    // staging registers of one GEMM1 chunk: thread (r = tid / 16, m8 = 8 (tid % 16)) carries 8 time steps of skip row r; threads tid and
    // FT + tid carry one float4 each of the [16 k][128 hidden] weight chunk
    const int sr = tid >> 4, m8 = 8 * (tid & 15);
    const bool whole = t0 + TN <= a.T && (a.T & 3) == 0;          // the tile lies inside the utterance and its rows are 16-byte aligned
    float4 sv0, sv1, wv0, wv1;
    auto load_chunk = [&](int hb, int kc) {
        const float* row = a.Skip + ((size_t)b * a.K + kc * KT + sr) * a.T + t0 + m8;
        if (whole) {
            sv0 = *reinterpret_cast<const float4*>(row);
            sv1 = *reinterpret_cast<const float4*>(row + 4);
        } else {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = t0 + m8 + e < a.T ? row[e] : 0.f;
            sv0 = make_float4(v[0], v[1], v[2], v[3]);
            sv1 = make_float4(v[4], v[5], v[6], v[7]);
        }
        const float* w = a.w_h1 + (size_t)(kc * KT + (tid >> 5)) * a.kp + HC * hb + 4 * (tid & 31);
        wv0 = *reinterpret_cast<const float4*>(w);
        wv1 = *reinterpret_cast<const float4*>(w + (size_t)8 * a.kp);
    };
    auto store_chunk = [&](int buf) {
        float* x = xt + buf * (KT * XT) + sr * XT + m8;
        *reinterpret_cast<float4*>(x) = make_float4(fmaxf(sv0.x * a.scale, 0.f), fmaxf(sv0.y * a.scale, 0.f), fmaxf(sv0.z * a.scale, 0.f), fmaxf(sv0.w * a.scale, 0.f));
        *reinterpret_cast<float4*>(x + 4) = make_float4(fmaxf(sv1.x * a.scale, 0.f), fmaxf(sv1.y * a.scale, 0.f), fmaxf(sv1.z * a.scale, 0.f), fmaxf(sv1.w * a.scale, 0.f));
        float4* w = reinterpret_cast<float4*>(wc + buf * (KT * HC));
        w[tid] = wv0;
        w[FT + tid] = wv1;
    };
    for (int hb = 0; hb < a.K / HC; ++hb) {                      // hidden channels [128 hb, 128 hb + 128)
        f16v hacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) hacc[i][v] = 0.f;
        // GEMM1 of this hidden block, software-pipelined: chunk kc + 1 (16 skip channels x 128 time steps, ReLU and scale applied on the
        // way -- wavenet.py:200-203 -- and W_h1 rows [16 kc, +16) x hidden columns [128 hb, +128)) travels global -> registers -> the
        // other LDS buffer while the MFMAs of chunk kc run: one barrier per chunk
        const int n1 = a.K / KT;
        load_chunk(hb, 0);
        store_chunk(0);
        __syncthreads();
        for (int kc = 0; kc < n1; ++kc) {
            const bool more = kc + 1 < n1;
            if (more) load_chunk(hb, kc + 1);
            const float* xc = xt + (kc & 1) * (KT * XT);
            const float* wk = wc + (kc & 1) * (KT * HC);
#pragma unroll
            for (int ks = 0; ks < KT / 2; ++ks) {
                const float bv = xc[(2 * ks + kl) * XT + 32 * wave + jl];
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[(2 * ks + kl) * HC + 32 * i + jl], bv, hacc[i], 0, 0, 0);
            }
            if (more) store_chunk((kc + 1) & 1);
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

// first_conv of one-hot / soft-input models (cin1 = out_channels > 1): the same product on the matrix pipe, D[channel][time] =
// W (128 x cin1) . x (cin1 x 32) per wave.  A workgroup of four waves takes 128 consecutive time steps of one utterance; both operands
// come straight from memory (w_first is k-major, so a half wave's A load is one 128-byte row piece; x is channel-major, so a half
// wave's B load is 32 consecutive time steps); the 128-KB matrix stays in the L2 / L1.  (The scalar loop above spent 4 ms on the
// 256-class models at the benchmark shape -- 22 % of the whole forward -- re-reading x 128 times through the caches.)
__global__ void __launch_bounds__(256) wnv_fwd_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bf,
                                                                 float* __restrict__ H, int cin1, long long T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const long long t0 = (long long)blockIdx.x * TN + 32 * wave;
    if (t0 >= T) return;
    const long long t = t0 + (lane & 31), tc = t < T ? t : T - 1;
    const int kh = lane >> 5;
    const float* xb = x + (size_t)b * cin1 * T + tc;
    const float* wa = wf + (lane & 31);
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = bf[32 * i + acc_row(v, lane)];
    for (int k0 = 0; k0 < cin1; k0 += 8) {                         // four k pairs per trip: 20 loads in flight, then 16 MFMAs
        float xv[4], wv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 2 * j + kh;
            const bool in = k < cin1;                              // cin1 not a multiple of 8: the empty k contribute zeros
            const int kc = in ? k : cin1 - 1;
            const float xl = xb[(size_t)kc * T];                   // (unconditional load of a clamped row, then a select: no branch)
            xv[j] = in ? xl : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[j][i] = wa[(size_t)kc * HC + 32 * i];
        }
        __builtin_amdgcn_sched_barrier(0);                         // all 20 loads issued before the first MFMA waits for its operands
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][i], xv[j], acc[i], 0, 0, 0);
    }
    if (t < T) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) H[((size_t)b * HC + 32 * i + acc_row(v, lane)) * T + t] = acc[i][v];
    }
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
    if (m.cin > 128) return "needs cin_channels <= 128";
    return nullptr;
}

size_t wnv_forward_scratch_floats(const WnvModelDev& m, int B, long long T) {
    const int cp = (m.cin + KT - 1) / KT * KT;
    return (size_t)B * T * (2 * HC + m.K + cp) + 1024;       // activations (ping, pong), skip sum, channel-major conditioning, slack: a tile's
                                                             // last B-operand loads may run up to 127 floats past a row
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
        if (m.cin1 > 1)
            hipLaunchKernelGGL(wnv_fwd_first_mfma_kernel, dim3((unsigned)((T + TN - 1) / TN), (unsigned)B), dim3(256), 0, s, a.x, d_W + m.w_first,
                               d_W + m.b_first, H0, m.cin1, T);
        else
            hipLaunchKernelGGL(wnv_fwd_first_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.x, d_W + m.w_first, d_W + m.b_first,
                               H0, m.cin1, T, n);
    }
    const int tiles = (int)((T + TN - 1) / TN);
    const int cp = (m.cin + KT - 1) / KT * KT;
    float* c_cm = Skip + (size_t)B * T * m.K;
    if (m.cin > 0)
        hipLaunchKernelGGL(wnv_fwd_ctrans_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)B), dim3(256), 0, s, a.c_up, c_cm, m.cin, cp, T);
    const size_t lds_l = ((size_t)2 * WCHL + 256 + HC + m.K) * sizeof(float);
    const size_t lds_h = ((size_t)2 * KT * XT + 2 * WCH) * sizeof(float);
    const bool odd = ((8 * m.kw + cp / KT) & 1) != 0;
    e = odd ? hipFuncSetAttribute((const void*)wnv_fwd_layer_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l)
            : hipFuncSetAttribute((const void*)wnv_fwd_layer_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l);
    if (e != hipSuccess) return e;
    float *in = H0, *out = H1;
    for (int l = 0; l < m.L; ++l) {
        const WnvLayerDev& Ld = layers_host[l];
        LayerArgs la{};
        la.Hin = in; la.Hout = out; la.Skip = Skip; la.c_cm = c_cm; la.cp = cp;
        la.zbias = a.zbias + (size_t)l * m.Gp; la.zb_bstride = a.zbias_bstride;
        la.w_in = d_W + Ld.w_in; la.w_os = d_W + Ld.w_os; la.b_os = d_W + Ld.b_os;
        la.T = T; la.tiles_per_utt = tiles; la.d = Ld.dilation; la.kw = m.kw; la.cin = m.cin; la.K = m.K; la.nosp = m.NOSp;
        if (odd) hipLaunchKernelGGL(wnv_fwd_layer_kernel<true>, dim3((unsigned)(B * tiles)), dim3(FT), lds_l, s, la);
        else hipLaunchKernelGGL(wnv_fwd_layer_kernel<false>, dim3((unsigned)(B * tiles)), dim3(FT), lds_l, s, la);
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
