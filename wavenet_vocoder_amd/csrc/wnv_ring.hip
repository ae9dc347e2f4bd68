// wnv_ring.hip -- the pipelined, weight-stationary sample-loop kernel for gfx950 ("ring" kernel).
//
// WHY.  One autoregressive step is a chain of L gated layers; for 8 utterances the per-step cost is the latency of
// that chain, not bandwidth or FLOPs.  No CU can hold the model and streaming the weights every
// step costs more than the chain itself, so the model is spread over the chip and stays there: every utterance slot
// gets a RING of persistent workgroups, one per CU, each owning ONE layer,
//
//     head -> stage 0 (layer 0) -> stage 1 (layer 1) -> ... -> stage L-1 -> head -> ...      (8 rings, ring r on XCD r)
//
// plus one TAP workgroup per layer that serves all rings.  One launch runs all T steps; the only synchronisation is the
// data flow itself.
//
//   * stage (run_stage): gate-to-gate chain -- with M_l = sqrt(.5) W_cur,l W_o,l-1 folded on the host, only
//     M_l u_{l-1} -> tanh.sigmoid -> send u_l  is on the chain (weights in VGPRs, 8-lane K split, reduce-scatter over
//     DPP, hardware exp/rcp); conv1x1_out (the h recurrence, evaluated exactly as the reference does), N_l h_{l-1},
//     the skip 1x1 and the hand-over to the tap workgroup all run behind the send;
//   * hand-off: the 128-float vector hops CU -> CU as data-tagged 8-byte granules {tag, value}: one store per value,
//     the consumer re-reads until every tag matches -- no flag, no fence.  Between workgroups that verified at start-up
//     (HW_REG_XCC_ID exchange) that they share an XCD the stores are plain and stay in that XCD's L2 (~0.45 us per hop
//     incl. LDS write + barrier, scripts/ubench_hop.hip); otherwise write-through (sc1) stores, placement-independent
//     (~0.65-0.8 us).  Tags are launch-unique, mailboxes are never re-zeroed;
//   * tap workgroup (run_tap): the dilated conv's older taps + the local-conditioning 1x1 of its layer for EVERY
//     utterance, matrix resident in VGPRs + LDS, history rings owned here; records of write-through tagged granules carry
//     h_l[t] in and pre_l[t+1] out with a whole step of slack;
//   * head (run_head): skip sum -> output MLP (registers) -> sampler -> first_conv of the next step.
//
// Every wait is bounded: a spin that exceeds its budget writes a code to `status` and every workgroup drains out
// (WNV_ERR_TIMEOUT).
//
// Reference semantics: wavenet.py:296-336 (loop), modules.py:127-163 (layer), conv.py:33-45 (history taps),
// mixture.py:118-156 / 221-270 (samplers).  Numerics: fp32 throughout; the association order of the dot products and
// the M/N folding differ from ATen's evaluation order (<= 5e-7 on the head outputs; parity tolerance 1e-4).
#include "wnv_ring.h"

#ifndef WNV_RING_IS_DEFAULT
#define WNV_RING_IS_DEFAULT 1   // the GPU parity suite of the ring kernel is green (tests/test_gpu_ring.py)
#endif

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "wnv_knobs.h"
#include "wnv_matvec.h"
#include "wnv_sample.h"

namespace {

constexpr int RT = 512;            // threads per workgroup
constexpr int RW = 8;              // waves per workgroup
constexpr int RC = 128;            // residual channels this kernel is specialised for
constexpr int GC = 256;            // gate channels
constexpr int QS = 36;             // LDS stride of one 32-float quarter (+4 pad: the 4 quarters hit disjoint banks)
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef float f2 __attribute__((ext_vector_type(2)));

struct RingParams {
    int n_rings, rstride, S, L, B, T, Tt, upr;
    int K, Kp, O, cin, kw, kpre, nz, dist;
    int NH, Op;                        // head parts per ring (= K / 128: part j owns hidden units [128 j, 128 j + 128)); partial-output stride
    int pstride;                       // LDS partial stride (floats) = max(256, Kp)
    int hist_floats;
    int allow_fast;                    // same-XCD hand-off through the XCD's L2 (plain stores) when placement allows
    unsigned tag_base;                 // tags of this launch are tag_base + t + 1: unique across launches, no re-zeroing
    float skip_scale;
    const float *w2img, *wnimg, *woimg, *wsimg, *bo, *wpre, *bskip, *cvec;
    const float *wh1img, *bh1, *wh2img, *bh2, *wfirst, *bfirst;   // one-hot models: wh2img holds two row images (rows i, 128 + i), wfirst is K-major [cin1][128]
    int cin1, softmax, quantize;       // first_conv input channels (1 = scalar input); categorical head switches (wavenet.py:332-335)
    int split, sA, qh;                 // split rings (run_stage_split): layers on TWO CUs; stages 1 .. sA share the head's XCD; qh = 2: two partial vectors per Q / skip slot
    const float *w2s, *wns, *wos, *wss; // split rings: per (layer, half) row images of M, N (wave-group mapping, 4 slots), conv1x1_out / conv1x1_skip K-halves
    int head_l0;                       // the head evaluates layer 0 itself (scalar-input models, K = 128): position 0 of a ring stays empty
    const float* l0vec;                // [4][256]: W_cur,0 w_first, W_cur,0 b_first (layer 0's pre-activation is affine in the sample); N_1 w_first, N_1 b_first (so is N_1 h_0)
    int* index_out;
    const float* zbias;                // effective conv bias of the generic pack: [B or 1][L][zb_ld], rows = the model's own G gate rows
    long long zbias_bstride;
    int zb_ld, gh;                     // its row stride (G padded to 4) and the model's G / 2: padded output n -> row (n >> 7) * gh + (n & 127)
    const int *lay_dil, *lay_histoff;
    unsigned long long *xmail, *hmail, *smail;   // chain inputs X[b][S+1][128]; residual increments Q[b][2 (t parity)][S+1][128] (slot l+1 = conv1x1_out(u_l) + b_o,l); skip sums
    unsigned long long *zmail;                   // head_l0: Z[b][256] = N_1 h_0[t] (affine in the sample: made by the head), read by stage 1
    unsigned long long *gmail;                   // layer inputs handed on: G[b][2 (t parity)][S+1][128], slot j = h_{j-1}[t] as stage j formed it (read by stage j + 1)
    unsigned long long *omail;                   // head parts j > 0 -> part 0: partial head outputs O[b][NH][Op]
    unsigned long long *fmail, *pmail;           // records: stage -> tap workgroup h_l[t]: F[b][L][step parity][128] granules; tap workgroup -> stage pre_l[t+1]: P[b][L][step parity][256]
    int ring_blocks, tap_parts;                  // blocks [0, ring_blocks) = rings, then tap_parts tap workgroups per layer (part q serves passes q, q + parts, ...)
    int tb;                                      // utterances per tap pass (<= TB)
    int kper, kreg_rows, klds_rows;              // tap workgroup: K rows per wave; of those resident in VGPRs / in LDS (the rest streams)
    unsigned int* xcc;                 // [grid] XCC id + 1 of every workgroup (placement handshake)
    float* hist;
    const float* zero_row;             // 128 zeros behind the history rings (packed slots: what lies before an utterance's first step)
    const float *c_up, *initial, *teacher, *noise;
    unsigned long long seed;
    int b0, noise_B;                   // this launch is utterances [b0, b0 + B) of a call of noise_B (noise tape / Philox stream addressing)
    const unsigned* noise_ready;       // streamed tape (coherent host memory): steps [0, *noise_ready) of `noise` are valid; null: all of it
    const int *seg_start, *seg_uid;    // PACKED SLOTS (wnv_generate_args, ABI 4): [B][T] each -- the step at which the utterance occupying slot b at step t
                                       // began, and its id in the job; null: one utterance per row.  An utterance's history before its first step reads as
                                       // zeros (tap workgroups), its first input is zeros / one-hot 127 and its noise stream is (uid, t - start) (heads)
    const int* seg_gid;                // packed slots of a model with global conditioning: [B][T] row of zbias (speaker / utterance of the job) of the
                                       // utterance occupying slot b at step t; zbias then has one row per speaker / utterance, not per slot
    float *out, *params_out;
    unsigned int* status;
    unsigned long long* trace;         // optional [T_trace][upr][S+1][16] wall-clock stamps of ring 0's utterances (debug)
    unsigned long long* trace_tap;     // optional [T_trace][16] stamps of the first pass of layer 0's tap workgroup, part 0
    int trace_t0, trace_n;
    int tap_mfma, tap_nj;              // (round 6) the tap role on the matrix pipe (run_tap_mf: the _mf kernels): K groups of 16 rows (<= TAP_NJR + TAP_NJL)
};

using u64 = unsigned long long;

// A wave-uniform int from memory the kernel never writes (the packed-slot maps): through the CONSTANT address space, so that the load is
// a scalar one (s_load_dword: no vector register, its own counter -- a wait for it does not wait for the acknowledgement of the chain
// stores in flight, as a wait for a vector load does on gfx9).  The caller makes the address uniform (readfirstlane where it is not provably so).
typedef const int __attribute__((address_space(4))) wnv_cint;
__device__ __forceinline__ int uniform_ld(const int* p) { return *reinterpret_cast<wnv_cint*>(reinterpret_cast<size_t>(p)); }

__device__ __forceinline__ u64 ld_granule(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1 (L1 bypass)
}
// One 8-byte {tag, value} granule.  slow: write-through (sc1) store, visible to every XCD.  fast: plain store that
// stays in THIS XCD's L2 -- legal only when the consumer workgroup was verified to sit on the same XCD (its sc1
// loads are served by that L2); ~0.2 us less per hop (scripts/ubench_hop.hip).
__device__ __forceinline__ void st_granule(u64* p, unsigned tag, float v, bool fast) {
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}

// two adjacent granules {tag, v0} {tag, v1} in one 16-byte store (a wave hands on a whole 128-value vector with one instruction)
__device__ __forceinline__ void st_granule2(u64* p, unsigned tag, float v0, float v1, bool fast) {
    typedef unsigned u4s __attribute__((ext_vector_type(4)));
    const u4s x = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    // (s_nop 1: a VMEM store of more than 64 bits reads its data registers AFTER issue -- a VALU write to them needs wait states in
    //  between, which the compiler's hazard recogniser inserts for stores it emits itself but cannot for one inside inline assembly.
    //  Found in round 4: the tap workgroups' second record store had its address computed into the last two data registers of the
    //  first -- lanes 12-15 of every 16 stored the POINTER where the tag belongs and the receiving stages waited for ever.)
    if (fast) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}

// RECORDS: the vectors nobody on the chain waits for -- stage -> tap workgroup h_l[t] (128 values), tap workgroup -> stage pre_l[t+1]
// (256 values), any XCD -- travel as data-tagged granules like everything else: 16-byte WRITE-THROUGH stores of two granules each,
// no drain, no flag.  (Until round 4 these were "bulk records": raw floats, the producer drained its write-through stores --
// s_waitcnt vmcnt(0), ~1 us -- and then published one tag granule.  Nobody on the chain waited for that drain, but with several
// utterances per ring the stage's wave 0 spent it once per utterance INSIDE the stage's occupancy, and a tap pass once per pass:
// profiles/r04_tap_bound_experiment.txt -- the rings alone saturated at 2.77 MSamples/s, 2.9 us per utterance and stage.)
// The reader first polls the record's FIRST 16 bytes from every lane (one request) at a relaxed cadence -- a record can be most of a
// step away, and hundreds of waves polling whole records would load the fabric the chain's hops share -- and, once that tag shows,
// the whole record until every tag matches.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4v ld16_sc1(const u64* p) {
    u4v x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
__device__ __forceinline__ void ld16x2_sc1(const u64* p, const u64* q, u4v& a, u4v& b) {   // granules p[0..1] and q[0..1]: one round trip
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void ld16x4_sc1(const u64* p0, const u64* p1, const u64* p2, const u64* p3, u4v& a, u4v& b, u4v& c, u4v& d) {   // ... and four
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// a 256-value record: granule n = value n; the receiving lane L takes granules 4L .. 4L + 3 in two 16-byte loads
__device__ __forceinline__ int rec4_a(int lane) { return 4 * lane; }
__device__ __forceinline__ int rec4_b(int lane) { return 4 * lane + 2; }
#ifdef WNV_DBG_MARK
#define WNV_MARK(m, x) do { if ((m) && lane == 0) __hip_atomic_store((m), (unsigned)(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define WNV_MARK(m, x) ((void)0)
#endif
template <int NG>
__device__ __forceinline__ bool rec_recv(const u64* rec, unsigned tag, float (&v)[2 * NG], unsigned int* status, unsigned code, int lane, unsigned* mark = nullptr) {
    static_assert(NG == 1 || NG == 2, "");
    unsigned spins = 0;
    WNV_MARK(mark, 0x10000u | (tag & 0xffffu));
    for (;;) {
        const u4v x = ld16_sc1(rec);                               // every lane the same 16 bytes: one request
        if (mark) WNV_MARK(mark + 256, x.y);
        if (x.y == tag) break;
        if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > (SPIN_LIMIT >> 3)) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
        __builtin_amdgcn_s_sleep(8);
    }
    spins = 0;
    WNV_MARK(mark, 0x20000u | (tag & 0xffffu));
    for (;;) {
        bool ok;
        if constexpr (NG == 1) {
            const u4v x = ld16_sc1(rec + 2 * lane);
            v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.z);
            ok = x.y == tag && x.w == tag;
        } else {
            u4v x, y;
            ld16x2_sc1(rec + rec4_a(lane), rec + rec4_b(lane), x, y);
            v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.z); v[2] = __uint_as_float(y.x); v[3] = __uint_as_float(y.z);
            ok = x.y == tag && x.w == tag && y.y == tag && y.w == tag;
#ifdef WNV_DBG_MARK
            if (mark && blockIdx.x == 16) {
                __hip_atomic_store(mark - blockIdx.x + 512 + lane, (x.y & 0xffu) | ((x.w & 0xffu) << 8) | ((y.y & 0xffu) << 16) | ((y.w & 0xffu) << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mark - blockIdx.x + 576 + lane, (unsigned)(size_t)(rec + rec4_a(lane)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mark - blockIdx.x + 640 + lane, x.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mark - blockIdx.x + 576 + lane, x.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
        }
        if (__all(ok)) { WNV_MARK(mark, 0x30000u | (tag & 0xffffu)); return true; }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}

// ONE wave receives a whole 128-value vector: two adjacent granules per lane in one 16-byte L1-bypassing load (load and wait
// are one asm statement: no register is ever in flight where the compiler can see it).  (Two polling waves see an arrival at
// the later of two independent poll phases; one wave sees it at its own.)
__device__ __forceinline__ bool wave_recv2(const u64* g2, unsigned tag, float& v0, float& v1, unsigned int* status,
                                           unsigned code, int lane, bool have_first, u4v first) {
    unsigned spins = 0;
    for (;;) {
        u4v x = first;
        if (!have_first) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(g2) : "memory");
        have_first = false;
        v0 = __uint_as_float(x.x); v1 = __uint_as_float(x.z);
        if (__all(x.y == tag && x.w == tag)) return true;
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) {
                if (lane == 0) atomicCAS(status, 0u, code);
                return false;
            }
        }
    }
}

// One wave waits until the granule of every ACTIVE lane carries `tag`; returns false on abort/timeout.
template <bool SLEEP>
__device__ __forceinline__ bool wave_recv(const u64* g, bool active, unsigned tag, float& v, unsigned int* status,
                                          unsigned code, int lane, bool have_first = false, u64 first = 0) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        if (active) {
            const u64 x = have_first ? first : ld_granule(g);
            have_first = false;
            v = __uint_as_float((unsigned)x);
            ok = (unsigned)(x >> 32) == tag;
        }
        if (__all(ok)) return true;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) {
                if (lane == 0) atomicCAS(status, 0u, code);
                return false;
            }
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(1);
    }
}

// ---- POLLS IN RESERVED REGISTERS ------------------------------------------------------------------------------------------
// A poll whose load is issued EARLY (so that its L2 round trip, ~0.25 us even when the granule is already there, overlaps the
// work in between) writes its destination registers whenever it lands, which the compiler knows nothing about: with a
// compiler-allocated destination it may copy or re-use the register while the load is still out (it did: a v_mov of the
// in-flight pair ahead of the wait).  So these polls land in PHYSICAL registers the compiler never allocates: the kernel is
// capped at 244 VGPRs (amdgpu_num_vgpr; tests/test_host_cpu.py checks the ISA) and v244 .. v255 -- three slots of up to 16 bytes
// per lane -- belong to the helpers below; values leave them through v_mov after an s_waitcnt.  Loads return in order, so
// "s_waitcnt vmcnt(n)" = everything but the n youngest polls has landed; older leftovers (stores) only make that wait longer,
// never shorter, and a re-issue into a slot whose previous load is still out is harmless (the younger load lands last).
// Nothing ever has to be drained.
//
// Measured and dropped: keeping SEVERAL polls of the same vector in flight while waiting (to see an arrival within a fraction of
// a round trip instead of a uniformly distributed 0 .. 1 round trips late).  egs/mol, B = 8: one poll at a time 432 kSamples/s,
// two in flight 408, three 391, two in flight on a replicated vector (different cache lines, a second store on the chain) 397:
// the hop itself gets slower when the line is polled harder.
#define WNV_RSV_CLOBBERS "memory", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
template <int SLOT> __device__ __forceinline__ void rpoll16_issue(const u64* p) {
    if constexpr (SLOT == 0) asm volatile("global_load_dwordx4 v[244:247], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
    else if constexpr (SLOT == 1) asm volatile("global_load_dwordx4 v[248:251], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
    else asm volatile("global_load_dwordx4 v[252:255], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
}
// waits until at most YOUNGER polls issued after this slot's are outstanding, then copies the slot out
template <int SLOT, int YOUNGER> __device__ __forceinline__ u4v rpoll16_take() {
    static_assert(YOUNGER == 2 || YOUNGER == 1 || YOUNGER == 0, "");
    unsigned x, y, z, w;
#define WNV_TAKE16(R0, R1, R2, R3)                                                                                        \
    if constexpr (YOUNGER == 2) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 "\n\tv_mov_b32 %2, " R2 "\n\tv_mov_b32 %3, " R3 \
                                             : "=v"(x), "=v"(y), "=v"(z), "=v"(w) :: WNV_RSV_CLOBBERS);                   \
    else if constexpr (YOUNGER == 1) asm volatile("s_waitcnt vmcnt(1)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 "\n\tv_mov_b32 %2, " R2 "\n\tv_mov_b32 %3, " R3 \
                                             : "=v"(x), "=v"(y), "=v"(z), "=v"(w) :: WNV_RSV_CLOBBERS);                   \
    else asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 "\n\tv_mov_b32 %2, " R2 "\n\tv_mov_b32 %3, " R3 \
                      : "=v"(x), "=v"(y), "=v"(z), "=v"(w) :: WNV_RSV_CLOBBERS);
    if constexpr (SLOT == 0) { WNV_TAKE16("v244", "v245", "v246", "v247") }
    else if constexpr (SLOT == 1) { WNV_TAKE16("v248", "v249", "v250", "v251") }
    else { WNV_TAKE16("v252", "v253", "v254", "v255") }
#undef WNV_TAKE16
    return u4v{x, y, z, w};
}
template <int SLOT> __device__ __forceinline__ void rpoll8_issue(const u64* p) {
    if constexpr (SLOT == 0) asm volatile("global_load_dwordx2 v[244:245], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
    else if constexpr (SLOT == 1) asm volatile("global_load_dwordx2 v[248:249], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
    else asm volatile("global_load_dwordx2 v[252:253], %0, off sc1" :: "v"(p) : WNV_RSV_CLOBBERS);
}
template <int SLOT, int YOUNGER> __device__ __forceinline__ void rpoll8_take(unsigned& val, unsigned& tg) {
#define WNV_TAKE8(R0, R1)                                                                                                                  \
    if constexpr (YOUNGER == 2) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 : "=v"(val), "=v"(tg) :: WNV_RSV_CLOBBERS); \
    else if constexpr (YOUNGER == 1) asm volatile("s_waitcnt vmcnt(1)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 : "=v"(val), "=v"(tg) :: WNV_RSV_CLOBBERS); \
    else asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 : "=v"(val), "=v"(tg) :: WNV_RSV_CLOBBERS);
    if constexpr (SLOT == 0) { WNV_TAKE8("v244", "v245") }
    else if constexpr (SLOT == 1) { WNV_TAKE8("v248", "v249") }
    else { WNV_TAKE8("v252", "v253") }
#undef WNV_TAKE8
}

// ONE wave receives a 128-value vector, two granules per lane.  PRE: the caller issued slot 0, then slot 1, earlier; both are
// looked at first, oldest first, while a third poll (slot 2) is already on its way; after that one poll at a time.
template <bool PRE>
__device__ __forceinline__ bool rpoll_recv2(const u64* g2, unsigned tag, float& v0, float& v1, unsigned int* status, unsigned code, int lane) {
#define WNV_HIT16(x) if (__all(x.y == tag && x.w == tag)) { v0 = __uint_as_float(x.x); v1 = __uint_as_float(x.z); return true; }
    rpoll16_issue<2>(g2);
    if (PRE) {
        u4v x = rpoll16_take<0, 2>();
        WNV_HIT16(x)
        x = rpoll16_take<1, 1>();
        WNV_HIT16(x)
    }
    unsigned spins = 0;
    for (;;) {
        const u4v x = rpoll16_take<2, 0>();
        WNV_HIT16(x)
        rpoll16_issue<2>(g2);
        if ((++spins & 255u) == 0u) {
            // (giving up: the poll just issued is taken first -- nothing is ever in flight outside a helper, on any path)
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { (void)rpoll16_take<2, 0>(); return false; }
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); (void)rpoll16_take<2, 0>(); return false; }
        }
    }
#undef WNV_HIT16
}
// One wave waits until the granule of every ACTIVE lane carries `tag`.  NPRE = how many of the slots 0, 1, 2 the caller has
// already issued (in that order, active lanes only): they are looked at first, oldest first; after that one poll at a time.
template <int NPRE>
__device__ __forceinline__ bool rpoll_recv(const u64* g, bool active, unsigned tag, float& v, unsigned int* status, unsigned code, int lane) {
#define WNV_RPOLL8(SLOT, YOUNGER, REISSUE)                                    \
    {                                                                         \
        bool ok = true;                                                       \
        if (active) {                                                         \
            unsigned val, tg;                                                 \
            rpoll8_take<SLOT, YOUNGER>(val, tg);                              \
            v = __uint_as_float(val);                                         \
            ok = tg == tag;                                                   \
        }                                                                     \
        if (__all(ok)) return true;                                           \
        if (REISSUE && active) rpoll8_issue<SLOT>(g);                         \
    }
    if (NPRE == 1) { WNV_RPOLL8(0, 0, false) }
    if (NPRE == 2) { WNV_RPOLL8(0, 1, false) WNV_RPOLL8(1, 0, false) }
    if (NPRE == 3) { WNV_RPOLL8(0, 2, false) WNV_RPOLL8(1, 1, false) WNV_RPOLL8(2, 0, false) }
    unsigned spins = 0;
    if (active) rpoll8_issue<0>(g);
    for (;;) {
        WNV_RPOLL8(0, 0, true)
        if ((++spins & 255u) == 0u) {
            const bool abort = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (abort || spins > SPIN_LIMIT) {
                if (!abort && lane == 0) atomicCAS(status, 0u, code);
                if (active) { unsigned val, tg; rpoll8_take<0, 0>(val, tg); }     // (the poll just issued is taken before giving up)
                return false;
            }
        }
    }
#undef WNV_RPOLL8
}

// ONE wave receives three 2-granule groups at once (stage 1 behind a head that evaluates layer 0: four rows of N_1 h_0 and two
// values of u_0 per lane): the three loads travel together, one L2 round trip per attempt instead of three polls in a row.
__device__ __forceinline__ bool rpoll_recv3(const u64* ga, const u64* gb, const u64* gc, unsigned tag, float (&v)[6], unsigned int* status, unsigned code, int lane) {
    unsigned spins = 0;
    for (;;) {
        rpoll16_issue<0>(ga); rpoll16_issue<1>(gb); rpoll16_issue<2>(gc);
        const u4v a = rpoll16_take<0, 2>(), b = rpoll16_take<1, 1>(), c = rpoll16_take<2, 0>();
        if (__all(a.y == tag && a.w == tag && b.y == tag && b.w == tag && c.y == tag && c.w == tag)) {
            v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.z); v[2] = __uint_as_float(b.x); v[3] = __uint_as_float(b.z);
            v[4] = __uint_as_float(c.x); v[5] = __uint_as_float(c.z);
            return true;
        }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}

// ... and two 2-granule groups (the two partial residual vectors of a split ring)
__device__ __forceinline__ bool rpoll_recv2x2(const u64* ga, const u64* gb, unsigned tag, float& a0, float& a1, float& b0, float& b1, unsigned int* status, unsigned code, int lane) {
    unsigned spins = 0;
    for (;;) {
        rpoll16_issue<1>(ga); rpoll16_issue<2>(gb);
        const u4v a = rpoll16_take<1, 1>(), b = rpoll16_take<2, 0>();
        if (__all(a.y == tag && a.w == tag && b.y == tag && b.w == tag)) {
            a0 = __uint_as_float(a.x); a1 = __uint_as_float(a.z); b0 = __uint_as_float(b.x); b1 = __uint_as_float(b.z);
            return true;
        }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}

// Placement handshake: publish this workgroup's XCC id, read those of the (up to two) workgroups that read what this one
// sends; true when all share an XCD (and its L2).
__device__ __forceinline__ bool same_xcd_as(const RingParams& p, int reader_a, int n_a, int reader_b, int* flag) {
    // readers: blocks reader_a, reader_a + rstride, ... (n_a of them) and reader_b
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x = (x & 0xfu) + 1u;
        __hip_atomic_store(p.xcc + blockIdx.x, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int k = 0; k <= n_a; ++k) {
            const int rdk = k < n_a ? reader_a + k * p.rstride : reader_b;
            unsigned y = 0, spins = 0;
            while ((y = __hip_atomic_load(p.xcc + rdk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                if (++spins > SPIN_LIMIT) break;                   // unknown placement: take the placement-independent path
                __builtin_amdgcn_s_sleep(2);
            }
            same = same && y == x;
        }
        *flag = (p.allow_fast && same) ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

// the same for an explicit list of reader blocks (split rings)
__device__ __forceinline__ bool same_xcd_list(const RingParams& p, const int* readers, int n, int* flag) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x = (x & 0xfu) + 1u;
        __hip_atomic_store(p.xcc + blockIdx.x, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int k = 0; k < n; ++k) {
            unsigned y = 0, spins = 0;
            while ((y = __hip_atomic_load(p.xcc + readers[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                if (++spins > SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(2);
            }
            same = same && y == x;
        }
        *flag = (p.allow_fast && same) ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}
// split rings: ring r lives on XCDs 2r (the head in slot 0, then both halves of stages 1 .. sA) and 2r + 1 (stages sA + 1 .. S - 1);
// block b runs on XCD b % 8 (verified by the host's census), slot = b / 8
__device__ __forceinline__ int split_block(const RingParams& p, int ring, int stage, int half) {
    if (stage >= p.S) return 2 * ring;                                          // the head
    return stage <= p.sA ? 2 * ring + 8 * (1 + 2 * (stage - 1) + half) : 2 * ring + 1 + 8 * (2 * (stage - p.sA - 1) + half);
}

// debug timeline (builds with -DWNV_FINE_TRACE only; the product kernel carries no stamp code -- the scalar bookkeeping of a
// run-time "is this step traced" test sat on the chain of every stage): a role notes the device-wide 100 MHz wall clock in
// REGISTERS where something happens (WNV_TS: one s_memrealtime, no wait, no store) and lane 0 of a wave writes the slots it owns
// once per step, behind everything that is timed (WNV_TS_FLUSH).
constexpr int TRW = 16;            // stamp slots per (step, position)
#ifdef WNV_FINE_TRACE
// (every utterance of ring 0 is stamped: utterance b = j n_rings is the j-th of that ring)
__device__ __forceinline__ bool ts_traced(const RingParams& p, int b, int t) {
    return p.trace && b % p.n_rings == 0 && t >= p.trace_t0 && t < p.trace_t0 + p.trace_n;
}
__device__ __forceinline__ unsigned long long* ts_row(const RingParams& p, int b, int t, int pos) {
    return p.trace + (((size_t)(t - p.trace_t0) * p.upr + b / p.n_rings) * (p.S + 1) + pos) * TRW;
}
// slots [k of tsv] -> [k + shift] of the row, for the bits set in mask
__device__ __forceinline__ void ts_flush(const RingParams& p, int b, int t, int pos, const unsigned long long (&tsv)[TRW], unsigned mask, int shift = 0) {
    if ((threadIdx.x & 63) != 0 || !ts_traced(p, b, t)) return;
    unsigned long long* row = ts_row(p, b, t, pos);
#pragma unroll
    for (int k = 0; k < TRW; ++k)
        if ((mask >> k) & 1u) row[k + shift] = tsv[k];
}
#define WNV_TS_DECL unsigned long long tsv[TRW] = {0}
#ifndef WNV_TRACE_TAP_LAYER
#define WNV_TRACE_TAP_LAYER 6           // the tap workgroup whose passes are stamped (6: dilation 1 in the 24-layer models)
#endif
#ifndef WNV_TRACE_TAP_WAVE
#define WNV_TRACE_TAP_WAVE 0            // ... and the wave of it that stamps (round 6: the waves of a SIMD do not finish a round together)
#endif
#define WNV_TS(k) (tsv[k] = __builtin_amdgcn_s_memrealtime())
// the less important stamps: every live stamp is an SGPR pair in a kernel that has none to spare (with all of them the trace build
// spills vector registers in its hot loops and runs 25 % slower than the product): -DWNV_FINE_TRACE=2 turns them on
#if WNV_FINE_TRACE + 0 >= 2
#define WNV_TSX(k) WNV_TS(k)
#else
#define WNV_TSX(k) ((void)0)
#endif
#define WNV_TS_FLUSH(b, t, pos, mask, shift) ts_flush(p, b, t, pos, tsv, mask, shift)
#else
#define WNV_TS_DECL
#define WNV_TS(k) ((void)0)
#define WNV_TSX(k) ((void)0)
#define WNV_TS_FLUSH(b, t, pos, mask, shift) ((void)0)
#endif

// sum over the four adjacent lanes of a quad (the four K-quarters of one output channel): two DPP quad_perm adds
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_allreduce(float v) { return dpp_add<0x4E>(dpp_add<0xB1>(v)); }   // [1,0,3,2] then [2,3,0,1]
template <int CTRL> __device__ __forceinline__ float dpp_fold(float keep, float send) {
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true));
}

// tanh(a) * sigmoid(g) with the hardware exp2 / rcp (absolute error ~1e-7; the generic kernel keeps libm's
// tanhf/expf and is the cross-check):  tanh(a) = sign(a) (1 - e)/(1 + e), e = exp(-2|a|);  sigmoid(g) = 1/(1 + exp(-g))
__device__ __forceinline__ float fast_gate(float a, float g) {
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -2.8853900817779268f);
    const float f = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}

// ROUND 5 (WNV_PHASE2): the gate on PRE-SCALED pre-activations.  The host folds the exp2 scales into everything that sums into z
// (gate_scale(): tanh rows x -2 log2 e, sigmoid rows x -log2 e -- M, N, c, the tap matrix, layer 0's affine terms; the tap workgroups
// scale the bias row they add), so a' and g' ARE the exp2 arguments, and tanh(a) sigmoid(g) = (2 r1 - 1) r2 with r1 = 1 / (1 + 2^a'),
// r2 = 1 / (1 + 2^g'): exp -> add -> rcp -> fma -> mul, five dependent levels instead of eight (mul, exp, add, mul, rcp, mul, select;
// scripts/ubench_phase.hip variant 13: -35 ns per chain phase).  2^a' = inf gives r1 = 0 -> tanh = -1, 2^g' = inf gives 0: no NaN.
#ifndef WNV_PHASE2
#define WNV_PHASE2 1
#endif
#ifndef WNV_CAT_LOG
#define WNV_CAT_LOG 1          // the ring's categorical head picks in the log domain (run_head_cat, LOGPICK) -- in EVERY instantiation since round 6
#endif
#ifndef WNV_TAP_DEFER
#define WNV_TAP_DEFER 1        // (round 6) tap workgroups: the last publish of a pass is held back behind the next pass's [B] and barrier (run_tap)
#endif
#ifndef WNV_TAP_ZLDS
#define WNV_TAP_ZLDS 1         // (round 6) tap workgroups of the packed instantiations: the bias rows of a pass come through LDS (run_tap)
#endif
#ifndef WNV_SKIP_DIRECT
#define WNV_SKIP_DIRECT 1      // K = 512: every stage hands its own skip term to the head parts (head_sum_skip_terms)
#endif
constexpr float GATE_SCALE_TANH = -2.8853900817779268f, GATE_SCALE_SIGM = -1.4426950408889634f;
__device__ __forceinline__ float ring_gate(float a, float g) {
#if WNV_PHASE2
    const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
    const float r2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g));
    return fmaf(2.0f, r1, -1.0f) * r2;
#else
    return fast_gate(a, g);
#endif
}

__device__ __forceinline__ void lds_read32(const float* p, float (&x)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = reinterpret_cast<const float4*>(p)[c];
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
}
// 32-long dot product with the weights packed in pairs along K (v_pk_fma_f32, two accumulators)
__device__ __forceinline__ float dot32p(const f2 (&w)[16], const float (&x)[32]) {
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        a0 = __builtin_elementwise_fma(w[k], f2{x[2 * k], x[2 * k + 1]}, a0);
        a1 = __builtin_elementwise_fma(w[k + 1], f2{x[2 * k + 2], x[2 * k + 3]}, a1);
    }
    a0 += a1;
    return a0.x + a0.y;
}
// (Round 5, measured and NOT kept: the head's Gumbel-max butterfly as four hand-written v_max_f32_dpp + a v_cmp straight into a lane
//  mask -- 4 dependent instructions instead of the 13 the compiler emits for fmaxf(m, dpp_mov(m)) with its canonicalising max pairs --
//  made the HEADLINE 0.8 % slower on the same box (511.7 -> 507.6 kSamples/s, cfg4 400 -> 390; profiles/r05_phase2_ab.txt): every role
//  is inlined into one function and a change in the head moves the register allocation of the stage loop, as in round 4.)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void lds_read16(const float* p, float (&x)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(p)[c];
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
}
__device__ __forceinline__ float dot16p(const f2 (&w)[8], const float (&x)[16]) {
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        a0 = __builtin_elementwise_fma(w[k], f2{x[2 * k], x[2 * k + 1]}, a0);
        a1 = __builtin_elementwise_fma(w[k + 1], f2{x[2 * k + 2], x[2 * k + 3]}, a1);
    }
    a0 += a1;
    return a0.x + a0.y;
}
// the same with the weight row read from an LDS image ([chunk][512 threads] float4, conflict-free 16 B per lane)
__device__ __forceinline__ float dot16l(const float4* w, const float (&x)[16]) {
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = w[(size_t)c * RT];
        a0 = __builtin_elementwise_fma(f2{v.x, v.y}, f2{x[4 * c], x[4 * c + 1]}, a0);
        a1 = __builtin_elementwise_fma(f2{v.z, v.w}, f2{x[4 * c + 2], x[4 * c + 3]}, a1);
    }
    a0 += a1;
    return a0.x + a0.y;
}
// dot16l for a row image that STREAMS from the L2 every step (K = 512: the skip passes that fit neither LDS nor registers): non-temporal
// loads (experiment build -DWNV_STREAM_NT=1: eight rings stream 3 MB each per step through 4-MB L2s that also hold every mailbox)
#ifndef WNV_STREAM_NT
#define WNV_STREAM_NT 0
#endif
__device__ __forceinline__ float dot16s(const float4* w, const float (&x)[16]) {
    f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        typedef float f4n __attribute__((ext_vector_type(4)));
        float4 v;
        if (WNV_STREAM_NT) {
            const f4n t = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(w + (size_t)c * RT));
            v = make_float4(t.x, t.y, t.z, t.w);
        } else {
            v = w[(size_t)c * RT];
        }
        a0 = __builtin_elementwise_fma(f2{v.x, v.y}, f2{x[4 * c], x[4 * c + 1]}, a0);
        a1 = __builtin_elementwise_fma(f2{v.z, v.w}, f2{x[4 * c + 2], x[4 * c + 3]}, a1);
    }
    a0 += a1;
    return a0.x + a0.y;
}
// one matrix row's 16-float K-slice: 4 chunks of 16 B, image layout [chunk][512 threads][4]
__device__ __forceinline__ void load_image8(const float* img, int tid, f2 (&w)[8]) {
    const float4* src = reinterpret_cast<const float4*>(img);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = src[(size_t)c * RT + tid];
        w[2 * c] = f2{v.x, v.y}; w[2 * c + 1] = f2{v.z, v.w};
    }
}
__device__ __forceinline__ void load_image16(const float* img, int tid, f2 (&w)[16]) {
    const float4* src = reinterpret_cast<const float4*>(img);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = src[(size_t)c * RT + tid];
        w[2 * c] = f2{v.x, v.y}; w[2 * c + 1] = f2{v.z, v.w};
    }
}
__device__ __forceinline__ int qidx(int i) { return QS * (i >> 5) + (i & 31); }   // channel -> strided LDS slot (head: 4 K-quarters)
// stage kernel: eight K-slices of 16 floats, stride 20 -- the eight slices of one ds_read_b128 hit disjoint banks
constexpr int ES = 20;
__device__ __forceinline__ int eidx(int i) { return ES * (i >> 4) + (i & 15); }

struct StageLds {
    float* hx;       // chain input X[l][t]  (u of the layer before, or h_0 at stage 0), eight padded K-slices
    float* hb;       // h_{l-1}[t]: input of the off-chain mat-vec N_l . h_{l-1}
    float* us;       // gate output u_l[t]
    float* hh;       // h_l[t], collected for the tap workgroup
    float* pre;      // [256] pre_l[t] from the layer's tap workgroup
    float* zin;      // [128][2] {tanh, sigmoid} rows of N_l h_{l-1}[t] + pre_l[t]: handed from the N waves to the chain waves
    int* flags;
    float* bsk;      // [512] conv1x1_skip bias
    float4* wsk;     // [passes in LDS][2 rows][4 chunks][512 threads] image of conv1x1_skip (64 KiB per 128 skip channels; off the chain)
};
// conv1x1_skip with K = 128 NK output channels = NK passes over the 128-channel thread mapping.  Passes [0, lds_passes(NK))
// live in LDS; K = 512: the last stage keeps pass 2 in the registers that hold conv1x1_out elsewhere (its residual output is
// never used), whatever is left streams from the L2 behind the send (64 KiB per pass and step, off the chain except at the
// last stage).
__host__ __device__ constexpr int lds_passes(int NK) { return NK < 2 ? NK : 2; }

__device__ __forceinline__ StageLds carve_stage(float* smem) {
    StageLds s;
    s.hx = smem;
    s.hb = smem + 8 * ES;
    s.us = smem + 16 * ES;
    s.hh = smem + 24 * ES;
    s.pre = smem + 32 * ES;
    s.zin = s.pre + GC;
    s.flags = reinterpret_cast<int*>(s.zin + GC);
    s.bsk = reinterpret_cast<float*>(s.flags + 16);
    s.wsk = reinterpret_cast<float4*>(s.bsk + 512);
    return s;
}
__host__ __device__ constexpr size_t stage_lds_floats(int NK) { return (size_t)32 * ES + 2 * GC + 16 + 512 + (size_t)lds_passes(NK) * 8 * RT * 4; }

// pre_l[step] of utterance b: one record per STEP PARITY.  A tap workgroup of a layer with dilation >= 2 publishes pre_l[t + 1] before it
// waits for h_l[t] (nothing in it depends on that row), i.e. while a ring that runs late may still be reading pre_l[t]: the two live in
// different slots, and pre_l[t + 2] -- the next writer of pre_l[t]'s slot -- is only made after h_l[t + 1] arrived, which that ring files
// at the end of its step t + 1 (the data flow orders it, not the timing).
// ... and h_l[step] the other way: a ring does not wait for such a layer's taps any more, so it may file h_l[t] before the tap workgroup has
// read h_l[t-1]; h_l[t+1] -- the next writer of h_l[t-1]'s slot -- needs pre_l[t+1], which that workgroup makes only after it has.
__device__ __forceinline__ size_t h_rec(const RingParams& p, int b, int l, int step) {
    return (((size_t)b * p.L + l) * 2 + (size_t)(step & 1)) * RC;                    // in granules
}
__device__ __forceinline__ size_t pre_rec(const RingParams& p, int b, int l, int step) {
    return (((size_t)b * p.L + l) * 2 + (size_t)(step & 1)) * GC;                    // in granules
}

// ---- tap workgroup (one per layer, shared by all rings) -----------------------------------------------------------------
// Everything of a layer that is known a step ahead -- the dilated conv's older taps and the local-conditioning 1x1,
//   pre_l[t+1] = b_l (+ W_g g) + c_l + sum_{k<kw-1} W_l[:, :, k] h_l[t+1 - (kw-1-k) d] + W_c,l c[t+1]        (conv.py:33-45)
// -- is computed here for EVERY utterance, with the [kw-1 taps + cin][256] matrix resident on this CU (VGPRs first, then
// LDS; only what fits neither streams from L2).  The stages forward h_l[t] (one write-through granule per value) and get
// pre_l[t+1] back the same way; both trips have a whole step of slack.  The workgroup also owns the history rings.
constexpr int KR_MAX = 32;         // K rows per lane slice held in VGPRs (32 float4 = 128 registers)
constexpr int KL_MAX = 16;         // further K rows per wave held in LDS
#ifndef WNV_KR_PACKED_SPEC
#define WNV_KR_PACKED_SPEC 32      // (28: round 5's form -- four rows in LDS to make room for the look's registers: 20-40 % slower, profiles/r06_tap_zlds_ab.txt)
#endif
constexpr int KR_PACKED_SPEC = WNV_KR_PACKED_SPEC; // ... in VGPRs in the packed-slot instantiations that keep the speculative look (run_tap; an experiment build)
#ifndef WNV_TAP_SPEC1
#define WNV_TAP_SPEC1 1            // the speculative look in the throughput instantiation's tap role (A/B switch, round 6)
#endif
#ifndef WNV_PACKED_SPEC
#define WNV_PACKED_SPEC 1          // the speculative look in the PACKED tap instantiations too (round 6: with the bias rows handed over through LDS -- WNV_TAP_ZLDS --
                                   // the look's four registers fit next to all 32 register rows, no spill: every packed job +2.5-3.5 %; round 5 had to move four
                                   // rows to LDS for it and still spilled)
#endif
constexpr int TB = 8;              // utterances per pass (one polling wave each; their latencies overlap)
struct TapLds {
    float* xin;      // [2][TB][8 * kper] mat-vec inputs (two buffers: the next pass's gather lands in the other one): tap rows then conditioning row, zero padded
    int* flags;      // [32]: 0 = give-up flag; 8 + buf * TB + u = packed slots: bias row (seg_gid) of utterance u of the pass whose inputs are in buffer buf
    float4* wl;      // [8 waves][klds_rows][64 lanes] LDS-resident rows
    float2* dv;      // [512] the held-back publish of a pass's last round (round 6: parked here, not in registers -- the tap role has none to spare)
    float* zl;       // [2][TB][256] (round 6, WNV_TAP_ZLDS) the effective conv bias row of every utterance of a pass, fetched with the pass's inputs
};
__device__ __forceinline__ TapLds carve_tap(float* smem, const RingParams& p) {
    TapLds s;
    s.xin = smem;
    s.flags = reinterpret_cast<int*>(smem + (size_t)2 * TB * RW * p.kper);
    s.dv = reinterpret_cast<float2*>(s.flags + 32);
    s.zl = reinterpret_cast<float*>(s.dv + RT);
    s.wl = reinterpret_cast<float4*>(s.zl + 2 * TB * GC);
    return s;
}
__host__ __device__ inline size_t tap_lds_floats(int kper, int klds_rows) {
    return (size_t)2 * TB * RW * kper + 32 + 2 * RT + (size_t)2 * TB * GC + (size_t)RW * klds_rows * 64 * 4;
}

// EXPERIMENT BUILDS ONLY (-DWNV_EXP_NOPRE=1|2; results are WRONG on purpose, timing only): 1 = stages and head do not wait for the tap
// workgroups' records (what would a zero-latency tap path buy?), 2 = the tap workgroups also exit at once (... and what do they cost
// the chain's hops by sharing the fabric?).  profiles/r04_tap_bound_experiment.txt.
#ifndef WNV_EXP_NOPRE
#define WNV_EXP_NOPRE 0
#endif
// (SPEC: a speculative look at the next pass's h record, issued with the gather -- four registers live across the mat-vec: it saves one
//  poll round trip per pass (+3 % at 40-64 utterances, A/B switch WNV_TAP_SPEC1).  On everywhere but the 256-skip-channel kernels since
//  round 6; rounds 4-5 had it off in the packed instantiations, where the slot masks and the bias loads had taken the last registers.)
// (PACKED: the launch runs packed slots -- RingParams::seg_start; a compile-time switch like SPEC)
// (DEFER: the throughput and packed instantiations hold a pass's last publish back -- see the pass loop; a compile-time switch: the
//  single-utterance-per-ring kernels never run two passes per workgroup and step at their batch sizes that matter, and the code alone moved
//  the headline kernel's tap role by 1.4 %)
template <bool SPEC, bool PACKED, bool DEFER>
__device__ __attribute__((always_inline)) void run_tap(const RingParams& p, int l, int part, float* smem) {
    if (WNV_EXP_NOPRE >= 2) return;
    const TapLds s = carve_tap(smem, p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = p.lay_dil[l];
    const int rows = (p.kw - 1) * d;
    const int hoff = (p.kw - 1) * RC;
    const int kx = RW * p.kper;                                    // padded K
    // MAT-VEC MAPPING (round 4): the eight WAVES split the 256 outputs (wave w: outputs [32 w, 32 w + 32)), the eight lanes ks = lane & 7 of
    // a lane group og = lane >> 3 split K (slice ks: rows [ks kper, ks kper + kper)) for the four outputs 32 w + 4 og .. + 3: the K slices meet
    // in three DPP steps inside the wave and every lane publishes from its registers.  (Until round 4 the WAVES split K: partial sums of four
    // utterances through 32 KB of LDS, a barrier, a reduce that waited for its bias loads, another barrier -- 3.1-3.8 us per round of four
    // utterances for 1.1 us of FMAs, profiles/r04_tap_pass_timeline.txt; the passes of the tap workgroups are what bounds the throughput
    // beyond 32 utterances per GPU.)
    const int ks = lane & 7, og = lane >> 3;
    const int ob = 32 * wave + 4 * og;                             // this lane's four outputs
    const int k0 = ks * p.kper;                                    // this lane's K rows: [k0, k0 + kper) of the padded matrix
    const float* Wt = p.wpre + (size_t)l * p.kpre * GC;           // K-major [kpre][256]
    // REDUCE-SCATTER WITHOUT SELECTS (as group_matvec8): after the FMAs a lane holds 16 partial sums -- 4 utterances x 4 outputs -- and
    // lane ks is to end up with outputs ob + 2 (ks & 1), + 1 of utterance ks >> 1.  Which utterance an accumulator GROUP g means and which
    // output pair comes first depend on the lane, so that every step adds the partner's "sent" registers to the own "kept" ones:
    //   step 1, row_half_mirror (ks <-> 7 - ks; the partner has the other parity, so pairs cross): groups 2, 3 are sent, 0, 1 kept;  step 2, ks <-> ks ^ 2: group 1 sent, 0 kept;
    //   step 3, ks <-> ks ^ 1: pair 1 sent, pair 0 kept.                                             14 DPP adds instead of 48.
    // Group g of lane ks = utterance ug[g]:  ug[0] = ks >> 1, ug[1] = (ks >> 1) ^ 1, ug[2], ug[3] = what lane 7 - ks keeps in groups 0, 1;
    // odd lanes hold their weights as (outputs 2, 3, 0, 1).
    const bool odd = (ks & 1) != 0;
    auto wload = [&](int k) {
        if (k >= p.kpre) return make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 w = *reinterpret_cast<const float4*>(Wt + (size_t)k * GC + ob);
        return odd ? make_float4(w.z, w.w, w.x, w.y) : w;
    };
    // (packed slots WITH the speculative look at the next pass's record: four rows less in registers -- they live in LDS --, which is what
    //  the look's four registers and the packed masks need: with 32 rows that combination spilled 6-8 registers; the host sets kreg_rows)
    constexpr int KR = (PACKED && SPEC) ? KR_PACKED_SPEC : KR_MAX;
    float4 wreg[KR];                                                // resident rows (registers), then LDS rows, then whatever streams
#pragma unroll
    for (int r = 0; r < KR; ++r) wreg[r] = r < p.kreg_rows ? wload(k0 + r) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < p.klds_rows; ++r) s.wl[((size_t)wave * p.klds_rows + r) * 64 + lane] = wload(k0 + p.kreg_rows + r);
    const int uq = ks >> 1, uqm = 3 - uq;                            // (7 - ks) >> 1 = 3 - (ks >> 1)
    const int ug[4] = {uq, uq ^ 1, uqm, uqm ^ 1};
    const int xo0 = ug[0] * kx + k0, xo1 = ug[1] * kx + k0, xo2 = ug[2] * kx + k0, xo3 = ug[3] * kx + k0;   // input of group g: xin[...][ug[g]][k0 ..]
    // what a lane publishes: after the all-reduce over ks every lane of a group holds all 16 sums (4 utterances x 4 outputs); lane ks sends
    // outputs ob + 2 (ks & 1), + 1 of utterance ks >> 1 of the round -- with their addends: c_l (a constant of the lane) and the effective
    // conv bias (per utterance when the model has a speaker embedding).  The bias rows are the MODEL's gate rows (tanh rows [0, G/2),
    // sigmoid rows [G/2, G)); this kernel's 256 outputs are tanh channels 0..127 then sigmoid channels 0..127, zero beyond G/2
    const int pu = uq, po = ob + 2 * (ks & 1);                     // utterance of the round, first of the two outputs
    const int zhalf = po >> 7, zch = po & 127;
    const float2 cvl = *reinterpret_cast<const float2*>(p.cvec + (size_t)l * GC + po);
    const float* zbase = p.zbias + (size_t)l * p.zb_ld + (size_t)zhalf * p.gh + zch;
    const bool z0 = zch < p.gh, z1 = zch + 1 < p.gh;
    const float zsc = WNV_PHASE2 ? (zhalf ? GATE_SCALE_SIGM : GATE_SCALE_TANH) : 1.0f;
    for (int i = tid; i < 2 * TB * kx; i += RT) s.xin[i] = 0.f;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();
    const int kres = p.kreg_rows + p.klds_rows;                    // rows of this wave that never touch memory again

    // A layer with dilation >= 2 needs NOTHING of this step's h for pre[t + 1] (its taps are h[t+1-d], h[t+1-2d], ...): its pass does not
    // wait for h[t] -- the row is waited for and filed by the NEXT pass of the utterance (tf = t - 1), a whole step later --, so the rings
    // never wait for this layer's taps (pre_rec: why the records come in two slots).  With d = 2 the youngest tap of step t + 1 is h[t-1],
    // the row that pass files: taken from the record, as a dilation-1 layer takes h[t].
    // slot of a step in the history ring: rows = (kw - 1) d is a power of two for every kernel size 3 model (d = 2^i): a mask, not the
    // ~40-instruction integer division the general case costs every lane of every gather (uniform: one compare)
    const int rmask = rows > 0 && (rows & (rows - 1)) == 0 ? rows - 1 : -1;
    auto ring_slot = [&](int step) -> int { return rmask >= 0 ? (step & rmask) : step % rows; };
    const bool early = d >= 2 && rows > 0;
    const int ntap4 = hoff / 4, ncin4 = (p.cin & 3) == 0 ? p.cin / 4 : 0;      // float4s of a mat-vec input: tap rows, conditioning row
    constexpr int GQ = 2;                                          // ... per lane ((kw - 1) 128 + cin <= 512 floats: why_not)
    const int pstride = p.tap_parts * p.tb;                        // this workgroup's passes of a step: utterances [b0, b0 + tb), b0 += pstride
    const int bfirst = part * p.tb;
    if (bfirst >= p.B) return;
    auto fresh_tap = [&](int tf_) { return ((d == 1 || early) && d <= 2 && tf_ >= 0 && rows > 0) ? p.kw - 2 : -1; };   // the tap that is h[tf] itself

    // ---- SOFTWARE PIPELINE (round 4).  A pass = [B] finish the inputs (wave w = utterance b0 + w: gathered rows -> LDS, h_l[tf] from its
    //      record -> history ring and LDS), barrier, [C] ISSUE the next pass's gather, [D] mat-vec + reduce + publish.  The gather -- the
    //      kw-1 older taps (one contiguous 512-byte history row each; zeros before t = 0: the rings start zeroed) and c[t + 1], 16-byte
    //      loads, two per lane -- is ~1.5 us of global-load latency; issued at [C] it lands under the mat-vec of the pass before (until
    //      round 4 it was paid in front of every pass: beyond 32 utterances per GPU the passes of the tap workgroups bound the
    //      throughput, profiles/r04_tap_bound_experiment.txt).  The next pass's h record is looked at speculatively in the same
    //      breath (two granules per lane): when the utterance's stage has filed it already -- the rule in that regime -- [B] takes it
    //      without a poll round trip.  What the next pass gathers was filed by THIS wave at the latest in [B] of this pass (a
    //      dilation-1 layer: h[t], one step back; in order in this CU's L1); the row the next pass files itself is never gathered.
    // The gathered rows go global -> LDS by DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land at base + 16 lane; no staging
    // registers -- next to 128 weight registers there are none to spare: the register-staged form spilled 27) into the OTHER input
    // buffer; the issuing wave waits for its own DMAs (vmcnt) in [B] of the pass that uses them, in front of the barrier.
    u64 hx0 = 0, hx1 = 0;
    // packed slots: seg_start of (utterance b0_ + wave, step t_ + 1) -- a scalar load (no vector register held across the DMA issue),
    // asked for ONE PASS AHEAD of the gather that needs it (round 5: read at the top of gather_issue, every pass began its mat-vec behind
    // a scalar-load round trip -- the packed instantiations ran 12-20 % below the padded ones at the same number of rows)
    auto seg_at = [&](int t_, int b0_) -> int {
        if (!PACKED || wave >= min(p.tb, p.B - b0_) || t_ + 1 >= p.T) return INT_MIN;
        return uniform_ld(p.seg_start + (size_t)__builtin_amdgcn_readfirstlane(b0_ + wave) * p.T + t_ + 1);
    };
    auto gather_issue = [&](int t_, int b, float* xu, int buf, int st_) {
        const int tp_ = t_ + 1, tf_ = early ? t_ - 1 : t_, kf = fresh_tap(tf_);
        const float* hb = p.hist + (size_t)b * p.hist_floats + p.lay_histoff[l];
        const float* cb = p.c_up + ((size_t)b * p.T + tp_) * p.cin;
        // packed slots: the utterance that occupies the slot at step tp_ began at step st_; what lies before reads as zeros (conv.py:34-36)
        // (the lane-constant pieces of the addresses are formed HERE, from a copy of the lane id the compiler cannot see through: kept live
        //  across the pass they were the registers that spilled once the deferred publish joined the loop -- a reload per pass against
        //  three VALU instructions)
        int lane_l = lane;
        if constexpr (DEFER) asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int i = 64 * q + lane_l;
            const float* src = nullptr;
            if (i < ntap4) {
                const int k = i >> 5, r4 = i & 31;                   // RC / 4 = 32 float4s per row
                // (a row from before the utterance's start comes from a row of zeros: a select on the address, no branch -- the tap role has
                //  no register to spare)
                if (k != kf) src = tp_ - (p.kw - 1 - k) * d >= st_ ? hb + (size_t)ring_slot(tp_ + k * d) * RC + 4 * r4 : p.zero_row + 4 * r4;
            } else if (i < ntap4 + ncin4) {
                src = cb + 4 * (i - ntap4);
            }
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)(xu + 256 * q));
            if (src) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds) : "memory");
        }
        hx0 = hx1 = 0;
        if (SPEC && tf_ >= 0) {                                     // speculative look at the record (L1-bypassing, not waited for here)
            const u64* rec = p.fmail + h_rec(p, b, l, tf_) + 2 * lane;
            hx0 = ld_granule(rec); hx1 = ld_granule(rec + 1);
        }
    };

    int t = -1, b0 = bfirst, cur = 0;                              // pass (t, b0) consumes h_l[tf], produces pre_l[t + 1]; its inputs: buffer cur
    int st_cur = seg_at(t, b0);                                      // ... of the pass about to run, of the next one
    if (wave < min(p.tb, p.B - b0)) gather_issue(t, b0 + wave, s.xin + (size_t)wave * kx, 0, st_cur);
    int st_next = INT_MIN;
    {
        int tn0 = t, bn0 = b0 + pstride;
        if (bn0 >= p.B) { bn0 = bfirst; tn0 = t + 1; }
        st_next = seg_at(tn0, bn0);
    }
#ifdef WNV_FINE_TRACE
#define TAP_STAMP(k) do { const int pk_ = (b0 - bfirst) / pstride; \
                          if (p.trace_tap && l == WNV_TRACE_TAP_LAYER && part == 0 && pk_ < 3 && (k) < 5 && tid == 64 * WNV_TRACE_TAP_WAVE && t >= p.trace_t0 && t < p.trace_t0 + p.trace_n) \
                              p.trace_tap[(size_t)(t - p.trace_t0) * TRW + 5 * pk_ + (k)] = wall_clock64(); } while (0)
#else
#define TAP_STAMP(k) ((void)0)
#endif
    // ---- [B] wave w finishes the mat-vec input of utterance b0_ + w of pass (t_, b0_) in buffer cur_: the h record it waits for (looked at
    //      speculatively with the gather: hx0, hx1) -> history ring and LDS; then its own DMAs (issued with the gather) have landed -------------
    auto do_B = [&](int t_, int b0_, int cur_, int st_) {
        const int tp_ = t_ + 1, nb_ = min(p.tb, p.B - b0_);
        const int tf_ = early ? t_ - 1 : t_;                        // the step whose h this pass waits for and files
        const int kfresh = fresh_tap(tf_);
        if (wave < nb_) {
            const int b = b0_ + wave;
            float* xu = s.xin + ((size_t)cur_ * TB + wave) * kx;
            // (round 6, WNV_TAP_ZLDS) the utterance's effective conv bias row -- b_l + W_g g: per utterance when the model has a speaker
            // embedding -- is fetched HERE, with the inputs, and handed to the rounds through LDS.  It used to be two global loads per lane at
            // the top of every round: in round 0 they were issued right behind the next pass's gather DMAs, loads return in order, and the
            // round's publish waited for the whole gather (history rows from the L2, the conditioning row from HBM).  PACKED instantiations
            // only: there [B] has a record round trip for the fetch to hide under (no speculative look) and a packed job gains 3.6-4.4 %
            // (egs/mol 100 utterances 2 292 -> 2 375, cfg4 128 utterances 1 166 -> 1 217 kSamples/s, same box); in the throughput
            // instantiation, whose speculative look usually hits, the fetch sits in the open in front of the pass's barrier: egs/mol
            // +0.4 %, mu-law -1.9 %, cfg4 -4 % (profiles/r06_tap_zlds_ab.txt) -- so not there.
            float4 zrow4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PACKED && WNV_TAP_ZLDS != 0) {
                size_t zr = (size_t)b;
                if (PACKED && p.seg_gid) zr = (size_t)uniform_ld(p.seg_gid + (size_t)__builtin_amdgcn_readfirstlane(b) * p.T + tp_);
                if (4 * lane < p.zb_ld) zrow4 = *reinterpret_cast<const float4*>(p.zbias + (size_t)l * p.zb_ld + zr * p.zbias_bstride + 4 * lane);
            }
            if (ncin4 == 0) {                                        // cin not a multiple of 4: scalar conditioning row
                const float* cb = p.c_up + ((size_t)b * p.T + tp_) * p.cin;
                for (int e = lane; e < p.cin; e += 64) xu[hoff + e] = cb[e];
            }
            if (tf_ >= 0) {
                const unsigned htag = p.tag_base + (unsigned)tf_ + 1u;
                float hv[2] = {__uint_as_float((unsigned)hx0), __uint_as_float((unsigned)hx1)};         // channels 2 lane, 2 lane + 1
#ifdef WNV_FINE_TRACE
                if (lane == 0) s.flags[24 + wave] = 0;                 // (trace: how this wave got its record -- 0 look, 1 direct look, 2 patient receive)
#endif
                if (!__all((unsigned)(hx0 >> 32) == htag && (unsigned)(hx1 >> 32) == htag)) {
                    // (round 6) ONE direct look at the whole record first: while the stages pace the passes the record lands during the pass
                    // before -- behind the speculative look, ahead of this one -- and the patient receive (first granule at a relaxed cadence,
                    // then the record: two round trips and a sleep at best) made every pass wait ~1.4 us at its barrier for the one wave
                    // whose speculative look had missed (profiles/r06_tap_pass_timeline.txt)
                    const u64* rec = p.fmail + h_rec(p, b, l, tf_);
                    const u4v x = ld16_sc1(rec + 2 * lane);
                    hv[0] = __uint_as_float(x.x); hv[1] = __uint_as_float(x.z);
#ifdef WNV_FINE_TRACE
                    if (lane == 0) s.flags[24 + wave] = __all(x.y == htag && x.w == htag) ? 1 : 2;
#endif
                    if (!__all(x.y == htag && x.w == htag)) {
                        if (!rec_recv<1>(rec, htag, hv, p.status, 0x600u + (unsigned)l, lane)) s.flags[0] = 1;
                    }
                }
                if (rows > 0) {
                    float* hist = p.hist + (size_t)b * p.hist_floats + p.lay_histoff[l];
                    *reinterpret_cast<float2*>(hist + (size_t)ring_slot(tf_) * RC + 2 * lane) = make_float2(hv[0], hv[1]);
                    if (kfresh >= 0) {
                        // (packed slots: the row is the previous utterance's when the one at step tp began later than tf)
                        const bool mine = !PACKED || tf_ >= st_;
                        *reinterpret_cast<float2*>(xu + kfresh * RC + 2 * lane) = mine ? make_float2(hv[0], hv[1]) : make_float2(0.f, 0.f);
                    }
                }
            }
            if constexpr (PACKED && WNV_TAP_ZLDS != 0) *reinterpret_cast<float4*>(s.zl + ((size_t)cur_ * TB + wave) * GC + 4 * lane) = zrow4;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs into buffer cur_ (issued with the gather) have landed
        }
    };
    // (round 6) THE LAST PUBLISH OF A PASS IS HELD BACK until the next pass's [B] and barrier are through (dv0, dv1 -> "deferred publish"
    // below).  A publish is a write-through store (the stage may sit on any XCD), its acknowledgement takes 1.5-2 us, and every wait of
    // this wave for a LOAD -- the h record's poll, the DMA wait -- waits for it too (vmcnt counts stores): with the last publish right in
    // front of [B] that drain was paid in the open once per pass -- the "wait for h" of profiles/r04_throughput_bound_final_stages.txt,
    // 1.6-2.6 us of a 5.9-us pass whatever the stages did (at 40+ utterances per GPU the step was 3 or 4 such passes: 17.6 / 23.5 us).
    // Now every publish is followed by a round of FMAs before this wave waits for anything.
    // ONLY where this workgroup runs at least two passes per step: with a single one the next pass is the same utterances' next step, whose
    // h (a dilation-1 layer's pass waits for it) cannot exist before the publish it would be holding back -- the ring would stop.
    // With two or more, pass N + 1 waits for h of ANOTHER group, which depends on publishes of pass N - 1 and earlier, all released.
    const bool defer = DEFER && WNV_TAP_DEFER != 0 && bfirst + pstride < p.B;
    int dbase = -1, du0 = 0, dnb = 0, dtp = 0;                       // the held publish: its pass's first utterance (-1: nothing held), round, utterances, step
    for (;;) {
        const int tp = t + 1;
        const int nb = min(p.tb, p.B - b0);
        TAP_STAMP(0);
        do_B(t, b0, cur, st_cur);
        TAP_STAMP(1);                                                // (this wave's h record filed, its DMAs landed)
        __syncthreads();                                             // the inputs of pass (t, b0) are complete in buffer cur
        if (s.flags[0]) return;
        TAP_STAMP(2);
#ifdef WNV_FINE_TRACE
        {   // slot 15 of the row: two bits per wave and pass -- how each wave got its record
            const int pk_ = (b0 - bfirst) / pstride;
            if (p.trace_tap && l == WNV_TRACE_TAP_LAYER && part == 0 && pk_ < 3 && tid == 0 && t >= p.trace_t0 && t < p.trace_t0 + p.trace_n) {
                unsigned long long m = 0;
                for (int w = 0; w < 8; ++w) m |= (unsigned long long)(s.flags[24 + w] & 3) << (2 * w);
                unsigned long long* q = p.trace_tap + (size_t)(t - p.trace_t0) * TRW + 15;
                *q = (pk_ == 0 ? 0ull : *q) | (m << (16 * pk_));
            }
        }
#endif
        if (dbase >= 0 && du0 + pu < dnb) {                             // the deferred publish of the pass before (see above the loop)
            const float2 dvv = s.dv[tid];
            st_granule2(p.pmail + pre_rec(p, dbase + du0 + pu, l, dtp) + po, p.tag_base + (unsigned)dtp + 1u, dvv.x, dvv.y, false);
        }
        dbase = -1;
        // ---- [C] the next pass of this workgroup: its gather is issued now and lands under the mat-vec below -------------------------
        int tn = t, bn = b0 + pstride;
        if (bn >= p.B) { bn = bfirst; tn = t + 1; }
        const bool more = tn + 1 < p.T;
        if (more && wave < min(p.tb, p.B - bn)) gather_issue(tn, bn + wave, s.xin + ((size_t)(cur ^ 1) * TB + wave) * kx, cur ^ 1, st_next);
        int st_nn = INT_MIN;                                         // the pass after the next: asked for now, used a pass from now
        if (PACKED && more) {
            int tnn = tn, bnn = bn + pstride;
            if (bnn >= p.B) { bnn = bfirst; tnn = tn + 1; }
            st_nn = seg_at(tnn, bnn);
        }
        // ---- [D] mat-vec for all utterances of the pass, four at a time (accumulators + weights must fit the register file):
        //      VGPR rows, then LDS rows, then whatever streams; no barrier inside (the inputs are read-only here, the next
        //      pass's land in the other buffer, and its [B] barrier is behind every wave's last read of this one) -----------
#pragma unroll 1
        for (int u0 = 0; u0 < nb; u0 += 4) {
            // packed FMAs (v_pk_fma_f32: two outputs per instruction, the input broadcast into both halves): this loop is what a
            // pass costs -- 336 rows x 256 outputs x 8 utterances = 688 k MACs per workgroup
            const bool pub = u0 + pu < nb;                                   // this lane has something to publish in this round
            const int rb = b0 + u0 + pu;
            float zb0 = 0.f, zb1 = 0.f;                                      // the utterance's effective conv bias: requested ahead of the FMAs
            if constexpr (PACKED && WNV_TAP_ZLDS != 0) {
                if (pub) {
                    const float* zq = s.zl + ((size_t)cur * TB + u0 + pu) * GC + zhalf * p.gh + zch;
                    if (z0) zb0 = zq[0];
                    if (z1) zb1 = zq[1];
                }
            } else
            if (pub) {
                // (throughput instantiation: these two global loads per lane and round stay.  Round 6 measured both alternatives on one box --
                //  the row fetched with the pass's inputs and handed over through LDS: egs/mol +0.4 %, mu-law -1.9 %, cfg4 -4 %; ONE copy in LDS
                //  for the whole launch where no global conditioning makes the row a constant: -3.2 % at 40-64 utterances.  The loads' wait at
                //  the end of a round is where this wave's earlier write-through publish gets drained, under the other waves' FMAs; without
                //  it the drain moves into the open in front of the next pass's barrier: profiles/r06_tap_zlds_ab.txt)
                // (packed slots: the bias row of the utterance that occupies the slot at step tp -- its speaker.  Read here, ahead of the
                //  FMAs; parked in LDS a pass ahead it cost the packed instantiations 2-4 spilled registers and 3 % -- round 5, measured)
                const float* zrow = zbase + (size_t)((PACKED && p.seg_gid) ? p.seg_gid[(size_t)rb * p.T + tp] : rb) * p.zbias_bstride;
                if (z0) zb0 = zrow[0];
                if (z1) zb1 = zrow[1];
            }
            f2 acc[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = f2{0.f, 0.f};
            const float* xr = s.xin + ((size_t)cur * TB + u0) * kx;
            const float* xg[4] = {xr + xo0, xr + xo1, xr + xo2, xr + xo3};
            // (loop order: a row's weights against two utterances -- four INDEPENDENT accumulators in a row; with one utterance
            //  innermost the compiler alternated two and every v_pk_fma_f32 waited for the one before the last: 6.8 clocks per
            //  instruction instead of 4.7, profiles/r04_tap_pass_timeline.txt; all four at once need 16 input registers: spills)
#pragma unroll
            for (int r4 = 0; r4 < KR / 4; ++r4) {
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {                      // two utterances at a time: four independent accumulators in a row
                    const float4 xa = *reinterpret_cast<const float4*>(xg[gp] + 4 * r4), xc = *reinterpret_cast<const float4*>(xg[gp + 1] + 4 * r4);
                    const float xs[2][4] = {{xa.x, xa.y, xa.z, xa.w}, {xc.x, xc.y, xc.z, xc.w}};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 w = wreg[4 * r4 + e];
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const f2 xx = f2{xs[g][e], xs[g][e]};
                            acc[gp + g][0] = __builtin_elementwise_fma(f2{w.x, w.y}, xx, acc[gp + g][0]);
                            acc[gp + g][1] = __builtin_elementwise_fma(f2{w.z, w.w}, xx, acc[gp + g][1]);
                        }
                    }
                }
            }
            for (int r = 0; r < p.klds_rows; r += 4) {                     // klds_rows is a multiple of 4: one 16-byte x read per utterance
                float4 w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = s.wl[((size_t)wave * p.klds_rows + r + e) * 64 + lane];
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    const float4 xa = *reinterpret_cast<const float4*>(xg[gp] + p.kreg_rows + r), xc = *reinterpret_cast<const float4*>(xg[gp + 1] + p.kreg_rows + r);
                    const float xs[2][4] = {{xa.x, xa.y, xa.z, xa.w}, {xc.x, xc.y, xc.z, xc.w}};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const f2 xx = f2{xs[g][e], xs[g][e]};
                            acc[gp + g][0] = __builtin_elementwise_fma(f2{w[e].x, w[e].y}, xx, acc[gp + g][0]);
                            acc[gp + g][1] = __builtin_elementwise_fma(f2{w[e].z, w[e].w}, xx, acc[gp + g][1]);
                        }
                }
            }
            for (int k = k0 + kres; k < k0 + p.kper && k < p.kpre; ++k) {
                const float4 w = wload(k);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float xs = xg[g][k - k0];
                    const f2 xx = f2{xs, xs};
                    acc[g][0] = __builtin_elementwise_fma(f2{w.x, w.y}, xx, acc[g][0]);
                    acc[g][1] = __builtin_elementwise_fma(f2{w.z, w.w}, xx, acc[g][1]);
                }
            }
            // the K slices meet (see REDUCE-SCATTER above)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[g][h] = f2{dpp_fold<0x141>(acc[g][h].x, acc[g + 2][1 - h].x), dpp_fold<0x141>(acc[g][h].y, acc[g + 2][1 - h].y)};   // (the mirror partner has the other parity: its OTHER pair holds these outputs)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[0][h] = f2{dpp_fold<0x4E>(acc[0][h].x, acc[1][h].x), dpp_fold<0x4E>(acc[0][h].y, acc[1][h].y)};
            // (WNV_PHASE2: everything that sums into z carries the gate's exp2 scale -- the matrix and c_l from the host, the bias row here)
            const float v0 = dpp_fold<0xB1>(acc[0][0].x, acc[0][1].x) + fmaf(zb0, zsc, cvl.x);
            const float v1 = dpp_fold<0xB1>(acc[0][0].y, acc[0][1].y) + fmaf(zb1, zsc, cvl.y);
            // pre_l[tp] of utterance rb leaves as tagged granules (write-through: the stage may sit on any XCD): no drain
            if (defer && more && u0 + 4 >= nb) {                          // the last round of the pass: held back (see above the loop)
                s.dv[tid] = make_float2(v0, v1);                         // (read back by this very thread)
                dbase = b0; du0 = u0; dnb = nb; dtp = tp;
            } else
            if (pub) st_granule2(p.pmail + pre_rec(p, rb, l, tp) + po, p.tag_base + (unsigned)tp + 1u, v0, v1, false);
            TAP_STAMP(min(3 + u0 / 4, 4));
        }
#undef TAP_STAMP
        if (!more) break;
        t = tn; b0 = bn; cur ^= 1;
        st_cur = st_next; st_next = st_nn;
    }
}

#ifndef WNV_MF_GATHER_AT
#define WNV_MF_GATHER_AT 1         // the K group of a unit's multiplication behind which the next unit's gathers and looks are issued (-1: in front of it)
#endif
constexpr int TAP_NJR = 16;        // (round 6, matrix-pipe tap role, run_tap_mf) K groups of 16 rows whose weights a lane holds in registers (2 tiles x 4 each: 128) ...
constexpr int TAP_NJL = 6;         // ... and at most this many more in LDS (two float4 per lane and group: TapLds::wl): K = (kw - 1) 128 + cin <= 352
constexpr int TAP_XPAD = 4;        // the pad of a mat-vec input row in LDS: the sixteen utterances of a multiplication then sit on sixteen different bank quads
constexpr int TAP_XBUF = 4;        // input buffers (two units of two passes)
__device__ __forceinline__ TapLds carve_tap_mf(float* smem, const RingParams& p) {
    TapLds s;
    s.xin = smem;
    s.flags = reinterpret_cast<int*>(smem + (size_t)TAP_XBUF * TB * (RW * p.kper + TAP_XPAD));
    s.dv = reinterpret_cast<float2*>(s.flags + 32);
    s.zl = reinterpret_cast<float*>(s.dv + 4 * RT);                  // (dv: 8 floats per thread -- a unit's four stores held back)
    s.wl = reinterpret_cast<float4*>(s.zl + 2 * TB * GC);
    return s;
}
__host__ __device__ inline size_t tap_lds_floats_mf(int kper, int klds_rows) {
    return (size_t)TAP_XBUF * TB * (RW * kper + TAP_XPAD) + 32 + 8 * RT + (size_t)2 * TB * GC + (size_t)RW * klds_rows * 64 * 4;
}

// ---- THE TAP ROLE ON THE MATRIX PIPE (round 6; kernels wnv_ring_kernel_mf, chosen by the host for models with ONE tap workgroup per layer at
// every batch size: the 30-layer models -- see "THE TAP ROLE ON THE MATRIX PIPE" in the host code).  Same inputs, records, history rings and
// publish format as run_tap; what differs is the mat-vec and the pass structure around it:
//  * v_mfma_f32_16x16x4_f32: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16], result register v of a lane = D[4 (lane / 16) + v][lane % 16].
//    A wave's 32 output rows are two tiles; K runs in groups of 16: lane (n = lane & 15, kq = lane >> 4) reads x[utterance n][16 J + 4 kq .. + 3]
//    with ONE 16-byte LDS read and feeds four MFMAs per tile with it (MFMA i of group J sums k = 16 J + 4 kq' + i over kq'); its weights
//    W[16 J + 4 kq + i][row n of the tile] live in registers for J < TAP_NJR (128) and as two float4 per group in LDS beyond.
//  * The sixteen columns are the utterances of a UNIT: two neighbouring passes (2 m, 2 m + 1) of eight -- or a last pass without a partner
//    (its eight columns repeated).  One input phase (every wave finishes its utterance of pass A, then of pass B: [B] of run_tap twice), one
//    barrier, the held publish's release, the multiplication with the next unit's gathers and looks issued from inside the MFMA stream, four
//    16-byte stores per lane -- held back behind the next unit's barrier where the workgroup runs two or more units per step.
//  * Why: run_tap's mat-vec -- 4 utterances per round, inputs broadcast from LDS to every lane group of every wave -- is issue-bound at 41 % of
//    the CU's FMA peak (4.7 clocks per v_pk_fma_f32 + 16 per ds_read_b128, the two waves of a SIMD do not hide each other); this form reads an
//    input once per sixteen columns and multiplies at 98 % of the matrix pipe in isolation (1.14 us per four utterances against 1.71:
//    profiles/r06_tap_waves.txt, scripts/ubench_tapmv.hip VAR 11).  A column's arithmetic does not depend on its neighbours, the choice of the
//    kernel depends on the model only: pre_l is the same bits whatever the batch size or the packing (tests/test_gpu_seed_determinism.py).
template <bool SPEC, bool PACKED, bool DEFER>
__device__ __attribute__((always_inline)) void run_tap_mf(const RingParams& p, int l, int part, float* smem) {
    if (WNV_EXP_NOPRE >= 2) return;
    const TapLds s = carve_tap_mf(smem, p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = p.lay_dil[l];
    const int rows = (p.kw - 1) * d;
    const int hoff = (p.kw - 1) * RC;
    const int kx = RW * p.kper + TAP_XPAD;                         // the stride of a mat-vec input row in LDS (4 x odd mod 64: sixteen rows, sixteen bank quads)
    const float* Wt = p.wpre + (size_t)l * p.kpre * GC;           // K-major [kpre][256]
    // column (an utterance) / row of a tile mn = lane & 15, K quarter mkq = lane >> 4; wa[(2 J + tile) 4 + i] = W[16 J + 4 mkq + i][32 wave + 16 tile + mn]
    const int mn = lane & 15, mkq = lane >> 4;
    const int nj = p.tap_nj;
    float wa[TAP_NJR * 8];
    {
        auto wk = [&](int J, int tl, int i_) { const int k = 16 * J + 4 * mkq + i_; return k < p.kpre ? Wt[(size_t)k * GC + 32 * wave + 16 * tl + mn] : 0.f; };
#pragma unroll
        for (int J = 0; J < TAP_NJR; ++J)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) wa[(2 * J + tl) * 4 + i_] = wk(J, tl, i_);
        for (int J = TAP_NJR; J < nj; ++J)
            for (int tl = 0; tl < 2; ++tl)
                s.wl[((size_t)wave * p.klds_rows + 2 * (J - TAP_NJR) + tl) * 64 + lane] = make_float4(wk(J, tl, 0), wk(J, tl, 1), wk(J, tl, 2), wk(J, tl, 3));
    }
    // (the bias rows are the MODEL's gate rows -- tanh rows [0, G/2), sigmoid rows [G/2, G) --; this kernel's 256 outputs are tanh channels 0..127
    //  then sigmoid channels 0..127, zero beyond G/2)
    for (int i = tid; i < TAP_XBUF * TB * kx; i += RT) s.xin[i] = 0.f;
    if (tid == 0) s.flags[0] = 0;
    __syncthreads();

    // A layer with dilation >= 2 needs NOTHING of this step's h for pre[t + 1] (its taps are h[t+1-d], h[t+1-2d], ...): its pass does not
    // wait for h[t] -- the row is waited for and filed by the NEXT pass of the utterance (tf = t - 1), a whole step later --, so the rings
    // never wait for this layer's taps (pre_rec: why the records come in two slots).  With d = 2 the youngest tap of step t + 1 is h[t-1],
    // the row that pass files: taken from the record, as a dilation-1 layer takes h[t].
    // slot of a step in the history ring: rows = (kw - 1) d is a power of two for every kernel size 3 model (d = 2^i): a mask, not the
    // ~40-instruction integer division the general case costs every lane of every gather (uniform: one compare)
    const int rmask = rows > 0 && (rows & (rows - 1)) == 0 ? rows - 1 : -1;
    auto ring_slot = [&](int step) -> int { return rmask >= 0 ? (step & rmask) : step % rows; };
    const bool early = d >= 2 && rows > 0;
    const int ntap4 = hoff / 4, ncin4 = (p.cin & 3) == 0 ? p.cin / 4 : 0;      // float4s of a mat-vec input: tap rows, conditioning row
    constexpr int GQ = 2;                                          // ... per lane ((kw - 1) 128 + cin <= 512 floats: why_not)
    // this workgroup's UNITS of a step: the two neighbouring passes (2 m, 2 m + 1) -- utterances [b, b + tb) and [b + tb, b + 2 tb): the partner of
    // a pass is the next token of every ring -- are multiplied together; the units are dealt to the layer's parts alternately: b = bfirst,
    // + pstride, ... < B.  (The host launches these kernels for models with ONE part per layer.)
    const int pstride = 2 * p.tap_parts * p.tb;
    const int bfirst = 2 * part * p.tb;
    if (bfirst >= p.B) return;
    auto fresh_tap = [&](int tf_) { return ((d == 1 || early) && d <= 2 && tf_ >= 0 && rows > 0) ? p.kw - 2 : -1; };   // the tap that is h[tf] itself

    // ---- SOFTWARE PIPELINE (round 4).  A pass = [B] finish the inputs (wave w = utterance b0 + w: gathered rows -> LDS, h_l[tf] from its
    //      record -> history ring and LDS), barrier, [C] ISSUE the next pass's gather, [D] mat-vec + reduce + publish.  The gather -- the
    //      kw-1 older taps (one contiguous 512-byte history row each; zeros before t = 0: the rings start zeroed) and c[t + 1], 16-byte
    //      loads, two per lane -- is ~1.5 us of global-load latency; issued at [C] it lands under the mat-vec of the pass before (until
    //      round 4 it was paid in front of every pass: beyond 32 utterances per GPU the passes of the tap workgroups bound the
    //      throughput, profiles/r04_tap_bound_experiment.txt).  The next pass's h record is looked at speculatively in the same
    //      breath (two granules per lane): when the utterance's stage has filed it already -- the rule in that regime -- [B] takes it
    //      without a poll round trip.  What the next pass gathers was filed by THIS wave at the latest in [B] of this pass (a
    //      dilation-1 layer: h[t], one step back; in order in this CU's L1); the row the next pass files itself is never gathered.
    // The gathered rows go global -> LDS by DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land at base + 16 lane; no staging
    // registers -- next to 128 weight registers there are none to spare: the register-staged form spilled 27) into the OTHER input
    // buffer; the issuing wave waits for its own DMAs (vmcnt) in [B] of the pass that uses them, in front of the barrier.
    u64 hxa0 = 0, hxa1 = 0, hxb0 = 0, hxb1 = 0;                      // the speculative looks in flight: the two passes of a unit
    // packed slots: seg_start of (utterance b0_ + wave, step t_ + 1) -- a scalar load (no vector register held across the DMA issue),
    // asked for ONE PASS AHEAD of the gather that needs it (round 5: read at the top of gather_issue, every pass began its mat-vec behind
    // a scalar-load round trip -- the packed instantiations ran 12-20 % below the padded ones at the same number of rows)
    auto seg_at = [&](int t_, int b0_) -> int {
        if (!PACKED || wave >= min(p.tb, p.B - b0_) || t_ + 1 >= p.T) return INT_MIN;
        return uniform_ld(p.seg_start + (size_t)__builtin_amdgcn_readfirstlane(b0_ + wave) * p.T + t_ + 1);
    };
    auto gather_issue = [&](int t_, int b, float* xu, int buf, int st_, u64& hx0, u64& hx1) {
        const int tp_ = t_ + 1, tf_ = early ? t_ - 1 : t_, kf = fresh_tap(tf_);
        const float* hb = p.hist + (size_t)b * p.hist_floats + p.lay_histoff[l];
        const float* cb = p.c_up + ((size_t)b * p.T + tp_) * p.cin;
        // packed slots: the utterance that occupies the slot at step tp_ began at step st_; what lies before reads as zeros (conv.py:34-36)
        // (the lane-constant pieces of the addresses are formed HERE, from a copy of the lane id the compiler cannot see through: kept live
        //  across the pass they were the registers that spilled once the deferred publish joined the loop -- a reload per pass against
        //  three VALU instructions)
        int lane_l = lane;
        if constexpr (DEFER) asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int i = 64 * q + lane_l;
            const float* src = nullptr;
            if (i < ntap4) {
                const int k = i >> 5, r4 = i & 31;                   // RC / 4 = 32 float4s per row
                // (a row from before the utterance's start comes from a row of zeros: a select on the address, no branch -- the tap role has
                //  no register to spare)
                if (k != kf) src = tp_ - (p.kw - 1 - k) * d >= st_ ? hb + (size_t)ring_slot(tp_ + k * d) * RC + 4 * r4 : p.zero_row + 4 * r4;
            } else if (i < ntap4 + ncin4) {
                src = cb + 4 * (i - ntap4);
            }
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)(xu + 256 * q));
            if (src) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds) : "memory");
        }
        hx0 = hx1 = 0;
        if (SPEC && tf_ >= 0) {                                     // speculative look at the record (L1-bypassing, not waited for here)
            const u64* rec = p.fmail + h_rec(p, b, l, tf_) + 2 * lane;
            hx0 = ld_granule(rec); hx1 = ld_granule(rec + 1);
        }
    };

#ifdef WNV_FINE_TRACE
#define TAP_STAMP(k) do { const int pk_ = (b0 - bfirst) / p.tb; \
                          if (p.trace_tap && l == WNV_TRACE_TAP_LAYER && part == 0 && pk_ < 3 && (k) < 5 && tid == 64 * WNV_TRACE_TAP_WAVE && t >= p.trace_t0 && t < p.trace_t0 + p.trace_n) \
                              p.trace_tap[(size_t)(t - p.trace_t0) * TRW + 5 * pk_ + (k)] = wall_clock64(); } while (0)
#else
#define TAP_STAMP(k) ((void)0)
#endif
    // ---- [B] wave w finishes the mat-vec input of utterance b0_ + w of pass (t_, b0_) in buffer cur_: the h record it waits for (looked at
    //      speculatively with the gather: hx0, hx1) -> history ring and LDS; then its own DMAs (issued with the gather) have landed -------------
    auto do_B = [&](int t_, int b0_, int cur_, int st_, u64 hx0, u64 hx1) {
        const int tp_ = t_ + 1, nb_ = min(p.tb, p.B - b0_);
        const int tf_ = early ? t_ - 1 : t_;                        // the step whose h this pass waits for and files
        const int kfresh = fresh_tap(tf_);
        if (wave < nb_) {
            const int b = b0_ + wave;
            float* xu = s.xin + ((size_t)cur_ * TB + wave) * kx;
            // (round 6, WNV_TAP_ZLDS) the utterance's effective conv bias row -- b_l + W_g g: per utterance when the model has a speaker
            // embedding -- is fetched HERE, with the inputs, and handed to the rounds through LDS.  It used to be two global loads per lane at
            // the top of every round: in round 0 they were issued right behind the next pass's gather DMAs, loads return in order, and the
            // round's publish waited for the whole gather (history rows from the L2, the conditioning row from HBM).  PACKED instantiations
            // only: there [B] has a record round trip for the fetch to hide under (no speculative look) and a packed job gains 3.6-4.4 %
            // (egs/mol 100 utterances 2 292 -> 2 375, cfg4 128 utterances 1 166 -> 1 217 kSamples/s, same box); in the throughput
            // instantiation, whose speculative look usually hits, the fetch sits in the open in front of the pass's barrier: egs/mol
            // +0.4 %, mu-law -1.9 %, cfg4 -4 % (profiles/r06_tap_zlds_ab.txt) -- so not there.
            float4 zrow4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PACKED && WNV_TAP_ZLDS != 0) {
                size_t zr = (size_t)b;
                if (PACKED && p.seg_gid) zr = (size_t)uniform_ld(p.seg_gid + (size_t)__builtin_amdgcn_readfirstlane(b) * p.T + tp_);
                if (4 * lane < p.zb_ld) zrow4 = *reinterpret_cast<const float4*>(p.zbias + (size_t)l * p.zb_ld + zr * p.zbias_bstride + 4 * lane);
            }
            if (ncin4 == 0) {                                        // cin not a multiple of 4: scalar conditioning row
                const float* cb = p.c_up + ((size_t)b * p.T + tp_) * p.cin;
                for (int e = lane; e < p.cin; e += 64) xu[hoff + e] = cb[e];
            }
            if (tf_ >= 0) {
                const unsigned htag = p.tag_base + (unsigned)tf_ + 1u;
                float hv[2] = {__uint_as_float((unsigned)hx0), __uint_as_float((unsigned)hx1)};         // channels 2 lane, 2 lane + 1
#ifdef WNV_FINE_TRACE
                if (lane == 0) s.flags[24 + wave] = 0;                 // (trace: how this wave got its record -- 0 look, 1 direct look, 2 patient receive)
#endif
                if (!__all((unsigned)(hx0 >> 32) == htag && (unsigned)(hx1 >> 32) == htag)) {
                    // (round 6) ONE direct look at the whole record first: while the stages pace the passes the record lands during the pass
                    // before -- behind the speculative look, ahead of this one -- and the patient receive (first granule at a relaxed cadence,
                    // then the record: two round trips and a sleep at best) made every pass wait ~1.4 us at its barrier for the one wave
                    // whose speculative look had missed (profiles/r06_tap_pass_timeline.txt)
                    const u64* rec = p.fmail + h_rec(p, b, l, tf_);
                    const u4v x = ld16_sc1(rec + 2 * lane);
                    hv[0] = __uint_as_float(x.x); hv[1] = __uint_as_float(x.z);
#ifdef WNV_FINE_TRACE
                    if (lane == 0) s.flags[24 + wave] = __all(x.y == htag && x.w == htag) ? 1 : 2;
#endif
                    if (!__all(x.y == htag && x.w == htag)) {
                        if (!rec_recv<1>(rec, htag, hv, p.status, 0x600u + (unsigned)l, lane)) s.flags[0] = 1;
                    }
                }
                if (rows > 0) {
                    float* hist = p.hist + (size_t)b * p.hist_floats + p.lay_histoff[l];
                    *reinterpret_cast<float2*>(hist + (size_t)ring_slot(tf_) * RC + 2 * lane) = make_float2(hv[0], hv[1]);
                    if (kfresh >= 0) {
                        // (packed slots: the row is the previous utterance's when the one at step tp began later than tf)
                        const bool mine = !PACKED || tf_ >= st_;
                        *reinterpret_cast<float2*>(xu + kfresh * RC + 2 * lane) = mine ? make_float2(hv[0], hv[1]) : make_float2(0.f, 0.f);
                    }
                }
            }
            if constexpr (PACKED && WNV_TAP_ZLDS != 0) *reinterpret_cast<float4*>(s.zl + ((size_t)(cur_ & 1) * TB + wave) * GC + 4 * lane) = zrow4;   // (slot = the pass's parity)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs into buffer cur_ (issued with the gather) have landed
        }
    };
    // A UNIT'S PUBLISH IS HELD BACK until the next unit's input phase and barrier are through (parked in LDS: TapLds::dv).  A publish is a
    // write-through store (the stage may sit on any XCD), its acknowledgement takes 1.5-2 us, and every wait of the wave for a LOAD -- the h
    // record's poll, the DMA wait -- waits for it too (vmcnt counts stores).  ONLY where this workgroup runs at least two units per step: with
    // a single one the next unit is the same utterances' next step, whose h cannot exist before the publish it would be holding back.
    int dlo_b0 = 0, dlo_nb = 0, dlo_tp = 0, dhi_b0 = 0, dhi_nb = 0, dhi_tp = 0;   // the held pair: first utterance / utterances / step of its low and high columns
    bool dheld = false;
    typedef float f4m __attribute__((ext_vector_type(4)));
    // what column mn of a multiplication means: utterance, step, valid
    auto colmap = [&](int lo_b0, int lo_nb, int lo_tp, int hi_b0, int hi_nb, int hi_tp, int& rb_, int& tp_, bool& pub_) {
        const bool lowc = mn < 8;
        rb_ = (lowc ? lo_b0 : hi_b0) + (mn & 7); tp_ = lowc ? lo_tp : hi_tp; pub_ = (mn & 7) < (lowc ? lo_nb : hi_nb);
    };
    auto mf_publish = [&](int rb_, int tp_, const f4m& v0_, const f4m& v1_) {          // eight values of one utterance: four 16-byte write-through stores
        u64* rec = p.pmail + pre_rec(p, rb_, l, tp_) + 32 * wave + 4 * mkq;
        const unsigned tg = p.tag_base + (unsigned)tp_ + 1u;
        st_granule2(rec, tg, v0_.x, v0_.y, false); st_granule2(rec + 2, tg, v0_.z, v0_.w, false);
        st_granule2(rec + 16, tg, v1_.x, v1_.y, false); st_granule2(rec + 18, tg, v1_.z, v1_.w, false);
    };
    {
        // ---- THE UNIT LOOP.  A unit = the two neighbouring passes (2 m, 2 m + 1) of a multiplication, or a last pass without a partner:
        //      ONE input phase (every wave finishes its utterance of pass A, then of pass B), ONE barrier, the held publish's release, the
        //      multiplication -- with the NEXT unit's gathers and looks issued from inside the MFMA stream (the matrix pipe runs on while the
        //      wave forms addresses and issues DMAs) --, publish or hold.
        auto unit_next = [&](int& t_, int& b_) { b_ = (b_ / (2 * p.tb)) * (2 * p.tb) + pstride; if (b_ >= p.B) { b_ = bfirst; t_ += 1; } };
        int ups = 0;                                                 // units of this workgroup per step
        { int t_ = 0, b_ = bfirst; do { ++ups; unit_next(t_, b_); } while (t_ == 0); }
        const bool hold_ok = DEFER && WNV_TAP_DEFER != 0 && ups >= 2;   // (the unit after a held one must not wait for an h that needs the held publish)
        int ut = -1, ub = bfirst, upc = 0;
        int stA = INT_MIN, stB = INT_MIN, stAn = INT_MIN, stBn = INT_MIN;
        auto issue_unit = [&](int t_, int b_, int base_, int& sa_, int& sb_) {       // the gathers + looks of a unit's passes, into buffers base_, base_ + 1
            sa_ = seg_at(t_, b_);
            if (wave < min(p.tb, p.B - b_)) gather_issue(t_, b_ + wave, s.xin + ((size_t)base_ * TB + wave) * kx, base_, sa_, hxa0, hxa1);
            const int bb = b_ + p.tb;
            sb_ = INT_MIN;
            if (bb < p.B) {
                sb_ = seg_at(t_, bb);
                if (wave < min(p.tb, p.B - bb)) gather_issue(t_, bb + wave, s.xin + ((size_t)(base_ + 1) * TB + wave) * kx, base_ + 1, sb_, hxb0, hxb1);
            }
        };
        issue_unit(ut, ub, 0, stA, stB);
        for (;;) {
            const int t = ut, b0 = ub; (void)t; (void)b0;                // (names the stamps use)
            const int tp = ut + 1, base = 2 * (upc & 1);
            const int bA = ub, bB = ub + p.tb;
            const bool hasB = bB < p.B;
            const int nbA = min(p.tb, p.B - bA), nbB = hasB ? min(p.tb, p.B - bB) : 0;
            TAP_STAMP(0);
            do_B(ut, bA, base, stA, hxa0, hxa1);
            if (hasB) do_B(ut, bB, base + 1, stB, hxb0, hxb1);
            TAP_STAMP(1);
            __syncthreads();                                             // the inputs of the unit are complete in buffers base, base + 1
            if (s.flags[0]) return;
            TAP_STAMP(2);
            if (dheld) {                                                 // the held unit's publish
                int rb_, tp_; bool pub_;
                colmap(dlo_b0, dlo_nb, dlo_tp, dhi_b0, dhi_nb, dhi_tp, rb_, tp_, pub_);
                if (pub_) {
                    const float4 a = reinterpret_cast<const float4*>(s.dv)[tid], c = reinterpret_cast<const float4*>(s.dv)[RT + tid];
                    mf_publish(rb_, tp_, f4m{a.x, a.y, a.z, a.w}, f4m{c.x, c.y, c.z, c.w});
                }
                dheld = false;
            }
            int tn = ut, bn = ub;
            unit_next(tn, bn);
            const bool more = tn + 1 < p.T;
            const int nbase = 2 * ((upc + 1) & 1);
            int rb, ctp; bool pubc;
            colmap(bA, nbA, tp, bB, nbB, tp, rb, ctp, pubc);
            // addends of this lane's outputs (rows 32 wave + 16 tile + 4 mkq + v of utterance rb), asked for ahead of the MFMAs
            float4 cv[2], zb[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int po_ = 32 * wave + 16 * tl + 4 * mkq, zh = po_ >> 7, zc = po_ & 127;
                cv[tl] = *reinterpret_cast<const float4*>(p.cvec + (size_t)l * GC + po_);
                float zz[4] = {0.f, 0.f, 0.f, 0.f};
                if (pubc) {
                    const float* zq;
                    if constexpr (PACKED && WNV_TAP_ZLDS != 0) zq = s.zl + ((size_t)(mn < 8 ? 0 : 1) * TB + (mn & 7)) * GC + zh * p.gh + zc;
                    else zq = p.zbias + (size_t)l * p.zb_ld + (size_t)zh * p.gh + zc + (size_t)((PACKED && p.seg_gid) ? p.seg_gid[(size_t)rb * p.T + ctp] : rb) * p.zbias_bstride;
#pragma unroll
                    for (int i_ = 0; i_ < 4; ++i_) if (zc + i_ < p.gh) zz[i_] = zq[i_];
                }
                zb[tl] = make_float4(zz[0], zz[1], zz[2], zz[3]);
            }
            const int xrow = base * TB + ((hasB || mn < 8) ? mn : (mn & 7));      // (a lone pass repeats its eight columns)
            const float* xr = s.xin + (size_t)xrow * kx + 4 * mkq;
            auto xread = [&](int J) { return *reinterpret_cast<const float4*>(xr + 16 * min(J, nj - 1)); };
            f4m d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#if WNV_MF_GATHER_AT < 0
            if (more) issue_unit(tn, bn, nbase, stAn, stBn);         // (in front of the multiplication)
#endif
            float4 xv = xread(0);
#pragma unroll
            for (int J = 0; J < TAP_NJR; ++J) {
                const float4 xn = xread(J + 1);
                __builtin_amdgcn_sched_barrier(0);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[(2 * J) * 4 + i_], xs[i_], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[(2 * J + 1) * 4 + i_], xs[i_], d1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                xv = xn;
#if WNV_MF_GATHER_AT >= 0
                if (J == WNV_MF_GATHER_AT) { if (more) issue_unit(tn, bn, nbase, stAn, stBn); __builtin_amdgcn_sched_barrier(0); }
#endif
            }
#pragma unroll 1
            for (int J = TAP_NJR; J < nj; ++J) {
                const float4 w0 = s.wl[((size_t)wave * p.klds_rows + 2 * (J - TAP_NJR)) * 64 + lane], w1 = s.wl[((size_t)wave * p.klds_rows + 2 * (J - TAP_NJR) + 1) * 64 + lane];
                const float4 xn = xread(J + 1);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i_], xs[i_], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i_], xs[i_], d1, 0, 0, 0);
                }
                xv = xn;
            }
            TAP_STAMP(3);
            const float zs_ = WNV_PHASE2 ? (wave >= 4 ? GATE_SCALE_SIGM : GATE_SCALE_TANH) : 1.0f;
            const f4m v0 = {d0.x + fmaf(zb[0].x, zs_, cv[0].x), d0.y + fmaf(zb[0].y, zs_, cv[0].y), d0.z + fmaf(zb[0].z, zs_, cv[0].z), d0.w + fmaf(zb[0].w, zs_, cv[0].w)};
            const f4m v1 = {d1.x + fmaf(zb[1].x, zs_, cv[1].x), d1.y + fmaf(zb[1].y, zs_, cv[1].y), d1.z + fmaf(zb[1].z, zs_, cv[1].z), d1.w + fmaf(zb[1].w, zs_, cv[1].w)};
            if (hold_ok && more) {
                reinterpret_cast<float4*>(s.dv)[tid] = make_float4(v0.x, v0.y, v0.z, v0.w);
                reinterpret_cast<float4*>(s.dv)[RT + tid] = make_float4(v1.x, v1.y, v1.z, v1.w);
                dlo_b0 = bA; dlo_nb = nbA; dlo_tp = tp; dhi_b0 = bB; dhi_nb = nbB; dhi_tp = tp; dheld = true;
            } else if (pubc) mf_publish(rb, ctp, v0, v1);
            TAP_STAMP(4);
            if (!more) break;
            ut = tn; ub = bn; ++upc; stA = stAn; stB = stBn;
        }
    }
#undef TAP_STAMP
}

// ---- the 256-row mat-vec of a WAVE GROUP (four waves = one wave per SIMD) -------------------------------------------------
// A stage's eight waves form two groups: waves 0-3 hold M_l and are the only ones that work while the chain waits (nothing shares
// their SIMDs), waves 4-7 hold N_l and prepare zin = N_l h_{l-1} + pre_l ahead of it.  (Until round 3 every wave held half of M and
// half of N: the two waves of a SIMD then finished the chain mat-vec one after the other -- the second wave's store left ~0.14 us
// after the first's, profiles/r02_ring_v12_fine_timeline.txt -- and every wave repeated the reduce and the gate.)
// Group thread gtid = (og = gtid >> 3, ks = gtid & 7): eight adjacent lanes split K = 128 (16 floats each) for the EIGHT rows
// {tanh, sigmoid} x channels 4og .. 4og + 3.  The reduce-scatter needs no selects because the register SLOT of a row depends on
// the lane (the host lays the image out that way, put_row8g): with b2 = ks >> 2, b1 = (ks >> 1) & 1 a lane's slot pairs hold
// channels 4og + {2b2 + b1, 2b2 + 1-b1, 2(1-b2) + 1-b1, 2(1-b2) + b1}; step 1 (row_half_mirror, lane j <-> 7 - j) adds the partner's
// slots 4-7 to slots 0-3, step 2 (lane j <-> j ^ 2) the partner's 2-3 to 0-1, step 3 (j <-> j ^ 1) completes both rows of
// channel 4og + (ks >> 1) in lanes ks and ks ^ 1: 8 DPP adds for 8 rows.
constexpr int GT = 256;            // threads of a wave group
__device__ __forceinline__ void load_image8g(const float* img, int gtid, f2 (&w)[8]) {
    const float4* src = reinterpret_cast<const float4*>(img);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = src[(size_t)c * GT + gtid];
        w[2 * c] = f2{v.x, v.y}; w[2 * c + 1] = f2{v.z, v.w};
    }
}
// (za, zb: two addends read from LDS ahead of the x slice -- their latency hides under the FMAs -- and added to a and g)
__device__ __forceinline__ void group_matvec8(const f2 (&w)[8][8], const float* xslice, const float* za, const float* zb, float& a, float& g) {
    float2 z = make_float2(*za, *zb);
    float x[16];
    lds_read16(xslice, x);
    f2 acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = __builtin_elementwise_fma(w[s][k], f2{x[2 * k], x[2 * k + 1]}, acc[s]);
    asm volatile("" : "+v"(z.x), "+v"(z.y));                     // the reads were issued up there, not behind the reduce
    float q[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) q[s] = acc[s].x + acc[s].y;
    const float n0 = dpp_fold<0x141>(q[0], q[4]), n1 = dpp_fold<0x141>(q[1], q[5]);
    const float n2 = dpp_fold<0x141>(q[2], q[6]), n3 = dpp_fold<0x141>(q[3], q[7]);
    const float m0 = dpp_fold<0x4E>(n0, n2), m1 = dpp_fold<0x4E>(n1, n3);
    a = dpp_fold<0xB1>(m0, m0) + z.x;
    g = dpp_fold<0xB1>(m1, m1) + z.y;
}

// ---- ROUND 5: the same mat-vec with a SHALLOWER dependent tail (scripts/ubench_phase.hip, profiles/r05_ubench_phase.txt) -----------
// The chain phase is 77 ns of barrier + store, 51 ns of LDS read, 116 ns of FMAs -- and 133 ns of reduce + gate: ~30 instructions that
// form ONE dependent chain of 14 levels on a SIMD that holds a single wave, ~7-10 ns per level whatever the instruction order (the
// hand-scheduled tail of variant 5 is not faster than the compiler's).  What helps is fewer LEVELS:
//   * ROW-PAIR accumulators: a packed FMA holds the {tanh, sigmoid} rows of one channel against a broadcast x (op_sel: free) instead of
//     two K halves of one row -- the x + y level (and its 8 adds) is gone; 4 accumulators x 16 k, still 64 v_pk_fma_f32;
//   * the gate on pre-scaled arguments as (2 r1 - 1) r2 (ring_gate: three levels less).
// Register slots: pair p = image slots 2p (tanh) and 2p + 1 (sigmoid) of put_row8g -- the host image is unchanged, the pairs are formed
// as the registers are loaded.
__device__ __forceinline__ void load_pair8g(const float* img_slot0, int gtid, f2 (&w)[4][16]) {
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
        const float4* sa = reinterpret_cast<const float4*>(img_slot0 + (size_t)(2 * pq) * 4 * GT * 4);
        const float4* sb = reinterpret_cast<const float4*>(img_slot0 + (size_t)(2 * pq + 1) * 4 * GT * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 va = sa[(size_t)c * GT + gtid], vb = sb[(size_t)c * GT + gtid];
            w[pq][4 * c] = f2{va.x, vb.x}; w[pq][4 * c + 1] = f2{va.y, vb.y}; w[pq][4 * c + 2] = f2{va.z, vb.z}; w[pq][4 * c + 3] = f2{va.w, vb.w};
        }
    }
}
// (the addends za, zb are read ahead of the x slice and added behind the reduce: as the init of the accumulators -- variant 12 -- the
//  first FMAs wait for them, +14 ns; the last DPP level as one hand-written instruction -- variant 11 -- buys nothing)
#ifndef WNV_PHASE2_ZACC
#define WNV_PHASE2_ZACC 1      // the addends as the init of accumulator pair 0 in the writer lane of each finishing pair (ubench variant 13: 298 against 309 ns;
                               // same-box A/B of the headline: 511.3 against 507.5 kSamples/s, profiles/r05_phase2_ab.txt); 0: added behind the reduce
#endif
__device__ __forceinline__ void group_matvec8r(const f2 (&w)[4][16], const float* xslice, const float* za, const float* zb, float& a, float& g, bool zlane = true) {
    float2 z = make_float2(*za, *zb);
    float x[16];
    lds_read16(xslice, x);
    f2 acc[4];
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) acc[pq] = f2{0.f, 0.f};
    if (WNV_PHASE2_ZACC) acc[0] = zlane ? f2{z.x, z.y} : f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) acc[pq] = __builtin_elementwise_fma(w[pq][k], f2{x[k], x[k]}, acc[pq]);
    if (!WNV_PHASE2_ZACC) asm volatile("" : "+v"(z.x), "+v"(z.y));   // the reads were issued up there, not behind the reduce
    const float n0 = dpp_fold<0x141>(acc[0].x, acc[2].x), n1 = dpp_fold<0x141>(acc[0].y, acc[2].y);
    const float n2 = dpp_fold<0x141>(acc[1].x, acc[3].x), n3 = dpp_fold<0x141>(acc[1].y, acc[3].y);
    const float m0 = dpp_fold<0x4E>(n0, n2), m1 = dpp_fold<0x4E>(n1, n3);
    a = dpp_fold<0xB1>(m0, m0);
    g = dpp_fold<0xB1>(m1, m1);
    if (!WNV_PHASE2_ZACC) { a += z.x; g += z.y; }
}

// One stage = one gated layer on one CU, weights resident in VGPRs.
//
// GATE-TO-GATE CHAIN.  The reference's layer is  z_l = W_cur,l h_l + pre_l ;  u_l = tanh . sigmoid (z_l) ;
// h_{l+1} = sqrt(.5) (W_o,l u_l + b_o,l + h_l)   (modules.py:127-163).  Substituting h_l into z_l,
//     z_l = M_l u_{l-1} + N_l h_{l-1} + c_l + pre_l ,   M_l = sqrt(.5) W_cur,l W_o,l-1 ,  N_l = sqrt(.5) W_cur,l ,  c_l = N_l b_o,l-1
// (M, N, c folded once on the host, in double).  h_{l-1} is known a whole layer earlier than u_{l-1}, so only
// M_l u_{l-1} -> gate -> send u_l  is on the chain (waves 0-3); the N_l mat-vec (waves 4-7, ahead of the chain), conv1x1_out
// (the h recurrence itself, bit for bit the reference's), the skip 1x1, the history push and the older taps all happen off it.
// The values exchanged:  X[l] = u_{l-1} (X[0] = h_0 from the head) over the chain mailbox;  H[l] = h_l, written by
// stage l-1 into a mailbox with one slot per step parity and read by stage l (residual, history) and stage l+1 (N);
// two slots because those reads are off the chain: a slot is rewritten two steps later, which the data flow itself
// orders behind every reader (the writer's gate of step t+2 needs the sample of step t+1, hence every stage's step t+1).
//
// Thread mapping of the work behind the send (all eight waves): eight adjacent lanes split the K = 128 contraction (16 floats
// each), a group of eight lanes owns channels 2og and 2og + 1; reductions are reduce-scatters (first DPP step row_half_mirror,
// lane j <-> 7 - j, hands lanes 0-3 the sums of channel 2og and lanes 4-7 those of 2og + 1; two quad_perm steps finish).
// (L0: the ring's head evaluates layer 0 -- run_head; ZMSG: this instantiation is stage 1 of such a ring.  Compile-time switches:
//  the stage loop is codegen-sensitive, a run-time flag in it costs every stage 3 %.)
// (MULTI: the instantiation for rings that carry several utterances -- more than 8 per GPU --, where the stage's OCCUPANCY per utterance
//  bounds the throughput: everything ahead of the chain input is asked for in one round trip.  A compile-time switch: the same
//  prologue in the single-utterance instantiation costs the headline 1.3 % through the register allocation of the stage loop
//  (same box: 504.8 / 503.4 against 497.7 / 497.2 kSamples/s), while 48 utterances gain 3.5 %.)
template <int NK, bool L0, bool ZMSG, bool MULTI>
__device__ __attribute__((always_inline)) void run_stage(const RingParams& p, int ring, int sidx, float* smem) {
    constexpr int NLDS = lds_passes(NK);
    constexpr bool RP = true;                   // pipelined polls in the reserved registers (until round 4 the K = 512 instantiation needed them itself)
    const StageLds s = carve_stage(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = tid & 7, og = tid >> 3;                      // K-slice; lane group = channels 2og, 2og + 1
    const bool hi = ks >= 4;                                    // lanes 4-7 finish channel 2og + 1, lanes 0-3 channel 2og
    const int ch = 2 * og + (hi ? 1 : 0);
    const bool writer = (ks & 3) == 0;                          // lanes 0 and 4 of the group publish
    // wave groups: grp 0 = waves 0-3 (M_l, the chain), grp 1 = waves 4-7 (N_l); group lane group gog owns channels 4gog .. 4gog + 3
    const int grp = tid >> 8, gtid = tid & (GT - 1);
    const int chc = 4 * (gtid >> 3) + (ks >> 1);                // the channel whose (tanh, sigmoid) rows end up in this lane
    const bool gwriter = (ks & 1) == 0;
    const int l = sidx;
    const bool first_stage = sidx == 0, last_stage = sidx == p.S - 1;
    const int S1 = p.S + 1;
    WNV_TS_DECL;

    // ---- resident weights (registers), pairs along K (every FMA on or near the chain is a v_pk_fma_f32):
    //      wmn = the eight rows of M_l (waves 0-3) or N_l (waves 4-7) this lane contracts, wo rows = {c0, c1} of conv1x1_out;
    //      conv1x1_skip (nobody waits for it) is read from an LDS image ----------------------------------------------
#if WNV_PHASE2
    f2 wmn[4][16], wo[2][8];                   // row pairs {tanh, sigmoid} of a channel against a broadcast x (group_matvec8r)
    load_pair8g((grp == 0 ? p.w2img : p.wnimg) + ((size_t)l * 8) * 4 * GT * 4, gtid, wmn);       // N_l is all zeros at stage 0
#else
    f2 wmn[8][8], wo[2][8];
#pragma unroll
    for (int slot = 0; slot < 8; ++slot)       // N_l is all zeros at stage 0
        load_image8g((grp == 0 ? p.w2img : p.wnimg) + ((size_t)l * 8 + slot) * 4 * GT * 4, gtid, wmn[slot]);
#endif
    const float4* wsk_g = reinterpret_cast<const float4*>(p.wsimg) + (size_t)l * NK * 8 * RT;    // this layer's conv1x1_skip image
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        // the last layer's conv1x1_out is never used (wavenet.py:310-313): its registers hold a skip pass there -- the first one that
        // does not fit LDS (K = 512), else pass 0: the last stage's skip term is the head's input, and a pass read from the LDS
        // image is LDS-bandwidth-bound (64 KiB per pass: ~0.2 us)
        if (last_stage)
            load_image8(reinterpret_cast<const float*>(wsk_g + ((size_t)(NK > NLDS ? NLDS : 0) * 2 + row) * 4 * RT), tid, wo[row]);
        else
            load_image8(p.woimg + ((size_t)l * 2 + row) * 4 * RT * 4, tid, wo[row]);
    }
    for (int c = 0; c < NLDS * 8; ++c) s.wsk[(size_t)c * RT + tid] = wsk_g[(size_t)c * RT + tid];
    const float bo_r = p.bo[(size_t)l * RC + ch];
    const float bs_r = p.bskip[(size_t)l * p.Kp + ch];                 // passes >= 1 read their bias from LDS (register budget)
    for (int k = tid; k < RC * NK; k += RT) s.bsk[k] = p.bskip[(size_t)l * p.Kp + k];
    if (tid == 0) s.flags[0] = 0;
    // readers of what this stage sends: the next stage (X, H) and the one behind it (H); the head behind the last stage
    // (every head part reads the skip sum)
    // (the stage before the last: the last stage and, for the skip sum it completes, every head part -- consecutive positions)
    const int rd1 = ring + (sidx + 1) * p.rstride, rd2 = ring + (sidx + 2 <= p.S ? sidx + 2 : sidx + 1) * p.rstride;
    const bool fast = same_xcd_as(p, rd1, last_stage ? p.NH : sidx == p.S - 2 ? 1 + p.NH : 1, rd2, s.flags + 1);
    // (K = 512: the skip terms are read by the head parts -- positions S .. S + NH - 1 of the ring)
    const bool fast_head = (NK > 2 && WNV_SKIP_DIRECT) ? same_xcd_as(p, ring + p.S * p.rstride, p.NH, ring + p.S * p.rstride, s.flags + 2) : false;

#ifdef WNV_DBG_MARK
    unsigned miss_pre = 0, miss_g = 0, miss_q = 0;                      // (diagnostic build) what the throughput prologue's first look did not find
#endif
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        const int par = t & 1;
        for (int j = 0; j < p.upr; ++j) {
            const int b = ring + j * p.n_rings;
            if (b >= p.B) continue;
            // every mailbox address of this step, pinned in registers before the first wait
            const u64* x_in = p.xmail + ((size_t)b * S1 + sidx) * RC + 2 * lane;      // wave 0: two granules per lane
            u64* x_out = p.xmail + ((size_t)b * S1 + sidx + 1) * RC + chc;            // chain waves
            u64* q_out = p.hmail + (((size_t)b * 2 + par) * S1 + sidx + 1) * RC + ch;    // conv1x1_out(u_l) + b_o,l, for stage l + 2
            const u64* sm_in = p.smail + ((size_t)b * S1 + sidx) * p.Kp + ch;
            u64* sm_out = p.smail + ((size_t)b * S1 + sidx + 1) * p.Kp + ch;
            asm volatile("" : "+v"(x_in), "+v"(x_out), "+v"(q_out), "+v"(sm_in), "+v"(sm_out));
            // ---- ahead of the chain: zin = N_l h_{l-1}[t] + pre_l[t]  (pre_l[t] comes from the layer's tap workgroup and has been on
            //      its way since the previous step).  THE h RECURRENCE IS CARRIED BY THE POLLERS (round 3): stage l - 2 publishes only
            //      q = conv1x1_out(u_{l-2}) + b_o -- right after its gate, without waiting for its own layer input -- and this
            //      stage's wave 0 forms  h_{l-1} = sqrt(.5) (q + h_{l-2})  (modules.py:157-162, the reference's own arithmetic) from
            //      q and the h_{l-2} stage l - 1 handed on when IT formed its input, hands h_{l-1} on to stage l + 1, and files it for
            //      layer l - 1's tap workgroup at the end of the step.  (Before, stage l - 2 waited for h_{l-2} to arrive, added and
            //      sent h_{l-1}: ~0.13 us of every stage's wait sat on the loop  u_l -> h_{l+1} -> N_{l+2} h_{l+1} -> u_{l+2}, which --
            //      not the chain of the u's -- was what bounded the step: profiles/r03_ring_e1_fine_timeline.txt.)
            //      ONE wave waits, pre at a relaxed cadence: the stage is idle here for most of a step, and hundreds of
            //      waves polling write-through lines would load the fabric that the chain's hops share.
            float hv0 = 0.f, hv1 = 0.f;                                         // wave 0: h_{l-1}[t], channels 2 lane, 2 lane + 1
            bool hand_on = false;
            constexpr bool zmsg = ZMSG;                                         // N_1 h_0 comes ready-made from the head (run_head)
            auto recv128 = [&](const u64* g2, unsigned code, float& v0, float& v1) {
                if constexpr (RP) return rpoll_recv2<false>(g2, tag, v0, v1, p.status, code, lane);
                else return wave_recv2(g2, tag, v0, v1, p.status, code, lane, false, u4v{0, 0, 0, 0});
            };
            if (wave == 0) {
                // MULTI (several utterances per ring): everything this stage needs ahead of the chain input -- pre_l[t] (256 values, from the
                // tap workgroup), the layer input h_{l-2} handed on by stage l - 1 and the residual increment q of stage l - 2 (128 values
                // each) -- is asked for in ONE round trip; what has not arrived is then waited for on its own.  (Otherwise: pre poll, pre
                // payload, h, q, one round trip after the other -- off the chain while a ring carries one utterance, but part of the
                // stage's OCCUPANCY per utterance once it carries several, which is what bounds 48+ utterances per GPU.)
                float pvv[4] = {0.f, 0.f, 0.f, 0.f};                             // pre_l[t], outputs 4 lane .. 4 lane + 3
                const u64* prec = p.pmail + pre_rec(p, b, l, t);
                bool pre_ok = WNV_EXP_NOPRE != 0, g_ok = false, q_ok = false;
                float g0 = 0.f, g1 = 0.f, q0 = 0.f, q1 = 0.f;
                const size_t slot = (((size_t)b * 2 + par) * S1 + sidx - 1) * RC + 2 * lane;
                // h_{l-2}: there long before q (stage 2 of a ring whose head evaluates layer 0 takes h_0 from the head itself; stage 1: h_0)
                const u64* gsrc = (L0 && sidx == 2) || sidx == 1 ? p.xmail + ((size_t)b * S1) * RC + 2 * lane : p.gmail + slot;
                const u64* qsrc = p.hmail + slot;
                if (MULTI && !zmsg && !first_stage && !WNV_EXP_NOPRE) {
                    u4v ra, rb2, rc, rd;
                    ld16x4_sc1(prec + rec4_a(lane), prec + rec4_b(lane), gsrc, sidx == 1 ? gsrc : qsrc, ra, rb2, rc, rd);
                    pre_ok = __all(ra.y == tag && ra.w == tag && rb2.y == tag && rb2.w == tag);
                    g_ok = __all(rc.y == tag && rc.w == tag);
                    q_ok = sidx != 1 && __all(rd.y == tag && rd.w == tag);
                    pvv[0] = __uint_as_float(ra.x); pvv[1] = __uint_as_float(ra.z); pvv[2] = __uint_as_float(rb2.x); pvv[3] = __uint_as_float(rb2.z);
                    g0 = __uint_as_float(rc.x); g1 = __uint_as_float(rc.z); q0 = __uint_as_float(rd.x); q1 = __uint_as_float(rd.z);
#ifdef WNV_DBG_MARK
                    miss_pre += !pre_ok; miss_g += !g_ok; miss_q += (sidx != 1 && !q_ok);
#endif
                }
                if (!pre_ok && !rec_recv<2>(prec, tag, pvv, p.status, 0x700u + (unsigned)sidx, lane, p.xcc + 256 + blockIdx.x)) s.flags[0] = 1;
#ifdef WNV_DBG_MARK
                if (MULTI && lane == 0 && t == p.T - 1) { p.xcc[256 + blockIdx.x] = miss_pre; p.xcc[512 + blockIdx.x] = miss_g; p.xcc[768 + blockIdx.x] = miss_q; }
#endif
                const float4 pv = make_float4(pvv[0], pvv[1], pvv[2], pvv[3]);
                *reinterpret_cast<float4*>(s.pre + 4 * lane) = pv;
                if constexpr (zmsg) {                                           // rows 4 lane .. 4 lane + 3 of N_1 h_0, plus pre_1: zin is complete;
                    if constexpr (RP) {                                         // the chain input u_0 comes with it
                        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        if (!rpoll_recv3(p.zmail + (size_t)b * GC + 4 * lane, p.zmail + (size_t)b * GC + 4 * lane + 2, x_in, tag, v, p.status,
                                         0x100u + (unsigned)sidx, lane)) s.flags[0] = 1;
                        WNV_TS(5); WNV_TS(8);
                        const int r = 4 * lane, half = r >> 7, c0 = r & (RC - 1);  // row r = half * 128 + channel; zin is [channel][half]
                        s.zin[2 * c0 + half] = v[0] + pv.x; s.zin[2 * (c0 + 1) + half] = v[1] + pv.y;
                        s.zin[2 * (c0 + 2) + half] = v[2] + pv.z; s.zin[2 * (c0 + 3) + half] = v[3] + pv.w;
                        *reinterpret_cast<float2*>(s.hx + eidx(2 * lane)) = make_float2(v[4], v[5]);
                    }
                } else if (!first_stage) {
                    bool ok;
                    if (sidx == 1) {                                            // h_0 is the chain input of stage 0 (from the head)
                        ok = g_ok || recv128(gsrc, 0x400u + (unsigned)sidx, g0, g1);
                        hv0 = g0; hv1 = g1;
                    } else {
                        ok = (g_ok || recv128(gsrc, 0x480u + (unsigned)sidx, g0, g1)) &&
                             (q_ok || recv128(qsrc, 0x400u + (unsigned)sidx, q0, q1));
                        hv0 = (q0 + g0) * 0.70710678118654752440f;
                        hv1 = (q1 + g1) * 0.70710678118654752440f;
                    }
                    if (!ok) s.flags[0] = 1;
                    WNV_TS(5);
                    *reinterpret_cast<float2*>(s.hb + eidx(2 * lane)) = make_float2(hv0, hv1);
                    hand_on = !last_stage && ok;
                }
            }
            __syncthreads();                                                    // pre_l and h_{l-1} in LDS
            if (wave == 0 && hand_on)                                           // (behind the barrier: the N waves start first)
                st_granule2(p.gmail + (((size_t)b * 2 + par) * S1 + sidx) * RC + 2 * lane, tag, hv0, hv1, fast);
            if (grp == 1 && !zmsg) {                                            // the N waves; the chain waves go on to the chain input
                WNV_TSX(12);
                __builtin_amdgcn_s_setprio(3);                                  // wave 4 shares its SIMD with the polling wave 0
                float a, g;
#if WNV_PHASE2
                if (!first_stage) group_matvec8r(wmn, s.hb + ES * ks, s.pre + chc, s.pre + RC + chc, a, g, gwriter);
#else
                if (!first_stage) group_matvec8(wmn, s.hb + ES * ks, s.pre + chc, s.pre + RC + chc, a, g);
#endif
                else { a = s.pre[chc]; g = s.pre[RC + chc]; }
                if (gwriter) *reinterpret_cast<float2*>(s.zin + 2 * chc) = make_float2(a, g);
                __builtin_amdgcn_s_setprio(0);
                WNV_TS(6);                                                      // zin ready
            } else if (wave == 0 && !zmsg) {
                // ---- the chain: receive X[l][t]  ->  M_l X + zin  ->  gate  ->  send u_l --------------------------------
                float v0 = 0.f, v1 = 0.f;
                bool got;
                if constexpr (RP) got = rpoll_recv2<false>(x_in, tag, v0, v1, p.status, 0x100u + (unsigned)sidx, lane);
                else got = wave_recv2(x_in, tag, v0, v1, p.status, 0x100u + (unsigned)sidx, lane, false, u4v{0, 0, 0, 0});
                if (!got) s.flags[0] = 1;
                WNV_TS(8);                                                      // the poll that carried every tag has returned
                *reinterpret_cast<float2*>(s.hx + eidx(2 * lane)) = make_float2(v0, v1);
            }
            __syncthreads();                                                    // X[l][t] and zin in LDS
            if (grp == 1) WNV_TS(0);                                            // (noted by the waves that are idle here)
            if (grp == 0) {
                float a, g;
#if WNV_PHASE2
                group_matvec8r(wmn, s.hx + ES * ks, s.zin + 2 * chc, s.zin + 2 * chc + 1, a, g, gwriter);
#else
                group_matvec8(wmn, s.hx + ES * ks, s.zin + 2 * chc, s.zin + 2 * chc + 1, a, g);
#endif
                const float u = ring_gate(a, g);                                // modules.py:154
                if (gwriter) {
                    if (!last_stage) st_granule(x_out, tag, u, fast);           // send on: nothing else is on the chain
                    s.us[eidx(chc)] = u;
                }
                WNV_TS(1);                                                      // this chain wave has issued its share of u
            }
            // ---- behind the send -----------------------------------------------------------------------------------
            __syncthreads();                                                    // u_l complete in LDS
            if (grp == 1) WNV_TSX(7);
            float xu[16];
            lds_read16(s.us + ES * ks, xu);
            // conv1x1_out + bias, published for stage l + 2's poller, which adds the residual (see above)
            auto h_phase = [&]() {
                if (!last_stage) {                  // the last layer's residual output is never used (wavenet.py:310-313)
                    const float o0 = dot16p(wo[0], xu), o1 = dot16p(wo[1], xu);
                    const float o = quad_allreduce((hi ? o1 : o0) + dpp_mov<0x141>(hi ? o0 : o1)) + bo_r;
                    if (writer) st_granule(q_out, tag, o, fast);
                }
            };
            // skip 1x1, accumulated stage to stage in the reference's layer order (wavenet.py:312).  The LAST stage publishes its own
            // term alone and the head adds it to the sum through stage S - 2 (same order of additions): the accumulated sum travels
            // as a chain of its own, one hop + one poll per stage, and runs ~0.6 us behind the u's -- the last stage used to wait
            // for it with the whole ring idle.
            auto skip_term = [&](int pp) -> float {                             // this stage's term of skip channels 128 pp + ch
                float m0, m1;
                if (last_stage && pp == (NK > NLDS ? NLDS : 0)) {
                    m0 = dot16p(wo[0], xu); m1 = dot16p(wo[1], xu);
                } else if (pp < NLDS) {
                    m0 = dot16l(s.wsk + (size_t)(8 * pp) * RT + tid, xu); m1 = dot16l(s.wsk + (size_t)(8 * pp + 4) * RT + tid, xu);
                } else {                                                        // streams from the L2 (image layout: coalesced 16-B loads)
                    // (the pass's base is wave-uniform: kept in SGPRs -- as eight per-lane 64-bit pointers it cost the K = 512
                    //  instantiation 22 spilled registers and a scratch reload in front of every load)
                    const unsigned long long sb = reinterpret_cast<unsigned long long>(wsk_g + (size_t)(8 * pp) * RT);
                    const float4* ub = reinterpret_cast<const float4*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)) << 32) |
                                                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb));
                    m0 = dot16s(ub + tid, xu); m1 = dot16s(ub + (size_t)4 * RT + tid, xu);
                }
                return quad_allreduce((hi ? m1 : m0) + dpp_mov<0x141>(hi ? m0 : m1)) + (pp == 0 ? bs_r : s.bsk[RC * pp + ch]);
            };
            auto skip_phase = [&]() {
                bool ok = true;
                if constexpr (NK <= 2) {                                        // term, poll, store per pass (K = 256 measured both ways:
#pragma unroll                                                                  // 414 this way, 400 kSamples/s the other)
                    for (int pp = 0; pp < NK; ++pp) {                           // skip channels 128 pp + ch
                        float m0, m1;
                        if (last_stage && pp == (NK > NLDS ? NLDS : 0)) {
                            m0 = dot16p(wo[0], xu); m1 = dot16p(wo[1], xu);
                        } else {
                            m0 = dot16l(s.wsk + (size_t)(8 * pp) * RT + tid, xu); m1 = dot16l(s.wsk + (size_t)(8 * pp + 4) * RT + tid, xu);
                        }
                        const float mine = quad_allreduce((hi ? m1 : m0) + dpp_mov<0x141>(hi ? m0 : m1)) + (pp == 0 ? bs_r : s.bsk[RC * pp + ch]);
                        float acc = 0.f;
                        if (sidx > 0 && !last_stage && ok)
                            ok = wave_recv<false>(sm_in + RC * pp, writer, tag, acc, p.status, 0x200u + (unsigned)sidx, lane);
                        if (writer && ok) st_granule(sm_out + RC * pp, tag, acc + mine, fast);
                    }
                } else {
                    // K = 512: all four terms first (two of them stream from the L2), then ONE poll loop for the four granules of the sum so
                    // far (first form: term, poll, store per pass -- three extra L2 round trips in every stage; the accumulated sum reached
                    // the head 3.2 us after the last gate, profiles/r03_ring_k512_fine_timeline.txt)
                    float mine[NK], acc[NK];
#pragma unroll
                    for (int pp = 0; pp < NK; ++pp) { mine[pp] = skip_term(pp); acc[pp] = 0.f; }
                    if (WNV_SKIP_DIRECT) {
                        // the terms go straight to the head parts, which add them up in layer order (head_sum_skip_terms): no wait here
                    } else if (sidx > 0 && !last_stage) {
                        unsigned spins = 0;
                        for (;;) {
                            bool hit = true;
                            if (writer) {
                                u64 x[NK];
#pragma unroll
                                for (int pp = 0; pp < NK; ++pp) x[pp] = ld_granule(sm_in + RC * pp);
#pragma unroll
                                for (int pp = 0; pp < NK; ++pp) { acc[pp] = __uint_as_float((unsigned)x[pp]); hit = hit && (unsigned)(x[pp] >> 32) == tag; }
                            }
                            if (__all(hit)) break;
                            if ((++spins & 255u) == 0u) {
                                if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                                if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(p.status, 0u, 0x200u + (unsigned)sidx); ok = false; break; }
                            }
                        }
                    }
                    if (writer && ok) {
#pragma unroll
                        for (int pp = 0; pp < NK; ++pp) st_granule(sm_out + RC * pp, tag, acc[pp] + mine[pp], WNV_SKIP_DIRECT ? fast_head : fast);
                    }
                }
                if (!ok) s.flags[0] = 1;
            };
            // the residual increment is what the next-but-one stage waits for; at the last stage the skip term (the head's input) is
            // the urgent one -- and at the stage before it too: the sum it completes is the other half of the head's input, while its
            // q only feeds the last stage's history push
            if (last_stage || sidx == p.S - 2) { skip_phase(); WNV_TSX(3); h_phase(); WNV_TS(2); }
            else { h_phase(); WNV_TS(2); skip_phase(); WNV_TSX(3); }
            // ---- history push + next step's pre-activations (its barriers fence the LDS vectors for the next step) -------
            __syncthreads();                        // fences the LDS vectors against the next step; makes flags[0] uniform
            if (s.flags[0]) return;                 // a bounded wait gave up somewhere: drain (status holds the code)
            if (wave == 0) {
                // layer inputs to their tap workgroups (history ring, older taps of the next step): records.  This stage knows
                // h_{l-1}[t] (its N input); the last stage also forms h_l[t] of its own layer, which nobody else needs
                auto file = [&](int layer, const float* vec) {
                    const float2 v = *reinterpret_cast<const float2*>(vec + eidx(2 * lane));     // channels 2 lane, 2 lane + 1
                    st_granule2(p.fmail + h_rec(p, b, layer, t) + 2 * lane, tag, v.x, v.y, false);
                };
                if (!first_stage && !zmsg) file(l - 1, s.hb);                   // (zmsg: the head files h_0 itself)
                if (last_stage) {
                    if (first_stage) {
                        file(l, s.hx);                                          // a one-layer model: h_0 is the chain input
                    } else {
                        float q0 = 0.f, q1 = 0.f;
                        if constexpr (zmsg) {                                   // (a two-layer model: this stage never fetched its own input h_0)
                            if (!recv128(p.xmail + ((size_t)b * S1) * RC + 2 * lane, 0x500u + (unsigned)sidx, hv0, hv1)) return;
                        }
                        if (!recv128(p.hmail + (((size_t)b * 2 + par) * S1 + sidx) * RC + 2 * lane, 0x500u + (unsigned)sidx, q0, q1)) return;   // (status holds the code; the others drain at their next wait)
                        *reinterpret_cast<float2*>(s.hh + eidx(2 * lane)) = make_float2((q0 + hv0) * 0.70710678118654752440f, (q1 + hv1) * 0.70710678118654752440f);
                        file(l, s.hh);
                    }
                }
            }
            WNV_TSX(4);
            // timeline slots: 0 X and zin in LDS | 1 u sent (wave 0) | 2 q sent | 3 skip sent | 4 step done | 5 h_{l-1} formed | 6 zin ready |
            // 7 barrier behind u | 8 chain input's poll hit | 9-11 u sent by waves 1-3
            if (wave == 0) WNV_TS_FLUSH(b, t, sidx, 0x013Eu, 0);
            else if (wave < 4) WNV_TS_FLUSH(b, t, sidx, 0x002u, 7 + wave);
            else if (wave == 4) WNV_TS_FLUSH(b, t, sidx, 0x10C1u, 0);            // 12: N waves released
            else if (wave == 5) WNV_TS_FLUSH(b, t, sidx, 0x0040u, 7);          // 13: zin ready as wave 5 saw it
        }
    }
}

// ---- SPLIT RINGS (round 3): one layer = TWO CUs ------------------------------------------------------------------------------
// A 256 x 128 mat-vec phase costs a CU ~0.29 us however its waves are arranged: four SIMDs issue 128 fp32 FMAs per clock, the
// phase is VALU-bound (profiles/r03_ring_c2_fine_timeline.txt), and a layer has one such phase on the chain (M_l u) and one on the
// loop u_l -> h_{l+1} -> N_{l+2} h_{l+1} -> u_{l+2} that runs neck and neck with it.  With 8 utterances per GPU the chip has room for
// TWO CUs per layer (4 rings x 2 utterances; a ring spans two XCDs, two cross-XCD hops per step): half h of stage l owns gate
// channels [64 h, 64 h + 64) -- 128 rows of M_l (waves 0-3) and of N_l (waves 4-7), 32 packed FMAs per lane instead of 64 -- and
// the K-half [64 h, 64 h + 64) of conv1x1_out and conv1x1_skip, whose inputs are the u channels it has just gated itself:
//   * both halves poll the whole chain input (mailboxes are multi-reader) and each publishes its 64 channels of u_l;
//   * conv1x1_out / conv1x1_skip become PARTIAL sums over the half's own 64 inputs (no wait for the sibling): Q and the skip
//     chain carry two partial vectors per slot (bias in half 0); the poller of stage l + 2 forms h_{l+1} = sqrt(.5) (qA + qB + h_l),
//     the head adds the two skip chains.  Re-association only (<= 1e-6 on the head outputs; parity tests).
// Needs the head to evaluate layer 0 (L0) and 128 skip channels.  Thread mapping of a wave group: lane (og = gtid >> 3, ks = gtid & 7):
// eight lanes split K for the four rows {tanh, sigmoid} x local channels 2og, 2og + 1; slots 0-1 = the channel the lane keeps
// (2og + (ks >= 4)), slots 2-3 = the other one: row_half_mirror adds the partner's slots 2-3 to 0-1, a quad all-reduce finishes.
__device__ __forceinline__ void group_matvec4(const f2 (&w)[4][8], const float* xslice, const float* za, const float* zb, float& a, float& g) {
    float2 z = make_float2(*za, *zb);
    float x[16];
    lds_read16(xslice, x);
    f2 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(w[s][k], f2{x[2 * k], x[2 * k + 1]}, acc[s]);
    asm volatile("" : "+v"(z.x), "+v"(z.y));
    const float q0 = acc[0].x + acc[0].y, q1 = acc[1].x + acc[1].y, q2 = acc[2].x + acc[2].y, q3 = acc[3].x + acc[3].y;
    a = quad_allreduce(dpp_fold<0x141>(q0, q2)) + z.x;
    g = quad_allreduce(dpp_fold<0x141>(q1, q3)) + z.y;
}
// ONE wave: two single granules + one pair of granules per lane in one round trip (stage 1 behind the head: its two rows of
// N_1 h_0 and two values of u_0)
__device__ __forceinline__ bool rpoll_recv_zx(const u64* ga, const u64* gb, const u64* gx, unsigned tag, float (&v)[4], unsigned int* status, unsigned code, int lane) {
    unsigned spins = 0;
    for (;;) {
        rpoll8_issue<0>(ga); rpoll8_issue<1>(gb); rpoll16_issue<2>(gx);
        unsigned va, ta, vb, tb;
        rpoll8_take<0, 2>(va, ta);
        rpoll8_take<1, 1>(vb, tb);
        const u4v c = rpoll16_take<2, 0>();
        if (__all(ta == tag && tb == tag && c.y == tag && c.w == tag)) {
            v[0] = __uint_as_float(va); v[1] = __uint_as_float(vb); v[2] = __uint_as_float(c.x); v[3] = __uint_as_float(c.z);
            return true;
        }
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}

template <bool ZMSG>
__device__ __attribute__((always_inline)) void run_stage_split(const RingParams& p, int ring, int sidx, int half, float* smem) {
    const StageLds s = carve_stage(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid >> 8, gtid = tid & (GT - 1);
    const int ks = gtid & 7, og = gtid >> 3;
    const int lc = 2 * og + (ks >= 4 ? 1 : 0);                  // local channel whose (tanh, sigmoid) rows end up in this lane
    const int chc = 64 * half + lc;                             // ... as a channel of the layer
    const bool gwriter = (ks & 3) == 0;
    const int r = tid >> 2, kq = tid & 3;                       // behind the send: four lanes split the half's 64 inputs of output row r
    const bool qwriter = kq == 0;
    const int l = sidx;
    const bool last_stage = sidx == p.S - 1;
    const int S1 = p.S + 1;
    constexpr float RS = 0.70710678118654752440f;
    WNV_TS_DECL;

    f2 wmn[4][8], wo[8], ws[8];
#pragma unroll
    for (int slot = 0; slot < 4; ++slot)
        load_image8g((grp == 0 ? p.w2s : p.wns) + (((size_t)l * 2 + half) * 4 + slot) * 4 * GT * 4, gtid, wmn[slot]);
    load_image8(p.wos + ((size_t)l * 2 + half) * 4 * RT * 4, tid, wo);
    load_image8(p.wss + ((size_t)l * 2 + half) * 4 * RT * 4, tid, ws);
    const float bo_r = half == 0 ? p.bo[(size_t)l * RC + r] : 0.f;          // the biases ride on half 0's partial sums
    const float bs_r = half == 0 ? p.bskip[(size_t)l * p.Kp + r] : 0.f;
    if (tid == 0) s.flags[0] = 0;
    // who reads what this half sends: u_l, the handed-on layer input and the skip chain -> stage l + 1 (the head behind the last
    // stage; the stage before the last completes the skip sum for the head as well); the residual partials -> stage l + 2
    if (tid == 0) {
        int near[4], far[2], nn = 0, nf = 0;
        if (!last_stage) { near[nn++] = split_block(p, ring, sidx + 1, 0); near[nn++] = split_block(p, ring, sidx + 1, 1); }
        if (sidx >= p.S - 2) near[nn++] = split_block(p, ring, p.S, 0);
        if (sidx + 2 < p.S) { far[nf++] = split_block(p, ring, sidx + 2, 0); far[nf++] = split_block(p, ring, sidx + 2, 1); }
        else if (!last_stage) { far[nf++] = split_block(p, ring, sidx + 1, 0); }    // the last stage reads them for its history push
        int* rl = s.flags + 4;
        rl[0] = nn; rl[1] = nf;
        for (int k = 0; k < nn; ++k) rl[2 + k] = near[k];
        for (int k = 0; k < nf; ++k) rl[6 + k] = far[k];
    }
    __syncthreads();
    const bool fast = same_xcd_list(p, s.flags + 6, s.flags[4], s.flags + 1);
    const bool fast_far = same_xcd_list(p, s.flags + 10, s.flags[5], s.flags + 2);

    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        const int par = t & 1;
        for (int j = 0; j < p.upr; ++j) {
            const int b = ring + j * p.n_rings;
            if (b >= p.B) continue;
            const u64* x_in = p.xmail + ((size_t)b * S1 + sidx) * RC + 2 * lane;      // wave 0: two granules per lane
            u64* x_out = p.xmail + ((size_t)b * S1 + sidx + 1) * RC + chc;            // chain waves
            u64* q_out = p.hmail + ((((size_t)b * 2 + par) * S1 + sidx + 1) * 2 + half) * RC + r;
            const u64* sm_in = p.smail + (((size_t)b * S1 + sidx) * 2 + half) * p.Kp + r;
            u64* sm_out = p.smail + (((size_t)b * S1 + sidx + 1) * 2 + half) * p.Kp + r;
            asm volatile("" : "+v"(x_in), "+v"(x_out), "+v"(q_out), "+v"(sm_in), "+v"(sm_out));
            float hv0 = 0.f, hv1 = 0.f;                                         // wave 0: h_{l-1}[t], channels 2 lane, 2 lane + 1
            bool hand_on = false;
            auto recv128 = [&](const u64* g2, unsigned code, float& v0, float& v1) {
                return rpoll_recv2<false>(g2, tag, v0, v1, p.status, code, lane);
            };
            if (wave == 0) {
                float pvv[4] = {0.f, 0.f, 0.f, 0.f};
                if (!rec_recv<2>(p.pmail + pre_rec(p, b, l, t), tag, pvv, p.status, 0x700u + (unsigned)sidx, lane)) s.flags[0] = 1;
                *reinterpret_cast<float4*>(s.pre + 4 * lane) = make_float4(pvv[0], pvv[1], pvv[2], pvv[3]);
                if constexpr (ZMSG) {                                           // lane L: rows 64 half + L (tanh) and 128 + 64 half + L (sigmoid)
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    const u64* zt = p.zmail + (size_t)b * GC + 64 * half + lane;
                    if (!rpoll_recv_zx(zt, zt + RC, x_in, tag, v, p.status, 0x100u + (unsigned)sidx, lane)) s.flags[0] = 1;
                    WNV_TS(5); WNV_TS(8);
                    *reinterpret_cast<float2*>(s.zin + 2 * lane) = make_float2(v[0] + s.pre[64 * half + lane], v[1] + s.pre[RC + 64 * half + lane]);
                    *reinterpret_cast<float2*>(s.hx + eidx(2 * lane)) = make_float2(v[2], v[3]);
                } else {
                    // h_{l-1} = sqrt(.5) (qA + qB + h_{l-2}): the two partial residual terms of stage l - 2 and the layer input stage
                    // l - 1 handed on (stage 2: h_0 straight from the head, whose q_0 is whole: its half-1 vector is zeros)
                    const size_t slot = (((size_t)b * 2 + par) * S1 + sidx - 1);
                    const u64* gsrc = sidx == 2 ? p.xmail + ((size_t)b * S1) * RC + 2 * lane : p.gmail + slot * RC + 2 * lane;
                    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    // (h_{l-2} is there long before the partials: fetched alone, then the two partials travel together)
                    const bool ok = recv128(gsrc, 0x480u + (unsigned)sidx, v[0], v[1]) &&
                                    rpoll_recv2x2(p.hmail + (slot * 2) * RC + 2 * lane, p.hmail + (slot * 2 + 1) * RC + 2 * lane, tag, v[2], v[3], v[4], v[5], p.status,
                                                  0x400u + (unsigned)sidx, lane);
                    hv0 = ((v[2] + v[4]) + v[0]) * RS;
                    hv1 = ((v[3] + v[5]) + v[1]) * RS;
                    if (!ok) s.flags[0] = 1;
                    WNV_TS(5);
                    *reinterpret_cast<float2*>(s.hb + eidx(2 * lane)) = make_float2(hv0, hv1);
                    hand_on = half == 0 && !last_stage && ok;
                }
            }
            __syncthreads();                                                    // pre_l and h_{l-1} in LDS
            if (wave == 0 && hand_on)
                st_granule2(p.gmail + (((size_t)b * 2 + par) * S1 + sidx) * RC + 2 * lane, tag, hv0, hv1, fast);
            if (grp == 1 && !ZMSG) {
                WNV_TSX(12);
                float a, g;
                group_matvec4(wmn, s.hb + ES * ks, s.pre + chc, s.pre + RC + chc, a, g);
                if (gwriter) *reinterpret_cast<float2*>(s.zin + 2 * lc) = make_float2(a, g);
                WNV_TS(6);
            } else if (wave == 0 && !ZMSG) {
                float v0 = 0.f, v1 = 0.f;
                if (!recv128(x_in, 0x100u + (unsigned)sidx, v0, v1)) s.flags[0] = 1;
                WNV_TS(8);
                *reinterpret_cast<float2*>(s.hx + eidx(2 * lane)) = make_float2(v0, v1);
            }
            __syncthreads();                                                    // X[l][t] and zin in LDS
            if (grp == 1) WNV_TS(0);
            if (grp == 0) {
                float a, g;
                group_matvec4(wmn, s.hx + ES * ks, s.zin + 2 * lc, s.zin + 2 * lc + 1, a, g);
                const float u = ring_gate(a, g);                                // modules.py:154
                if (gwriter) {
                    if (!last_stage) st_granule(x_out, tag, u, fast);
                    s.us[eidx(lc)] = u;
                }
                WNV_TS(1);
            }
            __syncthreads();                                                    // this half's 64 channels of u_l in LDS
            if (grp == 1) WNV_TSX(7);
            float xu[16];
            lds_read16(s.us + ES * kq, xu);
            auto h_phase = [&]() {
                if (!last_stage) {
                    const float o = quad_allreduce(dot16p(wo, xu)) + bo_r;
                    if (qwriter) st_granule(q_out, tag, o, fast_far);
                }
            };
            auto skip_phase = [&]() {
                const float mine = quad_allreduce(dot16p(ws, xu)) + bs_r;
                float acc = 0.f;
                bool ok = true;
                // (half 0's chain starts at the head, half 1's here at stage 1; the last stage publishes its own term alone)
                if (!last_stage && (sidx > 1 || half == 0)) ok = wave_recv<false>(sm_in, qwriter, tag, acc, p.status, 0x200u + (unsigned)sidx, lane);
                if (qwriter && ok) st_granule(sm_out, tag, acc + mine, fast);
                if (!ok) s.flags[0] = 1;
            };
            if (sidx >= p.S - 2) { skip_phase(); WNV_TSX(3); h_phase(); WNV_TS(2); }
            else { h_phase(); WNV_TS(2); skip_phase(); WNV_TSX(3); }
            __syncthreads();
            if (s.flags[0]) return;
            if (wave == 0 && half == 0) {                                       // history of layer l - 1 (and of the last layer): half 0 files it
                auto file = [&](int layer, const float* vec) {
                    const float2 v = *reinterpret_cast<const float2*>(vec + eidx(2 * lane));     // channels 2 lane, 2 lane + 1
                    st_granule2(p.fmail + h_rec(p, b, layer, t) + 2 * lane, tag, v.x, v.y, false);
                };
                if (!ZMSG) file(l - 1, s.hb);
                if (last_stage) {
                    if constexpr (ZMSG) {
                        if (!recv128(p.xmail + ((size_t)b * S1) * RC + 2 * lane, 0x500u + (unsigned)sidx, hv0, hv1)) return;
                    }
                    const size_t slot = (((size_t)b * 2 + par) * S1 + sidx);
                    float qa0 = 0.f, qa1 = 0.f, qb0 = 0.f, qb1 = 0.f;
                    if (!(recv128(p.hmail + (slot * 2) * RC + 2 * lane, 0x500u + (unsigned)sidx, qa0, qa1) &&
                          recv128(p.hmail + (slot * 2 + 1) * RC + 2 * lane, 0x500u + (unsigned)sidx, qb0, qb1))) return;
                    *reinterpret_cast<float2*>(s.hh + eidx(2 * lane)) = make_float2(((qa0 + qb0) + hv0) * RS, ((qa1 + qb1) + hv1) * RS);
                    file(l, s.hh);
                }
            }
            WNV_TSX(4);
            if (half == 0) {
                if (wave == 0) WNV_TS_FLUSH(b, t, sidx, 0x013Eu, 0);
                else if (wave < 4) WNV_TS_FLUSH(b, t, sidx, 0x002u, 7 + wave);
                else if (wave == 4) WNV_TS_FLUSH(b, t, sidx, 0x10C1u, 0);
                else if (wave == 5) WNV_TS_FLUSH(b, t, sidx, 0x0040u, 7);
            }
        }
    }
}

// ---- head -----------------------------------------------------------------------------------------------------------
// skip sum -> ReLU -> 1x1 (K x K) -> ReLU -> 1x1 (O x K) -> sampler -> first_conv of the next step (wavenet.py:313-336).
// With K = 128 NK skip channels the K x K matrix does not fit one CU, so the head is NK workgroups ("parts"): part j owns
// hidden units [128 j, 128 j + 128): it reads the whole skip vector, computes its hidden slice (W1 rows in VGPRs, 32 NK
// floats per thread) and the partial outputs W2[:, 128 j ..] . hidden_j; parts j > 0 send their partials to part 0, which adds
// them in part order, samples and feeds the ring.  NK = 1 is the single-workgroup head.
struct HeadLds {
    float* pre0;   // [256] pre_0[t+1] from layer 0's tap workgroup (head_l0)
    float* us0;    // u_0[t+1], eight padded K-slices (head_l0: input of conv1x1_out / conv1x1_skip of layer 0)
    float* vs;     // strided relu(skip * scale), 4 NK K-quarters
    float* hid;    // strided hidden slice
    float* obuf;   // [256] head output
    float* vbuf;   // [48]  mixture logit + Gumbel noise, padded with -inf
    int* flags;
};
__device__ __forceinline__ HeadLds carve_head(float* smem, int NK) {
    HeadLds s;
    s.pre0 = smem; s.us0 = smem + GC;
    s.vs = s.us0 + 8 * ES; s.hid = s.vs + 4 * NK * QS; s.obuf = s.hid + 4 * QS; s.vbuf = s.obuf + 256;
    s.flags = reinterpret_cast<int*>(s.vbuf + 48);
    return s;
}
__host__ __device__ constexpr size_t head_lds_floats(int NK) { return (size_t)GC + 8 * ES + (size_t)(4 * NK + 4) * QS + 256 + 48 + 16; }

// noise value `idx` of (t, b): from the tape (rng = "replay") or the in-kernel Philox stream
// (a STREAMED tape is host memory the CPU is still writing: its values are read with system-scope loads that bypass the vector L1 and
//  the L2 -- a line fetched for the last published step also holds the start of the next, unpublished one -- behind wait_noise's
//  acquire; a tape in device memory is complete before the launch and is read with plain loads)
// (tl, ub: the coordinates of the in-kernel stream -- step and utterance index of the call, or, with packed slots, the step within the
//  utterance and its id in the job)
__device__ __forceinline__ float head_noise(const RingParams& p, int t, int b, int idx, int kind, int tl, int ub) {
    if (!p.noise) return wnv_noise_gen(p.seed, tl, ub, idx, kind);
    const float* src = p.noise + ((size_t)t * p.noise_B + p.b0 + b) * p.nz + idx;
    if (p.noise_ready) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    return *src;
}
// STREAMED TAPE: the host is still drawing the tape while the kernel runs (coherent host memory; wnv_generate_args.noise_ready).
// Every wave that is about to read step t waits until the counter has passed it -- one PCIe read per tape chunk, the value seen is
// kept in `seen`.  Bounded like every wait (the host draws ~1e5 steps per second: the budget is seconds).  Returns false on abort
// -- the caller must not read the step then.  The counter is loaded with ACQUIRE semantics at system scope: the tape reads that
// follow are ordered behind it and cannot be served from a line cached before the host published the step.
__device__ __forceinline__ bool wait_noise(const RingParams& p, int t, unsigned& seen) {
    if (!p.noise_ready || (unsigned)t < seen) return true;
    unsigned spins = 0;
    for (;;) {
        seen = __hip_atomic_load(p.noise_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)t < seen) return true;
        if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if ((threadIdx.x & 63) == 0) atomicCAS(p.status, 0u, 0x800u); return false; }
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// the slice of the output MLP one head part owns: W1 rows [128 part, +128) x K and W2[:, 128 part .. +128) (NW2 row images: rows i,
// and rows 128 + i for one-hot models)
template <int NK, int NW2> struct HeadSlice {
    f2 wh1[NK][16];
    f2 wh2[NW2][16];
    float bh1;
    __device__ __forceinline__ void load(const RingParams& p, int part, int tid, int i) {
#pragma unroll
        for (int blk = 0; blk < NK; ++blk) load_image16(p.wh1img + ((size_t)part * NK + blk) * 8 * RT * 4, tid, wh1[blk]);
#pragma unroll
        for (int r = 0; r < NW2; ++r) load_image16(p.wh2img + ((size_t)part * 2 + r) * 8 * RT * 4, tid, wh2[r]);
        bh1 = p.bh1[RC * part + i];
    }
};

// skip sum of (b, t) -> s.vs (all parts): the sum through stage S - 2 (slot S - 1) + the last stage's own term (slot S), the
// reference's order of additions (wavenet.py:312);  returns false on abort
// K = 512 (round 5, WNV_SKIP_DIRECT): the skip sum no longer travels stage to stage.  With four skip passes per stage -- two of them
// streamed from the L2 -- the accumulated sum ran ~2.5 us behind the gates and the head waited 1.6 us for it after the last gate
// (profiles/r05_ring_k512_fine_timeline.txt), while every stage spent those microseconds polling its predecessor's sum.  Now every stage
// hands ITS OWN term to the head parts (slot l + 1 of the same mailbox) and the head parts, idle while the chain runs, add the terms up in
// layer order -- the reference's order of additions (wavenet.py:312) -- four slots per round trip, as they arrive.
__device__ __forceinline__ bool head_sum_skip_terms(const RingParams& p, int b, unsigned tag, float* vs, int tid, int lane) {
    const u64* slot = p.smail + ((size_t)b * (p.S + 1) + 1) * p.Kp + tid;       // stage 0's term; stage l's is Kp granules further per layer
    float acc = 0.f;
    for (int l0 = 0; l0 < p.S; l0 += 4) {
        const int n = min(4, p.S - l0);
        float v[4];
        unsigned spins = 0;
        for (;;) {
            u64 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = k < n ? ld_granule(slot + (size_t)(l0 + k) * p.Kp) : ((u64)tag << 32);
            bool hit = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = __uint_as_float((unsigned)x[k]); hit = hit && (unsigned)(x[k] >> 32) == tag; }
            if (__all(hit)) break;
            if ((++spins & 255u) == 0u) {
                if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
                if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(p.status, 0u, 0x300u); return false; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) acc += v[k];
    }
    vs[qidx(tid)] = fmaxf(acc * p.skip_scale, 0.f);                             // wavenet.py:313-316
    return true;
}

template <int NK>
__device__ __forceinline__ bool head_recv_skip(const RingParams& p, int b, unsigned tag, float* vs, int tid, int lane, int wave) {
    if constexpr (NK == 4 && WNV_SKIP_DIRECT != 0) return head_sum_skip_terms(p, b, tag, vs, tid, lane);
    bool ok = true;
    if (wave < 2 * NK) {
        const u64* own = p.smail + ((size_t)b * (p.S + 1) + p.S) * p.Kp + tid;
        float acc = 0.f, v = 0.f;
        if constexpr (NK <= 4) {
            if (p.S > 1) ok = rpoll_recv<0>(own - p.Kp, true, tag, acc, p.status, 0x300u, lane);
            ok = ok && rpoll_recv<0>(own, true, tag, v, p.status, 0x300u, lane);
        } else {
            if (p.S > 1) ok = wave_recv<false>(own - p.Kp, true, tag, acc, p.status, 0x300u, lane);
            ok = ok && wave_recv<false>(own, true, tag, v, p.status, 0x300u, lane);
        }
        vs[qidx(tid)] = fmaxf((acc + v) * p.skip_scale, 0.f);                   // wavenet.py:313-316
    }
    return ok;
}
// split rings: two partial chains (halves) per slot: (accA + accB) + ownA + ownB, all four granules of a lane in two round trips
__device__ __forceinline__ bool head_recv_skip_split(const RingParams& p, int b, unsigned tag, float* vs, int tid, int lane, int wave) {
    bool ok = true;
    if (wave < 2) {
        const u64* own = p.smail + (((size_t)b * (p.S + 1) + p.S) * 2) * p.Kp + tid;        // the last stage's terms: halves 0, 1
        const u64* acc = own - 2 * p.Kp;                                                     // the sums through stage S - 2
        float a0 = 0.f, a1 = 0.f, o0 = 0.f, o1 = 0.f;
        ok = rpoll_recv<0>(acc, true, tag, a0, p.status, 0x300u, lane) && rpoll_recv<0>(acc + p.Kp, true, tag, a1, p.status, 0x300u, lane) &&
             rpoll_recv<0>(own, true, tag, o0, p.status, 0x300u, lane) && rpoll_recv<0>(own + p.Kp, true, tag, o1, p.status, 0x300u, lane);
        vs[qidx(tid)] = fmaxf((((a0 + a1) + o0) + o1) * p.skip_scale, 0.f);               // wavenet.py:313-316
    }
    return ok;
}
// hidden slice (wavenet.py:317-318) into s.hid; needs a barrier before and after
template <int NK, int NW2>
__device__ __forceinline__ void head_hidden(const HeadSlice<NK, NW2>& w, const float* vs, float* hid, int q, int i) {
    float x[32];
    float acc = 0.f;
#pragma unroll
    for (int blk = 0; blk < NK; ++blk) {
        lds_read32(vs + QS * (4 * blk + q), x);
        acc += dot32p(w.wh1[blk], x);
    }
    const float h1 = fmaxf(quad_allreduce(acc) + w.bh1, 0.f);
    if (q == 0) hid[qidx(i)] = h1;
}

// head parts j > 0: hidden slice + partial outputs, sent to part 0
template <int NK, int NW2>
__device__ __attribute__((always_inline)) void run_head_part(const RingParams& p, int ring, int part, float* smem) {
    const HeadLds s = carve_head(smem, NK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, i = wave * 16 + (lane >> 2);
    HeadSlice<NK, NW2> w;
    w.load(p, part, tid, i);
    if (tid == 0) s.flags[0] = 0;
    const bool fast = same_xcd_as(p, ring + p.S * p.rstride, 1, ring + p.S * p.rstride, s.flags + 1);   // read by part 0
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int j = 0; j < p.upr; ++j) {
            const int b = ring + j * p.n_rings;
            if (b >= p.B) continue;
            if (!head_recv_skip<NK>(p, b, tag, s.vs, tid, lane, wave)) s.flags[0] = 1;
            __syncthreads();
            head_hidden<NK, NW2>(w, s.vs, s.hid, q, i);
            __syncthreads();
            float x[32];
            lds_read32(s.hid + QS * q, x);
            u64* om = p.omail + ((size_t)b * p.NH + part) * p.Op;
#pragma unroll
            for (int r = 0; r < NW2; ++r) {
                const float o = quad_allreduce(dot32p(w.wh2[r], x));              // rows i (and 128 + i) of W2, columns of this part
                if (q == 0 && RC * r + i < p.O) st_granule(om + RC * r + i, tag, o, fast);
            }
            if (s.flags[0]) return;                 // uniform: written before the barriers above
        }
    }
}

// part 0 collects the partial outputs of parts 1 .. NH-1 (lanes q == 0 own output rows row + RC r, r < NR); every granule a lane
// waits for is polled in the SAME round trip (round 2 polled part after part and row after row: up to three / two extra L2 round
// trips behind the arrival); the sums are formed in part order
template <int NR>
__device__ __forceinline__ bool head_collect(const RingParams& p, int b, unsigned tag, int row, const bool (&active)[NR], float (&o)[NR], int lane) {
    constexpr int MAXP = 3;                                                     // NH <= 4
    const u64* g = p.omail + ((size_t)b * p.NH + 1) * p.Op + row;
    const int np = p.NH - 1;
    float v[NR][MAXP];
    unsigned spins = 0;
    for (;;) {
        u64 x[NR][MAXP];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                x[r][k] = (active[r] && k < np) ? ld_granule(g + (size_t)k * p.Op + RC * r) : ((u64)tag << 32);
        bool ok = true;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int k = 0; k < MAXP; ++k) { v[r][k] = __uint_as_float((unsigned)x[r][k]); ok = ok && (unsigned)(x[r][k] >> 32) == tag; }
        if (__all(ok)) break;
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(p.status, 0u, 0x381u); return false; }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < MAXP; ++k) if (k < np) o[r] += v[r][k];
    return true;
}

template <int NK, bool L0, bool SPLIT, bool PACKED>
__device__ __attribute__((always_inline)) void run_head(const RingParams& p, int ring, float* smem) {
    WNV_TS_DECL;
    const HeadLds s = carve_head(smem, NK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, i = wave * 16 + (lane >> 2);
    const int S1 = p.S + 1;
    HeadSlice<NK, 1> w;
    w.load(p, 0, tid, i);
    const float bh2 = p.bh2[i];
    float wf = 0.f, bf = 0.f;
    if (tid < RC) { wf = p.wfirst[tid]; bf = p.bfirst[tid]; }
    // output distribution (mixture.py:118-156 / :221-270): which head outputs are mean / log-scale
    const bool single = p.dist == 2 && p.O <= 3;
    const int nmix = single ? 0 : p.O / 3;
    const int o_mean = single ? (p.O == 2 ? 0 : 1) : nmix, o_ls = single ? (p.O == 2 ? 1 : 2) : 2 * nmix;
    const int nchunk = (nmix + 3) >> 2;
    if (tid < 48) s.vbuf[tid] = -INFINITY;
    if (tid == 0) s.flags[0] = 0;
    // ---- LAYER 0 IN THE HEAD (p.head_l0; scalar-input models with 128 skip channels).  h_0 = w_first x + b_first is affine in the
    //      sample, so layer 0's pre-activation is  z_0 = (W_cur,0 w_first) x + W_cur,0 b_first + pre_0[t+1]: two FMAs per channel,
    //      no mat-vec.  The head therefore gates u_0 itself and sends it straight to stage 1 -- one hop and one mat-vec phase less
    //      per step (~0.6 us) -- followed by h_0 (stage 1's N input) and, behind the sends, layer 0's conv1x1_out / conv1x1_skip
    //      terms (the stage role's thread mapping and row images).  Position 0 of the ring stays empty.  Stage 1's N_1 h_0 is affine in
    //      the sample too and is sent along (Z mailbox): h_0 is no longer known a layer ahead of u_0.
    constexpr bool l0 = L0;
    const int ks = tid & 7, og = tid >> 3;
    const bool hi = ks >= 4, writer = (ks & 3) == 0;
    const int ch = 2 * og + (hi ? 1 : 0);
    f2 wo0[2][8], ws0[2][8];
    float a_t = 0.f, a_s = 0.f, c_t = 0.f, c_s = 0.f, bo0 = 0.f, bs0 = 0.f, n_t = 0.f, n_s = 0.f, d_t = 0.f, d_s = 0.f;
    if (l0) {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            load_image8(p.woimg + (size_t)row * 4 * RT * 4, tid, wo0[row]);
            load_image8(p.wsimg + (size_t)row * 4 * RT * 4, tid, ws0[row]);
        }
        bo0 = p.bo[ch]; bs0 = p.bskip[ch];
        if (tid < RC) {
            a_t = p.l0vec[tid]; a_s = p.l0vec[RC + tid]; c_t = p.l0vec[GC + tid]; c_s = p.l0vec[GC + RC + tid];
            n_t = p.l0vec[2 * GC + tid]; n_s = p.l0vec[2 * GC + RC + tid]; d_t = p.l0vec[3 * GC + tid]; d_s = p.l0vec[3 * GC + RC + tid];
        }
    }
    // readers of what the head sends: h_0 -> stages 0 and 1; with layer 0 here: u_0, h_0 -> stage 1, layer 0's residual term -> stage 2
    bool fast;
    if constexpr (SPLIT) {                                                      // both halves of stages 1 and 2
        if (tid == 0) { int* rl = s.flags + 4; rl[0] = split_block(p, ring, 1, 0); rl[1] = split_block(p, ring, 1, 1); rl[2] = split_block(p, ring, 2, 0); rl[3] = split_block(p, ring, 2, 1); }
        __syncthreads();
        fast = same_xcd_list(p, s.flags + 4, 4, s.flags + 1);
    } else {
        fast = l0 ? same_xcd_as(p, ring + p.rstride, 1, ring + (p.S > 2 ? 2 : 1) * p.rstride, s.flags + 1)
                  : same_xcd_as(p, ring, 1, ring + (p.S > 1 ? p.rstride : 0), s.flags + 1);
    }
    // wave 2 fetches pre_0 of the step whose input is about to be made (tag tg) into LDS; a barrier follows at the call sites
    auto fetch_pre0 = [&](int b, unsigned tg) {
        if (l0 && wave == 2) {
            float pvv[4] = {0.f, 0.f, 0.f, 0.f};
            if (!WNV_EXP_NOPRE && !rec_recv<2>(p.pmail + pre_rec(p, b, 0, (int)(tg - p.tag_base - 1u)), tg, pvv, p.status, 0x700u, lane, p.xcc + 256 + blockIdx.x)) s.flags[0] = 1;
            *reinterpret_cast<float4*>(s.pre0 + 4 * lane) = make_float4(pvv[0], pvv[1], pvv[2], pvv[3]);
        }
    };
    // the input of step (tag tg, parity parn) from the sample xs (waves 0-1): u_0 first -- it is what the chain waits for
    // (pt, ps: c + pre_0 of the lane's two rows, read from LDS ahead of time; every address comes pinned in registers: between the
    //  sample and the sends there is no address arithmetic and no reload of a spilled kernel argument)
    struct FeedAddr { u64 *x0, *x1, *z, *f, *q, *sk; };
    auto feed_addr = [&](int b, int parn) {
        FeedAddr a;
        a.x0 = p.xmail + ((size_t)b * S1) * RC + tid;
        a.x1 = a.x0 + RC;
        a.z = p.zmail + (size_t)b * GC + tid;
        a.f = p.fmail + h_rec(p, b, 0, parn) + tid;                     // (parn IS the parity of the step the input is made for)
        // (split rings: two partial vectors per slot; layer 0's terms are whole and go to half 0, half 1 of the residual slot gets zeros)
        a.q = p.hmail + ((((size_t)b * 2 + parn) * S1 + 1) * (SPLIT ? 2 : 1)) * RC + ch;
        a.sk = p.smail + (((size_t)b * S1 + 1) * (SPLIT ? 2 : 1)) * p.Kp + ch;
        asm volatile("" : "+v"(a.x0), "+v"(a.x1), "+v"(a.z), "+v"(a.f), "+v"(a.q), "+v"(a.sk));
        return a;
    };
    auto feed = [&](int b, unsigned tg, const FeedAddr& ad, float xs, float pt, float ps) {
        if (wave < 2) {
            const float h0 = fmaf(wf, xs, bf);
            if (l0) {
                const float u = ring_gate(fmaf(a_t, xs, pt), fmaf(a_s, xs, ps));
                st_granule(ad.x1, tg, u, fast);
                // N_1 h_0 for stage 1 (it would otherwise run its N mat-vec ON the chain: h_0 is not known a layer early any more)
                st_granule(ad.z, tg, fmaf(n_t, xs, d_t), fast);
                st_granule(ad.z + RC, tg, fmaf(n_s, xs, d_s), fast);
                s.us0[eidx(tid)] = u;
            }
            st_granule(ad.x0, tg, h0, fast);
            WNV_TS(0);
            // h_0 for layer 0's tap workgroup (history, older taps of the next step): filed HERE, at the start of the step -- the
            // tap workgroups serve all rings in one pass, and the ring that runs ahead of the others waits for pre_0 first
            if (l0) st_granule(ad.f, tg, h0, false);                            // (a tagged granule, written through: no drain, no flag)
        }
        if (l0) {                                                               // behind the sends: layer 0's residual and skip terms
            __syncthreads();
            float xu[16];
            lds_read16(s.us0 + ES * ks, xu);
            const float o0 = dot16p(wo0[0], xu), o1 = dot16p(wo0[1], xu);
            const float o = quad_allreduce((hi ? o1 : o0) + dpp_mov<0x141>(hi ? o0 : o1)) + bo0;
            if (writer) {
                st_granule(ad.q, tg, o, fast);                                  // (stage 2 is waiting for this one)
                if constexpr (SPLIT) st_granule(ad.q + RC, tg, 0.f, fast);
            }
            const float m0 = dot16p(ws0[0], xu), m1 = dot16p(ws0[1], xu);
            const float m = quad_allreduce((hi ? m1 : m0) + dpp_mov<0x141>(hi ? m0 : m1)) + bs0;
            if (writer) st_granule(ad.sk, tg, m, fast);
        }
    };

    // ---- prologue: the input of step 0 (wavenet.py:283-289, :297-308) ----------------------------------------
    for (int j = 0; j < p.upr; ++j) {
        const int b = ring + j * p.n_rings;
        if (b >= p.B) continue;
        fetch_pre0(b, p.tag_base + 1u);
        __syncthreads();
        const float xs = p.Tt > 0 ? p.teacher[(size_t)b * p.Tt] : (p.initial ? p.initial[b] : 0.f);
        feed(b, p.tag_base + 1u, feed_addr(b, 0), xs, l0 && tid < RC ? c_t + s.pre0[tid] : 0.f, l0 && tid < RC ? c_s + s.pre0[RC + tid] : 0.f);
    }

    // packed slots: the maps' entries of the iteration about to run -- (utterance j, step t) -> pf_*; called with the CURRENT (j, t) it
    // fetches the next iteration's: (j + 1, t) or (0, t + 1)
    int pf_start = 0, pf_uid = 0, pf_next = -1;
    const int nj_live = min(p.upr, (p.B - ring + p.n_rings - 1) / p.n_rings);
    auto seg_fetch = [&](int jn, int tn) {
        if constexpr (PACKED) {
            if (tn < p.T) {
                const int* base = p.seg_start + (size_t)(ring + jn * p.n_rings) * p.T + tn;
                pf_start = uniform_ld(base);
                pf_uid = uniform_ld(p.seg_uid + (size_t)(ring + jn * p.n_rings) * p.T + tn);
                pf_next = tn + 1 < p.T ? uniform_ld(base + 1) : -1;
            }
        }
    };
    auto seg_prefetch = [&](int j, int t) { if (j + 1 < nj_live) seg_fetch(j + 1, t); else seg_fetch(0, t + 1); };
    seg_fetch(0, 0);
    unsigned noise_seen = 0;
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int j = 0; j < p.upr; ++j) {
            const int b = ring + j * p.n_rings;
            if (b >= p.B) continue;
            // ---- everything that does not depend on the network, while the ring works ---------------------------
            const bool nz_ok = wait_noise(p, t, noise_seen);                        // (false: draining -- the step's noise is not read)
            if (!nz_ok) s.flags[0] = 1;
            int tl = t, ub = p.b0 + b;                                              // coordinates of the noise stream
            bool next_starts = false;                                               // packed slots: step t + 1 is the first of another utterance
            if constexpr (PACKED) {
                // (the maps' entries of THIS iteration were asked for an iteration ago -- seg_prefetch: a ring that carries several
                //  utterances is bound by its head's occupancy per utterance, and a load -> Philox -> log chain at the top of it was 1-2 us)
                tl = t - pf_start; ub = pf_uid;
                next_starts = pf_next == t + 1;
                seg_prefetch(j, t);
            }
            float gum = 0.f, lr = 0.f, forced = 0.f;
            if (i < nmix && nz_ok) gum = -logf(-logf(head_noise(p, t, b, i, 0, tl, ub)));   // Gumbel noise (mixture.py:138-140)
            if (wave < 2) {
                const float r = nz_ok ? head_noise(p, t, b, nmix, p.dist == 2 ? 1 : 0, tl, ub) : 0.5f;
                lr = p.dist == 1 ? logf(r) - logf(1.0f - r) : r;                     // mixture.py:151-152 / :265-267
                if (t + 1 < p.Tt) forced = p.teacher[(size_t)b * p.Tt + t + 1];
            }
            const FeedAddr ad = feed_addr(b, (t + 1) & 1);
            if (t + 1 < p.T) fetch_pre0(b, tag + 1u);
            // ---- wait for the accumulated skip vector of (b, t) -----------------------------------------------
            if (!(SPLIT ? head_recv_skip_split(p, b, tag, s.vs, tid, lane, wave) : head_recv_skip<NK>(p, b, tag, s.vs, tid, lane, wave))) s.flags[0] = 1;
            __syncthreads();
            WNV_TS(1);
            float pt = 0.f, ps = 0.f;
            if (l0 && tid < RC) { pt = c_t + s.pre0[tid]; ps = c_s + s.pre0[RC + tid]; }
            // (the teacher sample has long arrived: consume it here, or the compiler waits for "every outstanding memory operation" --
            //  the chain store just issued included -- when it re-uses the register behind the sample)
            asm volatile("" : "+v"(forced), "+v"(pt), "+v"(ps));
            head_hidden<NK, 1>(w, s.vs, s.hid, q, i);
            __syncthreads();
            WNV_TS(3);
            float x[32];
            lds_read32(s.hid + QS * q, x);
            float o = quad_allreduce(dot32p(w.wh2[0], x));                        // wavenet.py:319
            if (NK > 1) {
                const bool act[1] = {q == 0 && i < p.O};
                float ov[1] = {o};
                if (!head_collect<1>(p, b, tag, i, act, ov, lane)) s.flags[0] = 1;
                o = ov[0];
            }
            o += bh2;
            if (q == 0 && i < p.O) {
                s.obuf[i] = o;
                if (i < nmix) s.vbuf[i] = o + gum;
                if (p.params_out) p.params_out[((size_t)b * p.O + i) * p.T + t] = o;
            }
            __syncthreads();
            WNV_TS(4);
            // ---- sample in waves 0-1 (each on its own), then first_conv of step t+1.  Up to 16 mixture components: lane c evaluates
            //      component c -- its three LDS reads are independent, ONE round trip --, the Gumbel-max is a 16-lane DPP butterfly
            //      + ballot (first index wins ties) and the winner's sample comes back through v_readlane.  (Round 2 walked the keys
            //      in a loop of dependent LDS reads and then fetched mean / log-scale: ~0.49 us from barrier to send.) ----
            float xs = 0.f, xout = 0.f;
            if (wave < 2) {
                float xo;
                if (nmix <= 16) {
                    float key = -INFINITY, mean = 0.f, ls = 0.f;
                    if (nmix == 0) { mean = s.obuf[o_mean]; ls = s.obuf[o_ls]; }   // mixture.py:258-261
                    else if (lane < nmix) { key = s.vbuf[lane]; mean = s.obuf[o_mean + lane]; ls = s.obuf[o_ls + lane]; }   // mixture.py:143-146
                    float xc = p.dist == 1 ? mean + __expf(ls) * lr : lr * __expf(ls) + mean;
                    xc = fminf(fmaxf(xc, -1.0f), 1.0f);                           // mixture.py:154 / :269
                    if (nmix > 0) {                                                 // Gumbel-max (mixture.py:138-140), first index wins ties
                        float m = key;
                        m = fmaxf(m, dpp_mov<0xB1>(m)); m = fmaxf(m, dpp_mov<0x4E>(m));
                        m = fmaxf(m, dpp_mov<0x141>(m)); m = fmaxf(m, dpp_mov<0x140>(m));      // row_half_mirror, row_mirror: lanes 0-15
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
                        const float4 v = reinterpret_cast<const float4*>(s.vbuf)[c];
                        if (v.x > best) { best = v.x; bi = 4 * c; }
                        if (v.y > best) { best = v.y; bi = 4 * c + 1; }
                        if (v.z > best) { best = v.z; bi = 4 * c + 2; }
                        if (v.w > best) { best = v.w; bi = 4 * c + 3; }
                    }
                    const float mean = s.obuf[o_mean + bi], ls = s.obuf[o_ls + bi];
                    xo = p.dist == 1 ? mean + expf(ls) * lr : lr * expf(ls) + mean;
                    xo = fminf(fmaxf(xo, -1.0f), 1.0f);
                }
                xs = t + 1 < p.Tt ? forced : xo;                                    // wavenet.py:297-305
                if (next_starts) xs = 0.f;                                          // (a new utterance begins: wavenet.py:281-283)
                xout = xo;
            }
            if (t + 1 < p.T) feed(b, tag + 1u, ad, xs, pt, ps);
            if (tid == 0) p.out[(size_t)b * p.T + t] = xout;
            WNV_TS(2);
            if (wave == 0) { WNV_TS_FLUSH(b, t, p.S, 0x1Eu, 0); WNV_TS_FLUSH(b, t + 1, p.S, 0x1u, 0); }   // 0 input of step t+1 sent | 1 skip sum in LDS | 2 step done | 3 hidden layer in LDS | 4 head outputs in LDS
            if (s.flags[0]) return;                 // uniform: written before the barriers above
        }
    }
}

// ---- head of one-hot (mu-law categorical) models: wavenet.py:315-319 head, :332-335 softmax + OneHotCategorical, :297-308
// first_conv on the fed-back one-hot vector.  first_conv's matrix lives in LDS K-major, so the fed-back class is ONE row
// gather (bit-identical to F.linear with a one-hot input); teacher-forced inputs and fed-back probabilities
// (quantize = False, tests only) take the dense mat-vec.  Sampling is sample_categorical()'s arithmetic (wnv_sample.h:
// argmax_k exp(logit_k - max) / e_k, e ~ Exp(1) -- torch.multinomial's argmax(p_hat / e) without the common normalising factors), one
// class per lane on all eight waves since round 4; the general path (no softmax / quantize = False) calls sample_categorical() in wave 0.
struct CatLds {
    float* vs; float* hid; float* obuf; float* nzb; float* vin; float* part; int* ints; float* wfl;
};
__device__ __forceinline__ CatLds carve_cat(float* smem, int NK) {
    CatLds s;
    s.vs = smem; s.hid = smem + 4 * NK * QS; s.obuf = s.hid + 4 * QS; s.nzb = s.obuf + 256; s.vin = s.nzb + 256;
    s.part = s.vin + 256; s.ints = reinterpret_cast<int*>(s.part + 4 * RC); s.wfl = reinterpret_cast<float*>(s.ints + 16);
    return s;
}
__host__ __device__ constexpr size_t cat_lds_floats(int NK) { return (size_t)(4 * NK + 4) * QS + 3 * 256 + 4 * RC + 16 + (size_t)256 * RC; }

// (LOGPICK: the softmax + multinomial pick in the log domain -- see the fastcat block.  Round 5 had it in the throughput and packed
//  instantiations only (cfg1 at 48 utterances 2 381 -> 2 463 kSamples/s; at 1 / 8 utterances 56.2 -> 55.1 / 444.5 -> 436.2 through the
//  register allocation of the NK = 2 stage loop) -- which made a one-hot waveform a function of the batch size: the two forms can part at a
//  near tie.  Round 6: ONE form in every instantiation; -DWNV_CAT_LOG=0 builds the quotient form everywhere, for A/B runs.)
template <int NK, bool PACKED, bool LOGPICK>
__device__ __attribute__((always_inline)) void run_head_cat(const RingParams& p, int ring, float* smem) {
    WNV_TS_DECL;
    const CatLds s = carve_cat(smem, NK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, i = wave * 16 + (lane >> 2);
    const int S1 = p.S + 1, O = p.O;
    HeadSlice<NK, 2> w;
    w.load(p, 0, tid, i);
    const float bh2a = p.bh2[i], bh2b = p.bh2[RC + i];
    const float bf = tid < RC ? p.bfirst[tid] : 0.f;
    for (int k = tid; k < O * RC; k += RT) s.wfl[k] = p.wfirst[k];
    if (tid == 0) { s.ints[0] = 0; s.ints[1] = 0; }                  // ints[0] = abort flag, ints[1] = sampled class
    const bool fast = same_xcd_as(p, ring, 1, ring + (p.S > 1 ? p.rstride : 0), s.ints + 2);

    // first_conv of `dense` (O floats in global memory or LDS) or of the one-hot class `idx`, sent as the input of `tag_next`
    auto send_input = [&](int b, const float* dense, int idx, unsigned tag_next) {
        if (dense == nullptr) {
            if (tid < RC) st_granule(p.xmail + ((size_t)b * S1) * RC + tid, tag_next, s.wfl[(size_t)idx * RC + tid] + bf, fast);
            return;
        }
        for (int k = tid; k < O; k += RT) s.vin[k] = dense[k];
        __syncthreads();
        {
            const int r = tid & (RC - 1), part = tid >> 7;          // four K parts x 128 outputs
            const int kper = (O + 3) / 4, ka = part * kper, kb = min(O, ka + kper);
            float acc = 0.f;
            for (int k = ka; k < kb; ++k) acc = fmaf(s.wfl[(size_t)k * RC + r], s.vin[k], acc);
            s.part[part * RC + r] = acc;
        }
        __syncthreads();
        if (tid < RC)
            st_granule(p.xmail + ((size_t)b * S1) * RC + tid, tag_next,
                       ((s.part[tid] + s.part[RC + tid]) + (s.part[2 * RC + tid] + s.part[3 * RC + tid])) + bf, fast);
        __syncthreads();
    };

    // ---- prologue: the input of step 0 (wavenet.py:283-289: one-hot of class 127 unless given) ------------------
    for (int j = 0; j < p.upr; ++j) {
        const int b = ring + j * p.n_rings;
        if (b >= p.B) continue;
        const float* dense = p.Tt > 0 ? p.teacher + (size_t)b * p.Tt * O : (p.initial ? p.initial + (size_t)b * O : nullptr);
        send_input(b, dense, 127, p.tag_base + 1u);
    }

    // packed slots: the maps' entries of the iteration about to run -- (utterance j, step t) -> pf_*; called with the CURRENT (j, t) it
    // fetches the next iteration's: (j + 1, t) or (0, t + 1)
    int pf_start = 0, pf_uid = 0, pf_next = -1;
    const int nj_live = min(p.upr, (p.B - ring + p.n_rings - 1) / p.n_rings);
    auto seg_fetch = [&](int jn, int tn) {
        if constexpr (PACKED) {
            if (tn < p.T) {
                const int* base = p.seg_start + (size_t)(ring + jn * p.n_rings) * p.T + tn;
                pf_start = uniform_ld(base);
                pf_uid = uniform_ld(p.seg_uid + (size_t)(ring + jn * p.n_rings) * p.T + tn);
                pf_next = tn + 1 < p.T ? uniform_ld(base + 1) : -1;
            }
        }
    };
    auto seg_prefetch = [&](int j, int t) { if (j + 1 < nj_live) seg_fetch(j + 1, t); else seg_fetch(0, t + 1); };
    seg_fetch(0, 0);
    unsigned noise_seen = 0;
    for (int t = 0; t < p.T; ++t) {
        const unsigned tag = p.tag_base + (unsigned)t + 1u;
        for (int j = 0; j < p.upr; ++j) {
            const int b = ring + j * p.n_rings;
            if (b >= p.B) continue;
            // noise of this step, while the ring works: e ~ Exp(1) per class
            const bool nz_ok = wait_noise(p, t, noise_seen);                        // (false: draining -- the step's noise is not read)
            if (!nz_ok) s.ints[0] = 1;
            int tl = t, ub = p.b0 + b;
            bool next_starts = false;                                               // packed slots: step t + 1 is the first of another utterance
            if constexpr (PACKED) {
                tl = t - pf_start; ub = pf_uid;                                     // (asked for an iteration ago: seg_prefetch)
                next_starts = pf_next == t + 1;
                seg_prefetch(j, t);
            }
            // (WNV_CAT_LOG, softmax + multinomial: the pick is taken in the log domain, argmax_k logit_k - log e_k -- the noise term is
            //  prepared HERE, while the ring works.  The in-kernel stream's e is strictly positive -- wnv_u01 is exact and never 1, round 6 --,
            //  a tape's e = +0.0 scores +inf in this form as in the quotient form.)
            if constexpr (LOGPICK) {
                if (tid < O) {
                    const float ek = nz_ok ? head_noise(p, t, b, tid, 2, tl, ub) : 1.0f;
                    s.nzb[tid] = (p.quantize && p.softmax && O <= 256) ? logf(ek) : ek;
                }
            } else {
                if (tid < O) s.nzb[tid] = nz_ok ? head_noise(p, t, b, tid, 2, tl, ub) : 1.0f;
            }
            if (!head_recv_skip<NK>(p, b, tag, s.vs, tid, lane, wave)) s.ints[0] = 1;
            __syncthreads();
            WNV_TS(1);
            head_hidden<NK, 2>(w, s.vs, s.hid, q, i);
            __syncthreads();
            WNV_TSX(3);
            float x[32];
            lds_read32(s.hid + QS * q, x);
            float oa = quad_allreduce(dot32p(w.wh2[0], x));                       // wavenet.py:319, rows i and 128 + i
            float ob = quad_allreduce(dot32p(w.wh2[1], x));
            WNV_TSX(4);
            // (fastcat, below: lane q = 0 of a quad finishes class i, lane q = 1 class 128 + i -- each collects only its own row)
            const bool fastcat = p.quantize && p.softmax && O <= 256;
            const int q2 = fastcat ? 1 : 0;                                          // the lane of the quad that owns row 128 + i
            if (NK > 1) {
                const bool act[2] = {q == 0 && i < O, q == q2 && RC + i < O};
                float ov[2] = {oa, ob};
                if (!head_collect<2>(p, b, tag, i, act, ov, lane)) s.ints[0] = 1;
                oa = ov[0]; ob = ov[1];
            }
            WNV_TSX(5);
            oa += bh2a; ob += bh2b;
            if (p.params_out) {
                if (q == 0 && i < O) p.params_out[((size_t)b * O + i) * p.T + t] = oa;
                if (q == q2 && RC + i < O) p.params_out[((size_t)b * O + RC + i) * p.T + t] = ob;
            }
            // wavenet.py:332-335, then :297-308 for step t + 1.
            // SOFTMAX + MULTINOMIAL SPREAD OVER THE WORKGROUP (round 4).  After the quad reduce every lane of a quad holds the logits of
            // classes i and 128 + i: lane q = 0 takes the first, q = 1 the second -- one class per lane, 32 per wave.  The wave maxima meet
            // in LDS, every lane forms exp(logit - max) / e of ITS class (sample_categorical's arithmetic, wnv_sample.h: the normalising sums
            // are common factors the argmax does not see), the waves' best (value, class) pairs meet in LDS and every sending wave picks the
            // winner -- smallest class among equals.  (Until round 4 one wave did all of it, four classes per lane, with both
            // normalisations: 1.35 us of a 19.9 us step, profiles/r04_onehot_head_timeline.txt.)
            if (fastcat) {
                const int cls = q == 0 ? i : RC + i;
                const bool mine = q < 2 && cls < O;
                const float lg = mine ? (q == 0 ? oa : ob) : -INFINITY;
                const float ek = mine ? s.nzb[cls] : 1.f;
                float best;
                int bi;
                if constexpr (LOGPICK) {
                    // ROUND 5: argmax_k exp(logit_k - max) / e_k = argmax_k logit_k - log e_k (the logarithm is monotone, the maximum a
                    // common term): no pass for the maximum -- a wave reduction, an LDS exchange and a barrier --, no exp, no division on
                    // the chain; log e_k was formed with the noise.  A pick can move only where the top-2 margin is below the rounding of
                    // the difference (tests/test_sampler_arith_cpu.py: no flip in 600 000 draws at three logit spreads).
                    best = mine ? lg - ek : -INFINITY;
                    bi = mine ? cls : 0x7fffffff;
                } else {
                    const float mw = wave_max(lg);
                    if (lane == 0) s.part[wave] = mw;
                    __syncthreads();
                    const float4 ma = *reinterpret_cast<const float4*>(s.part), mb = *reinterpret_cast<const float4*>(s.part + 4);
                    const float mx = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
                    best = mine ? expf(lg - mx) / ek : -INFINITY;
                    bi = mine ? cls : 0x7fffffff;
                }
                wave_argmax(best, bi);
                if (lane == 0) { s.part[8 + wave] = best; s.part[16 + wave] = __int_as_float(bi); }
            } else if (q == 0) {
                if (i < O) s.obuf[i] = oa;
                if (RC + i < O) s.obuf[RC + i] = ob;
            }
            __syncthreads();
            // Waves 0 and 1 each pick / sample on their own (same inputs, same class) and send their 64 channels of first_conv's row at
            // once: no barrier and no LDS round trip between the argmax and the chain store
            const bool dense_next = t + 1 < p.Tt || !p.quantize;
            WNV_TSX(6);
            if (wave < (p.quantize ? 2 : 1)) {              // (quantize = False: the probabilities go back into obuf -- one wave)
                int idx;
                if (fastcat) {
                    float bv = s.part[8];
                    idx = __float_as_int(s.part[16]);
#pragma unroll
                    for (int k = 1; k < RW; ++k) {
                        const float v = s.part[8 + k];
                        const int c = __float_as_int(s.part[16 + k]);
                        if (v > bv || (v == bv && c < idx)) { bv = v; idx = c; }
                    }
                } else {
                    idx = sample_categorical(O, s.obuf, s.nzb, p.softmax, p.quantize, lane);
                }
                WNV_TSX(7);
                if (p.quantize) {
                    if (t + 1 < p.T && !dense_next) {
                        // (a new utterance begins with the one-hot vector of class 127: wavenet.py:284-289)
                        st_granule(p.xmail + ((size_t)b * S1) * RC + tid, tag + 1u, s.wfl[(size_t)(next_starts ? 127 : idx) * RC + tid] + bf, fast);
                        WNV_TS(0);
                    }
                    if (tid == 0) {
                        // out is pre-zeroed by the host.  (NULL = classes only, index_out: a PACKED-slot launch may ask for that -- a compile-time
                        //  condition elsewhere: the run-time test alone moved the one-hot instantiations' code, cfg1 B = 8 444 -> 436 kSamples/s)
                        if (!PACKED || p.out) p.out[((size_t)b * O + idx) * p.T + t] = 1.0f;
                        if (p.index_out) p.index_out[(size_t)b * p.T + t] = idx;
                    }
                } else if (wave == 0) {
                    for (int n = lane; n < O; n += 64) p.out[((size_t)b * O + n) * p.T + t] = s.obuf[n];
                }
            }
            __syncthreads();
            if (t + 1 < p.T && dense_next) {
                // teacher-forced input, or fed-back probabilities (quantize = False, tests only)
                send_input(b, t + 1 < p.Tt ? p.teacher + ((size_t)b * p.Tt + t + 1) * O : s.obuf, 0, tag + 1u);
                WNV_TS(0);
            }
            WNV_TS(2);
            // 0 input of step t+1 sent | 1 skip sum in LDS | 2 step done | 3 hidden half in LDS | 4 own partial outputs | 5 the other parts' collected |
            // 6 logits in LDS | 7 class sampled
            if (wave == 0) { WNV_TS_FLUSH(b, t, p.S, 0xFEu, 0); WNV_TS_FLUSH(b, t + 1, p.S, 0x1u, 0); }
            if (s.ints[0]) return;
        }
    }
}

// MODE: 0 = one to four utterances per ring, 1 = more (the stages' throughput prologue), 2 = packed slots (seg_start; any number)
template <int NK, bool L0, bool SPLIT, int MODE, bool MF = false>
__device__ __forceinline__ void ring_body(const RingParams& p) {
    constexpr bool MULTI = MODE >= 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (SPLIT) {
        // split rings (see run_stage_split): block b on XCD b % 8, slot b / 8; ring r = XCDs 2r (head, stages 1 .. sA) and 2r + 1
        if ((int)blockIdx.x >= p.ring_blocks) {
            const int k = (int)blockIdx.x - p.ring_blocks;
            if (k < p.tap_parts * p.L) run_tap<true, false, false>(p, k % p.L, k / p.L, smem);
            return;
        }
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, ring = xcd >> 1;
        if (ring >= p.n_rings) return;
        if ((xcd & 1) == 0) {
            if (slot == 0) { run_head<NK, true, true, false>(p, ring, smem); return; }
            const int stage = 1 + ((slot - 1) >> 1), half = (slot - 1) & 1;
            if (stage > p.sA) return;
            if (stage == 1) run_stage_split<true>(p, ring, stage, half, smem);
            else run_stage_split<false>(p, ring, stage, half, smem);
        } else {
            const int stage = p.sA + 1 + (slot >> 1), half = slot & 1;
            if (stage >= p.S) return;
            run_stage_split<false>(p, ring, stage, half, smem);
        }
        return;
    } else {
    const int P = p.S + p.NH;                   // workgroups of one ring: S stages + NH head parts
    // block i -> XCD i % 8 (observed, speed only): with rstride == 8 every workgroup of ring r sits on XCD r
    // tap workgroup k (layer k % L, part k / L) sits in the k-th block that is not part of a live ring: first the slots of
    // unused ring indices (they share an XCD with nothing but each other), then the blocks behind the rings
    const int free_slots = (p.rstride - p.n_rings) * P;
    if ((int)blockIdx.x >= p.ring_blocks) {
        const int k = free_slots + (int)blockIdx.x - p.ring_blocks;
        {
            if constexpr (MF) run_tap_mf<(MODE == 0 || (MODE == 1 && WNV_TAP_SPEC1 != 0) || (MODE == 2 && WNV_PACKED_SPEC != 0)) && NK != 2, MODE == 2, MODE != 0>(p, k % p.L, k / p.L, smem);
            else run_tap<(MODE == 0 || (MODE == 1 && WNV_TAP_SPEC1 != 0) || (MODE == 2 && WNV_PACKED_SPEC != 0)) && NK != 2, MODE == 2, MODE != 0>(p, k % p.L, k / p.L, smem);
        }
        return;
    }
    const int ring = blockIdx.x % p.rstride;
    const int pos = blockIdx.x / p.rstride;
    if (ring >= p.n_rings) {
        const int k = pos * (p.rstride - p.n_rings) + (ring - p.n_rings);
        if (k < p.tap_parts * p.L) {
            if constexpr (MF) run_tap_mf<(MODE == 0 || (MODE == 1 && WNV_TAP_SPEC1 != 0) || (MODE == 2 && WNV_PACKED_SPEC != 0)) && NK != 2, MODE == 2, MODE != 0>(p, k % p.L, k / p.L, smem);
            else run_tap<(MODE == 0 || (MODE == 1 && WNV_TAP_SPEC1 != 0) || (MODE == 2 && WNV_PACKED_SPEC != 0)) && NK != 2, MODE == 2, MODE != 0>(p, k % p.L, k / p.L, smem);
        }
        return;
    }
    if (L0 && pos == 0) return;                   // layer 0 is evaluated by the head (run_head)
    if (L0 && pos == 1) run_stage<NK, L0, L0, MULTI>(p, ring, pos, smem);
    else if (pos < p.S) run_stage<NK, L0, false, MULTI>(p, ring, pos, smem);
    else if (!L0 && p.cin1 > 1) {
        // (L0 instantiations serve scalar-input models only -- the categorical head is not compiled into them: every role of a kernel is
        //  inlined into ONE function, and a change in that head moved the register allocation of the stage loop -- 2 % of the headline)
        if constexpr (NK <= 2 && !L0) {         // one-hot models with 512 skip channels stay on the generic kernel (why_not)
            // (round 6: ONE pick form in every instantiation -- with bit-identical logits a seed gives the same classes whatever the batch size
            //  or the packing: tests/test_gpu_seed_determinism.py)
            if (pos == p.S) run_head_cat<NK, MODE == 2, WNV_CAT_LOG != 0>(p, ring, smem);
            else run_head_part<NK, 2>(p, ring, pos - p.S, smem);
        }
    } else {
        if (pos == p.S) run_head<NK, L0, false, MODE == 2>(p, ring, smem);
        else if constexpr (NK > 1) run_head_part<NK, 1>(p, ring, pos - p.S, smem);     // (128 skip channels: the head is one workgroup)
    }
    }
}

// NK <= 2: capped at 244 VGPRs -- v244 .. v255 are the poll slots (see "POLLS IN RESERVED REGISTERS")
template <int NK, bool L0, int MODE>
__global__ void __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(244))) wnv_ring_kernel(const RingParams p) { ring_body<NK, L0, false, MODE>(p); }
// (round 6) the same kernels with the tap role on the matrix pipe (run_tap_mf) -- separate instantiations, so that the kernels above compile
// from exactly the code they had: every role of a kernel is inlined into one function and a change in one moves the others' registers
template <int NK, bool L0, int MODE>
__global__ void __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(244))) wnv_ring_kernel_mf(const RingParams p) { ring_body<NK, L0, false, MODE, true>(p); }
// split rings: two CUs per layer (scalar-input models with 128 skip channels, up to 8 utterances)
__global__ void __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(244))) wnv_ring_kernel_split(const RingParams p) { ring_body<1, true, true, 0>(p); }
// K = 512: capped like the others since round 4 (two spilled registers in the MODE 0 instantiation, none in the others): its polls used to
// go one load at a time through compiler-allocated registers -- ~75 ns more per hop than the pipelined ones
template <int MODE>
__global__ void __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(244))) wnv_ring_kernel_k512(const RingParams p) { ring_body<4, false, false, MODE>(p); }

// Placement census (once per handle): every workgroup of a one-block-per-CU grid reports the XCC it runs on.  The host derives
// the number of XCDs and checks the block -> XCD mapping the ring layout relies on (block b on XCD b % n_xcd, observed; HIP
// promises nothing) instead of assuming it.
__global__ void __launch_bounds__(64) wnv_ring_census_kernel(unsigned int* xcc) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[blockIdx.x] = (x & 0xfu) + 1u;
    }
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================
struct WnvRingState {
    int device = 0;
    int L = 0, S = 0, K = 0, Kp = 0, O = 0, cin = 0, kw = 0, kpre = 0, cin1 = 1;
    float* d_w = nullptr;          // one blob, offsets below (floats)
    size_t o_wn = 0, o_cvec = 0, o_w2 = 0, o_wo = 0, o_bo = 0, o_wpre = 0, o_ws = 0, o_bskip = 0, o_wh1 = 0, o_bh1 = 0, o_wh2 = 0,
           o_bh2 = 0, o_wf = 0, o_bf = 0, o_l0 = 0, o_w2s = 0, o_wns = 0, o_wos = 0, o_wss = 0;
    bool has_split = false;        // split-ring images present (scalar input, 128 skip channels)
    int* d_dil = nullptr;
    int* d_histoff = nullptr;
    int hist_floats = 0;
    void* d_state = nullptr;       // status + placement table + mailboxes + history
    size_t state_cap = 0;
    unsigned tag_next = 0;         // tags handed out so far (mailboxes are only re-zeroed when this would wrap)
    size_t mail_bytes = 0;         // size of the mailbox region the tags in flight refer to
    // placement census of this device (see wnv_ring_census_kernel)
    int ncu = 0, n_xcd = 0;
    bool map_ok = false;           // block b ran on the same XCD as block b % n_xcd for every b of the census grid
    // asynchronous launches (WnvGenArgs::async): the status word lands in pinned host memory behind the kernel
    unsigned int* h_status = nullptr;
    bool pending = false;
    hipStream_t pending_stream = nullptr;
};

static const char* why_not(const wnv_config& c, int B) {
    if (!c.scalar_input && c.out_channels > 256) return "one-hot models need out_channels <= 256";
    // narrower models run zero-padded to the kernel's 128 / 256 / 128 n geometry (exact: padded channels stay 0 through every layer)
    if (c.residual_channels > RC || c.gate_channels > GC) return "needs residual_channels <= 128 and gate_channels <= 256";
    if (c.skip_out_channels > 512) return "needs skip_out_channels <= 512 (one head workgroup per 128 hidden units, at most four)";
    if (!c.scalar_input && c.skip_out_channels > 256) return "one-hot models need skip_out_channels <= 256";
    if (c.scalar_input && c.out_channels > 128) return "needs out_channels <= 128";
    if (c.kernel_size < 2 || c.kernel_size > 4) return "needs 2 <= kernel_size <= 4";
    if (c.cin_channels > 512 - (c.kernel_size - 1) * RC) return "too many local-conditioning channels";
    { int nk = (c.skip_out_channels + 127) / 128; if (nk == 3) nk = 4; if (c.layers + nk > 32) return "too many layers for one ring per XCD"; }
    return nullptr;
}
bool wnv_ring_supported(const wnv_config& c, int B) { return why_not(c, B) == nullptr; }
bool wnv_ring_default() {
    const char* e = wnv_knob("WNV_RING");
    if (e && *e) return e[0] != '0';
    return WNV_RING_IS_DEFAULT != 0;
}
const char* wnv_ring_why_not(const wnv_config& c, int B) { const char* w = why_not(c, B); return w ? w : "supported"; }

// -DWNV_GUARD: every device buffer the sample-loop kernel writes (the state block: status, placement table, mailboxes, records,
// history; the timeline buffer of trace builds) sits between two red zones of a fixed byte pattern that the host checks after the
// launch -- a write past either end is reported instead of landing in a neighbour's memory.
#ifdef WNV_GUARD
constexpr size_t GUARD_BYTES = 1 << 20;
constexpr int GUARD_BYTE = 0xC3;
static hipError_t guard_alloc(void** user, size_t bytes) {
    char* raw = nullptr;
    hipError_t e = hipMalloc((void**)&raw, bytes + 2 * GUARD_BYTES);
    if (e != hipSuccess) return e;
    e = hipMemset(raw, GUARD_BYTE, GUARD_BYTES);
    if (e == hipSuccess) e = hipMemset(raw + GUARD_BYTES + bytes, GUARD_BYTE, GUARD_BYTES);
    *user = raw + GUARD_BYTES;
    return e;
}
static hipError_t guard_free(void* user) { return user ? hipFree((char*)user - GUARD_BYTES) : hipSuccess; }
// true when both red zones are intact; otherwise says where the first foreign byte sits
static bool guard_check(const void* user, size_t bytes, const char* what) {
    std::vector<unsigned char> z(GUARD_BYTES);
    bool ok = true;
    for (int side = 0; side < 2; ++side) {
        const char* zone = side ? (const char*)user + bytes : (const char*)user - GUARD_BYTES;
        if (hipMemcpy(z.data(), zone, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "[wnv guard] %s: cannot read the red zone\n", what); return false; }
        for (size_t i = 0; i < GUARD_BYTES; ++i)
            if (z[i] != GUARD_BYTE) {
                const long long off = side ? (long long)i : (long long)i - (long long)GUARD_BYTES;
                fprintf(stderr, "[wnv guard] %s (%zu bytes): red zone %s it is overwritten, first at %s%lld bytes (value 0x%02x)\n", what, bytes,
                        side ? "behind" : "in front of", side ? "end + " : "start ", off, z[i]);
                ok = false;
                break;
            }
    }
    return ok;
}
#else
static hipError_t guard_alloc(void** user, size_t bytes) { return hipMalloc(user, bytes); }
static hipError_t guard_free(void* user) { return user ? hipFree(user) : hipSuccess; }
#endif

void wnv_ring_destroy(WnvRingState* st) {
    if (!st) return;
    if (st->pending) (void)hipStreamSynchronize(st->pending_stream);   // an asynchronous launch still reads all of the below
    if (st->d_w) (void)hipFree(st->d_w);
    if (st->d_dil) (void)hipFree(st->d_dil);
    if (st->d_histoff) (void)hipFree(st->d_histoff);
    if (st->d_state) (void)guard_free(st->d_state);
    if (st->h_status) (void)hipHostFree(st->h_status);
    delete st;
}

#define RING_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) { err = std::string(#expr) + " failed: " + hipGetErrorString(e__); return WNV_ERR_HIP; } \
    } while (0)

// thread tid of a 512-thread workgroup owns channel i = 16*(tid>>6) + ((tid&63)>>2) and K-quarter q = tid&3
static inline void tid_map(int tid, int& i, int& q) { i = 16 * (tid >> 6) + ((tid & 63) >> 2); q = tid & 3; }

// image of a (rows x 128) matrix M (row-major, M[o][k]) for the (o, q) register mapping:
// chunk c (0..7) of thread tid = M[row_of(tid)][32q + 4c .. +4]
// of the 128-column block starting at column col_offset of a row-major matrix with leading dimension ld
static void put_image(std::vector<float>& blob, size_t off, const float* M, int ld, int row_offset, int n_rows, int col_offset) {
    for (int tid = 0; tid < RT; ++tid) {
        int i, q;
        tid_map(tid, i, q);
        const int row = row_offset + i;
        for (int c = 0; c < 8; ++c)
            for (int e = 0; e < 4; ++e)
                blob[off + ((size_t)c * RT + tid) * 4 + e] = row < n_rows ? M[(size_t)row * ld + col_offset + 32 * q + 4 * c + e] : 0.f;
    }
}

// stage-kernel row image: thread tid (K-slice ks = tid & 7, lane group og = tid >> 3) holds
// M[row_offset + 2og + odd][16ks .. 16ks + 16) as 4 chunks of 4 floats, laid out [chunk][512 threads][4]
static void put_row8(std::vector<float>& blob, size_t off, const float* M, int row_offset, int odd) {
    for (int tid = 0; tid < RT; ++tid) {
        const int ks = tid & 7, og = tid >> 3;
        const float* src = M + (size_t)(row_offset + 2 * og + odd) * RC + 16 * ks;
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) blob[off + ((size_t)c * RT + tid) * 4 + e] = src[4 * c + e];
    }
}

// wave-group row image (see group_matvec8): group thread gtid (K-slice ks = gtid & 7, lane group og = gtid >> 3) holds in register
// slot `slot` the K-slice [16ks, 16ks + 16) of row (slot & 1 ? 128 : 0) + 4og + j, j = the lane-dependent channel of slot pair
// slot >> 1; laid out [chunk 4][256 threads][4]
static void put_row8g(std::vector<float>& blob, size_t off, const float* M, int slot) {
    for (int gtid = 0; gtid < GT; ++gtid) {
        const int ks = gtid & 7, og = gtid >> 3, b2 = ks >> 2, b1 = (ks >> 1) & 1;
        const int sp = slot >> 1;
        const int j = sp == 0 ? 2 * b2 + b1 : sp == 1 ? 2 * b2 + (1 - b1) : sp == 2 ? 2 * (1 - b2) + (1 - b1) : 2 * (1 - b2) + b1;
        const float* src = M + (size_t)(((slot & 1) ? RC : 0) + 4 * og + j) * RC + 16 * ks;
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) blob[off + ((size_t)c * GT + gtid) * 4 + e] = src[4 * c + e];
    }
}

// split rings (run_stage_split): wave-group row image of HALF h of a (256 x 128) matrix: group thread gtid (ks = gtid & 7, og = gtid >> 3)
// keeps local channel 2og + (ks >= 4) in slots 0-1 {tanh, sigmoid} and holds the other one of the pair in slots 2-3;
// laid out [chunk 4][256 threads][4]
static void put_row4s(std::vector<float>& blob, size_t off, const float* M, int half, int slot) {
    for (int gtid = 0; gtid < GT; ++gtid) {
        const int ks = gtid & 7, og = gtid >> 3, hi = ks >= 4 ? 1 : 0;
        const int lc = 2 * og + ((slot >> 1) == 0 ? hi : 1 - hi);
        const float* src = M + (size_t)(((slot & 1) ? RC : 0) + 64 * half + lc) * RC + 16 * ks;
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) blob[off + ((size_t)c * GT + gtid) * 4 + e] = src[4 * c + e];
    }
}
// ... and of the K-half [64 h, 64 h + 64) of a (128 x 128) matrix for the work behind the send: thread tid (row r = tid >> 2, kq = tid & 3)
// holds M[r][64 h + 16 kq .. + 16); laid out [chunk 4][512 threads][4]
static void put_rowq(std::vector<float>& blob, size_t off, const float* M, int half) {
    for (int tid = 0; tid < RT; ++tid) {
        const float* src = M + (size_t)(tid >> 2) * RC + 64 * half + 16 * (tid & 3);
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) blob[off + ((size_t)c * RT + tid) * 4 + e] = src[4 * c + e];
    }
}

wnv_status wnv_placement_census(int device, int* ncu_out, int* n_xcd_out, bool* map_ok_out, std::string& err) {
    int ncu = 0;
    RING_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    const int n = std::max(ncu, 1);
    unsigned int* d_x = nullptr;
    RING_HIP(hipMalloc((void**)&d_x, n * sizeof(unsigned int)));
    RING_HIP(hipMemset(d_x, 0, n * sizeof(unsigned int)));
    hipLaunchKernelGGL(wnv_ring_census_kernel, dim3(n), dim3(64), 0, 0, d_x);
    RING_HIP(hipGetLastError());
    std::vector<unsigned int> x(n);
    RING_HIP(hipMemcpy(x.data(), d_x, n * sizeof(unsigned int), hipMemcpyDeviceToHost));
    (void)hipFree(d_x);
    unsigned seen = 0;
    for (int b = 0; b < n; ++b) seen |= 1u << (x[b] & 31u);
    const int n_xcd = __builtin_popcount(seen & ~1u);
    bool map_ok = n_xcd >= 1;
    for (int b = 0; b < n && map_ok; ++b) map_ok = x[b] != 0u && x[b] == x[b % n_xcd];
    if (const char* e = wnv_knob("WNV_RING_CENSUS"); e && e[0] == '1')
        fprintf(stderr, "[wnv] device %d: %d CUs, %d XCDs, block -> XCD (b %% %d) mapping %s\n", device, ncu, n_xcd, n_xcd,
                map_ok ? "verified" : "NOT as assumed");
    *ncu_out = ncu; *n_xcd_out = n_xcd; *map_ok_out = map_ok;
    return WNV_OK;
}

static wnv_status build_state(WnvRingState** out, int device, const wnv_config& c, const TensorStore& store,
                              std::string& err) {
    WnvRingState* st = new WnvRingState();
    st->device = device;
    const int L = c.layers, kw = c.kernel_size, cin = c.cin_channels > 0 ? c.cin_channels : 0;
    // The kernel's geometry is fixed: 128 residual channels, 128 gate channels (256 gate rows), 128 n skip channels.  A narrower
    // model is embedded by ZERO PADDING: padded residual / gate / skip channels carry exact zeros through every layer
    // (z = 0 -> tanh(0) sigmoid(0) = 0; zero rows and biases), so the live channels see the same sums plus exact-zero terms.
    const int Ra = c.residual_channels, Ga = c.gate_channels, Gha = Ga / 2, Ka = c.skip_out_channels, O = c.out_channels;
    int NK = (Ka + RC - 1) / RC;                                    // skip passes per stage = head parts per ring: 1, 2 or 4 (the
    if (NK == 3) NK = 4;                                            // kernel is instantiated for those; 257 .. 384 channels pad to 512)
    const int K = NK * RC, Kp = K;                                  // padded skip width
    st->L = L; st->S = L; st->K = K; st->Kp = Kp; st->O = O; st->cin = cin; st->kw = kw;
    st->kpre = (kw - 1) * RC + cin;
    std::vector<float> blob;
    auto alloc = [&](size_t n) { size_t o = (blob.size() + 3) & ~(size_t)3; blob.resize(o + n, 0.f); return o; };
    auto T = [&](const std::string& n) -> const HostTensor& { return *store.get(n); };
    // padded gate row o (tanh channels 0..127, then sigmoid channels 0..127) -> the model's gate row, or -1
    auto gate_row = [&](int o) { const int ch = o & (RC - 1); return ch < Gha ? (o >> 7) * Gha + ch : -1; };
    // WNV_PHASE2: the gate's exp2 scales, folded into every term of z (ring_gate): padded row o is a tanh row (o < 128) or a sigmoid row
    auto gate_scale = [](int o) -> double { return WNV_PHASE2 ? (double)(o < RC ? GATE_SCALE_TANH : GATE_SCALE_SIGM) : 1.0; };
    // a (rows x cols) row-major matrix zero-padded to (prow x pcol)
    auto padded = [](const float* M, int rows, int cols, int prow, int pcol) {
        std::vector<float> P((size_t)prow * pcol, 0.f);
        for (int r = 0; r < rows; ++r) std::copy(M + (size_t)r * cols, M + (size_t)(r + 1) * cols, P.begin() + (size_t)r * pcol);
        return P;
    };
    st->o_w2 = alloc((size_t)L * 16 * RT * 4);
    st->o_wn = alloc((size_t)L * 16 * RT * 4);
    st->o_cvec = alloc((size_t)L * GC);
    st->o_wo = alloc((size_t)L * 8 * RT * 4);
    st->o_bo = alloc((size_t)L * RC);
    st->o_wpre = alloc((size_t)L * st->kpre * GC);
    st->o_ws = alloc((size_t)L * NK * 8 * RT * 4);
    st->o_bskip = alloc((size_t)L * Kp);
    st->has_split = c.scalar_input && NK == 1;
    if (st->has_split) {
        st->o_w2s = alloc((size_t)L * 2 * 4 * 4 * GT * 4);
        st->o_wns = alloc((size_t)L * 2 * 4 * 4 * GT * 4);
        st->o_wos = alloc((size_t)L * 2 * 4 * RT * 4);
        st->o_wss = alloc((size_t)L * 2 * 4 * RT * 4);
    }
    std::vector<int> dil(L), hoff(L);
    int hist = 0;
    const int per = L / c.stacks;
    std::vector<float> cur((size_t)GC * RC), mmat((size_t)GC * RC), nmat((size_t)GC * RC);
    const double rs = std::sqrt(0.5);
    for (int l = 0; l < L; ++l) {
        const std::string pfx = "conv_layers." + std::to_string(l) + ".";
        const HostTensor& wc = T(pfx + "conv.weight");                 // (G, R, kw)
        // newest tap (k = kw-1) as a padded (256 x 128) matrix
        std::fill(cur.begin(), cur.end(), 0.f);
        for (int o = 0; o < GC; ++o) {
            const int go = gate_row(o);
            if (go < 0) continue;
            for (int ii = 0; ii < Ra; ++ii) cur[(size_t)o * RC + ii] = wc.data[((size_t)go * Ra + ii) * kw + (kw - 1)];
        }
        // gate-to-gate chain (see run_stage): M_l = sqrt(.5) W_cur,l W_o,l-1, N_l = sqrt(.5) W_cur,l, c_l = N_l b_o,l-1, folded in
        // double and rounded once; layer 0 reads h_0 itself: M_0 = W_cur,0, no N term
        if (l == 0) {
            for (int o = 0; o < GC; ++o)
                for (int kk = 0; kk < RC; ++kk) mmat[(size_t)o * RC + kk] = (float)(gate_scale(o) * (double)cur[(size_t)o * RC + kk]);
        } else {
            const std::vector<float> wop = padded(T("conv_layers." + std::to_string(l - 1) + ".conv1x1_out.weight").data.data(), Ra, Gha, RC, RC);   // (R, G/2, 1)
            const std::vector<float> bop = padded(T("conv_layers." + std::to_string(l - 1) + ".conv1x1_out.bias").data.data(), 1, Ra, 1, RC);
            for (int o = 0; o < GC; ++o) {
                for (int kk = 0; kk < RC; ++kk) {
                    double acc = 0.0;
                    for (int m = 0; m < RC; ++m) acc += (double)cur[(size_t)o * RC + m] * (double)wop[(size_t)m * RC + kk];
                    mmat[(size_t)o * RC + kk] = (float)(gate_scale(o) * rs * acc);
                    nmat[(size_t)o * RC + kk] = (float)(gate_scale(o) * rs * (double)cur[(size_t)o * RC + kk]);
                }
                double cb = 0.0;
                for (int m = 0; m < RC; ++m) cb += (double)cur[(size_t)o * RC + m] * (double)bop[m];
                blob[st->o_cvec + (size_t)l * GC + o] = (float)(gate_scale(o) * rs * cb);
            }
        }
        const size_t rowsz = (size_t)4 * RT * 4;                   // one row image: 4 chunks x 512 threads x 4 floats
        for (int slot = 0; slot < 8; ++slot) {                     // M_l / N_l: eight register slots of the wave-group mapping
            put_row8g(blob, st->o_w2 + ((size_t)l * 8 + slot) * (rowsz / 2), mmat.data(), slot);
            if (l > 0) put_row8g(blob, st->o_wn + ((size_t)l * 8 + slot) * (rowsz / 2), nmat.data(), slot);
        }
        const std::vector<float> wo = padded(T(pfx + "conv1x1_out.weight").data.data(), Ra, Gha, RC, RC);          // (R, G/2, 1)
        put_row8(blob, st->o_wo + ((size_t)l * 2 + 0) * rowsz, wo.data(), 0, 0);
        put_row8(blob, st->o_wo + ((size_t)l * 2 + 1) * rowsz, wo.data(), 0, 1);
        if (st->has_split)
            for (int half = 0; half < 2; ++half) {
                for (int slot = 0; slot < 4; ++slot) {
                    put_row4s(blob, st->o_w2s + (((size_t)l * 2 + half) * 4 + slot) * (rowsz / 2), mmat.data(), half, slot);
                    if (l > 0) put_row4s(blob, st->o_wns + (((size_t)l * 2 + half) * 4 + slot) * (rowsz / 2), nmat.data(), half, slot);
                }
                put_rowq(blob, st->o_wos + ((size_t)l * 2 + half) * rowsz, wo.data(), half);
            }
        const HostTensor& bo = T(pfx + "conv1x1_out.bias");
        std::copy(bo.data.begin(), bo.data.end(), blob.begin() + st->o_bo + (size_t)l * RC);
        // deferred: older taps (oldest first) then local conditioning, K-major [kpre][256]
        float* wp = blob.data() + st->o_wpre + (size_t)l * st->kpre * GC;
        for (int o = 0; o < GC; ++o) {
            const int go = gate_row(o);
            if (go < 0) continue;
            for (int k = 0; k < kw - 1; ++k)
                for (int ii = 0; ii < Ra; ++ii) wp[(size_t)(k * RC + ii) * GC + o] = (float)(gate_scale(o) * (double)wc.data[((size_t)go * Ra + ii) * kw + k]);
            if (cin > 0) {
                const HostTensor& wcc = T(pfx + "conv1x1c.weight");        // (G, cin, 1)
                for (int jx = 0; jx < cin; ++jx) wp[(size_t)((kw - 1) * RC + jx) * GC + o] = (float)(gate_scale(o) * (double)wcc.data[(size_t)go * cin + jx]);
            }
        }
        const std::vector<float> ws = padded(T(pfx + "conv1x1_skip.weight").data.data(), Ka, Gha, K, RC);   // (K, G/2, 1): pass pp = skip channels [128 pp, 128 pp + 128)
        for (int pp = 0; pp < NK; ++pp) {
            put_row8(blob, st->o_ws + (((size_t)l * NK + pp) * 2 + 0) * rowsz, ws.data(), RC * pp, 0);
            put_row8(blob, st->o_ws + (((size_t)l * NK + pp) * 2 + 1) * rowsz, ws.data(), RC * pp, 1);
        }
        if (st->has_split)
            for (int half = 0; half < 2; ++half) put_rowq(blob, st->o_wss + ((size_t)l * 2 + half) * rowsz, ws.data(), half);
        const HostTensor& bs = T(pfx + "conv1x1_skip.bias");
        std::copy(bs.data.begin(), bs.data.end(), blob.begin() + st->o_bskip + (size_t)l * Kp);
        dil[l] = 1 << (l % per);
        hoff[l] = hist;
        hist += (kw - 1) * dil[l] * RC;
    }
    st->hist_floats = hist;
    // head: both 1x1s in the (o, q) register mapping.  Part j of the head (one workgroup per 128 hidden units) holds rows
    // [128 j, 128 j + 128) of the K x K matrix as NK column-block images and columns [128 j, 128 j + 128) of the O x K matrix
    // as two row images (rows i, and rows 128 + i for one-hot models), zero-padded
    const size_t imgsz = (size_t)8 * RT * 4;
    const std::vector<float> w1p = padded(T("last_conv_layers.1.weight").data.data(), Ka, Ka, K, K);
    const std::vector<float> w2p = padded(T("last_conv_layers.3.weight").data.data(), O, Ka, O, K);
    st->o_wh1 = alloc((size_t)NK * NK * imgsz);
    for (int part = 0; part < NK; ++part)
        for (int blk = 0; blk < NK; ++blk)
            put_image(blob, st->o_wh1 + ((size_t)part * NK + blk) * imgsz, w1p.data(), K, RC * part, K, RC * blk);
    st->o_bh1 = alloc(K);
    std::copy(T("last_conv_layers.1.bias").data.begin(), T("last_conv_layers.1.bias").data.end(), blob.begin() + st->o_bh1);
    st->o_wh2 = alloc((size_t)NK * 2 * imgsz);
    for (int part = 0; part < NK; ++part) {
        put_image(blob, st->o_wh2 + ((size_t)part * 2 + 0) * imgsz, w2p.data(), K, 0, O, RC * part);
        put_image(blob, st->o_wh2 + ((size_t)part * 2 + 1) * imgsz, w2p.data(), K, RC, O, RC * part);
    }
    st->o_bh2 = alloc(2 * RC);
    std::copy(T("last_conv_layers.3.bias").data.begin(), T("last_conv_layers.3.bias").data.end(), blob.begin() + st->o_bh2);
    // first_conv: (R, 1, 1) for scalar input; one-hot models: (R, cin1, 1) stored K-major [cin1][128] (row k = column k)
    const int cin1 = c.scalar_input ? 1 : O;
    st->cin1 = cin1;
    st->o_wf = alloc((size_t)cin1 * RC);
    {
        const HostTensor& wf = T("first_conv.weight");
        for (int r = 0; r < Ra; ++r)
            for (int k = 0; k < cin1; ++k) blob[st->o_wf + (size_t)k * RC + r] = wf.data[(size_t)r * cin1 + k];
    }
    st->o_bf = alloc(RC);
    std::copy(T("first_conv.bias").data.begin(), T("first_conv.bias").data.end(), blob.begin() + st->o_bf);
    // layer 0 evaluated by the head of scalar-input models (run_head): z_0 = (W_cur,0 w_first) x + W_cur,0 b_first + pre_0, folded in
    // double: [256] slopes then [256] offsets, rows = tanh channels 0..127 then sigmoid channels 0..127 of the padded geometry
    st->o_l0 = alloc(4 * GC);
    if (c.scalar_input) {
        const HostTensor& wc0 = T("conv_layers.0.conv.weight");          // (G, R, kw)
        const HostTensor& wf = T("first_conv.weight");                   // (R, 1, 1)
        const HostTensor& bf = T("first_conv.bias");
        for (int o = 0; o < GC; ++o) {
            const int go = gate_row(o);
            if (go < 0) continue;
            double a = 0.0, cc = 0.0;
            for (int ii = 0; ii < Ra; ++ii) {
                const double wv = (double)wc0.data[((size_t)go * Ra + ii) * kw + (kw - 1)];
                a += wv * (double)wf.data[ii];
                cc += wv * (double)bf.data[ii];
            }
            blob[st->o_l0 + o] = (float)(gate_scale(o) * a);
            blob[st->o_l0 + GC + o] = (float)(gate_scale(o) * cc);
            if (L >= 2) {                                                // N_1 = sqrt(.5) W_cur,1 (the tap workgroup adds c_1 = N_1 b_o,0)
                const HostTensor& wc1 = T("conv_layers.1.conv.weight");
                double a1 = 0.0, c1 = 0.0;
                for (int ii = 0; ii < Ra; ++ii) {
                    const double wv = rs * (double)wc1.data[((size_t)go * Ra + ii) * kw + (kw - 1)];
                    a1 += wv * (double)wf.data[ii];
                    c1 += wv * (double)bf.data[ii];
                }
                blob[st->o_l0 + 2 * GC + o] = (float)(gate_scale(o) * a1);
                blob[st->o_l0 + 3 * GC + o] = (float)(gate_scale(o) * c1);
            }
        }
    }
    *out = st;
    RING_HIP(hipMalloc((void**)&st->d_w, blob.size() * sizeof(float)));
    RING_HIP(hipMemcpy(st->d_w, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    RING_HIP(hipMalloc((void**)&st->d_dil, L * sizeof(int)));
    RING_HIP(hipMemcpy(st->d_dil, dil.data(), L * sizeof(int), hipMemcpyHostToDevice));
    RING_HIP(hipMalloc((void**)&st->d_histoff, L * sizeof(int)));
    RING_HIP(hipMemcpy(st->d_histoff, hoff.data(), L * sizeof(int), hipMemcpyHostToDevice));
    RING_HIP(hipHostMalloc((void**)&st->h_status, 64, hipHostMallocDefault));
    *st->h_status = 0;
    { const wnv_status cs = wnv_placement_census(device, &st->ncu, &st->n_xcd, &st->map_ok, err); if (cs != WNV_OK) return cs; }
    return WNV_OK;
}

wnv_status wnv_ring_wait(WnvRingState* st, std::string& err) {
    if (!st || !st->pending) return WNV_OK;
    st->pending = false;
    RING_HIP(hipStreamSynchronize(st->pending_stream));
    if (*st->h_status != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "ring kernel gave up waiting (code 0x%x: 0x1ss = activation into stage ss, 0x2ss = skip into stage ss, 0x300 = head, 0x800 = streamed noise tape)", *st->h_status);
        err = buf;
        return WNV_ERR_TIMEOUT;
    }
    return WNV_OK;
}

wnv_status wnv_ring_generate(WnvRingState** pst, int device, const wnv_config& c, const TensorStore& store,
                             const WnvGenArgs& ga, hipStream_t stream, std::string& err) {
    if (!*pst) {
        wnv_status st0 = build_state(pst, device, c, store, err);
        if (st0 != WNV_OK) { wnv_ring_destroy(*pst); *pst = nullptr; return st0; }
    }
    WnvRingState* st = *pst;
    constexpr int BSLICE = 64;                                       // utterances one launch pipelines through its rings
    if (ga.B > BSLICE) {
        // larger batches: slices of 64, one launch each (utterances are independent; the noise tape and the Philox stream are
        // addressed with the utterance's index in the whole call, so the samples do not depend on the slicing)
        const int cin1 = st->cin1, cin = st->cin, O = st->O;
        for (int b0 = 0; b0 < ga.B; b0 += BSLICE) {
            WnvGenArgs g = ga;
            g.B = std::min(BSLICE, ga.B - b0);
            g.b0 = ga.b0 + b0; g.noise_B = ga.noise_B > 0 ? ga.noise_B : ga.B;
            g.async = 0;
            if (ga.c_up) g.c_up = ga.c_up + (size_t)b0 * ga.T * cin;
            if (ga.initial) g.initial = ga.initial + (size_t)b0 * cin1;
            if (ga.teacher) g.teacher = ga.teacher + (size_t)b0 * ga.Tt * cin1;
            if (ga.zbias_bstride != 0 && !ga.seg_gid) g.zbias = ga.zbias + (size_t)b0 * ga.zbias_bstride;
            if (ga.out) g.out = ga.out + (size_t)b0 * cin1 * ga.T;
            if (ga.params_out) g.params_out = ga.params_out + (size_t)b0 * O * ga.T;
            if (ga.index_out) g.index_out = ga.index_out + (size_t)b0 * ga.T;
            if (ga.seg_start) { g.seg_start = ga.seg_start + (size_t)b0 * ga.T; g.seg_uid = ga.seg_uid + (size_t)b0 * ga.T; }
            if (ga.seg_gid) g.seg_gid = ga.seg_gid + (size_t)b0 * ga.T;
            const wnv_status s0 = wnv_ring_generate(pst, device, c, store, g, stream, err);
            if (s0 != WNV_OK) return s0;
        }
        return WNV_OK;
    }
    const int B = ga.B;
    { std::string perr; wnv_status pst = wnv_ring_wait(st, perr); if (pst != WNV_OK) { err = perr; return pst; } }
    const int ncu = st->ncu, NX = st->n_xcd;
    // The layout below (ring r = blocks r, r + NX, ... = one XCD; a per-XCD CU budget) needs the observed block -> XCD mapping and
    // the usual 8 XCDs; anything else (a partitioned GPU, a different dispatcher) is not a place for the persistent pipeline.
    if (!st->map_ok || NX != 8) {
        char buf[160];
        snprintf(buf, sizeof buf, "ring kernel: placement census found %d XCDs over %d CUs with the block -> XCD mapping %s (needs 8 XCDs, b %% 8)",
                 NX, ncu, st->map_ok ? "as assumed" : "NOT as assumed");
        err = buf;
        return WNV_ERR_UNSUPPORTED;
    }
    // one workgroup per CU, one ring per utterance slot; at most 8 rings (one per XCD) and never more than fit
    // Workgroups that must be co-resident: n_rings x (S + 1) ring workgroups + tap_parts x L tap workgroups, one per CU.
    // Block b runs on XCD b % 8 (observed; 32 CUs each), ring r occupies blocks r, r + 8, ... = XCD r, so the budget is per
    // XCD: (S + 1) + the tap workgroups that land there <= 32.  Tap workgroups first fill the block slots of unused ring
    // indices, the rest follow behind the rings (round-robin over the XCDs).  A tap workgroup serves 8 utterances per
    // pass (~4-5 us), so a second one per layer joins when there are more utterances and it fits.
    const int cus_per_xcd = ncu / 8;
    const int NK = st->K / RC;                                     // skip passes per stage = head parts per ring
    const int P = st->S + NK;                                      // workgroups (grid positions) of one ring
    // scalar-input models with 128 skip channels: the head evaluates layer 0 (run_head), position 0 of every ring exits at once
    bool head_l0 = st->cin1 == 1 && NK == 1 && st->S >= 2;
    { const char* e = wnv_knob("WNV_RING_L0"); if (e && e[0] == '0') head_l0 = false; }
    const int Plive = P - (head_l0 ? 1 : 0);
    auto rings_that_fit_of = [&](int parts, int B_) {
        for (int n = std::min(B_, 8); n >= 1; --n) {
            const int free_slots = (8 - n) * P;
            const int extra = std::max(0, parts * st->L - free_slots);
            if (Plive + (extra + 7) / 8 <= cus_per_xcd) return n;
        }
        return 0;
    };
    auto rings_that_fit = [&](int parts) { return rings_that_fit_of(parts, B); };
    // (round 6) THE TAP ROLE ON THE MATRIX PIPE (run_tap_mf, the _mf kernels) for the models whose layout has ONE tap workgroup per layer at
    // EVERY batch size -- rings too long for eight of them next to two parts: more than 24 stages --: there the single workgroup's passes are the
    // step's cadence beyond 16 utterances (up to 8 passes of 5.7 us per step): the 30-layer models gain 5 % at 32 utterances, 14-15 % at 48 / 64
    // and 24 % on BASELINE configs[3]'s 64-utterance job.  The choice depends on the MODEL, never on the batch size: pre_l is the same bits in
    // every launch of a model (seed determinism).  NOT the 24-layer models (two parts: every form of it measured slower at 40 / 48 utterances
    // and on the packed jobs), NOT K = 512 (one part too, but its stages' skip passes bound it as much: measured slower):
    // profiles/r06_tap_waves.txt #7, profiles/r06_tap_mfg_raw.txt.
    const bool mf_model = (st->kpre + 15) / 16 <= TAP_NJR + TAP_NJL && NK != 4
                          && rings_that_fit_of(2, 64) < 8 && rings_that_fit_of(1, 64) > rings_that_fit_of(2, 64);
    int tap_parts = B > TB ? 2 : 1;
    int tb = TB;                                                   // utterances per tap pass
    // WNV_RING_TAP=<parts>,<utterances per pass>: measurement knob (profiles/r04_tap_parts.txt)
    if (const char* e = wnv_knob("WNV_RING_TAP")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= 4 && b >= 1 && b <= TB) { tap_parts = a; tb = b; }
    }
    int n_rings = rings_that_fit(tap_parts);
    if (tap_parts == 2 && n_rings < std::min(B, 8) && rings_that_fit(1) > n_rings) {   // rather more rings than the second tap part
        tap_parts = 1;
        n_rings = rings_that_fit(1);
    }
    if (n_rings < 1) { err = "ring kernel: not enough CUs per XCD for one ring + the tap workgroups"; return WNV_ERR_UNSUPPORTED; }
    // SPLIT RINGS (run_stage_split): two CUs per layer, 4 rings over 2 XCDs each -- up to 8 utterances of a scalar-input model with 128
    // skip channels whose head evaluates layer 0.  XCD 2r: head + both halves of stages 1 .. sA; XCD 2r + 1: stages sA + 1 .. S - 1;
    // the tap workgroups follow behind the rings, round-robin over the XCDs.
    // MEASURED AND NOT THE DEFAULT (round 3, profiles/r03_ring_split_*): the mat-vec phases shrink as predicted (chain phase 301 -> 226 ns,
    // N phase 290 -> 215 ns) but every hand-off now waits for the slower of two CUs, twice as many pollers share the lines, and the
    // ring crosses XCDs (u, the residual partials and the skip sums: 0.5-0.8 us each): 441-453 against 472-479 kSamples/s on one box.
    // WNV_RING_SPLIT=1 selects it (parity-tested: tests/test_gpu_ring.py::test_split_rings_vs_oracle_and_generic).
    bool split = false;
    { const char* e = wnv_knob("WNV_RING_SPLIT"); if (e && e[0] == '1') split = head_l0 && st->has_split && st->S >= 3 && B <= 16; }
    int sA = 0, max_slots = 0;
    if (split) {
        sA = (2 * (st->S - 1) + 1) / 4;
        max_slots = std::max(1 + 2 * sA, 2 * (st->S - 1 - sA));
        if (max_slots + (st->L + 7) / 8 > cus_per_xcd) split = false;
    }
    if (split) { n_rings = std::min(B, 4); tap_parts = 1; }
    const int upr = (B + n_rings - 1) / n_rings;
    // Block b lands on XCD b % 8 (observed): a ring stride of 8 keeps every workgroup of a ring on one XCD, which the
    // kernel verifies at run time before it uses the same-XCD hand-off.  The grid always has 8 ring slots; the
    // workgroups of unused slots exit at once and free their CUs (for the tap workgroups, which are dispatched last).
    const int rstride = 8;
    RingParams p{};
    p.n_rings = n_rings; p.rstride = rstride; p.S = st->S; p.L = st->L; p.B = B; p.T = (int)ga.T; p.Tt = (int)ga.Tt; p.upr = upr;
    p.NH = NK; p.Op = 256;
    p.K = st->K; p.Kp = st->Kp; p.O = st->O; p.cin = st->cin; p.kw = st->kw; p.kpre = st->kpre; p.nz = ga.nz;
    p.dist = c.output_distribution;
    p.cin1 = st->cin1; p.softmax = ga.softmax; p.quantize = ga.quantize; p.index_out = ga.index_out;
    p.head_l0 = head_l0 ? 1 : 0;
    p.split = split ? 1 : 0; p.sA = sA; p.qh = split ? 2 : 1;
    p.pstride = std::max(GC, st->Kp);
    p.hist_floats = st->hist_floats;
    p.skip_scale = (float)std::sqrt(1.0 / st->L);
    { const char* e = wnv_knob("WNV_RING_FAST"); p.allow_fast = !(e && e[0] == '0'); }
    const float* w = st->d_w;
    p.w2img = w + st->o_w2; p.wnimg = w + st->o_wn; p.cvec = w + st->o_cvec; p.woimg = w + st->o_wo; p.bo = w + st->o_bo; p.wpre = w + st->o_wpre;
    p.wsimg = w + st->o_ws; p.bskip = w + st->o_bskip; p.wh1img = w + st->o_wh1; p.bh1 = w + st->o_bh1;
    p.wh2img = w + st->o_wh2; p.bh2 = w + st->o_bh2; p.wfirst = w + st->o_wf; p.bfirst = w + st->o_bf; p.l0vec = w + st->o_l0;
    p.w2s = w + st->o_w2s; p.wns = w + st->o_wns; p.wos = w + st->o_wos; p.wss = w + st->o_wss;
    p.zbias = ga.zbias; p.zbias_bstride = ga.zbias_bstride;
    p.zb_ld = (c.gate_channels + 3) & ~3; p.gh = c.gate_channels / 2;
    p.lay_dil = st->d_dil; p.lay_histoff = st->d_histoff;
    // state: [status 64 B][placement table 4 KiB][xmail B*(S+1)*128 u64][hmail B*2*(S+1)*128 u64][gmail, the same][smail B*(S+1)*Kp u64][omail B*NK*256 u64]
    //        [zmail B*256 u64][fmail B*L*2*128 u64][pmail B*L*2*256 u64][hist B*hist_floats f32]
    const size_t head_bytes = 64 + 4096;                           // status word, placement table
    const size_t n_h = (size_t)B * (st->S + 1) * RC, n_s = (size_t)B * (st->S + 1) * st->Kp;
    const size_t n_f = (size_t)B * st->L * 2 * RC, n_p = (size_t)B * st->L * 2 * GC;   // stage <-> tap-workgroup records (granules), two slots by step parity
    const size_t n_o = (size_t)B * NK * p.Op;                      // partial head outputs of parts 1 .. NK-1
    const size_t n_z = (size_t)B * GC;                             // N_1 h_0 from the head (head_l0)
    const size_t qh = split ? 2 : 1;                               // split rings: two partial vectors per residual / skip slot
    const size_t mail_bytes = (n_h + 2 * n_h * qh + 2 * n_h + n_s * qh + n_o + n_z + n_f + n_p) * sizeof(u64);
    const size_t hist_bytes = ((size_t)B * st->hist_floats + RC) * sizeof(float);    // (+ one row of zeros: RingParams::zero_row)
    const size_t bytes = head_bytes + mail_bytes + hist_bytes;
    bool fresh = false;
    if (bytes > st->state_cap) {
        if (st->d_state) { RING_HIP(guard_free(st->d_state)); st->d_state = nullptr; st->state_cap = 0; }
        RING_HIP(guard_alloc(&st->d_state, bytes));
        st->state_cap = bytes;
        fresh = true;
    }
    char* base = (char*)st->d_state;
    // Mailbox tags are unique across launches (tag_base + t + 1), so the mailboxes are zeroed only when the buffer is
    // new, when its layout changes, or before the 32-bit tag would wrap; the history rings are zeroed every call
    // (= clear_buffer, wavenet.py:241) together with the status word and the placement table.
    if (fresh || mail_bytes != st->mail_bytes || (unsigned long long)st->tag_next + (unsigned long long)ga.T + 2ull > 0xFFFFFFF0ull) {
        RING_HIP(hipMemsetAsync(base, 0, head_bytes + mail_bytes, stream));
        st->tag_next = 0;
        st->mail_bytes = mail_bytes;
    } else {
        RING_HIP(hipMemsetAsync(base, 0, head_bytes, stream));
    }
    RING_HIP(hipMemsetAsync(base + head_bytes + mail_bytes, 0, hist_bytes, stream));
    p.tag_base = st->tag_next;
    st->tag_next += (unsigned)ga.T + 1u;
    p.status = (unsigned int*)base;
    p.xcc = (unsigned int*)(base + 64);
    p.xmail = (u64*)(base + head_bytes);
    p.hmail = p.xmail + n_h;
    p.gmail = p.hmail + 2 * n_h * qh;
    p.smail = p.gmail + 2 * n_h;
    p.omail = p.smail + n_s * qh;
    p.zmail = p.omail + n_o;
    p.fmail = p.zmail + n_z;
    p.pmail = p.fmail + n_f;
    p.hist = (float*)(p.pmail + n_p);
    p.zero_row = p.hist + (size_t)B * st->hist_floats;
    p.c_up = ga.c_up; p.initial = ga.initial; p.teacher = ga.teacher; p.noise = ga.noise; p.seed = ga.seed;
    p.b0 = ga.b0; p.noise_B = ga.noise_B > 0 ? ga.noise_B : B;
    p.noise_ready = ga.noise_ready;
    p.seg_start = ga.seg_start; p.seg_uid = ga.seg_uid; p.seg_gid = ga.seg_gid;
    p.out = ga.out; p.params_out = ga.params_out;
    // LDS: the stage carve is the larger one
    // tap workgroups: K rows per wave (a multiple of 4), first in VGPRs, then in LDS, the remainder streams from L2
    p.kper = (((st->kpre + RW - 1) / RW) + 3) & ~3;
    p.kreg_rows = std::min(p.kper, (ga.seg_start && NK != 2 && WNV_PACKED_SPEC) ? KR_PACKED_SPEC : KR_MAX);      // (= run_tap's KR for the instantiation launched below)
    p.klds_rows = std::min(p.kper - p.kreg_rows, KL_MAX);                      // multiples of 4 (kper is one)
    while (p.klds_rows > 0 && tap_lds_floats(p.kper, p.klds_rows) * sizeof(float) > 158 * 1024) p.klds_rows -= 4;
    p.tap_nj = (st->kpre + 15) / 16;                                          // K groups of 16 rows (16 tap_nj <= 8 kper: the zero-padded input row)
    p.tap_mfma = (mf_model && !split && tap_parts == 1) ? 1 : 0;
    if (p.tap_mfma) {                                                          // (two float4 per lane and K group beyond the registers' sixteen, in LDS)
        const int rows = 2 * std::max(p.tap_nj - TAP_NJR, 0);
        if (tap_lds_floats_mf(p.kper, rows) * sizeof(float) <= 158 * 1024) p.klds_rows = rows;   // (= the stride of a wave's block of TapLds::wl)
        else p.tap_mfma = 0;
    }
    p.ring_blocks = split ? 8 * max_slots : rstride * P;
    const size_t lds = std::max(std::max(std::max(stage_lds_floats(NK), head_lds_floats(NK)), p.tap_mfma ? tap_lds_floats_mf(p.kper, p.klds_rows) : tap_lds_floats(p.kper, p.klds_rows)),
                                st->cin1 > 1 ? cat_lds_floats(NK) : (size_t)0) * sizeof(float);
    if (lds > 160 * 1024) { err = "ring kernel needs too much LDS"; return WNV_ERR_UNSUPPORTED; }
    // more than four utterances per ring (40+ per GPU): the stages' occupancy per utterance bounds the step -> their throughput
    // instantiation.  Up to four a ring is still bound by the chain's latency and the plain prologue is faster (same box, B = 16 / 32:
    // 1002 / 2014 against 970 / 1918 kSamples/s; B = 48: 2616 against 2742)
    // (K = 512: a stage is busy ~3.8 us per utterance -- four skip passes, two of them streamed --, so its occupancy counts from two utterances per
    //  ring on: the throughput instantiation measured +1-2 % at B = 16 / 32 (WNV_RING_MODE knob; both settings in profiles/r04_final_numbers.txt))
    int mode = ga.seg_start ? 2 : (upr > 4 || (NK == 4 && upr >= 2)) ? 1 : 0;
    if (const char* e = wnv_knob("WNV_RING_MODE")) { if (mode != 2 && (e[0] == '0' || e[0] == '1')) mode = e[0] - '0'; }   // measurement knob
    if (split && mode == 2) { err = "ring kernel: packed slots and split rings do not combine"; return WNV_ERR_UNSUPPORTED; }
#define WNV_PICK(NKV, L0V) (mode == 2 ? (const void*)wnv_ring_kernel<NKV, L0V, 2> : mode == 1 ? (const void*)wnv_ring_kernel<NKV, L0V, 1> : (const void*)wnv_ring_kernel<NKV, L0V, 0>)
#define WNV_PICK_MF(NKV, L0V) (mode == 2 ? (const void*)wnv_ring_kernel_mf<NKV, L0V, 2> : mode == 1 ? (const void*)wnv_ring_kernel_mf<NKV, L0V, 1> : (const void*)wnv_ring_kernel_mf<NKV, L0V, 0>)
    const void* kfn = split ? (const void*)wnv_ring_kernel_split
                    : p.tap_mfma ? (NK == 1 ? (head_l0 ? WNV_PICK_MF(1, true) : WNV_PICK_MF(1, false)) : WNV_PICK_MF(2, false))
                    : NK == 1 ? (head_l0 ? WNV_PICK(1, true) : WNV_PICK(1, false))
                    : NK == 2 ? WNV_PICK(2, false)
                              : (mode == 2 ? (const void*)wnv_ring_kernel_k512<2> : mode == 1 ? (const void*)wnv_ring_kernel_k512<1> : (const void*)wnv_ring_kernel_k512<0>);
#undef WNV_PICK
#undef WNV_PICK_MF
    RING_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    p.tap_parts = tap_parts; p.tb = tb;
    const int grid = split ? p.ring_blocks + tap_parts * st->L : p.ring_blocks + std::max(0, tap_parts * st->L - (8 - n_rings) * P);
    if (p.ring_blocks > ncu) { err = "ring kernel: too many layers for one ring per XCD"; return WNV_ERR_UNSUPPORTED; }
    // every LIVE workgroup must be resident at once (they wait for each other): ask the runtime how many fit a CU with this
    // register / LDS footprint instead of assuming one, and compare with what stays alive (the workgroups of unused ring slots
    // exit at once)
    {
        int per_cu = 0;
        RING_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, RT, lds));
        const int live = (split ? n_rings * (1 + 2 * (st->S - 1)) : n_rings * Plive) + tap_parts * st->L;
        if (per_cu < 1 || live > ncu * per_cu) {
            char buf[160];
            snprintf(buf, sizeof buf, "ring kernel: %d workgroups must be co-resident but the device holds %d (%d CUs x %d per CU)", live,
                     ncu * std::max(per_cu, 0), ncu, per_cu);
            err = buf;
            return WNV_ERR_UNSUPPORTED;
        }
    }
    // optional timeline (WNV_RING_TRACE=<file>): wall-clock stamps of utterance 0 for 8 steps in mid-run
    const char* trace_path = wnv_knob("WNV_RING_TRACE");
    unsigned long long* d_trace = nullptr;
    const int trace_n = 8;
    size_t trace_words = 0;
#ifndef WNV_FINE_TRACE
    if (trace_path && *trace_path) {
        static bool told = false;
        if (!told) fprintf(stderr, "[wnv] WNV_RING_TRACE needs a trace build of the library (python -m wavenet_vocoder_amd.build --out <lib> --flags -DWNV_FINE_TRACE, then WNV_LIB=<lib>); ignored\n");
        told = true;
        trace_path = nullptr;
    }
#endif
    if (trace_path && *trace_path && p.T > 64) {
        trace_words = (size_t)trace_n * upr * (st->S + 1) * TRW + (size_t)trace_n * TRW;
        RING_HIP(guard_alloc((void**)&d_trace, trace_words * sizeof(unsigned long long)));
        RING_HIP(hipMemsetAsync(d_trace, 0, trace_words * sizeof(unsigned long long), stream));
        p.trace = d_trace; p.trace_t0 = std::min(p.T / 2, 1000); p.trace_n = trace_n;
        p.trace_tap = d_trace + (size_t)trace_n * upr * (st->S + 1) * TRW;
    }
    {
        void* kargs[] = {(void*)&p};
        RING_HIP(hipLaunchKernel(kfn, dim3(grid), dim3(RT), kargs, lds, stream));
    }
    RING_HIP(hipGetLastError());
    // a bounded spin that gave up must reach the caller: the status word follows the kernel into pinned host memory; the call
    // waits for it here unless the caller asked for an asynchronous launch (then wnv_ring_wait reports it)
    RING_HIP(hipMemcpyAsync(st->h_status, p.status, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    if (ga.async && !d_trace) {
        st->pending = true;
        st->pending_stream = stream;
        return WNV_OK;
    }
    RING_HIP(hipStreamSynchronize(stream));
#ifdef WNV_GUARD
    if (!guard_check(st->d_state, st->state_cap, "ring state block")) { err = "WNV_GUARD: the ring state block's red zone is overwritten"; return WNV_ERR_HIP; }
    { static int told = 0; if (!told++) fprintf(stderr, "[wnv guard] red zones of %zu bytes around the ring state block (%zu bytes) checked after every launch\n", GUARD_BYTES, st->state_cap); }
#endif
    const unsigned int status = *st->h_status;
    if (d_trace) {
        std::vector<unsigned long long> tr(trace_words);
        RING_HIP(hipMemcpy(tr.data(), d_trace, trace_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#ifdef WNV_GUARD
        if (!guard_check(d_trace, trace_words * sizeof(unsigned long long), "timeline buffer")) { (void)guard_free(d_trace); err = "WNV_GUARD: the timeline buffer's red zone is overwritten"; return WNV_ERR_HIP; }
#endif
        (void)guard_free(d_trace);
        if (FILE* f = fopen(trace_path, "w")) {
            fprintf(f, "# step pos(S=head) stamps in ns relative to the head's send of the first traced step (100 MHz wall clock); utterance 0 of ring 0\n");
            const unsigned long long t00 = tr[(((size_t)0 * upr + 0) * (st->S + 1) + st->S) * TRW + 0];
            for (int tt = 0; tt < trace_n; ++tt)
                for (int pos = 0; pos <= st->S; ++pos) {
                    fprintf(f, "%d %d", p.trace_t0 + tt, pos);
                    for (int k = 0; k < TRW; ++k) {
                        const unsigned long long v = tr[(((size_t)tt * upr + 0) * (st->S + 1) + pos) * TRW + k];
                        fprintf(f, " %lld", v ? (long long)(v - t00) * 10 : -1LL);
                    }
                    fprintf(f, "\n");
                }
            for (int tt = 0; tt < trace_n; ++tt) {                   // first pass of layer 0's tap workgroup (part 0): "#tap step stamps"
                fprintf(f, "#tap %d", p.trace_t0 + tt);
                for (int k = 0; k < TRW; ++k) {
                    const unsigned long long v = tr[(size_t)trace_n * upr * (st->S + 1) * TRW + (size_t)tt * TRW + k];
                    if (k == 15) fprintf(f, " %lld", (long long)v);      // (not a stamp: two bits per wave and pass -- how each wave got its h record)
                    else
                    fprintf(f, " %lld", v ? (long long)(v - t00) * 10 : -1LL);
                }
                fprintf(f, "\n");
            }
            // the other utterances of ring 0 (when the ring carries several): "#u j step pos stamps" on the same clock
            for (int j = 1; j < upr; ++j)
                for (int tt = 0; tt < trace_n; ++tt)
                    for (int pos = 0; pos <= st->S; ++pos) {
                        fprintf(f, "#u %d %d %d", j, p.trace_t0 + tt, pos);
                        for (int k = 0; k < TRW; ++k) {
                            const unsigned long long v = tr[(((size_t)tt * upr + j) * (st->S + 1) + pos) * TRW + k];
                            fprintf(f, " %lld", v ? (long long)(v - t00) * 10 : -1LL);
                        }
                        fprintf(f, "\n");
                    }
            fclose(f);
        }
    }
#ifdef WNV_DBG_MARK
    if (status == 0 && wnv_knob("WNV_RING_MISS_COUNT")) {             // (diagnostic build) the throughput prologue's first look: what was not there yet
        std::vector<unsigned> mk(1024);
        (void)hipMemcpy(mk.data(), p.xcc, 4096, hipMemcpyDeviceToHost);
        fprintf(stderr, "[wnv miss] of %d utterance-steps per stage, ring 0, by stage: pre / h_{l-2} / q not there at the first look:", p.T * upr);
        for (int k = 0; k < 256; k += p.rstride) if (mk[256 + k] | mk[512 + k] | mk[768 + k]) fprintf(stderr, " %d:%u/%u/%u", k / p.rstride, mk[256 + k], mk[512 + k], mk[768 + k]);
        fprintf(stderr, "\n");
    }
#endif
    if (status != 0) {
        if (wnv_knob("WNV_RING_DEBUG_DUMP")) {                        // (failure path only) where did the records of utterance 0 get to?
            std::vector<u64> f(n_f / B), q(n_p / B);
            (void)hipMemcpy(f.data(), p.fmail, f.size() * sizeof(u64), hipMemcpyDeviceToHost);
            (void)hipMemcpy(q.data(), p.pmail, q.size() * sizeof(u64), hipMemcpyDeviceToHost);
#ifdef WNV_DBG_MARK
            {
                std::vector<unsigned> mk(1024);
                (void)hipMemcpy(mk.data(), p.xcc, 4096, hipMemcpyDeviceToHost);
                fprintf(stderr, "[wnv dump] tag_base %u; pre-receive marks by block (phase << 16 | tag; last first-granule tag seen):", p.tag_base);
                for (int k = 0; k < 256; ++k) if (mk[256 + k]) fprintf(stderr, " %d:%x/%x", k, mk[256 + k], mk[512 + k]);
                fprintf(stderr, "\n[wnv dump] block 16, per lane: tags seen | addr a | addr b (pmail = %llx):", (unsigned long long)p.pmail);
                for (int k = 0; k < 64; ++k) fprintf(stderr, " %d:%08x|%x|%x", k, mk[768 + k], mk[832 + k], mk[896 + k]);
                fprintf(stderr, "\n");
            }
#endif
            for (int l = 0; l < st->L; ++l)
                for (int par = 0; par < 2; ++par) {
                    fprintf(stderr, "[wnv dump] layer %2d slot %d  h tags (step + 1):", l, par);
                    for (int k : {0, 1, 2, 63, 64, 127}) fprintf(stderr, " %lld", (long long)(f[((size_t)l * 2 + par) * RC + k] >> 32) - (long long)p.tag_base);
                    fprintf(stderr, "   pre tags:");
                    for (int k : {0, 1, 2, 3, 4, 127, 128, 254, 255}) fprintf(stderr, " %lld", (long long)(q[((size_t)l * 2 + par) * GC + k] >> 32) - (long long)p.tag_base);
                    fprintf(stderr, "\n");
                }
            fprintf(stderr, "[wnv dump] layer 2 slot 0, pre granules 22..33 as raw u64:");
            for (int k = 22; k < 34; ++k) fprintf(stderr, " %d:%016llx", k, q[((size_t)2 * 2 + 0) * GC + k]);
            fprintf(stderr, "\n");
        }
        char buf[128];
        snprintf(buf, sizeof buf, "ring kernel gave up waiting (code 0x%x: 0x1ss = activation into stage ss, 0x2ss = skip into stage ss, 0x300 = head)", status);
        err = buf;
        return WNV_ERR_TIMEOUT;
    }
    return WNV_OK;
}
