// ubench_phase.hip -- what a ring stage's CHAIN PHASE costs, piece by piece, and what hand-scheduling can take out of it (VERDICT r04
// item 2: "hand-schedule the chain phase in the microbenchmark first; gate: <= 250 ns per phase").
//
// The phase as wnv_ring.hip's run_stage runs it on the chain (ISA of wnv_ring_kernel<1, true, 0>, round 5): barrier, 4 x ds_read_b128 of
// the lane's K slice (+ one ds_read_b64 of zin), 64 v_pk_fma_f32 on register-resident weights -- the compiler already starts them on
// the first 16 bytes (s_waitcnt lgkmcnt(4), (2), (1), (0)) --, 8 adds, a 3-level DPP reduce-scatter, the gate (2 v_exp, 1 v_rcp) and
// one granule store by every second lane.  One workgroup, 512 threads; the four chain waves do the work, all eight meet in the barrier.
//
// Variants (ns per phase incl. the barrier; 20 000 back-to-back phases, s_memrealtime):
//   0  the phase as the compiler schedules it (= scripts/ubench_load.hip mode 0)
//   1  no LDS read (x from registers)                     -> what the LDS latency costs on the chain
//   2  no reduce, no gate (sum of the accumulators stored) -> what the tail costs
//   3  barrier + store only                                -> the floor of the loop itself
//   4  FMAs only between barriers (no read, no tail)       -> the FMA phase alone
//   5  tail hand-scheduled: the eight rows' reduce steps interleaved pairwise, exp arguments formed before the last DPP level
//      returns, both v_exp issued back to back, 1/((1+e1)(1+e2)) as one v_rcp                  (inline asm, one block)
//   6  4-lane K split: 32 floats of x per lane, 4 rows per lane, two quad_perm levels, no row_half_mirror (8 x ds_read_b128)
//   7  as 0 with s_setprio 3 on the chain waves
//   8  as 0, the four reads as eight ds_read_b64 (first FMA after 8 bytes)
//   9  as 5 + the zin addends read with the first x read and added into the accumulators' init (no add behind the reduce)
// DEPTH of the dependent tail (round 5: variants 5 / 9 said that instruction ORDER buys nothing and every dependent level ~7 ns):
//  10  row-pair accumulators: a packed FMA holds two ROWS against a broadcast x (op_sel), 4 accumulators x 16 k -- the x + y add level is gone
//  11  10 + the last DPP level as ONE v_add_f32_dpp in every lane (the compiler splits it into v_mov_dpp + v_add when only the writer lanes gate)
//  12  11 + zin in the accumulators' init
//  13  12 + gate as (2 rcp(1 + e1) - 1) rcp(1 + e2) with the exp2 scales folded into the weights: exp -> add -> rcp -> fma -> mul
//  14  10 + the gate of 13, zin added behind the reduce = WHAT wnv_ring.hip RUNS since round 5 (WNV_PHASE2)
//  15  14 with zin added to one accumulator pair in the MIDDLE of the FMA sequence (a level hidden under the other accumulators' FMAs)
// hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_phase.bin scripts/ubench_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp_fold(float keep, float send) {
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gate(float a, float g) {                 // wnv_ring.hip fast_gate: tanh(a) * sigmoid(g), 2 exp + 1 rcp
    const float e1 = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(a)), e2 = __builtin_amdgcn_exp2f(-1.4426950408889634f * g);
    const float r = __builtin_amdgcn_rcpf((1.0f + e1) * (1.0f + e2));
    return copysignf((1.0f - e1) * r, a);
}

template <int V>
__global__ void __launch_bounds__(512) k(int reps, u64* mail, u64* result, float* sink) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid & 7;
    f2 w[8][8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) w[s][c] = f2{0.001f * (float)((tid + s) & 31) - 0.01f, 0.002f * (float)((lane - c) & 15) - 0.01f};
    for (int i = tid; i < 4096; i += 512) lds[i] = 0.01f * (float)(i & 63) - 0.3f;
    __syncthreads();
    float accum = 0.f;
    u64 t0 = 0, t1 = 0;
    for (int r = -8; r < reps; ++r) {
        if (r == 0) { __syncthreads(); t0 = __builtin_amdgcn_s_memrealtime(); }
        __syncthreads();
        if (tid < 256) {                                                  // the four chain waves
            if constexpr (V == 7) __builtin_amdgcn_s_setprio(3);
            const float* xs = lds + 20 * ks + ((r & 1) ? 256 : 0);
            float u;
            if constexpr (V == 3) {
                u = accum + 1.0f;
            } else if constexpr (V == 6) {
                // 4-lane K split: lane (og4 = tid >> 2, k4 = tid & 3) contracts 32 floats for 4 rows -- same 64 packed FMAs
                const float* xq = lds + 36 * (tid & 3) + ((r & 1) ? 256 : 0);
                float x[32];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(xq + 4 * q);
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
                f2 acc[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[s] = f2{accum * 1e-30f, 0.f};
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(w[2 * s + (c >> 3)][c & 7], f2{x[2 * c], x[2 * c + 1]}, acc[s]);
                float q4[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) q4[s] = acc[s].x + acc[s].y;
                const float n0 = dpp_fold<0x4E>(q4[0], q4[2]), n1 = dpp_fold<0x4E>(q4[1], q4[3]);      // lane j <-> j ^ 2
                const float a = dpp_fold<0xB1>(n0, n0), g = dpp_fold<0xB1>(n1, n1);                      // lane j <-> j ^ 1
                u = gate(a + xq[33], g + xq[34]);
            } else if constexpr (V >= 10) {
                float x[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(xs + 4 * q);
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
                const float2 z = *reinterpret_cast<const float2*>(lds + 1024 + 2 * (tid >> 1));
                // row pairs: acc[p] = rows (2p, 2p + 1) of this lane's eight; w[p][c] and w[p + 4][c] stand in for the 16 k of a pair
                f2 acc[4];
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) acc[pq] = f2{accum * 1e-30f, 0.f};
                if constexpr (V == 12 || V == 13) { acc[0].x += z.x; acc[0].y += z.y; }
#pragma unroll
                for (int c = 0; c < 16; ++c) {
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) acc[pq] = __builtin_elementwise_fma(w[pq + 4 * (c >> 3)][c & 7], f2{x[c], x[c]}, acc[pq]);
                    if constexpr (V == 15) { if (c == 9) acc[0] += ((tid & 1) == 0) ? f2{z.x, z.y} : f2{0.f, 0.f}; }
                }
                // reduce-scatter over the 8 K lanes: slots 0..7 = acc[0].x, acc[0].y, acc[1].x, ... ; level 1 sends slots 4-7, level 2 slots 2-3
                const float n0 = dpp_fold<0x141>(acc[0].x, acc[2].x), n1 = dpp_fold<0x141>(acc[0].y, acc[2].y);
                const float n2 = dpp_fold<0x141>(acc[1].x, acc[3].x), n3 = dpp_fold<0x141>(acc[1].y, acc[3].y);
                const float m0 = dpp_fold<0x4E>(n0, n2), m1 = dpp_fold<0x4E>(n1, n3);
                float a, g;
                if constexpr (V == 10) {
                    a = dpp_fold<0xB1>(m0, m0) + z.x; g = dpp_fold<0xB1>(m1, m1) + z.y;
                } else {
                    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : "=&v"(a), "=&v"(g) : "v"(m0), "v"(m1));
                    if constexpr (V == 11 || V == 14) { a += z.x; g += z.y; }
                }
                if constexpr (V >= 13) {
                    // (the weights carry -2 log2 e / -log2 e: a, g ARE the exp2 arguments) tanh(a') sigmoid(g') = (2 r1 - 1) r2
                    const float e1 = __builtin_amdgcn_exp2f(a), e2 = __builtin_amdgcn_exp2f(g);
                    const float r1 = __builtin_amdgcn_rcpf(1.0f + e1), r2 = __builtin_amdgcn_rcpf(1.0f + e2);
                    u = __builtin_fmaf(2.0f, r1, -1.0f) * r2;
                } else {
                    u = gate(a, g);
                }
            } else {
                float x[16];
                if constexpr (V == 1 || V == 4) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) x[q] = accum + 0.01f * (float)q;
                } else if constexpr (V == 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float2 v = *reinterpret_cast<const float2*>(xs + 2 * q);
                        x[2 * q] = v.x; x[2 * q + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(xs + 4 * q);
                        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                    }
                }
                float2 z = make_float2(0.f, 0.f);
                if constexpr (V != 1 && V != 4) z = *reinterpret_cast<const float2*>(lds + 1024 + 2 * (tid >> 1));
                f2 acc[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[s] = f2{accum * 1e-30f, 0.f};
                if constexpr (V == 9) { acc[0].x += z.x; acc[1].x += z.y; }     // (the row that ends up in this lane: see the reduce)
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc[s] = __builtin_elementwise_fma(w[s][c], f2{x[2 * c], x[2 * c + 1]}, acc[s]);
                if constexpr (V == 2 || V == 4) {
                    u = (acc[0].x + acc[1].y) + (acc[2].x + acc[3].y) + (acc[4].x + acc[5].y) + (acc[6].x + acc[7].y);
                } else if constexpr (V == 5 || V == 9) {
                    // ---- hand-scheduled tail: one asm block.  q[s] = acc[s].x + acc[s].y for the 8 rows (rows 0-3 plain adds, 4-7 as
                    //      the DPP source of level 1: v_add_f32_dpp takes ONE dpp operand, so the x + y of rows 4-7 must exist first).
                    //      Levels: n_s = q_s + mirror(q_{s+4}) (s = 0..3), m_0 = n_0 + perm2(n_2), m_1 = n_1 + perm2(n_3),
                    //      a = m_0 + perm1(m_0) (+ z.x), g = m_1 + perm1(m_1) (+ z.y); the two rows' chains are interleaved instruction
                    //      by instruction (DPP results need 2 wait states before a DPP read: the other row's op fills them),
                    //      then e1 = exp2(-2 log2e |a|), e2 = exp2(-log2e g) back to back, one rcp.
                    float a, g, e1, e2, t1v, t2v;
                    float q0, q1, q2, q3, q4, q5, q6, q7;
                    asm volatile(
                        "v_add_f32 %[q4], %[a4x], %[a4y]\n\t"
                        "v_add_f32 %[q5], %[a5x], %[a5y]\n\t"
                        "v_add_f32 %[q6], %[a6x], %[a6y]\n\t"
                        "v_add_f32 %[q7], %[a7x], %[a7y]\n\t"
                        "v_add_f32 %[q0], %[a0x], %[a0y]\n\t"
                        "v_add_f32 %[q1], %[a1x], %[a1y]\n\t"
                        "v_add_f32 %[q2], %[a2x], %[a2y]\n\t"
                        "v_add_f32 %[q3], %[a3x], %[a3y]\n\t"
                        "v_add_f32_dpp %[q0], %[q4], %[q0] row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32_dpp %[q1], %[q5], %[q1] row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32_dpp %[q2], %[q6], %[q2] row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32_dpp %[q3], %[q7], %[q3] row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "s_nop 0\n\t"
                        "v_add_f32_dpp %[q0], %[q2], %[q0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32_dpp %[q1], %[q3], %[q1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "s_nop 1\n\t"
                        "v_add_f32_dpp %[a], %[q0], %[q0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32_dpp %[g], %[q1], %[q1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_add_f32 %[a], %[a], %[zx]\n\t"
                        "v_add_f32 %[g], %[g], %[zy]\n\t"
                        "v_mul_f32_e64 %[e1], %[c1], |%[a]|\n\t"
                        "v_mul_f32_e64 %[e2], %[c2], %[g]\n\t"
                        "v_exp_f32 %[e1], %[e1]\n\t"
                        "v_exp_f32 %[e2], %[e2]\n\t"
                        "s_nop 1\n\t"
                        "v_add_f32 %[t1], 1.0, %[e1]\n\t"
                        "v_add_f32 %[t2], 1.0, %[e2]\n\t"
                        "v_sub_f32 %[e1], 1.0, %[e1]\n\t"
                        "v_mul_f32 %[t1], %[t1], %[t2]\n\t"
                        "v_rcp_f32 %[t1], %[t1]\n\t"
                        "s_nop 1\n\t"
                        "v_mul_f32 %[e1], %[e1], %[t1]\n\t"
                        : [a] "=&v"(a), [g] "=&v"(g), [e1] "=&v"(e1), [e2] "=&v"(e2), [t1] "=&v"(t1v), [t2] "=&v"(t2v),
                          [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [q4] "=&v"(q4), [q5] "=&v"(q5), [q6] "=&v"(q6), [q7] "=&v"(q7)
                        : [a0x] "v"(acc[0].x), [a0y] "v"(acc[0].y), [a1x] "v"(acc[1].x), [a1y] "v"(acc[1].y), [a2x] "v"(acc[2].x), [a2y] "v"(acc[2].y),
                          [a3x] "v"(acc[3].x), [a3y] "v"(acc[3].y), [a4x] "v"(acc[4].x), [a4y] "v"(acc[4].y), [a5x] "v"(acc[5].x), [a5y] "v"(acc[5].y),
                          [a6x] "v"(acc[6].x), [a6y] "v"(acc[6].y), [a7x] "v"(acc[7].x), [a7y] "v"(acc[7].y),
                          [zx] "v"(V == 9 ? 0.f : z.x), [zy] "v"(V == 9 ? 0.f : z.y), [c1] "s"(-2.885390081777927f), [c2] "s"(-1.4426950408889634f));
                    u = copysignf(e1, a);
                } else {
                    float q8[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) q8[s] = acc[s].x + acc[s].y;
                    const float n0 = dpp_fold<0x141>(q8[0], q8[4]), n1 = dpp_fold<0x141>(q8[1], q8[5]);
                    const float n2 = dpp_fold<0x141>(q8[2], q8[6]), n3 = dpp_fold<0x141>(q8[3], q8[7]);
                    const float m0 = dpp_fold<0x4E>(n0, n2), m1 = dpp_fold<0x4E>(n1, n3);
                    const float a = dpp_fold<0xB1>(m0, m0) + z.x, g = dpp_fold<0xB1>(m1, m1) + z.y;
                    u = gate(a, g);
                }
            }
            accum = u;
            if ((tid & 1) == 0) {                                           // the granule store of the send (every second lane, as gwriter)
                const u64 gr = ((u64)(unsigned)(r + 9) << 32) | (u64)__float_as_uint(u);
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(mail + 4096 + (tid >> 1)), "v"(gr) : "memory");
                lds[2048 + (tid >> 1)] = u;
            }
            if constexpr (V == 7) __builtin_amdgcn_s_setprio(0);
        }
    }
    __syncthreads();
    t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) result[0] = t1 - t0;
    sink[tid] = accum;
}

template <int V> static double run(int reps, u64* mail, u64* result, float* sink) {
    CK(hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    double best = 1e30;
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(512), 100 * 1024, 0, reps, mail, result, sink);
        CK(hipDeviceSynchronize());
        u64 ticks = 0;
        CK(hipMemcpy(&ticks, result, 8, hipMemcpyDeviceToHost));
        const double ns = (double)ticks * 10.0 / reps;
        if (pass > 0 && ns < best) best = ns;
    }
    return best;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20000;
    u64 *mail, *result;
    float* sink;
    CK(hipMalloc(&mail, (size_t)(1 << 16) * 8)); CK(hipMemset(mail, 0, (size_t)(1 << 16) * 8));
    CK(hipMalloc(&result, 64)); CK(hipMalloc(&sink, 4096 * 4));
    const char* names[] = {"0 the phase as compiled (baseline)", "1 no LDS read (x in registers)", "2 no reduce / gate", "3 barrier + store only",
                           "4 FMAs only (no read, no tail)", "5 hand-scheduled tail (asm)", "6 4-lane K split, two quad_perm levels", "7 baseline, s_setprio 3",
                           "8 reads as eight ds_read_b64", "9 hand-scheduled tail + zin in the accumulators", "10 row-pair accumulators (no x + y level)",
                           "11 = 10 + last DPP level as one instruction", "12 = 11 + zin in the accumulators", "13 = 12 + gate as (2 r1 - 1) r2, scales folded",
                           "14 = 10 + gate of 13, zin behind the reduce (= WNV_PHASE2)", "15 = 14, zin added mid-sequence"};
    double ns[16];
    ns[0] = run<0>(reps, mail, result, sink); ns[1] = run<1>(reps, mail, result, sink); ns[2] = run<2>(reps, mail, result, sink);
    ns[3] = run<3>(reps, mail, result, sink); ns[4] = run<4>(reps, mail, result, sink); ns[5] = run<5>(reps, mail, result, sink);
    ns[6] = run<6>(reps, mail, result, sink); ns[7] = run<7>(reps, mail, result, sink); ns[8] = run<8>(reps, mail, result, sink);
    ns[9] = run<9>(reps, mail, result, sink);
    ns[10] = run<10>(reps, mail, result, sink); ns[11] = run<11>(reps, mail, result, sink); ns[12] = run<12>(reps, mail, result, sink);
    ns[13] = run<13>(reps, mail, result, sink); ns[14] = run<14>(reps, mail, result, sink); ns[15] = run<15>(reps, mail, result, sink);
    for (int v = 0; v < 16; ++v) printf("variant %-52s %7.1f ns per phase (+ barrier)\n", names[v], ns[v]);
    printf("derived: barrier+store floor %.1f | FMA phase alone %.1f (256 x 128 MACs at 128 FMA/clk = 106.7) | LDS read on the chain %.1f | reduce+gate tail %.1f\n",
           ns[3], ns[4] - ns[3], ns[0] - ns[1], ns[0] - ns[2]);
    return 0;
}
