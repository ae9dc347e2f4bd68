// ubench_tapmv.hip -- the tap role's round (run_tap [D] in csrc/wnv_ring.hip: 4 utterances x 256 outputs x 336 K rows per workgroup,
// weights 32 rows per lane in VGPRs + 12 in LDS, inputs read from LDS as 16-byte broadcasts) in isolation: what does a round cost with
// 8 waves (two per SIMD) and with 4 (one per SIMD), and where does the time go -- the LDS reads or the packed FMAs?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_tapmv.bin scripts/ubench_tapmv.hip && scripts/ubench_tapmv.bin
// VAR 0: as in the kernel (two utterances at a time, reads a chunk ahead as the compiler schedules them)
// VAR 1: FMAs only (the inputs are read once, outside the timed loop)
// VAR 2: LDS reads only (the same 56 ds_read_b128 per round, consumed by one add each)
// VAR 3: all four utterances' inputs of a 4-row chunk read in one batch (4 reads in flight per wait instead of 2)
// VAR 4: explicit software pipeline -- the NEXT chunk's reads are issued before the current chunk's FMAs (8 more registers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)
constexpr int KR = 32, KL = 12, KPER = 44, KX = 8 * KPER, TB = 8;

template <int CTRL> __device__ __forceinline__ float dpp_fold(float keep, float send) {
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xF, 0xF, true));
}

template <int VAR>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(244))) k(const float* W, float* out, u64* cyc, int rounds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                  // [TB][KX]
    float4* wl = reinterpret_cast<float4*>(smem + TB * KX);   // [waves][KL][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = lane & 7;
    for (int i = tid; i < TB * KX; i += blockDim.x) xin[i] = 0.001f * (float)(i % 97);
    float4 wreg[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) wreg[r] = *reinterpret_cast<const float4*>(W + ((size_t)(r * 512 + tid) * 4));
    for (int r = 0; r < KL; ++r) wl[(wave * KL + r) * 64 + lane] = *reinterpret_cast<const float4*>(W + ((size_t)((KR + r) * 512 + tid) * 4));
    __syncthreads();
    const int k0 = ks * KPER;
    const int uq = ks >> 1, uqm = 3 - uq;
    const int ug[4] = {uq, uq ^ 1, uqm, uqm ^ 1};
    float sink = 0.f;
    const u64 t0 = wall_clock64();
    for (int it = 0; it < rounds; ++it) {
        const float* xr = xin + (size_t)((it & 1) * 4) * KX;
        const float* xg[4] = {xr + ug[0] * KX + k0, xr + ug[1] * KX + k0, xr + ug[2] * KX + k0, xr + ug[3] * KX + k0};
        f2 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = f2{0.f, 0.f};
        if constexpr (VAR == 0 || VAR == 1 || VAR == 2) {
#pragma unroll
            for (int r4 = 0; r4 < KR / 4; ++r4) {
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    float4 xa, xc;
                    if constexpr (VAR == 1) { xa = make_float4(sink, 1.f, 2.f, 3.f); xc = make_float4(3.f, sink, 1.f, 2.f); }
                    else { xa = *reinterpret_cast<const float4*>(xg[gp] + 4 * r4); xc = *reinterpret_cast<const float4*>(xg[gp + 1] + 4 * r4); }
                    if constexpr (VAR == 2) { acc[gp][0].x += xa.x + xa.w; acc[gp + 1][0].x += xc.y + xc.z; continue; }
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
            for (int r = 0; r < KL; r += 4) {
                float4 w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = VAR == 1 ? make_float4(1.f, sink, 2.f, 3.f) : wl[(wave * KL + r + e) * 64 + lane];
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    float4 xa, xc;
                    if constexpr (VAR == 1) { xa = make_float4(sink, 1.f, 2.f, 3.f); xc = make_float4(3.f, sink, 1.f, 2.f); }
                    else { xa = *reinterpret_cast<const float4*>(xg[gp] + KR + r); xc = *reinterpret_cast<const float4*>(xg[gp + 1] + KR + r); }
                    if constexpr (VAR == 2) { acc[gp][0].x += xa.x + xa.w + w[gp].x; acc[gp + 1][0].x += xc.y + xc.z + w[gp + 1].y; continue; }
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
        } else if constexpr (VAR == 3) {
            auto chunk = [&](const float4 (&x)[4], const float4 (&w)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float xv = e == 0 ? x[g].x : e == 1 ? x[g].y : e == 2 ? x[g].z : x[g].w;
                        const f2 xx = f2{xv, xv};
                        acc[g][0] = __builtin_elementwise_fma(f2{w[e].x, w[e].y}, xx, acc[g][0]);
                        acc[g][1] = __builtin_elementwise_fma(f2{w[e].z, w[e].w}, xx, acc[g][1]);
                    }
            };
#pragma unroll
            for (int r4 = 0; r4 < KR / 4; ++r4) {
                float4 x[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) x[g] = *reinterpret_cast<const float4*>(xg[g] + 4 * r4);
                const float4 w[4] = {wreg[4 * r4], wreg[4 * r4 + 1], wreg[4 * r4 + 2], wreg[4 * r4 + 3]};
                chunk(x, w);
            }
            for (int r = 0; r < KL; r += 4) {
                float4 w[4], x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = wl[(wave * KL + r + e) * 64 + lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) x[g] = *reinterpret_cast<const float4*>(xg[g] + KR + r);
                chunk(x, w);
            }
        } else if constexpr (VAR >= 5) {                  // VAR 5 / 6 / 7: the reads run AHEAD chunks in front of the FMAs (a ring of register pairs)
            constexpr int AHEAD = VAR - 3;                    // 2, 3, 4
            constexpr int NC = 2 * (KR + KL) / 4;
            float4 xq[AHEAD + 1][2];
            auto rd = [&](int c, float4& xa, float4& xc) {
                const int r4 = c >> 1, gp = 2 * (c & 1);
                xa = *reinterpret_cast<const float4*>(xg[gp] + 4 * r4); xc = *reinterpret_cast<const float4*>(xg[gp + 1] + 4 * r4);
            };
#pragma unroll
            for (int c = 0; c < AHEAD; ++c) rd(c, xq[c][0], xq[c][1]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c + AHEAD < NC) rd(c + AHEAD, xq[(c + AHEAD) % (AHEAD + 1)][0], xq[(c + AHEAD) % (AHEAD + 1)][1]);
                const int r4 = c >> 1, gp = 2 * (c & 1);
                float4 w[4];
                if (r4 < KR / 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = wreg[4 * r4 + e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = wl[(wave * KL + 4 * (r4 - KR / 4) + e) * 64 + lane];
                }
                const float4 xa = xq[c % (AHEAD + 1)][0], xc = xq[c % (AHEAD + 1)][1];
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
        } else {                                          // VAR 4: explicit pipeline over (r4, pair) chunks
            auto rd = [&](int c, float4& xa, float4& xc) {   // chunk c = 2 r4 + pair (KR part), then the LDS rows
                const int r4 = c >> 1, gp = 2 * (c & 1);
                xa = *reinterpret_cast<const float4*>(xg[gp] + 4 * r4); xc = *reinterpret_cast<const float4*>(xg[gp + 1] + 4 * r4);
            };
            float4 xa, xc, na, nc;
            rd(0, xa, xc);
            constexpr int NC = 2 * (KR + KL) / 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c + 1 < NC) rd(c + 1, na, nc);
                const int r4 = c >> 1, gp = 2 * (c & 1);
                float4 w[4];
                if (r4 < KR / 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = wreg[4 * r4 + e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = wl[(wave * KL + 4 * (r4 - KR / 4) + e) * 64 + lane];
                }
                const float xs[2][4] = {{xa.x, xa.y, xa.z, xa.w}, {xc.x, xc.y, xc.z, xc.w}};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const f2 xx = f2{xs[g][e], xs[g][e]};
                        acc[gp + g][0] = __builtin_elementwise_fma(f2{w[e].x, w[e].y}, xx, acc[gp + g][0]);
                        acc[gp + g][1] = __builtin_elementwise_fma(f2{w[e].z, w[e].w}, xx, acc[gp + g][1]);
                    }
                xa = na; xc = nc;
            }
        }
        // the K slices meet as in the kernel
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc[g][h] = f2{dpp_fold<0x141>(acc[g][h].x, acc[g + 2][1 - h].x), dpp_fold<0x141>(acc[g][h].y, acc[g + 2][1 - h].y)};
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[0][h] = f2{dpp_fold<0x4E>(acc[0][h].x, acc[1][h].x), dpp_fold<0x4E>(acc[0][h].y, acc[1][h].y)};
        sink += dpp_fold<0xB1>(acc[0][0].x, acc[0][1].x) + dpp_fold<0xB1>(acc[0][0].y, acc[0][1].y);
        asm volatile("" : "+v"(sink));
    }
    const u64 t1 = wall_clock64();
    out[blockIdx.x * 512 + tid] = sink;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR> void run(const float* W, float* out, u64* cyc, const char* what) {
    const size_t lds = (TB * KX + 8 * KL * 64 * 4) * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("VAR %d (%s):", VAR, what);
    const int rounds = 2000;
    for (int nw : {4, 8}) {
        hipLaunchKernelGGL((k<VAR>), dim3(1), dim3(64 * nw), lds, 0, W, out, cyc, rounds);
        CK(hipDeviceSynchronize());
        u64 c[8]; CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
        u64 mx = 0, mn = ~0ull; for (int w = 0; w < nw; ++w) { mx = c[w] > mx ? c[w] : mx; mn = c[w] < mn ? c[w] : mn; }
        printf("   %d waves: %.0f ns per round (fastest wave %.0f)", nw, 10.0 * mx / rounds, 10.0 * mn / rounds);
    }
    printf("\n");
}

// ---- VAR 8: the round on the matrix pipe.  v_mfma_f32_4x4x1_16B_f32: 16 blocks of (4 x 1) . (1 x 4); lane = 4 block + i holds A_block[i]
// and B_block[j = i]; D (4 VGPRs): register v of lane 4 block + j = D_block[v][j].  Block b: output rows 4 (b & 7) .. + 3 of the wave's 32,
// K half b >> 3; j = the round's utterance.  168 MFMAs (8 clocks each) + 42 ds_read_b128 per wave and round; the halves meet by
// v_permlane32_swap.
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int KH = 168, NQ = KH / 4, KXM = KX + 16;
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(244))) kmf(const float* Wkm, const float* x, float* out, u64* cyc, int rounds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                  // [TB][KXM]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = lane >> 2, ij = lane & 3, h = b >> 3;
    for (int i = tid; i < TB * KXM; i += blockDim.x) xin[i] = (i % KXM) < 336 ? x[(i / KXM) * 336 + (i % KXM)] : 0.f;
    float A[KH];
    const int row = 32 * wave + 4 * (b & 7) + ij;
#pragma unroll
    for (int s_ = 0; s_ < KH; ++s_) A[s_] = Wkm[(size_t)(h * KH + s_) * 256 + row];       // K-major [336][256]
    __syncthreads();
    float sink = 0.f;
    f4 dlast = {0.f, 0.f, 0.f, 0.f};
    const u64 t0 = wall_clock64();
    for (int it = 0; it < rounds; ++it) {
        const float* xr = xin + (size_t)((it & 1) * 4 + ij) * KXM + h * KH;
        f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f}, d3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * q);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[4 * q], xv.x, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[4 * q + 1], xv.y, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[4 * q + 2], xv.z, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[4 * q + 3], xv.w, d3, 0, 0, 0);
        }
        f4 d = (d0 + d1) + (d2 + d3);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[v]), __float_as_uint(d[v]), false, false);
            d[v] += __uint_as_float(h ? r[0] : r[1]);
        }
        if (it == 0) dlast = d;
        sink += h ? d.z + d.w : d.x + d.y;
        asm volatile("" : "+v"(sink));
    }
    const u64 t1 = wall_clock64();
    // the first round's results: out[utterance j][row] (every lane of the lower half writes its four rows)
    if (h == 0) for (int v = 0; v < 4; ++v) out[ij * 256 + 32 * wave + 4 * b + v] = dlast[v];
    out[4 * 256 + tid] = sink;
    if (lane == 0) cyc[wave] = t1 - t0;
}
void run_mf() {
    std::vector<float> W((size_t)336 * 256), x((size_t)8 * 336);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 0.01f * (float)((i * 2654435761u) % 101) - 0.5f;
    for (size_t i = 0; i < x.size(); ++i) x[i] = 0.02f * (float)((i * 40503u) % 89) - 0.8f;
    float *dW, *dx, *out; u64* cyc;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&out, 8192 * 4)); CK(hipMalloc(&cyc, 64));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)TB * KXM * sizeof(float);
    printf("VAR 8 (v_mfma_f32_4x4x1_16B_f32, 168 per wave and round + 42 ds_read_b128):");
    const int rounds = 2000;
    for (int nw : {4, 8}) {
        CK(hipMemset(out, 0, 8192 * 4));
        hipLaunchKernelGGL(kmf, dim3(1), dim3(64 * nw), lds, 0, dW, dx, out, cyc, rounds);
        CK(hipDeviceSynchronize());
        u64 c[8]; CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
        u64 mx = 0, mn = ~0ull; for (int w = 0; w < nw; ++w) { mx = c[w] > mx ? c[w] : mx; mn = c[w] < mn ? c[w] : mn; }
        printf("   %d waves: %.0f ns per round (fastest wave %.0f)", nw, 10.0 * mx / rounds, 10.0 * mn / rounds);
        if (nw == 8) {
            std::vector<float> o(4 * 256); CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int j = 0; j < 4; ++j) for (int r = 0; r < 256; ++r) {
                double ref = 0; for (int k = 0; k < 336; ++k) ref += (double)W[(size_t)k * 256 + r] * x[(size_t)j * 336 + k];
                worst = std::max(worst, std::abs(ref - o[j * 256 + r]));
            }
            printf("   max |err| vs double %.3g", worst);
        }
    }
    printf("\n");
}

// ---- VAR 11: a PASS OF SIXTEEN utterances on the matrix pipe.  v_mfma_f32_16x16x4_f32: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16],
// D register v of a lane = D[i = 4 (lane / 16) + v][j = lane % 16].  A wave's 32 output rows = two tiles; K = 336 = 21 groups of 16: lane (n, kq)
// reads x[n][16 J + 4 kq .. + 3] with ONE ds_read_b128 and feeds four MFMAs per tile with it (MFMA (J, i) sums k = 16 J + 4 kq' + i over kq').
// Weights: 16 groups in registers (128), 5 groups as 10 float4 per lane in LDS -- the legacy form's split.  Per wave and pass of 16 utterances:
// 168 MFMAs (32 clocks each) + 31 ds_read_b128, against 4 x (352 v_pk_fma_f32 + 56 ds_read_b128) for the same sixteen utterances today.
constexpr int NJ = 21, NJR = 16, KX16 = 388;
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(244))) kmf16(const float* Wkm, const float* x, float* out, u64* cyc, int passes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                           // [16][KX16]
    float4* wl = reinterpret_cast<float4*>(smem + 16 * KX16);   // [8 waves][(NJ - NJR) * 2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 16 * KX16; i += blockDim.x) xin[i] = (i % KX16) < 336 ? x[((i / KX16) & 7) * 336 + (i % KX16)] * (1.f + (float)((i / KX16) >> 3)) : 0.f;
    float A[NJR][2][4];
    auto wk = [&](int J, int tile, int i) { return Wkm[(size_t)(16 * J + 4 * kq + i) * 256 + 32 * wave + 16 * tile + n]; };
#pragma unroll
    for (int J = 0; J < NJR; ++J)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int i = 0; i < 4; ++i) A[J][tl][i] = wk(J, tl, i);
    for (int J = NJR; J < NJ; ++J)
        for (int tl = 0; tl < 2; ++tl) wl[((size_t)wave * (NJ - NJR) * 2 + (J - NJR) * 2 + tl) * 64 + lane] = make_float4(wk(J, tl, 0), wk(J, tl, 1), wk(J, tl, 2), wk(J, tl, 3));
    __syncthreads();
    float sink = 0.f;
    f4 first[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const u64 t0 = wall_clock64();
    for (int it = 0; it < passes; ++it) {
        const float* xr = xin + (size_t)n * KX16 + 4 * kq;
        f4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float4 xv = *reinterpret_cast<const float4*>(xr), w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
            float4 xn = xv, n0 = w0, n1 = w1;
            if (J + 1 < NJ) {
                xn = *reinterpret_cast<const float4*>(xr + 16 * (J + 1));
                if (J + 1 >= NJR) { n0 = wl[((size_t)wave * (NJ - NJR) * 2 + (J + 1 - NJR) * 2) * 64 + lane]; n1 = wl[((size_t)wave * (NJ - NJR) * 2 + (J + 1 - NJR) * 2 + 1) * 64 + lane]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float a0[4] = {J < NJR ? A[J < NJR ? J : 0][0][0] : w0.x, J < NJR ? A[J < NJR ? J : 0][0][1] : w0.y, J < NJR ? A[J < NJR ? J : 0][0][2] : w0.z, J < NJR ? A[J < NJR ? J : 0][0][3] : w0.w};
            const float a1[4] = {J < NJR ? A[J < NJR ? J : 0][1][0] : w1.x, J < NJR ? A[J < NJR ? J : 0][1][1] : w1.y, J < NJR ? A[J < NJR ? J : 0][1][2] : w1.z, J < NJR ? A[J < NJR ? J : 0][1][3] : w1.w};
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], xs[i], d[0], 0, 0, 0);
                d[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], xs[i], d[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            xv = xn; w0 = n0; w1 = n1;
        }
        if (it == 0) { first[0] = d[0]; first[1] = d[1]; }
        sink += d[0].x + d[0].w + d[1].y + d[1].z;
        asm volatile("" : "+v"(sink));
    }
    const u64 t1 = wall_clock64();
    for (int tl = 0; tl < 2; ++tl) for (int v = 0; v < 4; ++v) out[n * 256 + 32 * wave + 16 * tl + 4 * kq + v] = first[tl][v];
    out[16 * 256 + tid] = sink;
    if (lane == 0) cyc[wave] = t1 - t0;
}
void run_mf16() {
    std::vector<float> W((size_t)336 * 256), x((size_t)8 * 336);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 0.01f * (float)((i * 2654435761u) % 101) - 0.5f;
    for (size_t i = 0; i < x.size(); ++i) x[i] = 0.02f * (float)((i * 40503u) % 89) - 0.8f;
    float *dW, *dx, *out; u64* cyc;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&out, 8192 * 4)); CK(hipMalloc(&cyc, 64));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)16 * KX16 + (size_t)8 * (NJ - NJR) * 2 * 64 * 4) * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kmf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("VAR 11 (v_mfma_f32_16x16x4_f32, a pass of SIXTEEN utterances: 168 MFMAs + 31 ds_read_b128 per wave):");
    const int passes = 1000;
    for (int nw : {4, 8}) {
        CK(hipMemset(out, 0, 8192 * 4));
        hipLaunchKernelGGL(kmf16, dim3(1), dim3(64 * nw), lds, 0, dW, dx, out, cyc, passes);
        CK(hipDeviceSynchronize());
        u64 c[8]; CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
        u64 mx = 0, mn = ~0ull; for (int w = 0; w < nw; ++w) { mx = c[w] > mx ? c[w] : mx; mn = c[w] < mn ? c[w] : mn; }
        printf("   %d waves: %.0f ns per pass of 16 = %.0f per 4 utterances (fastest wave %.0f)", nw, 10.0 * mx / passes, 2.5 * mx / passes, 10.0 * mn / passes);
        if (nw == 8) {
            std::vector<float> o(16 * 256); CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int j = 0; j < 16; ++j) for (int r = 0; r < 256; ++r) {
                double ref = 0; for (int k = 0; k < 336; ++k) ref += (double)W[(size_t)k * 256 + r] * x[(size_t)(j & 7) * 336 + k] * (1.0 + (j >> 3));
                worst = std::max(worst, std::abs(ref - o[j * 256 + r]));
            }
            printf("   max |err| vs double %.3g", worst);
        }
    }
    printf("\n");
}

int main() {
    std::vector<float> h((size_t)(KR + KL) * 512 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 101) - 0.5f;
    float *W, *out; u64* cyc;
    CK(hipMalloc(&W, h.size() * 4)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 64 * 8));
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("tap round: 352 v_pk_fma_f32 + 56 ds_read_b128 per wave; FMA floor 352 x 4 clk = 587 ns per wave at 2.4 GHz (two waves per SIMD: 1173 ns)\n");
    run<0>(W, out, cyc, "as in the kernel");
    run<1>(W, out, cyc, "FMAs only");
    run<2>(W, out, cyc, "LDS reads only");
    run<3>(W, out, cyc, "four utterances per batch of reads");
    run<4>(W, out, cyc, "next chunk's reads ahead of the FMAs");
    run<5>(W, out, cyc, "reads 2 chunks ahead");
    run<6>(W, out, cyc, "reads 3 chunks ahead");
    run<7>(W, out, cyc, "reads 4 chunks ahead");
    run_mf();
    run_mf16();
    return 0;
}
