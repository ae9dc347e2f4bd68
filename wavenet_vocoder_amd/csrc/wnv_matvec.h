// wnv_matvec.h -- streaming K-major matrix-vector product shared by the generic and the ring kernels.
#pragma once
#include <hip/hip_runtime.h>

// y_partial[wave][0..Np) = sum over this wave's K-slice of Wt[k][0..Np) * x[k]
// Wt is K-major ([K][Np], Np % 4 == 0); a lane owns 4 consecutive outputs; waves split K.
template <int NW>
__device__ __noinline__ void matvec_partial(const float* __restrict__ Wt, int K, int Np,
                                               const float* __restrict__ x, float* __restrict__ part,
                                               int pstride, int wave, int lane) {
    const int kper = (K + NW - 1) / NW;
    const int k0 = wave * kper;
    int k1 = k0 + kper;
    if (k1 > K) k1 = K;
    for (int n0 = lane * 4; n0 < Np; n0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 < K) {
            const float* wp = Wt + (size_t)k0 * Np + n0;
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                float4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const float4*>(wp + (size_t)j * Np);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = x[k + j];
                    acc.x = fmaf(w[j].x, xv, acc.x);
                    acc.y = fmaf(w[j].y, xv, acc.y);
                    acc.z = fmaf(w[j].z, xv, acc.z);
                    acc.w = fmaf(w[j].w, xv, acc.w);
                }
                wp += (size_t)8 * Np;
            }
            if (k + 4 <= k1) {
                float4 w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(wp + (size_t)j * Np);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xv = x[k + j];
                    acc.x = fmaf(w[j].x, xv, acc.x);
                    acc.y = fmaf(w[j].y, xv, acc.y);
                    acc.z = fmaf(w[j].z, xv, acc.z);
                    acc.w = fmaf(w[j].w, xv, acc.w);
                }
                wp += (size_t)4 * Np;
                k += 4;
            }
            for (; k < k1; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(wp);
                const float xv = x[k];
                acc.x = fmaf(w.x, xv, acc.x);
                acc.y = fmaf(w.y, xv, acc.y);
                acc.z = fmaf(w.z, xv, acc.z);
                acc.w = fmaf(w.w, xv, acc.w);
                wp += Np;
            }
        }
        *reinterpret_cast<float4*>(part + (size_t)wave * pstride + n0) = acc;
    }
}

// init + part[0][n] + part[1][n] + ... in that order.  The reads are issued eight at a time and pinned before the first add: written
// as a plain loop, the compiler waits for every LDS read (pair) before it issues the next one (found in the group-ring kernel with
// deferred stamps: five LDS round trips for eight values; profiles/r03_wide_timeline.txt).
template <int NW>
__device__ __forceinline__ float reduce_part(const float* part, int pstride, int n, float init) {
    float v = init;
    if constexpr (NW % 8 == 0) {
#pragma unroll
        for (int w0 = 0; w0 < NW; w0 += 8) {
            const float* b = part + (size_t)w0 * pstride + n;
            float x0 = b[0], x1 = b[pstride], x2 = b[2 * pstride], x3 = b[3 * pstride];
            float x4 = b[4 * pstride], x5 = b[5 * pstride], x6 = b[6 * pstride], x7 = b[7 * pstride];
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            v = (((((((v + x0) + x1) + x2) + x3) + x4) + x5) + x6) + x7;
        }
    } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) v += part[w * pstride + n];
    }
    return v;
}


// Register-frugal, always-inlined variant for kernels that keep most of the register file pinned (ring kernel):
// at most four 16-B loads in flight per lane.
template <int NW>
__device__ __forceinline__ void matvec_partial_small(const float* __restrict__ Wt, int K, int Np,
                                                     const float* __restrict__ x, float* __restrict__ part,
                                                     int pstride, int wave, int lane) {
    const int kper = (K + NW - 1) / NW;
    const int k0 = wave * kper;
    int k1 = k0 + kper;
    if (k1 > K) k1 = K;
    for (int n0 = lane * 4; n0 < Np; n0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 < K) {
            const float* wp = Wt + (size_t)k0 * Np + n0;
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                float4 w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(wp + (size_t)j * Np);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xv = x[k + j];
                    acc.x = fmaf(w[j].x, xv, acc.x);
                    acc.y = fmaf(w[j].y, xv, acc.y);
                    acc.z = fmaf(w[j].z, xv, acc.z);
                    acc.w = fmaf(w[j].w, xv, acc.w);
                }
                wp += (size_t)4 * Np;
            }
            for (; k < k1; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(wp);
                const float xv = x[k];
                acc.x = fmaf(w.x, xv, acc.x);
                acc.y = fmaf(w.y, xv, acc.y);
                acc.z = fmaf(w.z, xv, acc.z);
                acc.w = fmaf(w.w, xv, acc.w);
                wp += Np;
            }
        }
        *reinterpret_cast<float4*>(part + (size_t)wave * pstride + n0) = acc;
    }
}
