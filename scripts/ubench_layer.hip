// ubench_layer.hip -- cycles per gated layer on ONE CU with the chain weights resident in VGPRs, no hand-off.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_layer.bin scripts/ubench_layer.hip && scripts/ubench_layer.bin
//
// One workgroup iterates   z = W2 h + pre ; u = tanh(z_a) sigmoid(z_g) ; hn = (Wo u + bo + h) sqrt(.5) ; h <- hn   N times
// (the residual layer of modules.py:127-163 with the history taps folded into `pre`), feeding hn back through LDS in
// place of the CU->CU mailbox.  Variants: the number of lanes KSL that split the K = 128 contraction (4 = the round-1
// ring kernel's quad mapping, 8, 16).  Reports shader cycles (s_memtime) and wall ns (100 MHz) per layer, i.e. also
// the sustained shader clock; the result vector is checked against a host reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

constexpr int RT = 512, RC = 128, GC = 256;

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// butterfly all-reduce over KSL adjacent lanes (KSL = 4, 8, 16)
template <int KSL> __device__ __forceinline__ float ks_allreduce(float v) {
    v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                       // quad_perm [2,3,0,1]
    if (KSL >= 8) v = dpp_add<0x141>(v);        // row_half_mirror
    if (KSL >= 16) v = dpp_add<0x140>(v);       // row_mirror
    return v;
}
__device__ __forceinline__ float fast_gate(float a, float g) {
    const float e = __expf(-2.0f * fabsf(a));
    const float f = __expf(-g);
    const float r = __frcp_rn((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}

template <int KSL> struct Map {
    static constexpr int SL = RC / KSL;          // K-slice length per lane
    static constexpr int NG = RT / KSL;          // output groups per workgroup
    static constexpr int CH = RC / NG;           // channels per group
    static constexpr int PS = SL + 4;            // padded LDS stride of one slice (bank spread)
};

template <int KSL>
__global__ void __launch_bounds__(RT) layer_kernel(const float* __restrict__ W2, const float* __restrict__ Wo,
                                                   const float* __restrict__ pre_g, const float* __restrict__ bo_g,
                                                   const float* __restrict__ h0, float* __restrict__ out, u64* stamps, int n) {
    using M = Map<KSL>;
    constexpr int SL = M::SL, CH = M::CH, PS = M::PS;
    __shared__ __attribute__((aligned(16))) float hs[KSL * PS];
    __shared__ __attribute__((aligned(16))) float us[KSL * PS];
    __shared__ float pre[GC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = lane & (KSL - 1);
    const int og = tid / KSL;
    // resident weights: rows (tanh half) CH*og + c, (sigmoid half) 128 + CH*og + c; columns SL*ks .. +SL
    float wa[CH][SL], wg[CH][SL], wo[CH][SL], bo[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            wa[c][k] = W2[(size_t)(CH * og + c) * RC + SL * ks + k];
            wg[c][k] = W2[(size_t)(RC + CH * og + c) * RC + SL * ks + k];
            wo[c][k] = Wo[(size_t)(CH * og + c) * RC + SL * ks + k];
        }
        bo[c] = bo_g[CH * og + c];
    }
    if (tid < GC) pre[tid] = pre_g[tid];
    if (tid < RC) hs[(tid / SL) * PS + (tid % SL)] = h0[tid];
    __syncthreads();
    const int myc = ks % CH;                         // the channel of the group this lane finishes
    const int mych = CH * og + myc;
    const int myslot = (mych / SL) * PS + (mych % SL);
    u64 c0 = 0, w0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    for (int it = 0; it < n; ++it) {
        float x[SL];
#pragma unroll
        for (int k = 0; k < SL; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&hs[ks * PS + k]);
            x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
        }
        const float hres = hs[myslot];
        float a[CH], g[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int k = 0; k < SL; k += 2) {
                a0 = fmaf(wa[c][k], x[k], a0); a1 = fmaf(wa[c][k + 1], x[k + 1], a1);
                g0 = fmaf(wg[c][k], x[k], g0); g1 = fmaf(wg[c][k + 1], x[k + 1], g1);
            }
            a[c] = a0 + a1; g[c] = g0 + g1;
            if (ks == 0) { a[c] += pre[CH * og + c]; g[c] += pre[RC + CH * og + c]; }
            a[c] = ks_allreduce<KSL>(a[c]);
            g[c] = ks_allreduce<KSL>(g[c]);
        }
        float am = a[0], gm = g[0];
#pragma unroll
        for (int c = 1; c < CH; ++c) { am = myc == c ? a[c] : am; gm = myc == c ? g[c] : gm; }
        const float u = fast_gate(am, gm);
        if (ks < CH) us[myslot] = u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SL; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&us[ks * PS + k]);
            x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
        }
        float o[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int k = 0; k < SL; k += 2) { o0 = fmaf(wo[c][k], x[k], o0); o1 = fmaf(wo[c][k + 1], x[k + 1], o1); }
            o[c] = ks_allreduce<KSL>(o0 + o1);
        }
        float om = o[0], bm = bo[0];
#pragma unroll
        for (int c = 1; c < CH; ++c) { om = myc == c ? o[c] : om; bm = myc == c ? bo[c] : bm; }
        const float hn = (om + bm + hres) * 0.70710678118654752440f;
        __syncthreads();                               // every lane has read hs (stands in for the mailbox hop)
        if (ks < CH) hs[myslot] = hn;
        __syncthreads();
    }
    if (tid == 0) { stamps[0] = __builtin_readcyclecounter() - c0; stamps[1] = wall_clock64() - w0; }
    if (tid < RC) out[tid] = hs[(tid / SL) * PS + (tid % SL)];
}

static void host_ref(const std::vector<float>& W2, const std::vector<float>& Wo, const std::vector<float>& pre,
                     const std::vector<float>& bo, std::vector<float> h, int n, std::vector<float>& out) {
    std::vector<double> z(GC), u(RC), hn(RC);
    for (int it = 0; it < n; ++it) {
        for (int r = 0; r < GC; ++r) { double s = pre[r]; for (int k = 0; k < RC; ++k) s += (double)W2[(size_t)r * RC + k] * h[k]; z[r] = s; }
        for (int i = 0; i < RC; ++i) u[i] = std::tanh(z[i]) / (1.0 + std::exp(-z[RC + i]));
        for (int r = 0; r < RC; ++r) { double s = bo[r]; for (int k = 0; k < RC; ++k) s += (double)Wo[(size_t)r * RC + k] * u[k]; hn[r] = (s + h[r]) * 0.70710678118654752440; }
        for (int i = 0; i < RC; ++i) h[i] = (float)hn[i];
    }
    out = h;
}

template <int KSL> static void run(const float* dW2, const float* dWo, const float* dpre, const float* dbo, const float* dh0,
                                   float* dout, u64* dst, int n, int grid, const std::vector<float>& ref) {
    hipLaunchKernelGGL(layer_kernel<KSL>, dim3(grid), dim3(RT), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, 48);   // warm + check
    CK(hipDeviceSynchronize());
    u64 st[2]; std::vector<float> out(RC);
    CK(hipMemcpy(out.data(), dout, RC * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < RC; ++i) err = std::fmax(err, std::fabs(out[i] - ref[i]));
    hipLaunchKernelGGL(layer_kernel<KSL>, dim3(grid), dim3(RT), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    printf("KSL=%2d grid=%3d : %8.1f cycles/layer  %7.1f ns/layer  (clock %.2f GHz)  max err vs f64 host %.2e\n", KSL, grid,
           (double)st[0] / n, (double)st[1] * 10.0 / n, (double)st[0] / ((double)st[1] * 10.0), err);
}

int main() {
    const int n = 20000;
    std::vector<float> W2((size_t)GC * RC), Wo((size_t)RC * RC), pre(GC), bo(RC), h0(RC), ref;
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : W2) v = rnd() * 0.15f;
    for (auto& v : Wo) v = rnd() * 0.15f;
    for (auto& v : pre) v = rnd();
    for (auto& v : bo) v = rnd() * 0.1f;
    for (auto& v : h0) v = rnd();
    host_ref(W2, Wo, pre, bo, h0, 48, ref);
    float *dW2, *dWo, *dpre, *dbo, *dh0, *dout; u64* dst;
    CK(hipMalloc(&dW2, W2.size() * 4)); CK(hipMalloc(&dWo, Wo.size() * 4)); CK(hipMalloc(&dpre, GC * 4));
    CK(hipMalloc(&dbo, RC * 4)); CK(hipMalloc(&dh0, RC * 4)); CK(hipMalloc(&dout, RC * 4)); CK(hipMalloc(&dst, 64));
    CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWo, Wo.data(), Wo.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpre, pre.data(), GC * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbo, bo.data(), RC * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh0, h0.data(), RC * 4, hipMemcpyHostToDevice));
    for (int grid : {1, 200}) {
        run<4>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<8>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<16>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
    }
    return 0;
}
