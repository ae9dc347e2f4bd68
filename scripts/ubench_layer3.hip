// ubench_layer3.hip -- the chain part of one gated layer on ONE CU, weights in VGPRs, generalised over
//   NW  = waves per workgroup (8 or 4)     KSL = lanes that split the K = 128 contraction (4 or 8)
// with v_pk_fma_f32 paired over OUTPUTS (tanh row, sigmoid row) x one broadcast input (op_sel), so no horizontal add.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_layer3.bin scripts/ubench_layer3.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

constexpr int RC = 128, GC = 256;

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int KSL> __device__ __forceinline__ float ks_allreduce(float v) {
    v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                       // quad_perm [2,3,0,1]
    if (KSL >= 8) v = dpp_add<0x141>(v);        // row_half_mirror
    return v;
}
__device__ __forceinline__ float gate(float a, float g) {
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -2.8853900817779268f);     // exp(-2|a|)
    const float f = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);           // exp(-g)
    const float r = __builtin_amdgcn_rcpf((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}

template <int NW, int KSL>
__global__ void __launch_bounds__(64 * NW) layer_kernel(const float* __restrict__ W2, const float* __restrict__ Wo,
                                                        const float* __restrict__ pre_g, const float* __restrict__ bo_g,
                                                        const float* __restrict__ h0, float* __restrict__ out, u64* stamps, int n) {
    constexpr int SL = RC / KSL;                     // K-slice per lane
    constexpr int NG = 64 * NW / KSL;                // output groups
    constexpr int CH = RC / NG;                      // channels per group
    constexpr int PS = SL + 4;                       // padded slice stride in LDS
    __shared__ __attribute__((aligned(16))) float hs[KSL * PS];
    __shared__ __attribute__((aligned(16))) float us[KSL * PS];
    const int tid = threadIdx.x;
    const int ks = tid & (KSL - 1), og = tid / KSL;
    auto slot = [](int ch) { return (ch / SL) * PS + (ch % SL); };
    f2 wz[CH][SL];                                   // (tanh row, sigmoid row) of channel c, column k of the slice
    f2 wo[(CH + 1) / 2][SL];                         // CH >= 2: (row 2p, row 2p+1) x column k; CH == 1: (col k, col k+1) pairs
    float bo[CH];
    f2 prez[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = CH * og + c;
#pragma unroll
        for (int k = 0; k < SL; ++k) wz[c][k] = f2{W2[(size_t)ch * RC + SL * ks + k], W2[(size_t)(RC + ch) * RC + SL * ks + k]};
        bo[c] = bo_g[ch];
        prez[c] = ks == 0 ? f2{pre_g[ch], pre_g[RC + ch]} : f2{0.f, 0.f};
    }
    if (CH >= 2) {
#pragma unroll
        for (int p = 0; p < CH / 2; ++p)
#pragma unroll
            for (int k = 0; k < SL; ++k)
                wo[p][k] = f2{Wo[(size_t)(CH * og + 2 * p) * RC + SL * ks + k], Wo[(size_t)(CH * og + 2 * p + 1) * RC + SL * ks + k]};
    } else {
#pragma unroll
        for (int k = 0; k < SL / 2; ++k) wo[0][k] = *reinterpret_cast<const f2*>(&Wo[(size_t)og * RC + SL * ks + 2 * k]);
    }
    if (tid < RC) hs[slot(tid)] = h0[tid];
    __syncthreads();
    const int myc = ks % CH;
    const int myslot = slot(CH * og + myc);
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
        f2 z[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) z[c] = prez[c];
#pragma unroll
        for (int k = 0; k < SL; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) z[c] = __builtin_elementwise_fma(wz[c][k], f2{x[k], x[k]}, z[c]);
        float am = 0.f, gm = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float a = ks_allreduce<KSL>(z[c].x), g = ks_allreduce<KSL>(z[c].y);
            if (c == 0 || myc == c) { am = a; gm = g; }
        }
        const float u = gate(am, gm);
        if (ks < CH) us[myslot] = u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SL; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&us[ks * PS + k]);
            x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
        }
        float om = 0.f, bm = bo[0];
        if (CH >= 2) {
            f2 o[(CH + 1) / 2];
#pragma unroll
            for (int p = 0; p < CH / 2; ++p) o[p] = f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < SL; ++k)
#pragma unroll
                for (int p = 0; p < CH / 2; ++p) o[p] = __builtin_elementwise_fma(wo[p][k], f2{x[k], x[k]}, o[p]);
#pragma unroll
            for (int p = 0; p < CH / 2; ++p) {
                const float e0 = ks_allreduce<KSL>(o[p].x), e1 = ks_allreduce<KSL>(o[p].y);
                if (p == 0 || myc == 2 * p) { om = e0; bm = bo[2 * p]; }
                if (myc == 2 * p + 1) { om = e1; bm = bo[2 * p + 1]; }
            }
        } else {
            f2 o0 = f2{0.f, 0.f}, o1 = f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < SL / 2; k += 2) {
                o0 = __builtin_elementwise_fma(wo[0][k], f2{x[2 * k], x[2 * k + 1]}, o0);
                o1 = __builtin_elementwise_fma(wo[0][k + 1], f2{x[2 * k + 2], x[2 * k + 3]}, o1);
            }
            o0 += o1;
            om = ks_allreduce<KSL>(o0.x + o0.y);
        }
        const float hn = (om + bm + hres) * 0.70710678118654752440f;
        __syncthreads();                               // every lane has read hs (stands in for the mailbox hop)
        if (ks < CH) hs[myslot] = hn;
        __syncthreads();
    }
    if (tid == 0) { stamps[0] = __builtin_readcyclecounter() - c0; stamps[1] = wall_clock64() - w0; }
    if (tid < RC) out[tid] = hs[slot(tid)];
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

template <int NW, int KSL> static void run(const float* dW2, const float* dWo, const float* dpre, const float* dbo, const float* dh0,
                                           float* dout, u64* dst, int n, int grid, const std::vector<float>& ref) {
    hipLaunchKernelGGL((layer_kernel<NW, KSL>), dim3(grid), dim3(64 * NW), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, 48);   // warm + check
    CK(hipDeviceSynchronize());
    u64 st[2]; std::vector<float> out(RC);
    CK(hipMemcpy(out.data(), dout, RC * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < RC; ++i) err = std::fmax(err, std::fabs(out[i] - ref[i]));
    hipLaunchKernelGGL((layer_kernel<NW, KSL>), dim3(grid), dim3(64 * NW), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    printf("NW=%d KSL=%d grid=%3d : %8.1f cycles/layer  %7.1f ns/layer  (clock %.2f GHz)  max err vs f64 host %.2e\n", NW, KSL, grid,
           (double)st[0] / n, (double)st[1] * 10.0 / n, (double)st[0] / ((double)st[1] * 10.0), err);
}

int main() {
    const int n = 20000;
    std::vector<float> W2((size_t)GC * RC), Wo((size_t)RC * RC), pre(GC), bo(RC), h0(RC), ref;
    srand(1);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
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
    for (int grid : {1}) {
        run<8, 4>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<4, 4>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<8, 8>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<4, 8>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
    }
    return 0;
}
