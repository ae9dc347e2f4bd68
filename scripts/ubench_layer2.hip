// ubench_layer2.hip -- hand-packed variant of ubench_layer.hip: v_pk_fma_f32 everywhere, pre-activation folded into
// the accumulator init, raw v_rcp/v_exp gate, quad K-split; NW = 8 waves (1 channel per quad) or 4 waves (2 per quad).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_layer2.bin scripts/ubench_layer2.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

constexpr int RC = 128, GC = 256, QS = 36;

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_allreduce(float v) { return dpp_add<0x4E>(dpp_add<0xB1>(v)); }
__device__ __forceinline__ float gate(float a, float g) {
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -2.8853900817779268f);     // exp(-2|a|)
    const float f = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);           // exp(-g)
    const float r = __builtin_amdgcn_rcpf((1.0f + e) * (1.0f + f));
    return copysignf((1.0f - e) * r, a);
}
__device__ __forceinline__ void lds_read32(const float* p, f2 (&x)[16]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = reinterpret_cast<const float4*>(p)[c];
        x[2 * c] = f2{v.x, v.y}; x[2 * c + 1] = f2{v.z, v.w};
    }
}
__device__ __forceinline__ int qidx(int i) { return QS * (i >> 5) + (i & 31); }

template <int NW, bool STAMP>
__global__ void __launch_bounds__(64 * NW) layer_kernel(const float* __restrict__ W2, const float* __restrict__ Wo,
                                                        const float* __restrict__ pre_g, const float* __restrict__ bo_g,
                                                        const float* __restrict__ h0, float* __restrict__ out, u64* stamps, int n) {
    constexpr int CH = 8 / NW;                       // channels per quad
    __shared__ __attribute__((aligned(16))) float hs[4 * QS];
    __shared__ __attribute__((aligned(16))) float us[4 * QS];
    const int tid = threadIdx.x;
    const int q = tid & 3, og = tid >> 2;
    f2 wa[CH][16], wg[CH][16], wo[CH][16];
    float bo[CH], prea[CH], preg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = CH * og + c;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            wa[c][k] = *reinterpret_cast<const f2*>(&W2[(size_t)ch * RC + 32 * q + 2 * k]);
            wg[c][k] = *reinterpret_cast<const f2*>(&W2[(size_t)(RC + ch) * RC + 32 * q + 2 * k]);
            wo[c][k] = *reinterpret_cast<const f2*>(&Wo[(size_t)ch * RC + 32 * q + 2 * k]);
        }
        bo[c] = bo_g[ch];
        prea[c] = q == 0 ? pre_g[ch] : 0.f;
        preg[c] = q == 0 ? pre_g[RC + ch] : 0.f;
    }
    if (tid < RC) hs[qidx(tid)] = h0[tid];
    __syncthreads();
    const int myc = q % CH;
    const int mych = CH * og + myc;
    const int myslot = qidx(mych);
    u64 c0 = 0, w0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    u64 ph[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
        u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        if (STAMP && tid < 64) s0 = __builtin_readcyclecounter();
        f2 x[16];
        lds_read32(hs + QS * q, x);
        const float hres = hs[myslot];
        float a[CH], g[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            f2 aa = f2{prea[c], 0.f}, gg = f2{preg[c], 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                aa = __builtin_elementwise_fma(wa[c][k], x[k], aa);
                gg = __builtin_elementwise_fma(wg[c][k], x[k], gg);
            }
            a[c] = quad_allreduce(aa.x + aa.y);
            g[c] = quad_allreduce(gg.x + gg.y);
        }
        float am = a[0], gm = g[0];
#pragma unroll
        for (int c = 1; c < CH; ++c) { am = myc == c ? a[c] : am; gm = myc == c ? g[c] : gm; }
        if (STAMP && tid < 64) { asm volatile("" :: "v"(am), "v"(gm)); s1 = __builtin_readcyclecounter(); }
        const float u = gate(am, gm);
        if (q < CH) us[myslot] = u;
        if (STAMP && tid < 64) { asm volatile("" :: "v"(u)); s2 = __builtin_readcyclecounter(); }
        __syncthreads();
        if (STAMP && tid < 64) s3 = __builtin_readcyclecounter();
        lds_read32(us + QS * q, x);
        float o[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            f2 oo = f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) oo = __builtin_elementwise_fma(wo[c][k], x[k], oo);
            o[c] = quad_allreduce(oo.x + oo.y);
        }
        float om = o[0], bm = bo[0];
#pragma unroll
        for (int c = 1; c < CH; ++c) { om = myc == c ? o[c] : om; bm = myc == c ? bo[c] : bm; }
        const float hn = (om + bm + hres) * 0.70710678118654752440f;
        if (STAMP && tid < 64) { asm volatile("" :: "v"(hn)); s4 = __builtin_readcyclecounter(); }
        __syncthreads();                               // every lane has read hs (stands in for the mailbox hop)
        if (q < CH) hs[myslot] = hn;
        __syncthreads();
        if (STAMP && tid < 64) { s5 = __builtin_readcyclecounter(); ph[0] += s1 - s0; ph[1] += s2 - s1; ph[2] += s3 - s2; ph[3] += s4 - s3; ph[4] += s5 - s4; }
    }
    if (STAMP && tid == 0) for (int k = 0; k < 5; ++k) stamps[2 + k] = ph[k];
    if (tid == 0) { stamps[0] = __builtin_readcyclecounter() - c0; stamps[1] = wall_clock64() - w0; }
    if (tid < RC) out[tid] = hs[qidx(tid)];
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

template <int NW, bool STAMP> static void run(const float* dW2, const float* dWo, const float* dpre, const float* dbo, const float* dh0,
                                  float* dout, u64* dst, int n, int grid, const std::vector<float>& ref) {
    hipLaunchKernelGGL((layer_kernel<NW, STAMP>), dim3(grid), dim3(64 * NW), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, 48);   // warm + check
    CK(hipDeviceSynchronize());
    u64 st[7]; std::vector<float> out(RC);
    CK(hipMemcpy(out.data(), dout, RC * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < RC; ++i) err = std::fmax(err, std::fabs(out[i] - ref[i]));
    hipLaunchKernelGGL((layer_kernel<NW, STAMP>), dim3(grid), dim3(64 * NW), 0, 0, dW2, dWo, dpre, dbo, dh0, dout, dst, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st, dst, 56, hipMemcpyDeviceToHost));
    printf("NW=%d grid=%3d : %8.1f cycles/layer  %7.1f ns/layer  (clock %.2f GHz)  max err vs f64 host %.2e\n", NW, grid,
           (double)st[0] / n, (double)st[1] * 10.0 / n, (double)st[0] / ((double)st[1] * 10.0), err);
    if (STAMP) printf("      phases (cycles, wave 0): read+z+reduce %.0f | gate+write %.0f | barrier %.0f | read+o+reduce %.0f | 2 barriers+write %.0f\n",
           (double)st[2] / n, (double)st[3] / n, (double)st[4] / n, (double)st[5] / n, (double)st[6] / n);
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
    for (int grid : {1}) {
        run<8, false>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<4, false>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<8, true>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
        run<4, true>(dW2, dWo, dpre, dbo, dh0, dout, dst, n, grid, ref);
    }
    return 0;
}
