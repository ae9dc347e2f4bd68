// ubench_hop4.hip -- how many polls should be in flight?  (round-2 experiment; not part of the product)
// Model from ubench_hop3: a plain store is visible in the XCD's L2 ~270 ns after issue; a poll is an L2 round trip of ~117 ns and
// sees the granule only if it reaches the L2 after that, so with ONE poll at a time the hop lands anywhere in [~327, ~444] ns
// depending on the poll phase (mean ~385; the ring kernel sits at the bad end).  K polls in flight, issued round-robin at
// RTT / K intervals, should cut the expected wait to RTT / (2K) -- unless reads of the line delay the store's visibility.
// Polls land in reserved physical registers (v100..v115, capped compiler allocation), as in the ring kernel.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_hop4.bin scripts/ubench_hop4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)
struct P { u64* box; u64* stamps; unsigned* status; int* xcc; int n, a, b, delay, gap; };

#define RSV "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
__device__ __forceinline__ void put(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
template <int S> __device__ __forceinline__ void issue(const u64* p) {
    if constexpr (S == 0) asm volatile("global_load_dwordx2 v[100:101], %0, off sc1" :: "v"(p) : RSV);
    if constexpr (S == 1) asm volatile("global_load_dwordx2 v[102:103], %0, off sc1" :: "v"(p) : RSV);
    if constexpr (S == 2) asm volatile("global_load_dwordx2 v[104:105], %0, off sc1" :: "v"(p) : RSV);
    if constexpr (S == 3) asm volatile("global_load_dwordx2 v[106:107], %0, off sc1" :: "v"(p) : RSV);
}
// wait until at most Y younger polls are outstanding, copy slot S out
template <int S, int Y> __device__ __forceinline__ unsigned take_tag() {
    unsigned t;
#define TK(R) \
    if constexpr (Y == 0) asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, " R : "=v"(t) :: RSV); \
    if constexpr (Y == 1) asm volatile("s_waitcnt vmcnt(1)\n\tv_mov_b32 %0, " R : "=v"(t) :: RSV); \
    if constexpr (Y == 2) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, " R : "=v"(t) :: RSV); \
    if constexpr (Y == 3) asm volatile("s_waitcnt vmcnt(3)\n\tv_mov_b32 %0, " R : "=v"(t) :: RSV);
    if constexpr (S == 0) { TK("v101") }
    if constexpr (S == 1) { TK("v103") }
    if constexpr (S == 2) { TK("v105") }
    if constexpr (S == 3) { TK("v107") }
#undef TK
    return t;
}
__device__ __forceinline__ void pause(int gap) { for (int i = 0; i < gap; ++i) asm volatile("s_nop 15" ::: "memory"); }   // 16 cycles each

// K polls in flight: slot s is examined when K-1 younger ones are out, then re-issued
template <int K>
__device__ __forceinline__ bool wait_tag(const u64* rx, unsigned tag, int gap, unsigned* status) {
    issue<0>(rx);
    if constexpr (K > 1) { pause(gap); issue<1>(rx); }
    if constexpr (K > 2) { pause(gap); issue<2>(rx); }
    if constexpr (K > 3) { pause(gap); issue<3>(rx); }
    for (unsigned spins = 0; spins < (1u << 20); ++spins) {
#define STEP(S) { const unsigned t = take_tag<S, K - 1>(); if (__all(t == tag)) return true; issue<S>(rx); }
        STEP(0)
        if constexpr (K > 1) STEP(1)
        if constexpr (K > 2) STEP(2)
        if constexpr (K > 3) STEP(3)
#undef STEP
    }
    atomicCAS(status, 0u, 1u);
    return false;
}

template <int K>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(96))) hop_kernel(P p) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        p.xcc[blockIdx.x] = (int)(x & 0xf);
    }
    const bool ping = blockIdx.x == p.a, pong = blockIdx.x == p.b;
    if (!ping && !pong) return;
    const int lane = threadIdx.x;
    u64* tx = p.box + (ping ? 0 : 64) + lane;
    const u64* rx = p.box + (ping ? 64 : 0) + lane;
    u64 t0 = 0;
    if (ping) t0 = wall_clock64();
    for (int r = 1; r <= p.n; ++r) {
        for (int half = 0; half < 2; ++half) {
            const bool sender = (half == 0) == ping;
            if (sender) {
                put(tx, ((u64)r << 32) | (unsigned)(r * 7));
                for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(1);
            } else if (!wait_tag<K>(rx, (unsigned)r, p.gap, p.status)) return;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: RSV);
    if (ping && threadIdx.x == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); }
}

template <int K> double run(P p) {
    CK(hipMemset(p.box, 0, 256 * 8)); CK(hipMemset(p.status, 0, 64)); CK(hipMemset(p.stamps, 0, 64));
    hipLaunchKernelGGL(hop_kernel<K>, dim3(16), dim3(64), 0, 0, p);
    CK(hipDeviceSynchronize());
    u64 s[8]; unsigned status;
    CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&status, p.status, 4, hipMemcpyDeviceToHost));
    return status ? -1.0 : (double)(s[1] - s[0]) * 10.0 / (2.0 * p.n);
}

int main() {
    P p{};
    CK(hipMalloc(&p.box, 256 * 8)); CK(hipMalloc(&p.stamps, 64)); CK(hipMalloc(&p.status, 64)); CK(hipMalloc(&p.xcc, 64 * 4));
    p.n = 4000; p.a = 0; p.b = 8;
    printf("same-XCD ping-pong of 64 granules; K polls in flight (spaced `gap` x 16 cycles at start-up); ns/hop over initial delays D = 0..14 (x64 cycles)\n");
    printf(" K gap |   D=0     2     4     6     8    10    12    14 |  mean   min   max\n");
    for (int K = 1; K <= 4; ++K)
        for (int gap = 0; gap <= (K == 1 ? 0 : 8); gap += 4) {
            double sum = 0, mn = 1e9, mx = 0;
            printf(" %d  %2d  |", K, gap);
            for (int d = 0; d <= 14; d += 2) {
                p.delay = d; p.gap = gap;
                const double v = K == 1 ? run<1>(p) : K == 2 ? run<2>(p) : K == 3 ? run<3>(p) : run<4>(p);
                printf(" %5.0f", v);
                sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
            }
            printf(" | %5.0f %5.0f %5.0f\n", sum / 8, mn, mx);
            fflush(stdout);
        }
    return 0;
}
