// ubench_hop3.hip -- does POLLING delay the store it waits for?  (round-2 experiment; not part of the product)
// ubench_hop2 put a same-XCD one-granule hop at ~444 ns of which a lone sc1 poll round trip is only ~117 ns: ~330 ns pass before a
// plain store is visible in the XCD's L2, and the ring kernel's sweep of poll depths showed the hop getting SLOWER when the line
// is polled harder.  If polling the line delays the store, a receiver that stays silent until shortly before the expected
// arrival would see it earlier.  Two workgroups on one XCD bounce one granule N times; the receiver sleeps D x 64 cycles
// (s_sleep D) after its own send before it issues its first poll; then one poll at a time as in the ring kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hop3 scripts/ubench_hop3.hip && /tmp/hop3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

struct P { u64* box; u64* stamps; unsigned* status; int* xcc; int n, a, b, delay, vec; };

__device__ __forceinline__ void put(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64 get(const u64* p) {
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// vec = 0: lane 0 sends one granule; vec = 1: 64 lanes send 64 granules (512 B, the ring kernel's per-wave share is 16 x 8 B)
__global__ void __launch_bounds__(64) hop_kernel(P p) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        p.xcc[blockIdx.x] = (int)(x & 0xf);
    }
    const bool ping = blockIdx.x == p.a, pong = blockIdx.x == p.b;
    if (!ping && !pong) return;
    const int lane = threadIdx.x;
    u64* tx = p.box + (ping ? 0 : 64) + (p.vec ? lane : 0);
    const u64* rx = p.box + (ping ? 64 : 0) + (p.vec ? lane : 0);
    const bool active = p.vec || lane == 0;
    u64 t0 = 0;
    if (ping) t0 = wall_clock64();
    for (int r = 1; r <= p.n; ++r) {
        for (int half = 0; half < 2; ++half) {
            const bool sender = (half == 0) == ping;
            if (sender) {
                if (active) put(tx, ((u64)r << 32) | (unsigned)(r * 7));
                for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(1);       // silent until the answer is about due
            } else {
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    if (active) { const u64 x = get(rx); ok = (unsigned)(x >> 32) == (unsigned)r; }
                    if (__all(ok)) break;
                    if (++spins > (1u << 20)) { atomicCAS(p.status, 0u, 1u); return; }
                }
            }
        }
    }
    if (ping && threadIdx.x == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); }
}

int main() {
    P p{};
    CK(hipMalloc(&p.box, 256 * 8)); CK(hipMalloc(&p.stamps, 64)); CK(hipMalloc(&p.status, 64)); CK(hipMalloc(&p.xcc, 64 * 4));
    p.n = 4000; p.a = 0; p.b = 8;
    printf("same-XCD ping-pong, receiver silent for D x 64 cycles after its own send (the partner's answer takes one hop + its turnaround)\n");
    for (int vec = 0; vec < 2; ++vec) {
        printf("%s\n  D   ns/hop\n", vec ? "64 granules per message (one per lane)" : "one granule per message");
        for (int d = 0; d <= 40; d += (d < 24 ? 2 : 4)) {
            p.delay = d; p.vec = vec;
            CK(hipMemset(p.box, 0, 256 * 8)); CK(hipMemset(p.status, 0, 64)); CK(hipMemset(p.stamps, 0, 64));
            hipLaunchKernelGGL(hop_kernel, dim3(16), dim3(64), 0, 0, p);
            CK(hipDeviceSynchronize());
            u64 s[8]; unsigned status; int xcc[16];
            CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&status, p.status, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(xcc, p.xcc, 64, hipMemcpyDeviceToHost));
            printf("  %2d  %7.1f  %s (xcc %d -> %d)\n", d, (double)(s[1] - s[0]) * 10.0 / (2.0 * p.n), status ? "TIMEOUT" : "", xcc[p.a], xcc[p.b]);
            fflush(stdout);
        }
    }
    return 0;
}
