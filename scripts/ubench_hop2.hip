// ubench_hop2.hip -- follow-up to ubench_hop.hip for the ring kernel's CU -> CU hand-off (next-round experiment, not part of the
// product).  The fine timeline of ring v11 puts a hop at ~290 ns store-to-visible + ~120 ns load return + ~30 ns LDS/barrier.
// Questions this answers on the GPU box (hipcc --offload-arch=gfx950 -O3 -o /tmp/hop2 scripts/ubench_hop2.hip && /tmp/hop2):
//   1. does another WRITE flavour land sooner?  plain | sc0 | atomic swap (no return) | atomic add (no return) | nt
//   2. does a SCALAR poll (s_load_dwordx2 glc: SQC -> L2, no TA/TCP) see the tag sooner than the vector sc1 poll?
//   3. how long is a lone poll round trip (vector sc1, scalar glc) with nothing to wait for?
// Same-XCD only (blocks 0 and 8); two workgroups bounce ONE granule {tag, value} N times; ns/hop = wall time / (2 N).
// Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

struct P {
    u64* box;            // [2][16] one 128-byte line per direction
    u64* stamps;         // [8]
    unsigned* status;
    int* xcc;
    int n, a, b, st, ld;
};

template <int ST> __device__ __forceinline__ void put(u64* p, u64 v) {
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_atomic_swap_x2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ u64 get(const u64* p) {
    u64 v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) {                                   // scalar path: uniform address, bypass the scalar cache
        const u64 pu = (u64)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
        const u64 ps = ((u64)hi << 32) | lo;
        u64 sv;
        asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(ps) : "memory");
        v = sv;
    }
    if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int ST, int LD>
__global__ void __launch_bounds__(64) hop1_kernel(P p) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        p.xcc[blockIdx.x] = (int)(x & 0xf);
    }
    const bool ping = blockIdx.x == p.a, pong = blockIdx.x == p.b;
    if (!ping && !pong) return;
    u64* tx = p.box + (ping ? 0 : 16);
    const u64* rx = p.box + (ping ? 16 : 0);
    u64 t0 = 0;
    if (ping) t0 = wall_clock64();
    for (int r = 1; r <= p.n; ++r) {
        for (int half = 0; half < 2; ++half) {
            const bool sender = (half == 0) == ping;
            if (sender) {
                if (threadIdx.x == 0) put<ST>(tx, ((u64)r << 32) | (unsigned)(r * 7));
            } else {
                unsigned spins = 0;
                for (;;) {
                    const u64 x = get<LD>(rx);
                    if ((unsigned)(x >> 32) == (unsigned)r) { if ((unsigned)x != (unsigned)(r * 7)) atomicCAS(p.status, 0u, 2u); break; }
                    if (++spins > (1u << 20)) { atomicCAS(p.status, 0u, 1u); return; }
                    if ((spins & 1023u) == 0 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
                }
            }
        }
    }
    if (ping && threadIdx.x == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); }
}

// lone round trips: N dependent polls of a line nobody writes
template <int LD>
__global__ void __launch_bounds__(64) rtt_kernel(P p) {
    if (blockIdx.x != 0) return;
    const u64 t0 = wall_clock64();
    u64 acc = 0;
    for (int r = 0; r < p.n; ++r) acc += get<LD>(p.box + (acc & 1));      // address depends on the previous value: serialised
    if (threadIdx.x == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); p.stamps[2] = acc; }
}

// lone round trips of WIDE polls: every lane its own granule(s): 64 x 8 B = 4 lines, 64 x 16 B = 8 lines (the ring kernel's poll)
template <int W>
__global__ void __launch_bounds__(64) rtt_wide_kernel(P p, u64* wide) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    const u64 t0 = wall_clock64();
    unsigned acc = 0;
    for (int r = 0; r < p.n; ++r) {
        if (W == 8) {
            u64 v;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(wide + lane + (acc & 1)) : "memory");
            acc += (unsigned)v;
        } else {
            typedef unsigned v4 __attribute__((ext_vector_type(4)));
            v4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(wide + 2 * lane + 2 * (acc & 1)) : "memory");
            acc += v.x + v.z;
        }
    }
    if (lane == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); p.stamps[2] = acc; }
}

typedef void (*kern_t)(P);
static kern_t pick(int st, int ld) {
#define C(S, L) if (st == S && ld == L) return hop1_kernel<S, L>;
    C(0, 0) C(0, 1) C(0, 2) C(1, 0) C(1, 1) C(2, 0) C(2, 1) C(3, 0) C(3, 1) C(4, 0) C(4, 1)
#undef C
    return nullptr;
}

int main() {
    P p{};
    CK(hipMalloc(&p.box, 64 * 8));
    CK(hipMalloc(&p.stamps, 8 * 8));
    CK(hipMalloc(&p.status, 64));
    CK(hipMalloc(&p.xcc, 64 * 4));
    p.n = 4000; p.a = 0; p.b = 8;
    const char* sn[] = {"plain", "sc0", "atomic_swap", "nt", "sc1"};
    const char* ln[] = {"vector sc1", "scalar glc", "vector sc0sc1"};
    printf("one granule, same XCD (blocks 0 and 8): ns per hop\n%-12s %-14s %9s  status\n", "store", "poll", "ns/hop");
    for (int st = 0; st < 5; ++st)
        for (int ld = 0; ld < 3; ++ld) {
            kern_t k = pick(st, ld);
            if (!k) continue;
            p.st = st; p.ld = ld;
            CK(hipMemset(p.box, 0, 64 * 8)); CK(hipMemset(p.status, 0, 64)); CK(hipMemset(p.stamps, 0, 64));
            hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, p);
            CK(hipDeviceSynchronize());
            u64 s[8]; unsigned status; int xcc[16];
            CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&status, p.status, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(xcc, p.xcc, 64, hipMemcpyDeviceToHost));
            printf("%-12s %-14s %9.1f  %u  (xcc %d -> %d)\n", sn[st], ln[ld], status == 1 ? -1.0 : (double)(s[1] - s[0]) * 10.0 / (2.0 * p.n), status,
                   xcc[p.a], xcc[p.b]);
            fflush(stdout);
        }
    printf("lone dependent poll round trips (nothing to wait for):\n");
    {
        CK(hipMemset(p.box, 0, 64 * 8));
        hipLaunchKernelGGL(rtt_kernel<0>, dim3(1), dim3(64), 0, 0, p);
        CK(hipDeviceSynchronize());
        u64 s[8]; CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost));
        printf("  vector sc1   %7.1f ns\n", (double)(s[1] - s[0]) * 10.0 / p.n);
        hipLaunchKernelGGL(rtt_kernel<1>, dim3(1), dim3(64), 0, 0, p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost));
        printf("  scalar glc   %7.1f ns\n", (double)(s[1] - s[0]) * 10.0 / p.n);
    }
    {
        u64* wide; CK(hipMalloc(&wide, 4096)); CK(hipMemset(wide, 0, 4096));
        u64 s[8];
        hipLaunchKernelGGL(rtt_wide_kernel<8>, dim3(1), dim3(64), 0, 0, p, wide);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost));
        printf("  vector sc1, 64 lanes x 8 B (4 lines)   %7.1f ns\n", (double)(s[1] - s[0]) * 10.0 / p.n);
        hipLaunchKernelGGL(rtt_wide_kernel<16>, dim3(1), dim3(64), 0, 0, p, wide);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(s, p.stamps, 64, hipMemcpyDeviceToHost));
        printf("  vector sc1, 64 lanes x 16 B (8 lines)  %7.1f ns\n", (double)(s[1] - s[0]) * 10.0 / p.n);
    }
    return 0;
}
