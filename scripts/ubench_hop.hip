// ubench_hop.hip -- CU -> CU hand-off latency on gfx950 for the ring kernel's 128-value activation vector.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_hop.bin scripts/ubench_hop.hip && scripts/ubench_hop.bin
//
// Two workgroups (ping, pong) bounce a 128-granule message N times; per-hop latency = wall time / (2 N).
// Every received value is checked (value == f(round)), every spin is bounded.  Variants:
//   store flavour : 0 = sc1 (write-through, agent atomics lower to this)   1 = plain   2 = sc0 sc1   3 = sc0
//   load  flavour : 0 = sc1                                                1 = sc0 sc1  2 = plain (expected stale)
//   placement     : same XCD (blocks 0 and 8) or cross XCD (blocks 0 and 1); the XCC id is read back and printed
//   producer shape: 0 = 2 waves x 64 lanes   1 = 8 waves x 16 lanes (every 4th lane, as the ring kernel's q==0 lanes)
//   consumer shape: 0 = 2 waves poll 64 granules each -> LDS -> barrier    1 = every wave polls all 128 (dwordx4 per lane), no barrier
//   poll pipelining: P loads in flight (1 = issue, wait, check)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

struct P {
    u64* box;       // [2][128] granules (direction 0: ping->pong, 1: pong->ping)
    u64* stamps;    // [4]
    unsigned* status;
    int* xcc;       // [grid]
    int n, a, b;    // rounds, block ids of ping and pong
    int st, ld, pshape, cshape;
};

template <int ST> __device__ __forceinline__ void st8(u64* p, u64 v) {
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ u64 ld8(const u64* p) {
    u64 v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int LD> __device__ __forceinline__ void ld16(const u64* p, u64& a, u64& b) {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 v;
    if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    a = ((u64)v.y << 32) | v.x;
    b = ((u64)v.w << 32) | v.z;
}

__device__ __forceinline__ float payload(int round, int i) { return (float)(round * 3 + i); }

template <int ST, int LD>
__global__ void __launch_bounds__(512) hop_kernel(P p) {
    __shared__ float hs[128];
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        p.xcc[blockIdx.x] = (int)(x & 0xf);
        bad = 0;
    }
    __syncthreads();
    const bool ping = blockIdx.x == p.a, pong = blockIdx.x == p.b;
    if (!ping && !pong) return;
    u64* tx = p.box + (ping ? 0 : 128);
    const u64* rx = p.box + (ping ? 128 : 0);
    u64 t0 = 0;
    if (ping && tid == 0) t0 = wall_clock64();
    for (int r = 1; r <= p.n; ++r) {
        for (int half = 0; half < 2; ++half) {
            const bool sender = (half == 0) == ping;
            if (sender) {
                // ---- produce 128 granules
                if (p.pshape == 0) { if (tid < 128) st8<ST>(tx + tid, ((u64)r << 32) | __float_as_uint(payload(r, tid))); }
                else { if ((lane & 3) == 0) { const int i = wave * 16 + (lane >> 2); st8<ST>(tx + i, ((u64)r << 32) | __float_as_uint(payload(r, i))); } }
            } else {
                // ---- consume
                if (p.cshape == 0) {
                    if (wave < 2) {
                        unsigned spins = 0;
                        float v = 0.f;
                        for (;;) {
                            const u64 x = ld8<LD>(rx + tid);
                            v = __uint_as_float((unsigned)x);
                            if (__all((unsigned)(x >> 32) == (unsigned)r)) break;
                            if (++spins > (1u << 18)) { bad = 1; atomicCAS(p.status, 0u, 1u); break; }
                            if ((spins & 255u) == 0 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
                        }
                        hs[tid] = v;
                    }
                    __syncthreads();
                    if (bad) return;
                    if (tid < 128 && hs[tid] != payload(r, tid)) { atomicCAS(p.status, 0u, 2u); }
                } else {
                    unsigned spins = 0;
                    u64 x0, x1;
                    for (;;) {
                        ld16<LD>(rx + 2 * lane, x0, x1);
                        if (__all((unsigned)(x0 >> 32) == (unsigned)r && (unsigned)(x1 >> 32) == (unsigned)r)) break;
                        if (++spins > (1u << 18)) { atomicCAS(p.status, 0u, 1u); return; }
                        if ((spins & 255u) == 0 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
                    }
                    if (__uint_as_float((unsigned)x0) != payload(r, 2 * lane) || __uint_as_float((unsigned)x1) != payload(r, 2 * lane + 1))
                        atomicCAS(p.status, 0u, 2u);
                }
            }
        }
    }
    if (ping && tid == 0) { p.stamps[0] = t0; p.stamps[1] = wall_clock64(); }
}

typedef void (*kern_t)(P);
static kern_t pick(int st, int ld) {
#define C(S, L) if (st == S && ld == L) return hop_kernel<S, L>;
    C(0, 0) C(0, 1) C(0, 2) C(1, 0) C(1, 1) C(1, 2) C(2, 0) C(2, 1) C(2, 2) C(3, 0) C(3, 1) C(3, 2)
#undef C
    return nullptr;
}

int main() {
    P p{};
    CK(hipMalloc(&p.box, 256 * 8));
    CK(hipMalloc(&p.stamps, 4 * 8));
    CK(hipMalloc(&p.status, 64));
    CK(hipMalloc(&p.xcc, 64 * 4));
    p.n = 4000;
    const char* sn[] = {"sc1", "plain", "sc0sc1", "sc0"};
    const char* ln[] = {"sc1", "sc0sc1", "plain"};
    printf("%-8s %-8s %-6s %-7s %-7s %9s  %s\n", "store", "load", "place", "pshape", "cshape", "ns/hop", "status (0 ok, 1 timeout, 2 wrong value)");
    for (int place = 0; place < 2; ++place)
        for (int st = 0; st < 4; ++st)
            for (int ld = 0; ld < 3; ++ld)
                for (int shape = 0; shape < 3; ++shape) {
                    if (shape > 0 && !(ld == 0 && (st == 0 || st == 1))) continue;
                    p.st = st; p.ld = ld; p.a = 0; p.b = place == 0 ? 8 : 1;
                    p.pshape = shape >= 1; p.cshape = shape == 2;
                    CK(hipMemset(p.box, 0, 256 * 8));
                    CK(hipMemset(p.status, 0, 64));
                    CK(hipMemset(p.stamps, 0, 32));
                    hipLaunchKernelGGL(pick(st, ld), dim3(16), dim3(512), 0, 0, p);
                    CK(hipDeviceSynchronize());
                    u64 s[4]; unsigned status; int xcc[16];
                    CK(hipMemcpy(s, p.stamps, 32, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&status, p.status, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(xcc, p.xcc, 64, hipMemcpyDeviceToHost));
                    const double ns = status == 1 ? -1.0 : (double)(s[1] - s[0]) * 10.0 / (2.0 * p.n);
                    printf("%-8s %-8s %-6s %-7d %-7d %9.1f  %u   (xcc %d -> %d)\n", sn[st], ln[ld], place == 0 ? "same" : "cross",
                           p.pshape, p.cshape, ns, status, xcc[p.a], xcc[p.b]);
                    fflush(stdout);
                }
    return 0;
}
