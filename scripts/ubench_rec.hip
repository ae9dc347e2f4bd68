// ubench_rec.hip -- the RECORD hand-off of wnv_ring.hip in isolation: workgroup W (block 1) publishes a 256-value record of tagged granules
// per step (tap-style: 64 lanes x two 16-byte write-through stores), workgroup R (block 0) receives it with rec_recv<2> and answers with
// a 128-value record (stage-style: one 16-byte store per lane), which W receives with rec_recv<1>.  Blocks 0 and 1 land on different
// XCDs.  Prints the round-trip time and checks every value.  hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_rec.bin scripts/ubench_rec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using u64 = unsigned long long;
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)
constexpr unsigned SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ void st_granule2(u64* p, unsigned tag, float v0, float v1) {
    typedef unsigned u4s __attribute__((ext_vector_type(4)));
    const u4s x = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ u4v ld16_sc1(const u64* p) {
    u4v x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
__device__ __forceinline__ void ld16x2_sc1(const u64* p, const u64* q, u4v& a, u4v& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory");
}
template <int NG>
__device__ __forceinline__ bool rec_recv(const u64* rec, unsigned tag, float (&v)[2 * NG], unsigned* status, unsigned code, int lane) {
    unsigned spins = 0;
    for (;;) {
        const u4v x = ld16_sc1(rec);
        if (x.y == tag) break;
        if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > (SPIN_LIMIT >> 3)) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
        __builtin_amdgcn_s_sleep(8);
    }
    spins = 0;
    for (;;) {
        bool ok;
        if constexpr (NG == 1) {
            const u4v x = ld16_sc1(rec + 2 * lane);
            v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.z);
            ok = x.y == tag && x.w == tag;
        } else {
            u4v x, y;
            ld16x2_sc1(rec + 2 * lane, rec + 128 + 2 * lane, x, y);
            v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.z); v[2] = __uint_as_float(y.x); v[3] = __uint_as_float(y.z);
            ok = x.y == tag && x.w == tag && y.y == tag && y.w == tag;
        }
        if (__all(ok)) return true;
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > SPIN_LIMIT) { if (lane == 0) atomicCAS(status, 0u, code); return false; }
        }
    }
}
__global__ void __launch_bounds__(512) k(u64* pre, u64* h, unsigned* status, unsigned* bad, u64* clk, int T, unsigned base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x >= 2) return;
    const u64 t0 = wall_clock64();
    for (int t = 0; t < T; ++t) {
        const unsigned tag = base + t + 1u;
        u64* prec = pre + (size_t)(t & 1) * 256;
        u64* hrec = h + (size_t)(t & 1) * 128;
        if (blockIdx.x == 1) {                       // W: publish pre[t] (wave 0), then wait for h[t] (wave 3)
            if (wave == 0) {
                st_granule2(prec + 2 * lane, tag, 4.f * lane + t, 4.f * lane + 1 + t);
                st_granule2(prec + 128 + 2 * lane, tag, 4.f * lane + 2 + t, 4.f * lane + 3 + t);
            }
            if (wave == 3) {
                float v[2];
                if (!rec_recv<1>(hrec, tag, v, status, 0x600u, lane)) return;
                if (v[0] != 2.f * lane - t || v[1] != 2.f * lane + 1 - t) atomicAdd(bad, 1u);
            }
            __syncthreads();
        } else {                                     // R: wait for pre[t] (wave 0), answer with h[t]
            if (wave == 0) {
                float v[4];
                if (!rec_recv<2>(prec, tag, v, status, 0x700u, lane)) return;
                for (int e = 0; e < 4; ++e) if (v[e] != 4.f * lane + e + t) atomicAdd(bad, 1u);
                st_granule2(hrec + 2 * lane, tag, 2.f * lane - t, 2.f * lane + 1 - t);
            }
            __syncthreads();
        }
    }
    if (tid == 0) clk[blockIdx.x] = wall_clock64() - t0;
}
int main() {
    u64 *pre, *h, *clk; unsigned *status, *bad;
    CK(hipMalloc(&pre, 2 * 256 * 8)); CK(hipMalloc(&h, 2 * 128 * 8)); CK(hipMalloc(&clk, 64)); CK(hipMalloc(&status, 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemset(pre, 0, 2 * 256 * 8)); CK(hipMemset(h, 0, 2 * 128 * 8)); CK(hipMemset(status, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(clk, 0, 64));
    const int T = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(2), dim3(512), 0, 0, pre, h, status, bad, clk, T, (unsigned)(rep * (T + 1)));
        CK(hipDeviceSynchronize());
        unsigned st, bd; u64 c[2];
        CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bd, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
        printf("rep %d: status 0x%x, %u wrong values, %.1f ns per round trip (two cross-XCD records)\n", rep, st, bd, (double)c[0] * 10.0 / T);
    }
    return 0;
}
