// ubench_lds.hip -- latency of a batch of N back-to-back LDS reads issued by each wave (then s_waitcnt), by width
// and by how many waves do it at once.  hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_lds.bin scripts/ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int W, int N>   // W = bytes per lane per read (4, 8, 16)
__global__ void k(float* out, u64* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    const int q = tid & 3;
    u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float* base = lds + q * 36 + (it & 7) * 256;
        if (W == 16) {
            float4 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = reinterpret_cast<const float4*>(base)[i];
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i].x + v[i].w;
        } else if (W == 8) {
            float2 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = reinterpret_cast<const float2*>(base)[i];
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i].x + v[i].y;
        } else {
            float v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = base[i];
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i];
        }
        asm volatile("" : "+v"(acc));
    }
    u64 t1 = __builtin_readcyclecounter();
    out[tid] = acc;
    if (tid == 0) cyc[0] = t1 - t0;
}
template <int W, int N> void run(float* d, u64* c) {
    printf("ds_read_b%-3d x %2d :", W * 8, N);
    for (int nw : {1, 4, 8, 16}) {
        hipLaunchKernelGGL((k<W, N>), dim3(1), dim3(64 * nw), 0, 0, d, c, 2000);
        CK(hipDeviceSynchronize());
        u64 cy; CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
        printf("  %2d waves: %6.1f cyc/batch", nw, (double)cy / 2000);
    }
    printf("\n");
}
int main() {
    float* d; u64* c;
    CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&c, 64));
    run<16, 1>(d, c); run<16, 2>(d, c); run<16, 4>(d, c); run<16, 8>(d, c); run<16, 16>(d, c);
    run<8, 1>(d, c); run<8, 4>(d, c); run<8, 8>(d, c); run<8, 16>(d, c);
    run<4, 1>(d, c); run<4, 8>(d, c); run<4, 16>(d, c);
    return 0;
}
