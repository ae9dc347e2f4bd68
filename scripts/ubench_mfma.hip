// ubench_mfma.hip -- checks the lane layout of v_mfma_f32_32x32x2_f32 assumed by csrc/wnv_forward.hip and its issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma.bin scripts/ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
using u64 = unsigned long long;
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

// C[32][32] = A[32][K] * B[K][32], one wave
__global__ void k(const float* A, const float* B, float* C, int K, u64* cyc, int reps) {
    const int l = threadIdx.x;
    f16v acc = {0};
    u64 t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
        for (int k0 = 0; k0 < K; k0 += 2) {
            const float a = A[(l % 32) * K + k0 + l / 32];
            const float b = B[(k0 + l / 32) * 32 + l % 32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    u64 t1 = __builtin_readcyclecounter();
    for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * (l / 32) + (v % 4), j = l % 32;
        C[i * 32 + j] = acc[v];
    }
    if (l == 0) cyc[0] = t1 - t0;
}
int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(K * 32), C(1024), R(1024, 0.f);
    srand(3);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : B) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; R[i * 32 + j] = (float)s; }
    float *dA, *dB, *dC; u64* dc;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 4096)); CK(hipMalloc(&dc, 64));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, dc, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(C[i] - R[i]));
    printf("32x32x2 f32 MFMA layout check: max |C - ref| = %.3e (%s)\n", err, err < 1e-5 ? "layout OK" : "LAYOUT WRONG");
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, dc, 200);
    CK(hipDeviceSynchronize());
    u64 cy; CK(hipMemcpy(&cy, dc, 8, hipMemcpyDeviceToHost));
    printf("dependent-accumulator MFMA + 2 L2 loads per step: %.1f cycles per MFMA\n", (double)cy / (200.0 * K / 2));
    return 0;
}
