// exp_fused_upsample.hip -- NEXT-ROUND EXPERIMENT (not part of the product, not validated on a GPU yet): the whole conditioning
// upsampler of the reference -- conv_in (valid Conv1d, k = 2 cin_pad + 1, no bias; upsample.py:69-85) followed by n stages of
// [nearest stretch x s, FIR of 2 s + 1 taps with zero padding, one filter shared by all channels] (upsample.py:29-66) and the
// (B, C, T) -> (B, T, C) transpose of wavenet.py:277-278 -- as ONE kernel: a workgroup owns F conv_in output frames of one
// utterance, keeps every intermediate level of its tile (plus the one-sample halo each FIR needs) in LDS and streams the
// time-major output in memory order.  Self-checking against a double-precision CPU restatement in this file.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fused_up scripts/exp_fused_upsample.hip && /tmp/fused_up
//
// Window recurrence: output j of a stage with scale s reads input indices floor((j + m - s) / s), m = 0 .. 2 s, i.e.
// floor(j / s) - 1 .. floor(j / s) + 1; so a tile that needs [lo_k, hi_k] at level k needs [floor(lo_k / s) - 1, floor(hi_k / s) + 1]
// at level k - 1.  Indices outside a level's valid range [0, T_k) are ZERO (the reference zero-pads every stage), never extrapolated.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)
constexpr int MAXS = 4;            // stages
constexpr int FT = 256;

struct Args {
    const float* c;                // (B, cin, Tin)   Tin = Tc + ks - 1
    const float* wconv;            // (cin, cin, ks)
    const float* wfir;             // stage k at wfir + 64 k, 2 s_k + 1 taps
    float* out;                    // (B, T, cin)
    int B, cin, Tin, ks, Tc, n, F, tiles;
    int s[MAXS];
    long long Tk[MAXS + 1];        // valid length of every level: Tk[0] = Tc, Tk[k + 1] = Tk[k] s[k]
    int lds_off[MAXS + 1];         // float offset of level k's window (levels 0 .. n-1 live in LDS) ; row stride = wlen[k] | 1
    int wlen[MAXS + 1];            // window length of level k for a full tile
};

// floor division for possibly negative numerators
__host__ __device__ inline long long fdiv(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

__global__ void __launch_bounds__(FT) fused_upsample_kernel(const Args a) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.tiles;
    const long long f0 = (long long)(blockIdx.x % a.tiles) * a.F;                 // first conv_in frame of the tile
    // window origins, top (level n = output) down to level 0
    long long lo[MAXS + 1], hi[MAXS + 1];
    long long span = a.F;
    for (int k = 0; k < a.n; ++k) span *= a.s[k];
    lo[a.n] = f0 * (span / a.F);
    hi[a.n] = lo[a.n] + span - 1;
    for (int k = a.n; k >= 1; --k) {
        lo[k - 1] = fdiv(lo[k], a.s[k - 1]) - 1;
        hi[k - 1] = fdiv(hi[k], a.s[k - 1]) + 1;
    }
    // ---- level 0: conv_in on the frames [lo0, hi0] (zero outside [0, Tc)) ----------------------------------------------------
    {
        const int n0 = (int)(hi[0] - lo[0] + 1), st0 = a.wlen[0] | 1;
        float* L0 = lds + a.lds_off[0];
        const float* cb = a.c + (size_t)b * a.cin * a.Tin;
        for (int i = tid; i < a.cin * n0; i += FT) {
            const int o = i / n0, p = i - o * n0;
            const long long f = lo[0] + p;
            float acc = 0.f;
            if (f >= 0 && f < a.Tc) {
                const float* wo = a.wconv + (size_t)o * a.cin * a.ks;
                for (int ci = 0; ci < a.cin; ++ci)
                    for (int k = 0; k < a.ks; ++k) acc = fmaf(wo[ci * a.ks + k], cb[(size_t)ci * a.Tin + f + k], acc);
            }
            L0[o * st0 + p] = acc;
        }
    }
    __syncthreads();
    // ---- levels 1 .. n-1 in LDS ----------------------------------------------------------------------------------------------
    for (int k = 1; k < a.n; ++k) {
        const int s = a.s[k - 1];
        const int nk = (int)(hi[k] - lo[k] + 1), stk = a.wlen[k] | 1, stp = a.wlen[k - 1] | 1;
        const float* Lp = lds + a.lds_off[k - 1];
        float* Lk = lds + a.lds_off[k];
        const float* w = a.wfir + 64 * (k - 1);
        for (int i = tid; i < a.cin * nk; i += FT) {
            const int ch = i / nk, p = i - ch * nk;
            const long long j = lo[k] + p;
            float acc = 0.f;
            if (j >= 0 && j < a.Tk[k]) {
                for (int m = 0; m <= 2 * s; ++m) {
                    const long long q = fdiv(j + m - s, s);                           // input index at level k-1
                    const float v = (q >= 0 && q < a.Tk[k - 1]) ? Lp[ch * stp + (int)(q - lo[k - 1])] : 0.f;
                    acc = fmaf(w[m], v, acc);
                }
            }
            Lk[ch * stk + p] = acc;
        }
        __syncthreads();
    }
    // ---- last stage: stream the time-major output in memory order -------------------------------------------------------------
    {
        const int k = a.n, s = a.s[k - 1], stp = a.wlen[k - 1] | 1;
        const float* Lp = lds + a.lds_off[k - 1];
        const float* w = a.wfir + 64 * (k - 1);
        const long long T = a.Tk[k];
        const long long nout = (hi[k] < T ? hi[k] : T - 1) - lo[k] + 1;
        float* dst = a.out + ((size_t)b * T + lo[k]) * a.cin;
        for (long long i = tid; i < nout * a.cin; i += FT) {
            const int tl = (int)(i / a.cin), ch = (int)(i - (long long)tl * a.cin);
            const long long j = lo[k] + tl;
            float acc = 0.f;
            for (int m = 0; m <= 2 * s; ++m) {
                const long long q = fdiv(j + m - s, s);
                const float v = (q >= 0 && q < a.Tk[k - 1]) ? Lp[ch * stp + (int)(q - lo[k - 1])] : 0.f;
                acc = fmaf(w[m], v, acc);
            }
            dst[i] = acc;
        }
    }
}

int main() {
    const int B = 8, cin = 80, ks = 5, Tc = 94, n = 4, F = 4;
    const int s[MAXS] = {4, 4, 4, 4};
    const int Tin = Tc + ks - 1;
    std::vector<float> c((size_t)B * cin * Tin), wconv((size_t)cin * cin * ks), wfir(64 * MAXS, 0.f);
    srand(1);
    auto rnd = [] { return (float)((double)rand() / (double)RAND_MAX) - 0.5f; };
    for (auto& v : c) v = rnd() * 2;
    for (auto& v : wconv) v = rnd() * 0.2f;
    for (int k = 0; k < n; ++k) for (int m = 0; m <= 2 * s[k]; ++m) wfir[64 * k + m] = rnd();
    // CPU reference in double
    long long Tk[MAXS + 1]; Tk[0] = Tc; for (int k = 0; k < n; ++k) Tk[k + 1] = Tk[k] * s[k];
    const long long T = Tk[n];
    std::vector<double> cur((size_t)B * cin * Tc), nxt;
    for (int b = 0; b < B; ++b) for (int o = 0; o < cin; ++o) for (int f = 0; f < Tc; ++f) {
        double acc = 0;
        for (int ci = 0; ci < cin; ++ci) for (int k = 0; k < ks; ++k) acc += (double)wconv[((size_t)o * cin + ci) * ks + k] * c[((size_t)b * cin + ci) * Tin + f + k];
        cur[((size_t)b * cin + o) * Tc + f] = acc;
    }
    for (int k = 0; k < n; ++k) {
        nxt.assign((size_t)B * cin * Tk[k + 1], 0.0);
        for (int r = 0; r < B * cin; ++r) for (long long j = 0; j < Tk[k + 1]; ++j) {
            double acc = 0;
            for (int m = 0; m <= 2 * s[k]; ++m) {
                const long long q = j + m - s[k];
                if (q >= 0 && q < Tk[k + 1]) acc += (double)wfir[64 * k + m] * cur[(size_t)r * Tk[k] + q / s[k]];
            }
            nxt[(size_t)r * Tk[k + 1] + j] = acc;
        }
        cur.swap(nxt);
    }
    // device
    Args a{};
    a.B = B; a.cin = cin; a.Tin = Tin; a.ks = ks; a.Tc = Tc; a.n = n; a.F = F; a.tiles = (Tc + F - 1) / F;
    for (int k = 0; k < n; ++k) a.s[k] = s[k];
    for (int k = 0; k <= n; ++k) a.Tk[k] = Tk[k];
    {   // window lengths of a full tile, top down
        long long lo = 0, hi = 1; for (int k = 0; k < n; ++k) hi *= s[k]; hi = hi * F - 1;
        long long wl[MAXS + 1]; wl[n] = hi - lo + 1;
        for (int k = n; k >= 1; --k) { lo = fdiv(lo, s[k - 1]) - 1; hi = fdiv(hi, s[k - 1]) + 1; wl[k - 1] = hi - lo + 1; }
        int off = 0;
        for (int k = 0; k < n; ++k) { a.wlen[k] = (int)wl[k]; a.lds_off[k] = off; off += cin * ((int)wl[k] | 1); }
        a.wlen[n] = (int)wl[n];
        printf("windows per tile (levels 0..%d):", n); for (int k = 0; k <= n; ++k) printf(" %lld", wl[k]); printf("   LDS %.1f KiB\n", off * 4 / 1024.0);
        const size_t ldsb = (size_t)off * sizeof(float);
        float *dc, *dw, *df, *dout;
        CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&dw, wconv.size() * 4)); CK(hipMalloc(&df, wfir.size() * 4));
        CK(hipMalloc(&dout, (size_t)B * T * cin * 4));
        CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, wconv.data(), wconv.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(df, wfir.data(), wfir.size() * 4, hipMemcpyHostToDevice));
        a.c = dc; a.wconv = dw; a.wfir = df; a.out = dout;
        CK(hipFuncSetAttribute((const void*)fused_upsample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(fused_upsample_kernel, dim3(B * a.tiles), dim3(FT), ldsb, 0, a);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(fused_upsample_kernel, dim3(B * a.tiles), dim3(FT), ldsb, 0, a);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<float> out((size_t)B * T * cin);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int b = 0; b < B; ++b) for (int ch = 0; ch < cin; ++ch) for (long long t = 0; t < T; ++t) {
            const double ref = cur[((size_t)b * cin + ch) * T + t], got = out[((size_t)b * T + t) * cin + ch];
            maxerr = std::fmax(maxerr, std::fabs(ref - got)); maxref = std::fmax(maxref, std::fabs(ref));
        }
        const double bytes = 4.0 * ((double)B * T * cin + c.size());
        printf("fused upsampler: %d x %lld samples x %d ch: %.4f ms, %.1f GB/s algorithmic, max |err| %.3e (max |ref| %.2f) -> %s\n", B, T, cin, ms,
               bytes / ms / 1e6, maxerr, maxref, maxerr <= 2e-5 * std::fmax(1.0, maxref) ? "PASS" : "FAIL");
    }
    return 0;
}
