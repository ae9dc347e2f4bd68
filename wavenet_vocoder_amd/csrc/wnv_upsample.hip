// wnv_upsample.hip -- one-shot local-conditioning upsampler (the prologue of the hot path).
//
// Reference: upsample.ConvInUpsampleNetwork.forward (upsample.py:83-85) = conv_in (valid Conv1d, k = 2*cin_pad+1,
// no bias) then UpsampleNetwork.forward (upsample.py:51-66) = per scale s: Stretch2d nearest x s along time
// (upsample.py:19-21) followed by Conv2d(1, 1, (1, 2s+1), padding=(0, s), bias=False) -- ONE filter shared by all
// mel bins -- and finally the (B, C, T) -> (B, T, C) transpose of wavenet.py:277-278.
//
// These are HBM-bound streaming kernels (the output, 4*cin bytes per audio sample per utterance, dominates).
// The LAST stage -- 80 % of the bytes: it reads (B, cin, T/s) and writes the time-major (B, T, cin) the sample loop reads one
// contiguous row of per step -- is LDS-tiled (wnv_stretch_fir_tm_kernel): a workgroup stages the [cin][64 + 2] input window of
// 64 s output samples with row-contiguous reads, then streams the outputs in memory order (consecutive lanes = consecutive
// channels of one sample, then the next sample): every global access is a full-line burst.  The earlier stages are 4x, 16x, ...
// smaller and keep the one-element-per-thread form.
#include "wnv_internal.h"

namespace {

__global__ void wnv_conv_in_kernel(const float* __restrict__ c, const float* __restrict__ w,
                                   float* __restrict__ out, int B, int cin, int Tin, int ks) {
    const int Tout = Tin - ks + 1;
    const long long total = (long long)B * cin * Tout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % Tout);
        const int o = (int)((i / Tout) % cin);
        const int b = (int)(i / ((long long)Tout * cin));
        const float* cb = c + (size_t)b * cin * Tin;
        const float* wo = w + (size_t)o * cin * ks;
        float acc = 0.f;
        for (int ci = 0; ci < cin; ++ci)
            for (int k = 0; k < ks; ++k) acc = fmaf(wo[ci * ks + k], cb[(size_t)ci * Tin + f + k], acc);
        out[i] = acc;
    }
}

// the optional per-stage activation of upsample.py:47-49 (`getattr(nn, upsample_activation)(**params)`): the ones with one parameter at most
__device__ __forceinline__ float up_act(float x, int act, float a) {
    switch (act) {
        case 1: return fmaxf(x, 0.f);                                   // nn.ReLU
        case 2: return x >= 0.f ? x : a * x;                            // nn.LeakyReLU(negative_slope = a)
        case 3: return tanhf(x);                                        // nn.Tanh
        case 4: return 1.0f / (1.0f + expf(-x));                        // nn.Sigmoid
        case 5: return x > 0.f ? x : a * expm1f(x);                     // nn.ELU(alpha = a)
        default: return x;
    }
}

// out[row][j] = act( sum_f sum_{m=0..2s} w[f][m] * rep[row + f - fk/2][j + m - s] ),  rep[r][q] = in[r][q / s] for 0 <= q < Tin*s and
// 0 <= r < cin, else 0: Conv2d(1, 1, (fk, 2s+1), padding=((fk-1)/2, s)) over the nearest-stretched map (upsample.py:38-45); fk = 1 in
// every reference preset
// sample q of the stretched row (upsample.py:19-21, F.interpolate(x, scale_factor=(1, s), mode=...)): nearest = in[q / s]; bilinear with
// align_corners = False (torch's default): source position ratio (q + 0.5) - 0.5 with ratio = 1 / s, clamped at 0, the right neighbour
// clamped at the last sample (ATen UpSample.h: area_pixel_compute_source_index / compute_source_index_and_lambda); the mel-bin axis has
// scale 1, i.e. is copied.  bicubic (mode 2): the same source position WITHOUT the clamp at 0, four taps floor(src) - 1 .. + 2 with their
// indices clamped to the row, Keys' kernel with A = -0.75 (ATen UpSample.h: cubic_convolution1/2, get_cubic_upsample_coefficients,
// upsample_get_value_bounded); along the mel-bin axis (scale 1) the fraction is 0 and the coefficients are exactly {0, 1, 0, 0}: a copy.
// ("area" = adaptive average pooling and "nearest-exact" pick the same single sample as "nearest" for an integer factor: the host maps
//  them to mode 0, tests/test_host_cpu.py checks that against F.interpolate.)
__device__ __forceinline__ float stretched(const float* row, long long q, int scale, long long Tin, int mode) {
    if (mode == 0) return row[q / scale];
    const float ratio = (float)(1.0 / (double)scale);
    if (mode == 2) {
        const float A = -0.75f;
        const float real = ratio * ((float)q + 0.5f) - 0.5f;
        const float fl = floorf(real);
        const long long ix = (long long)fl;
        const float t = fminf(fmaxf(real - fl, 0.f), 1.f);
        const float x0 = t + 1.0f, x2 = 1.0f - t, x3 = 2.0f - t;
        const float c0 = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
        const float c1 = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
        const float c2 = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
        const float c3 = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
        const long long hi = Tin - 1;
        const long long i0 = min(max(ix - 1, 0LL), hi), i1 = min(max(ix, 0LL), hi), i2 = min(max(ix + 1, 0LL), hi), i3 = min(max(ix + 2, 0LL), hi);
        float acc = row[i0] * c0;
        acc += row[i1] * c1; acc += row[i2] * c2; acc += row[i3] * c3;
        return acc;
    }
    float src = ratio * ((float)q + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const long long i0 = (long long)src;
    const long long i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    float l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    return (1.0f - l1) * row[i0] + l1 * row[i1];
}

__global__ void wnv_stretch_fir_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                       float* __restrict__ out, int B, int cin, long long Tin, int scale,
                                       int transpose_out, long long indent, int fk, int act, float act_p, int mode) {
    const long long Tfull = Tin * scale;
    const long long Tout = Tfull - 2 * indent;
    const long long total = (long long)B * cin * Tout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long j;
        int ch, b;
        if (transpose_out) {           // consecutive threads -> consecutive channels of one output row
            ch = (int)(i % cin);
            j = (i / cin) % Tout;
            b = (int)(i / ((long long)cin * Tout));
        } else {
            j = i % Tout;
            ch = (int)((i / Tout) % cin);
            b = (int)(i / (Tout * (long long)cin));
        }
        const long long jj = j + indent;
        float acc = 0.f;
        for (int f = 0; f < fk; ++f) {
            const int r = ch + f - (fk - 1) / 2;
            if (r < 0 || r >= cin) continue;
            const float* row = in + ((size_t)b * cin + r) * Tin;
            const float* wf = w + (size_t)f * (2 * scale + 1);
            for (int mtap = 0; mtap <= 2 * scale; ++mtap) {
                const long long q = jj + mtap - scale;
                if (q >= 0 && q < Tfull) acc = fmaf(wf[mtap], stretched(row, q, scale, Tin, mode), acc);
            }
        }
        acc = up_act(acc, act, act_p);
        if (transpose_out) out[((size_t)b * Tout + j) * cin + ch] = acc;
        else out[((size_t)b * cin + ch) * Tout + j] = acc;
    }
}

// Last stage, time-major output.  out[b][j][ch] = sum_m w[m] rep[j + indent + m - s],  rep[q] = in[b][ch][q / s] (0 outside);
// with J = (j0 + indent) / s (indent and the tile origin j0 are multiples of s) the window row of tap m of local sample tl is
// simply (tl + m) / s.  The taps are accumulated in the order m = 0 .. 2s, like the per-element kernel (bit-identical results).
constexpr int UT = 256;
__global__ void __launch_bounds__(UT) wnv_stretch_fir_tm_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                float* __restrict__ out, int cin, long long Tin, int scale,
                                                                long long indent, long long Tout, int ni, int tiles, int act, float act_p) {
    extern __shared__ float win[];                         // [cin][ni + 2 (+1: odd stride)]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles;
    const long long j0 = (long long)(blockIdx.x % tiles) * ni * scale;
    const long long ibase = (j0 + indent) / scale - 1;
    const int np = ni + 2, ls = np | 1;
    const float* src = in + (size_t)b * cin * Tin;
    for (int i = tid; i < cin * np; i += UT) {
        const int ch = i / np, p = i - ch * np;
        const long long q = ibase + p;
        win[ch * ls + p] = (q >= 0 && q < Tin) ? src[(size_t)ch * Tin + q] : 0.f;
    }
    float wr[33];
#pragma unroll
    for (int m = 0; m < 33; ++m) wr[m] = m <= 2 * scale ? w[m] : 0.f;
    __syncthreads();
    const long long nout = min((long long)ni * scale, Tout - j0);
    float* dst = out + ((size_t)b * Tout + j0) * cin;
    for (long long i = tid; i < nout * cin; i += UT) {
        const int tl = (int)(i / cin), ch = (int)(i - (long long)tl * cin);
        const float* row = win + ch * ls;
        float acc = 0.f;
        if (scale == 4) {
#pragma unroll
            for (int m = 0; m <= 8; ++m) acc = fmaf(wr[m], row[(tl + m) >> 2], acc);
        } else {
            for (int m = 0; m <= 2 * scale; ++m) acc = fmaf(w[m], row[(tl + m) / scale], acc);
        }
        dst[i] = up_act(acc, act, act_p);
    }
}

__global__ void wnv_transpose_bct_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int cin,
                                         long long T) {
    const long long total = (long long)B * cin * T;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cin);
        const long long t = (i / cin) % T;
        const int b = (int)(i / ((long long)cin * T));
        out[i] = in[((size_t)b * cin + ch) * T + t];
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;   // 256 CUs x 16 blocks, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

hipError_t wnv_launch_conv_in(const float* c, const float* w, float* out, int B, int cin, int Tin, int ks,
                              hipStream_t s) {
    const long long total = (long long)B * cin * (Tin - ks + 1);
    hipLaunchKernelGGL(wnv_conv_in_kernel, dim3(grid_for(total)), dim3(256), 0, s, c, w, out, B, cin, Tin, ks);
    return hipGetLastError();
}

hipError_t wnv_launch_stretch_fir(const float* in, const float* w, float* out, int B, int cin, long long Tin,
                                  int scale, int transpose_out, long long indent, int fk, int act, float act_p, int mode, hipStream_t s) {
    const long long total = (long long)B * cin * (Tin * scale - 2 * indent);
    if (mode == 0 && transpose_out && fk == 1 && scale <= 16 && indent % scale == 0 && cin <= 2048) {
        // LDS-tiled last stage: ni input samples per tile such that the window fits 48 KiB
        int ni = 64;
        while (ni > 1 && (size_t)cin * ((ni + 2) | 1) * sizeof(float) > 48 * 1024) ni >>= 1;
        const long long Tout = Tin * scale - 2 * indent;
        const long long tiles = (Tout + (long long)ni * scale - 1) / ((long long)ni * scale);
        if ((long long)B * tiles <= 0x7fffffffll && (size_t)cin * ((ni + 2) | 1) * sizeof(float) <= 64 * 1024) {
            const size_t lds = (size_t)cin * ((ni + 2) | 1) * sizeof(float);
            hipLaunchKernelGGL(wnv_stretch_fir_tm_kernel, dim3((unsigned)(B * tiles)), dim3(UT), lds, s, in, w, out, cin, Tin, scale,
                               indent, Tout, ni, (int)tiles, act, act_p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(wnv_stretch_fir_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, w, out, B, cin, Tin,
                       scale, transpose_out, indent, fk, act, act_p, mode);
    return hipGetLastError();
}

hipError_t wnv_launch_transpose_bct(const float* in, float* out, int B, int cin, long long T, hipStream_t s) {
    const long long total = (long long)B * cin * T;
    hipLaunchKernelGGL(wnv_transpose_bct_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, B, cin, T);
    return hipGetLastError();
}
