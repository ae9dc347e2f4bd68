// wnv_upsample.hip -- one-shot local-conditioning upsampler (the prologue of the hot path).
//
// Reference: upsample.ConvInUpsampleNetwork.forward (upsample.py:83-85) = conv_in (valid Conv1d, k = 2*cin_pad+1,
// no bias) then UpsampleNetwork.forward (upsample.py:51-66) = per scale s: Stretch2d nearest x s along time
// (upsample.py:19-21) followed by Conv2d(1, 1, (1, 2s+1), padding=(0, s), bias=False) -- ONE filter shared by all
// mel bins -- and finally the (B, C, T) -> (B, T, C) transpose of wavenet.py:277-278.
//
// These are HBM-bound streaming kernels (the output, 4*cin bytes per audio sample per utterance, dominates).
// The last stage writes the time-major layout directly so the sample loop reads one contiguous cin-float row per
// step.  Round-1 form: one launch per stage, one output element per thread; the stages run once per batch and
// are < 1 % of a synthesis call, so fusing them is scheduled behind the sample loop work.
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

// out[row][j] = sum_{m=0..2s} w[m] * rep[j + m - s],  rep[q] = in[row][q / s] for 0 <= q < Tin*s else 0
__global__ void wnv_stretch_fir_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                       float* __restrict__ out, int B, int cin, long long Tin, int scale,
                                       int transpose_out, long long indent) {
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
        const float* row = in + ((size_t)b * cin + ch) * Tin;
        const long long jj = j + indent;
        float acc = 0.f;
        for (int mtap = 0; mtap <= 2 * scale; ++mtap) {
            const long long q = jj + mtap - scale;
            if (q >= 0 && q < Tfull) acc = fmaf(w[mtap], row[q / scale], acc);
        }
        if (transpose_out) out[((size_t)b * Tout + j) * cin + ch] = acc;
        else out[((size_t)b * cin + ch) * Tout + j] = acc;
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
                                  int scale, int transpose_out, long long indent, hipStream_t s) {
    const long long total = (long long)B * cin * (Tin * scale - 2 * indent);
    hipLaunchKernelGGL(wnv_stretch_fir_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, w, out, B, cin, Tin,
                       scale, transpose_out, indent);
    return hipGetLastError();
}

hipError_t wnv_launch_transpose_bct(const float* in, float* out, int B, int cin, long long T, hipStream_t s) {
    const long long total = (long long)B * cin * T;
    hipLaunchKernelGGL(wnv_transpose_bct_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, B, cin, T);
    return hipGetLastError();
}
