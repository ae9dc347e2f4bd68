// wnv_internal.h -- launcher prototypes between the host API (wnv_host.cpp) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include "wnv_dev.h"

#define WNV_GENERIC_WAVES 16          // 1024-thread workgroup: one workgroup owns one utterance
#define WNV_GENERIC_THREADS (WNV_GENERIC_WAVES * 64)

// ---- generic single-workgroup kernels (wnv_generic.hip) ---------------------------------------
size_t wnv_generic_lds_bytes(const WnvModelDev& m);
hipError_t wnv_launch_generate_generic(const WnvModelDev& m, const WnvLayerDev* d_layers, const float* d_W,
                                       const WnvGenArgs& a, hipStream_t s);
// zbias[b][l][n] = b_in[l][n] + sum_j Wg[l][j][n] * gvec[b][j]; gvec from g (B,gin) or embed[ids[b]]
hipError_t wnv_launch_zbias(const WnvModelDev& m, const WnvLayerDev* d_layers, const float* d_W,
                            const float* g, const long long* ids, const float* embed, int B,
                            float* zbias, hipStream_t s);

struct WnvGluStepArgs {
    int B, t;                 // t = steps since reset
    const float *x, *c, *g;   // (B,R) (B,cin) (B,gin)
    float *x_out, *s_out;     // (B,R) (B,K)
    float* ring;              // (B, ring_rows*R)
};
hipError_t wnv_launch_glu_step(const WnvModelDev& m, const WnvLayerDev* d_layer, const float* d_W,
                               const WnvGluStepArgs& a, hipStream_t s);

struct WnvQconvDev {
    int cin, cout, coutp, kw, dilation, ring_rows;
    long long w, b;           // [kw*cin][coutp], [coutp]
};
hipError_t wnv_launch_qconv_step(const WnvQconvDev& q, const float* d_W, const float* x, float* y,
                                 float* ring, int B, int t, hipStream_t s);

// ---- local-conditioning upsampler (wnv_upsample.hip) ------------------------------------------
// conv_in: (B, cin, Tin) -> (B, cin, Tin - ks + 1), weight (cin, cin, ks), valid, no bias
hipError_t wnv_launch_conv_in(const float* c, const float* w, float* out, int B, int cin, int Tin, int ks,
                              hipStream_t s);
// one [nearest-stretch x s, FIR 2s+1, zero pad s] stage on (rows, Tin) -> (rows, Tin*s); when
// transpose_out != 0 writes (B, Tout_trim, cin) time-major, dropping `indent` samples at both ends.
// fk: taps along the channel (mel-bin) axis (zero padded); act / act_p: the stage's activation (wnv_upsample_act) and its parameter;
// mode: 0 nearest stretch, 1 bilinear stretch (F.interpolate, align_corners = False)
hipError_t wnv_launch_stretch_fir(const float* in, const float* w, float* out, int B, int cin, long long Tin,
                                  int scale, int transpose_out, long long indent, int fk, int act, float act_p, int mode, hipStream_t s);
// plain (B, cin, T) -> (B, T, cin) transpose for the no-upsampling case
hipError_t wnv_launch_transpose_bct(const float* in, float* out, int B, int cin, long long T, hipStream_t s);
