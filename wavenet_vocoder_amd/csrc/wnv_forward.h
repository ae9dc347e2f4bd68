// wnv_forward.h -- host interface of the teacher-forced batch evaluation kernels (wnv_forward.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "wnv_dev.h"

struct WnvForwardArgs {
    int B;
    long long T;
    const float* x;          // (B, cin1, T)
    const float* c_up;       // (B, T, cin) time-major, or null
    const float* zbias;      // (B or 1, L, Gp): conv bias (+ Wg g)
    long long zbias_bstride;
    float* scratch;          // wnv_forward_scratch_floats() floats
    float* out;              // (B, O, T)
    int softmax;
};

// null when the MFMA path covers the configuration, else the reason
const char* wnv_forward_why_not(const WnvModelDev& m);
size_t wnv_forward_scratch_floats(const WnvModelDev& m, int B, long long T);
hipError_t wnv_launch_forward(const WnvModelDev& m, const WnvLayerDev* layers_host, const float* d_W, const WnvForwardArgs& a,
                              hipStream_t s);
