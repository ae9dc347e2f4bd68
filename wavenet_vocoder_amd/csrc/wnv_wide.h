// wnv_wide.h -- host interface of the group-ring kernel for wide models (wnv_wide.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/wnv.h"
#include "wnv_dev.h"
#include "wnv_ring.h"
#include "wnv_store.h"

struct WnvWideState;

// Can the group-ring kernel run this configuration with B utterances in flight?
bool wnv_wide_supported(const wnv_config& c, int B);
const char* wnv_wide_why_not(const wnv_config& c, int B);
// Builds (once) the per-slice weight images from the fused host tensors and runs the whole loop (synchronous).
wnv_status wnv_wide_generate(WnvWideState** st, int device, const wnv_config& c, const TensorStore& store,
                             const WnvGenArgs& ga, hipStream_t s, std::string& err);
void wnv_wide_destroy(WnvWideState* st);
