// wnv_ring.h -- host interface of the pipelined ring kernel (wnv_ring.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/wnv.h"
#include "wnv_dev.h"
#include "wnv_store.h"

struct WnvRingState;

// Placement census of a device (one tiny launch): CU count, number of XCDs, and whether block b of a one-block-per-CU grid ran on
// the same XCD as block b % n_xcd for every b (the mapping the persistent kernels lay their workgroups out by).
wnv_status wnv_placement_census(int device, int* ncu, int* n_xcd, bool* map_ok, std::string& err);

// Can the ring kernel run this configuration with B utterances in flight?
bool wnv_ring_supported(const wnv_config& c, int B);
const char* wnv_ring_why_not(const wnv_config& c, int B);
// Is the ring kernel the automatic choice (kernel == 0)?  WNV_RING=0/1 in the environment overrides.
bool wnv_ring_default();
// Builds (once) the ring-specific weight images from the fused host tensors and runs the whole loop.
wnv_status wnv_ring_generate(WnvRingState** st, int device, const wnv_config& c, const TensorStore& store,
                             const WnvGenArgs& ga, hipStream_t s, std::string& err);
// Waits for a pending asynchronous launch (WnvGenArgs::async) and returns its status; WNV_OK when nothing is pending.
wnv_status wnv_ring_wait(WnvRingState* st, std::string& err);
void wnv_ring_destroy(WnvRingState* st);
