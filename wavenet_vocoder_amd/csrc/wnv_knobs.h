// wnv_knobs.h -- measurement / diagnostic knobs.
//
// The PRODUCT library (libwnv_hip.so) reads NO environment variable: layout, hand-off flavour, kernel instantiation and the
// per-device turn lock are decided by the code alone, per call, whatever the caller's environment says.
// The knobs the experiment scripts, the trace builds and a handful of variant tests use (WNV_RING_SPLIT, WNV_RING_L0, WNV_RING_FAST,
// WNV_RING_TAP, WNV_RING_MODE, WNV_RING, WNV_RING_CENSUS, WNV_NO_TURN, WNV_*_TRACE, WNV_RING_DEBUG_DUMP) exist only in a KNOB BUILD
// (-DWNV_KNOBS: wavenet_vocoder_amd/libwnv_test.so, selected with WNV_LIB=<path>; python -m wavenet_vocoder_amd.build builds both).
#pragma once
#include <cstdlib>

#ifdef WNV_KNOBS
static inline const char* wnv_knob(const char* name) { return std::getenv(name); }
#else
static inline const char* wnv_knob(const char*) { return nullptr; }
#endif
