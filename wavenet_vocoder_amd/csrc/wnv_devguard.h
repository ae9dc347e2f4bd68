// wnv_devguard.h -- scoped "make this the current HIP device, restore the caller's on exit".  Every entry point of the C ABI
// uses it: the calling thread's current device is also torch's current device, and an entry point must not change it.
#pragma once
#include <hip/hip_runtime.h>

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;                       // host-only handle (wnv_create with device = -1): nothing to guard
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
