// wnv_hostutil.h -- helpers shared by the host translation units (errors, device guard, scratch, packer).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/wnv.h"
#include "wnv_store.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
extern thread_local std::string wnv_g_err;

static inline wnv_status fail(wnv_status st, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    wnv_g_err = buf;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail(WNV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#include "wnv_devguard.h"

static inline int pad4(int n) { return (n + 3) & ~3; }

struct Expect { std::string name; std::vector<int64_t> shape; };

static inline bool shape_eq(const std::vector<int64_t>& a, const std::vector<int64_t>& b) { return a == b; }

// ------------------------------------------------------------------------------------------------
// the engine
// ------------------------------------------------------------------------------------------------
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};


// K-major packer ----------------------------------------------------------------------------------
struct Blob {
    std::vector<float> v;
    long long alloc(size_t n) {               // 16-byte aligned segments
        const size_t off = (v.size() + 3) & ~(size_t)3;
        v.resize(off + n, 0.f);
        return (long long)off;
    }
};

// W (Cout, Cin, kw) -> rows [k*Cin + i][Np] at blob offset `off`, column offset `col0`
static inline void put_kmajor(Blob& b, long long off, int Np, int col0, const HostTensor& w, int row0) {
    const int64_t co = w.shape[0], ci = w.shape[1], kw = w.shape.size() > 2 ? w.shape[2] : 1;
    for (int64_t o = 0; o < co; ++o)
        for (int64_t i = 0; i < ci; ++i)
            for (int64_t k = 0; k < kw; ++k)
                b.v[off + (size_t)(row0 + k * ci + i) * Np + col0 + o] = w.data[(o * ci + i) * kw + k];
}

