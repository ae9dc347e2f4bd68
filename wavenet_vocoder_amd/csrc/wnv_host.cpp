// wnv_host.cpp -- the C ABI of include/wnv.h: handle life cycle, state_dict ingestion (weight-norm fold,
// conv.py:51-62 linearisation, K-major re-layout), scratch management and kernel dispatch.
// Compiled by hipcc together with the kernel translation units into libwnv_hip.so; no torch dependency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wnv.h"
#ifdef WNV_TEST_HOOKS
#include "../../include/wnv_test.h"
#endif
#include "wnv_internal.h"
#include "wnv_store.h"
#include "wnv_ring.h"
#include "wnv_wide.h"
#include "wnv_forward.h"

#include "wnv_hostutil.h"
#include "wnv_knobs.h"

struct wnv_engine {
    wnv_config cfg{};
    int device = 0;
    TensorStore store;
    bool packed = false;
    // generic-kernel pack
    WnvModelDev m{};
    std::vector<WnvLayerDev> layers;
    float* d_W = nullptr;
    WnvLayerDev* d_layers = nullptr;
    size_t w_floats = 0;
    // upsampler weights (device): conv_in then the per-stage FIRs
    float* d_up = nullptr;
    std::vector<long long> up_off;     // offsets: [0] conv_in (or -1), [1+i] stage i
    long long embed_off = -1;          // inside d_W
    int64_t core_weights = 0;          // unpadded parameter count of the sample-loop network (with biases)
    int64_t core_macs = 0;
    Scratch ring, zbias, upA, upB, fwd;
    WnvRingState* ring_state = nullptr;   // pipelined kernel (wnv_ring.hip), built lazily
    WnvWideState* wide_state = nullptr;   // group-ring kernel for wide models (wnv_wide.hip), built lazily
    bool wide_disabled = false;           // auto mode: the persistent kernel cannot run on this device (WNV_ERR_UNSUPPORTED: permanent)
    bool ring_disabled = false;
    // auto mode after a WNV_ERR_TIMEOUT (the workgroups were not co-resident: CUs masked, or taken by somebody else for a moment):
    // the next `cooldown` calls are served by the generic kernel, then the persistent kernel is tried again; every further time-out
    // doubles the pause (2, 4, ... 32 calls), a launch that completes clears it, wnv_reset() makes the next call try at once
    int persist_cooldown = 0, persist_backoff = 0;
    int inject_timeouts = 0;              // wnv_debug_inject_timeouts: auto-mode ring launches still to be reported as timed out (tests)
    int last_kernel = 0;                  // 1 generic, 2 ring: what served the last wnv_generate
};

static std::vector<int> dilations_of(const wnv_config& c) {
    std::vector<int> d(c.layers);
    const int per = c.layers / c.stacks;
    for (int i = 0; i < c.layers; ++i) d[i] = 1 << (i % per);      // wavenet.py:125-126
    return d;
}

static std::vector<Expect> expected_tensors(const wnv_config& c) {
    std::vector<Expect> e;
    const int64_t R = c.residual_channels, G = c.gate_channels, K = c.skip_out_channels, O = c.out_channels;
    const int64_t kw = c.kernel_size, cin1 = c.scalar_input ? 1 : O;
    e.push_back({"first_conv.weight", {R, cin1, 1}});
    e.push_back({"first_conv.bias", {R}});
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = "conv_layers." + std::to_string(l) + ".";
        e.push_back({p + "conv.weight", {G, R, kw}});
        e.push_back({p + "conv.bias", {G}});
        if (c.cin_channels > 0) e.push_back({p + "conv1x1c.weight", {G, c.cin_channels, 1}});
        if (c.gin_channels > 0) e.push_back({p + "conv1x1g.weight", {G, c.gin_channels, 1}});
        e.push_back({p + "conv1x1_out.weight", {R, G / 2, 1}});
        e.push_back({p + "conv1x1_out.bias", {R}});
        e.push_back({p + "conv1x1_skip.weight", {K, G / 2, 1}});
        e.push_back({p + "conv1x1_skip.bias", {K}});
    }
    e.push_back({"last_conv_layers.1.weight", {K, K, 1}});
    e.push_back({"last_conv_layers.1.bias", {K}});
    e.push_back({"last_conv_layers.3.weight", {O, K, 1}});
    e.push_back({"last_conv_layers.3.bias", {O}});
    if (c.gin_channels > 0 && c.use_speaker_embedding)
        e.push_back({"embed_speakers.weight", {c.n_speakers, c.gin_channels}});
    if (c.upsample_kind != WNV_UPSAMPLE_NONE && c.cin_channels > 0) {
        std::string pre = "upsample_net.up_layers.";
        if (c.upsample_kind == WNV_UPSAMPLE_CONVIN) {
            e.push_back({"upsample_net.conv_in.weight", {c.cin_channels, c.cin_channels, 2 * c.cin_pad + 1}});
            pre = "upsample_net.upsample.up_layers.";
        }
        // up_layers: [Stretch2d, Conv2d(, activation)] per scale (upsample.py:38-49): the conv sits at index 2 i + 1, or 3 i + 1
        const int stride = c.upsample_activation != WNV_UPACT_NONE ? 3 : 2;
        for (int i = 0; i < c.n_upsample_scales; ++i)
            e.push_back({pre + std::to_string(stride * i + 1) + ".weight",
                         {1, 1, c.freq_axis_kernel_size, 2 * c.upsample_scales[i] + 1}});
    }
    return e;
}

static wnv_status validate_config(const wnv_config* c) {
    if (!c) return fail(WNV_ERR_INVALID_ARG, "config is NULL");
    if (c->abi_version != WNV_ABI_VERSION) return fail(WNV_ERR_INVALID_ARG, "abi_version %d != %d", c->abi_version, WNV_ABI_VERSION);
    if (c->layers <= 0 || c->stacks <= 0 || c->layers % c->stacks != 0)
        return fail(WNV_ERR_INVALID_ARG, "layers (%d) must be a positive multiple of stacks (%d)", c->layers, c->stacks);
    if (c->layers > WNV_MAX_LAYERS) return fail(WNV_ERR_UNSUPPORTED, "layers %d > %d", c->layers, WNV_MAX_LAYERS);
    if (c->layers / c->stacks > 24) return fail(WNV_ERR_UNSUPPORTED, "dilation 2^%d is too large", c->layers / c->stacks - 1);
    if (c->residual_channels <= 0 || c->gate_channels <= 0 || c->gate_channels % 2 || c->skip_out_channels <= 0 || c->out_channels <= 0)
        return fail(WNV_ERR_INVALID_ARG, "channel counts must be positive and gate_channels even");
    if (c->kernel_size < 1 || c->kernel_size > 16) return fail(WNV_ERR_UNSUPPORTED, "kernel_size %d outside [1,16]", c->kernel_size);
    if (c->output_distribution < 0 || c->output_distribution > 2) return fail(WNV_ERR_INVALID_ARG, "unknown output_distribution %d", c->output_distribution);
    if (c->scalar_input && c->output_distribution == WNV_DIST_CATEGORICAL)
        return fail(WNV_ERR_INVALID_ARG, "scalar_input requires Logistic or Normal output");   // wavenet.py:330 assert False
    if (!c->scalar_input && c->output_distribution != WNV_DIST_CATEGORICAL)
        ;  // the reference ignores output_distribution for one-hot input (wavenet.py:331-335)
    if (c->scalar_input) {
        const int O = c->out_channels;
        if (c->output_distribution == WNV_DIST_LOGISTIC && O % 3) return fail(WNV_ERR_INVALID_ARG, "Logistic needs out_channels %% 3 == 0 (mixture.py:130)");
        if (c->output_distribution == WNV_DIST_NORMAL && O != 2 && O % 3) return fail(WNV_ERR_INVALID_ARG, "Normal needs out_channels == 2 or %% 3 == 0 (mixture.py:234)");
    }
    if (c->residual_channels + c->skip_out_channels > WNV_GENERIC_THREADS || c->gate_channels / 2 > WNV_GENERIC_THREADS ||
        c->out_channels > WNV_GENERIC_THREADS)
        return fail(WNV_ERR_UNSUPPORTED, "residual+skip, gate/2 and out channels must each be <= %d", WNV_GENERIC_THREADS);
    if (c->upsample_kind < 0 || c->upsample_kind > 2) return fail(WNV_ERR_INVALID_ARG, "unknown upsample_kind");
    if (c->upsample_kind != WNV_UPSAMPLE_NONE) {
        if (c->n_upsample_scales < 0 || c->n_upsample_scales > WNV_MAX_UPSAMPLE_STAGES) return fail(WNV_ERR_INVALID_ARG, "n_upsample_scales");
        if (c->freq_axis_kernel_size < 1 || c->freq_axis_kernel_size > 15 || c->freq_axis_kernel_size % 2 == 0)
            return fail(WNV_ERR_UNSUPPORTED, "freq_axis_kernel_size %d: an odd size <= 15 is implemented", c->freq_axis_kernel_size);
        if (c->upsample_activation < 0 || c->upsample_activation > WNV_UPACT_ELU) return fail(WNV_ERR_INVALID_ARG, "unknown upsample_activation %d", c->upsample_activation);
        if (c->upsample_mode < 0 || c->upsample_mode > 2) return fail(WNV_ERR_UNSUPPORTED, "upsample_mode %d: 0 (nearest; also what area / nearest-exact do for integer factors), 1 (bilinear) and 2 (bicubic) are implemented", c->upsample_mode);
        for (int i = 0; i < c->n_upsample_scales; ++i)
            if (c->upsample_scales[i] < 1) return fail(WNV_ERR_INVALID_ARG, "upsample scale < 1");
    }
    if (c->gin_channels > 0 && c->use_speaker_embedding && c->n_speakers <= 0)
        return fail(WNV_ERR_INVALID_ARG, "use_speaker_embedding needs n_speakers (wavenet.py:144)");
    return WNV_OK;
}

extern "C" int32_t wnv_abi_version(void) { return WNV_ABI_VERSION; }

// ---- host helpers for a streamed replay tape (include/wnv.h) ---------------------------------------------------------------------
extern "C" wnv_status wnv_pinned_alloc(size_t bytes, void** host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr || bytes == 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_pinned_alloc");
    void* hp = nullptr;
    // portable: pinned for and mapped into EVERY device's address space, whichever device is current on the calling thread -- a
    // handle on another GPU than the thread's current one reads the same address (one unified virtual address space on ROCm)
    HIP_TRY(hipHostMalloc(&hp, bytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
    void* dp = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dp, hp, 0);
    if (e != hipSuccess) { (void)hipHostFree(hp); return fail(WNV_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    *host_ptr = hp; *device_ptr = dp;
    return WNV_OK;
}
extern "C" wnv_status wnv_pinned_free(void* host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return WNV_OK;
}
// The CPUs that share the last-level cache with `cpu` and that this process may run on (sysfs; empty when it cannot be read).
// The transform's workers are kept there: its input was written by the calling thread a moment ago and is REWRITTEN by it for the
// next chunk -- on the two-socket, 16-CCD host of the GPU box, workers scheduled anywhere left the staging buffer's lines in remote
// caches and torch's uniform_ then ran at 9.6 ns per draw instead of 2.3 (61 ns per cache line of ownership transfers).
static cpu_set_t llc_neighbours(int cpu, int* count) {
    cpu_set_t set, allowed;
    CPU_ZERO(&set);
    *count = 0;
    if (cpu < 0 || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return set;
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE* f = fopen(path, "r");
    if (!f) return set;
    char buf[512];
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return set;
    for (char* q = buf; *q;) {                                                  // "0-7,128-135"
        char* end;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        q = end;
        if (*q == '-') { b = strtol(q + 1, &end, 10); q = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed) && !CPU_ISSET((int)c, &set)) { CPU_SET((int)c, &set); ++*count; }
        while (*q == ',' || *q == ' ' || *q == '\n') ++q;
    }
    return set;
}
extern "C" wnv_status wnv_exponential_from_uniform(const double* u, float* out, int64_t n, int32_t threads) {
    if ((!u || !out) && n > 0) return fail(WNV_ERR_INVALID_ARG, "NULL buffer");
    if (n < 0) return fail(WNV_ERR_INVALID_ARG, "n < 0");
    // ATen, CPU: static_cast<float>(-1.0 / lambda * log1p(-u)) with lambda = 1.0, in double (TransformationHelper.h, exponential<double>)
    auto work = [u, out](int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) out[i] = static_cast<float>(-1.0 / 1.0 * std::log1p(-u[i])); };
    int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(threads, 64), n / 4096));
    if (nt <= 1) { work(0, n); return WNV_OK; }
    int near = 0;
    const cpu_set_t llc = llc_neighbours(sched_getcpu(), &near);
    if (near >= 2) nt = std::min(nt, near);
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int k = 1; k < nt; ++k) {
        pool.emplace_back(work, std::min<int64_t>(n, k * per), std::min<int64_t>(n, (k + 1) * per));
        if (near >= 2) (void)pthread_setaffinity_np(pool.back().native_handle(), sizeof llc, &llc);
    }
    work(0, std::min<int64_t>(n, per));                                          // the caller takes the first share
    for (auto& th : pool) th.join();
    return WNV_OK;
}
// torch's CPU generator, natively.  at::mt19937 (ATen/core/MT19937RNGEngine.h) is the 32-bit Mersenne Twister of Matsumoto & Nishimura;
// CPUGeneratorImpl::random64() joins two consecutive outputs, the first one high, and uniform_real_distribution<double> maps
// x -> (x & (2^53 - 1)) * 2^-53 * (to - from) + from (ATen/core/DistributionsHelper.h).  `state` is the blob torch.Generator.get_state()
// returns for a CPU generator: seed u64 @0, left i32 @8, seeded i32 @12, next u64 @16, state[624] as u64 @24 (one 32-bit word each).
// The engine's call is  if (--left == 0) next_state();  y = state[next++];  temper(y)  -- reproduced call for call, in bulk: the blob
// is advanced as n draws of uniform_(0, 1) on a float64 tensor would advance it.  (Why: that uniform_ call walks the generator element
// by element at 2.3 ns per draw alone and 5-10 ns inside incremental_forward on the GPU box's host -- the replay tape of a one-hot model
// needs B x 256 draws per step, and since the categorical head got faster the host was what bounded the public path.)
namespace {
constexpr int MT_N = 624, MT_M = 397;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define WNV_SIMD_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define WNV_SIMD_CLONES
#endif
inline uint32_t mt_twist(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu); }
WNV_SIMD_CLONES void mt_next_state(uint32_t* __restrict st) {                  // at::mt19937::next_state(): three runs, each reads only words not yet rewritten
    for (int i = 0; i < MT_N - MT_M; ++i) st[i] = st[i + MT_M] ^ mt_twist(st[i], st[i + 1]);
    for (int i = MT_N - MT_M; i < MT_N - 1; ++i) st[i] = st[i + MT_M - MT_N] ^ mt_twist(st[i], st[i + 1]);
    st[MT_N - 1] = st[MT_M - 1] ^ mt_twist(st[MT_N - 1], st[0]);
}
WNV_SIMD_CLONES void mt_temper_run(const uint32_t* __restrict st, uint32_t* __restrict out, int k) {
    for (int i = 0; i < k; ++i) {
        uint32_t y = st[i];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        out[i] = y;
    }
}
WNV_SIMD_CLONES void mt_join53(const uint32_t* __restrict w, double* __restrict out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t x = ((uint64_t)w[2 * i] << 32) | w[2 * i + 1];          // random64(): the first output is the high half
        out[i] = (double)(x & ((1ULL << 53) - 1)) * 0x1.0p-53;                 // uniform_real_distribution<double>, from 0 to 1
    }
}
}  // namespace
extern "C" wnv_status wnv_mt19937_uniform53(void* state, int64_t state_bytes, double* out, int64_t n) {
    if (!state || (!out && n > 0)) return fail(WNV_ERR_INVALID_ARG, "NULL buffer");
    if (n < 0 || state_bytes < 24 + 8 * MT_N) return fail(WNV_ERR_INVALID_ARG, "not a CPU generator state");
    unsigned char* blob = static_cast<unsigned char*>(state);
    int32_t left, seeded;
    uint64_t next, w64[MT_N];
    std::memcpy(&left, blob + 8, 4); std::memcpy(&seeded, blob + 12, 4); std::memcpy(&next, blob + 16, 8); std::memcpy(w64, blob + 24, sizeof w64);
    if (!seeded || left < 1 || left > MT_N || next + (uint64_t)(left - 1) > (uint64_t)MT_N) return fail(WNV_ERR_INVALID_ARG, "not a CPU generator state");
    uint32_t st[MT_N];
    for (int i = 0; i < MT_N; ++i) st[i] = (uint32_t)w64[i];
    // The engine's call is  if (--left == 0) next_state();  y = temper(state[next++]):  left - 1 words can be read before the next twist; the
    // call that finds left == 1 twists (left = N, next = 0) and reads word 0.  2 n words, run by run, then joined pairwise.
    thread_local std::vector<uint32_t> words;
    constexpr int64_t PIECE = 1 << 16;                                         // values per piece: the word buffer stays in the L2
    if ((int64_t)words.size() < 2 * PIECE) words.resize(2 * PIECE);
    for (int64_t done = 0; done < n; done += PIECE) {
        const int64_t m = std::min<int64_t>(PIECE, n - done), need = 2 * m;
        int64_t w = 0;
        const int64_t head = std::min<int64_t>(need, left - 1);
        mt_temper_run(st + next, words.data(), (int)head);
        next += (uint64_t)head; left -= (int32_t)head; w = head;
        while (w < need) {                                                      // left == 1 here
            mt_next_state(st);
            const int k = (int)std::min<int64_t>(MT_N, need - w);
            mt_temper_run(st, words.data() + w, k);
            w += k; left = MT_N + 1 - k; next = (uint64_t)k;
        }
        mt_join53(words.data(), out + done, m);
    }
    for (int i = 0; i < MT_N; ++i) w64[i] = st[i];
    std::memcpy(blob + 8, &left, 4); std::memcpy(blob + 16, &next, 8); std::memcpy(blob + 24, w64, sizeof w64);
    return WNV_OK;
}
thread_local std::string wnv_g_err;
extern "C" const char* wnv_last_error(void) { return wnv_g_err.c_str(); }

extern "C" int64_t wnv_receptive_field(int32_t layers, int32_t stacks, int32_t kernel_size) {
    if (layers <= 0 || stacks <= 0 || layers % stacks) return -1;
    const int per = layers / stacks;
    int64_t sum = 0;
    for (int i = 0; i < layers; ++i) sum += (int64_t)1 << (i % per);
    return (int64_t)(kernel_size - 1) * sum + 1;                               // wavenet.py:42-60
}

extern "C" int32_t wnv_noise_width(const wnv_config* c) {
    if (!c) return -1;
    if (!c->scalar_input) return c->out_channels;
    if (c->output_distribution == WNV_DIST_LOGISTIC) return c->out_channels / 3 + 1;
    if (c->output_distribution == WNV_DIST_NORMAL) return (c->out_channels == 2 || c->out_channels == 3) ? 1 : c->out_channels / 3 + 1;
    return -1;
}

extern "C" int64_t wnv_upsampled_length(const wnv_config* c, int64_t Tc_in) {
    if (!c || Tc_in < 0) return -1;
    if (c->upsample_kind == WNV_UPSAMPLE_NONE) return Tc_in;
    int64_t total = 1;
    for (int i = 0; i < c->n_upsample_scales; ++i) total *= c->upsample_scales[i];
    if (c->upsample_kind == WNV_UPSAMPLE_CONVIN) {
        const int64_t frames = Tc_in - 2 * c->cin_pad;                         // valid conv, k = 2*cin_pad+1
        return frames <= 0 ? -1 : frames * total;
    }
    const int64_t t = Tc_in * total - 2 * (int64_t)c->cin_pad * total;        // upsample.py:36,64-65
    return t <= 0 ? -1 : t;
}

extern "C" wnv_status wnv_create(const wnv_config* cfg, int32_t device, wnv_handle* out) {
    if (!out) return fail(WNV_ERR_INVALID_ARG, "out handle is NULL");
    *out = nullptr;
    wnv_status st = validate_config(cfg);
    if (st != WNV_OK) return st;
    if (device != -1) {                            // -1: host-only handle (checkpoint validation, packing, introspection; no launches)
        int ndev = 0;
        HIP_TRY(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) return fail(WNV_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    }
    wnv_engine* h = new wnv_engine();
    h->cfg = *cfg;
    h->device = device;
    *out = h;
    return WNV_OK;
}

static void free_dev(wnv_engine* h) {
    // (first: wnv_ring_destroy waits for a pending asynchronous launch, which still reads the handle's buffers)
    if (h->ring_state) { wnv_ring_destroy(h->ring_state); h->ring_state = nullptr; }
    if (h->wide_state) { wnv_wide_destroy(h->wide_state); h->wide_state = nullptr; }
    if (h->d_W) (void)hipFree(h->d_W);
    if (h->d_layers) (void)hipFree(h->d_layers);
    if (h->d_up) (void)hipFree(h->d_up);
    h->d_W = nullptr; h->d_layers = nullptr; h->d_up = nullptr;
    h->packed = false;
}

extern "C" wnv_status wnv_destroy(wnv_handle h) {
    if (!h) return WNV_OK;
    DeviceGuard g(h->device);
    free_dev(h);
    h->ring.release(); h->zbias.release(); h->upA.release(); h->upB.release(); h->fwd.release();
    delete h;
    return WNV_OK;
}

extern "C" wnv_status wnv_wait(wnv_handle h) {
    if (!h) return fail(WNV_ERR_INVALID_ARG, "handle is NULL");
    if (h->device < 0 || !h->ring_state) return WNV_OK;
    DeviceGuard g(h->device);
    std::string err;
    const wnv_status st = wnv_ring_wait(h->ring_state, err);
    return st == WNV_OK ? WNV_OK : fail(st, "%s", err.c_str());
}

extern "C" int32_t wnv_last_kernel(wnv_handle h) { return h ? h->last_kernel : 0; }
#ifdef WNV_TEST_HOOKS      // include/wnv_test.h: the test library only (libwnv_test.so)
extern "C" wnv_status wnv_debug_inject_timeouts(wnv_handle h, int32_t n) {
    if (!h || n < 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_debug_inject_timeouts");
    h->inject_timeouts = n;
    return WNV_OK;
}
#endif

// Which configurations a sample-loop kernel covers, decided from the configuration alone (pure host code, no device): "supported" or
// the reason -- the same strings wnv_generate reports with WNV_ERR_UNSUPPORTED.  kernel: 1 generic (covers everything), 2 pipelined
// ring, 3 group ring (wide models).
extern "C" const char* wnv_kernel_coverage(const wnv_config* cfg, int32_t kernel, int32_t B) {
    if (!cfg) return "cfg is NULL";
    if (cfg->abi_version != WNV_ABI_VERSION) return "abi_version mismatch";
    switch (kernel) {
        case 1: return "supported";
        case 2: return wnv_ring_why_not(*cfg, B);
        case 3: return wnv_wide_why_not(*cfg, B);
        default: return "unknown kernel selector";
    }
}

extern "C" wnv_status wnv_reset(wnv_handle h) {
    if (!h) return fail(WNV_ERR_INVALID_ARG, "handle is NULL");
    if (h->device < 0) return WNV_OK;
    DeviceGuard g(h->device);
    const wnv_status deferred = wnv_wait(h);                // a pending WNV_GEN_ASYNC launch reports here
    HIP_TRY(hipDeviceSynchronize());
    h->ring.release(); h->zbias.release(); h->upA.release(); h->upB.release(); h->fwd.release();
    h->persist_cooldown = 0;                                // a persistent kernel that timed out is tried again by the next call
    return deferred;
}

static wnv_status pack(wnv_engine* h) {
    const wnv_config& c = h->cfg;
    const auto exp = expected_tensors(c);
    for (const auto& e : exp) {
        const HostTensor* t = h->store.get(e.name);
        if (!t) return fail(WNV_ERR_NOT_LOADED, "missing tensor '%s'", e.name.c_str());
        if (!shape_eq(t->shape, e.shape)) {
            std::string got, want;
            for (auto s : t->shape) got += std::to_string(s) + ",";
            for (auto s : e.shape) want += std::to_string(s) + ",";
            return fail(WNV_ERR_INVALID_ARG, "size mismatch for %s: got (%s) expected (%s)", e.name.c_str(), got.c_str(), want.c_str());
        }
    }
    DeviceGuard g(h->device);
    free_dev(h);
    const int R = c.residual_channels, G = c.gate_channels, K = c.skip_out_channels, O = c.out_channels;
    const int kw = c.kernel_size, cin = c.cin_channels > 0 ? c.cin_channels : 0, gin = c.gin_channels > 0 ? c.gin_channels : 0;
    const int cin1 = c.scalar_input ? 1 : O, L = c.layers, H = G / 2;
    WnvModelDev& m = h->m;
    memset(&m, 0, sizeof m);
    m.L = L; m.R = R; m.G = G; m.K = K; m.O = O; m.kw = kw; m.cin = cin; m.gin = gin; m.cin1 = cin1;
    m.scalar_input = c.scalar_input; m.dist = c.scalar_input ? c.output_distribution : WNV_DIST_CATEGORICAL;
    m.Rp = pad4(R); m.Gp = pad4(G); m.NOSp = pad4(R + K); m.Kp = pad4(K); m.Op = pad4(O);
    m.nr_mix = (c.scalar_input && O % 3 == 0) ? O / 3 : 0;
    m.skip_scale = (float)std::sqrt(1.0 / L);
    Blob b;
    auto T = [&](const std::string& n) -> const HostTensor& { return *h->store.get(n); };
    m.w_first = b.alloc((size_t)cin1 * m.Rp);
    put_kmajor(b, m.w_first, m.Rp, 0, T("first_conv.weight"), 0);
    m.b_first = b.alloc(m.Rp);
    std::copy(T("first_conv.bias").data.begin(), T("first_conv.bias").data.end(), b.v.begin() + m.b_first);
    const auto dil = dilations_of(c);
    h->layers.assign(L, WnvLayerDev{});
    long long ring_off = 0;
    int64_t weights = (int64_t)R * cin1 + R, macs = (int64_t)R * cin1;
    for (int l = 0; l < L; ++l) {
        const std::string p = "conv_layers." + std::to_string(l) + ".";
        WnvLayerDev& Ld = h->layers[l];
        Ld.dilation = dil[l];
        Ld.ring_rows = (kw - 1) * dil[l];
        Ld.ring_off = ring_off;
        ring_off += (long long)Ld.ring_rows * R;
        const int Kin = kw * R + cin;
        Ld.w_in = b.alloc((size_t)Kin * m.Gp);
        put_kmajor(b, Ld.w_in, m.Gp, 0, T(p + "conv.weight"), 0);
        if (cin > 0) put_kmajor(b, Ld.w_in, m.Gp, 0, T(p + "conv1x1c.weight"), kw * R);
        Ld.b_in = b.alloc(m.Gp);
        std::copy(T(p + "conv.bias").data.begin(), T(p + "conv.bias").data.end(), b.v.begin() + Ld.b_in);
        Ld.w_g = -1;
        if (gin > 0) {
            Ld.w_g = b.alloc((size_t)gin * m.Gp);
            put_kmajor(b, Ld.w_g, m.Gp, 0, T(p + "conv1x1g.weight"), 0);
        }
        Ld.w_os = b.alloc((size_t)H * m.NOSp);
        put_kmajor(b, Ld.w_os, m.NOSp, 0, T(p + "conv1x1_out.weight"), 0);
        put_kmajor(b, Ld.w_os, m.NOSp, R, T(p + "conv1x1_skip.weight"), 0);
        Ld.b_os = b.alloc(m.NOSp);
        std::copy(T(p + "conv1x1_out.bias").data.begin(), T(p + "conv1x1_out.bias").data.end(), b.v.begin() + Ld.b_os);
        std::copy(T(p + "conv1x1_skip.bias").data.begin(), T(p + "conv1x1_skip.bias").data.end(), b.v.begin() + Ld.b_os + R);
        weights += (int64_t)G * R * kw + G + (int64_t)G * cin + (int64_t)G * gin + (int64_t)R * H + R + (int64_t)K * H + K;
        macs += (int64_t)G * R * kw + (int64_t)G * cin + (int64_t)R * H + (int64_t)K * H;   // Wg.g is hoisted
    }
    m.ring_floats = ring_off;
    m.w_h1 = b.alloc((size_t)K * m.Kp);
    put_kmajor(b, m.w_h1, m.Kp, 0, T("last_conv_layers.1.weight"), 0);
    m.b_h1 = b.alloc(m.Kp);
    std::copy(T("last_conv_layers.1.bias").data.begin(), T("last_conv_layers.1.bias").data.end(), b.v.begin() + m.b_h1);
    m.w_h2 = b.alloc((size_t)K * m.Op);
    put_kmajor(b, m.w_h2, m.Op, 0, T("last_conv_layers.3.weight"), 0);
    m.b_h2 = b.alloc(m.Op);
    std::copy(T("last_conv_layers.3.bias").data.begin(), T("last_conv_layers.3.bias").data.end(), b.v.begin() + m.b_h2);
    weights += (int64_t)K * K + K + (int64_t)O * K + O;
    macs += (int64_t)K * K + (int64_t)O * K;
    h->embed_off = -1;
    m.n_embed = (gin > 0 && c.use_speaker_embedding) ? c.n_speakers : 0;
    if (gin > 0 && c.use_speaker_embedding) {
        const HostTensor& e = T("embed_speakers.weight");
        h->embed_off = b.alloc(e.data.size());
        std::copy(e.data.begin(), e.data.end(), b.v.begin() + h->embed_off);
    }
    h->core_weights = weights;
    h->core_macs = macs;
    // LDS carve for the generic kernel
    const int nz = wnv_noise_width(&c);
    m.lds_xin = pad4(kw * R + cin);
    m.lds_u = pad4(std::max(H, K));
    m.lds_o = pad4(std::max(O, 4));
    m.lds_vin = pad4(std::max(cin1, 4));
    m.lds_nz = pad4(std::max(nz, 4));
    m.lds_part_stride = std::max(std::max(m.Gp, m.NOSp), std::max(std::max(m.Rp, m.Kp), m.Op));
    m.lds_taps = L * (kw - 1) * R;
    m.taps_in_lds = 1;
    if (wnv_generic_lds_bytes(m) > 150 * 1024) { m.lds_taps = 0; m.taps_in_lds = 0; }
    m.lds_taps = pad4(m.lds_taps);
    if (wnv_generic_lds_bytes(m) > 160 * 1024)
        return fail(WNV_ERR_UNSUPPORTED, "configuration needs %zu bytes of LDS (> 160 KiB)", wnv_generic_lds_bytes(m));
    // upload
    h->w_floats = b.v.size();
    if (h->device < 0) { h->packed = true; return WNV_OK; }       // host-only handle: validated and packed, nothing to upload
    HIP_TRY(hipMalloc((void**)&h->d_W, std::max<size_t>(b.v.size(), 4) * sizeof(float)));
    HIP_TRY(hipMemcpy(h->d_W, b.v.data(), b.v.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&h->d_layers, L * sizeof(WnvLayerDev)));
    HIP_TRY(hipMemcpy(h->d_layers, h->layers.data(), L * sizeof(WnvLayerDev), hipMemcpyHostToDevice));
    // upsampler
    h->up_off.clear();
    if (c.upsample_kind != WNV_UPSAMPLE_NONE && cin > 0) {
        std::vector<float> u;
        std::string pre = "upsample_net.up_layers.";
        if (c.upsample_kind == WNV_UPSAMPLE_CONVIN) {
            const HostTensor& w = T("upsample_net.conv_in.weight");
            h->up_off.push_back(0);
            u.insert(u.end(), w.data.begin(), w.data.end());
            pre = "upsample_net.upsample.up_layers.";
        } else {
            h->up_off.push_back(-1);
        }
        for (int i = 0; i < c.n_upsample_scales; ++i) {
            const HostTensor& w = T(pre + std::to_string((c.upsample_activation != WNV_UPACT_NONE ? 3 : 2) * i + 1) + ".weight");
            h->up_off.push_back((long long)u.size());
            u.insert(u.end(), w.data.begin(), w.data.end());
        }
        HIP_TRY(hipMalloc((void**)&h->d_up, std::max<size_t>(u.size(), 4) * sizeof(float)));
        HIP_TRY(hipMemcpy(h->d_up, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->packed = true;
    return WNV_OK;
}

extern "C" wnv_status wnv_load_weights(wnv_handle h, const wnv_tensor* tensors, int32_t n) {
    if (!h || (!tensors && n > 0) || n < 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_load_weights");
    const auto exp = expected_tensors(h->cfg);
    for (int i = 0; i < n; ++i) {
        const wnv_tensor& t = tensors[i];
        if (!t.name || !t.data || t.ndim < 1 || t.ndim > 4) return fail(WNV_ERR_INVALID_ARG, "tensor %d is malformed", i);
        std::string name = t.name;
        std::string base = name;
        if (ends_with(name, "weight_g") || ends_with(name, "weight_v")) base = name.substr(0, name.size() - 2);
        bool known = false;
        for (const auto& e : exp) if (e.name == base) { known = true; break; }
        if (!known) return fail(WNV_ERR_INVALID_ARG, "unexpected key '%s' in state_dict", t.name);
        h->store.put(t);
    }
    return pack(h);
}

extern "C" int64_t wnv_bytes_per_step(wnv_handle h, int32_t B) {
    if (!h || !h->packed) return -1;
    const wnv_config& c = h->cfg;
    const int64_t cin = c.cin_channels > 0 ? c.cin_channels : 0;
    // every weight once per step per utterance group + per utterance the kw ring taps read and the
    // one row written per layer, the conditioning row and the emitted sample
    int64_t wbytes = h->core_weights;
    if (c.gin_channels > 0) wbytes -= (int64_t)c.layers * c.gate_channels * c.gin_channels;   // hoisted out of the loop
    return wbytes * 4 + (int64_t)B * ((int64_t)c.layers * (c.kernel_size + 1) * c.residual_channels * 4 + cin * 4 + 4);
}

extern "C" int64_t wnv_macs_per_sample(wnv_handle h) { return (h && h->packed) ? h->core_macs : -1; }

// ------------------------------------------------------------------------------------------------
// upsampling prologue
// ------------------------------------------------------------------------------------------------
extern "C" wnv_status wnv_upsample(wnv_handle h, const float* c_in, int32_t B, int64_t Tc_in, float* c_up,
                                   int64_t T_expected, void* stream) {
    if (!h || !c_in || !c_up || B <= 0 || Tc_in <= 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_upsample");
    if (h->device < 0) return fail(WNV_ERR_INVALID_ARG, "host-only handle (created with device = -1): there is no CPU path");
    if (!h->packed) return fail(WNV_ERR_NOT_LOADED, "weights are not loaded");
    const wnv_config& c = h->cfg;
    if (c.cin_channels <= 0) return fail(WNV_ERR_INVALID_ARG, "model has no local conditioning");
    const int64_t T = wnv_upsampled_length(&c, Tc_in);
    if (T <= 0) return fail(WNV_ERR_SHAPE, "conditioning of %lld frames is too short", (long long)Tc_in);
    if (T_expected >= 0 && T != T_expected)
        return fail(WNV_ERR_SHAPE, "upsampled conditioning length %lld != T %lld (wavenet.py:276)", (long long)T, (long long)T_expected);
    DeviceGuard g(h->device);
    hipStream_t s = (hipStream_t)stream;
    const int cin = c.cin_channels;
    if (c.upsample_kind == WNV_UPSAMPLE_NONE) {
        HIP_TRY(wnv_launch_transpose_bct(c_in, c_up, B, cin, Tc_in, s));
        return WNV_OK;
    }
    const float* cur = c_in;
    int64_t Tcur = Tc_in;
    float* bufs[2];
    // largest intermediate: the input of the last stage
    int64_t total = 1;
    for (int i = 0; i < c.n_upsample_scales; ++i) total *= c.upsample_scales[i];
    const size_t inter = (size_t)B * cin * (size_t)(Tc_in * total) * sizeof(float);
    HIP_TRY(h->upA.ensure(inter));
    HIP_TRY(h->upB.ensure(inter));
    bufs[0] = (float*)h->upA.p; bufs[1] = (float*)h->upB.p;
    int which = 0;
    if (c.upsample_kind == WNV_UPSAMPLE_CONVIN) {
        const int ks = 2 * c.cin_pad + 1;
        HIP_TRY(wnv_launch_conv_in(cur, h->d_up + h->up_off[0], bufs[which], B, cin, (int)Tcur, ks, s));
        cur = bufs[which]; which ^= 1;
        Tcur = Tcur - ks + 1;
    }
    if (c.n_upsample_scales == 0) {
        HIP_TRY(wnv_launch_transpose_bct(cur, c_up, B, cin, Tcur, s));
        return WNV_OK;
    }
    for (int i = 0; i < c.n_upsample_scales; ++i) {
        const bool last = i == c.n_upsample_scales - 1;
        const int sc = c.upsample_scales[i];
        const long long indent = (last && c.upsample_kind == WNV_UPSAMPLE_PLAIN) ? (long long)c.cin_pad * total : 0;
        float* dst = last ? c_up : bufs[which];
        HIP_TRY(wnv_launch_stretch_fir(cur, h->d_up + h->up_off[1 + i], dst, B, cin, Tcur, sc, last ? 1 : 0, indent, c.freq_axis_kernel_size,
                                       c.upsample_activation, c.upsample_activation_param, c.upsample_mode, s));
        cur = dst; which ^= 1;
        Tcur *= sc;
    }
    return WNV_OK;
}

// ------------------------------------------------------------------------------------------------
// teacher-forced batch evaluation
// ------------------------------------------------------------------------------------------------
extern "C" wnv_status wnv_forward(wnv_handle h, const wnv_forward_args* a) {
    if (!h || !a) return fail(WNV_ERR_INVALID_ARG, "NULL handle or args");
    if (h->device < 0) return fail(WNV_ERR_INVALID_ARG, "host-only handle (created with device = -1): there is no CPU path");
    if (!h->packed) return fail(WNV_ERR_NOT_LOADED, "weights are not loaded");
    const WnvModelDev& m = h->m;
    if (a->B <= 0 || a->T <= 0 || a->T > (1LL << 23)) return fail(WNV_ERR_INVALID_ARG, "B and T must be positive (T <= 2^23: buffer offsets of the tile kernels stay below 2^31 bytes)");
    if (!a->x || !a->out) return fail(WNV_ERR_INVALID_ARG, "x / out is NULL");
    if (m.cin > 0 && !a->c_up) return fail(WNV_ERR_INVALID_ARG, "model has local conditioning but c_up is NULL");
    if (m.cin == 0 && a->c_up) return fail(WNV_ERR_INVALID_ARG, "c_up given but the model has no local conditioning");
    if (m.gin > 0 && !a->g && !a->g_ids) return fail(WNV_ERR_INVALID_ARG, "model has global conditioning but neither g nor g_ids is given");
    if (m.gin == 0 && (a->g || a->g_ids)) return fail(WNV_ERR_INVALID_ARG, "g given but the model has no global conditioning");
    if (const char* why = wnv_forward_why_not(m)) return fail(WNV_ERR_UNSUPPORTED, "the MFMA forward kernel does not cover this configuration: %s", why);
    DeviceGuard g(h->device);
    hipStream_t s = (hipStream_t)a->stream;
    if (h->ring_state) {                                             // the bias table is shared with a pending asynchronous generation
        std::string err;
        const wnv_status pst = wnv_ring_wait(h->ring_state, err);
        if (pst != WNV_OK) return fail(pst, "deferred from the previous asynchronous call: %s", err.c_str());
    }
    const bool has_g = m.gin > 0;
    const int Bz = has_g ? a->B : 1;
    HIP_TRY(h->zbias.ensure((size_t)Bz * m.L * m.Gp * sizeof(float)));
    HIP_TRY(wnv_launch_zbias(m, h->d_layers, h->d_W, has_g ? a->g : nullptr, (has_g && !a->g) ? (const long long*)a->g_ids : nullptr,
                             h->embed_off >= 0 ? h->d_W + h->embed_off : nullptr, Bz, (float*)h->zbias.p, s));
    HIP_TRY(h->fwd.ensure(wnv_forward_scratch_floats(m, a->B, a->T) * sizeof(float)));
    WnvForwardArgs fa{};
    fa.B = a->B; fa.T = a->T; fa.x = a->x; fa.c_up = a->c_up; fa.zbias = (const float*)h->zbias.p;
    fa.zbias_bstride = has_g ? (long long)m.L * m.Gp : 0; fa.scratch = (float*)h->fwd.p; fa.out = a->out; fa.softmax = a->softmax;
    HIP_TRY(wnv_launch_forward(m, h->layers.data(), h->d_W, fa, s));
    return WNV_OK;
}

// ------------------------------------------------------------------------------------------------
// the hot loop
// ------------------------------------------------------------------------------------------------
// ---- one persistent launch at a time per device ---------------------------------------------------------------------------
// The ring and group-ring kernels make progress only with ALL their workgroups resident.  Two of them in flight on one device
// (two handles on two streams or threads) can each hold part of the CUs and starve the other until the bounded spins give up
// -- WNV_ERR_TIMEOUT, and in auto mode a handle that stays on the generic kernel for no better reason than bad timing.  Inside
// one process they therefore take turns: the host side of a persistent launch runs under a per-device mutex, and the launch is
// ordered on the device behind an event recorded after the previous one (which matters for WNV_GEN_ASYNC launches, whose
// kernels are still running when the call returns).  Other processes on the same GPU are what the time-out fallback is for.
namespace {
struct PersistentTurn {
    std::mutex m;
    hipEvent_t done = nullptr;      // recorded behind the last persistent launch on this device
};
PersistentTurn g_turns[64];
class TurnGuard {                   // construct with the device current (DeviceGuard)
  public:
    TurnGuard(int device, hipStream_t s) : t_(g_turns[device & 63]), s_(s), on_(!wnv_knob("WNV_NO_TURN")) {   // (diagnostic knob:
        if (!on_) return;                                                // tests/test_gpu_zz_boundary.py shows what the turn prevents)
        t_.m.lock();
        if (t_.done) (void)hipStreamWaitEvent(s_, t_.done, 0);
    }
    ~TurnGuard() {
        if (!on_) return;
        if (!t_.done && hipEventCreateWithFlags(&t_.done, hipEventDisableTiming) != hipSuccess) t_.done = nullptr;
        if (t_.done) (void)hipEventRecord(t_.done, s_);
        t_.m.unlock();
    }
    TurnGuard(const TurnGuard&) = delete;
    TurnGuard& operator=(const TurnGuard&) = delete;
  private:
    PersistentTurn& t_;
    hipStream_t s_;
    bool on_;
};
}  // namespace

extern "C" wnv_status wnv_generate(wnv_handle h, const wnv_generate_args* a) {
    if (!h || !a) return fail(WNV_ERR_INVALID_ARG, "NULL handle or args");
    if (h->device < 0) return fail(WNV_ERR_INVALID_ARG, "host-only handle (created with device = -1): there is no CPU path");
    if (!h->packed) return fail(WNV_ERR_NOT_LOADED, "weights are not loaded");
    const wnv_config& c = h->cfg;
    const WnvModelDev& m = h->m;
    if (a->B <= 0 || a->T <= 0 || a->T > 0x7fffffffLL) return fail(WNV_ERR_INVALID_ARG, "B and T must be positive (T < 2^31)");
    // (out may be NULL for a one-hot model that samples classes: the caller then takes index_out only -- a (B, out_channels, T) one-hot
    //  output is 1 KB per sample for a 256-way model, index_out 4 bytes)
    if (!a->out && !(a->index_out && !c.scalar_input && a->quantize && (a->seg_start || a->kernel == 1)))
        return fail(WNV_ERR_INVALID_ARG, "out is NULL (allowed only for a one-hot model with quantize = 1 and index_out given, in a packed-slot launch or on the generic kernel)");
    if (m.cin > 0 && !a->c_up) return fail(WNV_ERR_INVALID_ARG, "model has local conditioning but c_up is NULL");
    if (m.cin == 0 && a->c_up) return fail(WNV_ERR_INVALID_ARG, "c_up given but the model has no local conditioning");
    if (m.gin > 0 && !a->g && !a->g_ids) return fail(WNV_ERR_INVALID_ARG, "model has global conditioning but neither g nor g_ids is given");
    if (m.gin == 0 && (a->g || a->g_ids)) return fail(WNV_ERR_INVALID_ARG, "g given but the model has no global conditioning");
    if (a->g_ids && !a->g && h->embed_off < 0) return fail(WNV_ERR_INVALID_ARG, "g_ids given but the model has no speaker embedding");
    if (a->Tt < 0 || a->Tt > a->T || (a->Tt > 0 && !a->teacher)) return fail(WNV_ERR_INVALID_ARG, "bad teacher-forcing arguments");
    if (a->kernel < 0 || a->kernel > 3) return fail(WNV_ERR_INVALID_ARG, "unknown kernel selector %d", a->kernel);
    if ((a->flags & WNV_GEN_ASYNC) && a->kernel != 2) return fail(WNV_ERR_INVALID_ARG, "WNV_GEN_ASYNC needs kernel = 2 (the ring kernel chosen explicitly: auto mode must see the launch's status to fall back)");
    if (a->noise_ready && !(a->noise && a->kernel == 2 && (a->flags & WNV_GEN_ASYNC)))
        return fail(WNV_ERR_INVALID_ARG, "a streamed noise tape (noise_ready) needs noise, kernel = 2 and WNV_GEN_ASYNC: the caller fills the tape while the kernel runs");
    if ((a->seg_start == nullptr) != (a->seg_uid == nullptr)) return fail(WNV_ERR_INVALID_ARG, "seg_start and seg_uid come together (packed slots)");
    if (a->seg_start) {
        if (a->noise || a->teacher || a->initial)
            return fail(WNV_ERR_INVALID_ARG, "packed slots take in-kernel noise and no teacher / initial input");
        if (m.gin > 0 && (!a->seg_gid || a->n_g <= 0))
            return fail(WNV_ERR_INVALID_ARG, "packed slots of a model with global conditioning need seg_gid (B, T) and n_g > 0 rows of g / g_ids");
        if (m.gin == 0 && (a->seg_gid || a->n_g != 0)) return fail(WNV_ERR_INVALID_ARG, "seg_gid / n_g given but the model has no global conditioning");
        if (a->kernel != 0 && a->kernel != 2) return fail(WNV_ERR_UNSUPPORTED, "packed slots run on the pipelined ring kernel only");
        if (!(wnv_ring_supported(c, a->B) && wnv_ring_default())) return fail(WNV_ERR_UNSUPPORTED, "packed slots need the pipelined ring kernel: %s", wnv_ring_why_not(c, a->B));
    }
    if (!a->seg_start && (a->seg_gid || a->n_g != 0)) return fail(WNV_ERR_INVALID_ARG, "seg_gid / n_g belong to packed slots (seg_start, seg_uid)");
    if ((a->flags & WNV_GEN_ASYNC) && a->B > 64) return fail(WNV_ERR_INVALID_ARG, "WNV_GEN_ASYNC takes at most 64 utterances per call (larger batches run as several launches)");
    DeviceGuard g(h->device);
    hipStream_t s = (hipStream_t)a->stream;
    // a pending asynchronous launch reads the handle's scratch (the bias table below) on every step: it reports -- and ends -- first
    if (h->ring_state) {
        std::string err;
        const wnv_status pst = wnv_ring_wait(h->ring_state, err);
        if (pst != WNV_OK) return fail(pst, "deferred from the previous asynchronous call: %s", err.c_str());
    }
    const bool has_g = m.gin > 0;
    const int Bz = has_g ? (a->seg_gid ? a->n_g : a->B) : 1;         // (packed slots: one row per speaker / utterance of the job, picked through seg_gid)
    HIP_TRY(h->zbias.ensure((size_t)Bz * m.L * m.Gp * sizeof(float)));
    HIP_TRY(wnv_launch_zbias(m, h->d_layers, h->d_W, has_g ? a->g : nullptr, (has_g && !a->g) ? (const long long*)a->g_ids : nullptr,
                             h->embed_off >= 0 ? h->d_W + h->embed_off : nullptr, Bz, (float*)h->zbias.p, s));
    auto zero_onehot_out = [&]() -> hipError_t {                     // the kernels write only the sampled class (wavenet.py:334)
        if (!c.scalar_input && a->quantize && a->out) return hipMemsetAsync(a->out, 0, (size_t)a->B * m.O * (size_t)a->T * sizeof(float), s);
        return hipSuccess;
    };

    WnvGenArgs ga{};
    ga.B = a->B; ga.T = a->T; ga.Tt = a->teacher ? a->Tt : 0;
    ga.c_up = a->c_up; ga.initial = a->initial; ga.teacher = a->teacher; ga.noise = a->noise;
    ga.zbias = (const float*)h->zbias.p; ga.zbias_bstride = has_g ? (long long)m.L * m.Gp : 0;
    ga.seed = a->seed; ga.softmax = a->softmax; ga.quantize = c.scalar_input ? 1 : a->quantize;
    ga.nz = wnv_noise_width(&c);
    ga.noise_ready = a->noise_ready;
    ga.seg_start = a->seg_start; ga.seg_uid = a->seg_uid; ga.seg_gid = a->seg_gid;
    ga.out = a->out; ga.params_out = a->params_out; ga.index_out = a->index_out;
    // asynchronous ring launches only when the caller chose the ring explicitly: auto mode must see the status to fall back
    ga.async = (a->flags & WNV_GEN_ASYNC) && a->kernel == 2;

    int kernel = a->kernel;
    if (kernel == 0) {
        // (the ring kernel takes any batch -- more than 64 utterances run in slices --, so the group ring is only for models that
        //  exceed the ring's geometry)
        if (!h->ring_disabled && wnv_ring_supported(c, a->B) && wnv_ring_default()) kernel = 2;
        else if (!h->wide_disabled && !wnv_ring_supported(c, a->B) && wnv_wide_supported(c, a->B) && wnv_ring_default()) kernel = 3;
        else kernel = 1;
        if (kernel != 1 && h->persist_cooldown > 0) { --h->persist_cooldown; kernel = 1; }     // pausing after a time-out (see wnv_engine)
        if (a->seg_start && kernel != 2) return fail(WNV_ERR_UNSUPPORTED, "packed slots need the pipelined ring kernel, which this handle is not using right now");
    }
    if (kernel == 3) {                     // wide models: one GROUP of 8 workgroups per layer (wnv_wide.hip); same fallback rules as the ring
        if (!wnv_wide_supported(c, a->B)) return fail(WNV_ERR_UNSUPPORTED, "the group-ring kernel does not cover this configuration: %s", wnv_wide_why_not(c, a->B));
        HIP_TRY(zero_onehot_out());
        std::string err;
        wnv_status st;
        {
            TurnGuard turn(h->device, s);
            st = wnv_wide_generate(&h->wide_state, h->device, c, h->store, ga, s, err);
        }
        if (st == WNV_OK) { h->last_kernel = 3; h->persist_backoff = 0; return WNV_OK; }
        const bool recoverable = st == WNV_ERR_UNSUPPORTED || st == WNV_ERR_TIMEOUT;
        if (!(a->kernel == 0 && recoverable)) return fail(st, "%s", err.c_str());
        if (st == WNV_ERR_TIMEOUT) {
            h->persist_backoff = h->persist_backoff ? std::min(2 * h->persist_backoff, 32) : 2;
            h->persist_cooldown = h->persist_backoff;
            fprintf(stderr, "[wnv] device %d: the group-ring kernel timed out (%s); its workgroups were not co-resident -- this call and the next %d "
                            "are served by the generic kernel, then it is tried again\n", h->device, err.c_str(), h->persist_cooldown);
        } else {
            h->wide_disabled = true;
            fprintf(stderr, "[wnv] device %d: the group-ring kernel cannot run here (%s); this handle now uses the generic kernel\n", h->device, err.c_str());
        }
    }
    if (kernel == 2) {
        if (!wnv_ring_supported(c, a->B)) return fail(WNV_ERR_UNSUPPORTED, "the pipelined ring kernel does not cover this configuration: %s", wnv_ring_why_not(c, a->B));
        HIP_TRY(zero_onehot_out());
        std::string err;
        wnv_status st;
#ifdef WNV_TEST_HOOKS
        if (a->kernel == 0 && h->inject_timeouts > 0) {              // test hook (wnv_debug_inject_timeouts): exercises the time-out policy
            --h->inject_timeouts;                                     // below without a device that cannot keep the launch resident
            st = WNV_ERR_TIMEOUT;                                     // (tests/test_gpu_zz_boundary.py; the real thing: the CU-mask tests)
            err = "injected by wnv_debug_inject_timeouts";
        } else
#endif
        {
            TurnGuard turn(h->device, s);
            st = wnv_ring_generate(&h->ring_state, h->device, c, h->store, ga, s, err);
        }
        if (st == WNV_OK) { h->last_kernel = 2; h->persist_backoff = 0; return WNV_OK; }
        // auto mode: a device that cannot host the persistent pipeline (fewer CUs than one ring + its tap workgroups per XCD, a
        // partitioned GPU: WNV_ERR_UNSUPPORTED from the occupancy / placement checks) or did not keep it co-resident (CUs masked
        // or busy with another process: WNV_ERR_TIMEOUT after the bounded spins) is served by the generic kernel.  UNSUPPORTED is
        // permanent for the handle; after a TIMEOUT the persistent kernel is tried again after a pause that doubles with every
        // consecutive time-out (persist_cooldown).  An explicit kernel = 2 reports the reason instead.
        const bool recoverable = (st == WNV_ERR_UNSUPPORTED || st == WNV_ERR_TIMEOUT) && !a->seg_start;   // (packed slots: the caller re-plans)
        if (!(a->kernel == 0 && recoverable)) return fail(st, "%s", err.c_str());
        if (st == WNV_ERR_TIMEOUT) {
            h->persist_backoff = h->persist_backoff ? std::min(2 * h->persist_backoff, 32) : 2;
            h->persist_cooldown = h->persist_backoff;
            fprintf(stderr, "[wnv] device %d: the pipelined ring kernel timed out (%s); its workgroups were not co-resident -- "
                            "this call and the next %d are served by the generic kernel, then it is tried again\n", h->device, err.c_str(), h->persist_cooldown);
        } else if (!h->ring_disabled) {
            h->ring_disabled = true;
            fprintf(stderr, "[wnv] device %d: the pipelined ring kernel cannot run here (%s); using the generic kernel\n", h->device, err.c_str());
        }
    }
    HIP_TRY(zero_onehot_out());
    const size_t ring_bytes = std::max<size_t>((size_t)a->B * m.ring_floats * sizeof(float), 16);
    HIP_TRY(h->ring.ensure(ring_bytes));
    HIP_TRY(hipMemsetAsync(h->ring.p, 0, ring_bytes, s));                       // history before t = 0 is zero (conv.py:34-36)
    ga.ring = (float*)h->ring.p;
    HIP_TRY(wnv_launch_generate_generic(m, h->d_layers, h->d_W, ga, s));
    h->last_kernel = 1;
    return WNV_OK;
}
