// wnv_layers.cpp -- layer-level drop-ins of include/wnv.h:
//   wnv_qconv_*  = conv.Conv1d.incremental_forward / clear_buffer        (reference conv.py:17-49)
//   wnv_glu_*    = ResidualConv1dGLU.incremental_forward / clear_buffer  (reference modules.py:112-169)
// Both keep their history ring on the device, created zeroed on the first step after a reset for that batch
// size (conv.py:34-36), and run the same device code as the whole-network kernel (wnv_generic.hip).
#include <cstring>

#include "wnv_hostutil.h"
#include "wnv_internal.h"

struct wnv_qconv {
    WnvQconvDev q{};
    int device = 0;
    float* d_W = nullptr;
    Scratch ring;
    int ring_B = 0;     // batch the ring was created for (0 = cleared)
    int t = 0;          // steps since the last reset (kept reduced modulo the ring length)
};

extern "C" wnv_status wnv_qconv_create(int32_t cin, int32_t cout, int32_t kernel_size, int32_t dilation,
                                       int32_t device, wnv_qconv_handle* out) {
    if (!out) return fail(WNV_ERR_INVALID_ARG, "out handle is NULL");
    *out = nullptr;
    if (cin <= 0 || cout <= 0 || kernel_size < 1 || kernel_size > 64 || dilation < 1)
        return fail(WNV_ERR_INVALID_ARG, "bad conv geometry");
    const size_t lds = ((size_t)pad4(kernel_size * cin) + (size_t)WNV_GENERIC_WAVES * pad4(cout)) * sizeof(float);
    if (lds > 160 * 1024) return fail(WNV_ERR_UNSUPPORTED, "conv needs %zu bytes of LDS", lds);
    wnv_qconv* q = new wnv_qconv();
    q->device = device;
    q->q.cin = cin; q->q.cout = cout; q->q.coutp = pad4(cout); q->q.kw = kernel_size; q->q.dilation = dilation;
    q->q.ring_rows = (kernel_size - 1) * dilation;
    *out = q;
    return WNV_OK;
}

extern "C" wnv_status wnv_qconv_set_weights(wnv_qconv_handle q, const float* weight, const float* bias) {
    if (!q || !weight) return fail(WNV_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard g(q->device);
    Blob b;
    HostTensor w;
    w.shape = {q->q.cout, q->q.cin, q->q.kw};
    w.data.assign(weight, weight + (size_t)q->q.cout * q->q.cin * q->q.kw);
    q->q.w = b.alloc((size_t)q->q.kw * q->q.cin * q->q.coutp);
    put_kmajor(b, q->q.w, q->q.coutp, 0, w, 0);                               // conv.py:51-62
    q->q.b = b.alloc(q->q.coutp);
    if (bias) std::copy(bias, bias + q->q.cout, b.v.begin() + q->q.b);
    if (q->d_W) { HIP_TRY(hipFree(q->d_W)); q->d_W = nullptr; }
    HIP_TRY(hipMalloc((void**)&q->d_W, b.v.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(q->d_W, b.v.data(), b.v.size() * sizeof(float), hipMemcpyHostToDevice));
    return WNV_OK;
}

extern "C" wnv_status wnv_qconv_step(wnv_qconv_handle q, const float* x, float* y, int32_t B, void* stream) {
    if (!q || !x || !y || B <= 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_qconv_step");
    if (!q->d_W) return fail(WNV_ERR_NOT_LOADED, "weights are not set");
    DeviceGuard g(q->device);
    hipStream_t s = (hipStream_t)stream;
    if (q->ring_B == 0) {                                                       // conv.py:34-36 lazy zero buffer
        const size_t bytes = std::max<size_t>((size_t)B * q->q.ring_rows * q->q.cin * sizeof(float), 16);
        HIP_TRY(q->ring.ensure(bytes));
        HIP_TRY(hipMemsetAsync(q->ring.p, 0, bytes, s));
        q->ring_B = B;
        q->t = 0;
    } else if (q->ring_B != B) {
        return fail(WNV_ERR_INVALID_ARG, "batch size changed from %d to %d without clear_buffer()", q->ring_B, B);
    }
    HIP_TRY(wnv_launch_qconv_step(q->q, q->d_W, x, y, (float*)q->ring.p, B, q->t, s));
    q->t += 1;                                                                  // only t mod ring_rows matters
    if (q->q.ring_rows > 0 && q->t >= 2 * q->q.ring_rows) q->t -= q->q.ring_rows;
    return WNV_OK;
}

extern "C" wnv_status wnv_qconv_reset(wnv_qconv_handle q) {
    if (!q) return fail(WNV_ERR_INVALID_ARG, "NULL handle");
    q->ring_B = 0;
    q->t = 0;
    return WNV_OK;
}

extern "C" wnv_status wnv_qconv_destroy(wnv_qconv_handle q) {
    if (!q) return WNV_OK;
    DeviceGuard g(q->device);
    if (q->d_W) (void)hipFree(q->d_W);
    q->ring.release();
    delete q;
    return WNV_OK;
}

struct wnv_glu {
    wnv_glu_config cfg{};
    int device = 0;
    TensorStore store;
    WnvModelDev m{};
    WnvLayerDev layer{};
    float* d_W = nullptr;
    WnvLayerDev* d_layer = nullptr;
    Scratch ring;
    int ring_B = 0, t = 0;
};

static std::vector<Expect> glu_expected(const wnv_glu_config& c) {
    std::vector<Expect> e;
    const int64_t R = c.residual_channels, G = c.gate_channels, K = c.skip_out_channels;
    e.push_back({"conv.weight", {G, R, c.kernel_size}});
    if (c.bias) e.push_back({"conv.bias", {G}});
    if (c.cin_channels > 0) e.push_back({"conv1x1c.weight", {G, c.cin_channels, 1}});
    if (c.gin_channels > 0) e.push_back({"conv1x1g.weight", {G, c.gin_channels, 1}});
    e.push_back({"conv1x1_out.weight", {R, G / 2, 1}});
    if (c.bias) e.push_back({"conv1x1_out.bias", {R}});
    e.push_back({"conv1x1_skip.weight", {K, G / 2, 1}});
    if (c.bias) e.push_back({"conv1x1_skip.bias", {K}});
    return e;
}

extern "C" wnv_status wnv_glu_create(const wnv_glu_config* cfg, int32_t device, wnv_glu_handle* out) {
    if (!out || !cfg) return fail(WNV_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (cfg->residual_channels <= 0 || cfg->gate_channels <= 0 || cfg->gate_channels % 2 || cfg->skip_out_channels <= 0 ||
        cfg->kernel_size < 1 || cfg->kernel_size > 16 || cfg->dilation < 1)
        return fail(WNV_ERR_INVALID_ARG, "bad ResidualConv1dGLU geometry");
    if (cfg->residual_channels + cfg->skip_out_channels > WNV_GENERIC_THREADS || cfg->gate_channels / 2 > WNV_GENERIC_THREADS)
        return fail(WNV_ERR_UNSUPPORTED, "channel counts exceed the %d-thread workgroup", WNV_GENERIC_THREADS);
    wnv_glu* g = new wnv_glu();
    g->cfg = *cfg;
    g->device = device;
    *out = g;
    return WNV_OK;
}

extern "C" wnv_status wnv_glu_load_weights(wnv_glu_handle g, const wnv_tensor* tensors, int32_t n) {
    if (!g || (!tensors && n > 0)) return fail(WNV_ERR_INVALID_ARG, "NULL argument");
    const wnv_glu_config& c = g->cfg;
    const auto exp = glu_expected(c);
    for (int i = 0; i < n; ++i) {
        std::string name = tensors[i].name ? tensors[i].name : "";
        std::string base = name;
        if (ends_with(name, "weight_g") || ends_with(name, "weight_v")) base = name.substr(0, name.size() - 2);
        bool known = false;
        for (const auto& e : exp) if (e.name == base) known = true;
        if (!known) return fail(WNV_ERR_INVALID_ARG, "unexpected key '%s'", name.c_str());
        g->store.put(tensors[i]);
    }
    for (const auto& e : exp) {
        const HostTensor* t = g->store.get(e.name);
        if (!t) return fail(WNV_ERR_NOT_LOADED, "missing tensor '%s'", e.name.c_str());
        if (!shape_eq(t->shape, e.shape)) return fail(WNV_ERR_INVALID_ARG, "size mismatch for %s", e.name.c_str());
    }
    DeviceGuard dg(g->device);
    const int R = c.residual_channels, G = c.gate_channels, K = c.skip_out_channels, kw = c.kernel_size;
    const int cin = c.cin_channels > 0 ? c.cin_channels : 0, gin = c.gin_channels > 0 ? c.gin_channels : 0, H = G / 2;
    WnvModelDev& m = g->m;
    memset(&m, 0, sizeof m);
    m.L = 1; m.R = R; m.G = G; m.K = K; m.O = 4; m.kw = kw; m.cin = cin; m.gin = gin; m.cin1 = 1;
    m.Rp = pad4(R); m.Gp = pad4(G); m.NOSp = pad4(R + K); m.Kp = pad4(K); m.Op = 4;
    Blob b;
    auto T = [&](const char* n) -> const HostTensor* { return g->store.get(n); };
    WnvLayerDev& Ld = g->layer;
    Ld.dilation = c.dilation; Ld.ring_rows = (kw - 1) * c.dilation; Ld.ring_off = 0;
    Ld.w_in = b.alloc((size_t)(kw * R + cin) * m.Gp);
    put_kmajor(b, Ld.w_in, m.Gp, 0, *T("conv.weight"), 0);
    if (cin > 0) put_kmajor(b, Ld.w_in, m.Gp, 0, *T("conv1x1c.weight"), kw * R);
    Ld.b_in = b.alloc(m.Gp);
    if (c.bias) std::copy(T("conv.bias")->data.begin(), T("conv.bias")->data.end(), b.v.begin() + Ld.b_in);
    Ld.w_g = -1;
    if (gin > 0) { Ld.w_g = b.alloc((size_t)gin * m.Gp); put_kmajor(b, Ld.w_g, m.Gp, 0, *T("conv1x1g.weight"), 0); }
    Ld.w_os = b.alloc((size_t)H * m.NOSp);
    put_kmajor(b, Ld.w_os, m.NOSp, 0, *T("conv1x1_out.weight"), 0);
    put_kmajor(b, Ld.w_os, m.NOSp, R, *T("conv1x1_skip.weight"), 0);
    Ld.b_os = b.alloc(m.NOSp);
    if (c.bias) {
        std::copy(T("conv1x1_out.bias")->data.begin(), T("conv1x1_out.bias")->data.end(), b.v.begin() + Ld.b_os);
        std::copy(T("conv1x1_skip.bias")->data.begin(), T("conv1x1_skip.bias")->data.end(), b.v.begin() + Ld.b_os + R);
    }
    m.ring_floats = (long long)Ld.ring_rows * R;
    m.lds_xin = pad4(kw * R + cin);
    m.lds_u = pad4(std::max(H, K));
    m.lds_o = m.Gp;                       // holds the effective conv bias (b_in + Wg.g)
    m.lds_vin = pad4(std::max(gin, 4));
    m.lds_nz = 4; m.lds_taps = 0; m.taps_in_lds = 0;
    m.lds_part_stride = std::max(m.Gp, m.NOSp);
    if (wnv_generic_lds_bytes(m) > 160 * 1024) return fail(WNV_ERR_UNSUPPORTED, "layer needs too much LDS");
    if (g->d_W) { HIP_TRY(hipFree(g->d_W)); g->d_W = nullptr; }
    if (g->d_layer) { HIP_TRY(hipFree(g->d_layer)); g->d_layer = nullptr; }
    HIP_TRY(hipMalloc((void**)&g->d_W, b.v.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(g->d_W, b.v.data(), b.v.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&g->d_layer, sizeof(WnvLayerDev)));
    HIP_TRY(hipMemcpy(g->d_layer, &g->layer, sizeof(WnvLayerDev), hipMemcpyHostToDevice));
    return WNV_OK;
}

extern "C" wnv_status wnv_glu_step(wnv_glu_handle g, const float* x, const float* c, const float* gcond,
                                   float* x_out, float* s_out, int32_t B, void* stream) {
    if (!g || !x || !x_out || !s_out || B <= 0) return fail(WNV_ERR_INVALID_ARG, "bad arguments to wnv_glu_step");
    if (!g->d_W) return fail(WNV_ERR_NOT_LOADED, "weights are not loaded");
    if (c && g->m.cin == 0) return fail(WNV_ERR_INVALID_ARG, "c given but the layer has no conv1x1c (modules.py:142 assert)");
    if (gcond && g->m.gin == 0) return fail(WNV_ERR_INVALID_ARG, "g given but the layer has no conv1x1g (modules.py:149 assert)");
    DeviceGuard dg(g->device);
    hipStream_t s = (hipStream_t)stream;
    if (g->ring_B == 0) {
        const size_t bytes = std::max<size_t>((size_t)B * g->m.ring_floats * sizeof(float), 16);
        HIP_TRY(g->ring.ensure(bytes));
        HIP_TRY(hipMemsetAsync(g->ring.p, 0, bytes, s));
        g->ring_B = B;
        g->t = 0;
    } else if (g->ring_B != B) {
        return fail(WNV_ERR_INVALID_ARG, "batch size changed from %d to %d without clear_buffer()", g->ring_B, B);
    }
    WnvGluStepArgs a{};
    a.B = B; a.t = g->t; a.x = x; a.c = c; a.g = gcond; a.x_out = x_out; a.s_out = s_out; a.ring = (float*)g->ring.p;
    HIP_TRY(wnv_launch_glu_step(g->m, g->d_layer, g->d_W, a, s));
    g->t += 1;
    if (g->layer.ring_rows > 0 && g->t >= 2 * g->layer.ring_rows) g->t -= g->layer.ring_rows;
    return WNV_OK;
}

extern "C" wnv_status wnv_glu_reset(wnv_glu_handle g) {
    if (!g) return fail(WNV_ERR_INVALID_ARG, "NULL handle");
    g->ring_B = 0;
    g->t = 0;
    return WNV_OK;
}

extern "C" wnv_status wnv_glu_destroy(wnv_glu_handle g) {
    if (!g) return WNV_OK;
    DeviceGuard dg(g->device);
    if (g->d_W) (void)hipFree(g->d_W);
    if (g->d_layer) (void)hipFree(g->d_layer);
    g->ring.release();
    delete g;
    return WNV_OK;
}
