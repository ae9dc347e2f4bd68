// wnv_generic.hip -- generic single-workgroup kernels for gfx950.
//
// One 1024-thread workgroup (16 wave64) owns ONE utterance and walks the whole network for every time
// step inside one launch: there is no host round trip between samples and no inter-workgroup traffic.
// Weights are streamed from L2 / Infinity Cache every step (K-major rows, 16 B per lane, eight loads in
// flight per wave); every shape the reference can express is supported, which is why this kernel is the
// fallback for configurations the pipelined ring kernel (wnv_ring.hip) does not cover, the engine behind
// the layer-level drop-ins, and the on-device cross-check of the ring kernel.
//
// Reference semantics implemented here (file:line under the reference tree):
//   conv.Conv1d.incremental_forward      conv.py:17-46      -> ring of (kw-1)*d rows, taps oldest first
//   ResidualConv1dGLU._forward(inc=True) modules.py:115-163 -> glu_layer()
//   WaveNet.incremental_forward loop     wavenet.py:296-336 -> wnv_generate_generic_kernel
//   sample_from_discretized_mix_logistic mixture.py:118-156 -> sample_scalar()
//   sample_from_mix_gaussian             mixture.py:221-270 -> sample_scalar()
//   softmax + OneHotCategorical          wavenet.py:332-335 -> sample_categorical()
#include "wnv_internal.h"
#include "wnv_matvec.h"
#include "wnv_sample.h"

namespace {

constexpr int NW = WNV_GENERIC_WAVES;
constexpr int NT = WNV_GENERIC_THREADS;

// One ResidualConv1dGLU step (modules.py:127-163) on the vector held in xin = [taps | h | c_t].
// Thread n = tid gets back: n < R -> new h (also stored to xin[hoff+n]); R <= n < R+K -> skip output s.
// Three barriers inside; the caller must have put a barrier between filling xin and this call.
__device__ __forceinline__ float glu_layer(const float* __restrict__ W, const WnvLayerDev& Ld,
                                           const WnvModelDev& m, const float* __restrict__ zb,
                                           float* xin, float* ubuf, float* part, int tid, int wave,
                                           int lane) {
    const int R = m.R, H = m.G >> 1, K = m.K;
    const int hoff = (m.kw - 1) * R;
    const int Kin = m.kw * R + (m.cin > 0 ? m.cin : 0);
    const int ps = m.lds_part_stride;
    float za = 0.f, zbv = 0.f, bo = 0.f;       // biases fetched ahead of the weight stream
    if (tid < H) { za = zb[tid]; zbv = zb[H + tid]; }
    if (tid < R + K) bo = W[Ld.b_os + tid];
    matvec_partial<NW>(W + Ld.w_in, Kin, m.Gp, xin, part, ps, wave, lane);
    __syncthreads();
    if (tid < H) {
        const float a = reduce_part<NW>(part, ps, tid, za);
        const float b = reduce_part<NW>(part, ps, H + tid, zbv);
        ubuf[tid] = tanhf(a) * wnv_sigmoid(b);                                  // modules.py:154
    }
    __syncthreads();
    matvec_partial<NW>(W + Ld.w_os, H, m.NOSp, ubuf, part, ps, wave, lane);
    __syncthreads();
    float ret = 0.f;
    if (tid < R + K) {
        const float o = reduce_part<NW>(part, ps, tid, bo);
        if (tid < R) {
            ret = (o + xin[hoff + tid]) * 0.70710678118654752440f;             // modules.py:162
            xin[hoff + tid] = ret;
        } else {
            ret = o;                                                            // modules.py:157
        }
    }
    return ret;
}

struct Lds {
    float *xin, *ubuf, *obuf, *vin, *taps, *nz, *part;
    int* ints;   // [0] previous sampled class, [1..] per-layer tables: rows, off, dil
    float* flt;  // [0] previous scalar sample
};
__device__ __forceinline__ Lds carve(float* smem, const WnvModelDev& m) {
    Lds s;
    s.xin = smem;
    s.ubuf = s.xin + m.lds_xin;
    s.obuf = s.ubuf + m.lds_u;
    s.vin = s.obuf + m.lds_o;
    s.taps = s.vin + m.lds_vin;
    s.nz = s.taps + m.lds_taps;
    s.part = s.nz + m.lds_nz;
    s.flt = s.part + (size_t)NW * m.lds_part_stride;
    s.ints = reinterpret_cast<int*>(s.flt + 4);
    return s;
}

__device__ __forceinline__ float tape_or_gen(const WnvGenArgs& a, int t, int b, int j, int kind) {
    if (a.noise) return a.noise[((size_t)t * a.B + b) * a.nz + j];
    return wnv_noise_gen(a.seed, t, b, j, kind);
}

// ---- the whole autoregressive loop for one utterance ------------------------------------------------
__global__ void __launch_bounds__(NT) wnv_generate_generic_kernel(const WnvModelDev m,
                                                                  const WnvLayerDev* __restrict__ layers,
                                                                  const float* __restrict__ W,
                                                                  const WnvGenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds s = carve(smem, m);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int R = m.R, K = m.K, O = m.O, kw = m.kw, cin = m.cin > 0 ? m.cin : 0, L = m.L;
    const int hoff = (kw - 1) * R, coff = kw * R;
    const int T = (int)a.T, Tt = (int)a.Tt;
    float* ring = a.ring + (size_t)b * m.ring_floats;
    const float* zb_base = a.zbias + (size_t)b * a.zbias_bstride;
    int* lay_rows = s.ints + 4;
    int* lay_off = lay_rows + L;
    int* lay_dil = lay_off + L;
    for (int l = tid; l < L; l += NT) {
        lay_rows[l] = layers[l].ring_rows;
        lay_off[l] = (int)layers[l].ring_off;
        lay_dil[l] = layers[l].dilation;
    }
    if (tid == 0) { s.ints[0] = 127; s.flt[0] = 0.f; }
    float skip_acc = 0.f;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // ---- step prologue: everything that does not depend on this step's input -----------------
        if (m.taps_in_lds && kw > 1) {
            const int total = L * hoff;
            for (int idx = tid; idx < total; idx += NT) {
                const int l = idx / hoff, rem = idx - l * hoff;
                const int k = rem / R, r = rem - k * R;
                const int slot = (t + k * lay_dil[l]) % lay_rows[l];     // time t - (kw-1-k)*d
                s.taps[idx] = ring[lay_off[l] + (size_t)slot * R + r];
            }
        }
        for (int j = tid; j < cin; j += NT) s.xin[coff + j] = a.c_up[((size_t)b * T + t) * cin + j];
        {
            const int kind = m.scalar_input ? 0 : 2;
            for (int j = tid; j < a.nz; j += NT) {
                int kd = kind;
                if (m.scalar_input && m.dist == 2 && j == a.nz - 1) kd = 1;
                s.nz[j] = tape_or_gen(a, t, b, j, kd);
            }
        }
        // ---- first_conv (wavenet.py:308) -----------------------------------------------------------
        if (m.cin1 == 1) {
            float xs;
            if (t < Tt) xs = a.teacher[(size_t)b * Tt + t];                      // wavenet.py:297-298
            else if (t == 0) xs = a.initial ? a.initial[b] : 0.f;                // wavenet.py:283
            else xs = s.flt[0];
            if (tid < R) s.xin[hoff + tid] = fmaf(W[m.w_first + tid], xs, W[m.b_first + tid]);
        } else {
            const float* dense = nullptr;
            int idx = -1;
            if (t < Tt) dense = a.teacher + ((size_t)b * Tt + t) * m.cin1;
            else if (t == 0) { if (a.initial) dense = a.initial + (size_t)b * m.cin1; else idx = 127; }
            else if (a.quantize) idx = s.ints[0];
            if (idx >= 0) {
                // one-hot input: F.linear(onehot, W) is exactly column idx (+ bias): a row of the K-major blob
                if (tid < R) s.xin[hoff + tid] = W[m.w_first + (size_t)idx * m.Rp + tid] + W[m.b_first + tid];
            } else {
                if (dense) { for (int i = tid; i < m.cin1; i += NT) s.vin[i] = dense[i]; }
                else { for (int i = tid; i < m.cin1; i += NT) s.vin[i] = s.obuf[i]; }   // fed-back probabilities
                __syncthreads();
                matvec_partial<NW>(W + m.w_first, m.cin1, m.Rp, s.vin, s.part, m.lds_part_stride, wave, lane);
                __syncthreads();
                if (tid < R) s.xin[hoff + tid] = reduce_part<NW>(s.part, m.lds_part_stride, tid, W[m.b_first + tid]);
            }
        }
        // prologue LDS writes (taps, c_t, noise) and h must be visible to every wave
        __syncthreads();
        // ---- residual stack (wavenet.py:310-312) ---------------------------------------------------
        for (int l = 0; l < L; ++l) {
            const WnvLayerDev Ld = layers[l];
            if (kw > 1) {
                if (m.taps_in_lds) {
                    for (int idx = tid; idx < hoff; idx += NT) s.xin[idx] = s.taps[l * hoff + idx];
                    if (tid < R) ring[Ld.ring_off + (size_t)(t % Ld.ring_rows) * R + tid] = s.xin[hoff + tid];
                } else {
                    // oldest tap and the row being written share a ring slot: the thread that reads
                    // element r of tap 0 is the one that overwrites it (read, then write)
                    for (int idx = tid; idx < hoff; idx += NT) {
                        const int k = idx / R, r = idx - k * R;
                        float* p = ring + Ld.ring_off + (size_t)((t + k * Ld.dilation) % Ld.ring_rows) * R + r;
                        const float v = *p;
                        if (k == 0) *p = s.xin[hoff + r];
                        s.xin[idx] = v;
                    }
                }
            }
            __syncthreads();
            const float v = glu_layer(W, Ld, m, zb_base + (size_t)l * m.Gp, s.xin, s.ubuf, s.part, tid, wave, lane);
            if (tid >= R && tid < R + K) skip_acc += v;                          // wavenet.py:312
        }
        // ---- head (wavenet.py:313-319) -------------------------------------------------------------
        __syncthreads();
        if (tid >= R && tid < R + K) {
            s.ubuf[tid - R] = fmaxf(skip_acc * m.skip_scale, 0.f);
            skip_acc = 0.f;
        }
        float bh1 = 0.f, bh2 = 0.f;
        if (tid < K) bh1 = W[m.b_h1 + tid];
        if (tid < O) bh2 = W[m.b_h2 + tid];
        __syncthreads();
        matvec_partial<NW>(W + m.w_h1, K, m.Kp, s.ubuf, s.part, m.lds_part_stride, wave, lane);
        __syncthreads();
        if (tid < K) s.ubuf[tid] = fmaxf(reduce_part<NW>(s.part, m.lds_part_stride, tid, bh1), 0.f);
        __syncthreads();
        matvec_partial<NW>(W + m.w_h2, K, m.Op, s.ubuf, s.part, m.lds_part_stride, wave, lane);
        __syncthreads();
        if (tid < O) {
            const float o = reduce_part<NW>(s.part, m.lds_part_stride, tid, bh2);
            s.obuf[tid] = o;
            if (a.params_out) a.params_out[((size_t)b * O + tid) * T + t] = o;
        }
        __syncthreads();
        // ---- sampling (wavenet.py:322-336) ---------------------------------------------------------
        if (wave == 0) {
            if (m.scalar_input) {
                const float x = sample_scalar(m.dist, m.O, s.obuf, s.nz, lane);
                if (lane == 0) { a.out[(size_t)b * T + t] = x; s.flt[0] = x; }
            } else {
                const int idx = sample_categorical(m.O, s.obuf, s.nz, a.softmax, a.quantize, lane);
                if (a.quantize) {
                    if (lane == 0) {
                        if (a.out) a.out[((size_t)b * O + idx) * T + t] = 1.0f;  // out is pre-zeroed (NULL: classes only, index_out)
                        if (a.index_out) a.index_out[(size_t)b * T + t] = idx;
                        s.ints[0] = idx;
                    }
                } else {
                    for (int n = lane; n < O; n += 64) a.out[((size_t)b * O + n) * T + t] = s.obuf[n];
                }
            }
        }
        __syncthreads();
    }
}

__global__ void wnv_zbias_kernel(const WnvModelDev m, const WnvLayerDev* __restrict__ layers,
                                 const float* __restrict__ W, const float* __restrict__ g,
                                 const long long* __restrict__ ids, const float* __restrict__ embed,
                                 float* __restrict__ zbias) {
    const int b = blockIdx.x, l = blockIdx.y;
    const WnvLayerDev Ld = layers[l];
    const bool has_g = (g != nullptr || ids != nullptr) && Ld.w_g >= 0;
    const float* gv = nullptr;
    if (has_g) {                                                                      // wavenet.py:264-268
        // an id outside the table is the host's IndexError (nn.Embedding); here it is clamped so that no launch can read past it
        long long id = ids ? ids[b] : 0;
        id = id < 0 ? 0 : (id >= m.n_embed ? (long long)m.n_embed - 1 : id);
        gv = g ? g + (size_t)b * m.gin : embed + (size_t)id * m.gin;
    }
    for (int n = threadIdx.x; n < m.Gp; n += blockDim.x) {
        float acc = 0.f;
        if (has_g)
            for (int j = 0; j < m.gin; ++j) acc = fmaf(W[Ld.w_g + (size_t)j * m.Gp + n], gv[j], acc);
        zbias[((size_t)b * m.L + l) * m.Gp + n] = W[Ld.b_in + n] + acc;
    }
}

// ---- layer-level drop-ins ---------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) wnv_glu_step_kernel(const WnvModelDev m, const WnvLayerDev* __restrict__ layer,
                                                          const float* __restrict__ W, const WnvGluStepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds s = carve(smem, m);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const WnvLayerDev Ld = layer[0];
    const int R = m.R, K = m.K, kw = m.kw, cin = m.cin > 0 ? m.cin : 0, gin = m.gin > 0 ? m.gin : 0;
    const int hoff = (kw - 1) * R, coff = kw * R;
    float* ring = a.ring + (size_t)b * m.ring_floats;
    if (tid < R) s.xin[hoff + tid] = a.x[(size_t)b * R + tid];
    for (int j = tid; j < cin; j += NT) s.xin[coff + j] = a.c ? a.c[(size_t)b * cin + j] : 0.f;
    for (int j = tid; j < gin; j += NT) s.vin[j] = a.g ? a.g[(size_t)b * gin + j] : 0.f;
    __syncthreads();
    // effective conv bias = b_in + Wg . g   (kept in obuf: [Gp])
    for (int n = tid; n < m.Gp; n += NT) {
        float acc = 0.f;
        if (Ld.w_g >= 0 && a.g)
            for (int j = 0; j < gin; ++j) acc = fmaf(W[Ld.w_g + (size_t)j * m.Gp + n], s.vin[j], acc);
        s.obuf[n] = W[Ld.b_in + n] + acc;
    }
    if (kw > 1) {
        for (int idx = tid; idx < hoff; idx += NT) {
            const int k = idx / R, r = idx - k * R;
            float* p = ring + (size_t)((a.t + k * Ld.dilation) % Ld.ring_rows) * R + r;
            const float v = *p;
            if (k == 0) *p = s.xin[hoff + r];
            s.xin[idx] = v;
        }
    }
    __syncthreads();
    const float v = glu_layer(W, Ld, m, s.obuf, s.xin, s.ubuf, s.part, tid, wave, lane);
    if (tid < R) a.x_out[(size_t)b * R + tid] = v;
    else if (tid < R + K) a.s_out[(size_t)b * K + (tid - R)] = v;
}

__global__ void __launch_bounds__(NT) wnv_qconv_step_kernel(const WnvQconvDev q, const float* __restrict__ W,
                                                            const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ ring_all, int t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int Kin = q.kw * q.cin, hoff = (q.kw - 1) * q.cin;
    float* xin = smem;
    float* part = smem + ((Kin + 3) & ~3);
    float* ring = ring_all + (size_t)b * q.ring_rows * q.cin;
    for (int i = tid; i < q.cin; i += NT) xin[hoff + i] = x[(size_t)b * q.cin + i];
    __syncthreads();
    for (int idx = tid; idx < hoff; idx += NT) {
        const int k = idx / q.cin, r = idx - k * q.cin;
        float* p = ring + (size_t)((t + k * q.dilation) % q.ring_rows) * q.cin + r;
        const float v = *p;
        if (k == 0) *p = xin[hoff + r];
        xin[idx] = v;
    }
    __syncthreads();
    matvec_partial<NW>(W + q.w, Kin, q.coutp, xin, part, q.coutp, wave, lane);
    __syncthreads();
    for (int n = tid; n < q.cout; n += NT) y[(size_t)b * q.cout + n] = reduce_part<NW>(part, q.coutp, n, W[q.b + n]);
}

}  // namespace

size_t wnv_generic_lds_bytes(const WnvModelDev& m) {
    size_t fl = (size_t)m.lds_xin + m.lds_u + m.lds_o + m.lds_vin + m.lds_taps + m.lds_nz +
                (size_t)NW * m.lds_part_stride + 4 /*flt*/ + 4 + 3 * (size_t)m.L /*ints*/;
    return fl * sizeof(float) + 16;
}

static hipError_t set_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

hipError_t wnv_launch_generate_generic(const WnvModelDev& m, const WnvLayerDev* d_layers, const float* d_W,
                                       const WnvGenArgs& a, hipStream_t s) {
    const size_t lds = wnv_generic_lds_bytes(m);
    hipError_t e = set_lds((const void*)wnv_generate_generic_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wnv_generate_generic_kernel, dim3(a.B), dim3(NT), lds, s, m, d_layers, d_W, a);
    return hipGetLastError();
}

hipError_t wnv_launch_zbias(const WnvModelDev& m, const WnvLayerDev* d_layers, const float* d_W,
                            const float* g, const long long* ids, const float* embed, int B,
                            float* zbias, hipStream_t s) {
    hipLaunchKernelGGL(wnv_zbias_kernel, dim3(B, m.L), dim3(256), 0, s, m, d_layers, d_W, g, ids, embed, zbias);
    return hipGetLastError();
}

hipError_t wnv_launch_glu_step(const WnvModelDev& m, const WnvLayerDev* d_layer, const float* d_W,
                               const WnvGluStepArgs& a, hipStream_t s) {
    const size_t lds = wnv_generic_lds_bytes(m);
    hipError_t e = set_lds((const void*)wnv_glu_step_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wnv_glu_step_kernel, dim3(a.B), dim3(NT), lds, s, m, d_layer, d_W, a);
    return hipGetLastError();
}

hipError_t wnv_launch_qconv_step(const WnvQconvDev& q, const float* d_W, const float* x, float* y,
                                 float* ring, int B, int t, hipStream_t s) {
    const size_t lds = ((size_t)((q.kw * q.cin + 3) & ~3) + (size_t)NW * q.coutp) * sizeof(float);
    hipError_t e = set_lds((const void*)wnv_qconv_step_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wnv_qconv_step_kernel, dim3(B), dim3(NT), lds, s, q, d_W, x, y, ring, t);
    return hipGetLastError();
}
