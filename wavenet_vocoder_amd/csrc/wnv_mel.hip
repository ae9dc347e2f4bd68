// wnv_mel.hip -- the mel front end on the device: audio.logmelspectrogram (audio.py:101-109)
// followed by the mean-variance scaling of preprocess_normalize.py:32-47, for a batch of waveforms.
//
//   y (B, n) float32
//     -> librosa.stft(y, n_fft, hop, win_length, window="hann", center=True, pad_mode)        (audio.py:128-132)
//     -> np.dot(librosa.filters.mel(sr, n_fft, fmin, fmax, n_mels), |D|)                       (audio.py:145-157)
//     -> log10(max(S, 1e-10))                                                                   (audio.py:108)
//     -> optionally (S - mean) / scale  per mel bin  (sklearn StandardScaler.transform)         (preprocess_normalize.py:44)
//   out (B, frames, n_mels)  [the "*-feats.npy" layout, datasets/wavallin.py:62]  or (B, n_mels, frames)
//
// librosa is the reference's un-vendored dependency (setup.py:25, unpinned); what is implemented is its published
// definition: frames of n_fft samples every `hop` samples of the signal padded by n_fft/2 on both sides (reflection
// without the edge sample, or zeros), times the periodic Hann window of win_length samples centred in the frame, DFT
// bins 0 .. n_fft/2; Slaney mel scale (linear below 1 kHz, log above, 27 steps per factor 6.4), triangular filters
// normalised to unit area ("slaney" norm).  The filterbank, window and twiddle table are built on the host in double.
//
// Kernel: ONE workgroup transforms TWO frames with one complex FFT (frame A in the real part, frame B in the imaginary
// part, separated afterwards: X_A[k] = (Z[k] + conj Z[N-k]) / 2, X_B[k] = (Z[k] - conj Z[N-k]) / 2i) -- a radix-4
// Stockham autosort FFT in LDS (one radix-2 pass when log2 N is odd), twiddles from an LDS table; the magnitudes stay in
// LDS and the (sparse: each filter touches its own bin range only) mel projection, log and scaling finish in the same
// launch.  The signal is read once per frame it appears in (n_fft / hop = 4 times at the preset; those re-reads are L2
// hits), the output written once: HBM-bound by construction, algorithmic bytes = 4 n + 4 frames n_mels per utterance.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wnv.h"
#include "wnv_devguard.h"

namespace {

constexpr int MT = 256;

struct MelDev {
    int N, logN, hop, nbins, n_mels, pad_mode;
    float floor_;
    const float2* tw;        // [N] exp(-2 pi i k / N)
    const float* window;     // [N] Hann window of win_length samples, zero-padded to N (centred)
    const int* flo;          // [n_mels] first bin with a non-zero weight
    const int* fcnt;         // [n_mels] number of bins
    const int* foff;         // [n_mels] offset into fw
    const float* fw;         // packed weights
    const float* mean;       // [n_mels] or null
    const float* scale;      // [n_mels] or null
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// sample i of the padded signal (i in [-N/2, n + N/2)): librosa.stft center=True -> np.pad(y, N/2, mode)
__device__ __forceinline__ float padded_sample(const float* __restrict__ y, long long n, long long i, int reflect) {
    if (i >= 0 && i < n) return y[i];
    if (!reflect) return 0.f;
    if (i < 0) i = -i;                       // np.pad "reflect": the edge sample is not repeated
    if (i >= n) i = 2 * (n - 1) - i;
    return (i >= 0 && i < n) ? y[i] : 0.f;
}

__global__ void __launch_bounds__(MT) wnv_logmel_kernel(const MelDev m, const float* __restrict__ wav, long long n, long long wav_stride,
                                                        long long frames, float* __restrict__ out, int transpose, int normalize) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = m.N, tid = threadIdx.x;
    float2* bufa = reinterpret_cast<float2*>(smem);
    float2* bufb = bufa + N;
    float2* tw = bufb + N;
    float* mag = reinterpret_cast<float*>(tw + N);            // [2][nbins]
    const int pairs = (int)((frames + 1) / 2);
    const int b = blockIdx.x / pairs;
    const long long f0 = 2ll * (blockIdx.x % pairs);
    const float* y = wav + (size_t)b * wav_stride;
    const bool have_b = f0 + 1 < frames;

    for (int k = tid; k < N; k += MT) {
        tw[k] = m.tw[k];
        const float w = m.window[k];
        const long long ia = f0 * m.hop - N / 2 + k;
        const float xa = padded_sample(y, n, ia, m.pad_mode) * w;
        const float xb = have_b ? padded_sample(y, n, ia + m.hop, m.pad_mode) * w : 0.f;
        bufa[k] = make_float2(xa, xb);
    }
    __syncthreads();

    // ---- Stockham autosort FFT, decimation in frequency: sub-transform length len, stride s, len * s == N ------------------------
    float2* x = bufa;
    float2* z = bufb;
    int len = N, s = 1;
    while (len >= 4) {
        const int n1 = len >> 2;
        for (int i = tid; i < (N >> 2); i += MT) {
            const int p = i / s, q = i - p * s;
            const float2 w1 = tw[p * s], w2 = tw[2 * p * s], w3 = tw[3 * p * s];
            const float2 a = x[q + s * p], bb = x[q + s * (p + n1)], c = x[q + s * (p + 2 * n1)], d = x[q + s * (p + 3 * n1)];
            const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
            const float2 bpd = make_float2(bb.x + d.x, bb.y + d.y);
            const float2 jbmd = make_float2(-(bb.y - d.y), bb.x - d.x);                 // i (b - d)
            z[q + s * (4 * p)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
            z[q + s * (4 * p + 1)] = cmul(w1, make_float2(amc.x - jbmd.x, amc.y - jbmd.y));
            z[q + s * (4 * p + 2)] = cmul(w2, make_float2(apc.x - bpd.x, apc.y - bpd.y));
            z[q + s * (4 * p + 3)] = cmul(w3, make_float2(amc.x + jbmd.x, amc.y + jbmd.y));
        }
        __syncthreads();
        float2* t = x; x = z; z = t;
        len >>= 2; s <<= 2;
    }
    if (len == 2) {
        for (int i = tid; i < (N >> 1); i += MT) {           // p == 0: the twiddle is 1
            const float2 a = x[i], bb = x[i + s];
            z[i] = make_float2(a.x + bb.x, a.y + bb.y);
            z[i + s] = make_float2(a.x - bb.x, a.y - bb.y);
        }
        __syncthreads();
        float2* t = x; x = z; z = t;
    }

    // ---- the two real spectra, magnitudes ---------------------------------------------------------------------------------------
    for (int k = tid; k < m.nbins; k += MT) {
        const float2 zk = x[k], zn = x[(N - k) & (N - 1)];
        const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);      // X_A = (Z[k] + conj Z[N-k]) / 2
        const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);     // X_B = (Z[k] - conj Z[N-k]) / 2i
        mag[k] = sqrtf(ar * ar + ai * ai);
        mag[m.nbins + k] = sqrtf(br * br + bi * bi);
    }
    __syncthreads();

    // ---- mel projection (audio.py:145-150), log10 floor (:108), scaling (preprocess_normalize.py:44) ---------------------------------
    for (int i = tid; i < 2 * m.n_mels; i += MT) {
        const int fr = i / m.n_mels, j = i - fr * m.n_mels;
        if (fr == 1 && !have_b) continue;
        const float* w = m.fw + m.foff[j];
        const float* g = mag + fr * m.nbins + m.flo[j];
        float acc = 0.f;
        for (int k = 0; k < m.fcnt[j]; ++k) acc = fmaf(w[k], g[k], acc);
        float v = log10f(fmaxf(acc, m.floor_));
        if (normalize) v = (v - m.mean[j]) / m.scale[j];
        const long long f = f0 + fr;
        if (transpose) out[((size_t)b * m.n_mels + j) * frames + f] = v;
        else out[((size_t)b * frames + f) * m.n_mels + j] = v;
    }
}

// ---- host: Slaney mel scale and filterbank as librosa.filters.mel builds them (htk = False, norm = "slaney") -----------------------
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double mel) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return mel >= min_log_mel ? min_log_hz * std::exp(logstep * (mel - min_log_mel)) : f_sp * mel;
}

}  // namespace

struct wnv_mel {
    int device = 0;
    wnv_mel_config cfg{};
    MelDev d{};
    std::vector<float> basis;        // dense [n_mels][nbins] copy (wnv_mel_basis)
    void* blob = nullptr;
    float* d_mean = nullptr;         // [2][n_mels]: mean, scale
    bool have_scaler = false;
};

extern thread_local std::string wnv_g_err;

#define MEL_HIP(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) { wnv_g_err = std::string(#expr) + " failed: " + hipGetErrorString(e__); return WNV_ERR_HIP; } \
    } while (0)

extern "C" wnv_status wnv_mel_create(const wnv_mel_config* c, int32_t device, wnv_mel_handle* out) {
    if (!c || !out) { wnv_g_err = "wnv_mel_create: null argument"; return WNV_ERR_INVALID_ARG; }
    const int N = c->fft_size;
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    if (N < 64 || N > 4096 || (1 << logN) != N) { wnv_g_err = "wnv_mel_create: fft_size must be a power of two in [64, 4096]"; return WNV_ERR_UNSUPPORTED; }
    if (c->hop_size <= 0 || c->win_length <= 0 || c->win_length > N || c->num_mels <= 0 || c->sample_rate <= 0) {
        wnv_g_err = "wnv_mel_create: bad hop_size / win_length / num_mels / sample_rate";
        return WNV_ERR_INVALID_ARG;
    }
    const double sr = c->sample_rate, fmax = c->fmax > 0.f ? c->fmax : sr / 2.0, fmin = c->fmin;
    if (fmax > sr / 2.0 || fmin < 0.0 || fmin >= fmax) { wnv_g_err = "wnv_mel_create: need 0 <= fmin < fmax <= sample_rate / 2 (audio.py:153-154)"; return WNV_ERR_INVALID_ARG; }
    if (c->pad_mode != 0 && c->pad_mode != 1) { wnv_g_err = "wnv_mel_create: pad_mode must be 0 (constant) or 1 (reflect)"; return WNV_ERR_INVALID_ARG; }
    wnv_mel* h = new wnv_mel();
    h->device = device;
    h->cfg = *c;
    const int nb = N / 2 + 1, M = c->num_mels;
    // mel_frequencies(n_mels + 2, fmin, fmax): equally spaced on the mel axis
    std::vector<double> mel_f(M + 2);
    const double m_lo = hz_to_mel(fmin), m_hi = hz_to_mel(fmax);
    for (int i = 0; i < M + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * (double)i / (double)(M + 1));
    h->basis.assign((size_t)M * nb, 0.f);
    std::vector<int> flo(M), fcnt(M), foff(M);
    std::vector<float> fw;
    for (int i = 0; i < M; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int lo = nb, hi = -1;
        for (int k = 0; k < nb; ++k) {
            const double f = (double)k * sr / (double)N;                      // np.fft.rfftfreq
            const double lower = (f - mel_f[i]) / fd0, upper = (mel_f[i + 2] - f) / fd1;
            const double w = std::max(0.0, std::min(lower, upper)) * enorm;
            const float wf = (float)w;
            h->basis[(size_t)i * nb + k] = wf;
            if (wf != 0.f) { lo = std::min(lo, k); hi = std::max(hi, k); }
        }
        flo[i] = hi >= lo ? lo : 0;
        fcnt[i] = hi >= lo ? hi - lo + 1 : 0;
        foff[i] = (int)fw.size();
        for (int k = 0; k < fcnt[i]; ++k) fw.push_back(h->basis[(size_t)i * nb + flo[i] + k]);
    }
    // twiddles and the centred periodic Hann window (scipy.signal.get_window("hann", win_length, fftbins=True), pad_center)
    std::vector<float> tw(2 * (size_t)N), win(N, 0.f);
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < N; ++k) {
        tw[2 * k] = (float)std::cos(-2.0 * PI * k / N);
        tw[2 * k + 1] = (float)std::sin(-2.0 * PI * k / N);
    }
    const int wl = c->win_length, lpad = (N - wl) / 2;
    for (int k = 0; k < wl; ++k) win[lpad + k] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * k / wl));
    // one device blob: tw | window | flo | fcnt | foff | fw
    const size_t b_tw = 0, b_win = b_tw + tw.size() * 4, b_flo = b_win + win.size() * 4, b_cnt = b_flo + (size_t)M * 4,
                 b_off = b_cnt + (size_t)M * 4, b_fw = b_off + (size_t)M * 4, total = b_fw + std::max<size_t>(fw.size(), 1) * 4;
    std::vector<char> host(total);
    memcpy(host.data() + b_tw, tw.data(), tw.size() * 4);
    memcpy(host.data() + b_win, win.data(), win.size() * 4);
    memcpy(host.data() + b_flo, flo.data(), (size_t)M * 4);
    memcpy(host.data() + b_cnt, fcnt.data(), (size_t)M * 4);
    memcpy(host.data() + b_off, foff.data(), (size_t)M * 4);
    if (!fw.empty()) memcpy(host.data() + b_fw, fw.data(), fw.size() * 4);
    *out = h;
    h->d.N = N; h->d.hop = c->hop_size; h->d.nbins = nb; h->d.n_mels = M; h->d.pad_mode = c->pad_mode;
    if (device < 0) return WNV_OK;                                // host-only handle: wnv_mel_basis / introspection, no launches
    DeviceGuard guard(device);
    if (!guard.ok) { wnv_g_err = "wnv_mel_create: cannot select the device"; return WNV_ERR_HIP; }
    MEL_HIP(hipMalloc(&h->blob, total));
    MEL_HIP(hipMemcpy(h->blob, host.data(), total, hipMemcpyHostToDevice));
    MEL_HIP(hipMalloc((void**)&h->d_mean, (size_t)2 * M * sizeof(float)));
    char* base = (char*)h->blob;
    h->d.N = N; h->d.logN = logN; h->d.hop = c->hop_size; h->d.nbins = nb; h->d.n_mels = M; h->d.pad_mode = c->pad_mode;
    h->d.floor_ = c->floor > 0.f ? c->floor : 1e-10f;
    h->d.tw = (const float2*)(base + b_tw); h->d.window = (const float*)(base + b_win);
    h->d.flo = (const int*)(base + b_flo); h->d.fcnt = (const int*)(base + b_cnt); h->d.foff = (const int*)(base + b_off);
    h->d.fw = (const float*)(base + b_fw);
    h->d.mean = h->d_mean; h->d.scale = h->d_mean + M;
    return WNV_OK;
}

extern "C" wnv_status wnv_mel_destroy(wnv_mel_handle h) {
    if (!h) return WNV_OK;
    if (h->blob) (void)hipFree(h->blob);
    if (h->d_mean) (void)hipFree(h->d_mean);
    delete h;
    return WNV_OK;
}

extern "C" wnv_status wnv_mel_set_scaler(wnv_mel_handle h, const float* mean, const float* scale) {
    if (!h || !mean || !scale) { wnv_g_err = "wnv_mel_set_scaler: null argument"; return WNV_ERR_INVALID_ARG; }
    if (h->device < 0) { wnv_g_err = "wnv_mel_set_scaler: host-only handle (device < 0)"; return WNV_ERR_INVALID_ARG; }
    const int M = h->cfg.num_mels;
    std::vector<float> v(2 * (size_t)M);
    for (int i = 0; i < M; ++i) {
        if (!(scale[i] != 0.f)) { wnv_g_err = "wnv_mel_set_scaler: zero scale"; return WNV_ERR_INVALID_ARG; }
        v[i] = mean[i];
        v[M + i] = scale[i];
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) { wnv_g_err = "wnv_mel: cannot select the device"; return WNV_ERR_HIP; }
    MEL_HIP(hipMemcpy(h->d_mean, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    h->have_scaler = true;
    return WNV_OK;
}

extern "C" int64_t wnv_mel_frames(const wnv_mel_config* c, int64_t n) {
    if (!c || c->hop_size <= 0 || n < 0) return -1;
    return 1 + n / c->hop_size;                                  // librosa.stft, center=True, even n_fft
}

extern "C" wnv_status wnv_mel_basis(wnv_mel_handle h, float* host_out) {
    if (!h || !host_out) { wnv_g_err = "wnv_mel_basis: null argument"; return WNV_ERR_INVALID_ARG; }
    memcpy(host_out, h->basis.data(), h->basis.size() * sizeof(float));
    return WNV_OK;
}

extern "C" wnv_status wnv_logmel(wnv_mel_handle h, const wnv_logmel_args* a) {
    if (!h || !a || !a->wav || !a->out || a->B <= 0 || a->n <= 0) { wnv_g_err = "wnv_logmel: bad arguments"; return WNV_ERR_INVALID_ARG; }
    if (h->device < 0) { wnv_g_err = "wnv_logmel: host-only handle (created with device < 0); there is no CPU path"; return WNV_ERR_INVALID_ARG; }
    if (a->normalize && !h->have_scaler) { wnv_g_err = "wnv_logmel: normalize without wnv_mel_set_scaler"; return WNV_ERR_NOT_LOADED; }
    const int N = h->d.N;
    if (h->d.pad_mode == 1 && a->n <= N / 2) { wnv_g_err = "wnv_logmel: reflect padding needs more than fft_size / 2 samples"; return WNV_ERR_SHAPE; }
    const long long frames = 1 + a->n / h->d.hop;
    const long long stride = a->wav_stride > 0 ? a->wav_stride : a->n;
    const long long pairs = (frames + 1) / 2;
    if ((long long)a->B * pairs > 0x7fffffffll) { wnv_g_err = "wnv_logmel: too many frames for one launch"; return WNV_ERR_UNSUPPORTED; }
    const size_t lds = (size_t)3 * N * sizeof(float2) + (size_t)2 * h->d.nbins * sizeof(float);
    DeviceGuard guard(h->device);
    if (!guard.ok) { wnv_g_err = "wnv_mel: cannot select the device"; return WNV_ERR_HIP; }
    MEL_HIP(hipFuncSetAttribute((const void*)wnv_logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(wnv_logmel_kernel, dim3((unsigned)(a->B * pairs)), dim3(MT), lds, (hipStream_t)a->stream, h->d, a->wav, (long long)a->n,
                       stride, frames, a->out, a->transpose, a->normalize);
    MEL_HIP(hipGetLastError());
    return WNV_OK;
}
