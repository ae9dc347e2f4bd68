/*
 * wnv.h -- C ABI of the MI355X-native WaveNet-vocoder synthesis engine (libwnv_hip.so).
 *
 * The reference (r9y9/wavenet_vocoder v0.2.0) is pure Python on PyTorch: it has NO native FFI for this
 * path, so there is no existing binding to mirror symbol-for-symbol.  Each entry point below therefore
 * names the reference Python function it replaces (file:line under the reference tree); the ctypes stub a
 * maintainer would add to the reference is shown in INTEGRATION.md and shipped as
 * wavenet_vocoder_amd/_lib.py.
 *
 * Conventions
 *   - plain C types only: pointers + sizes, no torch types.  "device" pointers are HIP device pointers on
 *     the device the handle was created for; "host" pointers are ordinary process memory.
 *   - all tensors are contiguous float32 unless stated.  Shapes are written reference-style.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream)
 *     except where it says "synchronous" -- wnv_generate is the notable case: on the pipelined ring kernel
 *     it returns after the launch has finished unless WNV_GEN_ASYNC is set (see wnv_generate_args.flags).
 *     The caller owns all in/out buffers; the handle owns packed weights, history rings and scratch.
 *   - every function returns a wnv_status; wnv_last_error() gives the message for the calling thread.
 *     The Python host maps WNV_ERR_TRAINING_MODE -> RuntimeError('incremental_forward only supports eval
 *     mode') (reference conv.py:19-20), WNV_ERR_SHAPE -> AssertionError (wavenet.py:276), and so on.
 *   - not re-entrant per handle (the reference's modules are not either: per-module mutable buffers,
 *     SURVEY.md section 8b "Threading"); different handles may be used from different threads.
 *     The persistent kernels (ring, group ring) need all their workgroups resident at once, so inside one
 *     process the library lets them take turns per device: launches of different handles are ordered on the
 *     device (an event behind the previous one) and their host side runs under a per-device mutex.  Other
 *     processes on the same GPU are what the WNV_ERR_TIMEOUT fallback of kernel = 0 is for.
 *   - the library reads NO environment variable (ABI 5): what it does is decided by its arguments alone.  The
 *     measurement knobs of the experiment scripts and the two test / bench hooks live in another build of the
 *     same sources, libwnv_test.so (include/wnv_test.h).
 */
#ifndef WNV_H_
#define WNV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WNV_ABI_VERSION 6
/* The library is built with -fvisibility=hidden: the entry points declared with WNV_API below are its ONLY dynamic symbols
 * (tests/test_host_cpu.py checks `nm -D`: no mangled C++ internals, nothing else defined). */
#if defined(__GNUC__) || defined(__clang__)
#define WNV_API __attribute__((visibility("default")))
#else
#define WNV_API
#endif
#define WNV_MAX_UPSAMPLE_STAGES 8

typedef enum wnv_status {
    WNV_OK = 0,
    WNV_ERR_INVALID_ARG = 1,   /* NULL pointer, negative size, unknown enum ...                         */
    WNV_ERR_NOT_LOADED = 2,    /* a required weight tensor was never loaded                              */
    WNV_ERR_HIP = 3,           /* a HIP runtime call failed (message carries hipGetErrorString)          */
    WNV_ERR_SHAPE = 4,         /* upsampled length != T  (reference: `assert c.size(-1) == T`)           */
    WNV_ERR_UNSUPPORTED = 5,   /* configuration outside what the engine implements                       */
    WNV_ERR_TIMEOUT = 6,       /* an in-kernel bounded spin gave up (persistent pipeline kernels)        */
    WNV_ERR_TRAINING_MODE = 7  /* reserved for hosts that track train/eval mode                          */
} wnv_status;

/* output_distribution: how the head output is turned into the next input.
 * reference wavenet.py:322-335 */
typedef enum wnv_dist {
    WNV_DIST_CATEGORICAL = 0,  /* one-hot input, softmax + OneHotCategorical (scalar_input == 0)          */
    WNV_DIST_LOGISTIC = 1,     /* mixture of logistics, mixture.py:118-156                                */
    WNV_DIST_NORMAL = 2        /* (mixture of) Gaussians, mixture.py:221-270                              */
} wnv_dist;

typedef enum wnv_upsample_act {    /* upsample_activation of upsample.UpsampleNetwork (upsample.py:30,47-49)                     */
    WNV_UPACT_NONE = 0, WNV_UPACT_RELU = 1, WNV_UPACT_LEAKY_RELU = 2, WNV_UPACT_TANH = 3, WNV_UPACT_SIGMOID = 4, WNV_UPACT_ELU = 5
} wnv_upsample_act;

typedef enum wnv_upsample_kind {
    WNV_UPSAMPLE_NONE = 0,     /* upsample_conditional_features=False: c arrives at sample rate           */
    WNV_UPSAMPLE_CONVIN = 1,   /* upsample.ConvInUpsampleNetwork  (upsample.py:69-85)                      */
    WNV_UPSAMPLE_PLAIN = 2     /* upsample.UpsampleNetwork        (upsample.py:29-66)                      */
} wnv_upsample_kind;

/* Mirrors the constructor of wavenet_vocoder.WaveNet (wavenet.py:98-111). */
typedef struct wnv_config {
    int32_t abi_version;            /* must be WNV_ABI_VERSION                                          */
    int32_t out_channels;
    int32_t layers;
    int32_t stacks;                 /* dilation of layer i = 2 ** (i % (layers / stacks))                 */
    int32_t residual_channels;
    int32_t gate_channels;          /* even                                                               */
    int32_t skip_out_channels;
    int32_t kernel_size;
    int32_t cin_channels;           /* <= 0 : no local conditioning                                       */
    int32_t gin_channels;           /* <= 0 : no global conditioning                                      */
    int32_t n_speakers;             /* > 0 with use_speaker_embedding: embed_speakers.weight is expected  */
    int32_t use_speaker_embedding;
    int32_t scalar_input;           /* 1: first_conv is Conv1d1x1(1, R); 0: Conv1d1x1(out_channels, R)    */
    int32_t output_distribution;    /* wnv_dist                                                           */
    int32_t upsample_kind;          /* wnv_upsample_kind                                                  */
    int32_t n_upsample_scales;
    int32_t upsample_scales[WNV_MAX_UPSAMPLE_STAGES];
    int32_t freq_axis_kernel_size;  /* odd, <= 15; taps of the upsampling FIRs along the mel-bin axis (1 in every preset) */
    int32_t cin_pad;
    int32_t upsample_activation;    /* wnv_upsample_act: nn module applied after every upsampling stage (upsample.py:47-49) */
    float   upsample_activation_param; /* LeakyReLU negative_slope / ELU alpha                             */
    int32_t upsample_mode;          /* Stretch2d mode (upsample.py:19-21): 0 "nearest" (every preset; = "area" and       */
                                    /* "nearest-exact" for integer factors), 1 "bilinear", 2 "bicubic"                      */
    int32_t reserved[5];
} wnv_config;

/* One named tensor of a reference state_dict.  `name` is the state_dict key, e.g.
 * "conv_layers.3.conv.weight_v" or "conv_layers.3.conv.weight"; both the weight-normed (weight_g /
 * weight_v) and the fused (weight) layouts are accepted and folded by the engine
 * (w = g * v / ||v||, norm over every dim but 0 == torch remove_weight_norm, wavenet.py:355-361). */
typedef struct wnv_tensor {
    const char*  name;
    const float* data;              /* HOST pointer, contiguous float32                                   */
    int32_t      ndim;
    int64_t      shape[4];
} wnv_tensor;

typedef struct wnv_engine* wnv_handle;

/* ---- life cycle ------------------------------------------------------------------------------ */

/* Replaces WaveNet.__init__ (wavenet.py:98-156).  Synchronous.  device = -1 creates a HOST-ONLY handle: wnv_load_weights
 * validates, folds and packs a checkpoint without touching a device (wnv_bytes_per_step / wnv_macs_per_sample work);
 * wnv_upsample / wnv_generate / wnv_forward refuse it -- there is no CPU path. */
WNV_API wnv_status wnv_create(const wnv_config* cfg, int32_t device, wnv_handle* out);
WNV_API wnv_status wnv_destroy(wnv_handle h);

/* Replaces load_state_dict + make_generation_fast_ (evaluate.py:145-153, wavenet.py:355-361) and
 * conv.Conv1d._get_linearized_weight (conv.py:51-62): takes the tensors of a reference state_dict,
 * folds weight norm, re-lays the weights out for the kernels and uploads them.  Synchronous.
 * Unknown names -> WNV_ERR_INVALID_ARG; a missing required tensor -> WNV_ERR_NOT_LOADED. */
WNV_API wnv_status wnv_load_weights(wnv_handle h, const wnv_tensor* tensors, int32_t n);

/* receptive_field_size (wavenet.py:42-60). */
WNV_API int64_t wnv_receptive_field(int32_t layers, int32_t stacks, int32_t kernel_size);
/* Number of float32 noise values consumed per utterance per step for this configuration
 * (tape layout: wavenet_vocoder_amd/noise.py; draw order of the reference: SURVEY.md A.3). */
WNV_API int32_t wnv_noise_width(const wnv_config* cfg);
/* Number of output samples the upsampling network produces for Tc_in input frames, or -1. */
WNV_API int64_t wnv_upsampled_length(const wnv_config* cfg, int64_t Tc_in);

/* ---- prologue: local-conditioning upsampling ----------------------------------------------------
 * Replaces ConvInUpsampleNetwork.forward / UpsampleNetwork.forward (upsample.py:83-85, :51-66) and the
 * transpose at wavenet.py:277-278.
 *   c      device (B, cin, Tc_in)                       [Tc_in includes the 2*cin_pad context frames]
 *   c_up   device (B, T, cin)   TIME-MAJOR              [T = wnv_upsampled_length(cfg, Tc_in)]
 * Returns WNV_ERR_SHAPE when T != T_expected (T_expected < 0 skips the check). */
WNV_API wnv_status wnv_upsample(wnv_handle h, const float* c, int32_t B, int64_t Tc_in,
                        float* c_up, int64_t T_expected, void* stream);

/* ---- the hot loop --------------------------------------------------------------------------------
 * Replaces WaveNet.incremental_forward (wavenet.py:215-343): clear_buffer, the T-step autoregressive
 * loop (first_conv, L x ResidualConv1dGLU.incremental_forward, skip sum, head, sampling) and the final
 * stack/transposes, as ONE launch with no host round trip between samples. */
typedef struct wnv_generate_args {
    int32_t B;                 /* utterances in this call                                               */
    int64_t T;                 /* steps to generate (already max(T, Tt), wavenet.py:255-258)             */
    const float* c_up;         /* device (B, T, cin) time-major, or NULL                                 */
    const float* g;            /* device (B, gin) global features, or NULL                               */
    const int64_t* g_ids;      /* device (B) speaker ids -> embed_speakers (wavenet.py:264-268), or NULL */
    const float* initial;      /* device (B, Cin) first input, Cin = 1 or out_channels; NULL = zeros /   */
                               /*   one-hot index 127 (wavenet.py:281-289)                               */
    const float* teacher;      /* device (B, Tt, Cin) teacher-forcing inputs (test_inputs), or NULL      */
    int64_t Tt;
    const float* noise;        /* device (T, B, wnv_noise_width) tape, or NULL = in-kernel Philox(seed)  */
    uint64_t seed;             /* THE IN-KERNEL STREAM (noise == NULL): value j of (utterance b, step t) is the first output word x of */
                               /*   Philox4x32-10(counter = (t lo, t hi, b, j), key = (seed lo, seed hi)) mapped to the float32         */
                               /*   u = ((x >> 9) + 0.5) / 2^23 -- exact, 0 < u < 1 -- and by the tape's position j (wnv_noise_width's   */
                               /*   layout) to U(1e-5, 1 - 1e-5) = fma(u, 1 - 2e-5, 1e-5) | N(0, 1) = sqrt(-2 ln u) cos(2 pi v), v from   */
                               /*   the second word | Exp(1) = -ln u > 0.  Packed slots: b = seg_uid, t = the step within the utterance. */
                               /*   A launch is a deterministic function of (weights, inputs, seed), and an utterance's waveform does   */
                               /*   not depend on the batch it ran in or on how a job was packed (one-hot models: one pick form,        */
                               /*   argmax logit_k - log e_k, in every instantiation of the ring kernel; ABI 6 changed the map from     */
                               /*   ((x >> 8) + 0.5) / 2^24, which rounded to 1.0 once in 2^24 draws).  tests/_philox.py restates the   */
                               /*   stream in numpy; tests/test_gpu_inkernel_noise.py checks every sample against it,                   */
                               /*   tests/test_gpu_seed_determinism.py the independence of the batch size.                              */
    int32_t softmax;           /* categorical only: apply softmax (wavenet.py:332)                       */
    int32_t quantize;          /* categorical only: sample a one-hot (wavenet.py:333-335)                */
    float* out;                /* device (B, C, T): C = 1 scalar samples | out_channels one-hot/probs;   */
                               /*   NULL allowed for a one-hot model with quantize = 1 and index_out     */
                               /*   given, in a packed-slot launch or with kernel = 1 (ABI 5): the       */
                               /*   sampled classes only                                                  */
    float* params_out;         /* optional device (B, out_channels, T): head output before sampling      */
    int32_t* index_out;        /* optional device (B, T): sampled class (categorical + quantize)         */
    int32_t kernel;            /* 0 = auto, 1 = generic single-workgroup kernel, 2 = pipelined ring,     */
                               /* 3 = group ring for wide models (8 workgroups per layer)                */
    int32_t flags;             /* WNV_GEN_* bits                                                         */
    void* stream;
    const uint32_t* noise_ready; /* ABI 3, optional: a STREAMED tape.  Device-visible address of a counter in coherent host    */
                               /* memory (wnv_pinned_alloc): steps [0, *noise_ready) of `noise` are valid, the caller keeps   */
                               /* filling the tape and advancing the counter while the kernel runs; the kernel waits (bounded) */
                               /* for every step it is about to read.  Needs kernel == 2 and WNV_GEN_ASYNC.  NULL: the whole   */
                               /* tape is valid at the call.                                                                   */
    const int32_t* seg_start;  /* ABI 4, optional: PACKED SLOTS (continuous batching).  The B rows of this call are not utterances    */
    const int32_t* seg_uid;    /* but SLOTS that run several utterances back to back: device (B, T) each; seg_start[b][t] = the step  */
                               /* at which the utterance occupying slot b at step t began, seg_uid[b][t] = its id in the job.  At a   */
                               /* boundary (seg_start[b][t] == t) the utterance starts as incremental_forward starts one: history     */
                               /* before its first step reads as zeros (conv.py:34-36), the first input is zeros / one-hot 127          */
                               /* (wavenet.py:281-289); the in-kernel noise stream is addressed with (seg_uid, t - seg_start), so a   */
                               /* waveform does not depend on how the job was packed.  c_up is the slots' concatenated conditioning.  */
                               /* Ring kernel only, in-kernel noise (noise, teacher, initial NULL); otherwise WNV_ERR_UNSUPPORTED /    */
                               /* WNV_ERR_INVALID_ARG.  NULL: one utterance per row.                                                  */
    const int32_t* seg_gid;    /* ABI 5, packed slots of a model WITH global conditioning (BASELINE configs[4]: speaker embedding,    */
    int32_t n_g;               /* wavenet.py:262-269): g / g_ids then have n_g rows -- one per speaker or per utterance of the job,   */
                               /* not per slot -- and seg_gid[b][t] (device (B, T)) is the row of the utterance occupying slot b at   */
                               /* step t: the hoisted bias table conv.bias + conv1x1g(g) (modules.py:146-150) has n_g rows and the    */
                               /* tap workgroups pick the row per slot and step.  NULL / 0 otherwise.  The maps live in device memory */
                               /* and are NOT range-checked (no host copy is made): the caller guarantees 0 <= seg_gid < n_g,         */
                               /* 0 <= seg_start[b][t] <= t, and seg_start / seg_uid constant over a segment.                         */
} wnv_generate_args;

/* The pipelined ring kernel is a persistent launch whose workgroups wait for each other; every wait is bounded and a
 * wait that gives up makes the whole launch drain with a status code, which only the host can turn into WNV_ERR_TIMEOUT.
 * Default: wnv_generate synchronises `stream` after that launch and reports the status itself (and, with kernel == 0,
 * re-runs the call on the generic kernel -- see below).  With WNV_GEN_ASYNC the call returns right after the launch;
 * the status is reported by wnv_wait(), or by the next wnv_generate / wnv_reset on the handle.  The generic kernel has no
 * cross-workgroup waits and is asynchronous either way. */
#define WNV_GEN_ASYNC 1

/* kernel == 0 ("auto") picks the ring kernel when it covers the configuration (wnv_ring_why_not) AND the device can keep
 * its grid resident (occupancy query x CU count, one ring per XCD as measured by a placement census at load time);
 * otherwise, or when a ring launch ends in WNV_ERR_TIMEOUT (CUs masked or taken by another process: the workgroups
 * were not co-resident), the call is served by the generic kernel and the reason goes to stderr.  "Cannot run here"
 * (WNV_ERR_UNSUPPORTED) keeps the handle on the generic kernel; after a time-out the persistent kernel is tried again
 * once 2 further calls have been served by the generic kernel (4, 8, ... 32 after consecutive time-outs; wnv_reset() makes
 * the next call try at once).  An explicit kernel == 2 reports the error instead.  WNV_GEN_ASYNC needs kernel == 2 and at
 * most 64 utterances (larger batches are run as several launches of 64). */
WNV_API wnv_status wnv_generate(wnv_handle h, const wnv_generate_args* args);
/* Waits for the handle's last WNV_GEN_ASYNC launch and returns its status (WNV_OK when nothing is pending).  Synchronous. */
WNV_API wnv_status wnv_wait(wnv_handle h);
/* Which kernel served the last wnv_generate of this handle: 1 generic, 2 ring, 3 group ring (wide models), 0 none yet. */
WNV_API int32_t wnv_last_kernel(wnv_handle h);

/* WaveNet.clear_buffer (wavenet.py:345-353): the engine re-zeroes its history at the start of every
 * wnv_generate (as incremental_forward does at :241), so this only releases scratch. */
WNV_API wnv_status wnv_reset(wnv_handle h);

/* ---- layer-level drop-ins ------------------------------------------------------------------------
 * conv.Conv1d.incremental_forward (conv.py:17-46): one step of a queue-cached dilated convolution. */
typedef struct wnv_qconv* wnv_qconv_handle;
WNV_API wnv_status wnv_qconv_create(int32_t cin, int32_t cout, int32_t kernel_size, int32_t dilation,
                            int32_t device, wnv_qconv_handle* out);
/* weight host (cout, cin, kernel_size) [nn.Conv1d layout]; bias host (cout) or NULL. */
WNV_API wnv_status wnv_qconv_set_weights(wnv_qconv_handle q, const float* weight, const float* bias);
/* x device (B, cin) -> y device (B, cout).  History is created zeroed on the first step after a reset
 * (conv.py:34-36) for that B. */
WNV_API wnv_status wnv_qconv_step(wnv_qconv_handle q, const float* x, float* y, int32_t B, void* stream);
WNV_API wnv_status wnv_qconv_reset(wnv_qconv_handle q);          /* conv.py:48-49 clear_buffer */
WNV_API wnv_status wnv_qconv_destroy(wnv_qconv_handle q);

/* ResidualConv1dGLU.incremental_forward (modules.py:112-163): one gated residual layer step. */
typedef struct wnv_glu* wnv_glu_handle;
typedef struct wnv_glu_config {
    int32_t residual_channels, gate_channels, kernel_size, skip_out_channels;
    int32_t cin_channels, gin_channels, dilation, bias;
} wnv_glu_config;
WNV_API wnv_status wnv_glu_create(const wnv_glu_config* cfg, int32_t device, wnv_glu_handle* out);
/* names as in ResidualConv1dGLU.state_dict(): "conv.weight[_g|_v]", "conv.bias", "conv1x1c.weight..",
 * "conv1x1g.weight..", "conv1x1_out.*", "conv1x1_skip.*". */
WNV_API wnv_status wnv_glu_load_weights(wnv_glu_handle g, const wnv_tensor* tensors, int32_t n);
/* x (B,R), c (B,cin)|NULL, gcond (B,gin)|NULL  ->  x_out (B,R), s_out (B,K); all device. */
WNV_API wnv_status wnv_glu_step(wnv_glu_handle g, const float* x, const float* c, const float* gcond,
                        float* x_out, float* s_out, int32_t B, void* stream);
WNV_API wnv_status wnv_glu_reset(wnv_glu_handle g);              /* modules.py:165-169 clear_buffer */
WNV_API wnv_status wnv_glu_destroy(wnv_glu_handle g);

/* ---- teacher-forced batch evaluation ------------------------------------------
 * Replaces WaveNet.forward (wavenet.py:164-213) after the upsampling step: first_conv, the L gated layers over all T at
 * once (f32 MFMA GEMMs), skip sum, head, optional softmax.  Needs residual_channels == 128, gate_channels == 256,
 * skip_out_channels % 128 == 0, out_channels <= 256 (else WNV_ERR_UNSUPPORTED: the Python host then uses torch ops). */
typedef struct wnv_forward_args {
    int32_t B;
    int64_t T;                 /* <= 2^23 samples per utterance (WNV_ERR_INVALID_ARG beyond)                */
    const float* x;            /* device (B, Cin, T): Cin = 1 (scalar input) or out_channels (one-hot)      */
    const float* c_up;         /* device (B, T, cin) time-major (wnv_upsample's output), or NULL            */
    const float* g;            /* device (B, gin) or NULL                                                   */
    const int64_t* g_ids;      /* device (B) speaker ids or NULL                                            */
    float* out;                /* device (B, out_channels, T)                                               */
    int32_t softmax;           /* F.softmax(x, dim=1) at the end (wavenet.py:211)                           */
    void* stream;
} wnv_forward_args;
WNV_API wnv_status wnv_forward(wnv_handle h, const wnv_forward_args* args);

/* ---- post-chain ------------------------------------------------------------
 * Replaces the tail of synthesis.batch_wavegen (synthesis.py:66-84) and the clip / int16 conversion of
 * evaluate.py:238, :43-48 on the device: argmax + inv_mulaw_quantize | inv_mulaw | raw, then the optional
 * audio.inv_preemphasis (audio.py:57-58), / global_gain_scale, clip, int16.  inv_mulaw* / inv_preemphasis are
 * nnmnkwii's published definitions (the reference's un-vendored dependency, setup.py:23). */
typedef struct wnv_post_args {
    int32_t B, C;              /* y is (B, C, T): C = 1 for scalar input types, quantize_channels for one-hot; C = 1 with */
                               /* "mulaw-quantize" (ABI 5): y holds the sampled CLASSES as floats (index_out), no argmax   */
    int64_t T;
    const float* y;            /* device: wnv_generate's `out`                                            */
    int32_t input_type;        /* 0 "raw", 1 "mulaw", 2 "mulaw-quantize"  (hparams.input_type)             */
    int32_t mu;                /* hparams.quantize_channels - 1 (unused for "raw")                        */
    float preemphasis;         /* > 0: postprocess = inv_preemphasis with this coefficient; 0: none       */
    float gain_scale;          /* > 0: divide by hparams.global_gain_scale                                */
    int32_t clip;              /* np.clip(gen, -1, 1)  (evaluate.py:238)                                  */
    float* wav;                /* device (B, T) float32                                                   */
    int16_t* pcm;              /* optional device (B, T): to_int16 (evaluate.py:43-48), needs clip        */
    void* stream;
} wnv_post_args;
WNV_API wnv_status wnv_postprocess(int32_t device, const wnv_post_args* args);

/* ---- mel front end ------------------------------------------------------------
 * Replaces audio.logmelspectrogram (audio.py:101-109: librosa.stft -> |D| -> librosa.filters.mel -> log10(max(., 1e-10)))
 * and the per-bin StandardScaler.transform of preprocess_normalize.py:44, i.e. what turns a waveform into the
 * "*-feats.npy" rows the synthesis path takes as `c` (datasets/wavallin.py:62).  librosa / scikit-learn are the
 * reference's un-vendored dependencies (setup.py:22-26); the engine implements their published definitions
 * (periodic Hann window centred in the frame, center=True framing with reflect / zero padding, Slaney mel scale with
 * unit-area triangular filters).  Fields mirror hparams.py:32-44. */
typedef struct wnv_mel_config {
    int32_t sample_rate;       /* hparams.sample_rate                                                       */
    int32_t fft_size;          /* hparams.fft_size: a power of two in [64, 4096]                            */
    int32_t hop_size;          /* audio.get_hop_size()                                                      */
    int32_t win_length;        /* audio.get_win_length() <= fft_size                                        */
    int32_t num_mels;          /* hparams.num_mels                                                          */
    float   fmin, fmax;        /* hparams.fmin / fmax; fmax <= 0 means sample_rate / 2                      */
    int32_t pad_mode;          /* 0 "constant" (zeros), 1 "reflect" (logmelspectrogram's default)           */
    float   floor;             /* the 1e-10 of audio.py:108; <= 0 selects 1e-10                             */
    int32_t reserved[4];
} wnv_mel_config;
typedef struct wnv_mel* wnv_mel_handle;
/* device < 0 creates a host-only handle (filterbank introspection via wnv_mel_basis; wnv_logmel refuses it). */
WNV_API wnv_status wnv_mel_create(const wnv_mel_config* cfg, int32_t device, wnv_mel_handle* out);   /* synchronous */
WNV_API wnv_status wnv_mel_destroy(wnv_mel_handle h);
/* StandardScaler.mean_ / scale_ (host, num_mels each): out = (logmel - mean) / scale.  Synchronous. */
WNV_API wnv_status wnv_mel_set_scaler(wnv_mel_handle h, const float* mean, const float* scale);
/* Number of frames librosa.stft(center=True) yields for n samples: 1 + n / hop_size; -1 on bad arguments. */
WNV_API int64_t wnv_mel_frames(const wnv_mel_config* cfg, int64_t n);
/* The filterbank the engine uses, host (num_mels, fft_size / 2 + 1) float32 = librosa.filters.mel(...). */
WNV_API wnv_status wnv_mel_basis(wnv_mel_handle h, float* host_out);
typedef struct wnv_logmel_args {
    int32_t B;
    int64_t n;                 /* samples per utterance                                                     */
    int64_t wav_stride;        /* floats between utterances (0 = n)                                         */
    const float* wav;          /* device (B, n)                                                             */
    float* out;                /* device (B, frames, num_mels) [feats layout] or (B, num_mels, frames)      */
    int32_t transpose;         /* 1: (B, num_mels, frames) = what logmelspectrogram itself returns          */
    int32_t normalize;         /* 1: apply the scaler (needs wnv_mel_set_scaler)                            */
    void* stream;
} wnv_logmel_args;
WNV_API wnv_status wnv_logmel(wnv_mel_handle h, const wnv_logmel_args* args);

/* ---- host helpers for a streamed replay tape (ABI 3) -------------------------------------------------
 * The reference draws its sampling noise from torch's CPU generator inside the loop (wavenet.py:334-335: OneHotCategorical ->
 * torch.multinomial's exponential race, B x 256 values per step); a replay of that stream is 49 M values for the benchmark batch
 * of a mu-law model -- longer to draw than the kernel takes to run.  These helpers let the host draw it WHILE the kernel runs. */
/* Coherent, device-mapped host memory (hipHostMalloc, coherent + mapped + portable: valid on every device of the process, whichever
 * is current on the calling thread): *host_ptr for the CPU, *device_ptr for wnv_generate_args. */
WNV_API wnv_status wnv_pinned_alloc(size_t bytes, void** host_ptr, void** device_ptr);
WNV_API wnv_status wnv_pinned_free(void* host_ptr);
/* out[i] = (float)(-log1p(-u[i])), u in [0, 1) double: the transform ATen's CPU exponential_ applies to its uniform draw
 * (TransformationHelper.h: -1 / lambda * log1p(-u), lambda = 1, computed in double, libm's log1p), on `threads` host threads.
 * With u = torch.empty(n, dtype=float64).uniform_(0, 1) this reproduces torch.empty(n).exponential_(1) bit for bit and leaves the
 * generator in the same state (tests/test_host_cpu.py).  Pure host code. */
WNV_API wnv_status wnv_exponential_from_uniform(const double* u, float* out, int64_t n, int32_t threads);
/* n draws of torch's CPU generator as `torch.empty(n, dtype=float64).uniform_(0, 1, generator=g)` makes them, natively: `state` is the
 * blob g.get_state() returns (CPUGeneratorImpl's legacy layout: seed u64 @0, left i32 @8, seeded i32 @12, next u64 @16, the 624 words of
 * at::mt19937 as u64 @24) and is ADVANCED in place -- g.set_state(state) afterwards leaves g where the torch call would have.  Two
 * consecutive 32-bit outputs make one draw, the first one high: (x & (2^53 - 1)) * 2^-53 (ATen/core/DistributionsHelper.h).  With
 * wnv_exponential_from_uniform this is the reference's exponential race (wavenet.py:334-335) at ~1.5 ns per value instead of the 5-10 ns
 * torch's element-by-element walk costs inside a busy process; the Python host checks it against torch once per process and falls back
 * to uniform_ when the numbers or the state differ (added within ABI 4, round 4; pure host code). */
WNV_API wnv_status wnv_mt19937_uniform53(void* state, int64_t state_bytes, double* out, int64_t n);

/* ---- misc --------------------------------------------------------------------------------------- */
WNV_API const char* wnv_last_error(void);
WNV_API int32_t wnv_abi_version(void);
/* Introspection used by bench.py's roofline: algorithmic bytes moved per time step of a B-utterance
 * group (weights once + ring taps + conditioning row + output) and MACs per sample. */
WNV_API int64_t wnv_bytes_per_step(wnv_handle h, int32_t B);
WNV_API int64_t wnv_macs_per_sample(wnv_handle h);
/* Which configurations a sample-loop kernel covers, from the configuration alone (pure host code): "supported", or the reason
 * wnv_generate would report with WNV_ERR_UNSUPPORTED.  kernel: 1 generic (everything the reference can express), 2 pipelined ring,
 * 3 group ring (wide models).  ABI 5. */
WNV_API const char* wnv_kernel_coverage(const wnv_config* cfg, int32_t kernel, int32_t B);
/* (ABI 5: the LDS-peak microbenchmark and the time-out injection hook left the product ABI: include/wnv_test.h, libwnv_test.so.) */

#ifdef __cplusplus
}
#endif
#endif /* WNV_H_ */
