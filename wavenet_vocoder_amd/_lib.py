"""ctypes binding of ``libwnv_hip.so`` (the C ABI declared in ``include/wnv.h``).

This is the whole "FFI stub" a maintainer of the reference would need (INTEGRATION.md shows it grafted
into ``wavenet_vocoder/wavenet.py``).  There is deliberately no fallback: if the shared library is
missing or was built for another ABI the import of the engine raises, it never silently runs on CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

__all__ = ["lib", "WnvError", "check", "Config", "Tensor", "GenerateArgs", "GluConfig", "PostArgs", "ForwardArgs", "MelConfig", "LogmelArgs", "LIB_PATH",
           "WNV_ABI_VERSION", "DIST", "UPSAMPLE"]

WNV_ABI_VERSION = 6
WNV_MAX_UPSAMPLE_STAGES = 8
WNV_GEN_ASYNC = 1
# WNV_LIB selects another build of the same sources: the TEST library (libwnv_test.so: knobs + the hooks of include/wnv_test.h) or a
# debug / trace build (python -m wavenet_vocoder_amd.build --out ... --flags ...)
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WNV_LIB") or os.path.join(_HERE, "libwnv_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "libwnv_test.so")

DIST = {"categorical": 0, "Logistic": 1, "Normal": 2}
UPSAMPLE = {None: 0, "none": 0, "ConvInUpsampleNetwork": 1, "UpsampleNetwork": 2}
# upsample_activation (upsample.py:30,47-49: getattr(nn, name)(**params)) -> (wnv_upsample_act, the keyword of its one parameter)
UPSAMPLE_ACT = {"none": (0, None), "ReLU": (1, None), "LeakyReLU": (2, "negative_slope"), "Tanh": (3, None), "Sigmoid": (4, None), "ELU": (5, "alpha")}
UPSAMPLE_ACT_DEFAULT = {"negative_slope": 0.01, "alpha": 1.0}
# Stretch2d mode (upsample.py:19-21: F.interpolate(x, scale_factor=(1, s), mode=mode)) -> wnv_config.upsample_mode
# ("area" -- adaptive average pooling -- and "nearest-exact" pick the same single sample as "nearest" when the factor is an integer, which
#  Stretch2d's factors are; "linear" / "trilinear" are for 3-D / 5-D inputs: F.interpolate itself refuses them for Stretch2d's 4-D map)
UPSAMPLE_MODE = {"nearest": 0, "area": 0, "nearest-exact": 0, "bilinear": 1, "bicubic": 2}

# status codes -> Python exceptions (include/wnv.h "Conventions")
_STATUS_EXC = {
    1: ValueError,          # WNV_ERR_INVALID_ARG
    2: RuntimeError,        # WNV_ERR_NOT_LOADED
    3: RuntimeError,        # WNV_ERR_HIP
    4: AssertionError,      # WNV_ERR_SHAPE         (reference: assert c.size(-1) == T, wavenet.py:276)
    5: NotImplementedError, # WNV_ERR_UNSUPPORTED
    6: TimeoutError,        # WNV_ERR_TIMEOUT
    7: RuntimeError,        # WNV_ERR_TRAINING_MODE (reference: conv.py:19-20)
}


class WnvError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("out_channels", C.c_int32), ("layers", C.c_int32), ("stacks", C.c_int32),
        ("residual_channels", C.c_int32), ("gate_channels", C.c_int32), ("skip_out_channels", C.c_int32),
        ("kernel_size", C.c_int32), ("cin_channels", C.c_int32), ("gin_channels", C.c_int32),
        ("n_speakers", C.c_int32), ("use_speaker_embedding", C.c_int32), ("scalar_input", C.c_int32),
        ("output_distribution", C.c_int32), ("upsample_kind", C.c_int32), ("n_upsample_scales", C.c_int32),
        ("upsample_scales", C.c_int32 * WNV_MAX_UPSAMPLE_STAGES), ("freq_axis_kernel_size", C.c_int32),
        ("cin_pad", C.c_int32), ("upsample_activation", C.c_int32), ("upsample_activation_param", C.c_float),
        ("upsample_mode", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class GenerateArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int64), ("c_up", C.c_void_p), ("g", C.c_void_p), ("g_ids", C.c_void_p),
        ("initial", C.c_void_p), ("teacher", C.c_void_p), ("Tt", C.c_int64), ("noise", C.c_void_p),
        ("seed", C.c_uint64), ("softmax", C.c_int32), ("quantize", C.c_int32), ("out", C.c_void_p),
        ("params_out", C.c_void_p), ("index_out", C.c_void_p), ("kernel", C.c_int32), ("flags", C.c_int32),
        ("stream", C.c_void_p), ("noise_ready", C.c_void_p), ("seg_start", C.c_void_p), ("seg_uid", C.c_void_p),
        ("seg_gid", C.c_void_p), ("n_g", C.c_int32),
    ]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int64), ("x", C.c_void_p), ("c_up", C.c_void_p), ("g", C.c_void_p),
        ("g_ids", C.c_void_p), ("out", C.c_void_p), ("softmax", C.c_int32), ("stream", C.c_void_p),
    ]


class PostArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("C", C.c_int32), ("T", C.c_int64), ("y", C.c_void_p), ("input_type", C.c_int32),
        ("mu", C.c_int32), ("preemphasis", C.c_float), ("gain_scale", C.c_float), ("clip", C.c_int32),
        ("wav", C.c_void_p), ("pcm", C.c_void_p), ("stream", C.c_void_p),
    ]


class MelConfig(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32), ("fft_size", C.c_int32), ("hop_size", C.c_int32), ("win_length", C.c_int32),
        ("num_mels", C.c_int32), ("fmin", C.c_float), ("fmax", C.c_float), ("pad_mode", C.c_int32), ("floor", C.c_float),
        ("reserved", C.c_int32 * 4),
    ]


class LogmelArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("n", C.c_int64), ("wav_stride", C.c_int64), ("wav", C.c_void_p), ("out", C.c_void_p),
        ("transpose", C.c_int32), ("normalize", C.c_int32), ("stream", C.c_void_p),
    ]


class GluConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("residual_channels", "gate_channels", "kernel_size",
                                         "skip_out_channels", "cin_channels", "gin_channels", "dilation",
                                         "bias")]


_PROTOS = {
    # name: (restype, argtypes)
    "wnv_abi_version": (C.c_int32, []),
    "wnv_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "wnv_pinned_free": (C.c_int, [C.c_void_p]),
    "wnv_exponential_from_uniform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "wnv_mt19937_uniform53": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "wnv_last_error": (C.c_char_p, []),
    "wnv_create": (C.c_int, [C.POINTER(Config), C.c_int32, C.POINTER(C.c_void_p)]),
    "wnv_destroy": (C.c_int, [C.c_void_p]),
    "wnv_load_weights": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_int32]),
    "wnv_receptive_field": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "wnv_noise_width": (C.c_int32, [C.POINTER(Config)]),
    "wnv_upsampled_length": (C.c_int64, [C.POINTER(Config), C.c_int64]),
    "wnv_upsample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "wnv_generate": (C.c_int, [C.c_void_p, C.POINTER(GenerateArgs)]),
    "wnv_reset": (C.c_int, [C.c_void_p]),
    "wnv_wait": (C.c_int, [C.c_void_p]),
    "wnv_last_kernel": (C.c_int32, [C.c_void_p]),
    "wnv_kernel_coverage": (C.c_char_p, [C.POINTER(Config), C.c_int32, C.c_int32]),
    "wnv_qconv_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "wnv_qconv_set_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "wnv_qconv_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "wnv_qconv_reset": (C.c_int, [C.c_void_p]),
    "wnv_qconv_destroy": (C.c_int, [C.c_void_p]),
    "wnv_glu_create": (C.c_int, [C.POINTER(GluConfig), C.c_int32, C.POINTER(C.c_void_p)]),
    "wnv_glu_load_weights": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_int32]),
    "wnv_glu_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "wnv_glu_reset": (C.c_int, [C.c_void_p]),
    "wnv_glu_destroy": (C.c_int, [C.c_void_p]),
    "wnv_postprocess": (C.c_int, [C.c_int32, C.POINTER(PostArgs)]),
    "wnv_forward": (C.c_int, [C.c_void_p, C.POINTER(ForwardArgs)]),
    "wnv_mel_create": (C.c_int, [C.POINTER(MelConfig), C.c_int32, C.POINTER(C.c_void_p)]),
    "wnv_mel_destroy": (C.c_int, [C.c_void_p]),
    "wnv_mel_set_scaler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "wnv_mel_frames": (C.c_int64, [C.POINTER(MelConfig), C.c_int64]),
    "wnv_mel_basis": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wnv_logmel": (C.c_int, [C.c_void_p, C.POINTER(LogmelArgs)]),
    "wnv_bytes_per_step": (C.c_int64, [C.c_void_p, C.c_int32]),
    "wnv_macs_per_sample": (C.c_int64, [C.c_void_p]),
}

# include/wnv_test.h: exported by the TEST library only (a product library that exports them is a build error: tests/test_host_cpu.py)
_TEST_PROTOS = {
    "wnv_debug_inject_timeouts": (C.c_int, [C.c_void_p, C.c_int32]),
    "wnv_measure_lds_read_peak": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOS))
TEST_HOOK_SYMBOLS = tuple(sorted(_TEST_PROTOS))

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WnvError(
            f"{LIB_PATH} is missing: the HIP engine has not been built.  Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m wavenet_vocoder_amd.build`) "
            f"at the repo root.  There is no CPU fallback for the synthesis path.")
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:  # pragma: no cover
            raise WnvError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _TEST_PROTOS.items():           # present in a test / knob build only
        fn = getattr(handle, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if handle.wnv_abi_version() != WNV_ABI_VERSION:
        raise WnvError(f"{LIB_PATH} has ABI {handle.wnv_abi_version()}, this package needs {WNV_ABI_VERSION}")
    _lib = handle
    return handle


def has_test_hooks() -> bool:
    """True when the loaded library is a test / knob build (WNV_LIB=.../libwnv_test.so)."""
    return hasattr(lib(), "wnv_debug_inject_timeouts")


_bench_tool: Optional[C.CDLL] = None


def measure_lds_read_peak(device: int):
    """(GB/s, CUs) from the microbenchmark of include/wnv_test.h.  A bench tool, not product: it lives in libwnv_test.so, which is
    loaded here (RTLD_LOCAL, next to the product library) for this one call when the process runs on the product library."""
    global _bench_tool
    h = lib() if has_test_hooks() else _bench_tool
    if h is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise WnvError(f"{TEST_LIB_PATH} is missing (python -m wavenet_vocoder_amd.build builds it next to the product library)")
        h = _bench_tool = C.CDLL(TEST_LIB_PATH, mode=C.RTLD_LOCAL)
        h.wnv_measure_lds_read_peak.restype, h.wnv_measure_lds_read_peak.argtypes = _TEST_PROTOS["wnv_measure_lds_read_peak"]
        h.wnv_last_error.restype = C.c_char_p
    gbs, ncu = C.c_double(0.0), C.c_int32(0)
    st = h.wnv_measure_lds_read_peak(int(device), C.byref(gbs), C.byref(ncu))
    if st != 0:
        msg = h.wnv_last_error()
        raise WnvError(msg.decode("utf-8", "replace") if msg else f"wnv status {st}")
    return gbs.value, int(ncu.value)


def check(status: int) -> None:
    if status == 0:
        return
    msg = lib().wnv_last_error()
    msg = msg.decode("utf-8", "replace") if msg else f"wnv status {status}"
    raise _STATUS_EXC.get(status, WnvError)(msg)
