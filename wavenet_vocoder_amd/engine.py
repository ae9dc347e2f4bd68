"""Thin object wrappers over the C ABI handles (``include/wnv.h``): ``Engine`` (whole network),
``QueueConv`` (conv.Conv1d.incremental_forward) and ``GluLayer`` (ResidualConv1dGLU.incremental_forward).

torch is used here for exactly three things: owning device memory (``tensor.data_ptr()``), naming the
HIP stream to launch on, and moving weights to the host once.  All arithmetic happens in libwnv_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import Config, GenerateArgs, GluConfig, Tensor, check

__all__ = ["Engine", "QueueConv", "GluLayer", "make_config", "require_gpu_tensor", "check_checkpoint"]


def require_gpu_tensor(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} lives on {t.device}: the MI355X engine only runs on a HIP device (model.to('cuda')); "
            f"there is no CPU fallback for the incremental path")


def make_config(*, out_channels, layers, stacks, residual_channels, gate_channels, skip_out_channels,
                kernel_size, cin_channels, gin_channels, n_speakers, use_speaker_embedding, scalar_input,
                output_distribution, upsample_net: Optional[str], upsample_scales: Sequence[int],
                freq_axis_kernel_size: int, cin_pad: int, upsample_activation: str = "none",
                upsample_activation_params: Optional[dict] = None, upsample_mode: str = "nearest") -> Config:
    cfg = Config()
    cfg.abi_version = _lib.WNV_ABI_VERSION
    cfg.out_channels = out_channels
    cfg.layers = layers
    cfg.stacks = stacks
    cfg.residual_channels = residual_channels
    cfg.gate_channels = gate_channels
    cfg.skip_out_channels = skip_out_channels
    cfg.kernel_size = kernel_size
    cfg.cin_channels = cin_channels
    cfg.gin_channels = gin_channels
    cfg.n_speakers = int(n_speakers or 0)
    cfg.use_speaker_embedding = int(bool(use_speaker_embedding))
    cfg.scalar_input = int(bool(scalar_input))
    if scalar_input:
        if output_distribution not in ("Logistic", "Normal"):
            raise AssertionError(f"unknown output_distribution {output_distribution!r}")   # wavenet.py:330
        cfg.output_distribution = _lib.DIST[output_distribution]
    else:
        cfg.output_distribution = _lib.DIST["categorical"]
    if upsample_net not in _lib.UPSAMPLE:
        raise NotImplementedError(f"upsample_net {upsample_net!r}")
    cfg.upsample_kind = _lib.UPSAMPLE[upsample_net]
    scales = list(upsample_scales or [])
    if len(scales) > _lib.WNV_MAX_UPSAMPLE_STAGES:
        raise NotImplementedError("too many upsample stages")
    cfg.n_upsample_scales = len(scales)
    for i, s in enumerate(scales):
        cfg.upsample_scales[i] = int(s)
    cfg.freq_axis_kernel_size = int(freq_axis_kernel_size)
    if upsample_activation not in _lib.UPSAMPLE_ACT:
        raise NotImplementedError(f"upsample_activation {upsample_activation!r} (implemented: {sorted(_lib.UPSAMPLE_ACT)})")
    kind, pname = _lib.UPSAMPLE_ACT[upsample_activation]
    params = dict(upsample_activation_params or {})
    params.pop("inplace", None)
    if set(params) - ({pname} if pname else set()):
        raise NotImplementedError(f"upsample_activation_params {sorted(params)} of {upsample_activation}")
    cfg.upsample_activation = kind
    if upsample_mode not in _lib.UPSAMPLE_MODE:
        raise NotImplementedError(f"upsampling mode {upsample_mode!r} (implemented: {sorted(_lib.UPSAMPLE_MODE)})")
    cfg.upsample_mode = _lib.UPSAMPLE_MODE[upsample_mode]
    cfg.upsample_activation_param = float(params.get(pname, _lib.UPSAMPLE_ACT_DEFAULT.get(pname, 0.0))) if pname else 0.0
    cfg.cin_pad = int(cin_pad)
    return cfg


def _tensor_table(named: Dict[str, torch.Tensor]):
    """state_dict -> (ctypes array of wnv_tensor, keep-alive list).  Host fp32 contiguous copies."""
    keep = []
    arr = (Tensor * len(named))()
    for i, (name, t) in enumerate(named.items()):
        h = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        nm = name.encode()
        keep.append((h, nm))
        arr[i].name = nm
        arr[i].data = h.data_ptr()
        arr[i].ndim = h.dim()
        for d in range(h.dim()):
            arr[i].shape[d] = h.shape[d]
    return arr, keep


def _ptr(t) -> Optional[int]:
    """A tensor's address, or an address given as an int (device pointers of PinnedBuffer)."""
    return None if t is None else (int(t) if isinstance(t, int) else t.data_ptr())


class PinnedBuffer:
    """Coherent, device-mapped host memory (wnv_pinned_alloc): ``.host`` / ``.dev`` addresses and zero-copy torch views -- what a
    replay tape that is still being drawn while the kernel reads it lives in (include/wnv.h, wnv_generate_args.noise_ready)."""

    def __init__(self, nbytes: int):
        host, dev = C.c_void_p(), C.c_void_p()
        check(_lib.lib().wnv_pinned_alloc(C.c_size_t(int(nbytes)), C.byref(host), C.byref(dev)))
        self.host, self.dev, self.nbytes = host.value, dev.value, int(nbytes)

    def view(self, dtype: torch.dtype, shape: Sequence[int], offset: int = 0) -> torch.Tensor:
        import numpy as np
        raw = (C.c_ubyte * (self.nbytes - offset)).from_address(self.host + offset)
        n = 1
        for d in shape:
            n *= int(d)
        return torch.from_numpy(np.frombuffer(raw, dtype=np.uint8))[: n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(*shape)

    def free(self):
        if self.host:
            _lib.lib().wnv_pinned_free(C.c_void_p(self.host))
            self.host = self.dev = 0

    # a copy (copy.deepcopy / pickle of a module that caches one) owns nothing: the memory belongs to the original
    def __deepcopy__(self, memo):
        other = object.__new__(PinnedBuffer)
        other.host = other.dev = other.nbytes = 0
        return other

    def __reduce__(self):
        return (_empty_pinned, ())

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


def _empty_pinned():
    b = object.__new__(PinnedBuffer)
    b.host = b.dev = b.nbytes = 0
    return b


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def check_checkpoint(cfg: Config, state: Dict[str, torch.Tensor], batch: int = 8) -> Dict[str, int]:
    """Run a ``state_dict`` through the engine's native checkpoint path WITHOUT a device (``wnv_create(device = -1)`` +
    ``wnv_load_weights``): key / shape validation, weight-norm fold, packing.  Raises what loading on a GPU would raise;
    returns the algorithmic work the engine derives from it."""
    h = C.c_void_p()
    check(_lib.lib().wnv_create(C.byref(cfg), -1, C.byref(h)))
    try:
        arr, keep = _tensor_table(state)
        check(_lib.lib().wnv_load_weights(h, arr, len(state)))
        del keep
        return {"macs_per_sample": int(_lib.lib().wnv_macs_per_sample(h)),
                "bytes_per_step": int(_lib.lib().wnv_bytes_per_step(h, int(batch))),
                "receptive_field": int(_lib.lib().wnv_receptive_field(cfg.layers, cfg.stacks, cfg.kernel_size))}
    finally:
        _lib.lib().wnv_destroy(h)


class Engine:
    """One ``wnv_handle``: packed weights + scratch for one model on one device."""

    # a deep copy / pickle of a module that holds an engine gets none: the handle belongs to the original, the copy packs its own
    # weights on its first call (EngineHost._get_engine keys the cache on the parameters' storage)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def __init__(self, cfg: Config, device: torch.device):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Engine needs a HIP ('cuda') device")
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        check(_lib.lib().wnv_create(C.byref(cfg), idx, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().wnv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state: Dict[str, torch.Tensor]) -> None:
        arr, keep = _tensor_table(state)
        check(_lib.lib().wnv_load_weights(self._h, arr, len(state)))
        del keep

    def noise_width(self) -> int:
        return int(_lib.lib().wnv_noise_width(C.byref(self.cfg)))

    def upsampled_length(self, tc_in: int) -> int:
        return int(_lib.lib().wnv_upsampled_length(C.byref(self.cfg), int(tc_in)))

    def bytes_per_step(self, B: int) -> int:
        return int(_lib.lib().wnv_bytes_per_step(self._h, int(B)))

    def macs_per_sample(self) -> int:
        return int(_lib.lib().wnv_macs_per_sample(self._h))

    def upsample(self, c: torch.Tensor, T_expected: int = -1) -> torch.Tensor:
        """(B, cin, Tc_in) device -> (B, T, cin) device, time-major."""
        require_gpu_tensor(c, "c")
        c = c.detach().to(torch.float32).contiguous()
        B, cin, tc = c.shape
        T = self.upsampled_length(tc)
        if T <= 0:
            raise AssertionError(f"conditioning of {tc} frames is too short for the upsampling network")
        out = torch.empty(B, T, cin, device=c.device, dtype=torch.float32)
        check(_lib.lib().wnv_upsample(self._h, c.data_ptr(), B, tc, out.data_ptr(), int(T_expected),
                                      _stream(c.device)))
        return out

    def forward(self, x: torch.Tensor, c_up=None, g=None, g_ids=None, softmax: bool = False) -> torch.Tensor:
        """Teacher-forced batch evaluation on the device (wnv_forward, f32 MFMA): x (B, Cin, T) -> (B, out_channels, T).
        Raises NotImplementedError for shapes the MFMA path does not cover."""
        require_gpu_tensor(x, "x")
        x = x.detach().float().contiguous()
        B, Cx, T = x.shape
        cfg = self.cfg
        if Cx != (1 if cfg.scalar_input else cfg.out_channels):
            raise ValueError(f"x has {Cx} channels, the model takes {1 if cfg.scalar_input else cfg.out_channels}")
        if c_up is not None:            # raw pointers cross the boundary: shape, dtype, device and layout are checked here
            require_gpu_tensor(c_up, "c_up")
            if c_up.dtype != torch.float32 or not c_up.is_contiguous() or tuple(c_up.shape) != (B, T, cfg.cin_channels):
                raise ValueError(f"c_up must be a contiguous float32 (B, T, cin) = {(B, T, cfg.cin_channels)} tensor, got {c_up.dtype} {tuple(c_up.shape)}")
        if g is not None:
            require_gpu_tensor(g, "g")
            if g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != (B, cfg.gin_channels):
                raise ValueError(f"g must be a contiguous float32 (B, gin) = {(B, cfg.gin_channels)} tensor")
        if g_ids is not None:
            require_gpu_tensor(g_ids, "g_ids")
            if g_ids.dtype != torch.int64 or not g_ids.is_contiguous() or g_ids.numel() != B:
                raise ValueError("g_ids must be a contiguous int64 (B,) tensor")
        out = torch.empty(B, self.cfg.out_channels, T, device=self.device, dtype=torch.float32)
        a = _lib.ForwardArgs(B=B, T=T, x=_ptr(x), c_up=_ptr(c_up), g=_ptr(g), g_ids=_ptr(g_ids), out=out.data_ptr(),
                             softmax=int(bool(softmax)), stream=_stream(self.device))
        check(_lib.lib().wnv_forward(self._h, C.byref(a)))
        return out

    def generate(self, *, B: int, T: int, c_up=None, g=None, g_ids=None, initial=None, teacher=None,
                 noise=None, seed: int = 0, softmax: bool = True, quantize: bool = True,
                 want_params: bool = False, want_index: bool = False, kernel: int = 0, asynchronous: bool = False,
                 noise_ready=None, seg_start=None, seg_uid=None, seg_gid=None, want_out: bool = True):
        """Runs the whole autoregressive loop.  Returns (out (B,C,T), params (B,O,T)|None, index (B,T)|None).
        ``asynchronous`` (ring kernel chosen explicitly, kernel=2): return right after the launch; ``wait()`` or the next
        call reports a bounded-spin timeout (WNV_GEN_ASYNC in include/wnv.h).  ``noise`` / ``noise_ready`` may be device
        addresses (ints) of a PinnedBuffer: a tape the caller keeps filling while the kernel runs (needs kernel=2, asynchronous).
        ``seg_start`` / ``seg_uid`` (+ ``seg_gid`` for models with global conditioning: ``g`` / ``g_ids`` then hold one row per speaker or
        per utterance of the job and ``seg_gid[b][t]`` picks the row): packed slots, include/wnv.h."""
        dev = self.device
        cfg = self.cfg
        C_out = 1 if cfg.scalar_input else cfg.out_channels
        if not want_out and (cfg.scalar_input or not quantize or not want_index or not (seg_start is not None or kernel == 1)):
            raise ValueError("want_out=False is for one-hot models that sample classes (quantize) and return them (want_index), in a "
                             "packed-slot launch or on the generic kernel (include/wnv.h)")
        out = torch.empty(B, C_out, T, device=dev, dtype=torch.float32) if want_out else None      # (one-hot output: 4 out_channels bytes per sample)
        params = torch.empty(B, cfg.out_channels, T, device=dev, dtype=torch.float32) if want_params else None
        index = torch.empty(B, T, device=dev, dtype=torch.int32) if want_index else None
        a = GenerateArgs()
        a.B, a.T = B, T
        a.c_up, a.g, a.g_ids = _ptr(c_up), _ptr(g), _ptr(g_ids)
        a.initial, a.teacher = _ptr(initial), _ptr(teacher)
        a.Tt = 0 if teacher is None else teacher.shape[1]
        a.noise = _ptr(noise)
        a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        a.softmax, a.quantize = int(bool(softmax)), int(bool(quantize))
        a.out, a.params_out, a.index_out = _ptr(out), _ptr(params), _ptr(index)
        a.kernel = int(kernel)
        a.flags = _lib.WNV_GEN_ASYNC if asynchronous else 0
        a.stream = _stream(dev)
        a.noise_ready = _ptr(noise_ready)
        if (seg_start is None) != (seg_uid is None):
            raise ValueError("seg_start and seg_uid come together (packed slots, include/wnv.h)")
        if seg_start is not None:       # packed slots: (B, T) int32 each; raw pointers cross the boundary, so shape / dtype / device are checked here
            for name, tns in (("seg_start", seg_start), ("seg_uid", seg_uid)):
                require_gpu_tensor(tns, name)
                if tns.dtype != torch.int32 or not tns.is_contiguous() or tuple(tns.shape) != (B, T):
                    raise ValueError(f"{name} must be a contiguous int32 (B, T) = {(B, T)} tensor, got {tns.dtype} {tuple(tns.shape)}")
            a.seg_start, a.seg_uid = _ptr(seg_start), _ptr(seg_uid)
            if seg_gid is not None:
                require_gpu_tensor(seg_gid, "seg_gid")
                if seg_gid.dtype != torch.int32 or not seg_gid.is_contiguous() or tuple(seg_gid.shape) != (B, T):
                    raise ValueError(f"seg_gid must be a contiguous int32 (B, T) = {(B, T)} tensor, got {seg_gid.dtype} {tuple(seg_gid.shape)}")
                rows = g if g is not None else g_ids
                if rows is None:
                    raise ValueError("seg_gid needs g or g_ids (one row per speaker / utterance of the job)")
                a.seg_gid, a.n_g = _ptr(seg_gid), int(rows.shape[0])
        elif seg_gid is not None:
            raise ValueError("seg_gid belongs to packed slots (seg_start, seg_uid)")
        check(_lib.lib().wnv_generate(self._h, C.byref(a)))
        return out, params, index

    def wait(self) -> None:
        """Status of the last asynchronous launch (raises TimeoutError if a bounded in-kernel wait gave up)."""
        check(_lib.lib().wnv_wait(self._h))

    def reset_buffers(self) -> None:
        """``wnv_reset``: waits for the device, frees the handle's scratch buffers (they are re-grown on demand) and lets a persistent
        kernel that timed out be tried again by the very next call (include/wnv.h).  The weights stay loaded."""
        check(_lib.lib().wnv_reset(self._h))

    def kernel_coverage(self, kernel: int, B: int = 1) -> str:
        """``wnv_kernel_coverage``: "supported", or why the kernel (2 ring, 3 group ring) does not take this configuration."""
        why = _lib.lib().wnv_kernel_coverage(C.byref(self.cfg), int(kernel), int(B))
        return why.decode("utf-8", "replace") if why else "supported"

    def last_kernel(self) -> int:
        """1 = generic kernel, 2 = pipelined ring kernel served the last ``generate`` (0: none yet)."""
        return int(_lib.lib().wnv_last_kernel(self._h))

    def inject_timeouts(self, n: int) -> None:
        """Test hook (``wnv_debug_inject_timeouts``, include/wnv_test.h): the next ``n`` ring launches of auto mode report a time-out
        unlaunched.  Exists in the TEST library only (WNV_LIB=.../libwnv_test.so)."""
        if not _lib.has_test_hooks():
            raise RuntimeError("wnv_debug_inject_timeouts is a hook of the test library: run with WNV_LIB=" + _lib.TEST_LIB_PATH)
        check(_lib.lib().wnv_debug_inject_timeouts(self._h, int(n)))


class QueueConv:
    """conv.Conv1d.incremental_forward state machine on the device (reference conv.py:17-49)."""

    def __deepcopy__(self, memo):          # (as Engine: the handle stays with the original, the copy builds its own on first use)
        return None

    def __reduce__(self):
        return (type(None), ())

    def __init__(self, cin: int, cout: int, kernel_size: int, dilation: int, device: torch.device):
        self.device = torch.device(device)
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.cin, self.cout = cin, cout
        check(_lib.lib().wnv_qconv_create(cin, cout, kernel_size, dilation, idx, C.byref(self._h)))

    def set_weights(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        check(_lib.lib().wnv_qconv_set_weights(self._h, w.data_ptr(), _ptr(b)))

    def step(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, cin) -> (B, cout)."""
        x = x.detach().to(torch.float32).contiguous()
        y = torch.empty(x.shape[0], self.cout, device=x.device, dtype=torch.float32)
        check(_lib.lib().wnv_qconv_step(self._h, x.data_ptr(), y.data_ptr(), x.shape[0], _stream(x.device)))
        return y

    def reset(self):
        check(_lib.lib().wnv_qconv_reset(self._h))

    def __del__(self):  # pragma: no cover
        try:
            if self._h.value:
                _lib.lib().wnv_qconv_destroy(self._h)
        except Exception:
            pass


class GluLayer:
    """ResidualConv1dGLU.incremental_forward state machine on the device (reference modules.py:112-169)."""

    def __deepcopy__(self, memo):          # (as Engine: the handle stays with the original, the copy builds its own on first use)
        return None

    def __reduce__(self):
        return (type(None), ())

    def __init__(self, *, residual_channels, gate_channels, kernel_size, skip_out_channels, cin_channels,
                 gin_channels, dilation, bias, device):
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        cfg = GluConfig(residual_channels, gate_channels, kernel_size, skip_out_channels,
                        cin_channels, gin_channels, dilation, int(bool(bias)))
        self.R, self.K = residual_channels, skip_out_channels
        self._h = C.c_void_p()
        check(_lib.lib().wnv_glu_create(C.byref(cfg), idx, C.byref(self._h)))

    def load_weights(self, state: Dict[str, torch.Tensor]):
        arr, keep = _tensor_table(state)
        check(_lib.lib().wnv_glu_load_weights(self._h, arr, len(state)))
        del keep

    def step(self, x, c=None, g=None):
        x = x.detach().to(torch.float32).contiguous()
        c = None if c is None else c.detach().to(torch.float32).contiguous()
        g = None if g is None else g.detach().to(torch.float32).contiguous()
        B = x.shape[0]
        xo = torch.empty(B, self.R, device=x.device, dtype=torch.float32)
        so = torch.empty(B, self.K, device=x.device, dtype=torch.float32)
        check(_lib.lib().wnv_glu_step(self._h, x.data_ptr(), _ptr(c), _ptr(g), xo.data_ptr(), so.data_ptr(), B,
                                      _stream(x.device)))
        return xo, so

    def reset(self):
        check(_lib.lib().wnv_glu_reset(self._h))

    def __del__(self):  # pragma: no cover
        try:
            if self._h.value:
                _lib.lib().wnv_glu_destroy(self._h)
        except Exception:
            pass
