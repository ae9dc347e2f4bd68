"""Host mirror of the mel front end of the reference's ``audio.py`` on top of ``wnv_logmel``.

``audio.logmelspectrogram`` (audio.py:101-109) = ``librosa.stft`` -> magnitude -> ``librosa.filters.mel`` projection ->
``log10(max(., 1e-10))``; the recipes then apply a per-bin ``StandardScaler`` (preprocess_normalize.py:44) and the result
is what the synthesis path receives as ``c``.  Here the whole chain is one HIP launch per batch (``csrc/wnv_mel.hip``).
The reference reads its settings from the global ``hparams``; this module takes any object with the same attribute names
(``sample_rate, fft_size, hop_size | frame_shift_ms, win_length | win_length_ms, window, num_mels, fmin, fmax``), so a
reference ``hparams`` instance can be handed over unchanged.  There is no CPU path: inputs are moved to the HIP device.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib

__all__ = ["MelFrontEnd", "logmelspectrogram", "default_hparams", "get_hop_size", "get_win_length"]


def default_hparams(**over) -> SimpleNamespace:
    """The audio fields of hparams.py:32-47 with the reference's defaults."""
    h = SimpleNamespace(sample_rate=22050, num_mels=80, fmin=125, fmax=7600, fft_size=1024, hop_size=256,
                        frame_shift_ms=None, win_length=1024, win_length_ms=-1.0, window="hann")
    h.__dict__.update(over)
    return h


def get_hop_size(hparams) -> int:
    """audio.get_hop_size (audio.py:112-117)."""
    hop_size = hparams.hop_size
    if hop_size is None:
        assert hparams.frame_shift_ms is not None
        hop_size = int(hparams.frame_shift_ms / 1000 * hparams.sample_rate)
    return hop_size


def get_win_length(hparams) -> int:
    """audio.get_win_length (audio.py:120-125)."""
    win_length = hparams.win_length
    if win_length < 0:
        assert hparams.win_length_ms > 0
        win_length = int(hparams.win_length_ms / 1000 * hparams.sample_rate)
    return win_length


_PAD = {"constant": 0, "reflect": 1}


class MelFrontEnd:
    """One ``wnv_mel_handle``: window, twiddles and the mel filterbank of one hparams set on one device."""

    def __init__(self, hparams=None, device="cuda", pad_mode: str = "reflect"):
        hparams = hparams if hparams is not None else default_hparams()
        if getattr(hparams, "window", "hann") != "hann":
            raise NotImplementedError("only window='hann' (every reference preset) is implemented")
        if pad_mode not in _PAD:
            raise NotImplementedError(f"pad_mode {pad_mode!r}: 'reflect' (logmelspectrogram's default) or 'constant'")
        if hparams.fmax is not None:
            assert hparams.fmax <= hparams.sample_rate // 2                       # audio.py:153-154
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MelFrontEnd needs a HIP ('cuda') device; there is no CPU path")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        cfg = _lib.MelConfig()
        cfg.sample_rate = int(hparams.sample_rate)
        cfg.fft_size = int(hparams.fft_size)
        cfg.hop_size = int(get_hop_size(hparams))
        cfg.win_length = int(get_win_length(hparams))
        cfg.num_mels = int(hparams.num_mels)
        cfg.fmin = float(hparams.fmin)
        cfg.fmax = float(hparams.fmax) if hparams.fmax is not None else 0.0
        cfg.pad_mode = _PAD[pad_mode]
        cfg.floor = 1e-10
        self.cfg = cfg
        self._h = C.c_void_p()
        _lib.check(_lib.lib().wnv_mel_create(C.byref(cfg), idx, C.byref(self._h)))
        self._has_scaler = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().wnv_mel_destroy(h)
            except Exception:
                pass
            h.value = None

    # ---- introspection ---------------------------------------------------------------------------
    def frames(self, n: int) -> int:
        return int(_lib.lib().wnv_mel_frames(C.byref(self.cfg), int(n)))

    def mel_basis(self) -> np.ndarray:
        """The filterbank in use, ``(num_mels, fft_size // 2 + 1)`` float32 (= ``audio._build_mel_basis()``)."""
        out = np.empty((self.cfg.num_mels, self.cfg.fft_size // 2 + 1), dtype=np.float32)
        _lib.check(_lib.lib().wnv_mel_basis(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_scaler(self, scaler=None, mean=None, scale=None) -> "MelFrontEnd":
        """Install a fitted ``sklearn.preprocessing.StandardScaler`` (``mean_``, ``scale_``) or explicit arrays."""
        if scaler is not None:
            mean, scale = scaler.mean_, scaler.scale_
        mean = np.ascontiguousarray(np.asarray(mean, dtype=np.float32))
        scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float32))
        if mean.shape != (self.cfg.num_mels,) or scale.shape != (self.cfg.num_mels,):
            raise ValueError(f"scaler statistics must have shape ({self.cfg.num_mels},)")
        _lib.check(_lib.lib().wnv_mel_set_scaler(self._h, mean.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p)))
        self._has_scaler = True
        return self

    # ---- compute ---------------------------------------------------------------------------------
    def _run(self, y, transpose: bool, normalize: bool) -> torch.Tensor:
        y = torch.as_tensor(y)
        single = y.dim() == 1
        if single:
            y = y.unsqueeze(0)
        if y.dim() != 2:
            raise ValueError("waveforms must be (n,) or (B, n)")
        y = y.to(self.device, torch.float32).contiguous()
        B, n = y.shape
        N = self.frames(n)
        shape = (B, self.cfg.num_mels, N) if transpose else (B, N, self.cfg.num_mels)
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        a = _lib.LogmelArgs()
        a.B, a.n, a.wav_stride = B, n, n
        a.wav, a.out = y.data_ptr(), out.data_ptr()
        a.transpose, a.normalize = int(transpose), int(normalize)
        a.stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().wnv_logmel(self._h, C.byref(a)))
        return out[0] if single else out

    def logmelspectrogram(self, y) -> torch.Tensor:
        """audio.logmelspectrogram: ``(n,)`` -> ``(num_mels, frames)`` (a leading batch axis is kept)."""
        return self._run(y, transpose=True, normalize=False)

    def feats(self, y, normalize: Optional[bool] = None) -> torch.Tensor:
        """The ``*-feats.npy`` rows: ``logmelspectrogram(y).T`` (datasets/wavallin.py:62), scaled when a scaler is set."""
        return self._run(y, transpose=False, normalize=self._has_scaler if normalize is None else normalize)


_default: dict = {}


def logmelspectrogram(y, pad_mode: str = "reflect", hparams=None):
    """Drop-in for ``audio.logmelspectrogram(y, pad_mode)``: numpy in -> numpy ``(num_mels, frames)`` out, torch in ->
    torch (device) out."""
    hp = hparams if hparams is not None else default_hparams()
    key = (tuple(sorted((k, v) for k, v in vars(hp).items() if not k.startswith("_") and isinstance(v, (int, float, str, type(None))))),
           pad_mode, torch.cuda.current_device())
    fe = _default.get(key)
    if fe is None:
        fe = _default[key] = MelFrontEnd(hp, pad_mode=pad_mode)
    out = fe.logmelspectrogram(y)
    return out.cpu().numpy() if isinstance(y, np.ndarray) else out
