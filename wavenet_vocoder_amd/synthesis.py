"""Host mirror of the reference's ``synthesis.batch_wavegen`` (synthesis.py:41-86) on top of the HIP engine.

Same call shape and return value (a ``(B, T)`` float32 numpy array of waveforms), but the whole chain runs on the
device: ``WaveNet.incremental_forward`` (``wnv_upsample`` + ``wnv_generate``) followed by ``wnv_postprocess``
(argmax / inv-mu-law / inv-preemphasis / gain).  The reference reads its settings from the global
``hparams`` object; here they are passed as an object with the same attribute names (any namespace works), so a
reference ``hparams`` instance can be handed over unchanged.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _stream, require_gpu_tensor
from .util import is_mulaw, is_mulaw_quantize, is_raw  # noqa: F401  (re-exported like synthesis.py:24)

__all__ = ["batch_wavegen", "postprocess", "default_hparams", "is_mulaw_quantize", "is_mulaw", "is_raw", "sanity_check"]

INPUT_TYPES = {"raw": 0, "mulaw": 1, "mulaw-quantize": 2}


def default_hparams(**over) -> SimpleNamespace:
    """The fields batch_wavegen reads, with the values of the reference's egs/mol preset (hparams.py)."""
    h = SimpleNamespace(input_type="raw", quantize_channels=65536, upsample_conditional_features=True, cin_pad=2,
                        hop_size=256, log_scale_min=-32.23619130191664, postprocess="inv_preemphasis",
                        global_gain_scale=0.55, preemphasis_coef=0.85)
    h.__dict__.update(over)
    return h


def sanity_check(model, c, g):
    """train.sanity_check (train.py:72-87), called by batch_wavegen (synthesis.py:43-44)."""
    if model.has_speaker_embedding():
        if g is None:
            raise RuntimeError("WaveNet expects speaker embedding, but speaker-id is not provided")
    else:
        if g is not None:
            raise RuntimeError("WaveNet expects no speaker embedding, but speaker-id is provided")
    if model.local_conditioning_enabled():
        if c is None:
            raise RuntimeError("WaveNet expects conditional features, but not given")
    else:
        if c is not None:
            raise RuntimeError("WaveNet expects no conditional features, but given")


def postprocess(y_hat: torch.Tensor, hparams, *, clip: bool = False, want_int16: bool = False):
    """Device post-chain of synthesis.py:66-84 on ``y_hat`` = incremental_forward's (B, C, T) output.
    Returns a (B, T) float32 device tensor (and the int16 tensor when ``want_int16``)."""
    require_gpu_tensor(y_hat, "y_hat")
    y = y_hat.detach().float().contiguous()
    B, Cc, T = y.shape
    wav = torch.empty(B, T, device=y.device, dtype=torch.float32)
    pcm = torch.empty(B, T, device=y.device, dtype=torch.int16) if want_int16 else None
    post = getattr(hparams, "postprocess", None)
    coef = 0.0
    if post not in (None, "", "none"):
        if post != "inv_preemphasis":
            raise NotImplementedError(f"postprocess '{post}' (the reference's audio module only offers inv_preemphasis)")
        coef = float(getattr(hparams, "preemphasis_coef", 0.85))     # audio.inv_preemphasis(x, coef=0.85), audio.py:57
    a = _lib.PostArgs(B=B, C=Cc, T=T, y=_ptr(y), input_type=INPUT_TYPES[hparams.input_type],
                      mu=int(hparams.quantize_channels) - 1, preemphasis=coef,
                      gain_scale=float(getattr(hparams, "global_gain_scale", 0.0) or 0.0), clip=int(clip or want_int16),
                      wav=_ptr(wav), pcm=_ptr(pcm), stream=_stream(y.device))
    _lib.check(_lib.lib().wnv_postprocess(y.device.index or 0, C.byref(a)))
    return (wav, pcm) if want_int16 else wav


def batch_wavegen(model, c=None, g=None, fast=True, tqdm=None, hparams=None) -> np.ndarray:
    """synthesis.batch_wavegen (synthesis.py:41-86): ``c`` is (B, cin, Tc + 2 cin_pad), ``g`` (B,) speaker ids or None."""
    hparams = hparams or default_hparams()
    sanity_check(model, c, g)
    assert c is not None
    model.eval()
    if fast:
        model.make_generation_fast_()
    dev = next(model.parameters()).device
    g = None if g is None else torch.as_tensor(g).to(dev)
    c = torch.as_tensor(c).to(dev)
    if hparams.upsample_conditional_features:
        length = (c.shape[-1] - hparams.cin_pad * 2) * hparams.hop_size          # synthesis.py:55-57
    else:
        length = c.shape[-1]
    with torch.no_grad():
        y_hat = model.incremental_forward(c=c, g=g, T=length, tqdm=tqdm or (lambda x: x), softmax=True, quantize=True,
                                          log_scale_min=hparams.log_scale_min)
    return postprocess(y_hat, hparams).cpu().numpy()
