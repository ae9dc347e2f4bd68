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

__all__ = ["batch_wavegen", "wavegen", "postprocess", "default_hparams", "is_mulaw_quantize", "is_mulaw", "is_raw", "sanity_check"]

INPUT_TYPES = {"raw": 0, "mulaw": 1, "mulaw-quantize": 2}


def default_hparams(**over) -> SimpleNamespace:
    """The fields batch_wavegen reads, with the values of the reference's egs/mol preset (hparams.py)."""
    h = SimpleNamespace(input_type="raw", quantize_channels=65536, upsample_conditional_features=True, cin_pad=2,
                        hop_size=256, log_scale_min=-32.23619130191664, postprocess="inv_preemphasis",
                        global_gain_scale=0.55, preemphasis_coef=0.85,
                        cin_channels=80, sample_rate=22050,
                        batch_size=None)        # read by evaluate.synthesize_dir: utterances per launch; None = from the measured
                                                # throughput curve (sharding.auto_group_size; the reference's recipes pass 32)
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


def postprocess(y_hat: torch.Tensor, hparams, *, clip: bool = False, want_int16: bool = False, mu: Optional[int] = None):
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
                      mu=int(hparams.quantize_channels) - 1 if mu is None else int(mu), preemphasis=coef,
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


def _to_numpy(x):
    """synthesis._to_numpy (synthesis.py:89-98)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray) or np.isscalar(x):
        return x
    if x.dim() == 3:                     # remove batch axis
        x = x.squeeze(0)
    return x.cpu().numpy()


def wavegen(model, length=None, c=None, g=None, initial_value=None, fast=False, tqdm=None, hparams=None) -> np.ndarray:
    """synthesis.wavegen (synthesis.py:101-186): ONE utterance; ``c`` is (Tc, cin) frames (numpy), ``g`` a scalar speaker id,
    ``initial_value`` the first input (class index for mulaw-quantize, else a float).  Returns the 1-D waveform.

    Kept quirks of the reference: the length is ``Tc * hop_size`` (no ``cin_pad`` context frames are expected, unlike
    batch_wavegen), and the mu-law decoders are called with ``quantize_channels`` here (synthesis.py:175-179) where batch_wavegen
    passes ``quantize_channels - 1`` (synthesis.py:68-74)."""
    hparams = hparams or default_hparams()
    sanity_check(model, c, g)
    c = _to_numpy(c)
    g = _to_numpy(g)
    model.eval()
    if fast:
        model.make_generation_fast_()
    dev = next(model.parameters()).device
    if c is None:
        assert length is not None
    else:
        if c.ndim != 2:
            raise RuntimeError("Expected 2-dim shape (T, {}) for the conditional feature, but {} was actually given.".format(
                getattr(hparams, "cin_channels", model.cin_channels), c.shape))
        Tc = c.shape[0]
        upsample_factor = hparams.hop_size
        length = Tc * upsample_factor                                        # synthesis.py:135-139
        if not hparams.upsample_conditional_features:
            c = np.repeat(c, upsample_factor, axis=0)                        # synthesis.py:142-143
        c = torch.from_numpy(np.ascontiguousarray(c.T, dtype=np.float32)).unsqueeze(0)      # B x C x T
    qc = int(hparams.quantize_channels)
    if initial_value is None:
        initial_value = int((0.0 + 1) / 2 * (qc - 1)) if is_mulaw_quantize(hparams.input_type) else 0.0   # P.mulaw_quantize(0, mu)
    if is_mulaw_quantize(hparams.input_type):
        assert initial_value >= 0 and initial_value < qc
        initial_input = torch.zeros(1, 1, qc)
        initial_input[0, 0, int(initial_value)] = 1.0                        # to_categorical
    else:
        initial_input = torch.zeros(1, 1, 1).fill_(float(initial_value))
    g = None if g is None else torch.as_tensor([int(g)], dtype=torch.long)
    initial_input = initial_input.to(dev)
    g = None if g is None else g.to(dev)
    c = None if c is None else c.to(dev)
    with torch.no_grad():
        y_hat = model.incremental_forward(initial_input, c=c, g=g, T=length, tqdm=tqdm or (lambda x: x), softmax=True,
                                          quantize=True, log_scale_min=hparams.log_scale_min)
    return postprocess(y_hat, hparams, mu=qc).cpu().numpy().reshape(-1)
