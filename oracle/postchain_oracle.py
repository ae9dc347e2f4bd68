"""CPU oracle of the post-chain of synthesis.batch_wavegen / evaluate.py (TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path).

The arithmetic lives in the reference's dependency **nnmnkwii** (`nnmnkwii >= 0.0.11`, setup.py:23 -- unpinned, not
vendored under /root/reference and not installed here), so the functions below restate its published definitions and are
anchored on the reference's call sites:

    synthesis.py:68-70   y_hat.max(1)[1] -> P.inv_mulaw_quantize(y, quantize_channels - 1)
    synthesis.py:72-74   P.inv_mulaw(y, quantize_channels - 1)
    synthesis.py:78-80   getattr(audio, hparams.postprocess)(y)   = audio.inv_preemphasis(x, coef=0.85) = P.inv_preemphasis
    synthesis.py:82-84   y /= hparams.global_gain_scale
    evaluate.py:238      np.clip(gen, -1, 1);   evaluate.py:43-48  to_int16: (x * 32767).astype(np.int16)

PARITY UNPINNED: the reference's tests hold no vector for these functions and nnmnkwii cannot be run here; what pins the
restatement are the closed-form properties in tests/test_postchain_cpu.py (mu-law companding is the inverse of the
published forward formula, the IIR is the inverse of the FIR pre-emphasis, known values at 0 and +-1).
"""
import numpy as np
from scipy.signal import lfilter


def mulaw(x, mu=255):                      # nnmnkwii.preprocessing.mulaw: sign(x) log(1 + mu|x|) / log(1 + mu)
    x = np.asarray(x, dtype=np.float64)
    return np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)


def inv_mulaw(y, mu=255):                  # nnmnkwii.preprocessing.inv_mulaw: sign(y) (1/mu) ((1 + mu)^|y| - 1)
    y = np.asarray(y, dtype=np.float64)
    return np.sign(y) * (1.0 / mu) * ((1.0 + mu) ** np.abs(y) - 1.0)


def mulaw_quantize(x, mu=255):             # nnmnkwii: ((mulaw(x) + 1) / 2 * mu).astype(int)
    return ((mulaw(x, mu) + 1) / 2 * mu).astype(np.int64)


def inv_mulaw_quantize(y, mu=255):         # nnmnkwii: inv_mulaw(2 y / mu - 1, mu)
    y = 2 * np.asarray(y, dtype=np.float64) / mu - 1
    return inv_mulaw(y, mu)


def preemphasis(x, coef=0.85):             # nnmnkwii: lfilter([1, -coef], [1], x)
    return lfilter([1, -coef], [1], np.asarray(x, dtype=np.float64))


def inv_preemphasis(x, coef=0.85):         # nnmnkwii: lfilter([1], [1, -coef], x)
    return lfilter([1], [1, -coef], np.asarray(x, dtype=np.float64))


def to_int16(x):                           # evaluate.py:43-48
    x = np.asarray(x)
    if x.dtype == np.int16:
        return x
    assert x.dtype == np.float32
    assert x.min() >= -1 and x.max() <= 1.0
    return (x * 32767).astype(np.int16)


def post_chain(y_hat, input_type="raw", quantize_channels=65536, postprocess="inv_preemphasis", coef=0.85,
               global_gain_scale=0.55, clip=False):
    """y_hat: (B, C, T) float array as returned by incremental_forward.  Mirrors synthesis.py:66-84 (+ evaluate.py:238)."""
    y_hat = np.asarray(y_hat)
    B = y_hat.shape[0]
    mu = quantize_channels - 1
    if input_type == "mulaw-quantize":
        y = y_hat.argmax(axis=1).reshape(B, -1).astype(np.float32)
        y = np.stack([inv_mulaw_quantize(y[i], mu) for i in range(B)])
    elif input_type == "mulaw":
        y = np.stack([inv_mulaw(y_hat.reshape(B, -1)[i], mu) for i in range(B)])
    else:
        y = y_hat.reshape(B, -1).astype(np.float64)
    if postprocess not in (None, "", "none"):
        y = np.stack([inv_preemphasis(y[i], coef) for i in range(B)])
    if global_gain_scale > 0:
        y = y / global_gain_scale
    if clip:
        y = np.clip(y, -1.0, 1.0)
    return y.astype(np.float32)
