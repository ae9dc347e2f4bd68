"""The in-kernel noise stream of the tape-less mode, restated in numpy (test infrastructure).

csrc/wnv_dev.h: wnv_philox (Philox4x32-10, Salmon et al. 2011 -- the Random123 known answers are in tests/test_philox_cpu.py),
wnv_u01 and wnv_noise_gen: value j of (utterance b, step t) under ``seed`` = Philox(counter = (t lo, t hi, b, j), key = (seed lo, seed hi));
the first output word gives the uniform, kind 0: U(1e-5, 1 - 1e-5), kind 2: Exp(1) = -log u.

The uniform is ((x >> 9) + 0.5) / 2^23: exact in float32 (the sum needs 24 bits, the scale is a power of two), never 0 and never 1, so
e = -log u > 0 for every draw and every form of the categorical pick is defined everywhere (round 6; until round 5 the map was
((x >> 8) + 0.5) / 2^24, which rounds to 1.0 once in 2^24 draws).  The margin helpers below still treat e <= 0 as "cannot be picked"."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Four uint32 output words (as uint64 arrays) for broadcastable counter words and a scalar key."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & _MASK for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> _S32, p0 & _MASK, p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def first_word(seed, t, b, j):
    t = np.asarray(t, dtype=np.uint64)
    return philox4x32_10(t & _MASK, t >> _S32, b, j, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)[0]


def uniform01(seed, t, b, j):
    """wnv_u01 of the first word: float32 arithmetic, round to nearest even"""
    x = (first_word(seed, t, b, j) >> np.uint64(9)).astype(np.float32)           # (< 2^23: exact, and so is the sum)
    return ((x + np.float32(0.5)).astype(np.float32) * np.float32(1.0 / 8388608.0)).astype(np.float32)


def exp_noise(seed, t, b, j):
    """kind 2: e = -log u, float64 of the float32 uniform (the device's logf is within an ulp of it)"""
    return -np.log(uniform01(seed, t, b, j).astype(np.float64))


def exp_grid(seed, uid, T, O):
    """(T, O) e ~ Exp(1) of utterance ``uid``: step t, class j"""
    return exp_noise(seed, np.arange(T, dtype=np.uint64)[:, None], np.uint64(uid), np.arange(O, dtype=np.uint64)[None, :])


def categorical_pick_margins(logits, classes, seed, uid, e=None):
    """``logits`` (O, T) head outputs the kernel reported for utterance ``uid``, ``classes`` (T,) what it picked under in-kernel noise
    of ``seed``: per step, how far (float64, log domain) the picked class's score logit_k - log e_k is below the best one -- 0 where the
    kernel picked the argmax.  A class whose e rounded to zero scores -inf (see the module docstring).  ``e``: ``exp_grid`` of the
    utterance (at least T rows) when the caller keeps it."""
    logits = np.asarray(logits, dtype=np.float64)
    O, T = logits.shape
    e = exp_grid(seed, uid, T, O) if e is None else e[:T]
    with np.errstate(divide="ignore"):
        score = np.where(e > 0, logits.T - np.log(np.where(e > 0, e, 1.0)), -np.inf)
    picked = score[np.arange(T), np.asarray(classes, dtype=np.int64)]
    return score.max(-1) - picked


def tape(seed, T, B, *, scalar_input, output_distribution="Logistic", out_channels=30, b0=0):
    """The (T, B, NZ) float32 noise tape the in-kernel stream stands for (wnv_noise_gen: "same semantics as the tape", layout of
    wavenet_vocoder_amd/noise.py): value j of (utterance b0 + b, step t), kind by position -- Logistic: nr_mix + 1 uniforms of
    U(1e-5, 1 - 1e-5) (kind 0); Normal: nr_mix uniforms then one N(0, 1) (kind 1: Box-Muller of the first two output words), or the
    single N(0, 1) for 2 / 3 output channels; one-hot: out_channels draws of Exp(1) (kind 2; strictly positive)."""
    if scalar_input:
        normal = output_distribution == "Normal"
        nz = 1 if (normal and out_channels in (2, 3)) else out_channels // 3 + 1
    else:
        normal, nz = False, out_channels
    t = np.arange(T, dtype=np.uint64)[:, None, None]
    b = (np.arange(B, dtype=np.uint64) + np.uint64(b0))[None, :, None]
    j = np.arange(nz, dtype=np.uint64)[None, None, :]
    w = philox4x32_10(t & _MASK, t >> _S32, b, j, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    f32 = np.float32

    def u01(x):
        return (((x >> np.uint64(9)).astype(f32) + f32(0.5)).astype(f32) * f32(1.0 / 8388608.0)).astype(f32)

    u = u01(w[0])
    if not scalar_input:
        with np.errstate(divide="ignore"):
            return (-np.log(u.astype(np.float64))).astype(f32)
    # kind 0: 1e-5f + u * (1.0f - 2e-5f) is ONE fused multiply-add on the device (v_fmac_f32 0x3727c5ac + 0x3f7ffeb0 * u: checked in the
    # ISA) -- a single rounding, emulated through float64 (the 48-bit product is exact there).  It matters: log(1 - u) near u = 1 - 1e-5
    # turns one ulp of u into 6e-3 of the logistic's argument.
    c = np.float64(f32(1.0) - f32(2e-5))
    out = (u.astype(np.float64) * c + np.float64(f32(1e-5))).astype(f32)
    if normal:
        v = u01(w[1])
        n = np.sqrt(-2.0 * np.log(u.astype(np.float64))) * np.cos(6.28318530717958647692 * v.astype(np.float64))
        out[:, :, nz - 1] = n[:, :, nz - 1].astype(f32)
    return out
