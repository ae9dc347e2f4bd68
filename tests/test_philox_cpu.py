"""The numpy restatement of the in-kernel noise stream (tests/_philox.py) against the published known answers of Philox4x32-10, and the
the property of the uniform map the categorical pick relies on: exact in float32, never 0, never 1 (round 6)."""
import re
from pathlib import Path

import numpy as np

from tests import _philox as P

ROOT = Path(__file__).resolve().parents[1]


def test_philox4x32_10_known_answers():
    """Random123's kat_vectors for philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in P.philox4x32_10(*ctr, *key)) == want
    # vectorised = element by element
    t = np.arange(5, dtype=np.uint64)[:, None]
    j = np.arange(7, dtype=np.uint64)[None, :]
    grid = P.philox4x32_10(t, 0, 3, j, 99, 0)[0]
    assert grid.shape == (5, 7) and int(grid[2, 4]) == int(P.philox4x32_10(2, 0, 3, 4, 99, 0)[0])


def test_the_device_code_is_that_algorithm():
    """csrc/wnv_dev.h spells the same rounds: the two multipliers, the two Weyl constants, ten rounds, counter = (t lo, t hi, b, j),
    key = (seed lo, seed hi), the uniform from the first word's top 23 bits."""
    src = (ROOT / "wavenet_vocoder_amd" / "csrc" / "wnv_dev.h").read_text()
    body = src[src.index("wnv_philox("):src.index("wnv_sigmoid")]
    for token in ("0xD2511F53u", "0xCD9E8D57u", "0x9E3779B9u", "0xBB67AE85u", "i < 10", "hi1 ^ c1 ^ k0", "hi0 ^ c3 ^ k1",
                  "(uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)b, (uint32_t)j", "(uint32_t)seed, (uint32_t)(seed >> 32)",
                  "(float)(x >> 9) + 0.5f", "1.0f / 8388608.0f", "return -logf(u)"):
        assert token in body, token
    assert re.search(r"const float u = wnv_u01\(r\[0\]\)", body)


def test_the_uniform_is_exact_and_never_one():
    """((x >> 9) + 0.5) / 2^23 in float32: the sum needs at most 24 bits and the scale is a power of two -- no rounding, 0 < u < 1, so
    e = -log u > 0 for EVERY draw and the quotient form (argmax x_k / e_k) and the log-domain form (argmax logit_k - log e_k) of the
    categorical pick have no edge to disagree on.  Until round 5 the map was ((x >> 8) + 0.5) / 2^24, whose top value 16777215.5 is not a
    float32 and rounded to u = 1.0, e = -0.0 once in 2^24 draws; the event that showed it: seed 99, utterance 12, step 24, class 144."""
    assert np.float32(16777215.0) + np.float32(0.5) == np.float32(16777216.0)          # the old map's top value: why it was replaced
    top = np.float32(8388607.0) + np.float32(0.5)
    assert float(top) == 8388607.5 and float(top * np.float32(1.0 / 8388608.0)) == 1.0 - 2.0 ** -24
    assert float((np.float32(0.0) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)) == 2.0 ** -24
    # every value of the map is exact: float32 arithmetic == exact rational arithmetic, on a sample of words and on the extremes
    x = np.concatenate([np.array([0, 1, 0x1FF, 0x200, 0xFFFFFFFF, 0xFFFFFE00, 0x80000000], dtype=np.uint64),
                        np.random.default_rng(1).integers(0, 2 ** 32, 100000, dtype=np.uint64)])
    f32 = (((x >> np.uint64(9)).astype(np.float32) + np.float32(0.5)).astype(np.float32) * np.float32(1.0 / 8388608.0)).astype(np.float32)
    assert np.array_equal(f32.astype(np.float64), ((x >> np.uint64(9)).astype(np.float64) + 0.5) / 8388608.0)
    assert float(f32.min()) > 0.0 and float(f32.max()) < 1.0
    # the draw that used to give u = 1.0
    w = int(P.first_word(99, 24, 12, 144))
    assert w >> 8 == 0xFFFFFF
    assert float(P.uniform01(99, 24, 12, 144)) == 1.0 - 2.0 ** -24
    assert 0.0 < float(P.exp_noise(99, 24, 12, 144)) < 1e-7
    t = np.arange(2304, dtype=np.uint64)[:, None]
    j = np.arange(256, dtype=np.uint64)[None, :]
    for uid in range(19):
        u = P.uniform01(99, t, np.uint64(uid), j)
        assert float(u.min()) > 0.0 and float(u.max()) < 1.0
    u = P.uniform01(5, t, np.uint64(0), j)
    assert abs(float(u.mean()) - 0.5) < 2e-3
    e = P.exp_noise(5, t, np.uint64(0), j)
    assert float(e.min()) > 0.0 and abs(float(e.mean()) - 1.0) < 5e-3


def test_pick_margins_of_a_host_made_pick():
    """categorical_pick_margins: zero for the argmax of logit - log e, positive otherwise, a class with e = 0 is never the best"""
    g = np.random.default_rng(3)
    logits = g.standard_normal((256, 64))
    e = P.exp_noise(99, np.arange(64, dtype=np.uint64)[:, None], np.uint64(12), np.arange(256, dtype=np.uint64)[None, :])
    with np.errstate(divide="ignore"):
        score = np.where(e > 0, logits.T - np.log(np.where(e > 0, e, 1.0)), -np.inf)
    best = score.argmax(-1)
    assert best[24] == 144                                       # (e = 6e-8 there: the class wins by 16.6 in the log domain)
    assert np.all(P.categorical_pick_margins(logits, best, 99, 12) == 0.0)
    other = (best + 1) % 256
    m = P.categorical_pick_margins(logits, other, 99, 12)
    assert np.all(m > 0)
    e0 = e.copy()
    e0[24, 7] = 0.0                                              # the helpers' convention for a tape value that is not positive
    forced = best.copy()
    forced[24] = 7
    assert np.isinf(P.categorical_pick_margins(logits, forced, 99, 12, e0)[24])


def test_tape_layouts_of_the_stream():
    """_philox.tape: the (T, B, NZ) tape the in-kernel stream stands for, per output distribution (layout of wavenet_vocoder_amd/noise.py)"""
    from wavenet_vocoder_amd.noise import noise_width
    for kw in (dict(scalar_input=True, output_distribution="Logistic", out_channels=30),
               dict(scalar_input=True, output_distribution="Normal", out_channels=2),
               dict(scalar_input=True, output_distribution="Normal", out_channels=9),
               dict(scalar_input=False, output_distribution="Logistic", out_channels=256)):
        tp = P.tape(7, 33, 3, **kw)
        assert tp.dtype == np.float32 and tp.shape == (33, 3, noise_width(kw["scalar_input"], kw["output_distribution"], kw["out_channels"]))
        # a slice of a larger call: utterance b0 + b
        assert np.array_equal(P.tape(7, 33, 1, b0=2, **kw)[:, 0], tp[:, 2])
    mol = P.tape(7, 4096, 2, scalar_input=True, output_distribution="Logistic", out_channels=30)
    assert 1e-5 <= float(mol.min()) and float(mol.max()) <= 1.0 - 1e-5 + 1e-7 and abs(float(mol.mean()) - 0.5) < 5e-3
    # kind 0 is one fused multiply-add of the float32 uniform (constants as the ISA holds them: 0x3f7ffeb0, 0x3727c5ac)
    u = P.uniform01(7, np.arange(4096, dtype=np.uint64)[:, None, None], np.arange(2, dtype=np.uint64)[None, :, None],
                    np.arange(11, dtype=np.uint64)[None, None, :]).astype(np.float64)
    c, a = np.float64(np.float32(1.0) - np.float32(2e-5)), np.float64(np.float32(1e-5))
    assert np.float32(c).view(np.uint32) == 0x3F7FFEB0 and np.float32(a).view(np.uint32) == 0x3727C5AC
    assert np.array_equal(mol, (u * c + a).astype(np.float32))
    two_step = (np.float32(1e-5) + (u.astype(np.float32) * np.float32(c)).astype(np.float32)).astype(np.float32)
    assert 0.05 < float((two_step != mol).mean()) < 0.6          # (the unfused form differs by an ulp in a good part of the draws)
    gauss = P.tape(7, 8192, 2, scalar_input=True, output_distribution="Normal", out_channels=2)
    assert abs(float(gauss.mean())) < 0.03 and abs(float(gauss.std()) - 1.0) < 0.03
    cat = P.tape(99, 32, 13, scalar_input=False, out_channels=256)
    assert np.array_equal(cat[:, 12], P.exp_grid(99, 12, 32, 256).astype(np.float32))
    assert 0.0 < cat[24, 12, 144] < 1e-7 and float(cat.min()) > 0.0
