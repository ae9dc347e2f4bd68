"""The numpy restatement of the in-kernel noise stream (tests/_philox.py) against the published known answers of Philox4x32-10, and the
one property of the uniform map the categorical pick has to know about: u can round to exactly 1.0."""
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
    key = (seed lo, seed hi), the uniform from the first word's top 24 bits."""
    src = (ROOT / "wavenet_vocoder_amd" / "csrc" / "wnv_dev.h").read_text()
    body = src[src.index("wnv_philox("):src.index("wnv_sigmoid")]
    for token in ("0xD2511F53u", "0xCD9E8D57u", "0x9E3779B9u", "0xBB67AE85u", "i < 10", "hi1 ^ c1 ^ k0", "hi0 ^ c3 ^ k1",
                  "(uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)b, (uint32_t)j", "(uint32_t)seed, (uint32_t)(seed >> 32)",
                  "(float)(x >> 8) + 0.5f", "1.0f / 16777216.0f", "return -logf(u)"):
        assert token in body, token
    assert re.search(r"const float u = wnv_u01\(r\[0\]\)", body)


def test_the_uniform_rounds_to_one_once_in_2_to_the_24():
    """((x >> 8) + 0.5) / 2^24 in float32: 16777215.5 is not representable and rounds to 2^24 -- u = 1.0, e = -log u = -0.0.  The event
    that showed it (round 5): seed 99, utterance 12, step 24, class 144 -- the one step at which the log-domain pick first differed from
    the quotient form in tests/test_gpu_packed.py (x / -0.0 = -inf against logit - log(-0.0) = +inf)."""
    assert np.float32(16777215.0) + np.float32(0.5) == np.float32(16777216.0)
    assert np.float32(16777214.0) + np.float32(0.5) == np.float32(16777214.0)          # (ties to even: every other top value stays below)
    w = int(P.first_word(99, 24, 12, 144))
    assert w >> 8 == 0xFFFFFF
    assert float(P.uniform01(99, 24, 12, 144)) == 1.0 and float(P.exp_noise(99, 24, 12, 144)) == 0.0
    # ... and nowhere else in that test's job (19 utterances of 512 ... 2304 steps, 256 classes)
    t = np.arange(2304, dtype=np.uint64)[:, None]
    j = np.arange(256, dtype=np.uint64)[None, :]
    hits = [(u, int(a), int(b)) for u in range(19) for a, b in np.argwhere(P.uniform01(99, t, np.uint64(u), j) >= 1.0)]
    assert hits == [(12, 24, 144)]
    u = P.uniform01(5, t, np.uint64(0), j)
    assert float(u.min()) > 0.0 and abs(float(u.mean()) - 0.5) < 2e-3
    e = P.exp_noise(5, t, np.uint64(0), j)
    assert abs(float(e.mean()) - 1.0) < 5e-3


def test_pick_margins_of_a_host_made_pick():
    """categorical_pick_margins: zero for the argmax of logit - log e, positive otherwise, a class with e = 0 is never the best"""
    g = np.random.default_rng(3)
    logits = g.standard_normal((256, 64))
    e = P.exp_noise(99, np.arange(64, dtype=np.uint64)[:, None], np.uint64(12), np.arange(256, dtype=np.uint64)[None, :])
    with np.errstate(divide="ignore"):
        score = np.where(e > 0, logits.T - np.log(np.where(e > 0, e, 1.0)), -np.inf)
    best = score.argmax(-1)
    assert best[24] != 144
    assert np.all(P.categorical_pick_margins(logits, best, 99, 12) == 0.0)
    other = (best + 1) % 256
    m = P.categorical_pick_margins(logits, other, 99, 12)
    assert np.all(m > 0)
    forced = best.copy()
    forced[24] = 144
    assert np.isinf(P.categorical_pick_margins(logits, forced, 99, 12)[24])


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
    assert cat[24, 12, 144] == 0.0 and np.signbit(cat[24, 12, 144]) and float(np.delete(cat.reshape(-1), (24 * 13 + 12) * 256 + 144).min()) > 0.0
