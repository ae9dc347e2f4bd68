"""The categorical sampler's arithmetic (csrc/wnv_sample.h: sample_categorical, the ring's categorical head) against the oracle's.

wavenet.py:332-335 is softmax -> OneHotCategorical(probs).sample(): Categorical renormalises, torch.multinomial takes
argmax(p_hat / e), e ~ Exp(1) (oracle.sample_categorical).  With x_k = exp(logit_k - max) that is
argmax_k ((x_k / s) / s2) / e_k -- s, s2 the two normalising sums, the SAME positive factors for every class.  Since round 4 the kernels
take argmax_k x_k / e_k.  This test restates both in float32 numpy and checks the claim the kernels rest on: the two picks differ only
where the top-2 margin of the choice (float64, log domain: tests/_margins.py's yardstick) is below the rounding of the quotients --
far inside what the GPU parity tests admit for a flip (1e-6 + twice the head-output difference)."""
import numpy as np
import pytest
import torch

from oracle.wavenet_oracle import sample_categorical as oracle_sample_categorical


def kernel_pick(logits, e):
    """sample_categorical, quantize and softmax: float32 throughout, first index among equals (np.argmax)"""
    x = np.exp((logits - logits.max(-1, keepdims=True)).astype(np.float32)).astype(np.float32)
    return np.argmax((x / e).astype(np.float32), -1)


def ring_head_pick(logits, e):
    """the ring kernel's categorical head since round 5 (run_head_cat, WNV_CAT_LOG): argmax_k logit_k - log e_k, float32"""
    return np.argmax((logits - np.log(e).astype(np.float32)).astype(np.float32), -1)


def oracle_pick(logits, e):
    p = torch.softmax(torch.from_numpy(logits), -1)                # wavenet.py:332
    return oracle_sample_categorical(p, torch.from_numpy(e)).numpy()


def margins(logits, e):
    s = torch.log_softmax(torch.from_numpy(logits).double(), -1) - torch.log(torch.from_numpy(e).double())
    top = s.topk(2, -1).values
    return (top[:, 0] - top[:, 1]).numpy()


@pytest.mark.parametrize("spread", [1.0, 4.0, 12.0])
@pytest.mark.parametrize("O", [256, 30])
def test_dropping_the_normalising_sums_moves_only_near_ties(spread, O):
    g = np.random.default_rng(1234 + O + int(spread))
    n = 200_000
    logits = (spread * g.standard_normal((n, O))).astype(np.float32)
    e = g.exponential(1.0, (n, O)).astype(np.float32)
    e = np.maximum(e, np.float32(1e-30))
    a, b = kernel_pick(logits, e), oracle_pick(logits, e)
    differ = a != b
    m = margins(logits, e)
    # every disagreement is a near tie, far below the 1e-6 the GPU tests admit
    assert differ.sum() <= 5, f"{int(differ.sum())} of {n} picks differ"
    if differ.any():
        assert float(m[differ].max()) < 5e-7, f"a pick differs at a top-2 margin of {float(m[differ].max()):.3e}"
    # ... and the margin distribution says how rare that is: a handful of steps per million sit below 1e-6
    assert (m < 1e-6).mean() < 1e-4


@pytest.mark.parametrize("spread", [1.0, 4.0, 12.0])
def test_the_log_domain_pick_of_the_ring_head_moves_only_near_ties(spread):
    """argmax_k exp(logit_k - max) / e_k = argmax_k logit_k - log e_k in exact arithmetic; in float32 the two (and the oracle's
    normalised form) may part only at a near tie of the choice."""
    g = np.random.default_rng(99 + int(spread))
    n, O = 200_000, 256
    logits = (spread * g.standard_normal((n, O))).astype(np.float32)
    e = np.maximum(g.exponential(1.0, (n, O)).astype(np.float32), np.float32(1e-30))
    a, b, k = ring_head_pick(logits, e), oracle_pick(logits, e), kernel_pick(logits, e)
    m = margins(logits, e)
    for other in (b, k):
        differ = a != other
        assert differ.sum() <= 5, f"{int(differ.sum())} of {n} picks differ"
        if differ.any():
            assert float(m[differ].max()) < 4e-6, f"a pick differs at a top-2 margin of {float(m[differ].max()):.3e}"


def test_common_factor_cannot_reorder_exactly_representable_cases():
    """scaling every quotient by the same power of two changes nothing at all: the picks agree exactly when s * s2 is one"""
    g = np.random.default_rng(7)
    logits = g.integers(-6, 1, (5000, 16)).astype(np.float32)      # max forced to 0 below
    logits[:, 0] = 0.0
    e = np.exp2(g.integers(-3, 4, (5000, 16))).astype(np.float32)
    x = np.exp(logits).astype(np.float32)
    for scale in (0.5, 0.25, 2.0):
        assert np.array_equal(np.argmax((x * np.float32(scale)) / e, -1), np.argmax(x / e, -1))
