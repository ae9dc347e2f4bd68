"""Pin the CPU oracle (oracle/wavenet_oracle.py) to the real reference's outputs (tests/golden)."""
import json

import numpy as np
import pytest
import torch

from oracle.wavenet_oracle import (Oracle, _Layer, _QueueConv, fold_weight_norm,
                                   receptive_field_size)
from tests._golden import CASE_NAMES, Case, load_layers, oracle_config

torch.set_num_threads(1)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_fold_equals_make_generation_fast(name):
    c = Case(name)
    folded = fold_weight_norm(c.wn)          # (for wn-only cases c.fused is the PRODUCT's fold: the two folds must agree)
    assert set(folded) == set(c.fused)
    for k in folded:
        assert torch.allclose(folded[k].float(), c.fused[k].float(), atol=1e-6, rtol=1e-6), k


@pytest.mark.parametrize("name", CASE_NAMES)
@pytest.mark.parametrize("layout", ["wn", "fused"])
def test_teacher_forced(name, layout):
    c = Case(name)
    o = Oracle(oracle_config(c.kwargs), getattr(c, layout))
    scalar = c.kwargs.get("scalar_input", False)
    x = c.get("x")
    # trained-magnitude cases (tests/_stress.py): head outputs of 10-20 instead of O(1) -- the last-bit differences of the fused
    # layout's fold (g v / |v| evaluated once, up front) scale with them; relative tolerance 1e-6 of the magnitude on top
    k = 1.0 + (float(c.get("fwd").abs().max()) if c.stress else 0.0)
    fwd = o.forward(x, c=c.get("c_tf"), g=c.get("g_tf"), softmax=not scalar)
    assert torch.allclose(fwd, c.get("fwd"), atol=2e-6 * k), (fwd - c.get("fwd")).abs().max()
    y, p = o.incremental_forward(test_inputs=x, c=c.get("c_tf"), g=c.get("g_tf"), T=x.size(-1),
                                 softmax=True, quantize=False, noise=c.get("tf_tape"),
                                 return_params=True)
    if scalar:
        assert torch.allclose(p, c.get("tf_params"), atol=2e-6 * k), (p - c.get("tf_params")).abs().max()
    if c.stress and scalar:
        # (a sample is mean + exp(log_scale) * noise, clamped: the same function of the parameters; a Gumbel pick may flip at a near tie)
        from tests._margins import assert_match_or_near_tie
        assert_match_or_near_tie(y, c.get("tf_out"), c.get("tf_params"), c.get("tf_tape"), c.kwargs, tol=1e-4)
    else:
        assert torch.allclose(y, c.get("tf_out"), atol=5e-6 * k), (y - c.get("tf_out")).abs().max()


@pytest.mark.parametrize("name", CASE_NAMES)
def test_free_running_with_tape(name):
    c = Case(name)
    o = Oracle(oracle_config(c.kwargs), c.wn)
    want = c.get("fr_out")
    y = o.incremental_forward(initial_input=c.get("fr_init"), c=c.get("c_fr"), g=c.get("g_fr"),
                              T=want.size(-1), softmax=True, quantize=True, noise=c.get("fr_tape"))
    if c.kwargs.get("scalar_input", False):
        assert torch.allclose(y, want, atol=1e-5), (y - want).abs().max()
    else:
        assert torch.equal(y.argmax(1), want.argmax(1))
        assert torch.equal(y, want)


def test_default_start_is_index_127():
    c = Case("onehot_nocond")
    o = Oracle(oracle_config(c.kwargs), c.wn)
    want = c.get("fr0_out")
    y = o.incremental_forward(T=want.size(-1), noise=c.get("fr0_tape"))
    assert torch.equal(y, want)


@pytest.mark.parametrize("name", [n for n in CASE_NAMES if "upsample" in n])
def test_upsampler(name):
    c = Case(name)
    o = Oracle(oracle_config(c.kwargs), c.wn)
    got = o.upsample(c.get("c_tf"))
    assert torch.allclose(got, c.get("c_up_tf"), atol=1e-6)


def test_receptive_field_known_answers():
    z = load_layers()
    for (l, s, k), want in zip(z["rf/args"], z["rf/want"]):
        assert receptive_field_size(int(l), int(s), int(k)) == int(want)
    assert receptive_field_size(30, 1, 3, dilation=lambda x: 1) == 61
    assert [receptive_field_size(30, 3, 3), receptive_field_size(24, 4, 3)] == [6139, 505]


@pytest.mark.parametrize("tag", ["glu_cg", "glu_plain", "glu_k2"])
def test_layer_step(tag):
    z = load_layers()
    kw = json.loads(str(z[f"{tag}/kwargs"]))
    st = fold_weight_norm({k[len(tag) + 4:]: torch.from_numpy(v) for k, v in z.items()
                           if k.startswith(f"{tag}/wn/")})
    lay = _Layer(st, "", kw.get("dilation", 1), kw["kernel_size"])
    x = torch.from_numpy(z[f"{tag}/x"])
    c = torch.from_numpy(z[f"{tag}/c"]) if f"{tag}/c" in z else None
    g = torch.from_numpy(z[f"{tag}/g"]) if f"{tag}/g" in z else None
    xs, ss = [], []
    for t in range(x.size(1)):
        xo, so = lay.step(x[:, t:t + 1], None if c is None else c[:, t:t + 1],
                          None if g is None else g[:, t:t + 1])
        xs.append(xo)
        ss.append(so)
    assert np.allclose(torch.cat(xs, 1).numpy(), z[f"{tag}/x_out"], atol=2e-6)
    assert np.allclose(torch.cat(ss, 1).numpy(), z[f"{tag}/s_out"], atol=2e-6)


@pytest.mark.parametrize("tag", ["conv_d3", "conv_k1", "conv_k4"])
def test_queue_conv(tag):
    z = load_layers()
    ci, co, k, d = [int(v) for v in z[f"{tag}/meta"]]
    q = _QueueConv(torch.from_numpy(z[f"{tag}/weight"]), torch.from_numpy(z[f"{tag}/bias"]), d)
    x = torch.from_numpy(z[f"{tag}/x"])
    y = torch.cat([q.step(x[:, t:t + 1]) for t in range(x.size(1))], 1)
    assert np.allclose(y.numpy(), z[f"{tag}/y"], atol=2e-6)
