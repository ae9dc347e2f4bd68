"""-m gpu: the HIP engine (through the C ABI / the drop-in WaveNet) against the golden fixtures generated
from the real reference.  Tolerance 1e-4 on teacher-forced outputs and distribution parameters (the
tolerance the reference's own online==offline tests state, tests/test_model.py:361-366); sampled classes of
the categorical models must be EXACT under the shared noise tape."""
import json

import numpy as np
import pytest
import torch

import wavenet_vocoder_amd as wnv
from wavenet_vocoder_amd.conv import Conv1d
from wavenet_vocoder_amd.modules import ResidualConv1dGLU
from tests._golden import CASE_NAMES, Case, load_layers
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie
from tests._stress import close_enough

pytestmark = pytest.mark.gpu
TOL = 1e-4
# (case, kernel) pairs: 1 = the generic single-workgroup kernel covers every case; 2 = the pipelined ring kernel takes the cases
# of its geometry (residual 128 / gate 256 / skip 128: the reference-made ring_* fixtures)
# stress_*: trained-magnitude weights (tests/_stress.py; round 4) -- the two R128 cases on the ring kernel too, the wide one on
# the group ring (kernel 3)
CASE_KERNELS = [(n, 1) for n in CASE_NAMES] + [(n, 2) for n in CASE_NAMES if n.startswith("ring_") or n.endswith("_r128")] \
    + [(n, 3) for n in CASE_NAMES if n.startswith("stress_wide")]


def params_close(got, want, stress):
    """1e-4 absolute on the reference's random-init cases (tests/test_model.py:361-366); trained-magnitude cases: 1e-4 absolute
    + 1e-5 relative (head outputs reach 10-20 there; the reference's own online / offline gap is recorded in the fixture)."""
    if stress:
        ok, excess, worst = close_enough(got, want)
        assert ok, f"head outputs differ by {worst:.3e} (beyond 1e-4 + 1e-5 |x| by {excess:.3e})"
    else:
        err = (got - want).abs().max().item()
        assert err < TOL, f"distribution parameters differ by {err}"


def model_on_gpu(c, layout="wn"):
    m = wnv.WaveNet(**c.kwargs).eval()
    m.load_state_dict(getattr(c, layout))
    return m.to("cuda")


def cuda(t):
    return None if t is None else t.to("cuda")


@pytest.mark.parametrize("name,kernel", CASE_KERNELS)
def test_teacher_forced_matches_reference(name, kernel):
    c = Case(name)
    m = model_on_gpu(c)
    m.kernel, m.capture_params = kernel, True
    scalar = c.kwargs.get("scalar_input", False)
    x = c.get("x")
    torch.manual_seed(c.meta["seed"] + 1)      # replay mode: same noise the reference consumed
    y = m.incremental_forward(test_inputs=cuda(x), c=cuda(c.get("c_tf")), g=cuda(c.get("g_tf")),
                              T=x.size(-1), softmax=True, quantize=False)
    y = y.cpu()
    assert y.shape == c.get("tf_out").shape
    assert m._get_engine().last_kernel() == kernel
    if scalar:
        p = m.last_params.cpu()
        params_close(p, c.get("tf_params"), c.stress)
        # samples: same noise, so they agree -- except where the Gumbel-max pick is a near tie in the reference's own numbers
        assert_match_or_near_tie(y, c.get("tf_out"), c.get("tf_params"), c.get("tf_tape"), c.kwargs, tol=TOL)
        # batch forward of the reference == our incremental parameters (online == offline; trained-magnitude cases: the reference's
        # own two paths part by io/ref_online_offline_err there, which comes on top)
        if c.stress:
            gap = float(c.get("ref_online_offline_err")[0])
            d = (p.double() - c.get("fwd").double()).abs() - (1e-4 + 1e-5 * c.get("fwd").double().abs())
            assert float(d.max()) <= gap, (float(d.max()), gap)
        else:
            assert (p - c.get("fwd")).abs().max().item() < TOL
    else:
        err = (y - c.get("tf_out")).abs().max().item()
        assert err < TOL, f"probabilities differ by {err}"
        assert (y - c.get("fwd")).abs().max().item() < TOL


@pytest.mark.parametrize("name,kernel", CASE_KERNELS)
def test_free_running_with_shared_tape(name, kernel):
    c = Case(name)
    m = model_on_gpu(c, "fused")
    m.kernel, m.capture_params = kernel, True
    want = c.get("fr_out")
    torch.manual_seed(c.meta["seed"] + 2)
    y = m.incremental_forward(initial_input=cuda(c.get("fr_init")), c=cuda(c.get("c_fr")), g=cuda(c.get("g_fr")),
                              T=want.size(-1), softmax=True, quantize=True).cpu()
    assert y.shape == want.shape
    if c.kwargs.get("scalar_input", False) and c.stress:
        # trained magnitudes: the trajectories may part only through a flipped mixture pick at a near tie (tests/_margins.py)
        assert_free_run_agrees_until_near_tie(y, want, m.last_params.cpu(), c.get("fr_params"), c.get("fr_tape"), c.kwargs)
    elif c.kwargs.get("scalar_input", False):
        # free running is chaotic in principle; on these short horizons the trajectories must still agree
        err = (y - want).abs().max().item()
        assert err < 5e-4, err
        assert (m.last_params.cpu() - c.get("fr_params")).abs().max().item() < 5e-4
    else:
        assert torch.equal(y.argmax(1), want.argmax(1)), "sampled classes differ"
        assert torch.equal(y, want)


def test_default_start_is_index_127():
    c = Case("onehot_nocond")
    m = model_on_gpu(c)
    m.kernel = 1
    want = c.get("fr0_out")
    torch.manual_seed(c.meta["seed"] + 3)
    y = m.incremental_forward(T=want.size(-1)).cpu()
    assert torch.equal(y, want)


@pytest.mark.parametrize("name", [n for n in CASE_NAMES if "upsample" in n])
def test_upsample_kernel(name):
    c = Case(name)
    m = model_on_gpu(c)
    eng = m._get_engine()
    got = eng.upsample(cuda(c.get("c_tf"))).cpu()            # (B, T, cin) time-major
    want = c.get("c_up_tf").transpose(1, 2)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 1e-5
    with pytest.raises(AssertionError):
        eng.upsample(cuda(c.get("c_tf")), T_expected=want.shape[1] + 1)   # wavenet.py:276


def test_mixed_teacher_then_free_run():
    """test_inputs shorter than T: forced for the first steps, then free running (wavenet.py:255-258,297-301)."""
    c = Case("mol_upsample_convin")
    m = model_on_gpu(c)
    m.kernel = 1
    x = c.get("x")[:, :, :16]
    T = c.get("c_tf").shape[-1] - 2 * c.kwargs["cin_pad"]
    T *= int(np.prod(c.kwargs["upsample_params"]["upsample_scales"]))
    torch.manual_seed(5)
    y = m.incremental_forward(test_inputs=cuda(x), c=cuda(c.get("c_tf")), T=T).cpu()
    # oracle on the same tape
    from oracle.wavenet_oracle import Oracle
    from tests._golden import oracle_config
    from wavenet_vocoder_amd.noise import make_noise_tape
    torch.manual_seed(5)
    tape = make_noise_tape(T, x.shape[0], scalar_input=True, output_distribution="Logistic", out_channels=30)
    o = Oracle(oracle_config(c.kwargs), c.wn)
    want = o.incremental_forward(test_inputs=x, c=c.get("c_tf"), T=T, noise=tape)
    assert (y - want).abs().max().item() < 5e-4


@pytest.mark.parametrize("tag", ["glu_cg", "glu_plain", "glu_k2"])
def test_layer_level_glu(tag):
    z = load_layers()
    kw = json.loads(str(z[f"{tag}/kwargs"]))
    m = ResidualConv1dGLU(**kw).eval()
    m.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith(f"{tag}/wn/")})
    m = m.to("cuda")
    x = torch.from_numpy(z[f"{tag}/x"]).cuda()
    c = torch.from_numpy(z[f"{tag}/c"]).cuda() if f"{tag}/c" in z else None
    g = torch.from_numpy(z[f"{tag}/g"]).cuda() if f"{tag}/g" in z else None
    for rep in range(2):                                   # second pass checks clear_buffer()
        m.clear_buffer()
        xs, ss = [], []
        for t in range(x.size(1)):
            xo, so = m.incremental_forward(x[:, t:t + 1], None if c is None else c[:, t:t + 1],
                                           None if g is None else g[:, t:t + 1])
            assert xo.shape == (x.size(0), 1, kw["residual_channels"])
            xs.append(xo)
            ss.append(so)
        assert np.abs(torch.cat(xs, 1).cpu().numpy() - z[f"{tag}/x_out"]).max() < 2e-5
        assert np.abs(torch.cat(ss, 1).cpu().numpy() - z[f"{tag}/s_out"]).max() < 2e-5


@pytest.mark.parametrize("tag", ["conv_d3", "conv_k1", "conv_k4"])
def test_layer_level_queue_conv(tag):
    z = load_layers()
    ci, co, k, d = [int(v) for v in z[f"{tag}/meta"]]
    m = Conv1d(ci, co, k, dilation=d, padding=(k - 1) * d).eval()
    m.load_state_dict({"weight": torch.from_numpy(z[f"{tag}/weight"]), "bias": torch.from_numpy(z[f"{tag}/bias"])})
    m = m.to("cuda")
    x = torch.from_numpy(z[f"{tag}/x"]).cuda()
    for rep in range(2):
        m.clear_buffer()
        y = torch.cat([m.incremental_forward(x[:, t:t + 1]) for t in range(x.size(1))], 1)
        assert np.abs(y.cpu().numpy() - z[f"{tag}/y"]).max() < 2e-5
    m.train()
    with pytest.raises(RuntimeError, match="only supports eval mode"):
        m.incremental_forward(x[:, :1])
