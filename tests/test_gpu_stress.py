"""-m gpu: parity at TRAINED-LIKE magnitudes (tests/_stress.py).

Every other parity test runs the reference's Kaiming initialisation, where pre-activations are O(1) and no gate saturates.  Here
the BASELINE configurations at full depth get weights whose gate pre-activations reach |z| of 10-40 (a third of the gate outputs
beyond 0.99), biases of +-2, log-scale biases in [-14, -5], mel frames with outliers up to +-6 and teacher samples on the rails, at
gain 2, 4 and 8 -- on the ring kernel, the generic kernel, the group ring and the batch `forward` kernels, against the oracle
(which the reference-made stress_* fixtures of tests/golden pin at these magnitudes: tests/test_oracle_golden.py).

Two families (tests/_stress.py):
  * mode "drive" -- gain 4, 8, 16 on what drives the gates of a trained vocoder (conditioning 1x1s, biases): gate pre-activations up
    to +-55 on a WELL-CONDITIONED map (ATen's f32 answer is within 1e-5 of the float64 one).  STRICT criteria: head outputs <= 1e-4
    absolute + 1e-5 relative against the oracle; a forced step's sample may differ only at a near tie of the sampler's discrete
    choice, a free run may part only through such a flip (tests/_margins.py); one-hot models: classes exact or near tie.  This is
    the test of v_exp_f32 / v_rcp_f32 in the ring kernel's gate on saturated arguments, of the M / N fold and of the skip-sum order.
  * mode "all" -- gain 2, 4, 8 on every weight_g (what VERDICT r03 asked for), which makes the 24-layer MAP ITSELF ill-conditioned:
    a perturbation grows ~gain-fold per layer, ATen's own f32 answer is 5e-4 (gain 4) and 1.5 (gain 8) away from the float64 one
    and its online and offline paths part by as much (measured: DESIGN.md section 3).  No f32 implementation can meet 1e-4 there,
    the reference included -- so the yardstick is the EXACT answer: the HIP path must be no further from the float64 oracle than
    4 x the reference arithmetic's own distance (or within the strict bound); sample-level criteria apply where the head outputs are
    within the strict bound.  The worst-case term is named by the measurement: it is the conditioning of the model -- the three
    kernel families (libm gate in the generic kernel, v_exp / v_rcp + folded M / N in the ring kernels, MFMA batch forward) land
    within a factor of 2 of each other and of ATen."""
import functools

import pytest
import torch

import wavenet_vocoder_amd as wnv
from oracle.wavenet_oracle import Oracle
from tests._configs import CONFIGS, MEL, build, inputs
from tests._golden import oracle_config
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie, choice_margin
from tests._stress import apply_stress_, close_enough, stress_mel, stress_teacher
from wavenet_vocoder_amd.noise import make_noise_tape

pytestmark = pytest.mark.gpu

WIDE8 = dict(out_channels=30, layers=8, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=256, kernel_size=3,
             dropout=0.0, scalar_input=True, output_distribution="Logistic", **MEL)
GAINS = [2.0, 4.0, 8.0]


def onehot_teacher(C, B, T, seed):
    idx = torch.randint(0, C, (B, T), generator=torch.Generator().manual_seed(seed))
    return torch.zeros(B, C, T).scatter_(1, idx.unsqueeze(1), 1.0)


@functools.lru_cache(maxsize=None)
def case(name, gain, mode="all"):
    """Model with trained-magnitude weights, stressed inputs, and the oracle's answers: Tt forced steps then free running to T.
    Also the EXACT head outputs of the forced part (float64 evaluation of the same f32 weights) and ATen's own distance to them."""
    if name == "wide8":
        kw, B, Tt, T = WIDE8, 2, 192, 256
        torch.manual_seed(0)
        m = wnv.WaveNet(**kw).eval()
    else:
        kw = CONFIGS[name]
        B, Tt, T = (3, 384, 512)
        m = build(name)
    apply_stress_(m, dict(gain=gain, seed=100 + int(gain), mode=mode))
    scalar = kw.get("scalar_input", False)
    _, gids = inputs(name, B, T) if name in CONFIGS else (None, None)
    c = stress_mel((B, 80, T // 256 + 2 * kw["cin_pad"]), 11 + int(gain))
    x = stress_teacher(B, Tt, 3) if scalar else onehot_teacher(kw["out_channels"], B, Tt, 3)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(2))
    o = Oracle(oracle_config(kw), m.state_dict())
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, g=gids, T=T, softmax=True, quantize=True, noise=tape, return_params=True)
    # the exact answer for the forced part: the batch forward in float64 on the f32 oracle's own upsampled conditioning
    o64 = Oracle(oracle_config(kw), m.state_dict(), dtype=torch.float64)
    c_up = o.upsample(c)[:, :, :Tt]
    for oo in (o, o64):
        oo.cfg.upsample_conditional_features = False
    try:
        truth = o64.forward(x, c=c_up, g=gids)
        wfwd = o.forward(x, c=c_up, g=gids)
    finally:
        o.cfg.upsample_conditional_features = True
    e_ref = float((wparams[:, :, :Tt].double() - truth).abs().max())
    e_ref_fwd = float((wfwd.double() - truth).abs().max())
    return dict(kw=kw, m=m, B=B, Tt=Tt, T=T, c=c, gids=gids, x=x, tape=tape, want=want, wparams=wparams, o=o, truth=truth, wfwd=wfwd,
                e_ref=e_ref, e_ref_fwd=e_ref_fwd, mode=mode)


def run_incremental(d, kernel):
    m = d["m"].to("cuda")
    eng = m._get_engine()
    c_up = eng.upsample(d["c"].cuda(), T_expected=d["T"])
    gi = None if d["gids"] is None else d["gids"][:, 0].cuda()
    out, params, _ = eng.generate(B=d["B"], T=d["T"], c_up=c_up, g_ids=gi, teacher=d["x"].transpose(1, 2).contiguous().cuda(),
                                  noise=d["tape"].cuda(), softmax=True, quantize=True, want_params=True, kernel=kernel)
    assert eng.last_kernel() == kernel
    return out.cpu(), params.cpu()


def check(d, out, params, what):
    kw, Tt, T = d["kw"], d["Tt"], d["T"]
    want, wparams, tape = d["want"], d["wparams"], d["tape"]
    strict, excess, worst = close_enough(params[:, :, :Tt], wparams[:, :, :Tt])
    e_hip = float((params[:, :, :Tt].double() - d["truth"]).abs().max())
    mag = float(wparams[:, :, :Tt].abs().max())
    print(f"{what}: forced head outputs vs oracle {worst:.2e}; distance to the float64 answer: HIP {e_hip:.2e}, ATen f32 {d['e_ref']:.2e} "
          f"(magnitudes up to {mag:.1f})")
    if d["mode"] == "drive":
        assert strict, f"{what}: head outputs differ by {worst:.3e}, {excess:.3e} beyond 1e-4 + 1e-5 |x| (magnitudes up to {mag:.1f})"
    else:
        assert strict or e_hip <= 4.0 * d["e_ref"], (f"{what}: {e_hip:.3e} from the exact answer where the reference arithmetic itself is "
                                                     f"{d['e_ref']:.3e} away (and {worst:.3e} from the f32 oracle)")
    if not strict:
        return                              # (ill-conditioned model: samples follow head outputs that legitimately differ)
    if kw.get("scalar_input", False):
        assert_match_or_near_tie(out[:, :, :Tt - 1], want[:, :, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw, what=f"{what}, forced part")
        assert_free_run_agrees_until_near_tie(out, want, params, wparams, tape, kw, t0=Tt - 1, tol=2e-3, what=f"{what}, free part")
    else:
        got_i, want_i = out.argmax(1), want.argmax(1)
        assert_match_or_near_tie(got_i[:, :Tt - 1], want_i[:, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw, what=f"{what}, forced classes")
        assert_free_run_agrees_until_near_tie(got_i, want_i, params, wparams, tape, kw, t0=Tt - 1, what=f"{what}, free classes")


MODE_GAINS = [("drive", 4.0), ("drive", 8.0), ("drive", 16.0), ("all", 2.0), ("all", 4.0), ("all", 8.0)]


@pytest.mark.parametrize("mode,gain", MODE_GAINS)
@pytest.mark.parametrize("kernel", [2, 1])
@pytest.mark.parametrize("name", ["cfg2_mol", "cfg1_mulaw256", "cfg4_mol_multispeaker"])
def test_trained_magnitudes_sample_loop_vs_oracle(name, kernel, mode, gain):
    d = case(name, gain, mode)
    out, params = run_incremental(d, kernel)
    check(d, out, params, f"{name} {mode} gain {gain:g} kernel {kernel}")
    d["m"].to("cpu")


@pytest.mark.parametrize("mode,gain", MODE_GAINS)
def test_trained_magnitudes_group_ring_vs_oracle(mode, gain):
    d = case("wide8", gain, mode)
    out, params = run_incremental(d, 3)
    check(d, out, params, f"wide 512/512/256 {mode} gain {gain:g} group ring")
    d["m"].to("cpu")


@pytest.mark.parametrize("mode,gain", MODE_GAINS)
@pytest.mark.parametrize("name", ["cfg2_mol", "cfg1_mulaw256", "cfg4_mol_multispeaker"])
def test_trained_magnitudes_batch_forward_vs_oracle(name, mode, gain):
    """f3 (wnv_forward, f32 MFMA) on the forced part: against the oracle's batch forward; ill-conditioned family: against the exact
    answer with ATen's own distance as the yardstick."""
    d = case(name, gain, mode)
    Tt, T = d["Tt"], d["T"]
    m = d["m"].to("cuda")
    eng = m._get_engine()
    c_up = eng.upsample(d["c"].cuda(), T_expected=T)[:, :Tt].contiguous()
    gi = None if d["gids"] is None else d["gids"][:, 0].cuda()
    y = eng.forward(d["x"].cuda(), c_up=c_up, g_ids=gi).cpu()
    strict, excess, worst = close_enough(y, d["wfwd"])
    e_hip = float((y.double() - d["truth"]).abs().max())
    print(f"{name} {mode} gain {gain:g}: forward vs oracle forward {worst:.2e}; distance to the float64 answer: HIP {e_hip:.2e}, ATen f32 {d['e_ref_fwd']:.2e}")
    if mode == "drive":
        assert strict, f"{name} gain {gain:g}: batch forward differs from the oracle's by {worst:.3e} ({excess:.3e} beyond the bound)"
        ok2, excess2, worst2 = close_enough(y, d["wparams"][:, :, :Tt], atol=2e-4, rtol=2e-5)      # online == offline: two roundings apart
        assert ok2, f"{name} gain {gain:g}: offline vs the oracle's online head outputs {worst2:.3e} ({excess2:.3e} beyond 2e-4 + 2e-5 |x|)"
    else:
        assert strict or e_hip <= 4.0 * d["e_ref_fwd"], (e_hip, d["e_ref_fwd"], worst)
    m.to("cpu")
