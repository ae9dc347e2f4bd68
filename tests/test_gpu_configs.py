"""-m gpu: the five BASELINE.json configurations at their real layer/channel sizes: HIP engine vs the CPU
oracle on the same seeded weights, mel and noise tape (teacher-forced parameters <= 1e-4; sampled classes
exact for the categorical models), plus size-independent properties at longer T where the oracle would be
too slow: prefix consistency, batch-member independence, determinism."""
import pytest
import torch

from oracle.wavenet_oracle import Oracle
from tests._configs import CONFIGS, build, inputs
from tests._golden import oracle_config
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie
from wavenet_vocoder_amd.noise import make_noise_tape

pytestmark = pytest.mark.gpu
TOL = 1e-4


def teacher(kw, B, T, seed=3):
    g = torch.Generator().manual_seed(seed)
    if kw.get("scalar_input", False):
        return torch.tanh(torch.randn(B, 1, T, generator=g) * 0.5)
    idx = torch.randint(0, kw["out_channels"], (B, T), generator=g)
    return torch.zeros(B, kw["out_channels"], T).scatter_(1, idx.unsqueeze(1), 1.0)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_teacher_forced_vs_oracle(name):
    kw = CONFIGS[name]
    B, T = 2, 256
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, gids = inputs(name, B, T)
    x = teacher(kw, B, T)
    scalar = kw.get("scalar_input", False)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(2))
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, g=gids, T=T, softmax=True, quantize=False,
                                          noise=tape, return_params=True)
    m = m.to("cuda")
    eng = m._get_engine()
    c_up = None if c is None else eng.upsample(c.cuda(), T_expected=T)
    tin = x.transpose(1, 2).contiguous().cuda()
    out, params, _ = eng.generate(B=B, T=T, c_up=c_up, g_ids=None if gids is None else gids[:, 0].cuda(),
                                  teacher=tin, noise=tape.cuda(), softmax=True, quantize=False,
                                  want_params=True, kernel=1)
    err = (params.cpu() - wparams).abs().max().item()
    assert err < TOL, f"{name}: head outputs differ by {err}"
    if scalar:
        assert_match_or_near_tie(out.cpu(), want, wparams, tape, kw, tol=TOL)
    else:
        assert (out.cpu() - want).abs().max().item() < TOL


@pytest.mark.parametrize("name", ["cfg0_mulaw256_small", "cfg2_mol", "cfg3_gaussian"])
def test_config_free_run_vs_oracle(name):
    kw = CONFIGS[name]
    B, T = (1, 192) if kw.get("cin_channels", -1) <= 0 else (2, 256)
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, gids = inputs(name, B, T)
    scalar = kw.get("scalar_input", False)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(4))
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(c=c, g=gids, T=T, noise=tape, return_params=True)
    m = m.to("cuda")
    eng = m._get_engine()
    c_up = None if c is None else eng.upsample(c.cuda(), T_expected=T)
    out, params, idx = eng.generate(B=B, T=T, c_up=c_up, noise=tape.cuda(), want_params=True,
                                    want_index=not scalar, kernel=1)
    # free running is chaotic in principle, but the only way two runs under one tape can part is a flipped
    # discrete choice (Gumbel pick / multinomial argmax) at a near tie: everything before that must agree, and a flip must be one
    if scalar:
        hz = assert_free_run_agrees_until_near_tie(out.cpu(), want, params.cpu(), wparams, tape, kw, what=name)
    else:
        hz = assert_free_run_agrees_until_near_tie(idx.cpu(), want.argmax(1), params.cpu(), wparams, tape, kw, what=name)
    print(f"{name}: free-run agreement horizon per utterance (of {T}): {hz}")
    assert min(hz) >= 32


@pytest.mark.parametrize("name", ["cfg2_mol", "cfg1_mulaw256"])
def test_properties_at_length(name):
    """Size-independent checks at a length the oracle cannot reach in seconds."""
    kw = CONFIGS[name]
    B, T = 4, 4096
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    scalar = kw.get("scalar_input", False)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(9)).cuda()
    full, _, _ = eng.generate(B=B, T=T, c_up=c_up, noise=tape, kernel=1)
    again, _, _ = eng.generate(B=B, T=T, c_up=c_up, noise=tape, kernel=1)
    assert torch.equal(full, again), "not deterministic"
    # prefix: generating fewer steps from the same inputs gives the same prefix
    T2 = 1024
    pre, _, _ = eng.generate(B=B, T=T2, c_up=c_up[:, :T2].contiguous(), noise=tape[:T2].contiguous(), kernel=1)
    assert torch.equal(pre, full[:, :, :T2])
    # batch members are independent: utterance 2 alone == utterance 2 in the batch
    solo, _, _ = eng.generate(B=1, T=T2, c_up=c_up[2:3, :T2].contiguous(), noise=tape[:T2, 2:3].contiguous(), kernel=1)
    assert torch.equal(solo[0], full[2, :, :T2])
    if scalar:
        assert float(full.abs().max()) <= 1.0 and float(full.std()) > 1e-3
    else:
        assert torch.equal(full.sum(1), torch.ones_like(full.sum(1)))      # exactly one class per step


def test_philox_mode_runs_and_is_seeded():
    m = build("cfg2_mol").to("cuda")
    eng = m._get_engine()
    c, _ = inputs("cfg2_mol", 2, 512)
    c_up = eng.upsample(c.cuda(), T_expected=512)
    a, _, _ = eng.generate(B=2, T=512, c_up=c_up, seed=123, kernel=1)
    b, _, _ = eng.generate(B=2, T=512, c_up=c_up, seed=123, kernel=1)
    d, _, _ = eng.generate(B=2, T=512, c_up=c_up, seed=124, kernel=1)
    assert torch.equal(a, b) and not torch.equal(a, d)
    assert float(a.abs().max()) <= 1.0 and torch.isfinite(a).all()


@pytest.mark.parametrize("net,cin_pad,scales,Tc", [("ConvInUpsampleNetwork", 2, [4, 4, 4, 4], 37), ("UpsampleNetwork", 2, [4, 4, 4, 4], 21),
                                                   ("ConvInUpsampleNetwork", 0, [16, 16], 9), ("ConvInUpsampleNetwork", 1, [2, 8, 4], 13)])
def test_upsampler_at_size_vs_oracle(net, cin_pad, scales, Tc):
    """The prologue at realistic sizes (the LDS-tiled time-major last stage, both networks, the trimmed `indent` path of the plain
    network, other scale sets) against the oracle's torch-CPU upsampler."""
    import numpy as np
    import wavenet_vocoder_amd as wnv
    hop = int(np.prod(scales))
    kw = dict(out_channels=30, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
              dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80, cin_pad=cin_pad,
              upsample_conditional_features=True, upsample_net=net,
              upsample_params=dict(upsample_scales=scales, **({"cin_channels": 80, "cin_pad": cin_pad} if net == "ConvInUpsampleNetwork" else {"cin_pad": cin_pad})))
    torch.manual_seed(4)
    m = wnv.WaveNet(**kw).eval()
    with torch.no_grad():                                       # the reference initialises the FIRs to constants: randomise them
        for name, p in m.named_parameters():
            if name.startswith("upsample_net"):
                p.copy_(torch.randn_like(p) * 0.3)
    o = Oracle(oracle_config(kw), m.state_dict())
    B = 3
    c = torch.randn(B, 80, Tc + 2 * cin_pad, generator=torch.Generator().manual_seed(5))
    want = o.upsample(c)                                        # (B, 80, T)
    T = Tc * hop
    assert want.shape == (B, 80, T)
    eng = m.to("cuda")._get_engine()
    got = eng.upsample(c.cuda(), T_expected=T)                  # (B, T, 80) time-major
    assert got.shape == (B, T, 80)
    err = (got.cpu().transpose(1, 2) - want).abs().max().item()
    assert err < 1e-5 * max(1.0, want.abs().max().item()), err          # values reach ~10 with the randomised filters
