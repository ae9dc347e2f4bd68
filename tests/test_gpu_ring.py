"""-m gpu: the pipelined ring kernel (csrc/wnv_ring.hip, kernel=2) against the CPU oracle, against the generic
single-workgroup kernel (kernel=1) on the same inputs, and through size-independent properties."""
import pytest
import torch

import wavenet_vocoder_amd as wnv
from oracle.wavenet_oracle import Oracle
from tests._testlib import needs_test_lib
from tests._configs import CONFIGS, build, inputs, tame_head_
from tests._golden import oracle_config
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie
from wavenet_vocoder_amd.noise import make_noise_tape

pytestmark = pytest.mark.gpu
TOL = 1e-4


def tape_for(kw, T, B, seed):
    return make_noise_tape(T, B, scalar_input=True, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(seed))


def run(eng, kernel, B, T, c_up=None, teacher=None, tape=None, g_ids=None, seed=0):
    return eng.generate(B=B, T=T, c_up=c_up, teacher=teacher, noise=tape, g_ids=g_ids, seed=seed,
                        want_params=True, kernel=kernel)


@pytest.mark.parametrize("name", ["cfg2_mol", "cfg3_gaussian"])
def test_ring_teacher_forced_vs_oracle(name):
    kw = CONFIGS[name]
    B, T = 3, 256
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, _ = inputs(name, B, T)
    x = torch.tanh(torch.randn(B, 1, T, generator=torch.Generator().manual_seed(3)) * 0.5)
    tape = tape_for(kw, T, B, 2)
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, T=T, noise=tape, return_params=True)
    m = m.to("cuda")
    eng = m._get_engine()
    c_up = eng.upsample(c.cuda(), T_expected=T)
    out, params, _ = run(eng, 2, B, T, c_up, x.transpose(1, 2).contiguous().cuda(), tape.cuda())
    err = (params.cpu() - wparams).abs().max().item()
    assert err < TOL, f"{name}: ring head outputs differ from the oracle by {err}"
    assert_match_or_near_tie(out.cpu(), want, wparams, tape, kw, tol=TOL)


@pytest.mark.parametrize("B", [1, 2, 5, 8, 11, 16, 24, 37, 100])     # (100: two launches of 64 + 36)
def test_ring_equals_generic_kernel(B):
    name = "cfg2_mol"
    kw = CONFIGS[name]
    T = 512
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = tape_for(kw, T, B, 7).cuda()
    x = torch.tanh(torch.randn(B, 128, 1, generator=torch.Generator().manual_seed(5)) * 0.5).cuda()   # 128 forced steps
    o1, p1, _ = run(eng, 1, B, T, c_up, x, tape)
    o2, p2, _ = run(eng, 2, B, T, c_up, x, tape)
    # teacher-forced part: same inputs, only the association order of the dot products differs
    assert (p1[:, :, :128] - p2[:, :, :128]).abs().max().item() < 2e-5
    # free-running part: trajectories stay together over this horizon
    d = (o1 - o2).abs()
    assert d.max().item() < 1e-3, d.max().item()
    # determinism
    o3, _, _ = run(eng, 2, B, T, c_up, x, tape)
    assert torch.equal(o2, o3)


@pytest.mark.parametrize("name,B", [("cfg2_mol", 8), ("cfg2_mol", 3), ("cfg3_gaussian", 5)])
@needs_test_lib                                             # (the knobs exist in libwnv_test.so only: the test re-runs itself there)
def test_split_rings_vs_oracle_and_generic(name, B, monkeypatch):
    """WNV_RING_SPLIT=1: two CUs per layer (run_stage_split; measured and not the default -- profiles/r03_ring_split_fine_timeline.txt).
    Teacher-forced head outputs against the oracle, then a free run against the generic kernel, and the layer-0-in-the-head switch
    (WNV_RING_L0=0: position 0 of every ring is a stage again) bit-compatible in the same sense."""
    kw = CONFIGS[name]
    T, Tt = 512, 256
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, _ = inputs(name, B, T)
    x = torch.tanh(torch.randn(B, 1, Tt, generator=torch.Generator().manual_seed(3)) * 0.5)
    tape = tape_for(kw, T, B, 2)
    torch.set_num_threads(8)
    _, wparams = o.incremental_forward(test_inputs=x, c=c, T=T, noise=tape, return_params=True)
    eng = m.to("cuda")._get_engine()
    c_up = eng.upsample(c.cuda(), T_expected=T)
    xt = x.transpose(1, 2).contiguous().cuda()
    ref, pref, _ = run(eng, 1, B, T, c_up, xt, tape.cuda())
    for env in ({"WNV_RING_SPLIT": "1"}, {"WNV_RING_L0": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out, params, _ = run(eng, 2, B, T, c_up, xt, tape.cuda())
        for k in env:
            monkeypatch.delenv(k)
        err = (params.cpu()[:, :, :Tt] - wparams[:, :, :Tt]).abs().max().item()
        assert err < TOL, (env, err)
        assert_free_run_agrees_until_near_tie(out.cpu(), ref.cpu(), params.cpu(), pref.cpu(), tape, kw, t0=Tt - 1, what=str(env))


def test_ring_30_layers_seven_rings_vs_oracle():
    """egs/gaussian as BASELINE.json words it (30 layers, 3 stacks: dilations up to 512): 8 x 31 ring workgroups + 30 tap
    workgroups do not fit 256 CUs, so 7 rings carry the 8 utterances (one ring pipelines two)."""
    kw = dict(out_channels=2, layers=30, stacks=3, residual_channels=128, gate_channels=256, skip_out_channels=128,
              kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Normal", cin_channels=80)
    torch.manual_seed(5)
    m = tame_head_(wnv.WaveNet(**kw).eval())
    o = Oracle(oracle_config(kw), m.state_dict())
    B, T = 8, 160
    g = torch.Generator().manual_seed(2)
    c_up = torch.randn(B, T, 80, generator=g)
    x = torch.tanh(torch.randn(B, 1, T, generator=g) * 0.5)
    tape = tape_for(kw, T, B, 6)
    torch.set_num_threads(8)
    _, wparams = o.incremental_forward(test_inputs=x, c=c_up.transpose(1, 2).contiguous(), T=T, noise=tape, return_params=True)
    eng = m.to("cuda")._get_engine()
    _, params, _ = run(eng, 2, B, T, c_up.cuda(), x.transpose(1, 2).contiguous().cuda(), tape.cuda())
    err = (params.cpu() - wparams).abs().max().item()
    assert err < TOL, err


def test_ring_free_run_vs_oracle():
    name = "cfg2_mol"
    kw = CONFIGS[name]
    B, T = 2, 256
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, _ = inputs(name, B, T)
    tape = tape_for(kw, T, B, 4)
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(c=c, T=T, noise=tape, return_params=True)
    m = m.to("cuda")
    eng = m._get_engine()
    c_up = eng.upsample(c.cuda(), T_expected=T)
    out, params, _ = run(eng, 2, B, T, c_up, None, tape.cuda())
    hz = assert_free_run_agrees_until_near_tie(out.cpu(), want, params.cpu(), wparams, tape, kw, what="ring free run")
    print(f"ring free-run agreement horizon per utterance (of {T}): {hz}")
    assert min(hz) >= 32


@pytest.mark.parametrize("variant", ["k2_s3", "nocond", "global", "k4"])
def test_ring_shape_variants_vs_generic(variant):
    base = dict(out_channels=30, residual_channels=128, gate_channels=256, skip_out_channels=128, dropout=0.0,
                scalar_input=True, output_distribution="Logistic")
    if variant == "k2_s3":
        kw = dict(layers=6, stacks=3, kernel_size=2, cin_channels=20, **base)
    elif variant == "nocond":
        kw = dict(layers=4, stacks=2, kernel_size=3, **base)
    elif variant == "global":
        kw = dict(layers=8, stacks=2, kernel_size=3, cin_channels=16, gin_channels=8, n_speakers=5,
                  use_speaker_embedding=True, **base)
    else:
        kw = dict(layers=4, stacks=1, kernel_size=4, cin_channels=80, **base)
        kw["out_channels"], kw["output_distribution"] = 2, "Normal"
    torch.manual_seed(11)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    B, T = 3, 300
    g = torch.Generator().manual_seed(1)
    cin = kw.get("cin_channels", -1)
    c_up = torch.randn(B, T, cin, generator=g).cuda() if cin > 0 else None
    gids = torch.randint(0, 5, (B,), generator=g).cuda() if kw.get("gin_channels", -1) > 0 else None
    tape = tape_for(kw, T, B, 3).cuda()
    x = torch.tanh(torch.randn(B, 200, 1, generator=g) * 0.5).cuda()
    o1, p1, _ = run(eng, 1, B, T, c_up, x, tape, gids)
    o2, p2, _ = run(eng, 2, B, T, c_up, x, tape, gids)
    assert (p1[:, :, :200] - p2[:, :, :200]).abs().max().item() < 2e-5
    assert (o1 - o2).abs().max().item() < 1e-3


def test_ring_properties_at_length():
    name = "cfg2_mol"
    kw = CONFIGS[name]
    B, T = 8, 8192
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = tape_for(kw, T, B, 9).cuda()
    full, _, _ = run(eng, 2, B, T, c_up, None, tape)
    T2 = 2048
    pre, _, _ = run(eng, 2, B, T2, c_up[:, :T2].contiguous(), None, tape[:T2].contiguous())
    assert torch.equal(pre, full[:, :, :T2]), "prefix property"
    solo, _, _ = run(eng, 2, 1, T2, c_up[5:6, :T2].contiguous(), None, tape[:T2, 5:6].contiguous())
    assert torch.equal(solo[0], full[5, :, :T2]), "batch members must be independent"
    assert torch.isfinite(full).all() and float(full.abs().max()) <= 1.0 and float(full.std()) > 1e-3


def test_unsupported_configs_say_so():
    kw = dict(CONFIGS["cfg2_mol"], residual_channels=256, gate_channels=512, skip_out_channels=256)      # wider than one CU's layer
    torch.manual_seed(0)
    eng = wnv.WaveNet(**kw).eval().to("cuda")._get_engine()
    cz = torch.zeros(1, 16, 80, device="cuda")
    with pytest.raises(NotImplementedError, match="ring kernel"):
        eng.generate(B=1, T=16, c_up=cz, kernel=2)
    out, _, _ = eng.generate(B=1, T=16, c_up=cz, kernel=0)            # auto: the group ring for wide models
    assert eng.last_kernel() == 3 and torch.isfinite(out).all()
    kw = dict(kw, skip_out_channels=512)                              # 512 skip channels on a wide model: the group ring since round 3
    eng = wnv.WaveNet(**kw).eval().to("cuda")._get_engine()
    out, _, _ = eng.generate(B=1, T=16, c_up=cz, kernel=0)
    assert eng.last_kernel() == 3 and torch.isfinite(out).all()
    kw = dict(kw, skip_out_channels=640)                              # beyond 512: neither persistent kernel
    eng = wnv.WaveNet(**kw).eval().to("cuda")._get_engine()
    with pytest.raises(NotImplementedError, match="group-ring kernel"):
        eng.generate(B=1, T=16, c_up=cz, kernel=3)
    out, _, _ = eng.generate(B=1, T=16, c_up=cz, kernel=0)            # auto: the generic kernel
    assert eng.last_kernel() == 1 and torch.isfinite(out).all()


# ---- models narrower than the kernel's 128 / 256 / 128 n geometry run zero-padded (exact: padded channels stay 0) ------------
NARROW = {
    "cfg0_mulaw256_small": CONFIGS["cfg0_mulaw256_small"],       # BASELINE cfg0: one-hot, R64 / G128 / K64, no conditioning
    "mol_r96_g160_k200": dict(out_channels=30, layers=6, stacks=3, residual_channels=96, gate_channels=160, skip_out_channels=200,
                              kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=24,
                              gin_channels=8, n_speakers=3, use_speaker_embedding=True),
    "gauss_r32_g64_k32_kw2": dict(out_channels=2, layers=4, stacks=2, residual_channels=32, gate_channels=64, skip_out_channels=32,
                                  kernel_size=2, dropout=0.0, scalar_input=True, output_distribution="Normal"),
}


@pytest.mark.parametrize("name", list(NARROW))
def test_ring_narrow_models_zero_padded_vs_oracle_and_generic(name):
    kw = NARROW[name]
    torch.manual_seed(17)
    m = tame_head_(wnv.WaveNet(**kw).eval())
    o = Oracle(oracle_config(kw), m.state_dict())
    scalar = kw.get("scalar_input", False)
    B, Tt, T = 3, 96, 192
    g = torch.Generator().manual_seed(2)
    cin, gin = kw.get("cin_channels", -1), kw.get("gin_channels", -1)
    c = torch.randn(B, cin, T, generator=g) if cin > 0 else None
    gids = torch.randint(0, kw["n_speakers"], (B, 1), generator=g) if gin > 0 else None
    from tests.test_gpu_configs import teacher
    x = teacher(kw, B, Tt)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(6))
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, g=gids, T=T, noise=tape, return_params=True)
    eng = m.to("cuda")._get_engine()
    res = {}
    for k in (1, 2):
        res[k] = eng.generate(B=B, T=T, c_up=None if c is None else c.transpose(1, 2).contiguous().cuda(),
                              g_ids=None if gids is None else gids[:, 0].cuda(), teacher=x.transpose(1, 2).contiguous().cuda(),
                              noise=tape.cuda(), want_params=True, want_index=not scalar, kernel=k)
        assert eng.last_kernel() == k
    out, params, idx = res[2]
    assert (params.cpu()[:, :, :Tt] - wparams[:, :, :Tt]).abs().max().item() < TOL            # ring vs oracle, forced part
    assert (params[:, :, :Tt] - res[1][1][:, :, :Tt]).abs().max().item() < 2e-5               # ring vs generic kernel
    if scalar:
        assert_match_or_near_tie(out.cpu()[:, :, :Tt - 1], want[:, :, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw, tol=TOL)
        assert_free_run_agrees_until_near_tie(out.cpu(), want, params.cpu(), wparams, tape, kw, t0=Tt - 1, what=name)
    else:
        assert_match_or_near_tie(idx.cpu()[:, :Tt - 1], want.argmax(1)[:, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw)
        assert_free_run_agrees_until_near_tie(idx.cpu(), want.argmax(1), params.cpu(), wparams, tape, kw, t0=Tt - 1, what=name)
    auto, _, _ = eng.generate(B=B, T=T, c_up=None if c is None else c.transpose(1, 2).contiguous().cuda(),
                              g_ids=None if gids is None else gids[:, 0].cuda(), noise=tape.cuda(), kernel=0)
    assert eng.last_kernel() == 2, "auto must pick the ring kernel for a model that fits its geometry after padding"


@needs_test_lib
def test_ring_slow_path_is_bit_identical(monkeypatch):
    """WNV_RING_FAST=0 forces the placement-independent write-through hand-offs; same arithmetic, same bits."""
    name = "cfg2_mol"
    kw = CONFIGS[name]
    B, T = 3, 512
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = tape_for(kw, T, B, 5).cuda()
    fast, pf, _ = run(eng, 2, B, T, c_up, None, tape)
    monkeypatch.setenv("WNV_RING_FAST", "0")
    slow, ps, _ = run(eng, 2, B, T, c_up, None, tape)
    assert torch.equal(fast, slow) and torch.equal(pf, ps)


def test_ring_back_to_back_launches_with_changing_shapes():
    """Mailbox tags are unique across launches (no re-zeroing): alternate B and T and compare with fresh engines."""
    name = "cfg2_mol"
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    eng = m._get_engine()
    outs = []

    def cond(B, T):          # conditioning at sample rate, any T (the upsampler is not what is tested here)
        return torch.randn(B, T, 80, generator=torch.Generator().manual_seed(100 * B + T)).cuda()

    for B, T in [(8, 300), (2, 700), (8, 300), (5, 129)]:
        outs.append(run(eng, 2, B, T, cond(B, T), None, tape_for(kw, T, B, 11).cuda())[0])
    assert torch.equal(outs[0], outs[2])
    eng2 = build(name).to("cuda")._get_engine()
    ref = run(eng2, 2, 5, 129, cond(5, 129), None, tape_for(kw, 129, 5, 11).cuda())[0]
    assert torch.equal(outs[3], ref)


# ---- one-hot (mu-law categorical) models on the ring kernel -----------------------------------------------------------
ONEHOT = dict(out_channels=256, layers=8, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128,
              kernel_size=3, dropout=0.0, cin_channels=80)


def onehot_model(seed=21, **over):
    kw = dict(ONEHOT, **over)
    torch.manual_seed(seed)
    return kw, tame_head_(wnv.WaveNet(**kw).eval())


def cat_tape(T, B, seed):
    return make_noise_tape(T, B, scalar_input=False, output_distribution="Logistic", out_channels=256,
                           generator=torch.Generator().manual_seed(seed))


def test_ring_onehot_teacher_forced_vs_oracle_and_generic():
    kw, m = onehot_model()
    o = Oracle(oracle_config(kw), m.state_dict())
    B, T = 3, 128
    g = torch.Generator().manual_seed(4)
    c_up = torch.randn(B, T, 80, generator=g)
    idx = torch.randint(0, 256, (B, T), generator=g)
    x = torch.zeros(B, 256, T).scatter_(1, idx.unsqueeze(1), 1.0)
    tape = cat_tape(T, B, 9)
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c_up.transpose(1, 2).contiguous(), T=T, softmax=True,
                                          quantize=False, noise=tape, return_params=True)
    eng = m.to("cuda")._get_engine()
    tin = x.transpose(1, 2).contiguous().cuda()
    res = {}
    for k in (1, 2):
        res[k] = eng.generate(B=B, T=T, c_up=c_up.cuda(), teacher=tin, noise=tape.cuda(), softmax=True, quantize=False,
                              want_params=True, kernel=k)
    assert (res[2][1].cpu() - wparams).abs().max().item() < TOL            # head outputs vs the oracle
    assert (res[2][0].cpu() - want).abs().max().item() < TOL               # probabilities vs the oracle
    assert (res[2][1] - res[1][1]).abs().max().item() < 2e-5               # vs the generic kernel


def test_ring_onehot_free_run_sampled_classes():
    kw, m = onehot_model(seed=22)
    o = Oracle(oracle_config(kw), m.state_dict())
    B, T = 2, 160
    c_up = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(5))
    tape = cat_tape(T, B, 10)
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(c=c_up.transpose(1, 2).contiguous(), T=T, noise=tape, return_params=True)      # (B, 256, T) one-hot
    eng = m.to("cuda")._get_engine()
    out, params, idx = eng.generate(B=B, T=T, c_up=c_up.cuda(), noise=tape.cuda(), want_index=True, want_params=True, kernel=2)
    out1, _, idx1 = eng.generate(B=B, T=T, c_up=c_up.cuda(), noise=tape.cuda(), want_index=True, kernel=1)
    assert torch.equal(out.sum(1), torch.ones(B, T, device="cuda")) and torch.equal(out.argmax(1).int(), idx.int())
    hz = assert_free_run_agrees_until_near_tie(idx.cpu(), want.argmax(1), params.cpu(), wparams, tape, kw, what="ring one-hot free run")
    print(f"ring one-hot free run: sampled classes equal the oracle's until a near tie; horizon per utterance (of {T}): {hz}")
    assert min(hz) >= 32                                                    # exact classes until an argmax tie flips
    assert (idx == idx1).float().mean().item() > 0.6                        # and largely the generic kernel's trajectory


def test_ring_mulaw256_intree_preset_properties():
    """egs/mulaw256 as shipped (30 layers, 3 stacks, 80-mel upsampling): 7 rings + categorical head; properties at length."""
    name = "cfg1b_mulaw256_intree"
    m = build(name).to("cuda")
    eng = m._get_engine()
    B, T = 8, 2048
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = cat_tape(T, B, 12).cuda()
    full, _, idx = eng.generate(B=B, T=T, c_up=c_up, noise=tape, want_index=True, kernel=2)
    assert torch.equal(full.sum(1), torch.ones(B, T, device="cuda"))
    again, _, _ = eng.generate(B=B, T=T, c_up=c_up, noise=tape, want_index=True, kernel=2)
    assert torch.equal(full, again), "determinism"
    T2 = 512
    pre, _, _ = eng.generate(B=B, T=T2, c_up=c_up[:, :T2].contiguous(), noise=tape[:T2].contiguous(), kernel=2)
    assert torch.equal(pre, full[:, :, :T2]), "prefix property"
    solo, _, _ = eng.generate(B=1, T=T2, c_up=c_up[3:4, :T2].contiguous(), noise=tape[:T2, 3:4].contiguous(), kernel=2)
    assert torch.equal(solo[0], full[3, :, :T2]), "batch members must be independent"
    gen, _, idx1 = eng.generate(B=B, T=256, c_up=c_up[:, :256].contiguous(), noise=tape[:256].contiguous(), want_index=True, kernel=1)
    assert (idx[:, :256] == idx1).float().mean().item() > 0.5


# ---- skip_out_channels 256 / 512: several skip passes per stage, one head workgroup per 128 hidden units -----------------
@pytest.mark.parametrize("name", ["cfg1_mulaw256", "cfg4_mol_multispeaker"])
def test_ring_wide_skip_baseline_configs_vs_oracle(name):
    """BASELINE.json's cfg1 (one-hot, 256 skip channels: two head parts) and cfg4 (MoL + speaker embedding, 512 skip channels:
    four head parts, two skip passes streamed) at full size: teacher-forced head outputs against the oracle."""
    from tests.test_gpu_configs import teacher
    kw = CONFIGS[name]
    B, T = 3, 256
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    c, gids = inputs(name, B, T)
    x = teacher(kw, B, T)
    scalar = kw.get("scalar_input", False)
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(2))
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, g=gids, T=T, softmax=True, quantize=False, noise=tape,
                                          return_params=True)
    eng = m.to("cuda")._get_engine()
    c_up = eng.upsample(c.cuda(), T_expected=T)
    gi = None if gids is None else gids[:, 0].cuda()
    res = {}
    for k in (1, 2):
        res[k] = eng.generate(B=B, T=T, c_up=c_up, g_ids=gi, teacher=x.transpose(1, 2).contiguous().cuda(), noise=tape.cuda(),
                              softmax=True, quantize=False, want_params=True, kernel=k)
    err = (res[2][1].cpu() - wparams).abs().max().item()
    assert err < TOL, f"{name}: ring head outputs differ from the oracle by {err}"
    assert (res[2][1] - res[1][1]).abs().max().item() < 3e-5               # vs the generic kernel
    if scalar:
        assert_match_or_near_tie(res[2][0].cpu(), want, wparams, tape, kw, tol=TOL)
    else:
        assert (res[2][0].cpu() - want).abs().max().item() < TOL           # probabilities


@pytest.mark.parametrize("K,B", [(256, 1), (256, 11), (512, 2), (512, 8), (512, 16), (512, 24)])
def test_ring_wide_skip_equals_generic_kernel(K, B):
    kw = dict(out_channels=30, layers=6, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=K,
              kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80)
    torch.manual_seed(31)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    T = 384
    g = torch.Generator().manual_seed(B)
    c_up = torch.randn(B, T, 80, generator=g).cuda()
    tape = tape_for(kw, T, B, 3).cuda()
    x = torch.tanh(torch.randn(B, 128, 1, generator=g) * 0.5).cuda()
    o1, p1, _ = run(eng, 1, B, T, c_up, x, tape)
    o2, p2, _ = run(eng, 2, B, T, c_up, x, tape)
    assert (p1[:, :, :128] - p2[:, :, :128]).abs().max().item() < 3e-5
    assert (o1 - o2).abs().max().item() < 1e-3
    o3, _, _ = run(eng, 2, B, T, c_up, x, tape)
    assert torch.equal(o2, o3)


def test_ring_wide_skip_onehot_free_run_and_properties():
    """cfg1 at full size, free running: sampled classes against the generic kernel, determinism, prefix, independence."""
    name = "cfg1_mulaw256"
    m = build(name).to("cuda")
    eng = m._get_engine()
    B, T = 4, 1024
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = cat_tape(T, B, 12).cuda()
    full, _, idx = eng.generate(B=B, T=T, c_up=c_up, noise=tape, want_index=True, kernel=2)
    assert torch.equal(full.sum(1), torch.ones(B, T, device="cuda"))
    again, _, _ = eng.generate(B=B, T=T, c_up=c_up, noise=tape, want_index=True, kernel=2)
    assert torch.equal(full, again), "determinism"
    T2 = 256
    pre, _, _ = eng.generate(B=B, T=T2, c_up=c_up[:, :T2].contiguous(), noise=tape[:T2].contiguous(), kernel=2)
    assert torch.equal(pre, full[:, :, :T2]), "prefix property"
    solo, _, _ = eng.generate(B=1, T=T2, c_up=c_up[3:4, :T2].contiguous(), noise=tape[:T2, 3:4].contiguous(), kernel=2)
    assert torch.equal(solo[0], full[3, :, :T2]), "batch members must be independent"
    _, _, idx1 = eng.generate(B=B, T=T2, c_up=c_up[:, :T2].contiguous(), noise=tape[:T2].contiguous(), want_index=True, kernel=1)
    assert (idx[:, :T2] == idx1).float().mean().item() > 0.5
