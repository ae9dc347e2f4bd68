"""-m gpu: the HIP kernels against the REAL reference, run on this box.

``oracle/_ref`` is r9y9/wavenet_vocoder's own package, byte-compiled from /root/reference by ``oracle/build_ref.py`` (it travels with the
tree like the built .so).  Every case below runs the UNMODIFIED ``WaveNet.incremental_forward`` (wavenet.py:215-343) on this machine's
CPU -- teacher-forced, then free-running, its noise drawn by torch's own generator under a seed -- and the HIP engine on the same
weights, mel, speaker ids and the replayed tape:

  * every BASELINE.json configuration in both wordings (tests/_configs.py) on the kernel ``auto`` picks AND on the generic kernel;
  * the two compile-time instantiations of the ring kernel that other suites only check HIP-against-HIP: MODE 1 (more
    than four utterances per ring: 48 utterances of egs/mol, 40 of the mu-law model, cfg4 at 16) and MODE 2 (packed slots);
  * criteria: head outputs <= 1e-4 (the reference's own tolerance, tests/test_model.py:361-366); a forced sample may differ only at a
    near tie of the sampler's discrete choice; free runs part only through such a flip (tests/_margins.py).

A box with a GPU but without oracle/_ref FAILS here (ReferenceMissing), it does not skip.
"""
import pytest
import torch

from tests._configs import CONFIGS
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie
from tests._refrun import reference_case, reference_model, tape_replay_is_exact

pytestmark = pytest.mark.gpu
TOL = 1e-4
_CASES = {}


def case(name, B, Tt, T, seed=11):
    key = (name, B, Tt, T, seed)
    if key not in _CASES:
        _CASES.clear()                                     # one case resident at a time (cfg1 at B = 40: 2.6 MB of head outputs per 10 steps)
        _CASES[key] = reference_case(name, B, Tt, T, seed=seed)
    return _CASES[key]


def run_hip(d, kernel):
    m = d["model"].to("cuda")
    eng = m._get_engine()
    kw, B, T = d["kw"], d["B"], d["T"]
    scalar = kw.get("scalar_input", False)
    c_up = None if d["c"] is None else eng.upsample(d["c"].cuda(), T_expected=T)
    out, params, idx = eng.generate(B=B, T=T, c_up=c_up, g_ids=None if d["gids"] is None else d["gids"][:, 0].cuda(),
                                    teacher=d["x"].transpose(1, 2).contiguous().cuda(), noise=d["tape"].cuda(),
                                    want_params=True, want_index=not scalar, kernel=kernel)
    torch.cuda.synchronize()
    ran = eng.last_kernel()
    m.to("cpu")
    return out.cpu(), params.cpu(), None if idx is None else idx.cpu(), ran


def compare(d, out, params, idx, what):
    kw, Tt, T = d["kw"], d["Tt"], d["T"]
    scalar = kw.get("scalar_input", False)
    want, wparams, tape = d["want"], d["wparams"], d["tape"]
    err = float((params[:, :, :Tt] - wparams[:, :, :Tt]).abs().max())
    assert err < TOL, f"{what}: forced head outputs differ from the reference's by {err:.3e}"
    if scalar:
        got_s, want_s = out, want
    else:
        assert torch.equal(out.sum(1), torch.ones_like(out.sum(1))), f"{what}: not one-hot"
        got_s, want_s = idx.long(), want.argmax(1)
        assert torch.equal(out.argmax(1), got_s)
    n_bad = 0
    if Tt > 1:                                             # forced part: sample t follows from forced inputs only
        sl = (slice(None), slice(None), slice(0, Tt - 1)) if scalar else (slice(None), slice(0, Tt - 1))
        n_bad = assert_match_or_near_tie(got_s[sl], want_s[sl], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw, tol=TOL, what=f"{what}, forced part")
    hz = assert_free_run_agrees_until_near_tie(got_s, want_s, params, wparams, tape, kw, t0=max(Tt - 1, 0), what=f"{what}, free part")
    return err, n_bad, hz


def test_the_reference_is_on_this_box_and_the_tape_replays_its_draws():
    from oracle import reference as R
    ref = R.load_reference()                               # ReferenceMissing = FAIL: build oracle/_ref where /root/reference exists
    assert ref.receptive_field_size(24, 4, 3) == 505
    assert tape_replay_is_exact()


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("kernel", [0, 1])
def test_baseline_configs_forced_then_free_vs_reference(name, kernel):
    d = case(name, 2, 160, 256)
    out, params, idx, ran = run_hip(d, kernel)
    if kernel == 0:
        assert ran in (2, 3), f"{name}: auto chose kernel {ran}, expected a persistent kernel"
    err, n_bad, hz = compare(d, out, params, idx, f"{name} kernel {ran}")
    print(f"{name} kernel {ran}: forced head outputs vs the REFERENCE {err:.2e}, {n_bad} near-tie flips, free-run horizon {hz} of 256")
    assert min(hz) >= 160 + 16


@pytest.mark.parametrize("name", ["cfg2_mol", "cfg4_mol_multispeaker", "cfg1_mulaw256", "cfg0_mulaw256_small"])
def test_free_run_from_the_default_first_input_vs_reference(name):
    """No teacher input at all: the reference starts from zeros / one-hot 127 (wavenet.py:281-289); one forced step carrying exactly
    that input tells it the batch size (tests/_refrun.py::first_input)."""
    d = case(name, 3, 1, 256, seed=5)
    out, params, idx, ran = run_hip(d, 0)
    err, _, hz = compare(d, out, params, idx, f"{name} free run, kernel {ran}")
    print(f"{name}: free run from the default first input, horizon {hz} of 256 (step-0 head outputs {err:.2e})")
    assert min(hz) >= 32


@pytest.mark.parametrize("name,B,Tt,T", [("cfg2_mol", 48, 96, 256), ("cfg1_mulaw256", 40, 40, 256), ("cfg4_mol_multispeaker", 16, 160, 256),
                                         ("cfg3b_gaussian30", 14, 96, 256), ("cfg4_mol_multispeaker", 40, 40, 256)])
def test_ring_throughput_instantiation_vs_reference(name, B, Tt, T):
    """MODE 1 of wnv_ring_kernel (csrc/wnv_ring.hip: more than four utterances per ring -- six at 48 utterances, five at 40 --, K = 512
    from two per ring; the 30-layer wording runs 7 rings, two utterances each at 14; K = 512 at 40: five per ring) against the
    reference itself."""
    d = case(name, B, Tt, T, seed=13)
    out, params, idx, ran = run_hip(d, 2)
    assert ran == 2
    err, n_bad, hz = compare(d, out, params, idx, f"{name} B={B}")
    print(f"{name} B={B} (MODE 1): forced head outputs vs the REFERENCE {err:.2e}, {n_bad} near-tie flips, free-run horizon min {min(hz)} of {T}")
    assert min(hz) >= Tt + 8


def _teacher_from_own_output(kw, y):
    """The inputs an utterance's kernel run saw: its first input (zeros / one-hot 127), then its own samples (C, T) -> (1, C, T)."""
    if kw.get("scalar_input", False):
        first = torch.zeros(1, 1)
    else:
        first = torch.zeros(kw["out_channels"], 1)
        first[127] = 1.0
    return torch.cat([first, y[:, :-1]], dim=1).unsqueeze(0)


@pytest.mark.parametrize("name,slots", [("cfg2_mol", 3), ("cfg1_mulaw256", 2), ("cfg4_mol_multispeaker", 3)])
def test_packed_slots_vs_reference(name, slots):
    """MODE 2 (packed slots: several utterances back to back in one row, in-kernel noise).  The kernel draws its own noise there, so the
    check is autoregressive consistency against the reference: feed the REFERENCE, per utterance and on its own, the inputs the kernel
    saw (its first input, then the kernel's own samples) -- the reference's head outputs must be the kernel's at every step: zero
    history at the utterance's first step (conv.py:34-36), its own conditioning, its own speaker."""
    from oracle import reference as R
    from wavenet_vocoder_amd import sharding
    kw = CONFIGS[name]
    ours, rm = reference_model(name)
    g = torch.Generator().manual_seed(17)
    frames = [3, 1, 2, 2, 1, 3, 1]
    mels = [torch.randn(80, f, generator=g) for f in frames]
    spk = None
    if kw.get("gin_channels", -1) > 0:
        spk = torch.randint(0, kw["n_speakers"], (len(mels),), generator=g).tolist()
    m = ours.to("cuda")
    par, st = [], {}
    extra = {} if spk is None else {"speaker_ids": spk}
    outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=kw["cin_pad"], slots=slots, seed=77, stats=st, params_out=par, **extra)
    torch.cuda.synchronize()
    m.to("cpu")
    assert st["slots"] == slots and max(st["utterances_per_slot"]) >= 2
    torch.set_num_threads(8)
    worst = 0.0
    for i, (mel, y, p) in enumerate(zip(mels, outs, par)):
        y, p = y.cpu(), p.cpu()
        T = mel.shape[-1] * 256
        assert y.shape[-1] == T and torch.isfinite(y).all()
        c = sharding.pad_group([mel], kw["cin_pad"])
        gi = None if spk is None else torch.tensor([[spk[i]]])
        _, wparams = R.incremental(rm, seed=1, T=T, c=c, g=gi, test_inputs=_teacher_from_own_output(kw, y))
        err = float((p - wparams[0]).abs().max())
        worst = max(worst, err)
        assert err < TOL, f"{name}: utterance {i} ({T} samples, slot layout {st['utterances_per_slot']}): head outputs differ by {err:.3e}"
    print(f"{name}: {len(mels)} utterances in {slots} packed slots, head outputs vs the REFERENCE run per utterance: {worst:.2e}")
