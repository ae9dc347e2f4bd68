"""-m gpu: PACKED SLOTS (continuous batching; wnv_generate_args.seg_start / seg_uid, sharding.synthesize_packed).

A row of a launch runs several utterances back to back.  The contract: every waveform is what the utterance gives on its own -- zero
history and the default first input at its first step (conv.py:34-36, wavenet.py:281-289), its own noise stream -- so a packed job
must reproduce, sample for sample, the same utterances run as one padded batch (whose rows are independent of each other: the
batch-member independence tests of tests/test_gpu_configs.py)."""
import numpy as np
import pytest
import torch

from tests import _philox
from tests._configs import CONFIGS, build
from wavenet_vocoder_amd import sharding

pytestmark = pytest.mark.gpu
HOP = 256


def job(n, cin, seed, lo=2, hi=9):
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
    return [torch.randn(cin, f, generator=g) for f in frames]


def own_conditioning(eng, mels, pad):
    """(B, T_max, cin): row i = utterance i's conditioning upsampled on its own (sharding.upsample_each), zeros behind its end."""
    lengths = [mm.shape[-1] * HOP for mm in mels]
    c_up = torch.zeros(len(mels), max(lengths), mels[0].shape[0], device="cuda")
    for k, cu in sharding.upsample_each(eng, mels, list(range(len(mels))), pad, HOP):
        c_up[k, :lengths[k]] = cu
    return c_up


@pytest.mark.parametrize("name,slots", [("cfg2_mol", 5), ("cfg3_gaussian", 8), ("cfg1_mulaw256", 3), ("cfg1b_mulaw256_intree", 4),
                                        ("cfg4_mol_multispeaker", 4)])
def test_packed_slots_reproduce_the_padded_batch(name, slots):
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    eng = m._get_engine()
    mels = job(19, 80, 7)                                       # 19 utterances of 2-9 frames (512 ... 2304 samples)
    pad = kw["cin_pad"]
    lengths = [mm.shape[-1] * HOP for mm in mels]
    spk, extra, gids = None, {}, None
    if kw.get("gin_channels", -1) > 0:                          # (round 5) a speaker per utterance: BASELINE configs[4]
        spk = torch.randint(0, kw["n_speakers"], (len(mels),), generator=torch.Generator().manual_seed(3)).tolist()
        extra, gids = {"speaker_ids": spk}, torch.tensor(spk, dtype=torch.int64).cuda()
    # comparison run: ONE batch, row i = utterance i on its own conditioning and speaker (in-kernel noise stream (step, i))
    T = max(lengths)
    want, _, _ = eng.generate(B=len(mels), T=T, c_up=own_conditioning(eng, mels, pad), g_ids=gids, seed=99, kernel=2)
    st = {}
    got = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=pad, slots=slots, seed=99, stats=st, **extra)
    assert st["slots"] == slots and sum(st["utterances_per_slot"]) == len(mels) and max(st["utterances_per_slot"]) >= 3
    assert st["padding_loss"] < 0.25
    for i, (y, n) in enumerate(zip(got, lengths)):
        assert y.shape[-1] == n
        # (round 6: one-hot models too, to the last class -- the packed and the padded instantiations pick in the same form on
        #  bit-identical logits; round 5 accepted a near tie of the two best scores here)
        assert torch.equal(y, want[i, :, :n]), f"utterance {i} (length {n}) differs from its stand-alone waveform by {float((y - want[i, :, :n]).abs().max())}"
    m.to("cpu")


@pytest.mark.parametrize("name", ["cfg1_mulaw256", "cfg1b_mulaw256_intree"])
def test_in_kernel_picks_are_the_argmax_of_their_own_scores(name):
    """One-hot models under IN-KERNEL noise (no tape the reference could replay): every class the kernels pick must be the argmax of
    logit_k - log e_k over the head outputs they report and the noise stream restated on the host (tests/_philox.py: Philox4x32-10 pinned
    by its published known answers; counter = (step within the utterance, utterance id, class)) -- or lie within 1e-5 of it (float32
    rounding of the scores).  Packed slots, the ring kernel at one utterance per ring and the generic kernel (quotient form); utterance 12
    of this seed holds the draw with the smallest e the stream can give (u = 1 - 2^-24, e = 6e-8: that class wins by 16 in the log domain --
    until round 5 the uniform rounded to 1.0 there and the class could not be picked)."""
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    eng = m._get_engine()
    mels = job(19, 80, 7)
    pad = kw["cin_pad"]
    lengths = [mm.shape[-1] * HOP for mm in mels]
    par = []
    got = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=pad, slots=3, seed=99, params_out=par)
    noise = [_philox.exp_grid(99, i, n, kw["out_channels"]) for i, n in enumerate(lengths)]
    worst = 0.0
    for i, (y, p) in enumerate(zip(got, par)):
        margins = _philox.categorical_pick_margins(p.cpu().numpy(), y.argmax(0).cpu().numpy(), 99, i, noise[i])
        assert np.isfinite(margins).all() and float(margins.max()) < 1e-5, f"packed: utterance {i}, step {int(margins.argmax())}: {float(margins.max()):.3e}"
        worst = max(worst, float(margins.max()))
    assert int(got[12][:, 24].argmax()) == 144
    # the same utterances as rows of one batch: utterance id = row, step = t
    c_up = own_conditioning(eng, mels, pad)
    for kernel in (2, 1):
        out, params, index = eng.generate(B=len(mels), T=max(lengths), c_up=c_up, seed=99, kernel=kernel, want_params=True, want_index=True)
        assert torch.equal(out.argmax(1).to(torch.int32), index)
        for i, n in enumerate(lengths):
            margins = _philox.categorical_pick_margins(params[i, :, :n].cpu().numpy(), index[i, :n].cpu().numpy(), 99, i, noise[i])
            assert np.isfinite(margins).all() and float(margins.max()) < 1e-5, f"kernel {kernel}: utterance {i}, step {int(margins.argmax())}: {float(margins.max()):.3e}"
            worst = max(worst, float(margins.max()))
        assert int(index[12, 24]) == 144
    print(f"{name}: picks under in-kernel noise, largest gap to the best score {worst:.2e}")
    m.to("cpu")


def test_a_long_job_runs_as_several_launches_with_the_same_waveforms():
    """``max_slot_steps`` bounds a launch (the slots' conditioning is resident): the split changes nothing an utterance can see."""
    m = build("cfg2_mol").to("cuda")
    mels = job(14, 80, 3, lo=6, hi=12)
    one = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5)
    st = {}
    many = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5, max_slot_steps=6000, stats=st)
    assert len(st["launches"]) >= 2
    # (round 5: every utterance's conditioning is upsampled on its own -- no neighbour can reach it --, so the waveforms are equal to the
    #  last sample; until round 4 the tail of an utterance depended on the padded group it was upsampled in)
    for a, b in zip(one, many):
        assert torch.equal(a, b)
    # ... and the byte bound of a launch splits the same way (a 256-way one-hot output is 1 KB per slot-step)
    st2 = {}
    capped = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5, max_launch_bytes=3 * 6000 * st["step_bytes"], stats=st2)
    assert len(st2["launches"]) >= 2 and all(torch.equal(a, b) for a, b in zip(one, capped))
    # sink: results are handed over launch by launch instead of being kept
    seen = {}
    res = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5, max_slot_steps=6000, sink=lambda i, y: seen.__setitem__(i, y.clone()))
    assert all(r is None for r in res) and sorted(seen) == list(range(len(mels))) and all(torch.equal(seen[i], one[i]) for i in seen)
    m.to("cpu")


def test_packed_slots_refuse_what_they_do_not_cover():
    m = build("cfg4_mol_multispeaker").to("cuda")                # a speaker embedding: every utterance names its speaker
    with pytest.raises(ValueError):
        sharding.synthesize_packed(m, job(4, 80, 1), hop_size=HOP, cin_pad=2)
    with pytest.raises(IndexError):
        sharding.synthesize_packed(m, job(4, 80, 1), hop_size=HOP, cin_pad=2, speaker_ids=[0, 1, 7, 2])
    assert sharding.packed_unsupported_reason(build("wide_mol_512").to("cuda")) is not None      # the ring kernel does not take wide models
    with pytest.raises(NotImplementedError):
        sharding.synthesize_packed(build("wide_mol_512").to("cuda"), job(2, 80, 1), hop_size=HOP, cin_pad=2)
    m2 = build("cfg2_mol").to("cuda")
    eng = m2._get_engine()
    T, B = 512, 2
    c_up = torch.zeros(B, T, 80, device="cuda")
    seg = torch.zeros(B, T, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):                              # both arrays or none
        eng.generate(B=B, T=T, c_up=c_up, seed=1, seg_start=seg)
    with pytest.raises(ValueError):                              # in-kernel noise only
        eng.generate(B=B, T=T, c_up=c_up, noise=torch.rand(T, B, 11, device="cuda"), seg_start=seg, seg_uid=seg)
    with pytest.raises(NotImplementedError):                     # the ring kernel only
        eng.generate(B=B, T=T, c_up=c_up, seed=1, seg_start=seg, seg_uid=seg, kernel=1)
    with pytest.raises(ValueError):
        eng.generate(B=B, T=T, c_up=c_up, seed=1, seg_start=seg[:, :100].contiguous(), seg_uid=seg[:, :100].contiguous())


@pytest.mark.parametrize("kernel", [1])
def test_classes_only_output_of_one_hot_models(kernel):
    """``out = NULL`` (ABI 5): a one-hot model that samples classes can return them alone -- 4 bytes per sample instead of 4 out_channels --
    on the generic kernel and in packed-slot launches of the ring kernel (the test below)."""
    name = "cfg0_mulaw256_small"
    m = build(name).to("cuda")
    eng = m._get_engine()
    B, T = 3, 600
    full, _, idx_full = eng.generate(B=B, T=T, seed=21, kernel=kernel, want_index=True)
    none, _, idx_only = eng.generate(B=B, T=T, seed=21, kernel=kernel, want_index=True, want_out=False)
    assert none is None and torch.equal(idx_only, idx_full) and torch.equal(full.argmax(1).int(), idx_full)
    with pytest.raises(ValueError):
        eng.generate(B=B, T=T, seed=21, kernel=kernel, want_out=False)                       # classes only means index_out
    with pytest.raises(ValueError):
        eng.generate(B=B, T=T, seed=21, kernel=2, want_index=True, want_out=False)           # not in a plain ring launch
    m2 = build("cfg2_mol").to("cuda")
    with pytest.raises(ValueError):
        m2._get_engine().generate(B=1, T=256, c_up=torch.zeros(1, 256, 80, device="cuda"), seed=1, want_index=True, want_out=False)
    m.to("cpu")


def test_packed_one_hot_job_as_classes_and_through_the_post_chain():
    """``as_index``: the packed job of a one-hot model returns classes; they equal the argmax of the one-hot outputs of the same job, and
    the post-chain (synthesis.postprocess, C = 1 with "mulaw-quantize") decodes them to the waveform the one-hot route gives."""
    from types import SimpleNamespace
    from wavenet_vocoder_amd import synthesis
    name = "cfg1_mulaw256"
    m = build(name).to("cuda")
    mels = job(7, 80, 11, lo=2, hi=5)
    st_a, st_b = {}, {}
    onehot = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=8, stats=st_a)
    classes = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=8, stats=st_b, as_index=True)
    assert st_b["step_bytes"] * 5 < st_a["step_bytes"] * 2                                    # (2 x 80 + 2 + 2 against 2 x 80 + 256 + 2 floats per slot-step)
    hp = synthesis.default_hparams()
    hp.input_type, hp.quantize_channels = "mulaw-quantize", 256
    for a, b in zip(onehot, classes):
        assert b.shape == (1, a.shape[-1]) and torch.equal(a.argmax(0).float(), b[0])
        wa = synthesis.postprocess(a.unsqueeze(0), hp)
        wb = synthesis.postprocess(b.unsqueeze(0), hp)
        assert torch.equal(wa, wb)
    m.to("cpu")
