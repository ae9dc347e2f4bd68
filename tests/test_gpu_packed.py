"""-m gpu: PACKED SLOTS (continuous batching; wnv_generate_args.seg_start / seg_uid, sharding.synthesize_packed).

A row of a launch runs several utterances back to back.  The contract: every waveform is what the utterance gives on its own -- zero
history and the default first input at its first step (conv.py:34-36, wavenet.py:281-289), its own noise stream -- so a packed job
must reproduce, sample for sample, the same utterances run as one padded batch (whose rows are independent of each other: the
batch-member independence tests of tests/test_gpu_configs.py)."""
import pytest
import torch

from tests._configs import CONFIGS, build
from wavenet_vocoder_amd import sharding

pytestmark = pytest.mark.gpu
HOP = 256


def job(n, cin, seed, lo=2, hi=9):
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
    return [torch.randn(cin, f, generator=g) for f in frames]


@pytest.mark.parametrize("name,slots", [("cfg2_mol", 5), ("cfg3_gaussian", 8), ("cfg1_mulaw256", 3), ("cfg1b_mulaw256_intree", 4)])
def test_packed_slots_reproduce_the_padded_batch(name, slots):
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    eng = m._get_engine()
    mels = job(19, 80, 7)                                       # 19 utterances of 2-9 frames (512 ... 2304 samples)
    pad = kw["cin_pad"]
    lengths = [mm.shape[-1] * HOP for mm in mels]
    # reference run: ONE padded batch, row i = utterance i (in-kernel noise stream (step, i))
    c = sharding.pad_group(mels, pad).cuda()
    T = max(lengths)
    want, _, _ = eng.generate(B=len(mels), T=T, c_up=eng.upsample(c, T_expected=T), seed=99, kernel=2)
    st = {}
    got = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=pad, slots=slots, seed=99, stats=st)
    assert st["slots"] == slots and sum(st["utterances_per_slot"]) == len(mels) and max(st["utterances_per_slot"]) >= 3
    assert st["padding_loss"] < 0.25
    for i, (y, n) in enumerate(zip(got, lengths)):
        assert y.shape[-1] == n
        assert torch.equal(y, want[i, :, :n]), f"utterance {i} (length {n}) differs from its stand-alone waveform by {float((y - want[i, :, :n]).abs().max())}"
    m.to("cpu")


def test_a_long_job_runs_as_several_launches_with_the_same_waveforms():
    """``max_slot_steps`` bounds a launch (the slots' conditioning is resident): the split changes nothing an utterance can see."""
    m = build("cfg2_mol").to("cuda")
    mels = job(14, 80, 3, lo=6, hi=12)
    one = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5)
    st = {}
    many = sharding.synthesize_packed(m, mels, hop_size=HOP, cin_pad=2, slots=3, seed=5, max_slot_steps=6000, stats=st)
    assert len(st["launches"]) >= 2
    # (the conditioning is upsampled in padded groups of neighbours, as the reference's padded batches are: an utterance that is not the
    #  longest of its group sees zeros behind its last frame instead of its replicated edge -- the last cin_pad frames and the FIR
    #  half-widths of its conditioning depend on its neighbours; everything before that is equal, sample for sample)
    for a, b in zip(one, many):
        n = a.shape[-1] - 4 * HOP
        assert n > 0 and torch.equal(a[..., :n], b[..., :n])
    m.to("cpu")


def test_packed_slots_refuse_what_they_do_not_cover():
    m = build("cfg4_mol_multispeaker").to("cuda")                # a speaker embedding: one bias table per row
    with pytest.raises(NotImplementedError):
        sharding.synthesize_packed(m, job(4, 80, 1), hop_size=HOP, cin_pad=2)
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
