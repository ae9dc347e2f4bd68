"""-m gpu: ONE SEED, ONE WAVEFORM -- whatever batch an utterance runs in (round 6).

Under in-kernel noise (``rng = "philox"``: what bench.py times) value j of (utterance u, step t) is a function of (seed, u, t, j) alone
(include/wnv.h, ``wnv_generate_args.seed``), the rows of a launch never see each other, and since round 6 every instantiation of the
ring kernel draws the SAME uniform (exact in float32, never 1.0: csrc/wnv_dev.h ``wnv_u01``) and picks a one-hot class in the SAME form
(``argmax logit_k - log e_k``) on bit-identical logits.  So utterance u must come out class for class the same

  * as one of 8 rows           -- one utterance per ring            (wnv_ring_kernel<.., MODE 0>),
  * as one of 40 rows          -- five per ring, the throughput mode (MODE 1),
  * packed two to a slot       -- continuous batching                (MODE 2: history, first input and noise restart at the seam).

Until round 5 MODE 0 picked in the quotient form and the uniform rounded to 1.0 once in 2^24 draws, where the two forms parted: a one-hot
waveform depended on the batch size about once per 65 k utterance-steps.  Reference semantics: wavenet.py:332-335 (softmax +
OneHotCategorical == argmax p_k / e_k, e ~ Exp(1): SURVEY.md A.3)."""
import numpy as np
import pytest
import torch

from tests._configs import CONFIGS, build, inputs

pytestmark = pytest.mark.gpu


def _ring_rows(eng, B, T, c_up, seed, want_params=False):
    out, params, index = eng.generate(B=B, T=T, c_up=c_up, seed=seed, kernel=2, want_index=True, want_params=want_params)
    assert eng.last_kernel() == 2
    assert torch.equal(out.argmax(1).to(torch.int32), index)
    return index, params


@pytest.mark.parametrize("name,T", [("cfg1_mulaw256", 32768), ("cfg0_mulaw256_small", 32768),
                                    # (round 6, last) a 30-layer model: its tap role runs on the matrix pipe, sixteen utterances (two passes) per
                                    # multiplication -- a lone pass at B = 8, pairs + a lone pass at B = 40, the packed kernel: same bits
                                    ("cfg1b_mulaw256_intree", 16384)])
def test_one_hot_classes_do_not_depend_on_the_batch_size_or_the_packing(name, T):
    """8 utterances x 32 768 steps = 2^18 utterance-steps x 256 draws: 2^26 draws of the stream -- four times the period at which the old
    uniform hit 1.0 -- compared class for class across the three instantiations."""
    kw = CONFIGS[name]
    seed, NU = 20260930, 8
    m = build(name).to("cuda")
    eng = m._get_engine()
    has_c = kw.get("cin_channels", -1) > 0
    c8 = c40 = None
    if has_c:
        c, _ = inputs(name, 40, T)
        c40 = eng.upsample(c.cuda(), T_expected=T)                    # rows are upsampled independently of each other
        c8 = c40[:NU].contiguous()
    a, pa = _ring_rows(eng, NU, T, c8, seed, want_params=True)       # MODE 0: one utterance per ring
    b, pb = _ring_rows(eng, 40, T, c40, seed, want_params=True)      # MODE 1: five per ring
    assert torch.equal(pa, pb[:NU]), "the logits themselves depend on the batch size"
    assert torch.equal(a, b[:NU]), f"first differing utterance-step: {torch.nonzero(a != b[:NU])[0].tolist()}"
    # MODE 2: the same 8 utterances packed two to a slot (slot s runs utterance s, then utterance s + 4)
    slots = NU // 2
    start = torch.zeros(slots, 2 * T, dtype=torch.int32)
    uid = torch.zeros(slots, 2 * T, dtype=torch.int32)
    start[:, T:] = T
    for s in range(slots):
        uid[s, :T], uid[s, T:] = s, s + slots
    cpk = torch.cat([c8[:slots], c8[slots:]], dim=1).contiguous() if has_c else None
    out, _, idx = eng.generate(B=slots, T=2 * T, c_up=cpk, seed=seed, kernel=2, want_index=True,
                               seg_start=start.cuda(), seg_uid=uid.cuda())
    packed = torch.cat([idx[:, :T], idx[:, T:]], dim=0)
    assert torch.equal(packed, a), f"first differing utterance-step: {torch.nonzero(packed != a)[0].tolist()}"
    # the run is a real one: many classes are in use and the utterances differ from each other
    ac = a.cpu().numpy()
    assert len(np.unique(ac)) > 128 and not np.array_equal(ac[0], ac[1])
    print(f"{name}: {NU} x {T} = {NU * T} utterance-steps identical at B = 8 (MODE 0), B = 40 (MODE 1) and packed (MODE 2)")
    m.to("cpu")


def test_scalar_models_do_not_depend_on_the_batch_size_either():
    """egs/mol: the samples are continuous, every instantiation evaluates the same arithmetic in the same order -- bit for bit.  (B = 8
    is run on the kernel auto picks for 8 rows and on the plain ring; the split-ring instantiation, when auto picks it, re-associates two
    sums and is held to 1e-5 instead.)"""
    name, T, seed = "cfg2_mol", 8192, 77
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, 40, T)
    c40 = eng.upsample(c.cuda(), T_expected=T)
    c8 = c40[:8].contiguous()
    a, _, _ = eng.generate(B=8, T=T, c_up=c8, seed=seed, kernel=2)
    b, _, _ = eng.generate(B=40, T=T, c_up=c40, seed=seed, kernel=2)
    assert torch.equal(a, b[:8])
    one, _, _ = eng.generate(B=1, T=T, c_up=c8[:1].contiguous(), seed=seed, kernel=2)
    assert torch.equal(one, a[:1])
    m.to("cpu")
