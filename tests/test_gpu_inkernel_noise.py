"""-m gpu: the mode bench.py times draws its noise IN THE KERNEL (``rng = "philox"``: wnv_noise_gen, csrc/wnv_dev.h) -- no tape the
reference could replay.  What makes those runs checkable: the stream is a documented function of (seed, utterance, step, position)
(Philox4x32-10, restated in numpy and pinned by its published known answers: tests/_philox.py, tests/test_philox_cpu.py), the kernels
report the head outputs every sample was drawn from (``params_out``; compared with the reference itself under teacher forcing and by
autoregressive consistency: tests/test_gpu_vs_reference.py), and the samplers are the reference's functions of (head output, noise)
(mixture.py:118-156, 221-270, restated in oracle/wavenet_oracle.py and pinned by reference-made fixtures).  So: rebuild the tape the
stream stands for on the host, apply the oracle's sampler to the kernel's own head outputs, and every sample of the launch must be that
-- 1e-4, or a near tie of the Gumbel pick (tests/_margins.py).  The scalar-output models here (MoL: the bench workload, at the bench
batch, on the throughput instantiation, on the generic kernel, with a speaker embedding, as packed slots; Gaussian: the in-tree and the
30-layer preset); the one-hot models' picks: tests/test_gpu_packed.py::test_in_kernel_picks_are_the_argmax_of_their_own_scores."""
import numpy as np
import pytest
import torch

from oracle.wavenet_oracle import sample_gaussian, sample_mol
from tests import _philox
from tests._configs import CONFIGS, build, inputs
from tests._margins import assert_match_or_near_tie
from wavenet_vocoder_amd import sharding

pytestmark = pytest.mark.gpu
TOL = 1e-4


def host_samples(kw, params, tape):
    """params (B, O, T), tape (T, B, NZ) -> (B, 1, T): the oracle's sampler at every (utterance, step)"""
    B, O, T = params.shape
    y = params.permute(0, 2, 1).reshape(B * T, O)
    nz = tape.permute(1, 0, 2).reshape(B * T, -1)
    fn = sample_mol if kw.get("output_distribution", "Logistic") == "Logistic" else sample_gaussian
    return fn(y, nz).view(B, 1, T)


def stream_tape(kw, seed, T, B, b0=0):
    return torch.from_numpy(_philox.tape(seed, T, B, scalar_input=True, output_distribution=kw.get("output_distribution", "Logistic"),
                                         out_channels=kw["out_channels"], b0=b0))


@pytest.mark.parametrize("name,B,kernel", [("cfg2_mol", 8, 2), ("cfg2_mol", 8, 1), ("cfg2_mol", 40, 2), ("cfg3_gaussian", 8, 2),
                                           ("cfg3b_gaussian30", 7, 2), ("cfg4_mol_multispeaker", 8, 2), ("wide_mol_512", 8, 3)])
def test_every_sample_is_the_samplers_function_of_head_output_and_stream(name, B, kernel):
    kw = CONFIGS[name]
    T, seed = 1024, 20260923
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, gids = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    out, params, _ = eng.generate(B=B, T=T, c_up=c_up, g_ids=None if gids is None else gids[:, 0].cuda(), seed=seed, want_params=True,
                                  kernel=kernel)
    assert eng.last_kernel() == kernel
    out, params = out.cpu(), params.cpu()
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0 and float(out.std()) > 1e-3
    tape = stream_tape(kw, seed, T, B)
    want = host_samples(kw, params, tape)
    flips = assert_match_or_near_tie(out, want, params, tape, kw, tol=TOL, what=f"{name} B={B} kernel={kernel}")
    same = (out - want).abs() < TOL
    print(f"{name} B={B} kernel {kernel}: {int(same.sum())} of {out.numel()} samples equal the host evaluation "
          f"(largest difference among them {float((out - want).abs()[same].max()):.2e}), {flips} near ties")
    m.to("cpu")


def test_packed_slots_draw_the_stream_of_the_utterance():
    """packed slots: the stream is addressed by (utterance id, step within the utterance), whatever slot and offset it runs at"""
    name, seed = "cfg2_mol", 31
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    g = torch.Generator().manual_seed(5)
    mels = [torch.randn(80, f, generator=g) for f in (3, 1, 2, 4, 1, 2, 3)]
    par = []
    outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=kw["cin_pad"], slots=3, seed=seed, params_out=par)
    for i, (y, p) in enumerate(zip(outs, par)):
        y, p = y.cpu().unsqueeze(0), p.cpu().unsqueeze(0)                     # (1, 1, T_i), (1, O, T_i)
        tape = stream_tape(kw, seed, y.shape[-1], 1, b0=i)
        want = host_samples(kw, p, tape)
        assert_match_or_near_tie(y, want, p, tape, kw, tol=TOL, what=f"packed utterance {i}")
    m.to("cpu")


@pytest.mark.parametrize("name", ["cfg1_mulaw256", "cfg1b_mulaw256_intree"])
def test_one_hot_picks_of_the_throughput_instantiation(name):
    """40 utterances of a mu-law model: more than four per ring (the throughput instantiation) -- under in-kernel noise every class it picks
    must be the argmax of logit_k - log e_k over its own head outputs and the host-restated stream (or within 1e-5 of it)."""
    kw = CONFIGS[name]
    B, T, seed = 40, 512, 20260923
    m = build(name).to("cuda")
    eng = m._get_engine()
    c, _ = inputs(name, B, T)
    out, params, index = eng.generate(B=B, T=T, c_up=eng.upsample(c.cuda(), T_expected=T), seed=seed, want_params=True, want_index=True, kernel=2)
    assert eng.last_kernel() == 2 and torch.equal(out.argmax(1).to(torch.int32), index)
    params, index = params.cpu().numpy(), index.cpu().numpy()
    worst = 0.0
    for b in range(B):
        margins = _philox.categorical_pick_margins(params[b], index[b], seed, b)
        assert np.isfinite(margins).all() and float(margins.max()) < 1e-5, f"utterance {b}, step {int(margins.argmax())}: {float(margins.max()):.3e}"
        worst = max(worst, float(margins.max()))
    assert len(np.unique(index)) > 64
    print(f"{name}: {B * T} picks of the throughput instantiation under in-kernel noise, largest gap to the best score {worst:.2e}")
    m.to("cpu")
