"""-m gpu: the group-ring kernel for WIDE models (csrc/wnv_wide.hip, kernel = 3): eight workgroups per layer, weights resident,
one all-gather hop per layer (the pair (u, h) travels together, gate-to-gate folding on the host) -- against the CPU oracle, against the generic kernel on the same inputs, and through
size-independent properties at the published 24-layer 512 / 512 / 256 geometry."""
import time

import numpy as np

import pytest
import torch

import wavenet_vocoder_amd as wnv
from oracle.wavenet_oracle import Oracle
from tests._configs import CONFIGS, build, inputs, tame_head_
from tests._golden import oracle_config
from tests._margins import assert_free_run_agrees_until_near_tie, assert_match_or_near_tie
from wavenet_vocoder_amd.noise import make_noise_tape

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = {
    # name: (kwargs, B, Tt, T)
    "mol_512_512_256_mel": (dict(out_channels=30, layers=6, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=256,
                                 kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80), 2, 96, 160),
    "gauss_256_384_192_k2_global": (dict(out_channels=2, layers=4, stacks=2, residual_channels=256, gate_channels=384, skip_out_channels=192,
                                         kernel_size=2, dropout=0.0, scalar_input=True, output_distribution="Normal", gin_channels=8,
                                         n_speakers=3, use_speaker_embedding=True), 3, 64, 128),
    "mol_320_512_256_nine_layers": (dict(out_channels=30, layers=9, stacks=3, residual_channels=320, gate_channels=512, skip_out_channels=256,
                                         kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=20), 8, 48, 96),
    # the reference constructor's DEFAULT geometry (wavenet.py:98-101: 512 / 512 / 512): two skip banks per slice, the head is four workgroups
    "mol_512_512_512_default_constructor": (dict(out_channels=30, layers=6, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=512,
                                                 kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80), 2, 96, 160),
    "gauss_384_512_400_five_utterances": (dict(out_channels=2, layers=5, stacks=1, residual_channels=384, gate_channels=512, skip_out_channels=400,
                                               kernel_size=2, dropout=0.0, scalar_input=True, output_distribution="Normal", cin_channels=7), 5, 40, 88),
    # more than 8 utterances: the tap stream takes them in two passes, the deferred history copies run two utterances behind
    "mol_512_384_256_thirteen_utterances": (dict(out_channels=30, layers=4, stacks=2, residual_channels=512, gate_channels=384, skip_out_channels=256,
                                                 kernel_size=3, dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=33), 13, 40, 72),
}


def tape_for(kw, T, B, seed):
    return make_noise_tape(T, B, scalar_input=True, output_distribution=kw.get("output_distribution", "Logistic"),
                           out_channels=kw["out_channels"], generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("name", list(CASES))
def test_wide_vs_oracle_and_generic(name):
    kw, B, Tt, T = CASES[name]
    torch.manual_seed(23)
    m = tame_head_(wnv.WaveNet(**kw).eval())
    o = Oracle(oracle_config(kw), m.state_dict())
    g = torch.Generator().manual_seed(2)
    cin, gin = kw.get("cin_channels", -1), kw.get("gin_channels", -1)
    c = torch.randn(B, cin, T, generator=g) if cin > 0 else None
    gids = torch.randint(0, kw["n_speakers"], (B, 1), generator=g) if gin > 0 else None
    x = torch.tanh(torch.randn(B, 1, Tt, generator=g) * 0.5)
    tape = tape_for(kw, T, B, 6)
    torch.set_num_threads(8)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, g=gids, T=T, noise=tape, return_params=True)
    eng = m.to("cuda")._get_engine()
    args = dict(B=B, T=T, c_up=None if c is None else c.transpose(1, 2).contiguous().cuda(), g_ids=None if gids is None else gids[:, 0].cuda(),
                teacher=x.transpose(1, 2).contiguous().cuda(), noise=tape.cuda(), want_params=True)
    out, params, _ = eng.generate(kernel=3, **args)
    assert eng.last_kernel() == 3
    gen_out, gen_params, _ = eng.generate(kernel=1, **args)
    out, params = out.cpu(), params.cpu()
    err = float((params[:, :, :Tt] - wparams[:, :, :Tt]).abs().max())
    print(f"{name}: forced head outputs vs oracle {err:.2e}, vs generic kernel {float((params[:, :, :Tt] - gen_params.cpu()[:, :, :Tt]).abs().max()):.2e}")
    assert err < TOL
    assert float((params[:, :, :Tt] - gen_params.cpu()[:, :, :Tt]).abs().max()) < 5e-5
    assert_match_or_near_tie(out[:, :, :Tt - 1], want[:, :, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw, tol=TOL)
    assert_free_run_agrees_until_near_tie(out, want, params, wparams, tape, kw, t0=Tt - 1, what=name)
    again, _, _ = eng.generate(kernel=3, **args)
    assert torch.equal(again.cpu(), out), "determinism"
    auto, _, _ = eng.generate(kernel=0, **args)
    assert eng.last_kernel() == 3 and torch.equal(auto.cpu(), out), "auto picks the group ring for a wide model"


ONEHOT_WIDE = dict(out_channels=256, layers=6, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=256, kernel_size=3,
                   dropout=0.0, cin_channels=80)        # the published CMU-ARCTIC geometry (mu-law 256 in / out), six layers of it


@pytest.mark.parametrize("K", [256, 512])
def test_wide_onehot_vs_oracle_and_generic(K):
    """(K = 512: the reference constructor's bare defaults -- wavenet.py:98-101: mu-law 256 in / out, 512 / 512 / 512 -- six layers of it
    with an 80-mel conditioning added: four hidden-layer workgroups + two output-layer workgroups.)
    One-hot wide models: the head is two workgroups (hidden layer | output layer + softmax + OneHotCategorical + first_conv row
    gather).  Teacher-forced probabilities against the oracle, sampled classes of a free run equal the oracle's until a near tie."""
    kw = dict(ONEHOT_WIDE, skip_out_channels=K)
    torch.manual_seed(29)
    m = tame_head_(wnv.WaveNet(**kw).eval())
    o = Oracle(oracle_config(kw), m.state_dict())
    B, Tt, T = 2, 64, 128
    g = torch.Generator().manual_seed(4)
    c = torch.randn(B, 80, T, generator=g)
    idx = torch.randint(0, 256, (B, Tt), generator=g)
    x = torch.zeros(B, 256, Tt).scatter_(1, idx.unsqueeze(1), 1.0)
    tape = make_noise_tape(T, B, scalar_input=False, output_distribution="Logistic", out_channels=256, generator=torch.Generator().manual_seed(9))
    torch.set_num_threads(8)
    want_p, wparams_tf = o.incremental_forward(test_inputs=x, c=c[:, :, :Tt], T=Tt, softmax=True, quantize=False, noise=tape[:Tt], return_params=True)
    want, wparams = o.incremental_forward(test_inputs=x, c=c, T=T, noise=tape, return_params=True)          # forced, then free-running classes
    eng = m.to("cuda")._get_engine()
    cu = c.transpose(1, 2).contiguous().cuda()
    tin = x.transpose(1, 2).contiguous().cuda()
    probs, params_tf, _ = eng.generate(B=B, T=Tt, c_up=cu[:, :Tt].contiguous(), teacher=tin, noise=tape[:Tt].cuda(), softmax=True, quantize=False,
                                       want_params=True, kernel=3)
    assert eng.last_kernel() == 3
    assert float((params_tf.cpu() - wparams_tf).abs().max()) < TOL and float((probs.cpu() - want_p).abs().max()) < TOL
    out, params, cls = eng.generate(B=B, T=T, c_up=cu, teacher=tin, noise=tape.cuda(), want_params=True, want_index=True, kernel=3)
    assert torch.equal(out.sum(1), torch.ones(B, T, device="cuda")) and torch.equal(out.argmax(1).int(), cls.int())
    assert_match_or_near_tie(cls.cpu()[:, :Tt - 1], want.argmax(1)[:, :Tt - 1], wparams[:, :, :Tt - 1], tape[:Tt - 1], kw)
    hz = assert_free_run_agrees_until_near_tie(cls.cpu(), want.argmax(1), params.cpu(), wparams, tape, kw, t0=Tt - 1, what="wide one-hot")
    print(f"wide one-hot: forced head outputs vs oracle {float((params_tf.cpu() - wparams_tf).abs().max()):.2e}; free-run horizon {hz} of {T}")
    gen, gparams, gcls = eng.generate(B=B, T=T, c_up=cu, teacher=tin, noise=tape.cuda(), want_params=True, want_index=True, kernel=1)
    assert float((params[:, :, :Tt] - gparams[:, :, :Tt]).abs().max()) < 5e-5
    free, _, fcls = eng.generate(B=B, T=64, c_up=cu[:, :64].contiguous(), noise=tape[:64].cuda(), want_index=True, kernel=3)       # implicit class-127 start
    free1, _, fcls1 = eng.generate(B=B, T=64, c_up=cu[:, :64].contiguous(), noise=tape[:64].cuda(), want_index=True, kernel=1)
    assert (fcls == fcls1).float().mean().item() > 0.9


def test_wide_published_geometry_properties_and_speed():
    """24 layers, 512 / 512 / 256, 80-mel conditioned MoL (the published CMU-ARCTIC geometry): properties at a length the oracle cannot
    reach, and the reason this kernel exists -- one utterance faster than real time."""
    name = "wide_mol_512"
    kw = CONFIGS[name]
    m = build(name).to("cuda")
    eng = m._get_engine()
    B, T = 2, 2048
    c, _ = inputs(name, B, T)
    c_up = eng.upsample(c.cuda(), T_expected=T)
    tape = tape_for(kw, T, B, 9).cuda()
    full, _, _ = eng.generate(B=B, T=T, c_up=c_up, noise=tape, kernel=3)
    T2 = 512
    pre, _, _ = eng.generate(B=B, T=T2, c_up=c_up[:, :T2].contiguous(), noise=tape[:T2].contiguous(), kernel=3)
    assert torch.equal(pre, full[:, :, :T2]), "prefix property"
    solo, _, _ = eng.generate(B=1, T=T2, c_up=c_up[1:2, :T2].contiguous(), noise=tape[:T2, 1:2].contiguous(), kernel=3)
    assert torch.equal(solo[0], full[1, :, :T2]), "batch members must be independent"
    B16 = 16
    c16, _ = inputs(name, B16, T2)
    cu16 = eng.upsample(c16.cuda(), T_expected=T2)
    tape16 = tape_for(kw, T2, B16, 10).cuda()
    many, _, _ = eng.generate(B=B16, T=T2, c_up=cu16, noise=tape16, kernel=3)
    lone, _, _ = eng.generate(B=1, T=T2, c_up=cu16[11:12].contiguous(), noise=tape16[:, 11:12].contiguous(), kernel=3)
    assert torch.equal(lone[0], many[11]), "utterance 11 of 16 must not depend on its neighbours"
    assert torch.isfinite(full).all() and float(full.abs().max()) <= 1.0 and float(full.std()) > 1e-3
    gen, _, _ = eng.generate(B=B, T=128, c_up=c_up[:, :128].contiguous(), noise=tape[:128].contiguous(), kernel=1)
    assert float((gen - full[:, :, :128]).abs().max()) < 1e-3                       # the generic kernel's trajectory over a short horizon
    for Bs in (1, 8, 16):
        cs, _ = inputs(name, Bs, 4096)
        cu = eng.upsample(cs.cuda(), T_expected=4096)
        eng.generate(B=Bs, T=4096, c_up=cu, seed=1, kernel=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(B=Bs, T=4096, c_up=cu, seed=2, kernel=3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_utt = 4096 / dt
        print(f"wide_mol_512, B = {Bs}: {Bs * per_utt / 1e3:.1f} kSamples/s, {per_utt / 1e3:.1f} kSamples/s per utterance = "
              f"{per_utt / 16000:.2f}x real time at 16 kHz, {per_utt / 22050:.2f}x at 22.05 kHz, {per_utt / 24000:.2f}x at 24 kHz")
        if Bs == 1:
            assert per_utt > 16000, "a single utterance of the published (16 kHz) geometry must run faster than real time"
    # the published models themselves are mu-law 256 in / out: the same stack with the one-hot head
    kw1 = dict(ONEHOT_WIDE, layers=24, stacks=4)
    torch.manual_seed(1)
    eng1 = tame_head_(wnv.WaveNet(**kw1).eval()).to("cuda")._get_engine()
    cu = torch.randn(1, 4096, 80, device="cuda")
    eng1.generate(B=1, T=4096, c_up=cu, seed=1, kernel=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out1, _, _ = eng1.generate(B=1, T=4096, c_up=cu, seed=2, kernel=3)
    torch.cuda.synchronize()
    per_utt = 4096 / (time.perf_counter() - t0)
    print(f"24-layer 512/512/256 mu-law-256 model, B = 1: {per_utt / 1e3:.1f} kSamples/s = {per_utt / 16000:.2f}x real time at 16 kHz")
    assert torch.equal(out1.sum(1), torch.ones(1, 4096, device="cuda")) and per_utt > 16000


def test_wide_onehot_model_through_batch_wavegen():
    """The caller's view: a published-geometry (512 / 512 / 256, mu-law 256, 80-mel + ConvInUpsampleNetwork) model through
    synthesis.batch_wavegen -- upsampling, the group-ring sample loop (auto), the device post-chain -- equals the post-chain oracle
    applied to the engine's own one-hot output under the same seed."""
    from oracle import postchain_oracle as P
    from wavenet_vocoder_amd import synthesis
    kw = dict(ONEHOT_WIDE, cin_pad=2, upsample_conditional_features=True,
              upsample_params=dict(upsample_scales=[4, 4, 4, 4], cin_channels=80, cin_pad=2))
    torch.manual_seed(31)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    B, frames = 2, 3
    c = torch.randn(B, 80, frames + 4, generator=torch.Generator().manual_seed(1))
    h = synthesis.default_hparams(input_type="mulaw-quantize", quantize_channels=256, out_channels=256, cin_pad=2, hop_size=256,
                                  upsample_conditional_features=True)
    torch.manual_seed(5)
    wav = synthesis.batch_wavegen(m, c=c, g=None, hparams=h)
    assert m._get_engine().last_kernel() == 3
    assert wav.shape == (B, frames * 256) and wav.dtype == np.float32 and np.isfinite(wav).all()
    torch.manual_seed(5)
    with torch.no_grad():
        y_hat = m.incremental_forward(c=c.cuda(), T=frames * 256, softmax=True, quantize=True)
    assert torch.equal(y_hat.sum(1), torch.ones(B, frames * 256, device="cuda"))
    want = P.post_chain(y_hat.cpu().numpy(), "mulaw-quantize", quantize_channels=256, postprocess=h.postprocess,
                        coef=h.preemphasis_coef, global_gain_scale=h.global_gain_scale)
    assert np.abs(wav - want).max() < 1e-4 * max(1.0, np.abs(want).max())


def test_wide_onehot_eleven_utterances_vs_the_generic_kernel():
    """More than 8 utterances on the one-hot head pair (the tap stream takes two passes, the history copies run two utterances behind):
    teacher-forced head outputs against the generic kernel on the same inputs, sampled classes equal wherever the choice is not a near tie."""
    kw = dict(ONEHOT_WIDE, layers=4, stacks=2)
    torch.manual_seed(41)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    B, T = 11, 96
    g = torch.Generator().manual_seed(6)
    cu = torch.randn(B, T, 80, generator=g).cuda()
    idx = torch.randint(0, 256, (B, T), generator=g)
    tin = torch.zeros(B, T, 256).scatter_(2, idx.unsqueeze(2), 1.0).cuda()
    tape = make_noise_tape(T, B, scalar_input=False, output_distribution="Logistic", out_channels=256, generator=torch.Generator().manual_seed(2)).cuda()
    out3, p3, c3 = eng.generate(B=B, T=T, c_up=cu, teacher=tin, noise=tape, want_params=True, want_index=True, kernel=3)
    assert eng.last_kernel() == 3
    out1, p1, c1 = eng.generate(B=B, T=T, c_up=cu, teacher=tin, noise=tape, want_params=True, want_index=True, kernel=1)
    assert float((p3 - p1).abs().max()) < 5e-5
    assert_match_or_near_tie(c3.cpu(), c1.cpu(), p1.cpu(), tape.cpu(), kw)
    lone, _, cl = eng.generate(B=1, T=T, c_up=cu[9:10].contiguous(), teacher=tin[9:10].contiguous(), noise=tape[:, 9:10].contiguous(), want_index=True, kernel=3)
    assert torch.equal(cl[0], c3[9]), "utterance 9 of 11 must not depend on its neighbours"


def test_wide_more_than_sixteen_utterances_run_in_slices():
    """A batch beyond what the groups pipeline at once is served in slices of 16 (one launch each), noise addressed by the utterance's
    index in the whole call: same numbers as the generic kernel on the whole batch (tape mode), and -- Philox mode -- an utterance
    alone with its own stream position equals its row in the batch."""
    kw, _, _, _ = CASES["mol_512_384_256_thirteen_utterances"]
    torch.manual_seed(23)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    B, Tt, T = 37, 24, 48
    g = torch.Generator().manual_seed(12)
    cu = torch.randn(B, T, kw["cin_channels"], generator=g).cuda()
    x = torch.tanh(torch.randn(B, Tt, 1, generator=g) * 0.5).cuda()
    tape = tape_for(kw, T, B, 3).cuda()
    out3, p3, _ = eng.generate(B=B, T=T, c_up=cu, teacher=x, noise=tape, want_params=True, kernel=3)
    assert eng.last_kernel() == 3
    out1, p1, _ = eng.generate(B=B, T=T, c_up=cu, teacher=x, noise=tape, want_params=True, kernel=1)
    assert float((p3[:, :, :Tt] - p1[:, :, :Tt]).abs().max()) < 5e-5
    assert_match_or_near_tie(out3.cpu()[:, :, :Tt - 1], out1.cpu()[:, :, :Tt - 1], p1.cpu()[:, :, :Tt - 1], tape.cpu()[:Tt - 1], kw, tol=TOL)
    auto, _, _ = eng.generate(B=B, T=T, c_up=cu, teacher=x, noise=tape, kernel=0)
    assert eng.last_kernel() == 3 and torch.equal(auto, out3), "auto serves a wide model with the group ring at any batch size"
    # Philox: the slices continue one stream (utterance b draws from position b), so slicing does not change a sample
    ph, _, _ = eng.generate(B=B, T=T, c_up=cu, seed=77, kernel=3)
    ph16, _, _ = eng.generate(B=16, T=T, c_up=cu[:16].contiguous(), seed=77, kernel=3)
    assert torch.equal(ph[:16], ph16)
    lone, _, _ = eng.generate(B=1, T=T, c_up=cu[:1].contiguous(), seed=77, kernel=3)                 # utterance 0 draws from stream position 0
    assert torch.equal(lone[0], ph[0])
