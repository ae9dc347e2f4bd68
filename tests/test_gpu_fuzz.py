"""-m gpu: seeded random configurations through every kernel that covers them -- the persistent kernels (ring: kernel 2, group ring:
kernel 3) against the generic kernel (kernel 1, the one that is checked against every reference-made fixture) on the same inputs and
noise tape: teacher-forced head outputs to 5e-5, then free running until a near tie.  The draw covers what the hand-picked cases do
not: conditioning widths that are not multiples of 4, kernel sizes 2-4, every padding amount, uneven stacks, odd batch sizes."""
import random

import pytest
import torch

import wavenet_vocoder_amd as wnv
from tests._configs import tame_head_
from tests._margins import assert_free_run_agrees_until_near_tie
from wavenet_vocoder_amd.noise import make_noise_tape

pytestmark = pytest.mark.gpu


def draw(seed, wide):
    r = random.Random(seed)
    stacks = r.choice([1, 2, 3])
    layers = stacks * r.choice([1, 2, 3, 4] if not wide else [1, 2, 3])
    scalar = r.random() < (0.7 if not wide else 0.6)
    if scalar:
        dist = r.choice(["Logistic", "Normal"])
        O = r.choice([3, 15, 30]) if dist == "Logistic" else r.choice([2, 3, 9])
    else:
        dist, O = "Logistic", r.choice([16, 100, 256])
    if wide:
        R, G, K = r.choice([136, 256, 384, 512]), 2 * r.choice([80, 128, 200, 256]), r.choice([64, 160, 256])
    else:
        R, G = r.choice([16, 64, 100, 128]), 2 * r.choice([8, 48, 100, 128])
        K = r.choice([32, 128, 200, 256]) if not scalar else r.choice([32, 128, 200, 256, 384, 512])
    kw = dict(out_channels=O, layers=layers, stacks=stacks, residual_channels=R, gate_channels=G, skip_out_channels=K,
              kernel_size=r.choice([2, 3, 3, 4]), dropout=0.0, scalar_input=scalar, output_distribution=dist)
    if r.random() < 0.75:
        kw["cin_channels"] = r.choice([1, 7, 20, 33, 80])
    if r.random() < 0.4:
        kw.update(gin_channels=r.choice([3, 16]), n_speakers=4, use_speaker_embedding=True)
    return kw, r.choice([1, 2, 3, 5, 8]), r.choice([40, 72]), r.choice([96, 130])


CASES = [(s, False) for s in range(100, 160)] + [(s, True) for s in range(200, 230)]


@pytest.mark.parametrize("seed,wide", CASES)
def test_persistent_kernels_equal_the_generic_kernel(seed, wide):
    kw, B, Tt, T = draw(seed, wide)
    torch.manual_seed(seed)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    g = torch.Generator().manual_seed(seed)
    cin, gin = kw.get("cin_channels", -1), kw.get("gin_channels", -1)
    c_up = torch.randn(B, T, cin, generator=g).cuda() if cin > 0 else None
    gids = torch.randint(0, 4, (B,), generator=g).cuda() if gin > 0 else None
    scalar = kw["scalar_input"]
    if scalar:
        x = torch.tanh(torch.randn(B, Tt, 1, generator=g) * 0.5).cuda()
    else:
        idx = torch.randint(0, kw["out_channels"], (B, Tt), generator=g)
        x = torch.zeros(B, Tt, kw["out_channels"]).scatter_(2, idx.unsqueeze(2), 1.0).cuda()
    tape = make_noise_tape(T, B, scalar_input=scalar, output_distribution=kw["output_distribution"], out_channels=kw["out_channels"],
                           generator=torch.Generator().manual_seed(seed + 1))
    args = dict(B=B, T=T, c_up=c_up, g_ids=gids, teacher=x, noise=tape.cuda(), want_params=True, want_index=not scalar)
    ref_out, ref_p, ref_i = eng.generate(kernel=1, **args)
    kernel = 3 if wide else 2
    try:
        out, p, i = eng.generate(kernel=kernel, **args)
    except NotImplementedError as e:                      # a draw outside the kernel's coverage (e.g. one-hot with 384 skip channels)
        pytest.skip(f"kernel {kernel} does not cover {kw}: {e}")
    assert eng.last_kernel() == kernel
    err = float((p[:, :, :Tt] - ref_p[:, :, :Tt]).abs().max())
    assert err < 5e-5, (kw, B, err)
    if scalar:
        assert_free_run_agrees_until_near_tie(out.cpu(), ref_out.cpu(), p.cpu(), ref_p.cpu(), tape, kw, t0=0, tol=1e-3, what=str(kw))
    else:
        assert_free_run_agrees_until_near_tie(i.cpu(), ref_i.cpu(), p.cpu(), ref_p.cpu(), tape, kw, t0=0, what=str(kw))
    auto, _, _ = eng.generate(kernel=0, **args)
    assert eng.last_kernel() == kernel and torch.equal(auto, out)


@pytest.mark.parametrize("seed", range(300, 312))
def test_deep_dilations_and_shared_rings_vs_generic(seed):
    """One stack of 6-9 layers (dilation up to 256) forced for long enough that the largest history ring wraps, with batch sizes that
    make several utterances share a ring (and the second tap workgroup join)."""
    r = random.Random(seed)
    layers, kwid = r.choice([6, 7, 8, 9]), r.choice([2, 3])
    wide = seed % 4 == 3
    kw = dict(out_channels=30, layers=layers, stacks=1, residual_channels=512 if wide else r.choice([96, 128]),
              gate_channels=512 if wide else 256, skip_out_channels=r.choice([128, 256]), kernel_size=kwid, dropout=0.0, scalar_input=True,
              output_distribution="Logistic", cin_channels=r.choice([5, 80]))
    B = r.choice([1, 8]) if wide else r.choice([9, 16, 24, 33])
    T = (kwid - 1) * 2 ** (layers - 1) * 2 + 40
    torch.manual_seed(seed)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    eng = m._get_engine()
    g = torch.Generator().manual_seed(seed)
    c_up = torch.randn(B, T, kw["cin_channels"], generator=g).cuda()
    x = torch.tanh(torch.randn(B, T, 1, generator=g) * 0.5).cuda()
    tape = make_noise_tape(T, B, scalar_input=True, output_distribution="Logistic", out_channels=30, generator=torch.Generator().manual_seed(seed)).cuda()
    _, ref_p, _ = eng.generate(B=B, T=T, c_up=c_up, teacher=x, noise=tape, want_params=True, kernel=1)
    _, p, _ = eng.generate(B=B, T=T, c_up=c_up, teacher=x, noise=tape, want_params=True, kernel=3 if wide else 2)
    err = (p - ref_p).abs()
    assert float(err.max()) < 5e-5, (kw, B, float(err.max()), int(err.flatten().argmax()))
