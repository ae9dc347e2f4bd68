"""Runs of the REAL reference (oracle/_ref, byte-compiled from /root/reference by oracle/build_ref.py) on seeded inputs, shared by
tests/test_gpu_vs_reference.py and tests/test_reference_build_cpu.py.

A case = teacher-forced for Tt steps, then free-running up to T under torch.manual_seed(seed) -- the reference's own mixed mode
(wavenet.py:296-303).  Returned: the inputs, the reference's output, the head outputs it handed to its sampler at every step, and the
noise tape wavenet_vocoder_amd.noise replays from the same seed (bit-identical to the draws the reference made: checked by
``tape_replay_is_exact``)."""
import functools

import torch

from oracle import reference as R
from tests._configs import CONFIGS, build, inputs
from wavenet_vocoder_amd.noise import make_noise_tape


def teacher(kw, B, T, seed=3):
    g = torch.Generator().manual_seed(seed)
    if kw.get("scalar_input", False):
        return torch.tanh(torch.randn(B, 1, T, generator=g) * 0.5)
    idx = torch.randint(0, kw["out_channels"], (B, T), generator=g)
    return torch.zeros(B, kw["out_channels"], T).scatter_(1, idx.unsqueeze(1), 1.0)


def first_input(kw, B):
    """The default first input (wavenet.py:281-289) as a one-step teacher input (B, C, 1): gives the reference its batch size
    (it learns B from test_inputs / c only, wavenet.py:253,273 -- a speaker-conditioned or unconditioned free run would be B = 1)."""
    if kw.get("scalar_input", False):
        return torch.zeros(B, 1, 1)
    x = torch.zeros(B, kw["out_channels"], 1)
    x[:, 127] = 1.0
    return x


def tape_for(kw, T, B, seed):
    torch.manual_seed(seed)
    return make_noise_tape(T, B, scalar_input=kw.get("scalar_input", False),
                           output_distribution=kw.get("output_distribution", "Logistic"), out_channels=kw["out_channels"])


@functools.lru_cache(maxsize=4)
def reference_model(name):
    ours = build(name)
    return ours, R.build_model(CONFIGS[name], ours.state_dict(), fast=True)


def reference_case(name, B, Tt, T, seed=11, threads=8):
    """dict(kw, model (this package's, CPU), c, gids, x (B, C, Tt) teacher input, tape (T, B, NZ), want (B, C, T), wparams (B, O, T))."""
    kw = CONFIGS[name]
    ours, rm = reference_model(name)
    c, gids = inputs(name, B, T)
    x = teacher(kw, B, Tt) if Tt > 1 else first_input(kw, B)
    torch.set_num_threads(threads)
    want, wparams = R.incremental(rm, seed=seed, T=T, c=c, g=gids, test_inputs=x, softmax=True, quantize=True)
    tape = tape_for(kw, T, B, seed)
    return dict(kw=kw, model=ours, c=c, gids=gids, x=x, tape=tape, want=want, wparams=wparams, B=B, Tt=x.shape[-1], T=T)


def tape_replay_is_exact():
    """The tape wavenet_vocoder_amd.noise draws from a seed, fed through the ORACLE's samplers, reproduces what the REFERENCE's own
    samplers draw from torch's generator under the same seed -- bit for bit, on this machine (the GPU box's CPU is not the authoring
    container's: vectorised paths of torch's generators depend on the batch size, see wavenet_vocoder_amd/noise.py)."""
    import importlib
    from oracle.wavenet_oracle import sample_categorical, sample_gaussian, sample_mol
    R.load_reference()
    mix = importlib.import_module("wavenet_vocoder.mixture")
    for dist, C, fn in (("Logistic", 30, mix.sample_from_discretized_mix_logistic), ("Normal", 2, mix.sample_from_mix_gaussian),
                        ("Normal", 3, mix.sample_from_mix_gaussian), ("Normal", 9, mix.sample_from_mix_gaussian)):
        for B in (1, 2, 8, 16, 48):
            T = 10
            ys = torch.randn(T, B, C, 1, generator=torch.Generator().manual_seed(B))
            torch.manual_seed(11)
            want = torch.stack([fn(ys[t]) for t in range(T)])
            torch.manual_seed(11)
            tape = make_noise_tape(T, B, scalar_input=True, output_distribution=dist, out_channels=C)
            ofn = sample_mol if dist == "Logistic" else sample_gaussian
            got = torch.stack([ofn(ys[t, :, :, 0], tape[t]).view(B, 1) for t in range(T)])
            assert torch.equal(want, got), (dist, C, B, float((want - got).abs().max()))
    for B in (1, 2, 8, 40):
        p = torch.softmax(torch.randn(6, B, 256, generator=torch.Generator().manual_seed(B)), -1)
        torch.manual_seed(12)
        want = torch.stack([torch.distributions.OneHotCategorical(p[t]).sample().argmax(-1) for t in range(6)])
        torch.manual_seed(12)
        tape = make_noise_tape(6, B, scalar_input=False, output_distribution="Logistic", out_channels=256)
        got = torch.stack([sample_categorical(p[t], tape[t]) for t in range(6)])
        assert torch.equal(want, got), B
    return True
