"""-m gpu: the device mel front end (wnv_logmel) against the CPU oracle (oracle/mel_oracle.py).

Tolerance: the oracle is float64; the device computes the FFT, magnitudes and the mel projection in float32 (as
librosa itself does for float32 input: complex64 STFT, float32 filterbank).  1e-4 absolute on log10 values -- the tolerance the
reference's own tests use for float comparisons (tests/test_model.py:361-366) -- holds wherever a band carries signal; bands
that are numerically empty (|S| within float32 round-off of the frame's peak) are compared on the linear scale instead."""
import os

import numpy as np
import pytest
import torch

from oracle import mel_oracle as M
from wavenet_vocoder_amd import audio
from wavenet_vocoder_amd.audio import MelFrontEnd, default_hparams

pytestmark = pytest.mark.gpu


def signal(n, seed, sr=22050.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    return (0.3 * np.sin(2 * np.pi * (200.0 * t + 3000.0 * t * t)) + 0.05 * rng.standard_normal(n)).astype(np.float32)


def close(got, want, atol=1e-4):
    got = np.asarray(got, np.float64)
    lin_ok = np.abs(10.0 ** got - 10.0 ** want) <= 2e-6 * (10.0 ** want).max()
    assert got.shape == want.shape
    bad = ~((np.abs(got - want) <= atol) | lin_ok)
    assert not bad.any(), (np.abs(got - want)[bad].max(), int(bad.sum()))


def test_golden_fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mel_preset.npz"))
    fe = MelFrontEnd(default_hparams())
    S = fe.logmelspectrogram(torch.from_numpy(z["y"]))
    assert S.shape == z["logmel"].shape == (80, 1 + len(z["y"]) // 256)
    close(S.cpu().numpy(), z["logmel"].astype(np.float64))
    fe.set_scaler(mean=z["mean"], scale=z["scale"])
    F = fe.feats(torch.from_numpy(z["y"]))
    np.testing.assert_allclose(F.cpu().numpy(), z["feats"], atol=3e-4)
    np.testing.assert_allclose(fe.feats(z["y"], normalize=False).cpu().numpy(), S.cpu().numpy().T, atol=0)


@pytest.mark.parametrize("over,pad", [
    ({}, "reflect"), ({}, "constant"),
    ({"win_length": 800}, "reflect"),
    ({"fft_size": 2048, "win_length": 1200, "hop_size": 300, "num_mels": 128, "fmin": 0, "fmax": None}, "reflect"),   # odd log2: radix-2 tail
    ({"fft_size": 512, "win_length": 512, "hop_size": 128, "num_mels": 40, "fmin": 60, "fmax": 7600, "sample_rate": 16000}, "reflect"),
    ({"fft_size": 256, "win_length": 256, "hop_size": 64, "num_mels": 20, "fmin": 0, "fmax": 4000, "sample_rate": 8000}, "constant"),
    ({"fft_size": 4096, "win_length": 4096, "hop_size": 1024, "num_mels": 80}, "reflect"),
])
@pytest.mark.parametrize("n", [2500, 7777])
def test_against_oracle(over, pad, n):
    hp = default_hparams(**over)
    y = signal(n, n, hp.sample_rate)
    fe = MelFrontEnd(hp, pad_mode=pad)
    got = fe.logmelspectrogram(y).cpu().numpy()
    want = M.logmelspectrogram(y, hp, pad_mode=pad)
    assert got.shape == (hp.num_mels, fe.frames(n))
    close(got, want)
    np.testing.assert_allclose(fe.mel_basis(), M.mel_filterbank(hp.sample_rate, hp.fft_size, hp.num_mels, hp.fmin, hp.fmax), rtol=1e-6, atol=1e-9)


def test_batch_layouts_and_module_function():
    hp = default_hparams()
    ys = np.stack([signal(6000, s) for s in range(5)])
    fe = MelFrontEnd(hp)
    S = fe.logmelspectrogram(ys)                                   # (B, 80, frames)
    F = fe.feats(ys)                                               # (B, frames, 80)
    assert S.shape == (5, 80, 24) and F.shape == (5, 24, 80)
    assert torch.equal(S.transpose(1, 2), F)
    for b in range(5):
        assert torch.equal(S[b], fe.logmelspectrogram(ys[b]))      # batch members are independent
        close(S[b].cpu().numpy(), M.logmelspectrogram(ys[b], hp))
    out = audio.logmelspectrogram(ys[0])                           # numpy in -> numpy out, like audio.logmelspectrogram
    assert isinstance(out, np.ndarray) and out.dtype == np.float32 and np.array_equal(out, S[0].cpu().numpy())


def test_silence_hits_the_floor_and_edge_lengths():
    hp = default_hparams()
    fe = MelFrontEnd(hp)
    assert (fe.logmelspectrogram(torch.zeros(3000)) + 10.0).abs().max().item() < 2e-6      # log10 of the 1e-10 floor
    for n in (513, 767, 768, 1024):                                 # fewer than one full frame; odd / even frame counts
        y = signal(n, n)
        close(fe.logmelspectrogram(y).cpu().numpy(), M.logmelspectrogram(y, hp))
    with pytest.raises(AssertionError):                             # reflect padding needs > fft_size / 2 samples (numpy raises too)
        fe.logmelspectrogram(signal(512, 0))
    fc = MelFrontEnd(hp, pad_mode="constant")
    close(fc.logmelspectrogram(signal(100, 1)).cpu().numpy(), M.logmelspectrogram(signal(100, 1), hp, "constant"))
    with pytest.raises(RuntimeError):
        fe.feats(signal(3000, 0), normalize=True)                   # no scaler installed


def test_properties_at_full_size():
    """10 s x 8 utterances (the bench batch's audio): homogeneity, shift by one hop, determinism."""
    hp = default_hparams()
    fe = MelFrontEnd(hp)
    ys = torch.from_numpy(np.stack([signal(220500, 100 + s) for s in range(8)])).cuda()
    S = fe.logmelspectrogram(ys)
    assert S.shape == (8, 80, 862) and torch.isfinite(S).all()
    assert torch.equal(S, fe.logmelspectrogram(ys))
    S2 = fe.logmelspectrogram(ys * 2)
    assert (S2 - S - float(np.log10(2.0))).abs().max().item() < 2e-5
    Sh = fe.logmelspectrogram(ys[:, 256:])                           # interior frames move by exactly one column
    assert (Sh[:, :, 3:-3] - S[:, :, 4:-3]).abs().max().item() < 1e-5
    close(S[3].cpu().numpy(), M.logmelspectrogram(ys[3].cpu().numpy(), hp))


def test_scaler_object():
    from sklearn.preprocessing import StandardScaler
    hp = default_hparams()
    y = signal(20000, 5)
    fe = MelFrontEnd(hp)
    raw = fe.feats(y, normalize=False).cpu().numpy()
    sc = StandardScaler().fit(raw.astype(np.float64))
    fe.set_scaler(sc)
    np.testing.assert_allclose(fe.feats(y).cpu().numpy(), sc.transform(raw.astype(np.float64)), atol=2e-5)
