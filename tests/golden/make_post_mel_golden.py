#!/usr/bin/env python3
"""Pin the f1 / f4 oracles to the REAL third-party arithmetic the moment it is available.

The post-chain (synthesis.py:66-84) and the mel front end (audio.py:101-109) get their arithmetic from the reference's
un-vendored, unpinned dependencies nnmnkwii (setup.py:23) and librosa (setup.py:25); neither is installed in the authoring
container and there is no network, so oracle/postchain_oracle.py and oracle/mel_oracle.py restate their published definitions
("parity unpinned").  This script closes the gap wherever those packages CAN be imported: it feeds seeded inputs through the real
functions at the reference's call sites and writes

    tests/golden/post_nnmnkwii.npz     nnmnkwii.preprocessing.{mulaw, inv_mulaw, mulaw_quantize, inv_mulaw_quantize,
                                       preemphasis, inv_preemphasis}  (+ the package version)
    tests/golden/mel_librosa.npz       librosa.stft / librosa.filters.mel / the reference's logmelspectrogram recipe
                                       (+ the package version)

tests/test_third_party_pins_cpu.py compares the oracles with these files when they exist and is skipped -- loudly -- when they
do not.  Committing the two files flips "parity unpinned" to "pinned" for rows f1 / f4 without touching any other code.

    python tests/golden/make_post_mel_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def post_inputs():
    rng = np.random.default_rng(11)
    x = np.concatenate([np.linspace(-1, 1, 2001), rng.uniform(-1, 1, 3000)]).astype(np.float64)
    codes = np.arange(256)
    sig = (0.3 * rng.standard_normal(4000)).astype(np.float64)
    return x, codes, sig


def mel_signal(n=22050, seed=7):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    return (0.3 * np.sin(2 * np.pi * (200.0 * t + 3000.0 * t * t)) + 0.05 * rng.standard_normal(n)).astype(np.float32)


def make_post():
    try:
        import nnmnkwii
        from nnmnkwii import preprocessing as P
    except Exception as e:                                   # noqa: BLE001
        print(f"nnmnkwii is not importable ({type(e).__name__}: {e}) -- post_nnmnkwii.npz NOT written; f1 stays parity-unpinned")
        return False
    x, codes, sig = post_inputs()
    out = dict(version=str(getattr(nnmnkwii, "__version__", "?")), x=x, codes=codes, sig=sig)
    for mu in (255, 65535):
        out[f"mulaw_{mu}"] = P.mulaw(x, mu)
        out[f"inv_mulaw_{mu}"] = P.inv_mulaw(x, mu)
        out[f"mulaw_quantize_{mu}"] = P.mulaw_quantize(x, mu)
    out["inv_mulaw_quantize_255"] = P.inv_mulaw_quantize(codes, 255)
    for coef in (0.85, 0.97):
        out[f"preemphasis_{coef}"] = P.preemphasis(sig, coef)
        out[f"inv_preemphasis_{coef}"] = P.inv_preemphasis(sig, coef)
    np.savez_compressed(os.path.join(HERE, "post_nnmnkwii.npz"), **out)
    print("wrote post_nnmnkwii.npz from nnmnkwii", out["version"])
    return True


def make_mel():
    try:
        import librosa
    except Exception as e:                                   # noqa: BLE001
        print(f"librosa is not importable ({type(e).__name__}: {e}) -- mel_librosa.npz NOT written; f4 stays parity-unpinned")
        return False
    from wavenet_vocoder_amd.audio import default_hparams
    hp = default_hparams()
    y = mel_signal()
    # the reference's recipe: logmelspectrogram(y, pad_mode="reflect") audio.py:101-109 -> _stft :128-132 (window="hann", center=True),
    # _linear_to_mel / _build_mel_basis :135-152
    from wavenet_vocoder_amd.audio import get_hop_size, get_win_length
    out = dict(version=str(librosa.__version__), y=y)
    basis = librosa.filters.mel(sr=hp.sample_rate, n_fft=hp.fft_size, fmin=hp.fmin, fmax=hp.fmax, n_mels=hp.num_mels)
    out["basis"] = basis.astype(np.float64)
    for pad_mode in ("reflect", "constant"):
        D = librosa.stft(y=y, n_fft=hp.fft_size, hop_length=get_hop_size(hp), win_length=get_win_length(hp), window=hp.window, pad_mode=pad_mode)
        out[f"stft_abs_{pad_mode}"] = np.abs(D).astype(np.float64)
        out[f"logmel_{pad_mode}"] = np.log10(np.maximum(np.dot(basis, np.abs(D)), 1e-10)).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "mel_librosa.npz"), **out)
    print("wrote mel_librosa.npz from librosa", librosa.__version__)
    return True


if __name__ == "__main__":
    a, b = make_post(), make_mel()
    sys.exit(0 if (a and b) else 3)
