"""Writes tests/golden/mel_preset.npz: a seeded waveform and the oracle's log-mel / scaled features for it at the
reference's preset (hparams.py:32-44).  librosa cannot be run here, so these are ORACLE outputs (oracle/mel_oracle.py,
pinned by tests/test_mel_cpu.py), committed so that the GPU parity test also has a fixed vector.

    python tests/golden/make_mel_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mel_oracle as M  # noqa: E402
from wavenet_vocoder_amd.audio import default_hparams  # noqa: E402


def signal(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    chirp = 0.3 * np.sin(2 * np.pi * (200.0 * t + 3000.0 * t * t))
    return (chirp + 0.05 * rng.standard_normal(n)).astype(np.float32)


if __name__ == "__main__":
    hp = default_hparams()
    y = signal(5000, 7)
    S = M.logmelspectrogram(y, hp)
    rng = np.random.default_rng(8)
    mean = rng.normal(-2.0, 0.5, hp.num_mels)
    scale = rng.uniform(0.5, 1.5, hp.num_mels)
    feats = M.standard_scale(S.T, mean, scale)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_preset.npz")
    np.savez_compressed(out, y=y, logmel=S.astype(np.float32), mean=mean.astype(np.float32), scale=scale.astype(np.float32),
                        feats=feats.astype(np.float32))
    print(out, S.shape)
