"""CPU (-m "not gpu"): pins of the mel front end's oracle (oracle/mel_oracle.py) and the host-built
filterbank of the C ABI.  librosa -- where the arithmetic lives -- is not installed, so the oracle is anchored on the known
answers of librosa's own docstrings, an independent STFT (scipy), closed-form properties and scikit-learn's scaler."""
import ctypes as C

import numpy as np
import pytest

from oracle import mel_oracle as M
from wavenet_vocoder_amd import _lib
from wavenet_vocoder_amd.audio import default_hparams, get_hop_size, get_win_length


def test_mel_scale_known_answers_from_librosa_docs():
    assert abs(M.hz_to_mel(60) - 0.9) < 1e-12
    np.testing.assert_allclose(M.hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(M.mel_to_hz(3) - 200.0) < 1e-9
    np.testing.assert_allclose(M.mel_to_hz([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], atol=5e-4)
    want = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856,
            1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686,
            2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009,
            7754.107, 8467.272, 9246.028, 10096.408, 11025.]
    np.testing.assert_allclose(M.mel_frequencies(40), want, atol=6e-4)         # librosa.mel_frequencies(n_mels=40)
    x = np.linspace(0, 11025, 1000)
    np.testing.assert_allclose(M.mel_to_hz(M.hz_to_mel(x)), x, atol=1e-8)


def test_filterbank_known_answer_and_properties():
    fb = M.mel_filterbank(22050, 2048)                                           # librosa.filters.mel(sr=22050, n_fft=2048)
    assert fb.shape == (128, 1025)
    assert round(fb[0, 1], 3) == 0.016 and fb[0, 0] == 0.0
    hp = default_hparams()
    fb = M.mel_filterbank(hp.sample_rate, hp.fft_size, hp.num_mels, hp.fmin, hp.fmax)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    df = hp.sample_rate / hp.fft_size
    freqs = np.arange(513) * df
    assert fb[:, freqs < hp.fmin].sum() == 0 and fb[:, freqs > hp.fmax].sum() == 0
    # slaney norm: every triangle has unit area on the Hz axis (the discrete sum approaches it for the wide filters)
    np.testing.assert_allclose(fb[40:].sum(1) * df, 1.0, atol=0.02)
    assert (np.diff(fb.argmax(1)) >= 0).all()                                   # peaks move up with the filter index


def test_stft_against_scipy():
    from scipy.signal import stft as sp_stft
    rng = np.random.default_rng(0)
    for n_fft, hop, win, n in [(1024, 256, 1024, 5000), (1024, 256, 800, 3000), (512, 128, 512, 2049)]:
        y = rng.standard_normal(n)
        D = M.stft(y, n_fft, hop, win, "reflect")
        assert D.shape == (n_fft // 2 + 1, 1 + n // hop)
        w = np.pad(M.hann_periodic(win), ((n_fft - win) // 2, n_fft - win - (n_fft - win) // 2))
        yp = np.pad(y, n_fft // 2, mode="reflect")
        _, _, Z = sp_stft(yp, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False, return_onesided=True)
        np.testing.assert_allclose(D, Z[:, :D.shape[1]] * w.sum(), atol=1e-9)
    # constant padding = the zero-padded signal
    D0 = M.stft(y, 512, 128, 512, "constant")
    np.testing.assert_allclose(D0[:, 0], np.fft.rfft(np.concatenate([np.zeros(256), y[:256]]) * M.hann_periodic(512)), atol=1e-10)


def test_logmel_closed_forms():
    hp = default_hparams()
    assert np.all(M.logmelspectrogram(np.zeros(4000), hp) == -10.0)              # log10 of the 1e-10 floor (audio.py:108)
    rng = np.random.default_rng(1)
    y = rng.standard_normal(6000) * 0.1
    a, b = M.logmelspectrogram(y, hp), M.logmelspectrogram(2 * y, hp)
    np.testing.assert_allclose(b - a, np.log10(2.0), atol=1e-12)                # |D| is homogeneous
    # a pure tone at a bin centre lands in the filters that cover that frequency
    k = 100
    tone = np.sin(2 * np.pi * k / hp.fft_size * np.arange(8000))
    S = M.logmelspectrogram(tone, hp)
    f = k * hp.sample_rate / hp.fft_size
    edges = M.mel_frequencies(hp.num_mels + 2, hp.fmin, hp.fmax)
    top = int(S[:, 10].argmax())
    assert edges[top] <= f <= edges[top + 2]


def test_standard_scale_is_sklearn():
    from sklearn.preprocessing import StandardScaler
    rng = np.random.default_rng(2)
    x = rng.normal(-3, 2, (500, 80))
    sc = StandardScaler().fit(x)
    np.testing.assert_allclose(M.standard_scale(x, sc.mean_, sc.scale_), sc.transform(x), atol=1e-12)


def test_golden_fixture_is_the_oracle():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mel_preset.npz"))
    S = M.logmelspectrogram(z["y"], default_hparams())
    np.testing.assert_allclose(S, z["logmel"], atol=2e-6)
    np.testing.assert_allclose(M.standard_scale(S.T, z["mean"], z["scale"]), z["feats"], atol=1e-5)


def _cfg(hp, pad_mode=1):
    c = _lib.MelConfig()
    c.sample_rate, c.fft_size, c.hop_size, c.win_length = hp.sample_rate, hp.fft_size, get_hop_size(hp), get_win_length(hp)
    c.num_mels, c.fmin, c.fmax, c.pad_mode, c.floor = hp.num_mels, hp.fmin, hp.fmax or 0.0, pad_mode, 1e-10
    return c


@pytest.mark.parametrize("over", [{}, {"fft_size": 2048, "win_length": 1200, "hop_size": 300, "num_mels": 128, "fmin": 0, "fmax": None},
                                  {"sample_rate": 16000, "fft_size": 512, "win_length": -1, "win_length_ms": 25.0, "hop_size": None,
                                   "frame_shift_ms": 10.0, "num_mels": 40, "fmin": 60, "fmax": 7600}])
def test_host_filterbank_of_the_c_abi_matches_the_oracle(over):
    """wnv_mel_create(device = -1) builds window/filterbank on the host only: no GPU needed."""
    hp = default_hparams(**over)
    lib = _lib.lib()
    cfg = _cfg(hp)
    h = C.c_void_p()
    _lib.check(lib.wnv_mel_create(C.byref(cfg), -1, C.byref(h)))
    try:
        fb = np.empty((hp.num_mels, hp.fft_size // 2 + 1), np.float32)
        _lib.check(lib.wnv_mel_basis(h, fb.ctypes.data_as(C.c_void_p)))
        want = M.mel_filterbank(hp.sample_rate, hp.fft_size, hp.num_mels, hp.fmin, hp.fmax)
        np.testing.assert_allclose(fb, want, rtol=1e-6, atol=1e-9)
        for n in (0, 1, 255, 256, 257, 22050):
            assert lib.wnv_mel_frames(C.byref(cfg), n) == 1 + n // cfg.hop_size
        a = _lib.LogmelArgs()
        a.B, a.n, a.wav, a.out = 1, 4000, 1, 1
        with pytest.raises(ValueError, match="no CPU path"):
            _lib.check(lib.wnv_logmel(h, C.byref(a)))
    finally:
        lib.wnv_mel_destroy(h)


def test_mel_create_argument_errors():
    lib = _lib.lib()
    h = C.c_void_p()
    for over, exc in [({"fft_size": 1000}, NotImplementedError), ({"fft_size": 8192}, NotImplementedError),
                      ({"win_length": 2048}, ValueError), ({"fmax": 20000}, ValueError), ({"num_mels": 0}, ValueError)]:
        cfg = _cfg(default_hparams(**over))
        with pytest.raises(exc):
            _lib.check(lib.wnv_mel_create(C.byref(cfg), -1, C.byref(h)))
    assert lib.wnv_mel_frames(None, 10) == -1


def test_host_mirror_argument_checks_need_no_gpu():
    """audio.MelFrontEnd refuses a CPU device (no CPU path), other windows and pad modes, and fmax above Nyquist (audio.py:153-154);
    get_hop_size / get_win_length follow audio.py:112-125."""
    from wavenet_vocoder_amd import audio
    hp = default_hparams()
    with pytest.raises(RuntimeError, match="no CPU path"):
        audio.MelFrontEnd(hp, device="cpu")
    with pytest.raises(NotImplementedError):
        audio.MelFrontEnd(default_hparams(window="hamming"), device="cpu")
    with pytest.raises(NotImplementedError):
        audio.MelFrontEnd(hp, device="cpu", pad_mode="edge")
    with pytest.raises(AssertionError):
        audio.MelFrontEnd(default_hparams(fmax=12000), device="cpu")
    assert get_hop_size(default_hparams(hop_size=None, frame_shift_ms=12.5)) == int(12.5 / 1000 * 22050)
    assert get_win_length(default_hparams(win_length=-1, win_length_ms=50.0)) == int(50.0 / 1000 * 22050)
    assert get_hop_size(hp) == 256 and get_win_length(hp) == 1024
