"""CPU (-m "not gpu"): the mel front end's oracle (oracle/mel_oracle.py) against a SECOND INDEPENDENT SOURCE.

librosa -- where the reference's arithmetic lives (audio.py:101-157) -- cannot be installed here, so the oracle stays "parity unpinned"
against librosa itself.  What IS in the image is Hugging Face `transformers`, whose `audio_utils` module is a published, independently
written restatement of the same definitions (Slaney mel scale and area normalisation, centred reflect-padded STFT with a periodic Hann
window, log10 with a floor) that its authors validate against librosa for the Whisper / SpeechT5 feature extractors.  Two restatements
written by different people from the same definitions agreeing to rounding is not the reference's own output -- but it is evidence of a
different kind from the oracle agreeing with itself.  Also torch.stft (a third STFT).  Skipped when `transformers` is absent."""
import numpy as np
import pytest

from oracle import mel_oracle as M
from wavenet_vocoder_amd.audio import default_hparams, get_hop_size, get_win_length

A = pytest.importorskip("transformers.audio_utils")


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(24000, 1024, 80, 80.0, 7600.0), (22050, 1024, 80, 125.0, 7600.0), (22050, 2048, 128, 0.0, None),
                                                     (16000, 512, 40, 0.0, 8000.0)])
def test_filterbank_is_the_slaney_filterbank_of_a_second_source(sr, n_fft, n_mels, fmin, fmax):
    ours = M.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)                                    # (n_mels, 1 + n_fft // 2)
    theirs = A.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin,
                               max_frequency=sr / 2.0 if fmax is None else fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney")
    assert theirs.shape == (1 + n_fft // 2, n_mels)
    np.testing.assert_allclose(ours, theirs.T, rtol=0, atol=2e-9)                              # (theirs is float64 arithmetic too)


def test_mel_scale_is_the_slaney_scale_of_a_second_source():
    f = np.array([0.0, 60.0, 200.0, 999.0, 1000.0, 1001.0, 4000.0, 7600.0, 11025.0])
    np.testing.assert_allclose(M.hz_to_mel(f), A.hertz_to_mel(f, mel_scale="slaney"), rtol=1e-12, atol=1e-12)
    m = np.array([0.0, 0.9, 3.0, 14.9, 15.0, 15.1, 30.0, 45.0])
    np.testing.assert_allclose(M.mel_to_hz(m), A.mel_to_hertz(m, mel_scale="slaney"), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("pad_mode", ["reflect", "constant"])
def test_logmel_against_a_second_source(pad_mode):
    """audio.logmelspectrogram (audio.py:101-109) at the reference's hparams: |STFT| -> Slaney filterbank -> log10(max(., 1e-10))."""
    hp = default_hparams()
    hop, win = get_hop_size(hp), get_win_length(hp)
    assert win == hp.fft_size                                                                  # (hparams.py: win_length = fft_size = 1024)
    rng = np.random.default_rng(5)
    y = (rng.standard_normal(24000) * 0.1 + 0.3 * np.sin(2 * np.pi * 440.0 / hp.sample_rate * np.arange(24000))).astype(np.float64)
    ours = M.logmelspectrogram(y, hp, pad_mode=pad_mode)                                       # (num_mels, frames)
    fb = A.mel_filter_bank(1 + hp.fft_size // 2, hp.num_mels, hp.fmin, hp.fmax, hp.sample_rate, norm="slaney", mel_scale="slaney")
    theirs = A.spectrogram(y, A.window_function(win, "hann", periodic=True), frame_length=win, hop_length=hop, fft_length=hp.fft_size,
                           power=1.0, center=True, pad_mode=pad_mode, mel_filters=fb, mel_floor=1e-10, log_mel="log10", dtype=np.float64)
    assert theirs.shape == ours.shape
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=1e-7)                                # (their FFT buffer is complex64: ~2e-8 on the log)


def test_stft_against_torch():
    import torch
    hp = default_hparams()
    hop, win = get_hop_size(hp), get_win_length(hp)
    rng = np.random.default_rng(6)
    y = rng.standard_normal(9000)
    ours = M.stft(y, hp.fft_size, hop, win)
    theirs = torch.stft(torch.from_numpy(y), hp.fft_size, hop_length=hop, win_length=win, window=torch.hann_window(win, periodic=True, dtype=torch.float64),
                        center=True, pad_mode="reflect", return_complex=True).numpy()
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=1e-9)
