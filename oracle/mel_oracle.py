"""CPU oracle of the mel front end.  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
product path.

Restates, in numpy float64, what the reference computes at

    audio.py:101-109   logmelspectrogram: D = _stft(y); S = _linear_to_mel(|D|); log10(max(S, 1e-10))
    audio.py:128-132   _stft = librosa.stft(y, n_fft=fft_size, hop_length, win_length, window="hann", pad_mode)   [center=True]
    audio.py:145-157   _linear_to_mel = np.dot(librosa.filters.mel(sample_rate, fft_size, fmin, fmax, n_mels), .)
    preprocess_normalize.py:44   scaler.transform(x)  (sklearn StandardScaler: (x - mean_) / scale_)

The arithmetic lives in **librosa** (setup.py:25, unpinned, not vendored under /root/reference, not installed here) and
**scikit-learn**; the functions below follow librosa's published definitions (librosa.core.spectrum.stft,
librosa.filters.mel with htk=False / norm="slaney", librosa.core.convert.hz_to_mel / mel_to_hz / mel_frequencies).

PARITY UNPINNED against librosa itself (it cannot be run here and the reference's tests hold no vector for this path).
What pins the restatement (tests/test_mel_cpu.py): the known answers printed in librosa's own docstrings (hz_to_mel(60) = 0.9,
mel_to_hz(3) = 200, mel_frequencies(n_mels=40) table, filters.mel(sr=22050, n_fft=2048)[0, 1] = 0.016), an independent
STFT (scipy.signal.stft on the same padded signal), unit filter areas, and scikit-learn's StandardScaler (installed) -- and, since
round 6 (tests/test_mel_second_source_cpu.py), a SECOND INDEPENDENT SOURCE: Hugging Face transformers.audio_utils (in the image; its
authors validate it against librosa for the Whisper / SpeechT5 feature extractors): the Slaney mel scale to 1e-12, the Slaney-normalised
filterbank at four geometries to 2e-9, the whole log-mel at the reference's hparams to 1e-7 (their FFT buffer is complex64), and
torch.stft as a third STFT to 1e-9.  Not librosa's own output -- the row stays "parity unpinned" -- but two restatements written by
different people from the same definitions agree.
"""
import numpy as np


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0):
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') -> (n_mels, 1 + n_fft // 2)."""
    if fmax is None:
        fmax = sr / 2.0
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, np.newaxis]


def hann_periodic(win_length):
    """scipy.signal.get_window('hann', win_length, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)


def stft(y, n_fft, hop_length, win_length, pad_mode="reflect"):
    """librosa.stft(y, n_fft, hop_length, win_length, window='hann', center=True, pad_mode) -> (1 + n_fft//2, frames)."""
    y = np.asarray(y, dtype=np.float64)
    w = hann_periodic(win_length)
    lpad = (n_fft - win_length) // 2
    w = np.pad(w, (lpad, n_fft - win_length - lpad))                      # librosa.util.pad_center
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    return np.fft.rfft(yp[idx] * w[None, :], axis=1).T


def logmelspectrogram(y, hp, pad_mode="reflect"):
    """audio.logmelspectrogram -> (num_mels, frames), float64."""
    hop = hp.hop_size if hp.hop_size is not None else int(hp.frame_shift_ms / 1000 * hp.sample_rate)
    win = hp.win_length if hp.win_length >= 0 else int(hp.win_length_ms / 1000 * hp.sample_rate)
    D = stft(y, hp.fft_size, hop, win, pad_mode)
    S = mel_filterbank(hp.sample_rate, hp.fft_size, hp.num_mels, hp.fmin, hp.fmax) @ np.abs(D)
    return np.log10(np.maximum(S, 1e-10))


def standard_scale(x, mean, scale):
    """StandardScaler.transform on (frames, num_mels) rows."""
    return (np.asarray(x, dtype=np.float64) - mean) / scale
