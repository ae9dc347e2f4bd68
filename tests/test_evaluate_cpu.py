"""CPU: the evaluate.py front end -- file collection, naming, padding/length bookkeeping, the wav
writer (byte-for-byte against scipy.io.wavfile) and the whole directory loop with a stand-in synthesiser."""
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from wavenet_vocoder_amd import evaluate as E


def make_dump(tmp_path, n=5, cin=80, with_meta=False, multi=False):
    rng = np.random.default_rng(0)
    frames = [7, 12, 5, 9, 12][:n]
    lines = []
    for i, fr in enumerate(frames):
        np.save(tmp_path / f"utt{i:02d}-feats.npy", rng.standard_normal((fr + 4, cin)).astype(np.float32))
        np.save(tmp_path / f"utt{i:02d}-wave.npy", np.zeros(fr * 256, dtype=np.float32))
        cols = [f"utt{i:02d}-wave.npy", f"utt{i:02d}-feats.npy", str(fr + 4), "dummy"]
        if multi:
            cols.append(str(i % 2))
        lines.append("|".join(cols))
    if with_meta:
        (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
    return frames


def test_collect_glob_sorted_and_named(tmp_path):
    make_dump(tmp_path)
    utts = E.collect_features(str(tmp_path))
    assert [os.path.basename(u.path) for u in utts] == [f"utt{i:02d}-feats.npy" for i in range(5)]
    assert E.output_name(utts[3]) == "utt03_gen.wav"
    assert len(E.collect_features(str(tmp_path), num_utterances=2)) == 2


def test_collect_from_metadata_with_speakers(tmp_path):
    make_dump(tmp_path, with_meta=True, multi=True)
    utts = E.collect_features(str(tmp_path))
    assert [u.speaker_id for u in utts] == [0, 1, 0, 1, 0] and utts[1].frames == 16
    assert E.output_name(utts[1]) == "speaker1_utt01_gen.wav"
    only = E.collect_features(str(tmp_path), speaker_id=1)
    assert [os.path.basename(u.path) for u in only] == ["utt01-feats.npy", "utt03-feats.npy"]
    assert all(u.speaker_id is None for u in only)          # filtered: used as a single-speaker set (train.py:201-209)


def test_load_features_checks_channels(tmp_path):
    make_dump(tmp_path, n=1, cin=20)
    u = E.collect_features(str(tmp_path))[0]
    assert E.load_features(u, 20).shape == (20, 11)
    with pytest.raises(RuntimeError, match="Invalid cin_channnels"):
        E.load_features(u, 80)


def test_write_wav_matches_scipy(tmp_path):
    pcm = (np.sin(np.arange(1000) * 0.1) * 20000).astype(np.int16)
    E.write_wav(str(tmp_path / "a.wav"), 22050, pcm)
    buf = io.BytesIO()
    wavfile.write(buf, 22050, pcm)
    assert (tmp_path / "a.wav").read_bytes() == buf.getvalue()
    rate, back = wavfile.read(str(tmp_path / "a.wav"))
    assert rate == 22050 and np.array_equal(back, pcm)


def test_to_int16_contract():
    assert E.to_int16(np.array([0.5, -1.0, 1.0], dtype=np.float32)).tolist() == [16383, -32767, 32767]
    with pytest.raises(AssertionError):
        E.to_int16(np.array([1.01], dtype=np.float32))


def test_directory_loop_lengths_and_padding(tmp_path):
    frames = make_dump(tmp_path)
    hp = SimpleNamespace(cin_channels=80, hop_size=256, cin_pad=2, batch_size=2, sample_rate=24000)
    seen = []

    def fake_synth(c, idx):      # (B, cin, max_frames + 2 cin_pad) -> (B, T)
        seen.append((tuple(c.shape), list(idx)))
        B, _, F = c.shape
        T = (F - 2 * hp.cin_pad) * hp.hop_size
        return torch.stack([torch.full((T,), 0.01 * (i + 1)) for i in idx])

    paths = E.synthesize_dir(None, str(tmp_path), str(tmp_path / "out"), hp, synth_group=fake_synth)
    assert [os.path.basename(p) for p in paths] == [f"utt{i:02d}_gen.wav" for i in range(5)]
    for i, p in enumerate(paths):
        rate, w = wavfile.read(p)
        # the features on disk already carry their 2*cin_pad context frames (train.py:218-226); the synthesised length is
        # frames_on_disk * hop before the extra replicate padding of evaluate.py:163-164
        assert rate == 24000 and len(w) == (frames[i] + 4) * 256
        assert np.all(w == int(np.float32(0.01 * (i + 1)) * 32767))
    # groups are packed longest-first in pairs and padded to their longest member + 2*cin_pad replicate frames
    assert all(s[0][0] <= 2 for s in seen) and sorted(i for s in seen for i in s[1]) == [0, 1, 2, 3, 4]
    assert seen[0][0][2] == max(frames) + 4 + 4


def test_packed_directory_loop_keeps_the_references_sanity_check(tmp_path):
    """train.sanity_check (train.py:72-87) guards batch_wavegen (synthesis.py:43-44); the packed path does not go through batch_wavegen, so
    synthesize_dir applies it itself: a speaker-embedding model without speaker ids (or the other way round) is the reference's RuntimeError,
    raised before anything touches a device."""
    import numpy as np
    from tests._configs import build
    from wavenet_vocoder_amd import evaluate as E, synthesis
    for i in range(2):
        np.save(tmp_path / f"u{i}-feats.npy", np.zeros((3, 80), np.float32))
    h = synthesis.default_hparams(cin_channels=80, cin_pad=2, hop_size=256, batch_size=None, sample_rate=24000)
    with pytest.raises(RuntimeError, match="expects speaker embedding"):
        E.synthesize_dir(build("cfg4_mol_multispeaker"), str(tmp_path), str(tmp_path / "o"), h, packed=True)
    (tmp_path / "train.txt").write_text("a|u0-feats.npy|3|t|1\nb|u1-feats.npy|3|t|2\n")
    with pytest.raises(RuntimeError, match="expects no speaker embedding"):
        E.synthesize_dir(build("cfg2_mol"), str(tmp_path), str(tmp_path / "o"), h, packed=True)


def test_packed_fallback_is_for_unsupported_timeout_and_oom_only(tmp_path, monkeypatch, capsys):
    """A rank whose packed run cannot proceed re-runs its share as padded groups -- but only for what "cannot proceed" means: the ring does
    not take the configuration (NotImplementedError = WNV_ERR_UNSUPPORTED), it gave up waiting (TimeoutError = WNV_ERR_TIMEOUT), or the
    launch's buffers did not fit (OutOfMemoryError).  Anything else -- an invalid argument, a HIP fault, a bug in the segment maps -- reaches
    the caller (the blanket RuntimeError catch masked real faults and could mix packed and padded waveforms under one seed)."""
    make_dump(tmp_path)
    hp = SimpleNamespace(cin_channels=80, hop_size=256, cin_pad=2, batch_size=2, sample_rate=24000)

    def fake_synth(c, idx):
        B, _, F = c.shape
        return torch.zeros(B, (F - 2 * hp.cin_pad) * hp.hop_size)

    def fault(*a, **k):
        raise RuntimeError("HIP fault in the packed launch")
    monkeypatch.setattr(E, "_packed_local", fault)
    with pytest.raises(RuntimeError, match="HIP fault"):
        E.synthesize_dir(None, str(tmp_path), str(tmp_path / "o1"), hp, synth_group=fake_synth, packed=True)
    for exc in (NotImplementedError("packed slots: not a ring configuration"), TimeoutError("ring kernel gave up waiting"),
                torch.cuda.OutOfMemoryError("out of memory")):
        def refuse(*a, _e=exc, **k):
            raise _e
        monkeypatch.setattr(E, "_packed_local", refuse)
        paths = E.synthesize_dir(None, str(tmp_path), str(tmp_path / "o2"), hp, synth_group=fake_synth, packed=True)
        assert len(paths) == 5 and "falling back to padded groups" in capsys.readouterr().out
