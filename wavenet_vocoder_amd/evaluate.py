"""Host mirror of the reference's ``evaluate.py`` front end: feature files -> padded batches ->
sharded synthesis -> trimmed, clipped int16 wav files.

    reference                                          here
    _NPYDataSource.collect_files   train.py:172-216    collect_features()   (train.txt metadata or *-feats.npy glob)
    dummy_collate / _pad_2d        evaluate.py:51-59   sharding.pad_group() (zero-pad to the batch maximum)
    F.pad(c, replicate, cin_pad)   evaluate.py:163-164 sharding.pad_group()
    input_lengths, gen[:length]    evaluate.py:53,215  lengths from the unpadded feature, trimmed per utterance
    np.clip + to_int16 + wavfile   evaluate.py:238-241 wnv_postprocess (clip, int16) + write_wav()
    "{name}_gen.wav" naming        evaluate.py:224-236 output_name()

The DataLoader / nnmnkwii FileSourceDataset machinery is replaced by a plain list of utterances handed to
``sharding.synthesize_sharded`` (one process per GPU, longest-first assignment, groups of <= batch_size).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from glob import glob
from os.path import basename, exists, join, splitext
from typing import Callable, List, Optional

import numpy as np
import torch

from . import sharding

__all__ = ["Utterance", "collect_features", "load_features", "output_name", "write_wav", "to_int16", "synthesize_dir"]


@dataclass
class Utterance:
    path: str                      # the *-feats.npy file
    frames: Optional[int] = None   # from train.txt (col 2) when present
    speaker_id: Optional[int] = None


def collect_features(data_dir: str, speaker_id: Optional[int] = None, num_utterances: int = -1) -> List[Utterance]:
    """MelSpecDataSource(max_steps=None).collect_files: ``train.txt`` lines ``wave|feats|frames|text[|speaker]`` when the
    file exists (train.py:173-216, col 1), else the sorted ``*-feats.npy`` glob (train.py:175-176)."""
    meta = join(data_dir, "train.txt")
    utts: List[Utterance] = []
    if not exists(meta):
        utts = [Utterance(p) for p in sorted(glob(join(data_dir, "*-feats.npy")))]
    else:
        with open(meta, "rb") as f:
            lines = [ln.decode("utf-8").rstrip("\n").split("|") for ln in f.readlines() if ln.strip()]
        assert all(len(l) in (4, 5) for l in lines), "train.txt: expected 4 or 5 '|'-separated columns"
        multi = len(lines[0]) == 5
        for l in lines:
            sid = int(l[-1]) if multi else None
            if speaker_id is not None and multi and sid != speaker_id:
                continue
            utts.append(Utterance(join(data_dir, l[1]), int(l[2]), sid if speaker_id is None else None))
    if num_utterances > 0:
        utts = utts[:num_utterances]
    return utts


def load_features(u: Utterance, cin_channels: int) -> torch.Tensor:
    """(frames, cin) float32 on disk -> (cin, frames) tensor; the channel check of evaluate.py:79-82."""
    x = np.load(u.path)
    if x.ndim != 2 or x.shape[-1] != cin_channels:
        raise RuntimeError("Invalid cin_channnels {}. Expectd to be {}.".format(cin_channels, x.shape[-1]))
    return torch.from_numpy(np.ascontiguousarray(x.T, dtype=np.float32))


def output_name(u: Utterance) -> str:
    """evaluate.py:221-236: ``{name}_gen.wav`` / ``speaker{g}_{name}_gen.wav`` with '-feats' stripped."""
    name = splitext(basename(u.path))[0].replace("-feats", "")
    return f"{name}_gen.wav" if u.speaker_id is None else f"speaker{u.speaker_id}_{name}_gen.wav"


def to_int16(x: np.ndarray) -> np.ndarray:
    """evaluate.py:43-48."""
    if x.dtype == np.int16:
        return x
    assert x.dtype == np.float32
    assert x.min() >= -1 and x.max() <= 1.0
    return (x * 32767).astype(np.int16)


def write_wav(path: str, sample_rate: int, pcm: np.ndarray) -> None:
    """Mono 16-bit PCM RIFF file, byte-identical to ``scipy.io.wavfile.write(path, rate, int16_array)`` (evaluate.py:240)."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


def _synthesize_packed(model, mels, hparams, group):
    """The job as packed slots: every rank packs its share (longest-first over the ranks), the post-chain of synthesis.py:66-84 runs
    per UTTERANCE (the inverse pre-emphasis is an IIR: its state must not leak across a slot's boundaries) on padded batches of the
    finished waveforms, rank 0 gathers.  Returns the list of clipped float waveforms on rank 0, None elsewhere."""
    import torch.distributed as dist
    from . import synthesis
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    model.eval()
    lengths = [int(m.shape[-1]) * hparams.hop_size for m in mels]
    mine = sharding.lpt_assign(lengths, world)[rank]
    with torch.no_grad():
        outs = sharding.synthesize_packed(model, mels, hop_size=hparams.hop_size, cin_pad=hparams.cin_pad, indices=mine)
    local = {}
    order = sorted(range(len(mine)), key=lambda k: -lengths[mine[k]])
    for a in range(0, len(order), 32):                              # post-chain on padded batches of neighbouring length
        grp = order[a:a + 32]
        Tm = max(lengths[mine[k]] for k in grp)
        y = torch.zeros(len(grp), outs[grp[0]].shape[0], Tm, device=outs[grp[0]].device)
        for row, k in enumerate(grp):
            y[row, :, :lengths[mine[k]]] = outs[k]
        wav = synthesis.postprocess(y, hparams).clamp_(-1.0, 1.0)
        for row, k in enumerate(grp):
            local[mine[k]] = wav[row, :lengths[mine[k]]].detach().to("cpu")
    if not distributed:
        return [local[i] for i in range(len(mels))]
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0, group=group)
    if rank != 0:
        return None
    merged = {}
    for part in parts:
        merged.update(part)
    return [merged[i] for i in range(len(mels))]


def synthesize_dir(model, data_dir: str, dst_dir: str, hparams, *, num_utterances: int = -1,
                   speaker_id: Optional[int] = None, group=None,
                   synth_group: Optional[Callable[[torch.Tensor, List[int]], torch.Tensor]] = None,
                   packed: Optional[bool] = None) -> List[str]:
    """The main loop of evaluate.py (:155-251) without reference wavs: every ``*-feats.npy`` under ``data_dir`` becomes
    ``dst_dir/{name}_gen.wav``.  Utterances are sharded over the ranks of ``group`` (one process per GPU); rank 0
    writes the files and returns their paths (other ranks return []).  Utterances per launch: ``hparams.batch_size`` when it is
    set (the reference's recipes pass 32, egs/mol/run.sh:31), otherwise from the measured throughput curve
    (``sharding.auto_group_size``: up to 48 per GPU).

    ``packed`` (round 4): run the job as PACKED SLOTS -- continuous batching, ``sharding.synthesize_packed``: no padding to a group's
    longest member, every waveform is what the utterance gives on its own -- instead of padded groups.  None (default): when nothing
    fixes the grouping (no ``synth_group``, no ``hparams.batch_size``), the model has no speaker embedding and the ring kernel takes
    it; a model or device the packed path does not cover falls back to padded groups."""
    from . import synthesis
    utts = collect_features(data_dir, speaker_id=speaker_id, num_utterances=num_utterances)
    assert len(utts) > 0, f"no *-feats.npy under {data_dir}"
    mels = [load_features(u, hparams.cin_channels) for u in utts]

    def default_group(c: torch.Tensor, idx: List[int]) -> torch.Tensor:
        g = None
        if utts[idx[0]].speaker_id is not None:
            g = torch.tensor([utts[i].speaker_id for i in idx], dtype=torch.long)
        wav = synthesis.batch_wavegen(model, c=c, g=g, hparams=hparams)          # (B, T) float32, post-chain applied
        return torch.from_numpy(np.clip(wav, -1.0, 1.0))                         # evaluate.py:238

    wavs = None
    if packed is None:                                    # (nothing to pack while every utterance can have a row of its own)
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        packed = synth_group is None and getattr(hparams, "batch_size", None) is None and not model.has_speaker_embedding() \
            and next(model.parameters()).is_cuda and len(mels) > sharding.THROUGHPUT_GROUP * world
    if packed:
        try:
            wavs = _synthesize_packed(model, mels, hparams, group)
        except (NotImplementedError, TimeoutError) as e:          # not a ring configuration / the ring does not fit: padded groups
            print(f"[wnv] packed slots not used ({str(e)[:120]}); falling back to padded groups", flush=True)
            wavs = None
            packed = False
    if not packed:
        wavs = sharding.synthesize_sharded(mels, synth_group or default_group, hop_size=hparams.hop_size,
                                           cin_pad=hparams.cin_pad, group_size=getattr(hparams, "batch_size", None), group=group)
    if wavs is None:
        return []
    os.makedirs(dst_dir, exist_ok=True)
    out = []
    for u, w in zip(utts, wavs):
        path = join(dst_dir, output_name(u))
        write_wav(path, hparams.sample_rate, to_int16(w.numpy().astype(np.float32)))
        out.append(path)
    return out
