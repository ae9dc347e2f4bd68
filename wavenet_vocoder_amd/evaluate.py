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


def _packed_local(model, mels, hparams, mine, speaker_ids=None) -> dict:
    """This rank's utterances ``mine`` as packed slots -> {utterance index: clipped float waveform on the CPU}.  The post-chain of
    synthesis.py:66-84 runs per UTTERANCE (the inverse pre-emphasis is an IIR: its state must not leak across a slot's boundaries) as
    soon as the utterance's launch is done (``sink``): only one launch's buffers live on the device, whatever the size of the job
    (a 256-way one-hot output is 1 KB per sample).  No collective in here: a rank may fall back to padded groups on its own."""
    from . import synthesis
    local = {}

    def sink(i, y):
        wav = synthesis.postprocess(y.unsqueeze(0), hparams).clamp_(-1.0, 1.0)
        local[i] = wav[0, : y.shape[-1]].detach().to("cpu")

    with torch.no_grad():
        sharding.synthesize_packed(model, mels, hop_size=hparams.hop_size, cin_pad=hparams.cin_pad, indices=mine, sink=sink,
                                   speaker_ids=speaker_ids, as_index=True)     # (one-hot models: classes, not 256-wide one-hot vectors)
    return local


def synthesize_dir(model, data_dir: str, dst_dir: str, hparams, *, num_utterances: int = -1,
                   speaker_id: Optional[int] = None, group=None,
                   synth_group: Optional[Callable[[torch.Tensor, List[int]], torch.Tensor]] = None,
                   packed: Optional[bool] = None) -> List[str]:
    """The main loop of evaluate.py (:155-251) without reference wavs: every ``*-feats.npy`` under ``data_dir`` becomes
    ``dst_dir/{name}_gen.wav``.  Utterances are sharded over the ranks of ``group`` (one process per GPU); rank 0
    writes the files and returns their paths (other ranks return []).  Utterances per launch: ``hparams.batch_size`` when it is
    set (the reference's recipes pass 32, egs/mol/run.sh:31), otherwise from the measured throughput curve
    (``sharding.auto_group_size``: up to 48 per GPU).

    ``packed``: run the job as PACKED SLOTS -- continuous batching, ``sharding.synthesize_packed``: no padding to a group's longest
    member, every waveform is what the utterance gives on its own -- instead of padded groups.  Packed slots draw their noise IN THE
    KERNEL (Philox streams addressed by utterance and step) and leave the kernel choice to the engine, so None (default) packs only when
    that is what the model is set to anyway -- ``model.rng == "philox"`` and ``model.kernel`` in (0, 2) --, nothing fixes the grouping
    (no ``synth_group``, no ``hparams.batch_size``), the job is larger than one launch per GPU and the ring kernel covers the model
    (speaker-conditioned models included since round 5: the utterances' speaker ids from train.txt).  ``packed=True`` asks for it
    explicitly (a ``rng = "replay"`` model then still draws in-kernel noise).  A rank whose packed run cannot proceed (model or device
    not covered, time-out, out of memory) runs its share as padded groups; every rank meets in the one gather at the end."""
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

    _, rank, world, lengths, mine = sharding._local_share(mels, hparams.hop_size, group)
    has_spk = bool(model is not None and model.has_speaker_embedding())      # (model None: the caller's synth_group does the synthesis)
    spk = [u.speaker_id for u in utts] if has_spk and all(u.speaker_id is not None for u in utts) else None
    if packed is None:
        # automatic only where it changes nothing the caller set: nothing fixes the grouping, the model draws its noise in the kernel
        # anyway (rng = "philox"; the class default "replay" reproduces the reference's CPU stream for a seed -- packed slots cannot) and
        # leaves the kernel choice to the engine, the job is larger than one launch per GPU, and the packed path covers the model
        # (decided from the configuration alone: identical on every rank)
        packed = (model is not None and synth_group is None and getattr(hparams, "batch_size", None) is None
                  and getattr(model, "rng", "replay") == "philox" and int(getattr(model, "kernel", 0)) in (0, 2)
                  and (not has_spk or spk is not None)
                  and len(mels) > sharding.packed_group_size(model) * world
                  and sharding.packed_unsupported_reason(model) is None)
    local = None
    if packed and model is not None:
        # the reference's sanity_check (train.py:72-87, called by batch_wavegen, synthesis.py:43-44), for the path that does not go
        # through batch_wavegen: a speaker-embedding model needs every utterance's speaker, any other model must not get one
        synthesis.sanity_check(model, mels[0], spk if has_spk else ([u.speaker_id for u in utts][0] if utts[0].speaker_id is not None else None))
    if packed:
        try:
            local = _packed_local(model, mels, hparams, mine, spk)
        except (NotImplementedError, TimeoutError, torch.cuda.OutOfMemoryError) as e:   # not a ring configuration (WNV_ERR_UNSUPPORTED) /
            # the ring does not fit or gave up waiting (WNV_ERR_TIMEOUT) / out of memory: THIS rank runs its share as padded groups --
            # neither path holds a collective, the one gather below is reached by every rank whichever path it took.  Anything else (an
            # invalid argument, a HIP fault, a bug in the segment maps) propagates: it would be masked by a silent re-run, and a job
            # must not mix packed and padded waveforms for one seed without the caller hearing about it.
            print(f"[wnv] rank {rank}: packed slots not used ({type(e).__name__}: {str(e)[:160]}); falling back to padded groups", flush=True)
            local = None
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
    if local is None:
        local = sharding.synthesize_local_padded(mels, synth_group or default_group, mine, lengths, cin_pad=hparams.cin_pad,
                                                 group_size=getattr(hparams, "batch_size", None))
    wavs = sharding.gather_results(local, len(mels), group=group, gather_to=0)
    if wavs is None:
        return []
    os.makedirs(dst_dir, exist_ok=True)
    out = []
    for u, w in zip(utts, wavs):
        path = join(dst_dir, output_name(u))
        write_wav(path, hparams.sample_rate, to_int16(w.numpy().astype(np.float32)))
        out.append(path)
    return out
