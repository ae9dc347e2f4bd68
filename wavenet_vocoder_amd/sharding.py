"""Utterance-level sharding of a synthesis job over the GPUs of one node.

Utterances share only read-only weights and the sample loop of one utterance never talks to another, so
the path shards with NO data-path collective: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests), every rank
  1. holds a replica of the weights (loaded from the checkpoint, or broadcast once from rank 0),
  2. takes its share of the utterances (longest-processing-time-first over total samples),
  3. packs them into groups of similar length (a group runs to its longest member, exactly like the reference's
     zero-padded batches, evaluate.py:55-57,215); the group size comes from the MEASURED throughput curve of the ring
     kernel unless the caller fixes it (``auto_group_size``: the reference's recipes synthesise 32 utterances per batch,
     egs/mol/run.sh:31),
  4. synthesises group by group, trims every waveform to its true length,
and the results are gathered to rank 0 (variable-length, one object gather at the very end).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

__all__ = ["lpt_assign", "pack_groups", "pad_group", "broadcast_weights", "synthesize_sharded", "auto_group_size",
           "padding_loss", "THROUGHPUT_GROUP", "plan_slots", "synthesize_packed", "synthesize_local_padded", "gather_results",
           "packed_group_size", "packed_unsupported_reason", "upsample_each"]

# Utterances per launch at which the ring kernel's aggregate rate stops growing.  Round 4 (profiles/r04_final_numbers.txt; kSamples/s
# per GPU at B = 8 / 16 / 32 / 40 / 48 / 56 / 64): 504 / 1000 / 2014 / 2257 / 2712 / 2388 / 2703 -- up to 32 utterances every utterance advances at
# the chain latency (~2.6x real time at 24 kHz), 48 still gain a third in aggregate at 2.35x real time each, beyond that nothing is
# gained (a ring carries whole utterances: 56 = seven per ring run slower than 48 = six).  (Round 3: the plateau began at 32 -- 1.98
# MSamples/s -- because the tap workgroups' passes bound the step there.)
THROUGHPUT_GROUP = 48


def auto_group_size(n_pending: int) -> int:
    """Group size for ``n_pending`` utterances waiting on ONE GPU when the caller did not fix one: everything in one launch while it
    fits the plateau (few utterances: a group of <= 8 is one utterance per ring, the lowest latency), groups of ``THROUGHPUT_GROUP``
    otherwise -- 40 pending utterances run as ONE launch (17.7 us per step against 2 x 16.1 for 32 + 8), 100 as 48 + 48 + 4."""
    return max(1, min(int(n_pending), THROUGHPUT_GROUP))


def padding_loss(groups: Sequence[Sequence[int]], lengths: Sequence[int]) -> float:
    """Fraction of the synthesised samples that is padding: a group runs to its longest member (evaluate.py:55-57,215)."""
    padded = sum(len(g) * max(int(lengths[i]) for i in g) for g in groups if len(g))
    true = sum(int(lengths[i]) for g in groups for i in g)
    return 0.0 if padded == 0 else 1.0 - true / padded


def lpt_assign(lengths: Sequence[int], n_bins: int) -> List[List[int]]:
    """Longest-processing-time-first: utterance indices per bin, balanced on total length."""
    bins: List[List[int]] = [[] for _ in range(n_bins)]
    load = [0] * n_bins
    for i in sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i)):
        k = min(range(n_bins), key=lambda k: (load[k], k))
        bins[k].append(i)
        load[k] += int(lengths[i])
    return bins


def pack_groups(indices: Sequence[int], lengths: Sequence[int], group_size: Optional[int] = None) -> List[List[int]]:
    """Groups of at most ``group_size`` utterances of neighbouring length (descending).  ``group_size=None``: chosen from the
    measured throughput curve (``auto_group_size``)."""
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i))
    if group_size is None:
        group_size = auto_group_size(len(order))
    if int(group_size) < 1:
        raise ValueError(f"group_size must be >= 1, got {group_size}")
    group_size = int(group_size)
    return [order[k:k + group_size] for k in range(0, len(order), group_size)]


def pad_group(mels: Sequence[torch.Tensor], cin_pad: int) -> torch.Tensor:
    """Stack (cin, frames_i) mels into (B, cin, max_frames + 2*cin_pad): zero-pad to the longest
    (evaluate.py:55-57) then replicate-pad ``cin_pad`` context frames at both ends (evaluate.py:163-164)."""
    fmax = max(int(m.shape[-1]) for m in mels)
    out = torch.zeros(len(mels), mels[0].shape[0], fmax, dtype=torch.float32)
    for i, m in enumerate(mels):
        out[i, :, : m.shape[-1]] = m
    if cin_pad > 0:
        out = torch.nn.functional.pad(out, (cin_pad, cin_pad), mode="replicate")
    return out


def broadcast_weights(model: torch.nn.Module, src: int = 0, group=None) -> None:
    """One-time replication of the weights from ``src`` (one flat buffer, one collective: the whole egs/mol
    model is 14.8 MB, far below anything worth bucketing).  The copy goes through ``Parameter.copy_`` under
    ``no_grad`` so that the tensors' version counters move (the packed-weight caches of the engine key on them), and
    any engine built before the broadcast is dropped explicitly as well."""
    import torch.distributed as dist
    params = list(model.parameters())
    with torch.no_grad():
        flat = torch.cat([p.reshape(-1) for p in params])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n
    for m in model.modules():
        if hasattr(m, "invalidate_engine"):
            m.invalidate_engine()


def _local_share(mels, hop_size, group=None):
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    lengths = [int(m.shape[-1]) * hop_size for m in mels]
    return distributed, rank, world, lengths, lpt_assign(lengths, world)[rank]


def synthesize_local_padded(mels, synth_group, mine, lengths, *, cin_pad: int, group_size: Optional[int] = None,
                            stats: Optional[dict] = None) -> dict:
    """This rank's utterances ``mine`` as padded groups of neighbouring length: {utterance index: trimmed CPU waveform}.  No collective."""
    local = {}
    groups = pack_groups(mine, lengths, group_size)
    if stats is not None:
        stats.update(groups=[list(g) for g in groups], true_samples=sum(lengths[i] for i in mine),
                     padded_samples=sum(len(g) * max(lengths[i] for i in g) for g in groups if len(g)),
                     padding_loss=padding_loss(groups, lengths))
    for grp in groups:
        c = pad_group([mels[i] for i in grp], cin_pad)
        wav = synth_group(c, list(grp))
        for row, i in enumerate(grp):
            local[i] = wav[row, : lengths[i]].detach().to("cpu")
    return local


def gather_results(local: dict, n: int, *, group=None, gather_to: Optional[int] = 0) -> Optional[List[torch.Tensor]]:
    """The ONE collective of a job: every rank's {utterance index: waveform} to rank ``gather_to`` (every rank if None), in job order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local[i] for i in range(n)]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if gather_to is None:
        parts = [None] * world
        dist.all_gather_object(parts, local, group=group)
    else:
        parts = [None] * world if rank == gather_to else None
        dist.gather_object(local, parts, dst=gather_to, group=group)
        if rank != gather_to:
            return None
    merged = {}
    for part in parts:
        merged.update(part)
    return [merged[i] for i in range(n)]


def synthesize_sharded(mels: Sequence[torch.Tensor], synth_group: Callable[[torch.Tensor, List[int]], torch.Tensor],
                       *, hop_size: int, cin_pad: int, group_size: Optional[int] = None, group=None,
                       gather_to: Optional[int] = 0, stats: Optional[dict] = None) -> Optional[List[torch.Tensor]]:
    """Distribute ``mels`` (list of (cin, frames) tensors, identical on every rank) over the ranks.

    ``synth_group(c, idx)`` receives the padded batch ``c`` (B, cin, frames + 2*cin_pad) and the global utterance
    indices, and returns the waveforms (B, T) -- on the GPU box that is
    ``model.incremental_forward(c=c.cuda(), T=frames*hop)[:, 0]``.  Returns the list of trimmed waveforms in the
    original order on rank ``gather_to`` (on every rank if ``gather_to`` is None), None elsewhere.
    ``group_size=None`` (default): from the measured throughput curve, ``auto_group_size`` -- up to 48 utterances per launch; a
    number (``hparams.batch_size`` of the caller) wins.  ``stats``: filled with this rank's groups, true / padded samples."""
    _, _, _, lengths, mine = _local_share(mels, hop_size, group)
    local = synthesize_local_padded(mels, synth_group, mine, lengths, cin_pad=cin_pad, group_size=group_size, stats=stats)
    return gather_results(local, len(mels), group=group, gather_to=gather_to)


# ---- packed slots: continuous batching ---------------------------------------------------------------------------------------------
# A padded batch runs to its longest member (evaluate.py:55-57,215): with utterances of 1-8 s a quarter to a third of the samples a GPU
# makes is padding, whatever the grouping.  The ring kernel does not need that: a ROW of a launch can be a SLOT that runs several
# utterances back to back (wnv_generate_args.seg_start / seg_uid, ABI 4) -- at a boundary the next utterance starts exactly as
# incremental_forward starts one (zero history, zero / index-127 first input) and draws its noise from its own stream, so every
# waveform is what the utterance gives on its own.  The planner below fills THROUGHPUT_GROUP slots longest-first.

def plan_slots(lengths: Sequence[int], n_slots: int) -> List[List[int]]:
    """Utterance indices per slot, in running order: longest-processing-time-first over ``n_slots`` bins (a slot's cost is the SUM of
    its utterances' lengths); no more slots than utterances."""
    n_slots = max(1, min(int(n_slots), len(lengths)))
    return [b for b in lpt_assign(lengths, n_slots) if b]


def plan_launches(lengths: Sequence[int], n_slots: int, steps_cap: int) -> List[List[int]]:
    """Positions into ``lengths`` per launch of ``n_slots`` packed slots, no slot longer than ``steps_cap`` steps (an utterance that is
    longer than the cap on its own still gets its launch: the cap bounds packing, it does not refuse work).  As few launches as the
    bound allows, and BALANCED: the length-sorted utterances are dealt out to the launches in turn, so every launch gets the same mix of
    lengths and the same total (round 5: filling the first launch to the cap left a short second one whose slots ran mostly empty -- 200
    utterances: 2.00 against 2.31 MSamples/s).  A pure function of its arguments: every rank of a job plans the same launches."""
    lengths = [int(x) for x in lengths]
    if not lengths:
        return []
    n_slots, steps_cap = max(1, int(n_slots)), max(1, int(steps_cap))
    order = sorted(range(len(lengths)), key=lambda k: (-lengths[k], k))
    n_launch = max(1, -(-sum(lengths) // (n_slots * steps_cap)))
    while True:
        launches = [m for m in (order[i::n_launch] for i in range(n_launch)) if m]
        # (a slot's sum can exceed the average: check the planned launches against the cap, split further when one does)
        worst = max(max(sum(lengths[m[k]] for k in b) for b in plan_slots([lengths[k] for k in m], n_slots)) for m in launches)
        if worst <= steps_cap or n_launch >= len(order):
            return launches
        n_launch += 1


def packed_group_size(model) -> int:
    """Slots per launch for THIS model: where its throughput curve stops growing (profiles/r04_final_numbers.txt).  128 skip channels
    (egs/mol, egs/gaussian) and the 256-way one-hot models gain up to 48 utterances per GPU; with 512 skip channels (BASELINE
    configs[4]) a stage is busy 3.8 us per utterance and the curve is flat from 32 on (1.37 MSamples/s; below real time per utterance at 64)."""
    k = int(getattr(model, "skip_out_channels", 0) or 0)
    if k <= 0:
        try:
            k = int(model.conv_layers[0].conv1x1_skip.out_channels)
        except Exception:
            k = 128
    return 32 if k > 256 else THROUGHPUT_GROUP


def packed_unsupported_reason(model) -> Optional[str]:
    """Why ``synthesize_packed`` would refuse this model -- decided WITHOUT a launch, identically on every rank (the ranks of a job must
    agree on packed versus padded before anybody enters a collective); None when it is covered."""
    p = next(model.parameters(), None)
    if p is None or not p.is_cuda:
        return "the model is not on a GPU"
    if int(getattr(model, "gin_channels", -1) or -1) > 0 and getattr(model, "embed_speakers", None) is None:
        return "global conditioning without a speaker embedding (external g vectors per utterance are not packed)"
    if int(getattr(model, "cin_channels", -1) or -1) <= 0:
        return "no local conditioning: a packed job is a list of mel spectrograms (an unconditioned model has no utterances to pack)"
    if getattr(model, "upsample_net", None) is None:
        return "local conditioning without an upsampling network"
    try:
        eng = model._get_engine()
        why = eng.kernel_coverage(2)
        if why != "supported":
            return "the pipelined ring kernel does not cover the model: " + why
    except Exception as e:       # (no engine on this device: the launch would say so too)
        return str(e)[:160]
    return None


def synthesize_packed(model, mels: Sequence[torch.Tensor], *, hop_size: int, cin_pad: int, slots: Optional[int] = None,
                      seed: Optional[int] = None, indices: Optional[Sequence[int]] = None, stats: Optional[dict] = None,
                      max_slot_steps: int = 1 << 20, max_launch_bytes: int = 32 << 30, params_out: Optional[list] = None,
                      speaker_ids: Optional[Sequence[int]] = None,
                      sink: Optional[Callable[[int, torch.Tensor], None]] = None, as_index: bool = False) -> List[Optional[torch.Tensor]]:
    """The waveforms (network outputs ``(C, T_i)`` on the model's device, one per mel of ``mels[i] for i in indices``) of a job run as
    PACKED SLOTS on this process's GPU.  ``model``: an ``EngineHost`` WaveNet on the device that the ring kernel covers (otherwise
    NotImplementedError: the caller falls back to padded groups).  ``slots``: rows of a launch (default ``packed_group_size(model)``);
    ``seed``: of the in-kernel noise streams (default: drawn from torch's generator, as ``rng = "philox"`` does).
    ``speaker_ids`` (round 5; one per mel of ``mels``, for models with a speaker embedding -- BASELINE configs[4],
    wavenet.py:262-269): the hoisted bias table gets one row per SPEAKER of the embedding and every slot-step names its row (``seg_gid``).
    ``params_out``: a list that receives, per utterance, the head outputs ``(O, T_i)`` the sampler was handed at every step (parity tests).
    ``sink(i, y)``: called with utterance ``i`` (index into ``mels``) and its ``(C, T_i)`` output as soon as its launch is done, INSTEAD
    of keeping the output (the returned list then holds None): a long job keeps only one launch's buffers on the device.
    ``as_index`` (one-hot models): the outputs are the sampled CLASSES, ``(1, T_i)`` float32, instead of ``(C, T_i)`` one-hot vectors --
    the launch then carries 4 bytes of output per slot-step instead of 4 C (1 KB for a 256-way model: a job of 100 utterances needed two
    launches of half-empty slots under the byte bound), and ``synthesis.postprocess`` takes them as they are.
    One launch is bounded by ``max_slot_steps`` steps per slot AND by ``max_launch_bytes`` of resident per-step buffers -- the slots'
    conditioning (``cin`` floats per step, and as much again for the upsampler's output of the utterance being placed), the output (``C``
    floats per step: 1 KB for a 256-way one-hot model; ``as_index``: the int32 classes and their float copy) and the maps; a longer job runs
    as several launches.  The default byte bound is 32 GiB of the GPU's 288 GB: 48 slots of egs/mol may run 2^20 steps (43 s of audio) each --
    with 12 GiB (round 5) and the honest byte count of round 6 a job of 200 utterances split into two launches and lost 5 % to padding.  An utterance's waveform does not depend on any of this: its conditioning is upsampled on its own
    (what ``incremental_forward`` gives for the utterance alone, its edges replicate-padded as evaluate.py:163-164 does for a batch of
    one), its noise stream is (utterance id, step within the utterance)."""
    why = packed_unsupported_reason(model)
    if why is not None:
        raise NotImplementedError("packed slots: " + why)
    has_spk = getattr(model, "embed_speakers", None) is not None
    if has_spk:
        if speaker_ids is None or len(speaker_ids) != len(mels):
            raise ValueError("packed slots: a model with a speaker embedding needs speaker_ids, one per mel")
        n_spk = int(model.embed_speakers.weight.shape[0])
        bad = [int(g) for g in speaker_ids if not 0 <= int(g) < n_spk]
        if bad:
            raise IndexError(f"speaker id {bad[0]} outside the embedding table of {n_spk} speakers")
    elif speaker_ids is not None:
        raise ValueError("speaker_ids given but the model has no speaker embedding")
    idx = list(range(len(mels))) if indices is None else list(indices)
    if not idx:
        return []
    n_slots = packed_group_size(model) if slots is None else int(slots)
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    eng = model._get_engine()
    cin = int(mels[idx[0]].shape[0])
    as_index = bool(as_index) and not eng.cfg.scalar_input
    c_out = 1 if (eng.cfg.scalar_input or as_index) else int(eng.cfg.out_channels)
    # resident bytes per slot-step of a launch: conditioning row, output (as_index: the int32 classes AND their float32 copy for the
    # post-chain, both alive at once), the two segment maps (+ the speaker map), the head outputs when asked for; the per-utterance
    # upsampler output is one utterance at a time (<= a slot's worth of conditioning: one more cin per step covers it)
    step_bytes = 4 * (2 * cin + (2 if as_index else c_out) + 2 + (1 if has_spk else 0) + (int(eng.cfg.out_channels) if params_out is not None else 0))
    steps_cap = max(hop_size, min(int(max_slot_steps), int(max_launch_bytes) // (n_slots * step_bytes)))
    lengths_all = [int(mels[i].shape[-1]) * hop_size for i in idx]
    launches = plan_launches(lengths_all, n_slots, steps_cap)
    res: List[Optional[torch.Tensor]] = [None] * len(idx)
    par: Optional[list] = None if params_out is None else [None] * len(idx)
    agg = dict(slots=0, slot_steps=0, true_samples=0, padded_samples=0, utterances_per_slot=[], launches=[], step_bytes=step_bytes)
    for members in launches:
        out_m, st_m, par_m = _packed_launch(model, mels, [idx[k] for k in members], hop_size, cin_pad, n_slots, seed,
                                            want_params=params_out is not None, speaker_ids=speaker_ids, as_index=as_index)
        for j, (k, y) in enumerate(zip(members, out_m)):
            if sink is not None:
                sink(idx[k], y)
            else:
                res[k] = y.clone()                                                   # (a copy: the launch's output buffer goes away)
            if par is not None:
                par[k] = par_m[j].clone()
        del out_m, par_m
        agg["slots"] = max(agg["slots"], st_m["slots"])
        agg["slot_steps"] += st_m["slot_steps"]
        agg["true_samples"] += st_m["true_samples"]
        agg["padded_samples"] += st_m["padded_samples"]
        agg["utterances_per_slot"] += st_m["utterances_per_slot"]
        agg["launches"].append((st_m["slots"], st_m["slot_steps"]))
    if stats is not None:
        agg["padding_loss"] = 1.0 - agg["true_samples"] / float(max(agg["padded_samples"], 1))
        stats.update(agg)
    if params_out is not None:
        params_out[:] = par
    return res


def upsample_each(eng, mels: Sequence[torch.Tensor], ids: Sequence[int], cin_pad: int, hop_size: int):
    """Yields (position in ``ids``, (T_i, cin) conditioning on the device) -- every utterance upsampled ON ITS OWN: its edges
    replicate-padded (evaluate.py:163-164 for a batch of one), zeros nowhere.  Utterances of equal length share a launch (their padded
    batch has no zero padding, so rows do not see each other)."""
    by_frames = {}
    for k, i in enumerate(ids):
        by_frames.setdefault(int(mels[i].shape[-1]), []).append(k)
    for f, ks in by_frames.items():
        for a in range(0, len(ks), 32):
            grp = ks[a:a + 32]
            c = pad_group([mels[ids[k]] for k in grp], cin_pad).to(eng.device)
            cu = eng.upsample(c, T_expected=f * hop_size)                            # (len(grp), T, cin) time-major
            for row, k in enumerate(grp):
                yield k, cu[row]


def segment_maps(bins: Sequence[Sequence[int]], lengths: Sequence[int], ids: Sequence[int], speaker_ids: Optional[Sequence[int]], T: int, device):
    """The per-slot-step maps of a packed launch (``wnv_generate_args.seg_start / seg_uid / seg_gid``, each ``(len(bins), T)`` int32 on
    ``device``) and ``where``: position in ``ids`` -> (slot, first step).  ``bins[s]``: positions into ``ids`` / ``lengths`` in running
    order.  At slot-step (s, t) the maps name the utterance that runs there: the step it started at, its id in the job (``ids[k]``: its
    noise stream), its speaker (``speaker_ids[ids[k]]``: its bias row; None without a speaker embedding).  A slot that ends before T
    keeps its last utterance running to T (ignored by the caller).
    8-12 bytes per slot-step next to 4 cin of conditioning: the kernel's roles look a step up without carrying a cursor per slot in
    registers they do not have.  Built ON THE DEVICE from one small segment table (start, id, speaker, run length per segment) -- as host
    arrays they were 8-12 T n bytes of fills and of pageable upload per launch."""
    where = {}
    seg_rows = []                               # (start, uid, gid, run length) per segment, slot after slot
    for s, b in enumerate(bins):
        off = 0
        for j, k in enumerate(b):
            where[k] = (s, off)
            run = lengths[k] if j + 1 < len(b) else T - off
            seg_rows.append((off, ids[k], int(speaker_ids[ids[k]]) if speaker_ids is not None else 0, run))
            off += lengths[k]
    n = len(bins)
    table = torch.tensor(seg_rows, dtype=torch.int32).to(device)
    runs = table[:, 3].to(torch.int64)
    seg_start = torch.repeat_interleave(table[:, 0], runs, output_size=n * T).view(n, T)
    seg_uid = torch.repeat_interleave(table[:, 1], runs, output_size=n * T).view(n, T)
    seg_gid = torch.repeat_interleave(table[:, 2], runs, output_size=n * T).view(n, T) if speaker_ids is not None else None
    return where, seg_start, seg_uid, seg_gid


def _packed_launch(model, mels, ids, hop_size, cin_pad, n_slots, seed, want_params=False, speaker_ids=None, as_index=False):
    """One launch of packed slots over the utterances ``ids`` (indices into ``mels``, which are also their ids in the job).
    Returns views into the launch's output buffers (the caller copies or consumes them before the next launch)."""
    eng = model._get_engine()
    dev = eng.device
    cin = int(mels[ids[0]].shape[0])
    frames = [int(mels[i].shape[-1]) for i in ids]
    lengths = [f * hop_size for f in frames]
    bins = plan_slots(lengths, n_slots)                                               # positions into ids
    n, T = len(bins), max(sum(lengths[k] for k in b) for b in bins)
    c_slot = torch.zeros(n, T, cin, device=dev, dtype=torch.float32)
    where, seg_start, seg_uid, seg_gid = segment_maps(bins, lengths, ids, speaker_ids, T, dev)
    for k, cu in upsample_each(eng, mels, ids, cin_pad, hop_size):
        s, off = where[k]
        c_slot[s, off:off + lengths[k]] = cu
    g_rows = None
    if seg_gid is not None:                     # one bias row per speaker of the embedding table (tiny: n_speakers x L x G floats)
        g_rows = torch.arange(int(model.embed_speakers.weight.shape[0]), dtype=torch.int64, device=dev)
    out, params, index = eng.generate(B=n, T=T, c_up=c_slot, seed=seed, seg_start=seg_start, seg_uid=seg_uid, seg_gid=seg_gid, g_ids=g_rows,
                                      kernel=0, want_params=want_params, want_index=as_index, want_out=not as_index)
    if as_index:                                # the classes as a (n, 1, T) float tensor: what the post-chain takes with C = 1
        out = index.to(torch.float32).unsqueeze(1)
        del index
    st = dict(slots=n, slot_steps=T, true_samples=sum(lengths), padded_samples=n * T, utterances_per_slot=[len(b) for b in bins])
    res, par = [], []
    for k in range(len(ids)):
        s, off = where[k]
        res.append(out[s, :, off:off + lengths[k]])
        if want_params:
            par.append(params[s, :, off:off + lengths[k]])
    return res, st, par
