"""N > 1 path on CPU: two processes, gloo backend, utterance sharding + gather (no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavenet_vocoder_amd.sharding import (THROUGHPUT_GROUP, auto_group_size, broadcast_weights, lpt_assign, pack_groups,
                                          pad_group, padding_loss, plan_launches, plan_slots, segment_maps, synthesize_sharded)

HOP, PAD = 4, 2


def fake_synth(c, idx):
    """Stand-in for the GPU engine: a deterministic function of each utterance's own mel only."""
    B, cin, fr = c.shape
    frames = fr - 2 * PAD
    body = c[:, :, PAD:PAD + frames]
    return body.mean(1).repeat_interleave(HOP, dim=1) + torch.tensor(idx, dtype=torch.float32).view(-1, 1)


def make_mels(n=13, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(5, int(torch.randint(3, 40, (1,), generator=g)), generator=g) for _ in range(n)]


def expected(mels):
    return [m.mean(0).repeat_interleave(HOP) + i for i, m in enumerate(mels)]


def test_lpt_and_packing():
    lengths = [10, 200, 30, 40, 50, 60, 70, 5, 90]
    bins = lpt_assign(lengths, 3)
    assert sorted(sum(bins, [])) == list(range(9))
    loads = [sum(lengths[i] for i in b) for b in bins]
    assert max(loads) <= 200 + 5            # the long utterance dominates; the rest balances
    groups = pack_groups(range(9), lengths, 4)
    assert [len(g) for g in groups] == [4, 4, 1]
    assert lengths[groups[0][0]] == 200 and all(
        lengths[a] >= lengths[b] for g in groups for a, b in zip(g, g[1:]))
    c = pad_group([torch.ones(2, 3), 2 * torch.ones(2, 5)], cin_pad=2)
    assert c.shape == (2, 2, 9)
    assert c[0, 0].tolist() == [1, 1, 1, 1, 1, 0, 0, 0, 0] and c[1, 0].tolist() == [2] * 9


def test_group_size_from_the_measured_curve():
    """No group size given: everything in one launch up to the throughput plateau (48 utterances per GPU since round 4; the reference
    recipe's own inference batch is 32, egs/mol/run.sh:31), groups of 48 beyond it -- 40 pending utterances run as one launch, 56 as
    48 + 8; a number wins."""
    assert THROUGHPUT_GROUP == 48
    assert [auto_group_size(n) for n in (0, 1, 5, 8, 9, 32, 40, 48, 49, 1000)] == [1, 1, 5, 8, 9, 32, 40, 48, 48, 48]
    lengths = [1000 + 37 * ((7 * i) % 40) for i in range(56)]
    assert [len(g) for g in pack_groups(range(40), lengths)] == [40]
    groups = pack_groups(range(56), lengths)
    assert [len(g) for g in groups] == [48, 8]
    assert sorted(sum(groups, [])) == list(range(56))
    assert min(lengths[i] for i in groups[0]) >= max(lengths[i] for i in groups[1])       # neighbouring lengths, descending
    assert [len(g) for g in pack_groups(range(40), lengths, 8)] == [8] * 5                  # hparams.batch_size wins
    assert [len(g) for g in pack_groups(range(6), lengths)] == [6]
    assert [len(g) for g in pack_groups(range(100), lengths * 2)] == [48, 48, 4]
    with pytest.raises(ValueError):
        pack_groups(range(4), lengths, 0)
    # padding loss: a group runs to its longest member
    assert padding_loss([[0, 1]], [100, 50]) == pytest.approx(0.25)
    assert padding_loss([[0], [1]], [100, 50]) == 0.0
    assert padding_loss(groups, lengths) < padding_loss([list(range(56))], lengths)         # two sorted groups pad less than one
    # the stats the job mode of bench.py reports
    st = {}
    mels = make_mels(56, seed=3)
    got = synthesize_sharded(mels, fake_synth, hop_size=HOP, cin_pad=PAD, stats=st)
    for a, b in zip(got, expected(mels)):
        assert torch.allclose(a, b, atol=1e-6)
    assert [len(g) for g in st["groups"]] == [48, 8]
    assert st["true_samples"] == sum(m.shape[-1] * HOP for m in mels) and st["padded_samples"] >= st["true_samples"]
    assert st["padding_loss"] == pytest.approx(1 - st["true_samples"] / st["padded_samples"])


def test_slot_plan_of_packed_jobs():
    """Continuous batching: a slot's cost is the SUM of its utterances (not the longest): longest-first over the slots leaves a few
    per cent of idle tail where padded groups of the same job lose a quarter."""
    g = torch.Generator().manual_seed(5)
    lengths = [int(x) * 256 for x in torch.randint(94, 751, (100,), generator=g)]
    bins = plan_slots(lengths, 48)
    assert len(bins) == 48 and sorted(sum(bins, [])) == list(range(100))
    loads = [sum(lengths[i] for i in b) for b in bins]
    packed_loss = 1 - sum(lengths) / (len(bins) * max(loads))
    padded_loss = padding_loss(pack_groups(range(100), lengths), lengths)
    assert packed_loss < 0.10 < 0.2 < padded_loss, (packed_loss, padded_loss)
    assert [len(b) for b in plan_slots([5, 3, 9], 48)] == [1, 1, 1]                  # never more slots than utterances
    assert plan_slots([7], 4) == [[0]]


def test_launch_plan_of_packed_jobs():
    """sharding.plan_launches: every utterance in exactly one launch; no slot beyond the cap unless one utterance is; as few launches as the
    total allows (+ the split a tight cap forces); the launches balanced to within the longest utterance; the same plan from the same
    arguments (every rank of a job plans alone)."""
    g = torch.Generator().manual_seed(11)
    lengths = [int(x) * 256 for x in torch.randint(94, 751, (200,), generator=g)]
    for n_slots, cap in ((48, 1 << 20), (48, 400_000), (48, 200_000), (32, 250_000), (3, 6000 * 40)):
        plan = plan_launches(lengths, n_slots, cap)
        assert sorted(sum(plan, [])) == list(range(200))
        assert plan == plan_launches(list(lengths), n_slots, cap)
        lower = -(-sum(lengths) // (n_slots * cap))
        # (few slots under a cap of 1.25 utterances fragment: a fifth more launches than the total alone would need)
        assert lower <= len(plan) <= lower + 1 + lower // 5, (n_slots, cap, len(plan), lower)
        totals = [sum(lengths[k] for k in m) for m in plan]
        assert max(totals) - min(totals) <= max(lengths) * 2, totals
        for m in plan:
            slots = plan_slots([lengths[k] for k in m], n_slots)
            assert max(sum(lengths[m[k]] for k in b) for b in slots) <= cap
    # the cap bounds packing, it does not refuse work: an utterance longer than the cap runs alone in its slot
    plan = plan_launches([1000, 10, 10, 10], 2, 100)
    assert sorted(sum(plan, [])) == [0, 1, 2, 3]
    for m in plan:
        for b in plan_slots([[1000, 10, 10, 10][k] for k in m], 2):
            load = sum([1000, 10, 10, 10][m[k]] for k in b)
            assert load <= 100 or len(b) == 1
    assert plan_launches([], 48, 100) == []
    assert plan_launches([7], 48, 1) == [[0]]


def test_segment_maps_of_a_packed_launch():
    """sharding.segment_maps (the (slots, T) int32 maps the ring kernel's roles read, built with torch ops from one small table) against
    the literal per-step loop: at slot-step (s, t) the start step, the job id and the speaker of the utterance that runs there; a slot
    that ends early keeps its last utterance to T."""
    g = torch.Generator().manual_seed(3)
    lengths = [int(x) * 8 for x in torch.randint(1, 12, (23,), generator=g)]
    ids = [100 + 3 * k for k in range(23)]                              # ids in the job (positions into the job's mel list)
    speakers = {i: int(torch.randint(0, 7, (1,), generator=g)) for i in ids}
    spk_list = [speakers.get(i, -1) for i in range(max(ids) + 1)]
    for n_slots in (1, 4, 23, 48):
        bins = plan_slots(lengths, n_slots)
        T = max(sum(lengths[k] for k in b) for b in bins)
        for with_spk in (False, True):
            where, start, uid, gid = segment_maps(bins, lengths, ids, spk_list if with_spk else None, T, "cpu")
            assert start.dtype == uid.dtype == torch.int32 and start.shape == uid.shape == (len(bins), T) and start.is_contiguous()
            assert (gid is None) == (not with_spk)
            assert sorted(where) == list(range(23))
            for s, b in enumerate(bins):
                off = 0
                for j, k in enumerate(b):
                    assert where[k] == (s, off)
                    end = off + lengths[k] if j + 1 < len(b) else T
                    assert torch.all(start[s, off:end] == off) and torch.all(uid[s, off:end] == ids[k])
                    if with_spk:
                        assert torch.all(gid[s, off:end] == speakers[ids[k]])
                    off += lengths[k]
            # the kernel's reading of the maps: the step within the utterance is t - start, a new utterance begins where start == t
            t = torch.arange(T).unsqueeze(0)
            assert torch.all(start <= t) and int((start == t).sum()) == 23


def test_single_process_matches():
    mels = make_mels()
    got = synthesize_sharded(mels, fake_synth, hop_size=HOP, cin_pad=PAD, group_size=3)
    for a, b in zip(got, expected(mels)):
        assert torch.allclose(a, b, atol=1e-6)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lin = torch.nn.Linear(7, 3)
        if rank != 0:
            with torch.no_grad():
                for p in lin.parameters():
                    p.zero_()
        ref = [p.detach().clone() for p in lin.parameters()] if rank == 0 else None
        broadcast_weights(lin, src=0)
        mels = make_mels()
        got = synthesize_sharded(mels, fake_synth, hop_size=HOP, cin_pad=PAD, group_size=3, gather_to=0)
        every = synthesize_sharded(mels, fake_synth, hop_size=HOP, cin_pad=PAD, group_size=3, gather_to=None)
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(every, expected(mels)))
        if rank == 0:
            ok = ok and all(torch.allclose(a, b, atol=1e-6) for a, b in zip(got, expected(mels)))
            ok = ok and all(torch.equal(a, b.detach()) for a, b in zip(ref, lin.parameters()))
        else:
            ok = ok and got is None and float(sum(p.abs().sum() for p in lin.parameters())) > 0
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}
