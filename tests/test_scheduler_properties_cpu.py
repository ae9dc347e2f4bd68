"""Property tests (hypothesis) of the job schedulers in wavenet_vocoder_amd/sharding.py -- pure functions every rank of a job evaluates on
its own and must evaluate identically: lpt_assign (utterances -> ranks), pack_groups (padded groups), plan_slots (packed slots),
plan_launches (launches of a packed job under a step cap), segment_maps (the maps a packed launch hands to the kernel)."""
import pytest
import torch

pytest.importorskip("hypothesis")           # (collection on a box without it must not fail the run: -m gpu imports every module)
from hypothesis import given, settings, strategies as st  # noqa: E402

from wavenet_vocoder_amd.sharding import lpt_assign, pack_groups, plan_launches, plan_slots, segment_maps

lengths_st = st.lists(st.integers(min_value=1, max_value=2000), min_size=1, max_size=60)


@settings(max_examples=60, deadline=None)
@given(lengths_st, st.integers(min_value=1, max_value=9))
def test_lpt_assign_partitions_and_balances(lengths, n):
    bins = lpt_assign(lengths, n)
    assert len(bins) == n and sorted(sum(bins, [])) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in b) for b in bins]
    # the classic bound of longest-processing-time-first: no bin is more than the longest item above the lightest one
    assert max(loads) - min(loads) <= max(lengths)
    assert bins == lpt_assign(list(lengths), n)


@settings(max_examples=60, deadline=None)
@given(lengths_st, st.integers(min_value=1, max_value=64))
def test_pack_groups_keeps_neighbouring_lengths_together(lengths, size):
    groups = pack_groups(range(len(lengths)), lengths, size)
    assert sorted(sum(groups, [])) == list(range(len(lengths))) and all(1 <= len(g) <= size for g in groups)
    flat = [lengths[i] for g in groups for i in g]
    assert flat == sorted(flat, reverse=True)


@settings(max_examples=60, deadline=None)
@given(lengths_st, st.integers(min_value=1, max_value=64))
def test_plan_slots_never_opens_more_slots_than_utterances(lengths, n_slots):
    bins = plan_slots(lengths, n_slots)
    assert 1 <= len(bins) <= min(n_slots, len(lengths)) and all(bins)
    assert sorted(sum(bins, [])) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in b) for b in bins]
    assert max(loads) - min(loads) <= max(lengths)


@settings(max_examples=60, deadline=None)
@given(lengths_st, st.integers(min_value=1, max_value=48), st.integers(min_value=1, max_value=6000))
def test_plan_launches_respects_the_cap_wherever_packing_decides(lengths, n_slots, cap):
    plan = plan_launches(lengths, n_slots, cap)
    assert sorted(sum(plan, [])) == list(range(len(lengths))) and all(plan)
    assert plan == plan_launches(tuple(lengths), n_slots, cap)
    for m in plan:
        for b in plan_slots([lengths[k] for k in m], n_slots):
            load = sum(lengths[m[k]] for k in b)
            # over the cap only where a single utterance is (the cap bounds packing, it does not refuse work) -- or where the plan has
            # already given every utterance a launch of its own
            assert load <= cap or len(b) == 1 or len(plan) == len(lengths)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=1, max_value=40), min_size=1, max_size=25), st.integers(min_value=1, max_value=8), st.booleans())
def test_segment_maps_name_the_running_utterance(lengths, n_slots, with_speakers):
    ids = [7 + 2 * k for k in range(len(lengths))]
    speakers = [k % 5 for k in range(max(ids) + 1)] if with_speakers else None
    bins = plan_slots(lengths, n_slots)
    T = max(sum(lengths[k] for k in b) for b in bins)
    where, start, uid, gid = segment_maps(bins, lengths, ids, speakers, T, "cpu")
    assert start.shape == uid.shape == (len(bins), T) and (gid is None) == (not with_speakers)
    for k, (s, off) in where.items():
        assert torch.all(start[s, off:off + lengths[k]] == off) and torch.all(uid[s, off:off + lengths[k]] == ids[k])
        if with_speakers:
            assert torch.all(gid[s, off:off + lengths[k]] == speakers[ids[k]])
    t = torch.arange(T).unsqueeze(0)
    assert torch.all(start <= t) and int((start == t).sum()) == len(lengths)


def test_the_bench_jobs_fit_one_packed_launch_under_the_default_byte_bound():
    """Round 6: the honest byte count of a packed launch (conditioning + the upsampler's output of the utterance being placed + output + maps:
    652 bytes per slot-step for egs/mol) under round 5's 12-GiB bound split bench.py's 200-utterance job into two launches -- 7.6 % padding
    instead of 2.5 %, 5 % slower.  The default bound is 32 GiB of the GPU's 288 GB: that job, and a 400-utterance one, are ONE launch of 48
    slots; what bounds a launch then is max_slot_steps (2^20 steps per slot)."""
    import inspect
    from wavenet_vocoder_amd import sharding
    sig = inspect.signature(sharding.synthesize_packed)
    max_bytes, max_steps = sig.parameters["max_launch_bytes"].default, sig.parameters["max_slot_steps"].default
    assert max_bytes == 32 << 30 and max_steps == 1 << 20
    step_bytes = 4 * (2 * 80 + 1 + 2)                                     # egs/mol: scalar output, no speaker map (sharding.synthesize_packed)
    for n_utts, want_launches in ((200, 1), (400, 1)):
        gen = torch.Generator().manual_seed(2024)                          # bench.py job_inputs: 1.0 .. 8.0 s at 24 kHz, hop 256
        lengths = [f * 256 for f in torch.randint(94, 751, (n_utts,), generator=gen).tolist()]
        cap = max(256, min(max_steps, max_bytes // (48 * step_bytes)))
        launches = sharding.plan_launches(lengths, 48, cap)
        assert len(launches) == want_launches
        worst = max(sum(lengths[m[k]] for k in b) for m in launches for b in sharding.plan_slots([lengths[k] for k in m], 48))
        assert worst <= cap and 48 * worst * step_bytes <= max_bytes
