#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

metric   : audio kSamples/s (egs/mol: 24-layer 10-mixture MoL WaveNet, 80-mel conditioned, batch = 8 utterances
           per GPU); `value` is the whole-job aggregate over all N GPUs.
step     : ONE pass of the hot path over one batch: wnv_upsample (mel -> sample rate) + wnv_generate (the whole
           T-sample autoregressive loop incl. sampling, in-kernel Philox) for B = 8 utterances of T = 24064
           samples (94 frames x hop 256 = 1.003 s at 24 kHz).  Inputs (mel, weights) are resident in HBM before
           the timed region.
scaling  : weak -- every rank synthesises its own 8 utterances; no collective on the data path (utterances
           are independent); only the barrier/max-reduce around the timed region uses RCCL.
kernel   : auto = the pipelined ring kernel (csrc/wnv_ring.hip) for this configuration.
launch   : `python bench.py --gpus N` starts its own N ranks (one process per GPU, LOCAL_RANK = GPU index, rendezvous on
           127.0.0.1) when WORLD_SIZE is not set; under `python -m torch.distributed.run ... bench.py --gpus N` it uses the
           ranks it is given.  `--dry-run` exercises the same launch / barrier / max-reduce / single-JSON-line plumbing on
           CPU (gloo, no engine) -- tests/test_bench_launch_cpu.py.
roofline : the dominant kernel is the sample-loop kernel.  achieved = algorithmic bytes per launch
           (wnv_bytes_per_step(B) x T, SURVEY.md 8d: every weight once per step per utterance group + ring taps
           + conditioning row + sample) / the kernel's duration measured with HIP events on its own stream.
           `roofline` prices it against the LDS read peak -- the roofline BASELINE.json and SURVEY.md 8d ask for: the
           weights are on chip (VGPRs + LDS), nothing streams from HBM (MI355X_MICROARCH.md: 256 CU x 256 B/clk x
           2.4 GHz = 157 TB/s); `roofline_hbm` prices the same bytes against 8 TB/s for reference; `roofline_latency` is
           the bound this kernel actually has: the serial chain of L layers, floor = L x (one CU -> CU hop as measured
           by scripts/ubench_hop2.hip + the on-chain 256x128 mat-vec at the CU's fp32 FMA peak) + the head.
           `traffic` (HBM/fabric bytes per launch from rocprofv3 PMC passes) cannot be collected from inside this
           process: it is quoted from profiles/traffic_latest.json together with its source file and commit.
cpu_baseline : kind "reference" -- the UNMODIFIED reference package (oracle/_ref: byte-compiled from /root/reference by
           oracle/build_ref.py, shipped with the tree like the built .so) timed on rank 0's host cores in this same run:
           WaveNet.incremental_forward after make_generation_fast_(), no_grad, the SAME weights / mel / batch 8, a bounded sample
           (T_cpu steps or 12 s per thread setting through the reference's own tqdm hook), 1 / 4 (the reference's own
           setting, synthesis.py:37) / 16 threads, each pinned to one NUMA node.  Only when oracle/_ref is absent:
           kind "port" (oracle/wavenet_oracle.py with the ratio measured in the authoring container).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOAD = "cfg2_mol"
B_PER_GPU = 8
T_SAMPLES = 94 * 256          # 24064 samples = 1.003 s @ 24 kHz
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 256 * 256 * 2.4  # CUs x B/clk/CU (ds_read_b64/b128) x GHz = 157286 GB/s


def numa_local_cpus(want):
    """Up to `want` CPUs of the NUMA node this thread runs on, nearest ids first (the block of 8 that holds the current CPU -- one
    CCD / last-level cache on the GPU box's host -- then its neighbours).  None when sysfs or the affinity API is unavailable."""
    try:
        import ctypes
        import glob
        cur = ctypes.CDLL(None).sched_getcpu()
        allowed = os.sched_getaffinity(0)
        for path in glob.glob("/sys/devices/system/node/node*/cpulist"):
            cpus = []
            for part in open(path).read().strip().split(","):
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
            if cur in cpus:
                cpus = sorted(set(cpus) & allowed, key=lambda x: (abs(x // 8 - cur // 8), x))
                return cpus[:max(1, want)] if cpus else None
    except Exception:
        pass
    return None


def _thread_settings():
    ncores = os.cpu_count() or 1
    return ncores, sorted({1, 4, min(ncores, 16)})       # 4 = the reference's own setting (synthesis.py:37)


def _pinned_runs(run_one):
    """run_one(threads) -> (kSamples/s, steps) for every thread setting, each on CPUs of ONE NUMA node, nearest first (round 3 measured
    4 threads SLOWER than 1 on the GPU box's two-socket, 256-CPU host: the intra-op workers were scheduled anywhere and the 600-KB
    layer weights bounced between sockets)."""
    ncores, settings = _thread_settings()
    affinity0 = None
    try:
        affinity0 = os.sched_getaffinity(0)
    except Exception:
        pass
    results, steps, pinned = {}, {}, {}
    for threads in settings:
        local = numa_local_cpus(threads) if affinity0 is not None else None
        if local:
            try:
                os.sched_setaffinity(0, local)
                pinned[threads] = len(local)
            except Exception:
                pass
        torch.set_num_threads(threads)
        results[threads], steps[threads] = run_one(threads)
        if affinity0 is not None:
            try:
                os.sched_setaffinity(0, affinity0)
            except Exception:
                pass
    return ncores, results, steps, pinned


def cpu_baseline_reference(model_cpu, kw, c, gids, T_cpu, budget_s=12.0, B=8):
    """kind = "reference": the UNMODIFIED reference package (oracle/_ref, byte-compiled from /root/reference by oracle/build_ref.py;
    it travels with the tree like the built .so) timed on this box's host cores: `WaveNet.incremental_forward` (wavenet.py:215-343)
    after `make_generation_fast_()` (wavenet.py:355-361), under `torch.no_grad()`, same weights / mel / speaker ids / batch as the
    GPU run, upsampling included, torch's own generator for the noise.  Bounded through the reference's own `tqdm` hook
    (wavenet.py:217,296): the iterator it wraps around `range(T)` stops handing out steps after `budget_s` seconds -- the per-step
    cost is constant once the history buffers exist (conv.py:34-36), so a truncated run measures the same rate."""
    from oracle import reference as R
    rm = R.build_model(kw, model_cpu.state_dict(), fast=True)
    frames = T_cpu // 256
    c_cpu = None if c is None else c[:B, :, : frames + 2 * kw["cin_pad"]].contiguous()
    g_cpu = None if gids is None else gids[:B]
    # the reference learns the batch size from test_inputs / c only (wavenet.py:253,273): one forced first step -- the default
    # first input itself (zeros / one-hot 127, wavenet.py:281-289) -- tells it B for unconditioned and speaker-conditioned models
    if kw.get("scalar_input", False):
        first = torch.zeros(B, 1, 1)
    else:
        first = torch.zeros(B, kw["out_channels"], 1)
        first[:, 127] = 1.0
    done = {}

    def bounded(rng):
        t0 = time.perf_counter()
        n = 0
        for t in rng:
            if n >= 16 and time.perf_counter() - t0 > budget_s:
                break
            yield t
            n += 1
        done["steps"], done["seconds"] = n, time.perf_counter() - t0

    def run_one(threads):
        import warnings
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rm.incremental_forward(c=None if c_cpu is None else c_cpu[:, :, : 1 + 2 * kw["cin_pad"]], g=g_cpu, T=256 if c_cpu is not None else 32,
                                   test_inputs=first, tqdm=lambda r: (t for t in r if t < 32), log_scale_min=-16.0)      # warm-up
            torch.manual_seed(2)
            rm.incremental_forward(c=c_cpu, g=g_cpu, T=T_cpu, test_inputs=first, tqdm=bounded, softmax=True, quantize=True,
                                   log_scale_min=-16.0)
        return B * done["steps"] / done["seconds"] / 1e3, done["steps"]

    ncores, results, steps, pinned = _pinned_runs(run_one)
    best = max(results, key=results.get)
    return {"value": round(results[best], 4), "unit": "kSamples/s", "cores": best, "kind": "reference",
            "sample": f"the unmodified reference WaveNet.incremental_forward (oracle/_ref: r9y9/wavenet_vocoder byte-compiled by "
                      f"oracle/build_ref.py) after make_generation_fast_(), no_grad, same weights/mel, B={B}, up to T={T_cpu} steps or "
                      f"{budget_s:.0f} s per thread setting through its tqdm hook (steps done: {steps}); host has {ncores} cores, each thread "
                      f"setting pinned to CPUs of one NUMA node ({pinned}); 4 threads is the reference's own setting (synthesis.py:37)",
            "all_threads_kSamples_s": {str(k): round(v, 4) for k, v in results.items()}}


def cpu_baseline(model_cpu, kw, c, T_cpu, budget_s=12.0, B=8, gids=None):
    """The CPU leg of the line.  kind "reference" (oracle/_ref present: the real reference package, see cpu_baseline_reference) or,
    only when that build product is absent, kind "port": the oracle (oracle/wavenet_oracle.py -- checker used as the measured CPU
    path, the one place that is allowed) with the ratio to the real reference measured in the authoring container."""
    from oracle import reference as R
    if R.available():
        return cpu_baseline_reference(model_cpu, kw, c, gids, T_cpu, budget_s, B)
    from oracle.wavenet_oracle import Oracle
    from tests._golden import oracle_config
    from wavenet_vocoder_amd.noise import make_noise_tape
    o = Oracle(oracle_config(kw), model_cpu.state_dict())
    frames = T_cpu // 256
    c_cpu = None if c is None else c[:B, :, : frames + 2 * kw["cin_pad"]].contiguous()
    Bo = B if c is not None else 1                       # (the oracle takes the batch size from c; unconditioned: one utterance)
    tape = make_noise_tape(T_cpu, Bo, scalar_input=kw.get("scalar_input", False),
                           output_distribution=kw.get("output_distribution", "Logistic"), out_channels=kw["out_channels"],
                           generator=torch.Generator().manual_seed(2))
    c_up = None if c_cpu is None else o.upsample(c_cpu).contiguous()   # upsampled once; the timed loop is the sample loop
    saved = o.cfg.upsample_conditional_features
    o.cfg.upsample_conditional_features = False

    def run_one(threads):
        with torch.no_grad():
            o.incremental_forward(c=None if c_up is None else c_up[:, :, :32], T=32, noise=tape)     # warm-up
            o.incremental_forward(c=c_up, T=T_cpu, noise=tape, max_seconds=budget_s)
        return Bo * o.last_steps / o.last_seconds / 1e3, o.last_steps

    ncores, results, steps, pinned = _pinned_runs(run_one)
    o.cfg.upsample_conditional_features = saved
    best = max(results, key=results.get)
    ratio, est = None, None
    try:
        rj = json.load(open(os.path.join(ROOT, "profiles", "cpu_ref_ratio.json")))["by_threads"]
        ratio = {k: round(v["oracle_over_reference"], 3) for k, v in rj.items()}
        r_best = ratio.get(str(best), max(ratio.values()))
        est = round(results[best] / r_best, 4)
    except Exception:
        pass
    return {"value": round(results[best], 4), "unit": "kSamples/s", "cores": best, "kind": "port", "batch": Bo,
            "sample": f"oracle/_ref ABSENT (run oracle/build_ref.py where /root/reference exists) -> oracle/wavenet_oracle.py (torch-CPU "
                      f"restatement of the reference op sequence incl. its per-step queue shift), same weights/mel, B={Bo}, up to T={T_cpu} "
                      f"steps or {budget_s:.0f} s per thread setting (steps done: {steps}); host has {ncores} cores, pinned ({pinned}); the oracle "
                      f"runs {ratio} x the real reference's speed by thread count (profiles/r02_cpu_reference_vs_oracle.txt), so the "
                      f"reference itself would measure ~{est} kSamples/s here",
            "oracle_over_reference_speed": ratio, "reference_estimate_kSamples_s": est,
            "all_threads_kSamples_s": {str(k): round(v, 4) for k, v in results.items()}}


def describe(kw):
    """One-line shape of a tests/_configs.py entry, e.g. 'L24/S4 R128/G256/K128 O30 10-mix MoL, 80-mel + ConvInUpsample x256'."""
    O = kw["out_channels"]
    if not kw.get("scalar_input", False):
        head = f"O{O} softmax (one-hot input)"
    elif kw.get("output_distribution", "Logistic") == "Logistic":
        head = f"O{O} {O // 3}-mix MoL"
    else:
        head = f"O{O} Gaussian"
    s = (f"L{kw['layers']}/S{kw['stacks']} R{kw['residual_channels']}/G{kw['gate_channels']}/K{kw['skip_out_channels']} {head}")
    if kw.get("cin_channels", -1) > 0:
        s += f", {kw['cin_channels']}-mel + ConvInUpsample x256"
    if kw.get("gin_channels", -1) > 0:
        s += f", speaker embedding gin={kw['gin_channels']}"
    return s


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU), wait for them,
    return the worst exit code.  Rank 0's stdout is ours, so exactly one JSON line comes out."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting in a rendezvous or a barrier for ever: stop them (by PID)
    codes = [None] * n
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.05)
    return max(abs(c) for c in codes)


def dry_run(args, world, rank):
    """The distributed skeleton of main() without the engine: rendezvous (gloo), warm-up, barrier, K timed no-op steps,
    barrier, MAX-reduce of the elapsed time, ONE JSON line from rank 0 with the same keys."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, T = args.batch, args.T
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))                   # ranks finish at different times: the reduce must take the MAX
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ranks = rank_report(dist, torch.device("cpu"), 2.0 * (1 + rank), 2)          # (the same gather as the real run; "kernel" 2 = ring)
    job = None
    if args.job > 0:
        # JOB dry run: the scheduler's decisions for this job on `world` ranks (what the multi-GPU BASELINE jobs -- configs[3]: 64
        # utterances, configs[4]: 128 -- would launch), gathered exactly as job_mode gathers its per-rank figures; strong scaling
        from tests._configs import CONFIGS
        kw = CONFIGS[args.workload]
        slots = args.job_group if args.job_group > 0 else (32 if kw["skip_out_channels"] > 256 else 48)       # = sharding.packed_group_size
        mine = list(job_plan(args, kw, world, rank, slots))
        if dist is not None:
            rows = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(rows, torch.tensor(mine, dtype=torch.float64))
            per_rank = [[int(x) for x in r.tolist()] for r in rows]
        else:
            per_rank = [mine]
        true_total, padded_total = sum(r[0] for r in per_rank), sum(r[1] for r in per_rank)
        job = {"utterances": args.job, "true_samples": true_total, "padded_samples": padded_total,
               "padding_loss": round(1.0 - true_total / max(padded_total, 1), 4), "per_rank_true_padded_launches": per_rank,
               "load_imbalance": round(max(r[1] for r in per_rank) / (padded_total / world), 4),
               "scheduler": ("packed slots x%d" % slots) if args.packed else "padded groups"}
    if rank == 0:
        line = {"metric": "audio kSamples/sec (24 kHz MoL egs/mol, batch=8 per GPU), whole job", "value": 0.0,
                "unit": "kSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
                "scaling": "strong" if job else "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)", "dry_run": True,
                "config": {"workload": f"{args.workload} dry run, " + (f"JOB of {args.job} utterances" if job else f"B={B} x T={T}"),
                           "parallelism": f"utterance-sharded x{world}"},
                "ranks": ranks}
        if job:
            line["job"] = job
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def measured_floor_us(kw, n_head_parts=1):
    """The serial-chain floor from MEASURED constants only (round 5; profiles/r05_latency_constants.json): per stage on the chain the
    best same-XCD hop of the own microbenchmarks + the best isolated chain phase of scripts/ubench_phase.hip (barrier, LDS read, mat-vec,
    reduce, gate, store: what one CU needs for one layer when nothing else is in its way); the head as in latency_floor_us."""
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "r05_latency_constants.json")))
        hop, phase = float(cj["hop_us"]), float(cj["phase_us"])
    except Exception:
        hop, phase = 0.276, 0.298
    fma_per_us = 128 * 2400.0
    K, O = kw["skip_out_channels"], kw["out_channels"]
    stages = kw["layers"] - (1 if kw.get("scalar_input", False) and K <= 128 and kw["layers"] >= 2 else 0)
    head = 2 * hop + (K * K / max(n_head_parts, 1) + O * K) / fma_per_us + (hop if n_head_parts > 1 else 0.0)
    return stages * (hop + phase) + head, hop, phase


def latency_floor_us(kw, n_head_parts=1, hop=0.276):
    """Serial-chain floor of the one-layer-per-CU design, from the committed microbenchmarks: per stage on the chain one same-XCD
    CU -> CU hop + the on-chain G x G/2 mat-vec at the CU's fp32 FMA peak (128 FMA/clk at 2.4 GHz: 256 x 128 MACs = 0.107 us); head:
    the skip hop and the hop to the first stage + the K x K and O x K mat-vecs at the same FMA peak (+ one more hop when the head is
    split over several workgroups).  `hop`: 0.276 us = the BEST same-XCD ping-pong the microbenchmarks reach in the ring's own
    configuration (profiles/ubench/hop234_same_box.txt, ubench_hop4: polls in reserved registers); 0.444 us =
    ubench_hop2, what round 2 quoted.  Scalar-input models with 128 skip channels have one stage less on the chain: the head
    evaluates layer 0 (two FMAs per channel)."""
    fma_per_us = 128 * 2400.0
    G, K, O = kw["gate_channels"], kw["skip_out_channels"], kw["out_channels"]
    stages = kw["layers"] - (1 if kw.get("scalar_input", False) and K <= 128 and kw["layers"] >= 2 else 0)
    layer = hop + (G * (G // 2)) / fma_per_us
    head = 2 * hop + (K * K / max(n_head_parts, 1) + O * K) / fma_per_us + (hop if n_head_parts > 1 else 0.0)
    return stages * layer + head


def dist_info(dist, world, args, dev=None):
    """Which collective library carried the barriers / reduces of this line (None: a single process without --force-dist).
    ``ranks_seen``: an all_reduce(SUM) of one 1 per rank over that library -- the line ASSERTS it equals --gpus, so a launch whose ranks
    did not all join the RCCL communicator cannot post a number (VERDICT r05 next #6c)."""
    if dist is None:
        return None
    info = {"backend": str(dist.get_backend()), "world_size": world, "forced_single_rank": bool(getattr(args, "force_dist", False) and world == 1),
            "collectives": "barrier x2, all_reduce(MAX), all_reduce(SUM: ranks_seen), all_gather (rank report)"
                           + (", gather_object (waveforms)" if getattr(args, "job", 0) > 0 else "")}
    if dev is not None:
        one = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        info["ranks_seen"] = int(round(float(one.item())))
        assert info["ranks_seen"] == world == dist.get_world_size(), (info["ranks_seen"], world)
    return info


def strong_job(name, n_utts, dev, dist, world, rank):
    """One STRONG-SCALED job of BASELINE.json -- configs[3]: 64 utterances of the 30-layer Gaussian model, configs[4]: 128 of the
    speaker-conditioned K = 512 model -- the same job whatever the number of ranks: seeded lengths 1-8 s, longest-first assignment to
    the ranks (sharding.lpt_assign), packed slots on every rank, ONE gather of the waveforms to rank 0 (sharding.gather_results).
    With a process group the weights are REPLICATED FROM RANK 0 by sharding.broadcast_weights (one RCCL broadcast of the flat parameter
    buffer) -- every other rank first perturbs its own copy, so the check that all ranks ended with the same waveform statistics is a
    check of the broadcast.  Every rank runs this (it holds collectives); returns the report on rank 0, None elsewhere.
    Reported per rank: true samples, slot-steps incl. padding, seconds of synthesis; the gather's seconds; the imbalance of both."""
    from tests._configs import CONFIGS, build
    from wavenet_vocoder_amd import sharding
    kw = CONFIGS[name]
    hop, pad = 256, int(kw.get("cin_pad", 0))
    gen = torch.Generator().manual_seed(2024)
    frames = torch.randint(94, 751, (n_utts,), generator=gen).tolist()
    mels = [torch.randn(kw["cin_channels"], f, generator=gen) for f in frames]
    spk = torch.randint(0, kw["n_speakers"], (n_utts,), generator=gen).tolist() if kw.get("gin_channels", -1) > 0 else None
    lengths = [f * hop for f in frames]
    model = build(name, seed=0).to(dev)
    t_bcast = None
    if dist is not None:
        if rank != 0:                                       # (what the broadcast has to undo)
            with torch.no_grad():
                for prm in model.parameters():
                    prm.add_(0.01)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sharding.broadcast_weights(model, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t1
    model.rng = "philox"
    mine = sharding.lpt_assign(lengths, world)[rank]
    warm = sorted(mine, key=lambda i: lengths[i])[:2]       # engine, scratch and mailboxes exist before the clock starts
    if warm:
        sharding.synthesize_packed(model, mels, hop_size=hop, cin_pad=pad, indices=warm, seed=1, speaker_ids=spk)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = {}
    outs = sharding.synthesize_packed(model, mels, hop_size=hop, cin_pad=pad, indices=mine, stats=st, seed=4321, speaker_ids=spk) if mine else []
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    local = {i: o[0].detach().to("cpu") for i, o in zip(mine, outs)}
    t1 = time.perf_counter()
    wavs = sharding.gather_results(local, n_utts, gather_to=0) if dist is not None else [local[i] for i in range(n_utts)]
    t_gather = time.perf_counter() - t1
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mine_row = [float(sum(lengths[i] for i in mine)), float(st.get("padded_samples", 0)), t_local, t_gather, elapsed, float(len(mine))]
    if dist is not None:
        rows = [torch.zeros(len(mine_row), dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(mine_row, dtype=torch.float64, device=dev))
        rows = [r.tolist() for r in rows]
    else:
        rows = [mine_row]
    if rank != 0:
        return None
    assert wavs is not None and len(wavs) == n_utts and all(w.numel() == n for w, n in zip(wavs, lengths))
    assert all(torch.isfinite(w).all() and float(w.std()) > 1e-3 for w in wavs), "dead or non-finite waveform in the job"
    true_total = sum(lengths)
    wall = max(r[4] for r in rows)
    t_loc = [r[2] for r in rows]
    return {"workload": f"{name}: {describe(kw)}", "utterances": n_utts, "audio_s_24k": round(true_total / 24000.0, 1),
            "scheduler": f"lpt_assign over {world} rank(s) + packed slots ({sharding.packed_group_size(model)} per GPU)", "scaling": "strong",
            "kSamples_per_s": round(true_total / wall / 1e3, 1), "wall_s": round(wall, 3),
            "x_real_time_24k_whole_job": round(true_total / 24000.0 / wall, 2),
            "per_rank": {"utterances": [int(r[5]) for r in rows], "true_samples": [int(r[0]) for r in rows], "slot_steps_incl_padding": [int(r[1]) for r in rows],
                         "synthesis_s": [round(x, 3) for x in t_loc], "gather_s": [round(r[3], 4) for r in rows]},
            "imbalance_true_samples": round(max(r[0] for r in rows) / (true_total / world), 4),
            "imbalance_synthesis_time": round(max(t_loc) / (sum(t_loc) / world), 4),
            "padding_loss": round(1.0 - true_total / max(sum(r[1] for r in rows), 1.0), 4),
            "broadcast_weights_s": None if t_bcast is None else round(t_bcast, 4),
            "collectives": None if dist is None else "broadcast (weights, rank 0 -> all), barrier x2, gather_object (waveforms -> rank 0), all_gather (report)"}


def rank_report(dist, dev, kernel_ms, last_kernel):
    """What every rank ran, gathered on all ranks: a rank that silently fell back to the generic kernel would otherwise hide inside a
    MAX-reduced time.  Returns {"kernel_ms": [...], "last_kernel": [...]} by rank."""
    mine = torch.tensor([float(kernel_ms), float(last_kernel)], dtype=torch.float64, device=dev)
    if dist is None:
        rows = [mine]
    else:
        rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
    names = {0: "none", 1: "generic", 2: "ring", 3: "group ring"}
    ms = [round(float(r[0]), 3) for r in rows]
    return {"kernel_ms_by_rank": ms, "kernel_ms_min": min(ms), "kernel_ms_max": max(ms),
            "last_kernel_by_rank": [names.get(int(r[1]), "?") for r in rows]}


def job_inputs(args, kw):
    """The seeded job: frames per utterance (1.0 .. 8.0 s at 24 kHz, hop 256), mels, speaker ids (speaker-conditioned models)."""
    gen = torch.Generator().manual_seed(2024)
    frames = torch.randint(94, 751, (args.job,), generator=gen).tolist()
    mels = [torch.randn(kw["cin_channels"], f, generator=gen) for f in frames]
    spk = None
    if kw.get("gin_channels", -1) > 0:
        spk = torch.randint(0, kw["n_speakers"], (args.job,), generator=gen).tolist()
    return frames, mels, spk


def job_plan(args, kw, world, rank, slots):
    """What the scheduler does with the job on this rank, without synthesising anything: (true samples, padded samples, launches) --
    sharding.lpt_assign over the ranks, then packed slots (plan_slots: a slot's cost is the SUM of its utterances) or padded groups
    (pack_groups: a group runs to its longest member)."""
    from wavenet_vocoder_amd import sharding
    frames, _, _ = job_inputs(args, kw)
    lengths = [f * 256 for f in frames]
    mine = sharding.lpt_assign(lengths, world)[rank]
    true = sum(lengths[i] for i in mine)
    if args.packed:
        bins = sharding.plan_slots([lengths[i] for i in mine], slots) if mine else []
        T = max((sum(lengths[mine[k]] for k in b) for b in bins), default=0)
        return true, len(bins) * T, 1 if mine else 0
    groups = sharding.pack_groups(mine, lengths, args.job_group if args.job_group > 0 else None)
    return true, sum(len(g) * max(lengths[i] for i in g) for g in groups if len(g)), len(groups)


def job_mode(args, model, kw, dev, dist, world, rank):
    """N utterances of different lengths through the scheduler the evaluate front end uses (wavenet_vocoder_amd/sharding.py;
    reference evaluate.py:51-92,204-215 + egs/mol/run.sh:31): longest-first assignment to the ranks, groups of neighbouring length
    (group size from the measured throughput curve unless --job-group fixes it), every group run to its longest member, waveforms
    trimmed.  One "step" = the whole job.  `value` counts TRUE samples (padding excluded); the padding loss is reported beside it.
    Strong scaling: the job is the same whatever the number of ranks."""
    from wavenet_vocoder_amd import sharding
    hop, pad = 256, int(kw.get("cin_pad", 0))
    frames, mels, spk = job_inputs(args, kw)
    lengths = [f * hop for f in frames]
    model.rng = "philox"                                                                      # in-kernel noise, as the fixed-batch line
    launches = []

    def synth(c, idx):
        T = (c.shape[-1] - 2 * pad) * hop
        launches.append((len(idx), T))
        g = None if spk is None else torch.tensor([[spk[i]] for i in idx], dtype=torch.long, device=dev)
        y = model.incremental_forward(c=c.to(dev), g=g, T=T)
        return y[:, 0]

    group = args.job_group if args.job_group > 0 else None

    def run_padded():
        st = {}
        launches.clear()
        wavs = sharding.synthesize_sharded(mels, synth, hop_size=hop, cin_pad=pad, group_size=group, stats=st, gather_to=0)
        return wavs, st

    def run_packed():
        """Every rank packs its share of the job (longest-first over the ranks) into slots of one launch."""
        st = {}
        launches.clear()
        mine = sharding.lpt_assign(lengths, world)[rank]
        outs = sharding.synthesize_packed(model, mels, hop_size=hop, cin_pad=pad, slots=group, indices=mine, stats=st, seed=4321,
                                          speaker_ids=spk)
        launches.append((st.get("slots", 0), st.get("slot_steps", 0)))
        st["groups"] = [mine]
        local = {i: o[0].detach().to("cpu") for i, o in zip(mine, outs)}
        if dist is None:
            return [local[i] for i in range(len(mels))], st
        parts = [None] * world if rank == 0 else None
        dist.gather_object(local, parts, dst=0)
        if rank != 0:
            return None, st
        merged = {}
        for part in parts:
            merged.update(part)
        return [merged[i] for i in range(len(mels))], st

    run = run_packed if args.packed else run_padded

    for _ in range(max(args.warmup, 1)):                                                     # engine, scratch, mailboxes exist
        run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wavs, st = run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mine = [st["true_samples"], st["padded_samples"], len(st["groups"])]
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rows = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(mine, dtype=torch.float64, device=dev))
        per_rank = [[int(x) for x in r.tolist()] for r in rows]
    else:
        per_rank = [mine]
    ranks = rank_report(dist, dev, elapsed / args.steps * 1e3, model._get_engine().last_kernel())     # (every rank: an all_gather)
    if rank == 0:
        assert wavs is not None and len(wavs) == args.job and all(w.numel() == n for w, n in zip(wavs, lengths))
        assert all(torch.isfinite(w).all() and float(w.std()) > 1e-3 for w in wavs), "dead or non-finite waveform in the job"
        true_total, padded_total = sum(lengths), sum(r[1] for r in per_rank)
        value = true_total * args.steps / elapsed / 1e3
        line = {"metric": "audio kSamples/sec (24 kHz MoL egs/mol), whole job, TRUE samples of a variable-length job", "value": round(value, 3),
                "unit": "kSamples/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic (seeded N(0,1) mels of seeded lengths 1-8 s, random-init weights, in-kernel Philox noise)",
                "config": {"workload": f"{args.workload}: {describe(kw)}; JOB of {args.job} utterances, {min(frames)}-{max(frames)} frames "
                                       f"({true_total / 24000.0:.1f} s of audio), scheduler = "
                                       + (f"lpt_assign + PACKED SLOTS (continuous batching, {sharding.packed_group_size(model) if group is None else group} slots per GPU)" if args.packed
                                          else f"lpt_assign + pack_groups({'auto' if group is None else group})"),
                           "parallelism": f"utterance-sharded x{world}"},
                "job": {"utterances": args.job, "true_samples": true_total, "padded_samples": padded_total,
                        "padding_loss": round(1.0 - true_total / padded_total, 4),
                        "kSamples_per_s_incl_padding": round(padded_total * args.steps / elapsed / 1e3, 1),
                        "rank0_launches_B_x_T": launches, "per_rank_true_padded_groups": per_rank,
                        "load_imbalance": round(max(r[1] for r in per_rank) / (padded_total / world), 4),
                        "x_real_time_24k_whole_job": round(true_total / 24000.0 / (elapsed / args.steps), 2)},
                "ranks": ranks, "distributed": dist_info(dist, world, args)}
        print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 ring, 3 group ring (wide models)")
    ap.add_argument("--T", type=int, default=T_SAMPLES)
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--cpu-steps", type=int, default=1024, help="T of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduce / print plumbing only, on CPU with gloo (no engine, no GPU)")
    ap.add_argument("--job", type=int, default=0,
                    help="JOB MODE: synthesise N utterances of seeded lengths 1-8 s through the scheduler (sharding.lpt_assign + "
                         "pack_groups) instead of one fixed batch; reports true kSamples/s and the padding loss")
    ap.add_argument("--job-group", type=int, default=0, help="job mode: utterances per launch (0 = sharding.auto_group_size)")
    ap.add_argument("--packed", action="store_true",
                    help="job mode: PACKED SLOTS (continuous batching, sharding.synthesize_packed) instead of padded groups")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed leg even with ONE rank: init_process_group('nccl') = RCCL with world size 1, both barriers, "
                         "the MAX all_reduce, the all_gather of the rank report and (job mode) the gather_object -- what a multi-GPU "
                         "launch executes, on one GPU (tests/test_gpu_zz_boundary.py)")
    ap.add_argument("--no-strong-jobs", dest="strong_jobs", action="store_false",
                    help="skip the two strong-scaled BASELINE jobs (configs[3] / configs[4]) that follow the timed batch")
    ap.add_argument("--strong-utts", default="64,128",
                    help="utterances of the two strong-scaled jobs (BASELINE: 64,128); another value also runs them under --no-extras (tests)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (no throughput_mode, no cpu_baseline): what the rocprofv3 summaries are taken with")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))                  # one process per GPU; rank 0 prints the JSON line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run:
        return dry_run(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                     # --force-dist without a launcher: a rendezvous of one
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from tests._configs import CONFIGS, build, inputs
    name, B, T = args.workload, args.batch, args.T
    kw = CONFIGS[name]
    model_cpu = build(name, seed=0)                       # same weights on every rank (replicated)
    c, gids = inputs(name, B, T, seed=1 + rank)           # each rank: its own 8 utterances (weak scaling)
    import copy
    model = copy.deepcopy(model_cpu).to(dev)
    eng = model._get_engine()
    c_dev = None if c is None else c.to(dev)             # (BASELINE configs[0] has no conditioning)
    g_dev = None if gids is None else gids[:, 0].to(dev)

    if args.job > 0:
        rc = job_mode(args, model, kw, dev, dist, world, rank)
        if dist is not None:
            dist.destroy_process_group()
        return rc

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + args.warmup))]
    kern_ms = []

    def one_step(i):
        c_up = None if c_dev is None else eng.upsample(c_dev, T_expected=T)
        ev[2 * i].record()
        out, _, _ = eng.generate(B=B, T=T, c_up=c_up, g_ids=g_dev, seed=1000 + i, kernel=args.kernel)
        ev[2 * i + 1].record()
        return out

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = one_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        out = one_step(i)
    fence()
    elapsed = time.perf_counter() - t0
    for i in range(args.warmup, args.warmup + args.steps):
        kern_ms.append(ev[2 * i].elapsed_time(ev[2 * i + 1]))
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0 + 1e-6
    assert float(out.float().std()) > 1e-3, "the timed kernel wrote a constant waveform: a dead kernel must not post a number"
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ranks = rank_report(dist, dev, sum(kern_ms) / len(kern_ms), eng.last_kernel())
    dinfo = dist_info(dist, world, args, dev)             # (every rank: it holds an all_reduce)
    # THE STRONG-SCALED JOBS of BASELINE.json (configs[3]: 64 utterances, configs[4]: 128) behind the weak-scaled batch: the weak line grows
    # ~N-fold by construction, these say what N GPUs do to ONE job (VERDICT r05 next #6a).  Every rank takes part; skipped with --no-extras.
    strong = None
    n3, n4 = (int(x) for x in args.strong_utts.split(","))
    if args.strong_jobs and args.batch == B_PER_GPU and args.workload == WORKLOAD and (not args.no_extras or args.strong_utts != "64,128"):
        strong = []
        for jname, jn in (("cfg3b_gaussian30", n3), ("cfg4_mol_multispeaker", n4)):
            try:
                strong.append(strong_job(jname, jn, dev, dist, world, rank))
            except Exception as e:                        # (a single process only: with a process group a rank must not leave the others waiting)
                if dist is not None:
                    raise
                strong.append({"workload": jname, "error": str(e)[:160]})

    if rank == 0:
        total_samples = world * B * T * args.steps
        value = total_samples / elapsed / 1e3
        per_utt = T * args.steps / elapsed
        kdur = sum(kern_ms) / len(kern_ms) / 1e3                      # seconds per sample-loop launch
        alg_bytes = eng.bytes_per_step(B) * T                          # per launch
        ach = alg_bytes / kdur / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = {k: tj.get(k) for k in ("source", "commit", "kernel_version", "collected_with") if tj.get(k) is not None}
                traffic_src["note"] = "quoted from a separate rocprofv3 --pmc run (counters cannot be read from inside this process)"
                # (round 6) where the waves' time goes, from the issue-side counters of the same passes: shares of
                # SQ_WAVE_CYCLES over the whole launch (all roles together -- the counters are per kernel, not per role; per-role
                # timelines come from the trace builds' in-kernel stamps, profiles/r0*_timeline.txt)
                iss = tj.get("issue_side") or {}
                if iss:
                    traffic_src["issue_side"] = {"valu_busy_share_of_wave_time": round(iss.get("wave_time_valu_busy_share", 0.0), 4),
                                                 "issuing_any_instruction_share": round(iss.get("wave_time_issuing_share", 0.0), 4),
                                                 "valu_insts_per_vmem_read_poll": round(iss.get("valu_insts_per_vmem_read", 0.0), 1),
                                                 "lds_active_over_sq_busy": round(iss.get("lds_active_over_sq_busy", 0.0), 4),
                                                 "l2_hit_rate": round(tj.get("l2_hit_rate", 0.0), 4)}
            except Exception:
                traffic, traffic_src = None, None
        us_step = kdur / T * 1e6
        # the MEASURED LDS read peak of this box next to the nominal one (csrc/wnv_ubench.hip, ~5 ms)
        peak_meas, peak_cus = None, None
        try:
            from wavenet_vocoder_amd import _lib
            gbs, peak_cus = _lib.measure_lds_read_peak(local_rank)           # a bench tool of libwnv_test.so (include/wnv_test.h), not product
            peak_meas = round(gbs, 1)
        except Exception as e:
            peak_meas = None
            print(f"[bench] LDS peak microbenchmark failed: {e}", file=sys.stderr)
        nparts = max(kw["skip_out_channels"] // 128, 1)
        floor, floor_hop2 = latency_floor_us(kw, nparts), latency_floor_us(kw, nparts, hop=0.444)
        floor_meas, hop_meas, phase_meas = measured_floor_us(kw, nparts)
        line = {
            "metric": "audio kSamples/sec (24 kHz MoL egs/mol, batch=8 per GPU), whole job",
            "value": round(value, 3), "unit": "kSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded N(0,1) mel, random-init weights of the egs/mol architecture, in-kernel Philox noise)",
            "config": {"workload": f"{name}: {describe(kw)}, B={B} utterances/GPU x T={T} samples", "batch_per_gpu": B, "T": T,
                       "kernel": {0: "auto", 1: "generic", 2: "ring", 3: "group ring"}[args.kernel] + f" (ran: {({1: 'generic', 2: 'ring', 3: 'group ring'}).get(eng.last_kernel(), '?')})",
                       "parallelism": f"utterance-sharded x{world}"},
            "kSamples_per_s_per_gpu": round(value / world, 3),
            "samples_per_s_per_utterance": round(per_utt, 1),
            "rtf_24k": round(per_utt / 24000.0, 4), "rtf_22k05": round(per_utt / 22050.0, 4),
            "roofline": {"bound": "lds", "peak_name": "LDS read bandwidth, 256 CU x 256 B/clk x 2.4 GHz (BASELINE.json / SURVEY.md 8d: weights resident on chip)",
                         "achieved": round(ach, 2), "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": round(ach / LDS_PEAK_GBS, 6),
                         "peak_measured": peak_meas, "frac_of_peak_measured": None if not peak_meas else round(ach / peak_meas, 6),
                         "peak_measured_how": f"wnv_measure_lds_read_peak on this box: {peak_cus} CUs x 16 waves of conflict-free ds_read_b128 "
                                              "(csrc/wnv_ubench.hip), bytes read / HIP-event time, best of 5 launches",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": round(kdur * 1e3, 3), "algorithmic_bytes_per_launch": alg_bytes},
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach / HBM_PEAK_GBS, 5),
                             "note": "for reference only: the sample loop does not stream from HBM (traffic << algorithmic bytes)"},
            "roofline_latency": {"bound": "serial chain latency", "floor_us_per_step": round(floor, 3), "achieved_us_per_step": round(us_step, 3),
                                 "frac": round(floor / us_step, 4),
                                 "model": "HARDWARE constants (what rounds 3-4 reported; comparable across rounds): stages on the chain x (best same-XCD CU->CU hop "
                                          "0.276 us [ubench_hop4] + the chain phase's 256x128 mat-vec at the CU's fp32 FMA peak, 0.107 us) + head (2 hops + KxK and "
                                          "OxK mat-vecs at the FMA peak)",
                                 "floor_us_per_step_measured_phase": round(floor_meas, 3), "frac_vs_measured_phase": round(floor_meas / us_step, 4),
                                 "model_measured_phase": f"the same with the chain phase priced at this implementation's own best isolated phase, {phase_meas} us "
                                          f"[scripts/ubench_phase.hip: barrier, LDS read, 256x128 mat-vec, reduce, gate, store; hop {hop_meas} us; "
                                          "profiles/r05_latency_constants.json] -- NOT a roofline (circular: built from the implementation's own phase), "
                                          "it says how much of the step is outside hop + phase",
                                 "floor_us_per_step_hop2_0p444": round(floor_hop2, 3), "frac_hop2_0p444": round(floor_hop2 / us_step, 4)},
            "ranks": ranks,
            "distributed": dinfo,
            "strong_scaled_jobs": strong,
        }
        if world == 1 and args.batch == B_PER_GPU and args.workload == WORKLOAD and not args.no_extras:
            # informative only (not `value`): the same kernel with 48 utterances per GPU -- the rings pipeline six
            # utterances each like a systolic array (DESIGN.md 5.2 "Throughput vs. batch")
            try:
                B2, T2 = 48, 8192                      # (round 4: the plateau moved from 32 to 48 utterances per GPU)
                c2, g2 = inputs(name, B2, T2, seed=7)
                c2 = c2.to(dev)
                eng.generate(B=B2, T=T2, c_up=eng.upsample(c2, T_expected=T2), seed=1, kernel=args.kernel)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.generate(B=B2, T=T2, c_up=eng.upsample(c2, T_expected=T2), seed=2, kernel=args.kernel)
                torch.cuda.synchronize()
                line["throughput_mode"] = {"batch_per_gpu": B2, "T": T2,
                                           "kSamples_per_s_per_gpu": round(B2 * T2 / (time.perf_counter() - t1) / 1e3, 1)}
            except Exception as e:  # the headline line must not depend on this extra
                line["throughput_mode"] = {"error": str(e)[:120]}
        if world == 1 and not args.no_extras:
            # informative only: the same batch through the reference-compatible public entry point -- WaveNet.incremental_forward with
            # its default rng = "replay" (the noise tape the reference's CPU generator would have produced for the current seed is
            # built on the host first), upsampling included
            try:
                torch.manual_seed(0)
                init = None
                if c_dev is None and gids is None:                # (unconditioned: the batch size comes with the initial input, wavenet.py:283-289)
                    if kw.get("scalar_input", False):
                        init = torch.zeros(B, 1, 1, device=dev)
                    else:
                        init = torch.zeros(B, kw["out_channels"], 1, device=dev)
                        init[:, 127] = 1.0
                model.incremental_forward(initial_input=init, c=c_dev, g=None if gids is None else gids.to(dev), T=T)      # warm: engine + scratch exist
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.incremental_forward(initial_input=init, c=c_dev, g=None if gids is None else gids.to(dev), T=T)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                line["api_path"] = {"entry": "WaveNet.incremental_forward(c, T) [rng='replay']", "kSamples_per_s_per_gpu": round(B * T / dt / 1e3, 1),
                                    "ms_per_call": round(dt * 1e3, 2)}
            except Exception as e:
                line["api_path"] = {"error": str(e)[:120]}
        if world == 1 and args.batch == B_PER_GPU and args.workload == WORKLOAD and not args.no_extras:
            # informative only: the two other device paths of SURVEY.md section 8 at their own bench shapes, so that a driver-run line
            # carries them too -- f3 (teacher-forced batch `forward` on the f32 matrix cores, same model and batch) and the group-ring
            # kernel on the published wide geometry (24 layers 512 / 512 / 256), one and eight utterances
            try:
                gx = torch.Generator().manual_seed(3)             # a seeded teacher input (the bench model takes a scalar waveform)
                if kw.get("scalar_input", False):
                    xt = torch.tanh(torch.randn(B, 1, T, generator=gx) * 0.5).to(dev)
                else:
                    xi = torch.randint(0, kw["out_channels"], (B, T), generator=gx)
                    xt = torch.zeros(B, kw["out_channels"], T).scatter_(1, xi.unsqueeze(1), 1.0).to(dev)
                c_up_f = None if c_dev is None else eng.upsample(c_dev, T_expected=T)
                gi = None if gids is None else gids[:, 0].to(dev)
                for _ in range(2):
                    eng.forward(xt, c_up=c_up_f, g_ids=gi)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(3):
                    eng.forward(xt, c_up=c_up_f, g_ids=gi)
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / 3
                tf = 2.0 * eng.macs_per_sample() * B * T / ms / 1e9
                line["f3_forward"] = {"entry": "wnv_forward (WaveNet.forward, teacher-forced, f32 MFMA)", "ms_per_call": round(ms, 3),
                                      "TFLOP_per_s": round(tf, 1), "frac_of_f32_mfma_peak_157.3": round(tf / 157.3, 4)}
                del xt, c_up_f
            except Exception as e:
                line["f3_forward"] = {"error": str(e)[:120]}
            try:
                wname, Tw = "wide_mol_512", 4096
                mw = build(wname).to(dev)
                ew = mw._get_engine()
                wide = {"model": "24 layers 512/512/256, 80-mel MoL (group-ring kernel)", "T": Tw}
                for Bw in (1, 8):
                    cw, _ = inputs(wname, Bw, Tw)
                    cuw = ew.upsample(cw.to(dev), T_expected=Tw)
                    ew.generate(B=Bw, T=Tw, c_up=cuw, seed=1, kernel=0)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    ew.generate(B=Bw, T=Tw, c_up=cuw, seed=2, kernel=0)
                    torch.cuda.synchronize()
                    wide[f"kSamples_per_s_B{Bw}"] = round(Bw * Tw / (time.perf_counter() - t1) / 1e3, 1)
                wide["kernel"] = {1: "generic", 2: "ring", 3: "group ring"}.get(ew.last_kernel(), "?")
                line["wide_model"] = wide
                del mw, ew
            except Exception as e:
                line["wide_model"] = {"error": str(e)[:120]}
        if world == 1 and args.cpu_steps > 0 and not args.no_extras:
            line["cpu_baseline"] = cpu_baseline(model_cpu, kw, c, args.cpu_steps, B=B, gids=gids)
            line["cpu_baseline"]["batch"] = line["cpu_baseline"].get("batch", B)
            # like for like: aggregate rate of the SAME number of utterances on both sides (a "port" leg of an unconditioned model
            # runs one utterance: then the per-utterance rates are compared)
            cb = line["cpu_baseline"]
            line["speedup_vs_cpu_baseline"] = round((value / world / B) / (cb["value"] / cb["batch"]), 1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
