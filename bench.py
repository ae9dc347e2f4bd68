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
roofline : the dominant kernel is the sample-loop kernel.  achieved = algorithmic bytes per launch
           (wnv_bytes_per_step(B) x T, SURVEY.md 8d: every weight once per step per utterance group + ring taps
           + conditioning row + sample) / the kernel's duration measured with HIP events on its own stream.
           `roofline` prices it against the 8 TB/s HBM peak (schema bound "hbm"), `roofline_lds` against the
           LDS read peak BASELINE.json asks for (MI355X_MICROARCH.md: 256 CU x 256 B/clk x 2.4 GHz = 157 TB/s).
cpu_baseline : the CPU oracle (oracle/wavenet_oracle.py, a torch-CPU restatement of the reference's op
           sequence incl. its per-step queue shift) timed on rank 0's host cores on a bounded sample of the SAME
           workload (same weights, mel, batch 8; T_cpu steps), reference thread setting (4) and all cores.
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


def cpu_baseline(model_cpu, kw, c, T_cpu, budget_s=12.0):
    """Time the oracle on the host (checker used as the measured CPU path -- the one place that is allowed).
    Bounded: each thread setting gets at most `budget_s` seconds of wall time (the per-step cost is constant
    once the history buffers exist, so a truncated run measures the same rate)."""
    from oracle.wavenet_oracle import Oracle
    from tests._golden import oracle_config
    from wavenet_vocoder_amd.noise import make_noise_tape
    o = Oracle(oracle_config(kw), model_cpu.state_dict())
    B = c.shape[0]
    frames = T_cpu // 256
    c_cpu = c[:, :, : frames + 2 * kw["cin_pad"]].contiguous()
    tape = make_noise_tape(T_cpu, B, scalar_input=kw.get("scalar_input", False),
                           output_distribution=kw.get("output_distribution", "Logistic"), out_channels=kw["out_channels"],
                           generator=torch.Generator().manual_seed(2))
    results, steps = {}, {}
    ncores = os.cpu_count() or 1
    c_up = o.upsample(c_cpu).contiguous()                 # upsampled once; the timed loop is the sample loop
    saved = o.cfg.upsample_conditional_features
    o.cfg.upsample_conditional_features = False
    for threads in sorted({1, 4, min(ncores, 16)}):       # 4 = the reference's own setting (synthesis.py:37)
        torch.set_num_threads(threads)
        with torch.no_grad():
            o.incremental_forward(c=c_up[:, :, :32], T=32, noise=tape)                     # warm-up
            o.incremental_forward(c=c_up, T=T_cpu, noise=tape, max_seconds=budget_s)
        results[threads] = B * o.last_steps / o.last_seconds / 1e3
        steps[threads] = o.last_steps
    o.cfg.upsample_conditional_features = saved
    best = max(results, key=results.get)
    return {"value": round(results[best], 4), "unit": "kSamples/s", "cores": best, "kind": "port",
            "sample": f"oracle/wavenet_oracle.py (torch-CPU restatement of the reference op sequence incl. its per-step "
                      f"queue shift), same weights/mel, B={B}, up to T={T_cpu} steps or {budget_s:.0f} s per thread setting "
                      f"(steps done: {steps}); host has {ncores} cores",
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 ring")
    ap.add_argument("--T", type=int, default=T_SAMPLES)
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--cpu-steps", type=int, default=1024, help="T of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (no throughput_mode, no cpu_baseline): what the rocprofv3 summaries are taken with")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from tests._configs import CONFIGS, build, inputs
    name, B, T = args.workload, args.batch, args.T
    kw = CONFIGS[name]
    model_cpu = build(name, seed=0)                       # same weights on every rank (replicated)
    c, gids = inputs(name, B, T, seed=1 + rank)           # each rank: its own 8 utterances (weak scaling)
    import copy
    model = copy.deepcopy(model_cpu).to(dev)
    eng = model._get_engine()
    c_dev = c.to(dev)
    g_dev = None if gids is None else gids[:, 0].to(dev)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + args.warmup))]
    kern_ms = []

    def one_step(i):
        c_up = eng.upsample(c_dev, T_expected=T)
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
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total_samples = world * B * T * args.steps
        value = total_samples / elapsed / 1e3
        per_utt = T * args.steps / elapsed
        kdur = sum(kern_ms) / len(kern_ms) / 1e3                      # seconds per sample-loop launch
        alg_bytes = eng.bytes_per_step(B) * T                          # per launch
        ach = alg_bytes / kdur / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "audio kSamples/sec (24 kHz MoL egs/mol, batch=8 per GPU), whole job",
            "value": round(value, 3), "unit": "kSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded N(0,1) mel, random-init weights of the egs/mol architecture, in-kernel Philox noise)",
            "config": {"workload": f"{name}: {describe(kw)}, B={B} utterances/GPU x T={T} samples", "batch_per_gpu": B, "T": T,
                       "kernel": "auto" if args.kernel == 0 else ("generic" if args.kernel == 1 else "ring"),
                       "parallelism": f"utterance-sharded x{world}"},
            "kSamples_per_s_per_gpu": round(value / world, 3),
            "samples_per_s_per_utterance": round(per_utt, 1),
            "rtf_24k": round(per_utt / 24000.0, 4), "rtf_22k05": round(per_utt / 22050.0, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": round(kdur * 1e3, 3), "algorithmic_bytes_per_launch": alg_bytes},
            "roofline_lds": {"bound": "lds", "achieved": round(ach, 2), "peak": LDS_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach / LDS_PEAK_GBS, 6)},
        }
        if world == 1 and args.batch == B_PER_GPU and args.workload == WORKLOAD and not args.no_extras:
            # informative only (not `value`): the same kernel with 32 utterances per GPU -- the rings pipeline four
            # utterances each like a systolic array (DESIGN.md 5.2 "Throughput vs. batch")
            try:
                B2, T2 = 32, 8192
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
        if world == 1 and args.cpu_steps > 0 and not args.no_extras:
            line["cpu_baseline"] = cpu_baseline(model_cpu, kw, c, args.cpu_steps)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
