#!/usr/bin/env python3
"""Kernel rate of one library for a list of batch sizes, no output checks (experiment builds produce wrong samples on purpose).
    WNV_LIB=<lib.so> python scripts/exp_rate.py <workload> <T> <B,B,...> [label]
prints one line per batch: label workload B kSamples/s us_per_step (best of 2 timed launches after one warm-up launch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tests._configs import build, inputs  # noqa: E402

name, T = sys.argv[1], int(sys.argv[2])
batches = [int(x) for x in sys.argv[3].split(",")]
label = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(os.environ.get("WNV_LIB", "product"))
dev = torch.device("cuda", 0)
m = build(name, seed=0).to(dev)
eng = m._get_engine()
for B in batches:
    c, gids = inputs(name, B, T, seed=1)
    g = None if gids is None else gids[:, 0].to(dev)
    c_up = eng.upsample(c.to(dev), T_expected=T)
    best = 0.0
    try:
        for i in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.generate(B=B, T=T, c_up=c_up, g_ids=g, seed=100 + i, kernel=int(os.environ.get("EXP_KERNEL", "0")))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if i > 0:
                best = max(best, B * T / dt / 1e3)
        print(f"{label} {name} B={B} {best:.1f} kSamples/s  {B / best * 1e3:.2f} us/step  kernel={eng.last_kernel()} tap={os.environ.get('WNV_RING_TAP', '-')}", flush=True)
    except Exception as e:
        print(f"{label} {name} B={B} FAILED {type(e).__name__}: {str(e)[:160]}", flush=True)
