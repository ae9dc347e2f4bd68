"""Timeline of one sample step of the group-ring kernel (WNV_WIDE_TRACE): per group, when its inputs were gathered, its partial sums
were in LDS, u was published, the own h was gathered and the next step's pre-activations were ready.  usage: python scripts/trace_wide.py out.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wide_trace.txt"
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["WNV_WIDE_TRACE"] = out
import torch
from tests._configs import build, inputs
B, T = int(os.environ.get("B", 1)), 4096
m = build("wide_mol_512").to("cuda")
eng = m._get_engine()
c, _ = inputs("wide_mol_512", B, T)
eng.generate(B=B, T=T, c_up=eng.upsample(c.cuda(), T_expected=T), seed=1, kernel=3)
torch.cuda.synchronize()
rows = {(int(f[0]), int(f[1])): [int(x) for x in f[2:]] for f in (l.split() for l in open(out) if not l.startswith("#"))}
L = max(k[1] for k in rows)
steps = sorted({k[0] for k in rows})
t = steps[3]
print("step period (head send -> head send), ns:", [rows[(b, L)][1] - rows[(a, L)][1] for a, b in zip(steps, steps[1:])])
prev_pub = rows[(t - 1, L)][1]
acc = {}
for l in range(L):
    v = rows[(t, l)]
    line = (f" group {l:2d}: gathered {v[0]:7d} (hop {v[0] - prev_pub:5d}) | pass +{v[1] - v[0]:4d} | u published +{v[2] - v[1]:4d} | rest of the batch + own h stored +{v[3] - v[2]:6d} "
            f"| tap inputs +{v[5] - v[3]:5d} | stream +{v[6] - v[5]:6d} | pre ready +{v[4] - v[6]:5d}")
    if len(v) >= 13 and v[12] >= 0 and v[10] >= 0:
        # deferred stamps of the chain (trace builds since the end of round 3): u / h arrival as the polling waves saw it, wave 0's own share of
        # the pass, the wait at the barrier behind it, the sum of the partial sums, gate + store, and when h left
        d = dict(u_arrived_before_barrier=v[0] - v[7] if v[7] >= 0 else None, h_arrived_before_barrier=v[0] - v[8], pass_wave0=v[10] - v[0],
                 barrier_behind_pass=v[1] - v[10], sum_partials=v[12] - v[1], gate_store=v[2] - v[12], h_published_after_barrier=v[11] - v[1])
        line += " || " + " ".join(f"{k} {x}" for k, x in d.items())
        if l >= 1:
            for k, x in d.items():
                if x is not None:
                    acc.setdefault(k, []).append(x)
    print(line)
    prev_pub = v[2]
v = rows[(t, L)]
print(f" head    : skip gathered {v[0]:7d} (after the last group's u {v[0] - prev_pub:5d}) | next input sent +{v[1] - v[0]:5d}")
for k, x in acc.items():
    print(f"  mean {k:28s} {sum(x) / len(x):7.1f}  (min {min(x)}, max {max(x)})")
