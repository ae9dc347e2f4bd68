"""Run the ring kernel once with WNV_RING_TRACE and print a per-stage timeline (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ring_trace.txt"
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.environ.get("TRACE_OFF"):
    os.environ["WNV_RING_TRACE"] = out
import torch
from tests._configs import build, inputs
B, T = int(os.environ.get("B", 8)), 4096
CFG = os.environ.get("CFG", "cfg2_mol")
m = build(CFG).to("cuda")
eng = m._get_engine()
c, g = inputs(CFG, B, T)
c_up = eng.upsample(c.cuda(), T_expected=T)
eng.generate(B=B, T=T, c_up=c_up, seed=1, kernel=2, **({"g_ids": g.cuda()} if g is not None else {}))
torch.cuda.synchronize()
if os.environ.get("TRACE_OFF"):
    print("ran without the trace switch, last kernel", eng.last_kernel()); sys.exit(0)
rows = [l.split() for l in open(out) if not l.startswith("#")]
S = max(int(r[1]) for r in rows)
steps = sorted({int(r[0]) for r in rows})
print("step-to-step period (head send -> next head send), ns:")
hs = {int(r[0]): int(r[2]) for r in rows if int(r[1]) == S}
print([hs[b] - hs[a] for a, b in zip(steps, steps[1:])])
t = steps[2]
print(f"timeline of step {t} (ns, relative to the head's send of this step):")
base = hs[t]
prev_send = base
for pos in list(range(S)) + [S]:
    r = [x for x in rows if int(x[0]) == t and int(x[1]) == pos][0]
    v = [int(x) - base if int(x) >= 0 else None for x in r[2:]]
    if pos < S and v[0] is None:                     # (position 0 of a ring whose head evaluates layer 0 is empty)
        continue
    if pos < S:
        print(f" stage {pos:2d}: recv {v[0]:6d} (hop {v[0]-prev_send:5d})  gate/send-u +{v[1]-v[0]:5d}  h-sent +{v[2]-v[1]:5d}  skip-sent +{v[3]-v[2]:5d}  deferred-done +{v[4]-v[3]:5d}")
        prev_send = v[1]
        last_skip = v[3]
    else:
        # head stamps of step t: [0] send (this step), [1] skip received, [2] sample done
        print(f" head    : skip recv {v[1]:6d} (after last stage's skip send {v[1]-last_skip:5d})  sample done +{v[2]-v[1]:5d}")
