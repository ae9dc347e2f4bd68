"""How do the utterances that share a ring follow each other?  Prints, for one traced step, when each utterance of ring 0 was
received (slot 0) / gated (slot 1) / finished (slot 4) at a few stages and at the head.  usage: B=32 python scripts/convoy_trace.py out.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["WNV_RING_TRACE"] = out
import torch
from tests._configs import build, inputs
B, T = int(os.environ.get("B", 32)), 4096
m = build("cfg2_mol").to("cuda")
eng = m._get_engine()
c, _ = inputs("cfg2_mol", B, T)
eng.generate(B=B, T=T, c_up=eng.upsample(c.cuda(), T_expected=T), seed=1, kernel=2)
torch.cuda.synchronize()
rows = {}
taps = []
for ln in open(out):
    f = ln.split()
    if ln.startswith("#tap"):
        taps.append([int(x) for x in f[1:]])
        continue
    if ln.startswith("#u"):
        j, t, pos, v = int(f[1]), int(f[2]), int(f[3]), [int(x) for x in f[4:]]
    elif ln.startswith("#"):
        continue
    else:
        j, t, pos, v = 0, int(f[0]), int(f[1]), [int(x) for x in f[2:]]
    rows[(j, t, pos)] = v
S = max(k[2] for k in rows)
J = max(k[0] for k in rows) + 1
steps = sorted({k[1] for k in rows})
t = steps[3]
print(f"B = {B}: {J} utterances per ring; step {t}; ns on one clock")
for pos in (0, 1, 2, 12, 22, 23):
    print(f" stage {pos:2d}: " + " | ".join(f"j{j}: recv {rows[(j, t, pos)][0]:7d} gate {rows[(j, t, pos)][1]:7d} done {rows[(j, t, pos)][4]:7d}" for j in range(J)))
print(" head    : " + " | ".join(f"j{j}: skip {rows[(j, t, S)][1]:7d} sampled {rows[(j, t, S)][2]:7d} sent-next {rows[(j, t + 1, S)][0] if (j, t + 1, S) in rows else -1:7d}" for j in range(J)))
for r in taps[2:5]:
    v = r[1:]
    print(f" tap WG (layer 0, part 0, first pass) step {r[0]}: start {v[0]} | records in +{v[1]-v[0]} | gathered +{v[2]-v[1]} | round 0 +{v[3]-v[2]} | round 1 +{v[4]-v[3]} | drained +{v[5]-v[4]}")
