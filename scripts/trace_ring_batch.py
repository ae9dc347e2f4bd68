"""Per-utterance service time of a ring stage when the ring carries several utterances (B = 8 x upr): from the "#u j step pos stamps" rows
of a -DWNV_FINE_TRACE=2 trace.   usage: B=64 WNV_LIB=<trace lib> python scripts/trace_ring_batch.py [raw trace file]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raw = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ring_trace_batch.txt"
if not os.path.exists(raw) or os.environ.get("RUN", "1") == "1":
    os.makedirs(os.path.dirname(raw), exist_ok=True)
    os.environ["WNV_RING_TRACE"] = raw
    import torch
    from tests._configs import build, inputs
    B, T = int(os.environ.get("B", 64)), 4096
    NAME = os.environ.get("NAME", "cfg2_mol")
    m = build(NAME).to("cuda")
    eng = m._get_engine()
    c, gid = inputs(NAME, B, T)
    eng.generate(B=B, T=T, c_up=eng.upsample(c.cuda(), T_expected=T), g_ids=None if gid is None else gid[:, 0].cuda(), seed=1, kernel=2)
    torch.cuda.synchronize()
rows = {}
for l in open(raw):
    f = l.split()
    if l.startswith("#u"):
        rows[(int(f[1]), int(f[2]), int(f[3]))] = [int(x) for x in f[4:]]
    elif not l.startswith("#"):
        rows[(0, int(f[0]), int(f[1]))] = [int(x) for x in f[2:]]
upr = 1 + max(k[0] for k in rows)
steps = sorted({k[1] for k in rows})
S = max(k[2] for k in rows)
t = steps[3]
names = {8: "X hit", 5: "h formed", 12: "N released", 6: "zin ready", 0: "X+zin in LDS", 1: "u sent", 7: "bar behind u", 2: "q sent", 3: "skip sent", 4: "done"}
for pos in (2, 11, 22):
    print(f"stage {pos}, step {t}: per utterance of ring 0 (ns after the first utterance's X hit)")
    base = rows[(0, t, pos)][8]
    prev_done = None
    for j in range(upr):
        v = rows[(j, t, pos)]
        order = [8, 5, 12, 6, 0, 1, 7, 2, 3, 4]
        print(f"  utt {j}: " + " | ".join(f"{names[k]} {v[k] - base if v[k] >= 0 else None}" for k in sorted(order, key=lambda k: v[k] if v[k] >= 0 else 10**9)))
h = [rows[(j, t, S)] for j in range(upr)]
print("head (ns after its first send of the step):", [[x - h[0][0] if x >= 0 else None for x in hv[:5]] for hv in h])
per = [rows[(0, b, S)][0] - rows[(0, a, S)][0] for a, b in zip(steps, steps[1:])]
print("step period:", per)
# head of ring 0, per utterance of the traced step: 1 skip sum in LDS | 3 hidden layer in LDS | 4 head outputs in LDS | 2 step done | 0 input of the next step sent
hn = {1: "skip sum", 3: "hidden", 4: "outputs", 2: "done", 0: "sent"}
print(f"head, step {t}: per utterance (ns after utterance 0's skip sum)")
hb = rows[(0, t, S)][1]
for j in range(upr):
    v = rows[(j, t, S)]
    v1 = rows.get((j, t + 1, S))
    print(f"  utt {j}: " + " | ".join(f"{hn[k]} {v[k] - hb if v[k] >= 0 else None}" for k in (1, 3, 4, 2)) + (f" | next input sent {v1[0] - hb}" if v1 and v1[0] >= 0 else ""))
