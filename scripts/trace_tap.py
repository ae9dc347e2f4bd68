"""Timeline of the passes of one tap workgroup (layer 6 -- dilation 1 --, part 0) from a -DWNV_FINE_TRACE build: per pass
start | inputs written (before the wait for h) | barrier passed (h filed) | round 0 done | round 1 done, in ns relative to the first pass's start;
[LDP per wave: the h record came from the speculative Look / the Direct look / the Patient receive].
    B=48 WNV_LIB=<trace lib> python scripts/trace_tap.py [raw file]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raw = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tap_trace_raw.txt"
os.makedirs(os.path.dirname(raw), exist_ok=True)
os.environ["WNV_RING_TRACE"] = raw
import torch
from tests._configs import build, inputs
B, T = int(os.environ.get("B", 48)), 4096
m = build("cfg2_mol").to("cuda")
eng = m._get_engine()
c, _ = inputs("cfg2_mol", B, T)
eng.generate(B=B, T=T, c_up=eng.upsample(c.cuda(), T_expected=T), seed=1, kernel=2)
torch.cuda.synchronize()
rows = [[int(x) for x in l.split()[1:]] for l in open(raw) if l.startswith("#tap")]
print(f"B = {B}: passes of the tap workgroup of layer 6, part 0 (ns from the step's first pass start)")
prev = None
for r in rows[2:int(os.environ.get('ROWS', 7))]:
    step, v = r[0], r[1:]
    base = v[0]
    out = []
    for k in range(3):
        w = v[5 * k:5 * k + 5]
        if w[0] < 0:
            continue
        if os.environ.get("REL") == "barrier":      # every stamp relative to the barrier (common to all waves): compare builds that stamp different waves
            out.append(f"pass {k}: start {w[0] - w[2]} [B] done {w[1] - w[2]} | barrier 0 | round 0 done +{w[3] - w[2]} | round 1 done +{w[4] - w[2]} | next barrier +{(v[5 * k + 7] if k < 2 and v[5 * k + 7] >= 0 else 0) - w[2]}")
            continue
        how = "".join("LDP?"[(v[15] >> (16 * k + 2 * w_)) & 3] for w_ in range(8)) if len(v) > 15 and v[15] >= 0 else ""
        out.append((f"[{how}] " if how else "") + f"pass {k}: start {w[0] - base} | wave 0: h filed, DMAs landed +{w[1] - w[0]} | barrier +{w[2] - w[1]} | round 0 +{w[3] - w[2]}" + (f" | round 1 +{w[4] - w[3]}" if w[4] >= 0 else ""))
    print(f" step {step}" + (f" (period {base - prev})" if prev is not None else "") + ": " + " || ".join(out))
    prev = base
