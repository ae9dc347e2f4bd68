import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._configs import build, inputs
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mol"
B, T = int(sys.argv[2]) if len(sys.argv) > 2 else 1, 256
m = build(name).to("cuda")
eng = m._get_engine()
c, g = inputs(name, B, T)
c_up = eng.upsample(c.cuda(), T_expected=T)
try:
    out, _, _ = eng.generate(B=B, T=T, c_up=c_up, g_ids=None if g is None else g[:, 0].cuda(), seed=1, kernel=2)
    ref, _, _ = eng.generate(B=B, T=T, c_up=c_up, g_ids=None if g is None else g[:, 0].cuda(), seed=1, kernel=1)
    print(name, B, "ring ran; max diff vs generic", float((out - ref).abs().max()), flush=True)
except Exception as e:
    print(name, B, "ring failed:", type(e).__name__, str(e)[:200], flush=True)
