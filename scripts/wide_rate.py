"""Rate of the group-ring kernel on a wide model.  usage: [B=1,8] [T=4096] [CASE=wide_mol_512] python scripts/wide_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests._configs import build, inputs
case = os.environ.get("CASE", "wide_mol_512")
T = int(os.environ.get("T", 4096))
m = build(case).to("cuda")
eng = m._get_engine()
for B in [int(x) for x in os.environ.get("B", "1,8").split(",")]:
    c, _ = inputs(case, B, T)
    cu = eng.upsample(c.cuda(), T_expected=T)
    best = 0.0
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.generate(B=B, T=T, c_up=cu, seed=1, kernel=3)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, B * T / dt)
    print(f"{case} B={B} quiet={os.environ.get('WNV_WIDE_QUIET_NS', 'default')}: {best / 1e3:.1f} kSamples/s ({1e6 * B / best:.2f} us/step)")
