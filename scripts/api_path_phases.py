"""Where does WaveNet.incremental_forward (rng = "replay", one-hot model, streamed tape) spend its time?  cProfile of one warm call."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests._configs import build, inputs
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_mulaw256"
B, T = 8, 8192
m = build(name).to("cuda")
c, g = inputs(name, B, T)
c = c.cuda()
torch.manual_seed(0)
m.incremental_forward(c=c, T=T); torch.cuda.synchronize()
eng = m._get_engine()
c_up = eng.upsample(c, T_expected=T)
torch.cuda.synchronize(); t = time.perf_counter(); eng.generate(B=B, T=T, c_up=c_up, seed=1, kernel=2); torch.cuda.synchronize()
print("kernel alone (Philox): %.1f ms" % ((time.perf_counter() - t) * 1e3))
pr = cProfile.Profile()
torch.cuda.synchronize(); t = time.perf_counter()
pr.enable(); m.incremental_forward(c=c, T=T); torch.cuda.synchronize(); pr.disable()
print("incremental_forward: %.1f ms" % ((time.perf_counter() - t) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
