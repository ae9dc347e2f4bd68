"""Throughput of the MFMA teacher-forced batch evaluation (wnv_forward) at the bench shape:
egs/mol, B = 8 utterances x T = 24064 samples.  Reports TFLOP/s against the f32 MFMA peak (157.3 TFLOP/s)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests._configs import CONFIGS, build, inputs
from tests.test_gpu_configs import teacher
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mol"
B, T = 8, 94 * 256
m = build(name).to("cuda")
eng = m._get_engine()
c, gids = inputs(name, B, T)
c_up = eng.upsample(c.cuda(), T_expected=T)
x = teacher(CONFIGS[name], B, T).cuda()
gi = None if gids is None else gids[:, 0].cuda()
for _ in range(2):
    y = eng.forward(x, c_up=c_up, g_ids=gi)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 5
ev[0].record()
for _ in range(n):
    y = eng.forward(x, c_up=c_up, g_ids=gi)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
flops = 2.0 * eng.macs_per_sample() * B * T
with torch.enable_grad():
    ref = m(x[:1, :, :2048], c=c[:1, :, :8 + 4].cuda()) if False else None
print(json.dumps({"workload": name, "B": B, "T": T, "ms_per_forward": round(ms, 3), "TFLOP_per_s": round(flops / ms / 1e9, 2),
                  "frac_of_f32_mfma_peak": round(flops / ms / 1e9 / 157.3, 4), "samples_per_s": round(B * T / ms * 1e3)}))
