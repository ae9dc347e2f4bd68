"""HBM roofline of the conditioning upsampler (the prologue of the hot path, wnv_upsample): algorithmic bytes = the time-major output
(B T cin 4 bytes, written once) + the mel input; the bench batch (8 x 24064 samples) and a 32 x 96256 batch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests._configs import build, inputs

m = build("cfg2_mol").to("cuda")
eng = m._get_engine()
for B, T in ((8, 94 * 256), (32, 376 * 256)):
    c, _ = inputs("cfg2_mol", B, T)
    c = c.cuda()
    for _ in range(3):
        out = eng.upsample(c, T_expected=T)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 10
    ev[0].record()
    for _ in range(reps):
        out = eng.upsample(c, T_expected=T)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    alg = 4.0 * (out.numel() + c.numel())
    print(json.dumps({"workload": f"upsample {B} x {T} samples, 80 mel, x256", "ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1),
                      "frac_of_hbm_peak_8TBps": round(alg / ms / 1e6 / 8000.0, 4)}))
