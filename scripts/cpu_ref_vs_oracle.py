#!/usr/bin/env python3
"""How fast is the CPU oracle relative to the REAL reference?  (authoring container only: /root/reference does not travel)

bench.py's cpu_baseline times oracle/wavenet_oracle.py on the GPU box's host cores because the reference package is not there
(kind "port").  This script times both on the same machine, same weights / mel / noise, so the bench line can say what the
reference itself would have measured:  reference kSamples/s ~= oracle kSamples/s / ratio.

    python scripts/cpu_ref_vs_oracle.py [T]      -> profiles/cpu_ref_ratio.json + a text report on stdout
"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
import torch  # noqa: E402

import wavenet_vocoder as ref  # noqa: E402
from oracle.wavenet_oracle import Oracle  # noqa: E402
from tests._configs import CONFIGS, build, inputs  # noqa: E402
from tests._golden import oracle_config  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name, B = "cfg2_mol", 8
kw = CONFIGS[name]
ours = build(name)
rm = ref.WaveNet(**kw).eval()
rm.make_generation_fast_()
rm.load_state_dict(ours.state_dict())
o = Oracle(oracle_config(kw), ours.state_dict())
c, _ = inputs(name, B, T)
res = {}
print(f"{name}, B = {B}, T = {T} steps, host has {os.cpu_count()} cores")
for threads in (1, 4):
    torch.set_num_threads(threads)
    with torch.no_grad():
        rm.incremental_forward(c=c[:, :, :5], T=256)                    # warm-up
        t0 = time.perf_counter(); torch.manual_seed(0)
        y_ref = rm.incremental_forward(c=c, T=T, softmax=True, quantize=True, log_scale_min=-16.0)
        t_ref = time.perf_counter() - t0
        o.incremental_forward(c=c[:, :, :5], T=256, noise=torch.rand(256, B, 11) * 0.9 + 0.05)
        tape = torch.rand(T, B, 11) * 0.9 + 0.05
        t0 = time.perf_counter()
        o.incremental_forward(c=c, T=T, noise=tape)
        t_or = time.perf_counter() - t0
    res[threads] = dict(reference_kSamples_s=B * T / t_ref / 1e3, oracle_kSamples_s=B * T / t_or / 1e3, oracle_over_reference=t_ref / t_or)
    print(f"  {threads} thread(s): reference {res[threads]['reference_kSamples_s']:.3f} kSamples/s, oracle {res[threads]['oracle_kSamples_s']:.3f} kSamples/s "
          f"-> the oracle is {res[threads]['oracle_over_reference']:.2f}x the reference's speed")
json.dump({"workload": f"{name} B={B} T={T}", "host_cores": os.cpu_count(), "by_threads": {str(k): v for k, v in res.items()},
           "note": "measured in the authoring container (the reference cannot travel to the GPU box); divide bench.py's cpu_baseline by "
                   "oracle_over_reference to estimate what the reference itself would measure"},
          open(os.path.join(ROOT, "profiles", "cpu_ref_ratio.json"), "w"), indent=1)
