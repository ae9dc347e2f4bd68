#!/bin/bash
# round 5, GPU call 18 (experiment): the log-domain pick in the PACKED one-hot instantiation too -- do the packed == padded tests still hold, what does the job gain
set -u
OUT=gpurun_out/r05r
mkdir -p $OUT
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_plog.so timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_vs_reference.py tests/test_gpu_postchain.py -m gpu -q -k "packed or classes" 2>&1 | tail -6 | tee $OUT/pytest.log
python - <<'PY' 2>&1 | tee $OUT/job_cfg1.txt
import os, subprocess, sys
code = r'''
import sys, time, torch
sys.path.insert(0, ".")
import bench
from types import SimpleNamespace
from tests._configs import CONFIGS, build
from wavenet_vocoder_amd import sharding
name = "cfg1_mulaw256"; kw = CONFIGS[name]
m = build(name).to("cuda")
frames, mels, _ = bench.job_inputs(SimpleNamespace(job=100), kw)
true = sum(f * 256 for f in frames)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=2, seed=3, as_index=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{true / dt / 1e3:.1f} kSamples/s true")
'''
for lib in ("libwnv_hip.so", "libwnv_plog.so", "libwnv_hip.so", "libwnv_plog.so"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WNV_LIB=os.path.join(os.getcwd(), "wavenet_vocoder_amd", lib)), capture_output=True, text=True)
    print(lib, "cfg1 job of 100 packed as classes:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
PY
