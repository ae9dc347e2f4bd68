#!/bin/bash
# round 5, GPU call 21 (the last one: 5 GPU-minutes left): the packed one-hot instantiations with the log-domain pick as the DEFAULT (the only two
# kernels whose code changed: wnv_ring_kernel<1,false,2>, <2,false,2>) -- every test that can launch a packed-slot kernel, the new check of the picks
# against the in-kernel noise restated on the host, the job number, smoke
set -u
OUT=gpurun_out/r05s
mkdir -p $OUT
timeout 130 python -m pytest tests/test_gpu_packed.py tests/test_gpu_postchain.py tests/test_gpu_vs_reference.py -m gpu -q -k "test_gpu_packed or test_gpu_postchain or packed" -s 2>&1 | tail -15 | tee $OUT/pytest.log
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 60 python - <<'PY' 2>&1 | tail -3 | tee $OUT/job_cfg1.txt
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
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=2, seed=3, as_index=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"cfg1 job of 100 packed as classes: {true / dt / 1e3:.1f} kSamples/s true", flush=True)
PY
