#!/bin/bash
# round 5, GPU call 10: classes-only output of one-hot models -- parity, the one-hot instantiations' A/B (their code moved), the packed job
set -u
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_postchain.py tests/test_gpu_vs_reference.py tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
for a in "--workload cfg1_mulaw256 --batch 1 --T 8192" "--workload cfg1_mulaw256 --batch 8 --T 8192" "--workload cfg1_mulaw256 --batch 48 --T 8192" "--workload cfg1b_mulaw256_intree --batch 8 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_prev.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_prev.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_cfg1.txt
done
python - <<'PY' 2>&1 | tee $OUT/job_cfg1.txt
# the 100-utterance job of the one-hot model through evaluate's own packed path (classes + post-chain per utterance) against padded groups
import sys, time, torch
sys.path.insert(0, ".")
import bench
from types import SimpleNamespace
from tests._configs import CONFIGS, build
from wavenet_vocoder_amd import sharding, synthesis
name = "cfg1_mulaw256"; kw = CONFIGS[name]
m = build(name).to("cuda"); m.rng = "philox"
args = SimpleNamespace(job=100)
frames, mels, _ = bench.job_inputs(args, kw)
true = sum(f * 256 for f in frames)
for label, kwargs in (("one-hot outputs", {}), ("classes (as_index)", {"as_index": True})):
    for rep in range(2):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=2, seed=3, stats=st, **kwargs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name} job of 100, packed, {label}: {true / dt / 1e3:.1f} kSamples/s true, launches {st['launches']}, padding {st['padding_loss']:.3f}")
    del outs
PY
