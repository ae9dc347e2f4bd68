#!/bin/bash
# after the native Mersenne Twister (host code only): the replay-tape tests and the public path of the one-hot configurations
set -u
OUT=gpurun_out/${1:-r04fin3}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_zz_boundary.py tests/test_gpu_golden.py tests/test_gpu_ring.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/pytest.txt
for w in cfg1_mulaw256 cfg1b_mulaw256_intree cfg0_mulaw256_small cfg2_mol; do
python bench.py --workload $w --steps 2 --T 8192 --cpu-steps 0 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); api = d.get("api_path") or {}
print("%-26s B = %d  kernel %7.1f  incremental_forward %s" % (sys.argv[1], d["config"]["batch_per_gpu"], d["value"], api.get("kSamples_per_s_per_gpu")))' $w
done | tee $OUT/api.txt
python scripts/api_path_phases.py cfg1_mulaw256 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/phases.txt
