#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04k}; mkdir -p $OUT
L=$PWD/wavenet_vocoder_amd
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_zz_boundary.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/pytest.txt
{
WNV_LIB=$L/libwnv_prev.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8,48 prev_commit
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 1,8,16,32,40,48,56,64 multi_variant
WNV_LIB=$L/libwnv_prev.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8 prev_commit
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8 multi_variant
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
bash scripts/guard_runs.sh 34 $OUT/guard_runs.txt
echo "== default bench line"
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
echo "== job mode"
timeout 600 python bench.py --job 40 --steps 1 --warmup 1 2>$OUT/job.err | tail -1 > $OUT/job40.json; python -c "import json; j=json.load(open('$OUT/job40.json')); print(j['value'], j['job'])"
timeout 600 python bench.py --job 100 --steps 1 --warmup 1 2>$OUT/job.err | tail -1 > $OUT/job100.json; python -c "import json; j=json.load(open('$OUT/job100.json')); print(j['value'], j['job'])"
