#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04l}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/pytest.txt
{
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8,16,32,48,64 packed_build
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
for J in 40 100 200; do
timeout 600 python bench.py --job $J --steps 1 --warmup 1 2>$OUT/job.err | tail -1 > $OUT/job${J}_padded.json; python -c "import json; j=json.load(open('$OUT/job${J}_padded.json')); print('padded', $J, j['value'], j['job']['padding_loss'], j['job']['rank0_launches_B_x_T'])"
timeout 600 python bench.py --job $J --steps 1 --warmup 1 --packed 2>$OUT/jobp.err | tail -1 > $OUT/job${J}_packed.json; python -c "import json; j=json.load(open('$OUT/job${J}_packed.json')); print('packed', $J, j['value'], j['job']['padding_loss'], j['job']['rank0_launches_B_x_T'])" || tail -5 $OUT/jobp.err
done
