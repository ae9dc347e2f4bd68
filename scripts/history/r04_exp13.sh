#!/bin/bash
# item 5: K = 512 kernel with the reserved-register polls -- parity, rates
set -u
OUT=gpurun_out/${1:-r04x}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_ring.py tests/test_gpu_stress.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_parity_depth.py tests/test_gpu_packed.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/pytest.txt
{
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 1,8,16,32 rp
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8 rp
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8,48 rp
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
