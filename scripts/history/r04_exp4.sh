#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04i}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/pytest.txt
{
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8,32,40,48,56,64 ${2:-extra_d1}
WNV_RING_TAP_EXTRA=0 timeout 400 python scripts/exp_rate.py cfg2_mol 8192 48,64 no_extra
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 48,64 ${2:-extra_d1}
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 32,48 ${2:-extra_d1}
timeout 300 python scripts/exp_rate.py cfg3_gaussian 8192 8,48 ${2:-extra_d1}
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
