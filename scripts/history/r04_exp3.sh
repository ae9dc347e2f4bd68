#!/bin/bash
# Round 4: software-pipelined tap passes.  Ring parity suites, then the batch curve, then the rings alone (no tap workgroups).
set -u
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/pytest.txt
{
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 1,8,32,40,48,56,64 ${2:-tap_v2}
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 8,32,64 ${2:-tap_v2}
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 16,32 ${2:-tap_v2}
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_nopre2.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8,32,48,64 rings_alone
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
