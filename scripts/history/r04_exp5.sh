#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04j}; mkdir -p $OUT
L=$PWD/wavenet_vocoder_amd
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/pytest.txt
{
WNV_LIB=$L/libwnv_prev.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8,32,48,64 prev_commit
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 1,8,32,40,48,56,64 ${2:-prologue1rtt}
WNV_LIB=$L/libwnv_prev.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8 prev_commit
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8 ${2:-prologue1rtt}
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8,48,64 ${2:-prologue1rtt}
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16,32,48 ${2:-prologue1rtt}
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
