#!/bin/bash
# Round 4: drain-free records (tagged granules both ways).  Parity of the ring suites, then the batch curve.
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/pytest.txt
{
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 1,8,16,32,48,64 records_v2
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8 records_v2
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16 records_v2
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
