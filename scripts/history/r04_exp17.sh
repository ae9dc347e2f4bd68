#!/bin/bash
# throughput instantiation: no store in wave 0 between prologue and chain poll / at the end of an utterance (hand-on by wave 1, filing by wave 5
# behind the chain-input barrier); tap workgroups: the wait for the gather in front of the first publish instead of behind the last.
# same-box A/B against the library of HEAD, then parity.
set -u
OUT=gpurun_out/${1:-r04ac}; mkdir -p $OUT
{
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8,32,40,48,56,64 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8,32,40,48,56,64 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 48,64 base
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 48,64 product
timeout 300 python scripts/exp_rate.py cfg3_gaussian 8192 48 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48,64 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48,64 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_packed.py tests/test_gpu_golden.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py tests/test_gpu_postchain.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/pytest.txt
