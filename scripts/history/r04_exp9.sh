#!/bin/bash
# prefetch of the next utterance's pre record in registers (wave 7): hit counters, A/B against HEAD, parity
set -u
OUT=gpurun_out/${1:-r04s}; mkdir -p $OUT
{
WNV_RING_PF_COUNT=1 WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_pfmark.so timeout 300 python scripts/exp_rate.py cfg2_mol 4096 48,64 pfmark
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 32,40,48,56,64 regpf
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_head.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 32,40,48,56,64 head
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 48,64 regpf
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_head.so timeout 400 python scripts/exp_rate.py cfg2_mol 8192 48,64 head
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_ring.py tests/test_gpu_parity_depth.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/pytest.txt
