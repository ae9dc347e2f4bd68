#!/bin/bash
# what bounds 40-64 utterances per GPU with the final stages: tap passes per workgroup (quantised in pairs of utterances per ring)?
set -u
OUT=gpurun_out/${1:-r04t}; mkdir -p $OUT
{
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 40,48,56,64 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_nopre2.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 32,40,48,56,64 nopre2
WNV_RING_TAP=3,8 timeout 300 python scripts/exp_rate.py cfg2_mol 8192 42,48,56 tap3x8
WNV_RING_TAP=1,8 timeout 300 python scripts/exp_rate.py cfg2_mol 8192 32,48,64 tap1x8
WNV_RING_TAP=2,4 timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48,64 tap2x4
WNV_RING_TAP=4,4 timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48,64 tap4x4
WNV_RING_TAP=4,8 timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48 tap4x8
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
{
B=64 WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace.so timeout 200 python scripts/trace_tap.py $OUT/tap6_raw.txt
B=64 WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace3.so timeout 200 python scripts/trace_tap.py $OUT/tap3_raw.txt
B=48 WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace3.so timeout 200 python scripts/trace_tap.py $OUT/tap3_48_raw.txt
} 2>&1 | grep -v amdgpu.ids | tee $OUT/taps.txt
