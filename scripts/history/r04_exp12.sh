#!/bin/bash
# item 5: categorical head with the softmax spread over the workgroup -- parity, rates, timeline
set -u
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_ring.py tests/test_gpu_stress.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_wide.py tests/test_gpu_parity_depth.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/pytest.txt
{
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8,16,32 fastcat
timeout 300 python scripts/exp_rate.py cfg1b_mulaw256_intree 8192 1,8 fastcat
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
export WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace.so
for c in "cfg1_mulaw256 1" "cfg1_mulaw256 8"; do
  set -- $c
  CFG=$1 B=$2 timeout 200 python scripts/trace_ring.py $OUT/raw_$1_$2.txt > $OUT/ring_$1_$2.txt 2>&1
  python scripts/fine_trace.py $OUT/raw_$1_$2.txt > $OUT/fine_$1_$2.txt 2>&1
  echo "== $1 B=$2"; grep -A40 "^means" $OUT/fine_$1_$2.txt | grep "head\|step"
done
