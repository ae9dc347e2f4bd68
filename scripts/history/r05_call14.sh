#!/bin/bash
# round 5, GPU call 14: packed slots with the speculative look in the tap workgroups (-DWNV_PACKED_SPEC=1: 28 register rows, 2-4 spilled registers) -- parity, then the jobs A/B
set -u
OUT=gpurun_out/r05n
mkdir -p $OUT
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_pspec.so timeout 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -3
for lib in libwnv_hip.so libwnv_pspec.so libwnv_hip.so libwnv_pspec.so; do
for args in "--workload cfg2_mol --job 100 --packed" "--workload cfg4_mol_multispeaker --job 128 --packed"; do
  WNV_LIB=$PWD/wavenet_vocoder_amd/$lib timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$lib $args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'])" | tee -a $OUT/jobs.txt
done; done
