#!/bin/bash
# round 5, GPU call 5: ubench variants 14 / 15; fine timelines of cfg4 (K = 512) and cfg1 (one-hot, K = 256); packed jobs with device-built maps
set -u
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
echo "== ubench_phase"; timeout 120 scripts/ubench_phase.bin 2>&1 | tee $OUT/ubench_phase.txt | tail -8
for cfg in cfg4_mol_multispeaker cfg1_mulaw256; do
  CFG=$cfg WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace.so timeout 200 python scripts/trace_ring.py $OUT/raw_$cfg.txt > $OUT/ring_$cfg.txt 2>&1
  python scripts/fine_trace.py $OUT/raw_$cfg.txt > $OUT/fine_$cfg.txt 2>&1; echo "== $cfg"; tail -30 $OUT/fine_$cfg.txt
done
echo "== packed tests + jobs"
timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_vs_reference.py -m gpu -x -q -k "packed" 2>&1 | tail -3
for args in "--workload cfg4_mol_multispeaker --job 128 --packed" "--workload cfg2_mol --job 100 --packed" "--workload cfg2_mol --job 200 --packed"; do
  timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee -a $OUT/jobs.txt
done
