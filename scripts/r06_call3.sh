#!/bin/bash
# round 6, call 3: per-utterance stage timeline of the pair stages at B = 64 and 40 (trace build)
set -u
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
for B in 64 40; do
  B=$B WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace.so timeout 300 python scripts/trace_ring_batch.py $OUT/raw_pair_B$B.txt > $OUT/pair_timeline_B$B.txt 2>&1
  tail -40 $OUT/pair_timeline_B$B.txt
done
