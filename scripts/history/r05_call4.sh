#!/bin/bash
# round 5, GPU call 4: fine timelines of the ring stages with and without WNV_PHASE2 (trace builds of the same sources)
set -u
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
for v in trace0 trace; do
  WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_$v.so timeout 200 python scripts/trace_ring.py $OUT/raw_$v.txt > $OUT/ring_$v.txt 2>&1
  python scripts/fine_trace.py $OUT/raw_$v.txt > $OUT/fine_$v.txt 2>&1; echo "== $v"; tail -24 $OUT/fine_$v.txt
done
