#!/bin/bash
# Same-box A/B of library builds on the headline workload: every library is benchmarked ROUNDS times, interleaved.
#   scripts/ab_bench.sh "<bench args>" libA.so libB.so ...        (kSamples/s per run)
args=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    v=$(WNV_LIB=$PWD/$lib python bench.py $args --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["value"])')
    echo "$lib round $r: $v"
  done
done
