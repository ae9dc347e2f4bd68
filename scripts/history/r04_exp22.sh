#!/bin/bash
# shader clock and power while the ring kernel runs 8 / 48 / 64 utterances per GPU (do the stages' compute phases stretch because the clock drops?)
set -u
OUT=gpurun_out/${1:-r04ah}; mkdir -p $OUT
for B in 8 48 64; do
  steps=$(( B == 8 ? 60 : 40 ))
  ( python bench.py --batch $B --steps $steps --warmup 1 --T 8192 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("   bench:", d["value"], "kSamples/s")' ) &
  BP=$!
  sleep 4
  echo "== B = $B"
  for i in $(seq 1 8); do
    kill -0 $BP 2>/dev/null || break
    echo "   $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power" | sed 's/^GPU\[0\][^:]*: //' | tr '\n' '|' | cut -c1-200)"
    sleep 0.3
  done
  wait $BP
done 2>&1 | tee $OUT/clocks.txt
