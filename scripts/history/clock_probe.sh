#!/bin/bash
# shader clock / power while the ring kernel runs (is the headline clock-limited by the power manager?)
rocm-smi --showperflevel 2>/dev/null | grep -i "level"
( python bench.py --steps 12 --warmup 1 --T 24064 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | cut -c1-120 ) &
BP=$!
for i in $(seq 1 40); do
  echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*level: //; s/.*(W): /P=/' | tr '\n' ' ')"
  kill -0 $BP 2>/dev/null || break
done
wait $BP
