#!/bin/bash
rocm-smi --showclkfrq 2>/dev/null | grep -v "^$\|====" | head -30
b() { python bench.py --steps 6 --warmup 1 --T 24064 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["value"])'; }
echo "auto: $(b)"
rocm-smi --setperflevel high 2>&1 | grep -v "^$\|====" | head -3
rocm-smi --showclocks 2>/dev/null | grep -i "clk" | head -6
echo "high: $(b)"
rocm-smi --setperflevel auto 2>&1 | grep -v "^$\|====" | head -2
echo "auto again: $(b)"
