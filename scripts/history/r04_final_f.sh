#!/bin/bash
# BASELINE configs[0] through bench.py with the batch carried by the initial input (the public path ran one utterance before)
set -u
OUT=gpurun_out/${1:-r04fin5}; mkdir -p $OUT
for B in 8 1; do
python bench.py --workload cfg0_mulaw256_small --steps 2 --T 8192 --batch $B --cpu-steps 0 2>$OUT/cfg0_$B.err | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); api = d.get("api_path") or {}
print("cfg0_mulaw256_small        B = %d  kernel %7.1f  incremental_forward %s" % (d["config"]["batch_per_gpu"], d["value"], api))'
done | tee $OUT/cfg0.txt
