#!/bin/bash
# round 6, last call: the job lines after the byte bound of a packed launch went to 32 GiB (egs/mol 100 / 200 utterances, the mu-law model as classes)
for J in 100 200; do for M in "" "--packed"; do
  python bench.py --job $J --steps 1 --warmup 1 $M 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); j = d["job"]
print("egs/mol job of %3d utterances %-8s: %7.1f kSamples/s true, padding %4.1f %%, launches %s" % (j["utterances"], sys.argv[1] or "padded", d["value"], 100 * j["padding_loss"], j["rank0_launches_B_x_T"]))' "$M"
done; done
for M in "" "--packed"; do
  python bench.py --workload cfg1_mulaw256 --job 100 --steps 1 --warmup 1 $M 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); j = d["job"]
print("mu-law job of %3d utterances %-8s: %7.1f kSamples/s true, padding %4.1f %%, launches %s" % (j["utterances"], sys.argv[1] or "padded", d["value"], 100 * j["padding_loss"], j["rank0_launches_B_x_T"]))' "$M"
done
