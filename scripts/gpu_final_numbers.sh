#!/bin/bash
# kSamples/s of every BASELINE configuration on the final library, one GPU-box call: kernel rate (in-kernel Philox) and the public
# WaveNet.incremental_forward path (rng = "replay"), B = 8; single utterance for the two recipe configurations.
# usage: scripts/gpu_final_numbers.sh > gpurun_out/final_numbers.txt
for w in cfg2_mol cfg1_mulaw256 cfg3_gaussian cfg4_mol_multispeaker cfg1b_mulaw256_intree cfg0_mulaw256_small; do
  python bench.py --workload $w --steps 2 --T 8192 --cpu-steps 0 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
api = d.get("api_path") or {}
print("%-26s B = %d  kernel %7.1f  incremental_forward %s  (%s)" % (sys.argv[1], d["config"]["batch_per_gpu"], d["value"], api.get("kSamples_per_s_per_gpu"), d["config"]["kernel"]))' $w
done
for w in cfg2_mol cfg1_mulaw256; do
  python bench.py --workload $w --steps 2 --T 8192 --batch 1 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print("%-26s B = 1  kernel %7.1f  (%.2f x real time at 24 kHz)" % (sys.argv[1], d["value"], d["value"] / 24.0))' $w
done
for B in 16 32 64; do
  python bench.py --workload cfg2_mol --steps 2 --T 8192 --batch $B --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print("cfg2_mol                   B = %d  kernel %7.1f" % (d["config"]["batch_per_gpu"], d["value"]))'
done
