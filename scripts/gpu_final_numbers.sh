#!/bin/bash
# kSamples/s of every BASELINE configuration (both variants where BASELINE.json and the in-tree presets differ) on the final library, one
# GPU-box call: kernel rate (in-kernel Philox) and the public WaveNet.incremental_forward path (rng = "replay"), B = 8; single
# utterance; the batch curve; the job mode (padded groups against packed slots); the wide model.
# usage: scripts/gpu_final_numbers.sh > gpurun_out/final_numbers.txt
for w in cfg2_mol cfg1_mulaw256 cfg1b_mulaw256_intree cfg3_gaussian cfg3b_gaussian30 cfg4_mol_multispeaker cfg0_mulaw256_small; do
  python bench.py --workload $w --steps 2 --T 8192 --cpu-steps 0 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
api = d.get("api_path") or {}
print("%-26s B = %d  kernel %7.1f  incremental_forward %s  (%s)" % (sys.argv[1], d["config"]["batch_per_gpu"], d["value"], api.get("kSamples_per_s_per_gpu"), d["config"]["kernel"]))' $w
done
for w in cfg2_mol cfg1_mulaw256 cfg1b_mulaw256_intree cfg3b_gaussian30 cfg4_mol_multispeaker; do
  python bench.py --workload $w --steps 2 --T 8192 --batch 1 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print("%-26s B = 1  kernel %7.1f  (%.2f x real time at 24 kHz)" % (sys.argv[1], d["value"], d["value"] / 24.0))' $w
done
for w in cfg2_mol cfg1_mulaw256 cfg4_mol_multispeaker; do
for B in 16 32 48 64; do
  python bench.py --workload $w --steps 2 --T 8192 --batch $B --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print("%-26s B = %d  kernel %7.1f  (%.2f x real time at 24 kHz per utterance)" % (sys.argv[1], d["config"]["batch_per_gpu"], d["value"], d["value"] / d["config"]["batch_per_gpu"] / 24.0))' $w
done
done
# BASELINE configs[3] / configs[4] as jobs on one GPU: 64 utterances of the 30-layer Gaussian, 128 of the speaker-conditioned model (packed slots
# with a speaker per utterance since round 5)
for spec in "cfg3b_gaussian30 64" "cfg4_mol_multispeaker 128"; do set -- $spec
  for M in "" "--packed"; do
  python bench.py --workload $1 --job $2 --steps 1 --warmup 1 $M 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); j = d["job"]
print("%-22s job of %3d utterances %-7s: %7.1f kSamples/s true (%7.1f incl. padding), padding %4.1f %%, launches %s" % (sys.argv[2], j["utterances"], sys.argv[1] or "padded", d["value"], j["kSamples_per_s_incl_padding"], 100 * j["padding_loss"], j["rank0_launches_B_x_T"][:4]))' "$M" $1
  done
done
for J in 40 100 200; do
  for M in "" "--packed"; do
  python bench.py --job $J --steps 1 --warmup 1 $M 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); j = d["job"]
print("job of %3d utterances (1-8 s, %.0f s of audio) %-7s: %7.1f kSamples/s true, padding %4.1f %%, %5.1f x real time for the whole job, launches %s" % (j["utterances"], j["true_samples"] / 24000.0, sys.argv[1] or "padded", d["value"], 100 * j["padding_loss"], j["x_real_time_24k_whole_job"], j["rank0_launches_B_x_T"]))' "$M"
  done
done
python scripts/wide_rate.py 2>/dev/null | tail -4
