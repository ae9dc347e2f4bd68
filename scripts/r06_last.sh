#!/bin/bash
# round 6, last record: the whole GPU suite, smoke, the driver's bench command, and the numbers of the models that run the matrix-pipe tap units
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_last2_bench.json; python - <<'PY'
import json; d=json.load(open("gpurun_out/r06_last2_bench.json"))
print("headline", d["value"], d["ms_per_step"], "throughput_mode", d.get("throughput_mode"))
for j in d.get("strong_scaled_jobs") or []: print(j.get("workload","")[:40], j.get("kSamples_per_s"), j.get("padding_loss"))
PY
for W in cfg4_mol_multispeaker; do for r in 1 2 3; do python bench.py --workload $W --batch 8 --T 8192 --steps 2 --warmup 1 --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], 8, d["value"])' $W; done; done
for W in cfg3b_gaussian30 cfg1b_mulaw256_intree; do for B in 1 8 16 32 48 64; do python bench.py --workload $W --batch $B --T 8192 --steps 2 --warmup 1 --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], sys.argv[2], d["value"], d.get("api_path"))' $W $B; done; done
python bench.py --workload cfg3b_gaussian30 --job 64 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print("configs[3] job 64 packed", j["value"], j["job"]["padding_loss"])'
python bench.py --workload cfg3b_gaussian30 --job 64 --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print("configs[3] job 64 padded groups", j["value"], j["job"]["padding_loss"])'
