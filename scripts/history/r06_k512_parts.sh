#!/bin/bash
# round 6: K = 512 (cfg4) and egs/mol beyond 32 utterances -- fewer rings, more tap workgroups?  (knob build; WNV_RING_KEEP_PARTS / WNV_RING_TAP)
OUT=gpurun_out/${1:-r06n}; mkdir -p $OUT
export WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_test.so
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --steps 2 --warmup 1 --cpu-steps 0 --no-extras 2>$OUT/err.txt | python -c 'import sys,json
L=[l for l in sys.stdin.readlines() if l.startswith("{")]
j=json.loads(L[-1]) if L else None
print(sys.argv[1], (j["config"]["batch_per_gpu"], j["value"], round(j["roofline"]["kernel_ms"]*1e3/j["config"]["T"],2)) if j else "FAILED")' "$label"
}
for B in 16 32 48 64; do
  run "cfg4 default      " -- --workload cfg4_mol_multispeaker --batch $B --T 8192
  run "cfg4 keep 2 parts " WNV_RING_KEEP_PARTS=1 -- --workload cfg4_mol_multispeaker --batch $B --T 8192
done
for B in 48 56 64; do
  run "egs/mol default   " -- --batch $B --T 8192
  run "egs/mol 3 parts   " WNV_RING_TAP=3,8 -- --batch $B --T 8192
done
