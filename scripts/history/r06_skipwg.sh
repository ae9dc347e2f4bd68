#!/bin/bash
# round 6: K = 512 with SKIP workgroups -- parity first, then the rates (knob build: WNV_RING_SKIPWG=0 is the round-5 layout)
OUT=gpurun_out/${1:-r06o}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_vs_reference.py -x -q -k "throughput_instantiation and cfg4" 2>&1 | tail -25 | cut -c1-400
export WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_test.so
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" --steps 2 --warmup 1 --cpu-steps 0 --no-extras 2>$OUT/err.txt | python -c 'import sys,json
L=[l for l in sys.stdin.readlines() if l.startswith("{")]
j=json.loads(L[-1]) if L else None
print(sys.argv[1], (j["config"]["batch_per_gpu"], j["value"], round(j["roofline"]["kernel_ms"]*1e3/j["config"]["T"],2)) if j else "FAILED")' "$label"
  [ -s $OUT/err.txt ] && tail -2 $OUT/err.txt
}
for B in 16 24 32 40 48 56 64; do
  run "cfg4 round-5 layout " WNV_RING_SKIPWG=0 -- --workload cfg4_mol_multispeaker --batch $B --T 8192
  run "cfg4 skip workgroups" WNV_RING_SKIPWG=1 -- --workload cfg4_mol_multispeaker --batch $B --T 8192
done
