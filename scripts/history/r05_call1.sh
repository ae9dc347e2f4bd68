#!/bin/bash
# round 5, GPU call 1: the reference on the box, the whole -m gpu suite (ABI 5, test library, packed speakers), smoke, the default
# bench line (cpu_baseline.kind = reference), the chain-phase microbenchmark, the cfg4 job lines.
set -u
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== oracle/_ref on the box"; ls -la oracle/_ref oracle/_ref/wavenet_vocoder 2>&1 | tail -12
echo "== ubench_phase"; timeout 120 scripts/ubench_phase.bin 2>&1 | tee $OUT/ubench_phase.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -45 | tee $OUT/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench default"
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-1500 $OUT/bench_default.json; tail -3 $OUT/bench.err
echo "== cfg4 jobs"
for args in "--job 128 --packed" "--job 128"; do
  timeout 600 python bench.py --workload cfg4_mol_multispeaker $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | tee -a $OUT/cfg4_jobs.json | cut -c1-900
done
echo "== cfg4 / cfg1 fixed batches"
for a in "cfg4_mol_multispeaker 8" "cfg4_mol_multispeaker 16" "cfg1_mulaw256 1" "cfg1_mulaw256 8"; do set -- $a
  timeout 300 python bench.py --workload $1 --batch $2 --T 8192 --steps 3 --warmup 1 --no-extras 2>>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 B=$2', j['value'], j['config']['kernel'])" | tee -a $OUT/configs.txt
done
