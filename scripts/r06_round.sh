#!/bin/bash
# round 6: one GPU-box call -- the whole -m gpu suite, smoke, the default bench line, rocprofv3 kernel stats of the same command, the PMC passes
set -u
TAG=${1:-r06m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -25 > $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_default.log 2>&1; grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json; grep real $OUT/bench_default.log
python - <<PY
import json; d=json.load(open("$OUT/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("throughput_mode"), d.get("api_path"))
for j in d.get("strong_scaled_jobs") or []: print(j.get("workload","")[:30], j.get("kSamples_per_s"), j.get("wall_s"), j.get("padding_loss"), j.get("error"))
print(d["cpu_baseline"] if "cpu_baseline" in d else None)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o wnv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-extras > $ROOT/$OUT/prof_bench.log 2>&1 )
grep '^{' $OUT/prof_bench.log | tail -1 > $OUT/bench_under_rocprof.json
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -6 $f | cut -c1-200; done
bash scripts/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -45 $OUT/pmc.log | head -60
