#!/bin/bash
# round 5, the final evidence on the final library: the default bench line, rocprofv3 kernel stats of the same command, the PMC passes,
# the numbers of every BASELINE configuration.  Outputs -> gpurun_out/r05final/ (copied to profiles/r05_final_* afterwards).
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r05final
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench default"
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-400 $OUT/bench_default.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o wnv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-extras > $OUT/bench_under_rocprof.log 2>&1 )
grep '^{' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json; cut -c1-200 $OUT/bench_under_rocprof.json
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -6 $f | cut -c1-200; done
echo "== PMC"
bash scripts/gpu_pmc.sh r05final/pmc > $OUT/pmc.log 2>&1; tail -25 $OUT/pmc.log
echo "== numbers"
bash scripts/gpu_final_numbers.sh 2>&1 | tee $OUT/final_numbers.txt
