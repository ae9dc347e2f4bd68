#!/bin/bash
# Round-end evidence for the headline kernel, in one GPU-box call: default bench line, rocprofv3 kernel stats of the same command,
# PMC passes (HBM traffic, L2 hit rate, LDS activity), fine timeline (trace build).   usage: scripts/gpu_profile_round.sh <tag>
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== default bench"; timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o wnv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-extras > $OUT/bench_under_rocprof.json 2>$OUT/prof.err )
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -8 $f | cut -c1-200; done
tail -1 $OUT/bench_under_rocprof.json | cut -c1-200
echo "== PMC"; bash scripts/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -25 $OUT/pmc.log
if [ -f wavenet_vocoder_amd/libwnv_trace.so ]; then
  echo "== fine timeline"; WNV_LIB=$ROOT/wavenet_vocoder_amd/libwnv_trace.so python scripts/trace_ring.py $OUT/trace_raw.txt > $OUT/trace.txt 2>&1; python scripts/fine_trace.py $OUT/trace_raw.txt > $OUT/fine_timeline.txt; tail -26 $OUT/fine_timeline.txt
fi
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_l2 $OUT/pmc_lds 2>/dev/null
