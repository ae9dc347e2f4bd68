#!/bin/bash
# instruction-cache behaviour of the ring kernel at 8 / 48 / 64 utterances per GPU (one PMC pass each; rocprofv3 --pmc with --kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r04ab}; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for B in 8 48 64; do
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d $OUT/ic_$B -o p -- \
      python $ROOT/bench.py --batch $B --T 8192 --steps 1 --warmup 1 --no-extras --cpu-steps 0 > $OUT/ic_$B.log 2>&1
  f=$(find $OUT/ic_$B -name '*counter_collection.csv' | head -1)
  echo "== B = $B ($(tail -1 $OUT/ic_$B.log | cut -c1-120))"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    if 'wnv_ring_kernel' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(acc): print(f"   {k:22s} {acc[k]:16.0f}  ({n[k]} rows)")
if acc.get('SQC_ICACHE_REQ'): print(f"   hit rate {acc['SQC_ICACHE_HITS'] / acc['SQC_ICACHE_REQ']:.4f}   misses per step and workgroup (T = 8192, 2 launches, ~248 workgroups): {acc['SQC_ICACHE_MISSES'] / 2 / 8192 / 248:.2f}")
PY
  rm -rf $OUT/ic_$B
done
