#!/bin/bash
# round 5, GPU call 13: L2 hit rate and fabric reads of the K = 512 ring at 1 / 8 / 32 utterances
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for B in 1 8 32; do
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_B$B -o p -- python $ROOT/bench.py --workload cfg4_mol_multispeaker --batch $B --T 8192 --steps 1 --warmup 1 --no-extras > $OUT/pmc_B$B.log 2>&1
  f=$(find $OUT/pmc_B$B -name '*counter_collection.csv' | head -1)
  echo "== cfg4 B=$B"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "wnv_ring_kernel" in r["Kernel_Name"]: d[r["Counter_Name"]]+=float(r["Counter_Value"])/2
for k in sorted(d): print(k, "%.4g per launch" % d[k])
if d.get("TCC_HIT_sum"): print("L2 hit rate %.4f" % (d["TCC_HIT_sum"]/(d["TCC_HIT_sum"]+d["TCC_MISS_sum"])))
PY
  tail -1 $OUT/pmc_B$B.log | cut -c1-120
done
