#!/bin/bash
# round 5, GPU call 6: A/B of the addends-as-accumulator-init form (WNV_PHASE2_ZACC); instruction-cache counters of the K = 512 ring against egs/mol
set -u
OUT=$PWD/gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B zacc"
bash scripts/ab_bench.sh "--steps 5 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_zacc.so 2>&1 | tee $OUT/ab_zacc.txt
echo "== counters available"
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|SQ_INSTS_VALU\b|SQ_WAIT_INST|SQ_IFETCH|SQ_INST_LEVEL" | head -20) | tee $OUT/avail.txt
ROOT=$PWD
cd /tmp
for wl in cfg4_mol_multispeaker cfg2_mol; do
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$tag -o p -- python $ROOT/bench.py --workload $wl --T 8192 --steps 1 --warmup 1 --no-extras > $OUT/pmc_${wl}_$tag.log 2>&1
    f=$(find $OUT/pmc_${wl}_$tag -name '*counter_collection.csv' | head -1)
    echo "== $wl $grp"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "wnv_ring_kernel" in r["Kernel_Name"]:
        d[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in d: print(k, d[k], "rows", n[k])
PY
    tail -2 $OUT/pmc_${wl}_$tag.log | cut -c1-200
  done
done
