#!/bin/bash
# SQ counters of the f3 layer kernel (rocprofv3 --pmc in its own runs, --kernel-trace only): instruction mix and where the cycles go.
# usage: scripts/gpu_pmc_forward.sh <tag>     outputs -> gpurun_out/<tag>/pmc_fwd*/
set -u
TAG=${1:-pmcf}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_[A-Z0-9_]*\|SQ_ACTIVE_INST_[A-Z0-9_]*\|SQ_INST_CYCLES_[A-Z0-9_]*\|SQ_VALU_MFMA[A-Z0-9_]*" | sort -u > $OUT/avail.txt
run() {   # name, counters...
  local n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- python $ROOT/scripts/bench_forward.py > $OUT/$n.log 2>&1 || echo "$n failed: $(tail -2 $OUT/$n.log | head -1)"
}
run pmc_fwd SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32
run pmc_fwd2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run pmc_fwd3 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $ROOT
python - <<PY
import csv, glob, collections
for d in ("pmc_fwd", "pmc_fwd2", "pmc_fwd3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-40:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    for k, v in acc.items():
        if "fwd_layer" not in k: continue
        print(d, k, "dispatches", len(n[k]))
        for c, x in sorted(v.items()):
            print("   %-32s %.6g per dispatch" % (c, x / len(n[k])))
PY
