#!/bin/bash
# SQ counters of the f3 layer kernel (rocprofv3 --pmc in its own run, --kernel-trace only): where do the wave cycles go?
# usage: scripts/gpu_pmc_forward.sh <tag>     outputs -> gpurun_out/<tag>/pmc_fwd/
set -u
TAG=${1:-pmcf}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 \
    --kernel-trace --output-format csv -d $OUT/pmc_fwd -o p -- python $ROOT/scripts/bench_forward.py > $OUT/pmc_fwd.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_fwd2 -o p -- python $ROOT/scripts/bench_forward.py > $OUT/pmc_fwd2.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for d in ("pmc_fwd", "pmc_fwd2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-40:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    for k, v in acc.items():
        if "fwd_layer" not in k and "fwd_head" not in k: continue
        print(d, k, "dispatches", len(n[k]))
        for c, x in sorted(v.items()):
            print("   %-32s %.6g per dispatch" % (c, x / len(n[k])))
PY
