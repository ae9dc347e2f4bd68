#!/bin/bash
# PMC passes for the sample-loop kernel (each counter group in its own rocprofv3 run, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes): HBM/fabric traffic (FETCH_SIZE, WRITE_SIZE), L2 hit rate, LDS activity, and (round 6) the issue side.
# usage: scripts/gpu_pmc.sh <tag> [bench args]     outputs -> gpurun_out/<tag>/pmc_*/
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- \
      python $ROOT/bench.py --steps 1 --warmup 1 --no-extras ${BENCH_ARGS:-} > $OUT/pmc_$name.log 2>&1
  echo "== $name: $(grep -c . $(find $OUT/pmc_$name -name '*counter_collection.csv' | head -1) 2>/dev/null) rows"
}
BENCH_ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
# round 6: the issue side -- where wave time goes: VALU issued / VALU busy, scalar and vector-memory instructions (the polls),
# cycles a wave waits for any instruction (s_waitcnt, s_barrier, s_sleep)
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
run vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT
run wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
cd $ROOT
python scripts/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt
