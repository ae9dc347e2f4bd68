#!/bin/bash
# round 5, GPU call 17: the whole suite on the final library (log-domain pick in the one-hot throughput instantiation only); cfg1 at 40 / 48 / 64
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for B in 8 40 48 64; do bash scripts/ab_any.sh "--workload cfg1_mulaw256 --batch $B --T 8192 --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/cfg1.txt; done
