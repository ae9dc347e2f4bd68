#!/bin/bash
# Round 4, first GPU call: what bounds the ring -- the tap workgroups (waiting for pre / sharing the fabric) or the chain?
set -u
OUT=gpurun_out/r04a; mkdir -p $OUT
L=$PWD/wavenet_vocoder_amd
T=8192
{
echo "== product, default tap layout"
timeout 300 python scripts/exp_rate.py cfg2_mol $T 1,4,8,16,32,48,64 product
echo "== product, 2 tap workgroups per layer x 4 utterances per pass (B = 8)"
WNV_RING_TAP=2,4 timeout 200 python scripts/exp_rate.py cfg2_mol $T 8,16 product_tap2x4
echo "== product, B = 4: 1 part x 4 against 4 parts x 1 (fully decoupled rings)"
WNV_RING_TAP=1,4 timeout 200 python scripts/exp_rate.py cfg2_mol $T 4 product_tap1x4
WNV_RING_TAP=4,1 timeout 200 python scripts/exp_rate.py cfg2_mol $T 4 product_tap4x1
WNV_RING_TAP=2,2 timeout 200 python scripts/exp_rate.py cfg2_mol $T 4 product_tap2x2
echo "== experiment build 1: nobody waits for the tap workgroups' records (WRONG samples, timing only)"
WNV_LIB=$L/libwnv_nopre1.so timeout 300 python scripts/exp_rate.py cfg2_mol $T 1,8,32,48,64 nopre1
echo "== experiment build 2: ... and the tap workgroups exit at once"
WNV_LIB=$L/libwnv_nopre2.so timeout 300 python scripts/exp_rate.py cfg2_mol $T 1,8,32,48,64 nopre2
echo "== product again (drift check)"
timeout 200 python scripts/exp_rate.py cfg2_mol $T 8 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/exp1.txt
echo "== default bench line"
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-400 $OUT/bench_default.json
echo "== job mode"
timeout 600 python bench.py --job 40 --steps 1 --warmup 1 2>$OUT/job.err | tail -1 > $OUT/job40.json; cut -c1-1200 $OUT/job40.json
tail -3 $OUT/job.err
