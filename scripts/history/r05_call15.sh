#!/bin/bash
# round 5, GPU call 15: the new evaluate tests (packed directory loop of a mu-law and a speaker-conditioned model); a second sample of the default line
set -u
OUT=gpurun_out/r05o
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_postchain.py tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
