#!/bin/bash
# round 5, GPU call 22: the new check of the in-kernel noise mode (what bench.py times) -- every sample of a launch against the oracle's sampler applied
# to the kernel's own head outputs and the host-restated Philox stream (tests/test_gpu_inkernel_noise.py)
set -u
OUT=gpurun_out/r05t
mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_inkernel_noise.py -m gpu -q -s ${WNV_K:+-k "$WNV_K"} 2>&1 | tail -25 | tee $OUT/pytest.log
