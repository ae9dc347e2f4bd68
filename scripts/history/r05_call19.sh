#!/bin/bash
set -u
mkdir -p gpurun_out/r05s
timeout 600 python scripts/r05_logpick_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05s/logpick_check.txt
