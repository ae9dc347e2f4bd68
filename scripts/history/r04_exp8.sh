#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04q}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_ring.py tests/test_gpu_parity_depth.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/pytest.txt
{
timeout 400 python scripts/exp_rate.py cfg2_mol 8192 8,32,40,48,56,64 ${2:-prefetch}
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 48,64 ${2:-prefetch}
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 32,48 ${2:-prefetch}
WNV_RING_MODE=1 timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 16,32 ${2:-prefetch}_mode1
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
timeout 600 python bench.py --job 100 --steps 1 --warmup 1 --packed 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('packed 100', j['value'], j['job']['padding_loss'])"
