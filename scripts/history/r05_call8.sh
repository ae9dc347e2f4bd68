#!/bin/bash
# round 5, GPU call 8: the head's Gumbel-max butterfly as four v_max_f32_dpp (same-box A/B), then the whole suite on that library
set -u
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B head"
bash scripts/ab_bench.sh "--steps 5 --warmup 1" wavenet_vocoder_amd/libwnv_prev.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee $OUT/ab_head.txt
bash scripts/ab_any.sh "--workload cfg4_mol_multispeaker --batch 8 --T 8192 --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_prev.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_head.txt
bash scripts/ab_any.sh "--workload cfg3_gaussian --batch 8 --T 8192 --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_prev.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_head.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.log
