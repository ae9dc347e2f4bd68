#!/bin/bash
# round 6: the round's tap-workgroup changes (deferred publish parked in LDS, ring-slot mask) and the exact-uniform heads under RED ZONES
# (-DWNV_GUARD: every device buffer the sample-loop kernels write sits between two guard regions checked after every launch):
# the throughput / packed / determinism / ring suites and the stress script on the guarded library.
OUT=gpurun_out/r06_guard; mkdir -p $OUT
export WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_guard.so
timeout 2400 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_packed.py tests/test_gpu_seed_determinism.py tests/test_gpu_ring.py tests/test_gpu_inkernel_noise.py -x -q 2>$OUT/pytest_stderr.txt | tail -3 | tee $OUT/pytest_tail.txt
echo "guard banners: $(grep -c 'wnv guard. red zones of' $OUT/pytest_stderr.txt); red-zone violations: $(grep -c overwritten $OUT/pytest_stderr.txt)" | tee -a $OUT/pytest_tail.txt
timeout 1200 python scripts/stress_ring.py 2>$OUT/stress_stderr.txt | tee $OUT/stress.txt
echo "stress: guard banners $(grep -c 'wnv guard. red zones of' $OUT/stress_stderr.txt); violations $(grep -c overwritten $OUT/stress_stderr.txt)" | tee -a $OUT/stress.txt
