#!/bin/bash
# scripts/ab_any.sh "<bench args>" lib1.so lib2.so ...   (one run each; prints the library, the batch and kSamples/s)
args=$1; shift
for lib in "$@"; do
  WNV_LIB=$PWD/$lib python bench.py $args --cpu-steps 0 --no-extras 2>gpurun_out/ab_err.txt | python -c 'import sys,json
L=sys.stdin.readlines()
print(sys.argv[1], (json.loads(L[-1])["config"]["batch_per_gpu"], json.loads(L[-1])["value"]) if L else "FAILED")' $lib
done
