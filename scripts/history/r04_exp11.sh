#!/bin/bash
# item 5: where do the one-hot (cfg1) and K = 512 (cfg4) rings spend their step?
set -u
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
export WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_trace.so
for c in "cfg1_mulaw256 1" "cfg1_mulaw256 8" "cfg4_mol_multispeaker 8" "cfg4_mol_multispeaker 16" "cfg2_mol 1"; do
  set -- $c
  CFG=$1 B=$2 timeout 200 python scripts/trace_ring.py $OUT/raw_$1_$2.txt > $OUT/ring_$1_$2.txt 2>&1
  python scripts/fine_trace.py $OUT/raw_$1_$2.txt > $OUT/fine_$1_$2.txt 2>&1
  echo "== $1 B=$2"; grep -A40 "^means" $OUT/fine_$1_$2.txt
done
