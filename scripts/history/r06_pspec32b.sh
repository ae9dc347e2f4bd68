#!/bin/bash
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vPS.so
for lib in $A $Z $A $Z $A $Z; do
  for spec in "cfg2_mol 100" "cfg2_mol 200" "cfg3b_gaussian30 64" "cfg3_gaussian 100"; do set -- $spec
    echo -n "$lib packed job $1 $2: "; WNV_LIB=$PWD/$lib python bench.py --workload $1 --job $2 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
  done
done
