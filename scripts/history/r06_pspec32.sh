#!/bin/bash
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vPS.so
WNV_LIB=$PWD/$Z timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_vs_reference.py -x -q -k "packed" 2>&1 | tail -2
for lib in $A $Z $A $Z; do
  for spec in "cfg2_mol 100" "cfg4_mol_multispeaker 128"; do set -- $spec
    echo -n "$lib packed job $1 $2: "; WNV_LIB=$PWD/$lib python bench.py --workload $1 --job $2 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
  done
done
