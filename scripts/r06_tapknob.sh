L=$PWD/wavenet_vocoder_amd/libwnv_test.so
for r in 1 2 3; do
  for K in "" "2,4"; do
    v=$(WNV_LIB=$L WNV_RING_TAP=$K python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["value"])')
    echo "B=8 WNV_RING_TAP='$K' run $r: $v"
  done
done
for K in "" "2,4"; do
  v=$(WNV_LIB=$L WNV_RING_TAP=$K python bench.py --batch 4 --T 8192 --steps 2 --warmup 1 --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["value"])')
  echo "B=4 WNV_RING_TAP='$K': $v"
done
