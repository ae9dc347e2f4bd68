#!/bin/bash
# throughput vs batch on the ring kernel (bench.py --batch B --T 8192 --no-extras): kSamples/s per GPU
for B in 8 16 32 48 64; do
  python bench.py --batch $B --T 8192 --steps 3 --warmup 1 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('B =', j['config']['batch_per_gpu'], ' kSamples/s', j['value'], ' per-utterance x real time @24k', j['rtf_24k'])"
done
