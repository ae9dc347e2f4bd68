#!/bin/bash
# after the bench.py change (a workload without conditioning): BASELINE configs[0] through bench.py, and the default line once more
set -u
OUT=gpurun_out/${1:-r04fin2}; mkdir -p $OUT
for B in 8 1; do
python bench.py --workload cfg0_mulaw256_small --steps 2 --T 8192 --batch $B --cpu-steps 0 2>$OUT/cfg0_$B.err | tail -1 > $OUT/cfg0_$B.json
python - $OUT/cfg0_$B.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).readline())
    api = d.get("api_path") or {}
    print("%-26s B = %d  kernel %7.1f  incremental_forward %s  (%s)" % ("cfg0_mulaw256_small", d["config"]["batch_per_gpu"], d["value"], api.get("kSamples_per_s_per_gpu"), d["config"]["kernel"]))
except Exception as e:
    print("cfg0 failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
python bench.py --workload cfg0_mulaw256_small --steps 1 --T 2048 --batch 1 --cpu-steps 1024 --no-extras 2>/dev/null | tail -1 | cut -c1-400
python bench.py --workload cfg1_mulaw256 --steps 2 --T 8192 --cpu-steps 0 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline()); api = d.get("api_path") or {}
print("cfg1_mulaw256 again: kernel %.1f incremental_forward %s" % (d["value"], api))'
echo "== default bench line"
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-200 $OUT/bench_default.json
