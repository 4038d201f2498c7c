#!/bin/bash
# dev helper: run bench.py under several UR_DEBUG ablation values on ONE box; prints value + per-call ms
for d in "$@"; do
  UR_DEBUG=$d timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/ab_$d.json || exit 1
  python - "$d" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d["per_call_ms"])
PY
done
