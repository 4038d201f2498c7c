#!/bin/bash
# dev helper: one library, several values of a dev environment switch.  usage: ab_env.sh VAR rounds v1 v2 ...
V=$1; R=$2; shift; shift
for i in $(seq $R); do
  for x in "$@"; do
    env $V=$x timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/ab_tmp.json || exit 1
    python - "$V=$x" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], d["value"], d["roofline"]["kernels_ms"])
PY
  done
done
