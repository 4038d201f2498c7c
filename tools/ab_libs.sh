#!/bin/bash
# dev helper: A/B two builds of the library on ONE box, alternating runs.  usage: ab_libs.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    UNICORN_RENDER_LIB=$L timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/ab_tmp.json || exit 1
    python - "$L" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].split("/")[-1], d["value"], d["per_call_ms"])
PY
  done
done
