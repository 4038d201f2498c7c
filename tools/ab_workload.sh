#!/bin/bash
# dev helper: one workload, several library builds.  usage: ab_workload.sh WORKLOAD lib1.so lib2.so ...
W=$1; shift
for L in "$@"; do
  UNICORN_RENDER_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --workload $W 2>/dev/null > gpurun_out/ab_tmp.json || exit 1
  python - "$L" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].split("/")[-1], d["config"]["workload"], d["value"], d["roofline"]["kernels_ms"])
PY
done
