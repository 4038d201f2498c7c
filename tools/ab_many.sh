#!/bin/bash
# dev helper: time several builds of the library on ONE box, alternating.  usage: ab_many.sh rounds lib1.so lib2.so ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    UNICORN_RENDER_LIB=$L timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/ab_tmp.json || exit 1
    python - "$L" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].split("/")[-1], d["value"], d["roofline"]["kernels_ms"], d["per_call_ms"])
PY
  done
done
