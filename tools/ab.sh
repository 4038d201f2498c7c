#!/bin/bash
# dev helper: time several builds of the library on ONE box, alternating.  usage: ab.sh rounds workload lib1.so lib2.so ...
R=$1; W=$2; shift; shift
for i in $(seq $R); do
  for L in "$@"; do
    UNICORN_RENDER_LIB=$PWD/$L timeout -k 10 200 python bench.py --no-cpu-baseline --workload $W 2>/dev/null > gpurun_out/ab_tmp.json || { echo "$L failed"; continue; }
    python - "$L" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].split("/")[-1].ljust(16), round(d["value"]), {k["kernel"]: k["kernel_ms"] for k in d["roofline"]["kernels"]}, d["per_call_ms"], "eager", round(d["eager_images_per_s"]))
PY
  done
done
