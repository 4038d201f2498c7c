#!/bin/bash
# usage (on the GPU box): bash tools/round_profile.sh <tag>: bench line, kernel stats, PMC passes, other workloads
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${TAG}.log 2>&1 || exit 2
bash $R/tools/prof_counters.sh $TAG || exit 3
for w in C2_B32_64x64_K50_ico3 C2p_B192_64x64_K50_ico3 C4b_B64_128x128_K50_ico4 C5_B16_256x256_K100_ico5; do
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline --workload $w >> $R/gpurun_out/${TAG}_workloads.jsonl 2>> $R/gpurun_out/${TAG}_workloads.err || exit 4
done
