#!/bin/bash
# usage (on the GPU box): bash tools/prof_workload.sh <tag> <workload> : kernel stats + four counter passes of
# `bench.py --workload <workload>`; every pass is a run of its own (rocprofv3 does not split counters over passes),
# counters only with --pmc (no trace domains).  Output under gpurun_out/<tag>_<workload>/.
TAG=$1; WL=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline --workload $WL > $OUT/stats.log 2>&1 || exit 2
find $OUT/stats -name "*_kernel_trace.csv" -delete   # (one row per dispatch: tens of MB; the stats summary is what is kept)
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python $R/bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --workload $WL > $OUT/pmc_$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES || exit 3
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS || exit 4
run fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 5
run write WRITE_SIZE || exit 6
echo "profiled $WL"
