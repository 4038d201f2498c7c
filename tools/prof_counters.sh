#!/bin/bash
# usage (on the GPU box): bash tools_prof.sh <tag> ; writes gpurun_out/pmc_<tag>_*.csv
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
