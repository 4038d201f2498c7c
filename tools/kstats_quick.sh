#!/bin/bash
# dev helper (GPU box): per-kernel average durations of one bench workload under rocprofv3.  usage: kstats_quick.sh <workload> [lib.so]
WL=$1; R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export UNICORN_RENDER_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksq && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksq -- python $R/bench.py --no-cpu-baseline --workload $WL > /tmp/ksq.log 2>&1 || { tail -5 /tmp/ksq.log; exit 1; }
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ksq/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.reader(open(f)))[1:]:
    if r[0].startswith(("(anonymous namespace)::k_", "k_")) or "k_" in r[0][:40]:
        print("%-60s calls %5s avg %9.1f us" % (r[0][:60], r[1], float(r[3]) / 1e3))
PY
