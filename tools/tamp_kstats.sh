#!/bin/bash
# GPU box: rocprofv3 kernel stats of the TAMP-shaped batch (bash tools/tamp_kstats.sh [queries] [K]) -> gpurun_out/kstats_tamp.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ks_tamp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/tamp_probe.py ${1:-1024}:${2:-128} > $OUT/probe.txt 2> $OUT/trace.log
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $R/gpurun_out/kstats_tamp.csv
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - $R/gpurun_out/kstats_tamp.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-34s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"].split("(")[0].replace("void porrt::", "").replace("porrt::", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cat $OUT/probe.txt
rm -rf $OUT/trace
