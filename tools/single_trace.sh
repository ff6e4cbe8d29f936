#!/bin/bash
# developer probe (GPU box): kernel stats of single queries under a pipeline mode: bash tools/single_trace.sh <mode>
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-1}
OUT=$R/gpurun_out/st_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/coop_probe.py 111500 $M > $OUT/out.txt 2> $OUT/err.txt
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - $S <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-40s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"].split("(")[0].replace("void porrt::", "").replace("porrt::", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 $OUT/out.txt
rm -rf $OUT/trace
