#!/bin/bash
# GPU box: per-kernel averages of a pure batch run (QS=<queries>)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr3 -o t -- python3 $R/tools/batch_probe.py > $R/gpurun_out/tr3.log 2>&1
cd $R
grep Q= gpurun_out/tr3.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tr3/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-40s calls %6s avg %9.1f us  total %8.1f ms" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
rm -rf gpurun_out/tr3
