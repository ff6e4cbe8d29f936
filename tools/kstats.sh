#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run (bash tools/kstats.sh tag [bench args...]) -> gpurun_out/kstats_<tag>.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-query --no-belief --no-profile "$@" > $OUT/bench.json 2> $OUT/trace.log
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $R/gpurun_out/kstats_$TAG.csv
rm -rf $OUT/trace
python3 - $R/gpurun_out/kstats_$TAG.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-34s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"].split("(")[0].replace("void porrt::", "").replace("porrt::", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -c 300 $OUT/bench.json | head -c 300
