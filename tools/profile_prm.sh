#!/bin/bash
# Kernel trace + stats of porrt_grow_prm (tools/prm_probe.py).  Run on the GPU box: bash tools/profile_prm.sh <tag>
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/prm_probe.py 200000 1000000 > $OUT/probe_traced.txt 2> $OUT/trace.log
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
rm -rf $OUT/trace
grep -E "k_prm|k_scan" $OUT/kernel_stats.csv
cat $OUT/probe_traced.txt
