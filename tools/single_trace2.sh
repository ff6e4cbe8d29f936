#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-1}
OUT=$R/gpurun_out/st2_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/coop_probe.py 111500 $M > $OUT/out.txt 2> $OUT/err.txt
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 $R/tools/single_gaps.py $T
rm -rf $OUT/trace
