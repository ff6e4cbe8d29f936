#!/bin/bash
# Kernel trace + stats of porrt_build_belief_graph (tools/belief_probe.py), then FETCH_SIZE / WRITE_SIZE in their own
# --pmc passes.  Run on the GPU box:  bash tools/profile_belief.sh <tag> [n_iter]   (outputs under gpurun_out/bprof_<tag>/)
set -e
TAG=${1:-r1}
NIT=${2:-5000}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/belief_probe.py $NIT > $OUT/probe_traced.txt 2> $OUT/trace.log
# the counter passes run the 12-world problem only (WORLDS=12: two identical builds), so that per-launch averages belong to one size
export WORLDS=12
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/tools/belief_probe.py $NIT > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/tools/belief_probe.py $NIT > /dev/null 2> $OUT/write.log
unset WORLDS
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/fetch $OUT/write $OUT/trace
grep -E "k_bg|k_scan" $OUT/kernel_stats.csv
