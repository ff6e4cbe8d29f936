#!/bin/bash
# Evidence for profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes.
# Run on the GPU box:  bash tools/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-query --no-belief > $OUT/bench_traced.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-single-query --no-belief > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-single-query --no-belief > /dev/null 2> $OUT/write.log
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $OUT/fetch $OUT/write
find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/trace
head -12 $OUT/kernel_stats.csv
