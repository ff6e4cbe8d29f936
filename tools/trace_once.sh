#!/bin/bash
# GPU box: kernel trace of a short bench run -> per-step timeline + per-kernel averages
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr2 -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $R
python3 tools/step_timeline.py $(find gpurun_out/tr2 -name "*kernel_trace.csv")
head -9 $(find gpurun_out/tr2 -name "*kernel_stats.csv") | cut -c1-140
rm -rf gpurun_out/tr2
