#!/bin/bash
# GPU box: kernel trace of one eager batch (tools/step_probe.py ROWS batch_streams=1) -> a window of its timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${1:-128}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr3 -o t -- python3 $R/tools/step_probe.py $ROWS batch_streams=1 graph=0 > /dev/null 2>&1
cd $R
python3 tools/batch_timeline.py $(find gpurun_out/tr3 -name "*kernel_trace.csv") ${2:-40} ${3:-36}
rm -rf gpurun_out/tr3
