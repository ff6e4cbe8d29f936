#!/bin/bash
# GPU box: parity tests, a short bench, and the per-step kernel timeline of one traced query.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 || exit 1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('BENCH', d['value'], d['ms_per_step'])" || exit 1
if [ "$1" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr2 -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  cd $R
  python3 tools/step_timeline.py $(find gpurun_out/tr2 -name "*kernel_trace.csv")
  head -9 $(find gpurun_out/tr2 -name "*kernel_stats.csv") | cut -d, -f1-7 | sed 's/porrt::RunConst const\*, //; s/unsigned int/u/g'
  rm -rf gpurun_out/tr2
fi
