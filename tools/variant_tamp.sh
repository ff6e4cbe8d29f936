#!/bin/bash
# GPU box: the TAMP-shaped row and the main row rebuilt with compile-time variants, one line each:  bash tools/variant_tamp.sh "-DFLAG=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  PORRT_CXXFLAGS="$v" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  t=$(timeout -k 10 200 python tools/tamp_probe.py 1024 128 5 2>&1 | tail -3 | awk '{print $3}' | tr '\n' ' ')
  timeout -k 10 300 python bench.py --steps 5 --no-belief --no-cpu-baseline --no-pmc --no-single-query > gpurun_out/var.json 2> gpurun_out/var.err || exit 1
  m=$(python3 -c "
import json
d=json.loads(open('gpurun_out/var.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('%.1f M/s conn2 %.1f nn2 %.1f' % (d['value']/1e6, k['k_conn2']['avg_launch_us'], k['k_nn2']['avg_launch_us']))")
  echo "[$v] TAMP ms: $t | main: $m"
done
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
