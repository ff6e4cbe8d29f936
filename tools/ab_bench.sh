#!/bin/bash
# GPU box: the bench's main row with the library in the tree and with another build of it (old_lib.so at the repo root), alternating, on one box
#   bash tools/ab_bench.sh [rounds] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}; shift
cd $R
cp po_rrt_amd/libporrt_hip.so /tmp/ab_new.so && cp old_lib.so /tmp/ab_old.so || exit 1
for i in $(seq $N); do
  for w in new old; do
    cp /tmp/ab_$w.so po_rrt_amd/libporrt_hip.so
    timeout -k 10 300 python bench.py --steps 5 --no-belief --no-cpu-baseline --no-pmc --no-single-query "$@" > gpurun_out/ab_$w.json 2> gpurun_out/ab_$w.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$w.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('$w', '%.1f M/s' % (d['value']/1e6), 'conn2 %.1f us' % k['k_conn2']['avg_launch_us'], 'nn2 %.1f us' % k['k_nn2']['avg_launch_us'])
"
  done
done
cp /tmp/ab_new.so po_rrt_amd/libporrt_hip.so
