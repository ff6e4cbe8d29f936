#!/bin/bash
# developer probe (GPU box): phase timers of a step kernel (-DPORRT_TIMING=<n>: 1 k_nn2, 2 k_conn2, 3 k_kd_locate), 64 queries in one sequence
N=${1:-2}
PORRT_CXXFLAGS="-DPORRT_TIMING=$N" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py 64 batch_streams=1 2>&1 | grep "phase" | tail -8
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
