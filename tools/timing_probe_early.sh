#!/bin/bash
# developer probe (GPU box): k_nn2's phase timers over the first 27 steps of 64 queries (the tree does not cover the map yet)
PORRT_CXXFLAGS="-DPORRT_TIMING=1" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py 64 batch_streams=1 n_iter=${1:-27648} 2>&1 | grep "phase" | tail -8
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
