#!/bin/bash
# developer probe (GPU box): losers / rounds / time of k_kd_claim (-DPORRT_CLAIM_PROBE build), then the normal build again
PORRT_CXXFLAGS="-DPORRT_CLAIM_PROBE" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py 64 batch_streams=1 2>&1 | grep "batch member 0\|phase" | tail -6
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
