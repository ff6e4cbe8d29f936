#!/bin/bash
# developer probe (GPU box): losers / rounds / time of k_kd_claim in the TAMP-shaped batch (-DPORRT_CLAIM_PROBE build), then the normal build again
PORRT_CXXFLAGS="-DPORRT_CLAIM_PROBE" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/tamp_steps.py ${1:-512} ${2:-128} 2>&1 | grep "batch member 0\|phase\|claim" | tail -8
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
