#!/bin/bash
# developer probe (GPU box): losers / rounds / time of k_kd_claim (-DPORRT_CLAIM_PROBE build; "phase 0" = the workgroup's run time per
# launch, "phase 1" = rounds of its last wave, "g_nd" = whole-workgroup rounds over all launches), then the normal build again
# bash tools/claim_probe.sh [rows] [option=value ...]
ROWS=${1:-64}; shift
PORRT_CXXFLAGS="-DPORRT_CLAIM_PROBE" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py $ROWS batch_streams=1 "$@" 2>&1 | grep "batch member 0\|phase" | tail -3
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
