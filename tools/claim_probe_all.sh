#!/bin/bash
# developer probe (GPU box): the longest and the mean k_kd_claim workgroup of every member of a batch (-DPORRT_CLAIM_PROBE build)
ROWS=${1:-128}; shift
PORRT_CXXFLAGS="-DPORRT_CLAIM_PROBE" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG_ALL=1 python tools/step_probe.py $ROWS batch_streams=1 "$@" 2>&1 | grep "longest" | tail -$ROWS | sort -k7 -n -r | awk 'NR<=6 || NR%32==0'
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
