#!/bin/bash
# developer probe (GPU box): k_conn2's phase timers over the steps from <from> on (the steady state without the dense first steps), Q queries in one sequence
FROM=${1:-20}; Q=${2:-128}; WHICH=${4:-2}      # 2: k_conn2, 1: k_nn2
PORRT_CXXFLAGS="-DPORRT_TIMING=$WHICH -DPORRT_TIMING_FROM=$FROM $3" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py $Q batch_streams=1 2>&1 | grep "phase" | tail -8
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
