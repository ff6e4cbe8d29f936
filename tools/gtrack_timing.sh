#!/bin/bash
# developer probe (GPU box): where g_track_step's time goes in a single query (-DPORRT_GTRACK_TIMING: cumulative marks 0 entry work, 4 coordinates + bisection, 5 exit levels written, 1 parallel part, 2 sequential part, 3 end)
PORRT_CXXFLAGS="-DPORRT_GTRACK_TIMING" python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1 || exit 1
PORRT_DEBUG=1 python tools/step_probe.py 1 kd_lazy=2 2>&1 | grep "phase" | tail -8
python -c "from po_rrt_amd import build as b; b.build(force=True)" > /dev/null 2>&1
