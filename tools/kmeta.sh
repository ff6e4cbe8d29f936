#!/bin/bash
# developer tool (CPU only): registers, scratch and LDS of the kernels from the code object's metadata (bash tools/kmeta.sh [name-substring ...] ; PORRT_CXXFLAGS honoured)
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/kmeta.XXXXXX)
cd $D && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -mllvm -disable-promote-alloca-to-lds $PORRT_CXXFLAGS --save-temps -o $D/k.so $R/po_rrt_amd/csrc/porrt_engine.hip -L/opt/rocm/lib -lrccl 2>/dev/null
S=$D/porrt_engine-hip-amdgcn-amd-amdhsa-gfx950.s
[ $# -eq 0 ] && set -- k_conn2 k_nn2
python3 - $S "$@" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
md = s[s.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", md)[1:]:
    m = re.search(r"\n\s+\.name:\s+(_Z\S+)", blk)
    if not m:
        continue
    name = m.group(1)
    if not any(t in name for t in sys.argv[2:]):
        continue
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    print("%-60s vgpr %s agpr %s sgpr %s spill_v %s spill_s %s scratch %s lds %s" % (name[:60], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
          g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
rm -rf $D
