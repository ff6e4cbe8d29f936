#!/bin/bash
# developer tool (CPU only): where the step kernels spill -- scratch loads/stores per source line (bash tools/spillmap.sh [mangled-name-substring ...])
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -mllvm -disable-promote-alloca-to-lds -gline-tables-only --save-temps -o /tmp/spillmap.so $R/po_rrt_amd/csrc/porrt_engine.hip -L/opt/rocm/lib -lrccl 2>/dev/null
S=/tmp/porrt_engine-hip-amdgcn-amd-amdhsa-gfx950.s
[ $# -eq 0 ] && set -- k_conn2ILi16 k_nn2ILi16
for f in "$@"; do
  for name in $(grep -o "^_ZN5porrt[A-Za-z0-9_]*${f}[A-Za-z0-9_]*:" $S | tr -d ':' | sort -u); do
    awk "/^$name:/,/\.Lfunc_end/" $S > /tmp/f.s
    echo "== $name: $(grep -c . /tmp/f.s) lines, $(grep -c scratch_ /tmp/f.s) scratch ops, VGPRs $(grep -m1 'NumVgprs' /tmp/f.s | awk '{print $NF}')"
    # (.loc lines read "<n>:\t.loc\t<file> <line> <col> ...": keep file and line of the last one before each scratch access)
    awk '$1 == ".loc" {loc = "file " $2 " line " $3} /scratch_/ {print loc}' /tmp/f.s | sort | uniq -c | sort -rn | head -12
  done
done
