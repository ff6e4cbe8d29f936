#!/bin/bash
# developer probe (GPU box): rebuild with -D flags and run the TAMP-shaped row (tools/tamp_probe.py): bash tools/variant_tamp.sh tag "-DA=1" "-DA=2" ...
tag=$1; shift
i=0
for flags in "$@"; do
  PORRT_CXXFLAGS="$flags" python -c "from po_rrt_amd import build as b; b.build(force=True)" > gpurun_out/${tag}_build_$i.log 2>&1 || { tail -5 gpurun_out/${tag}_build_$i.log; exit 1; }
  echo "[$flags] $(timeout -k 10 200 python tools/tamp_probe.py 1024:128 2>&1 | tail -1)"
  i=$((i+1))
done
