#!/bin/bash
# developer probe (GPU box): rebuild with -D flags and run the short bench: bash tools/variant_bench.sh tag "-DA=1" "-DA=2" ...
tag=$1; shift
i=0
for flags in "$@"; do
  PORRT_CXXFLAGS="$flags" python -c "from po_rrt_amd import build as b; b.build(force=True)" > gpurun_out/${tag}_build_$i.log 2>&1 || { tail -5 gpurun_out/${tag}_build_$i.log; exit 1; }
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-belief --no-pmc > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err || { tail -5 gpurun_out/${tag}_$i.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_$i.json").read().strip().splitlines()[-1])
r=d.get("roofline",{}).get("kernels",{})
print("$flags", "value %.1f M/s on-device %.1f" % (d["value"]/1e6, d.get("value_trees_on_device",0)/1e6), "single %.2f ms" % (d.get("single_query") or {}).get("ms_per_query", 0),
      "near us %.0f conn us %.0f" % (r.get("k_nn2",{}).get("avg_launch_us",0), r.get("k_conn2",{}).get("avg_launch_us",0)), flush=True)
PY
  i=$((i+1))
done
