#!/bin/bash
# developer probe: the headline step under several engine options (bash tools/quick_bench.sh tag "opt=val ..." ...)
tag=$1; shift
i=0
for opts in "$@"; do
  args=""
  for o in $opts; do args="$args --opt $o"; done
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-belief $args > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_$i.json").read().strip().splitlines()[-1])
r=d.get("roofline",{}).get("kernels",{})
print("$opts", "value %.1f M/s" % (d["value"]/1e6), "ms/step %.1f" % d["ms_per_step"], "single", (d.get("single_query") or {}).get("ms_per_query"),
      "near us %.0f conn us %.0f" % (r.get("k_nn2",{}).get("avg_launch_us",0), r.get("k_conn2",{}).get("avg_launch_us",0)))
PY
  i=$((i+1))
done
