#!/bin/bash
# Evidence for profiles/ (round 3 form).  Run on the GPU box:  bash tools/profile_round.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
#   1. rocprofv3 --kernel-trace --stats of a short bench run (timed steps + the eager roofline pass; batch launches only)
#   2. the default bench line (python bench.py): it measures roofline.traffic itself, by two rocprofv3 --pmc child runs
#      (FETCH_SIZE, WRITE_SIZE: one pass each) -- pmc_traffic.json is cut out of that line
#   3. kernel stats of single queries (tools/coop_probe.py, the default launch form) and of the belief-space rows (tools/belief_probe.py)
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-single-query --no-belief --no-pmc > $OUT/bench_traced_under_rocprof.json 2> $OUT/trace.log
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
cd $R
python3 bench.py --steps 10 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python3 - $OUT/bench_n1.json $OUT/pmc_traffic.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
json.dump({"source": r["traffic_source"], "kernels": r["traffic_raw"]}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("value %.1f M/s, trees on device %.1f, single %.2f ms, k_conn2 frac %.4f" % (d["value"] / 1e6, d["value_trees_on_device"] / 1e6, d["single_query"]["ms_per_query"], r["frac"]))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strace -o t -- python3 $R/tools/coop_probe.py 111500 4 > $OUT/single_probe_traced.txt 2> $OUT/strace.log
cp $(find $OUT/strace -name '*kernel_stats.csv' | head -1) $OUT/single_kernel_stats.csv
rm -rf $OUT/strace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/btrace -o t -- python3 $R/tools/belief_probe.py 20000 > $OUT/belief_probe_traced.txt 2> $OUT/btrace.log
cp $(find $OUT/btrace -name '*kernel_stats.csv' | head -1) $OUT/belief_kernel_stats.csv
rm -rf $OUT/btrace
head -8 $OUT/kernel_stats.csv | cut -d, -f1-7
