#!/bin/bash
# GPU box: counter sets per kernel for one batch step (diagnostics): bash tools/pmc2.sh Q "--opt a=b" "SET1" "SET2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
Q=$1; OPTS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for set in "$@"; do
  rm -rf $R/gpurun_out/pmcp
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcp -o p -- python3 $R/bench.py --steps 1 --warmup 0 --queries $Q --no-cpu-baseline --no-profile --no-single-query --no-belief $OPTS > /dev/null 2> $R/gpurun_out/pmc2.err || tail -5 $R/gpurun_out/pmc2.err
  python3 - "$(find $R/gpurun_out/pmcp -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if not any(t in k for t in ("k_near", "k_connect", "k_nn2", "k_conn2", "k_kd_", "k_tie")): continue
    print(k, {c: "%.4g" % (v / max(n[(k, c)], 1)) for c, v in agg[k].items()})
PY
done
rm -rf $R/gpurun_out/pmcp
