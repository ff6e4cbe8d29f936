#!/bin/bash
# developer probe (GPU box): instruction-cache and issue counters of the step kernels over one grow step of 128 queries
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rm -rf $R/gpurun_out/pmcl
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcl -o p -- python3 $R/bench.py --pmc-child --queries 128 > /dev/null 2> $R/gpurun_out/pmcl.err || { echo "pass [$set] failed / timed out"; continue; }
  python3 - "$(find $R/gpurun_out/pmcl -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if not any(t in k for t in ("k_nn2", "k_conn2", "k_kd_locate")): continue
    print(k, {c: "%.4g" % (v / max(n[(k, c)], 1)) for c, v in agg[k].items()})
PY
done
rm -rf $R/gpurun_out/pmcl
