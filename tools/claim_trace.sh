#!/bin/bash
# developer probe (GPU box): durations of the side-chain kernels by group index over one bench step
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ct
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-single-query --no-belief --no-pmc --no-profile --opt batch_streams=1 > /dev/null 2> $OUT/err.txt
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - $T <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
# last grow: after the last k_batch_prep
gi = [i for i, r in enumerate(rows) if nm(r) == "k_batch_prep"]
q = rows[gi[-1]:]
for name in ("k_kd_claim<2048u, false>", "k_kd_locate<1>", "k_kd_link", "k_conn2<16>", "k_nn2<16>"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in q if nm(r) == name]
    print(name, len(d), "total %.1f ms" % (sum(d) / 1e3), " ".join("%.0f" % x for x in d[:60]))
PY
rm -rf $OUT/trace
