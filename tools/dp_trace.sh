#!/bin/bash
# GPU box: duration of every k_dp_level_* dispatch of one expected-cost computation on the bench's 4095-belief graph, grouped by level
#   bash tools/dp_trace.sh [tag]  ->  gpurun_out/dp_levels_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
D=$R/gpurun_out/dptrace_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/dp_probe.py 1 > $R/gpurun_out/dptrace_$TAG.log 2>&1 || exit 1
F=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - $F > $R/gpurun_out/dp_levels_$TAG.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_dp_level" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lv = []
for r in rows:
    nm = r["Kernel_Name"]
    st, en = int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3
    if "init" in nm:
        lv.append({"t0": st, "init": en - st, "sw": [], "kind": "", "grid": 0})
    elif "sweep" in nm and lv:
        lv[-1]["sw"].append((st, en - st))
        lv[-1]["kind"] = "4 lanes per row" if "4u" in nm else "1 lane per row"
        lv[-1]["grid"] = r.get("Grid_Size_X") or r.get("Grid_Size")
tot = 0.0
for i, l in enumerate(lv):
    d = [x[1] for x in l["sw"]]
    end = l["sw"][-1][0] + l["sw"][-1][1]
    tot += end - l["t0"]
    print("level %2d: %s threads per sweep (%s), %d sweeps, kernels %.1f us, first launch to last end %.1f us, init %.1f us" % (i, l["grid"], l["kind"], len(d), sum(d), end - l["t0"], l["init"]))
    print("    " + " ".join("%.0f" % x for x in d))
print("all levels: %.1f us" % tot)
PY
rm -rf $D
tail -1 $R/gpurun_out/dp_levels_$TAG.txt
