#!/bin/bash
# GPU box: counters of the expected-cost sweeps (k_dp_level_*), one rocprofv3 --pmc pass per set over ONE computation on the bench's
# 4095-belief graph (tools/dp_probe.py 1).   bash tools/pmc_dp.sh [tag]  ->  gpurun_out/pmc_dp_<tag>/set<i>.csv (one row per dispatch)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
OUT=$R/gpurun_out/pmc_dp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=(
 "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for set in "${SETS[@]}"; do
  d=$OUT/raw$i
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $d -o p -- python3 $R/tools/dp_probe.py 1 > $OUT/set$i.log 2>&1
  rc=$?
  echo "set $i rc=$rc: $set"
  if [ $rc -eq 0 ]; then
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - $f $OUT/set$i.csv <<'PY'
import csv, sys
rd = csv.DictReader(open(sys.argv[1]))
rows = [r for r in rd if "k_dp_" in r["Kernel_Name"]]
# one line per dispatch: kernel, grid, then counter=value pairs (rocprofv3 writes one row per dispatch and counter)
by = {}
for r in rows:
    key = int(r["Dispatch_Id"])
    nm = r["Kernel_Name"]
    e = by.setdefault(key, {"k": "init" if "level_init" in nm else ("sweep4" if "sweep<4u" in nm else ("sweep1" if "sweep<1u" in nm else nm.split("(")[0].split("::")[-1])), "grid": r["Grid_Size"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(sys.argv[2], "w") as o:
    for key in sorted(by):
        e = by[key]
        o.write("%d %s %s %s\n" % (key, e["k"], e["grid"], " ".join("%s=%.0f" % (c, v) for c, v in e.items() if c not in ("k", "grid"))))
PY
  fi
  rm -rf $d
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && break
  i=$((i+1))
done
# the traffic of the whole computation (what bench.py's expected_costs.roofline.traffic reads): FETCH_SIZE / WRITE_SIZE are in KB
python3 - $OUT <<'PY'
import json, sys, collections
out = sys.argv[1]
def load(i):
    rows = []
    try:
        for l in open("%s/set%d.csv" % (out, i)):
            f = l.split()
            rows.append((f[1], {kv.split("=")[0]: float(kv.split("=")[1]) for kv in f[3:]}))
    except OSError:
        pass
    return rows
res = collections.OrderedDict()
for i, c in ((4, "FETCH_SIZE"), (5, "WRITE_SIZE")):
    for k, d in load(i):
        e = res.setdefault("k_dp_level_" + k if k in ("init", "sweep1", "sweep4") else k, {"launches": 0, "fetch_bytes_raw": 0.0, "write_bytes": 0.0})
        if c == "FETCH_SIZE":
            e["launches"] += 1
            e["fetch_bytes_raw"] += 1024.0 * d.get(c, 0.0)
        else:
            e["write_bytes"] += 1024.0 * d.get(c, 0.0)
tot_f = sum(e["fetch_bytes_raw"] for e in res.values()); tot_w = sum(e["write_bytes"] for e in res.values())
res["_all"] = {"fetch_bytes_raw": tot_f, "write_bytes": tot_w, "hbm_bytes_2x_fetch_plus_write": 2.0 * tot_f + tot_w,
               "note": "one expected-cost computation on the bench's 4095-belief graph (tools/dp_probe.py 1), separate --pmc passes for FETCH_SIZE and WRITE_SIZE; "
                       "FETCH_SIZE doubled per the guide's gfx950 correction (an upper bound for these 8-byte-per-lane gathers)"}
json.dump(res, open(out + "/dp_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["_all"]))
PY
ls -la $OUT
