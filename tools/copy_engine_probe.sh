#!/bin/bash
# developer probe (GPU box): are the tree fetch's device-to-host copies shader (blit) kernels or SDMA transfers?  A short bench under rocprofv3
# --kernel-trace --stats with the runtime's copy-engine switches: calls of __amd_rocclr_copyBuffer and the bench's value per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {   # label, then VAR=val ...
  local label=$1; shift
  local d=$R/gpurun_out/cep_$label
  rm -rf $d; mkdir -p $d
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-single-query --no-belief --no-pmc --no-profile > $d/bench.json 2> $d/bench.err )
  local s=$(find $d/trace -name '*kernel_stats.csv' | head -1)
  python3 - "$label" "$s" $d/bench.json <<'PY'
import csv, json, sys
label, stats, bj = sys.argv[1:4]
calls = tot = 0
try:
    for r in csv.DictReader(open(stats)):
        if "copyBuffer" in r["Name"]:
            calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])
except Exception as ex:
    print(label, "no kernel stats:", ex)
try:
    d = json.loads(open(bj).read().strip().splitlines()[-1])
    print("%-28s copyBuffer kernels %5d calls, %.1f ms;  value %.1f M/s, trees on device %.1f" % (label, calls, tot / 1e6, d["value"] / 1e6, d["value_trees_on_device"] / 1e6), flush=True)
except Exception as ex:
    print(label, "bench failed:", ex)
PY
  rm -rf $d/trace
}
run default
run sdma_forced HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
run blit_engine_2 GPU_BLIT_ENGINE_TYPE=2
