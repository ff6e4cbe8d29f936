#!/bin/bash
# GPU box: what bounds the step kernels -- one rocprofv3 --pmc pass per counter set (never beside a trace), each over ONE grow step
# (`bench.py --pmc-child`: one porrt_grow_batch of Q configs[1] queries, a single launch sequence so that a launch = Q queries),
# at Q = 64 / 128 / 256 queries per launch.   bash tools/pmc_r4.sh [tag] ["Q ..."] [extra bench args]
#   -> gpurun_out/pmc_<tag>/latency.txt    per kernel and Q: every counter's mean per launch + the derived figures
#   -> gpurun_out/pmc_<tag>/per_step.csv   FETCH_SIZE / WRITE_SIZE of every k_nn2 / k_conn2 dispatch by step index (Q = 128)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
QS=${2:-"64 128 256"}
EXTRA=${3:-}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT/raw
cd /tmp && export TMPDIR=/tmp
SETS=(
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
 "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_TC_BUSY GRBM_UTCL2_BUSY"
 "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
one_pass() {   # Q index set...: 0 ok, 1 ordinary failure, 2 timed out (then no further GPU step in this call)
  local Q=$1 i=$2; shift 2
  local d=$OUT/raw/q${Q}_s$i
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $d -o p -- python3 $R/bench.py --pmc-child --queries $Q --opt batch_streams=1 $EXTRA > /dev/null 2> $OUT/raw/q${Q}_s$i.err
  local rc=$?
  if [ $rc -eq 0 ]; then
    local f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp $f $OUT/raw/q${Q}_s$i.csv
    echo "Q=$Q set $i ok: $*"
  else
    echo "Q=$Q set $i rc=$rc: $*" | tee -a $OUT/failed.txt
    tail -3 $OUT/raw/q${Q}_s$i.err
  fi
  rm -rf $d
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && return 2
  [ $rc -eq 0 ] && return 0
  return 1
}
STOP=0
for Q in $QS; do
  i=0
  for set in "${SETS[@]}"; do
    one_pass $Q $i $set
    [ $? -eq 2 ] && { STOP=1; break 2; }
    i=$((i+1))
  done
done
# (a pass with TA counters has hung on this pool before: once, last, bounded)
[ $STOP -eq 0 ] && one_pass 128 99 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
python3 $R/tools/pmc_r4_summary.py $OUT
# the raw per-dispatch tables are large: keep only the step kernels' rows
for f in $OUT/raw/*.csv; do
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "k_conn2" in r["Kernel_Name"] or "k_nn2" in r["Kernel_Name"]]
if rows:
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    with open(sys.argv[1], "w") as fo:
        w = csv.writer(fo)
        w.writerow(cols)
        for r in keep:
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].split("::")[-1], r["Counter_Name"], r["Counter_Value"]])
PY
done
rm -f $OUT/raw/*.err
ls $OUT
