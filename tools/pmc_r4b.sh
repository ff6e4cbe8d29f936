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
 "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"
 "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LEVEL_WAVES"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum"
 "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum"
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
# (TA counter passes hang on this pool -- twice now: not run; TD counters, unknown, last and bounded)
[ $STOP -eq 0 ] && one_pass 128 99 TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
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
