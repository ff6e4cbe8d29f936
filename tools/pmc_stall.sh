#!/bin/bash
# GPU box: where the step kernels' cycles go -- a few --pmc passes of one bench step (separate passes, no tracing beside them),
# per-kernel averages per launch -> gpurun_out/pmc_stall_<tag>.txt.    bash tools/pmc_stall.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
OUT=$R/gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do        # (a pass with TA_* counters never finished on this pool: left out)
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-single-query --no-belief > /dev/null 2> $OUT/p$i.log || echo "pass $i ($set) failed" >> $OUT/fail.txt
  i=$((i+1))
done
python3 - $OUT $R/gpurun_out/pmc_stall_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void porrt::", "").replace("porrt::", "")
        c = out[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
with open(sys.argv[2], "w") as fo:
    for k in sorted(out):
        if not any(t in k for t in ("k_nn2", "k_conn2", "k_kd_locate", "k_kd_claim")):
            continue
        fo.write(k + "\n")
        for c in sorted(out[k]):
            v, n = out[k][c]
            fo.write("    %-34s per launch %16.1f   (launches %d)\n" % (c, v / n, n))
print(open(sys.argv[2]).read())
PY
cat $OUT/fail.txt 2>/dev/null
rm -rf $OUT/p*/
