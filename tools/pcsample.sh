#!/bin/bash
# GPU box: where the waves of the step kernels stand -- rocprofv3 PC sampling (beta) of ONE grow step of Q queries, the library rebuilt with line
# tables so that a PC maps to a source line.   bash tools/pcsample.sh [tag] [Q] [method: stochastic|host_trap] [interval]
#   -> gpurun_out/pcs_<tag>/summary.txt  (per kernel: samples by source line, by instruction, by stall reason where the method gives one)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}; Q=${2:-128}; METHOD=${3:-stochastic}; INTERVAL=${4:-262144}
OUT=$R/gpurun_out/pcs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
PORRT_CXXFLAGS="-gline-tables-only" python3 -c "from po_rrt_amd import build as b; b.build(force=True)" > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL --kernel-trace \
   --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --pmc-child --queries $Q --opt batch_streams=1 > $OUT/run.out 2> $OUT/run.err
rc=$?
echo "rocprofv3 rc=$rc"; tail -5 $OUT/run.err
[ $rc -ne 0 ] && exit 0
find $OUT/raw -type f | head -20
for f in $(find $OUT/raw -name '*.csv'); do echo "== $f: $(wc -l < $f) lines"; head -3 $f | cut -c1-600; done
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$OUT/fat.bin $R/po_rrt_amd/libporrt_hip.so
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$OUT/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT/porrt.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d -l --no-show-raw-insn $OUT/porrt.co > $OUT/porrt.dis 2>/dev/null
python3 $R/tools/pcsample_summary.py $OUT > $OUT/summary.txt 2> $OUT/summary.err; tail -3 $OUT/summary.err
head -120 $OUT/summary.txt
rm -rf $OUT/raw $OUT/fat.bin $OUT/porrt.co $OUT/porrt.dis
