#!/usr/bin/env python3
"""Summary of a rocprofv3 PC-sampling run (tools/pcsample.sh): samples of the step kernels by source line / instruction / stall reason."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
# disassembly with line info: address -> (function, file:line, instruction)
addr = {}
func, loc = None, None
for ln in open(os.path.join(out, "porrt.dis"), errors="replace"):
    m = re.match(r"^([0-9a-f]+) <(.+)>:", ln)
    if m:
        func = m.group(2)
        continue
    if ln.startswith(";") and ":" in ln:
        loc = ln[1:].strip()
        continue
    m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", ln)
    if m and func:
        addr[int(m.group(2), 16)] = (func, loc, m.group(1).strip())
files = glob.glob(os.path.join(out, "raw", "**", "*pc_sampling*.csv"), recursive=True)
print("pc sampling files:", [os.path.basename(f) for f in files], "; disassembled instructions:", len(addr))
for f in files:
    rows = csv.DictReader(open(f))
    cols = rows.fieldnames
    print("columns:", cols)
    by_line, by_inst, by_reason, by_kernel = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    n = 0
    off_col = next((c for c in cols if "offset" in c.lower()), None)
    reason_cols = [c for c in cols if "stall" in c.lower() or "reason" in c.lower() or "issued" in c.lower() or "inst_type" in c.lower()]
    for r in rows:
        n += 1
        try:
            a = int(r[off_col], 0) if off_col else None
        except Exception:
            a = None
        fn, lc, ins = addr.get(a, ("?", "?", "?"))
        short = re.sub(r"\(.*", "", fn).replace("void porrt::", "").replace("porrt::", "")[:40]
        by_kernel[short] += 1
        if "k_conn2" in fn or "k_nn2" in fn or "connect_rrt_sample" in fn or "heavy_sample" in fn or "group_nn" in fn:
            by_line[(short, lc)] += 1
            by_inst[(short, hex(a) if a is not None else "?", ins[:60], lc)] += 1
            for c in reason_cols:
                by_reason[(short, c, r[c])] += 1
    print("samples:", n)
    print("-- by kernel / function")
    for k, v in by_kernel.most_common(25):
        print("  %8d  %5.1f%%  %s" % (v, 100.0 * v / max(n, 1), k))
    print("-- step kernels: by source line (top 80)")
    for (k, lc), v in by_line.most_common(80):
        print("  %8d  %-30s %s" % (v, k, lc))
    print("-- step kernels: by instruction (top 80)")
    for (k, a, ins, lc), v in by_inst.most_common(80):
        print("  %8d  %-24s %-10s %-60s %s" % (v, k, a, ins, lc))
    print("-- step kernels: by reason columns")
    for (k, c, val), v in sorted(by_reason.items(), key=lambda kv: -kv[1])[:80]:
        print("  %8d  %-24s %-28s %s" % (v, k, c, val))
