#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_r4.sh: <dir>/raw/q<Q>_s<i>.csv -> <dir>/latency.txt, <dir>/per_step.csv.

Every figure is a mean per launch of the kernel (one launch = Q queries, one grow step).  Derived figures:
  tcp_read_latency   = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ      cycles from a vector L1 miss to its data (L2 hit or beyond)
  ea_read_latency    = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ              cycles an L2 miss spends in the fabric (Infinity Cache / HBM)
  ea_write_latency   = TCC_EA0_WRREQ_LEVEL / TCC_EA0_WRREQ
  vmem_latency       = SQ_INST_LEVEL_VMEM / (SQ_INSTS_VMEM_RD + _WR)    cycles a vector memory instruction is outstanding
  utcl1_miss_rate    = MISS / (HIT + MISS)                              translation misses of the vector L1's TLB
  wait_frac          = SQ_WAIT_ANY / SQ_WAVE_CYCLES                      share of wave time parked on a waitcnt
  icache_miss_rate   = SQC_ICACHE_MISSES / SQC_ICACHE_REQ
  l2_hit_rate        = TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    return name.split("(")[0].split("::")[-1].replace("void ", "")


def main():
    out = sys.argv[1]
    data = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])))
    series = collections.defaultdict(lambda: collections.defaultdict(list))      # (Q, kernel) -> counter -> [per dispatch]
    for f in sorted(glob.glob(os.path.join(out, "raw", "q*_s*.csv"))):
        Q = int(re.search(r"q(\d+)_s", f).group(1))
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if not (k.startswith("k_conn2") or k.startswith("k_nn2")):
                continue
            c = data[Q][k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                series[(Q, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out, "latency.txt"), "w") as fo:
        for Q in sorted(data):
            for k in sorted(data[Q]):
                m = {c: v[0] / max(v[1], 1) for c, v in data[Q][k].items()}
                n = max(v[1] for v in data[Q][k].values())
                fo.write("== %s  Q = %d queries per launch  (%d launches)\n" % (k, Q, n))
                for c in sorted(m):
                    fo.write("    %-52s %18.1f\n" % (c, m[c]))

                def ratio(name, a, b):
                    num = sum(m.get(x, 0.0) for x in a) if all(x in m for x in a) else None
                    den = sum(m.get(x, 0.0) for x in b) if all(x in m for x in b) else None
                    if num is not None and den:
                        fo.write("  > %-50s %18.3f\n" % (name, num / den))
                ratio("tcp_read_latency [cycles]", ["TCP_TCC_READ_REQ_LATENCY_sum"], ["TCP_TCC_READ_REQ_sum"])
                ratio("tcp_write_latency [cycles]", ["TCP_TCC_WRITE_REQ_LATENCY_sum"], ["TCP_TCC_WRITE_REQ_sum"])
                ratio("ea_read_latency [cycles]", ["TCC_EA0_RDREQ_LEVEL_sum"], ["TCC_EA0_RDREQ_sum"])
                ratio("ea_write_latency [cycles]", ["TCC_EA0_WRREQ_LEVEL_sum"], ["TCC_EA0_WRREQ_sum"])
                ratio("vmem_latency [cycles per instruction]", ["SQ_INST_LEVEL_VMEM"], ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"])
                ratio("smem_latency [cycles per instruction]", ["SQ_INST_LEVEL_SMEM"], ["SQ_INSTS_SMEM"])
                ratio("ifetch_latency [cycles]", ["SQ_IFETCH_LEVEL"], ["SQ_IFETCH"])
                ratio("utcl1_miss_rate", ["TCP_UTCL1_TRANSLATION_MISS_sum"], ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum"])
                ratio("wait_frac", ["SQ_WAIT_ANY"], ["SQ_WAVE_CYCLES"])
                ratio("issue_frac", ["SQ_ACTIVE_INST_ANY"], ["SQ_WAVE_CYCLES"])
                ratio("icache_miss_rate", ["SQC_ICACHE_MISSES"], ["SQC_ICACHE_REQ"])
                ratio("dcache_miss_rate", ["SQC_DCACHE_MISSES"], ["SQC_DCACHE_REQ"])
                ratio("l2_hit_rate", ["TCC_HIT_sum"], ["TCC_HIT_sum", "TCC_MISS_sum"])
                ratio("vmem_rd_per_wave", ["SQ_INSTS_VMEM_RD"], ["SQ_WAVES"])
                ratio("vmem_wr_per_wave", ["SQ_INSTS_VMEM_WR"], ["SQ_WAVES"])
                ratio("wave_cycles_per_wave [quad-cycles]", ["SQ_WAVE_CYCLES"], ["SQ_WAVES"])
                ratio("ta_busy_frac_of_gui", ["GRBM_TA_BUSY"], ["GRBM_GUI_ACTIVE"])
                ratio("tc_busy_frac_of_gui", ["GRBM_TC_BUSY"], ["GRBM_GUI_ACTIVE"])
                if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                    fo.write("  > %-50s %18.1f\n" % ("HBM MB per launch: 2 x FETCH + WRITE", (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6))
                    fo.write("  > %-50s %18.1f\n" % ("    FETCH raw MB", m["FETCH_SIZE"] * 1024 / 1e6))
                    fo.write("  > %-50s %18.1f\n" % ("    WRITE MB", m["WRITE_SIZE"] * 1024 / 1e6))
                fo.write("\n")
    with open(os.path.join(out, "per_step.csv"), "w") as fo:
        fo.write("queries_per_launch,kernel,step,fetch_raw_bytes,write_bytes\n")
        for (Q, k) in sorted(series):
            f, w = series[(Q, k)].get("FETCH_SIZE", []), series[(Q, k)].get("WRITE_SIZE", [])
            for s in range(max(len(f), len(w))):
                fo.write("%d,%s,%d,%s,%s\n" % (Q, k, s, "%.0f" % (f[s] * 1024) if s < len(f) else "", "%.0f" % (w[s] * 1024) if s < len(w) else ""))
    print(open(os.path.join(out, "latency.txt")).read())


if __name__ == "__main__":
    main()
