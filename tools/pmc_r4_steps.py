#!/usr/bin/env python3
"""Per-step view of the counter passes of tools/pmc_r4.sh: one row per grow step of k_conn2 (or k_nn2) from the per-dispatch tables
<dir>/raw/q<Q>_s*.csv.   python tools/pmc_r4_steps.py <dir> [Q] [kernel-prefix]
Columns: duration from GRBM_GUI_ACTIVE / 8 XCDs at 2.1 GHz; occupancy = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x duration in cycles)."""
import collections
import csv
import glob
import sys

out, Q, kern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "128"), (sys.argv[3] if len(sys.argv) > 3 else "k_conn2")
D = collections.defaultdict(list)
for f in sorted(glob.glob("%s/raw/q%s_s*.csv" % (out, Q))):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        if r["Kernel_Name"].startswith(kern):
            D[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = len(D["GRBM_GUI_ACTIVE"])
print("# %s, %s queries per launch, one row per grow step (counter mode: kernels serialised)" % (kern, Q))
print("step      us   waves  waves/SIMD  wait%  issue%  VALU/wave  SALU/wave  loads/wave  stores/wave  LDS/wave  FETCH_MB  WRITE_MB  tcp_lat  ea_lat  l2_hit")
g = lambda c, s: D[c][s] if c in D and s < len(D[c]) else float("nan")
for s in range(n):
    gui = g("GRBM_GUI_ACTIVE", s) / 8
    w = g("SQ_WAVES", s)
    print("%4d %7.0f %7.0f %10.2f %6.1f %7.1f %10.0f %10.0f %11.0f %12.0f %9.0f %9.1f %9.1f %8.0f %7.0f %7.2f" % (
        s, gui / 2100, w, g("SQ_WAVE_CYCLES", s) * 4 / (1024 * gui), 100 * g("SQ_WAIT_ANY", s) / g("SQ_WAVE_CYCLES", s),
        100 * g("SQ_ACTIVE_INST_ANY", s) / g("SQ_WAVE_CYCLES", s), g("SQ_INSTS_VALU", s) / w, g("SQ_INSTS_SALU", s) / w, g("SQ_INSTS_VMEM_RD", s) / w,
        g("SQ_INSTS_VMEM_WR", s) / w, g("SQ_INSTS_LDS", s) / w, g("FETCH_SIZE", s) * 1024 / 1e6, g("WRITE_SIZE", s) * 1024 / 1e6,
        g("TCP_TCC_READ_REQ_LATENCY_sum", s) / g("TCP_TCC_READ_REQ_sum", s), g("TCC_EA0_RDREQ_LEVEL_sum", s) / g("TCC_EA0_RDREQ_sum", s),
        g("TCC_HIT_sum", s) / (g("TCC_HIT_sum", s) + g("TCC_MISS_sum", s))))
tot = sum(D["GRBM_GUI_ACTIVE"]) / 8 / 2100
print("# sum %.0f us; steps 0-8: %.0f us (%.1f %%)" % (tot, sum(D["GRBM_GUI_ACTIVE"][:9]) / 8 / 2100, 100 * sum(D["GRBM_GUI_ACTIVE"][:9]) / sum(D["GRBM_GUI_ACTIVE"])))
