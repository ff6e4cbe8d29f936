#!/usr/bin/env python3
"""developer probe: gaps between a single query's main-stream step kernels from a rocprofv3 kernel trace (python tools/single_gaps.py trace.csv)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
main = [r for r in rows if nm(r).startswith(("k_step", "k_file_commit", "k_near", "k_commit_rrt"))]
# the last query: after the last k_gen_samples
gi = [i for i, r in enumerate(rows) if nm(r) == "k_gen_samples"]
t_last = int(rows[gi[-1]]["Start_Timestamp"])
main = [r for r in main if int(r["Start_Timestamp"]) > t_last]
gaps, durs = [], []
for a, b in zip(main, main[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
print("kernels", len(main), "span us %.0f" % ((int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])) / 1e3),
      "sum of durations %.0f" % sum(durs), "sum of gaps %.0f" % sum(gaps))
import collections
g = sorted(gaps)
print("gap us: median %.1f  p90 %.1f  max %.1f" % (g[len(g) // 2], g[int(0.9 * len(g))], g[-1]))
print("gaps:", " ".join("%.0f" % x for x in gaps))
print("durations:", " ".join("%.0f" % x for x in durs))
side = [r for r in rows if int(r["Start_Timestamp"]) > t_last and nm(r).startswith(("k_kd", "k_tie"))]
import collections
agg = collections.defaultdict(list)
for r in side: agg[nm(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items(): print("side", k, "calls", len(v), "first 8 us:", " ".join("%.0f" % x for x in v[:8]), "median %.0f" % sorted(v)[len(v) // 2])
print("names:", " ".join(nm(r)[2:8] for r in main[:12]))
