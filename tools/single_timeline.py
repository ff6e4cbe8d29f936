"""Developer probe: kernel timeline of one single query from a rocprofv3 kernel trace (python tools/single_timeline.py trace.csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].split('::')[-1].replace('void ', '').split('<')[0]
gi = [i for i, r in enumerate(rows) if nm(r) == 'k_gen_samples']
q = rows[gi[-1]:]
t0 = int(q[0]['Start_Timestamp'])
main = [(nm(r), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in q]
for n, t, d in main[300:360]:
    print("%-16s start %8.1f  dur %6.1f" % (n, t, d))
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for n, t, d in main:
    agg[n][0] += 1; agg[n][1] += d
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-18s calls %4d total %8.1f us avg %6.1f" % (n, c, d, d / c))
print("span us", (int(q[-1]['End_Timestamp']) - t0) / 1e3)
