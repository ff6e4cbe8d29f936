"""Developer probe: a window of a batch's kernel trace, main and side kernels with start / end relative to the window
(python tools/batch_timeline.py kernel_trace.csv [first_step] [n_kernels])"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
gi = [i for i, r in enumerate(rows) if nm(r) == "k_gen_samples"]
q = rows[gi[-1]:]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
# the first_step-th k_nn2 after the last sample generation
nn = [i for i, r in enumerate(q) if nm(r).startswith("k_nn2")]
w = q[nn[first]:nn[first] + n]
t0 = int(w[0]["Start_Timestamp"])
for r in w:
    print("%-28s q%-3s start %8.1f  end %8.1f  dur %7.1f" % (nm(r)[:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
