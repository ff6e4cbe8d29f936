#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB (hbm_bytes = counter * 1024).  On gfx950 FETCH_SIZE under-reports wide
(16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md, HBM); the scans stage nodes with 4- and 8-byte
coalesced loads, a width that guide leaves uncalibrated, so both the raw and the doubled figure are kept.
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n * 1024.0) for k, (n, v) in agg.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (0, 0.0)), write.get(k, (0, 0.0))
        out[k] = {"launches": max(f[0], w[0]), "fetch_bytes_per_launch_raw": f[1], "fetch_bytes_per_launch_x2": 2 * f[1],
                  "write_bytes_per_launch": w[1], "hbm_bytes_per_launch_raw": f[1] + w[1]}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_raw"])[:12]:
        print("%-28s fetch %9.0f B (x2 %9.0f)  write %9.0f B per launch" % (k, v["fetch_bytes_per_launch_raw"], v["fetch_bytes_per_launch_x2"], v["write_bytes_per_launch"]))


if __name__ == "__main__":
    main()
