#!/usr/bin/env python3
"""developer probe: the TAMP-shaped row of bench.py at several (queries, K)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench
opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
for nq, K in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:] if ":" in a] or [(1024, 128)]:
    r = bench.tamp_queries(0, False, nq, K, opts)
    print(nq, K, "ms %.1f" % r["ms_wall"], "queries/s %.0f" % r["queries_per_s"], "exp/s %.1fM" % (r["node_expansions_per_s"] / 1e6),
          "mean its %.0f" % r["mean_iterations_per_query"], "solved", r["queries_solved"], "make ms %.0f" % r["ms_creating_the_contexts_once"], flush=True)
