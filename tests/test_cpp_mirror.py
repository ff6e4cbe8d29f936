"""The C++ host mirror (include/porrt.hpp) of the reference's RRT / PTO interface.

CPU: the header and the example compile with plain g++ against the C ABI and the program fails loudly
without a GPU.  GPU: the example's tree digest equals the oracle's for the same map / seed / batch."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import cases
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build_tools", "plan_rrt")
EXE_PTO = os.path.join(ROOT, "build_tools", "plan_pto")
MAP = os.path.join(ROOT, "tests", "golden", "maps", "map_benchmark_like.pgm")


@pytest.fixture(scope="module")
def exe():
    from po_rrt_amd import build
    build.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(os.path.join(ROOT, "include", "porrt.hpp")):
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", EXE, os.path.join(ROOT, "examples", "plan_rrt.cpp"),
                        "-L" + os.path.join(ROOT, "po_rrt_amd"), "-lporrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "po_rrt_amd")],
                       check=True)
    if not os.path.exists(EXE_PTO) or os.path.getmtime(EXE_PTO) < max(os.path.getmtime(os.path.join(ROOT, "include", "porrt.hpp")),
                                                                     os.path.getmtime(os.path.join(ROOT, "examples", "plan_pto.cpp"))):
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", EXE_PTO, os.path.join(ROOT, "examples", "plan_pto.cpp"),
                        "-L" + os.path.join(ROOT, "po_rrt_amd"), "-lporrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "po_rrt_amd")],
                       check=True)
    return EXE


def test_cpp_example_builds_and_needs_a_gpu(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([exe, MAP, "100", "100", "16", "0"], capture_output=True, text=True)
    assert out.returncode == 1 and "no usable HIP device" in out.stderr


def fnv_digest(xy, parent):
    h = 1469598103934665603
    for j in range(len(parent)):
        p = int(parent[j]) if parent[j] >= 0 else -1
        for v in (p & (2 ** 64 - 1), struct.unpack("<Q", struct.pack("<d", xy[j, 0]))[0], struct.unpack("<Q", struct.pack("<d", xy[j, 1]))[0]):
            for b in range(8):
                h ^= (v >> (8 * b)) & 0xFF
                h = (h * 1099511628211) & (2 ** 64 - 1)
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("K,nq", [(1, 1), (256, 1), (256, 3)])
def test_cpp_rrt_plan_matches_oracle(exe, K, nq):
    """nq > 1: RRT::plan_batch plans the query together with nq - 1 others; its own result must not change."""
    n = 400 if K == 1 else 3000
    out = subprocess.run([exe, MAP, str(n), str(n), str(K), "5"] + ([str(nq)] if nq > 1 else []), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    tok = out.stdout.split()
    case = cases.cfg2(n, seed=5)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
    xy, parent, _ = o.tree()
    assert int(tok[1]) == len(parent)
    assert int(tok[3], 16) == fnv_digest(xy, parent)
    sol = o.best_solution()
    if sol is None:
        assert "No solution found" in out.stdout
    else:
        assert int(tok[5]) == len(sol[0]) and float(tok[7]) == sol[1]


def fnv_words(h, words):
    for v in words:
        v = int(v)
        for b in range(8):
            h ^= (v >> (8 * b)) & 0xFF
            h = (h * 1099511628211) & (2 ** 64 - 1)
    return h


@pytest.mark.gpu
def test_cpp_pto_belief_graph_matches_oracle(exe):
    """PTO::grow_graph + PTO::build_belief_graph of the C++ mirror: digests of types, children and parents lists."""
    n = 16000                        # enough for both goals: a policy exists
    out = subprocess.run([EXE_PTO, os.path.join(ROOT, "tests", "golden", "maps", "map1_2_goals_like.pgm"),
                          os.path.join(ROOT, "tests", "golden", "maps", "map1_2_goals_like_zone_ids.pgm"), str(n), "64", "0"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    head, tok = out.stdout.split()[:8], out.stdout.split()[8:]
    case = cases.cfg3(n, n)
    o = cases.configure(orc.Oracle(), case)
    rc = cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
    o.build_belief_graph([0.5, 0.5])
    beliefs, types, (coff, cid), (poff, pid) = o.belief_graph()
    h0 = 1469598103934665603
    hc, hp = h0, h0
    for i in range(len(types)):
        hc = fnv_words(fnv_words(hc, [coff[i + 1] - coff[i]]), cid[int(coff[i]):int(coff[i + 1])])
        hp = fnv_words(fnv_words(hp, [poff[i + 1] - poff[i]]), pid[int(poff[i]):int(poff[i + 1])])
    assert int(tok[1]) == (1 if rc == 0 else 0)
    assert int(tok[3]) == o.num_nodes() and int(tok[5]) == len(beliefs) and int(tok[7]) == len(types) and int(tok[9]) == len(cid)
    assert int(tok[11], 16) == fnv_words(h0, types)
    assert int(tok[13], 16) == hc and int(tok[15], 16) == hp
    assert int(tok[17]) == len(types)          # one validity in the shelf domain: every pair is compatible
    d = o.expected_costs()
    assert int(head[1], 16) == fnv_words(h0, d.view(np.uint64))
    assert np.isfinite(d[0])
    oid, par, leaf = o.extract_policy(d)
    assert int(head[3]) == len(oid) and int(head[5]) == int(leaf.sum())
    hpol = h0
    for k in range(len(oid)):
        hpol = fnv_words(hpol, [oid[k], par[k] if par[k] >= 0 else 2 ** 64 - 1])
    assert int(head[7], 16) == hpol
    # the PRM roadmap printed on the second line: adjacency lists in the reference's push order
    tok2 = out.stdout.splitlines()[1].split()
    o2 = orc.Oracle()
    o2.set_grid(cases.load_map("map1_2_goals_like"), (-1.0, -1.0), (1.0, 1.0), cases.SHELF)
    o2.set_zones(cases.load_map("map1_2_goals_like_zone_ids"), 0.5)
    o2.set_sampler((-1.0, -1.0), (1.0, 1.0), 100)
    o2.grow_prm((-0.8, -0.8), 0.1, 5.0, 2000)
    f, t, _ = o2.edges()
    adj = [[] for _ in range(o2.num_nodes())]
    e0 = 0
    while e0 < len(t):
        e1 = e0
        while e1 < len(t) and t[e1] == t[e0]:
            e1 += 1
        for k in range(e0, e1):
            adj[f[k]].append(int(t[k]))
        for k in range(e0, e1):
            adj[t[k]].append(int(f[k]))
        e0 = e1
    hr = h0
    for a in adj:
        hr = fnv_words(fnv_words(hr, [len(a)]), a)
    assert int(tok2[1]) == o2.num_nodes() and int(tok2[3]) == 2 * len(f) and int(tok2[5], 16) == hr
    path = o2.prm_plan_path((-0.8, -0.8), (-0.5, 0.5))
    assert int(tok2[7]) == len(path) > 2 and int(tok2[9], 16) == fnv_words(h0, path.reshape(-1).view(np.uint64))
