"""On-disk formats either side of the hot path, host code of libporrt_hip.so -- no GPU needed.

PGM: what image 0.23's PNM decoder hands MapShelfDomain::open / Map::open (src/map_shelves_io.rs:88-103, src/map_io.rs:90-105).
JSON: PTOGraph save / load (src/pto_graph.rs:22-118); the fixture is the reference's own minimal graph
(create_minimal_graph, src/pto_graph.rs:540-562, serialised by test_graph_serialization :566-572)."""
import glob
import json
import os

import numpy as np
import pytest

import cases
from make_maps import read_pgm as py_read_pgm
from po_rrt_amd import engine

MAPS = sorted(glob.glob(os.path.join(cases.MAPS, "*.pgm")))


@pytest.mark.parametrize("path", MAPS, ids=[os.path.basename(p) for p in MAPS])
def test_committed_rasters_in_both_encodings(path, tmp_path):
    ref = py_read_pgm(path)
    got = engine.read_pgm(path)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)
    # the same raster as P2 (ASCII), with comments and ragged whitespace, and with another maxval: samples are not rescaled
    h, w = ref.shape
    body = "\n".join(" ".join(str(v) for v in row) for row in ref)
    p2 = ("P2\n# a comment\n%d # width\n%d\n# before maxval\n255\n%s\n" % (w, h, body)).encode()
    assert np.array_equal(engine.read_pgm(data=p2), ref)
    f = tmp_path / "m.pgm"
    f.write_bytes(p2)
    assert np.array_equal(engine.read_pgm(str(f)), ref)


def test_pnm_corner_cases():
    px = bytes([0, 127, 128, 200, 254, 99])
    assert engine.read_pgm(data=b"P5\n3 2\n254\n" + px).tolist() == [[0, 127, 128], [200, 254, 99]]       # maxval < 255: as stored
    assert engine.read_pgm(data=b"P5 3 2 255\t" + px).tolist() == [[0, 127, 128], [200, 254, 99]]          # any single whitespace byte
    assert engine.read_pgm(data=b"P5\n3 2\n255\n\n" + px[:5]).tolist() == [[10, 0, 127], [128, 200, 254]]   # a second newline is a sample
    assert engine.read_pgm(data=b"P2\n2 2\n15\n0 15\n7 3").tolist() == [[0, 15], [7, 3]]
    assert engine.read_pgm(data=b"P1\n3 1\n1 0 1").tolist() == [[0, 255, 0]]                               # bitmap: 1 = black
    assert engine.read_pgm(data=b"P4\n10 1\n" + bytes([0b10000000, 0b01000000])).tolist() == [[0] + [255] * 8 + [0]]
    for bad in (b"P5\n2 1\n65535\n\x00\x01\x00\x02",      # 16-bit gray: ImageLuma16, the reference says "Wrong image format!"
                b"P6\n1 1\n255\n\x00\x00\x00",              # colour
                b"P5\n3 2\n255\n" + px[:5],                 # short raster
                b"P2\n2 1\n255\n1 300", b"P2\n2 1\n255\n1 x", b"P5\n0 2\n255\n", b"PX\n", b""):
        with pytest.raises(engine.PorrtError) as ex:
            engine.read_pgm(data=bad)
        assert ex.value.code == -1
    with pytest.raises(engine.PorrtError) as ex:
        engine.read_pgm("/nonexistent/map.pgm")
    assert ex.value.code == -7


def minimal_graph():
    """create_minimal_graph (pto_graph.rs:540-562): 4 nodes, worlds [10, 01, 11], a diamond of bi-edges"""
    xy = [[0.0, 0.0], [1.0, 1.0], [1.0, -1.0], [2.0, 0.0]]
    node_validity = [2, 0, 1, 2]
    edges = []                                               # add_bi_edge(a, b, v) = add_edge(a, b, v); add_edge(b, a, v)
    for a, b, v in ((0, 1, 0), (0, 2, 1), (1, 3, 0), (2, 3, 1)):
        edges += [(a, b, v), (b, a, v)]
    children = [[(t, v) for f, t, v in edges if f == n] for n in range(4)]
    parents = [[(f, v) for f, t, v in edges if t == n] for n in range(4)]
    return xy, node_validity, children, parents, [[True, False], [False, True], [True, True]]


def csr(lists):
    off = np.cumsum([0] + [len(l) for l in lists])
    flat = [e for l in lists for e in l]
    return off, [e[0] for e in flat], [e[1] for e in flat]


def test_graph_json_is_serde_pretty_and_round_trips(tmp_path):
    xy, nv, children, parents, validities = minimal_graph()
    path = str(tmp_path / "test_graph_serialization.json")
    engine.graph_write_json(path, xy, nv, csr(children), csr(parents), validities)
    text = open(path).read()
    # what serde_json::to_writer_pretty writes for SerializablePTOGraph (field order of the structs, 2-space indent)
    expected = {"nodes": [{"state": s, "validity_id": v, "parents": [{"id": i, "validity_id": w} for i, w in p],
                           "children": [{"id": i, "validity_id": w} for i, w in c]} for s, v, p, c in zip(xy, nv, parents, children)],
                "validities": validities}
    assert json.loads(text) == expected
    assert text == json.dumps(expected, indent=2)            # python's indent=2 layout is serde_json's pretty layout for this shape
    assert text.startswith('{\n  "nodes": [\n    {\n      "state": [\n        0.0,\n        0.0\n      ],\n      "validity_id": 2,\n      "parents": [\n        {\n          "id": 1,')
    g = engine.graph_load_json(path)
    assert np.array_equal(g["xy"], np.array(xy)) and g["node_validity"].tolist() == nv
    for key, lists in (("children", children), ("parents", parents)):
        off, ids, vals = csr(lists)
        assert g[key][0].tolist() == off.tolist() and g[key][1].tolist() == ids and g[key][2].tolist() == vals
    assert g["validities"].tolist() == validities


def test_json_reader_takes_what_serde_takes_and_names_what_it_does_not(tmp_path):
    p = tmp_path / "g.json"
    p.write_text('{"validities":[[true,false]],"extra":{"a":[1,2,{"b":null}]},\n "nodes":[{"children":[],"parents":[],"validity_id":0,"state":[1e-7,-2.5E+3]}]}')
    g = engine.graph_load_json(str(p))
    assert g["xy"].tolist() == [[1e-7, -2500.0]] and g["validities"].tolist() == [[True, False]] and len(g["children"][1]) == 0
    for bad in ('{"nodes": []}', '{"nodes": [{"state": [0.0, 0.0], "validity_id": 0, "parents": []}], "validities": []}',
                '{"nodes": [{"state": [0.0], "validity_id": 0, "parents": [], "children": []}], "validities": []}',
                '{"nodes": [{"state": [0.0, 0.0], "validity_id": 0, "parents": [], "children": [{"id": 5, "validity_id": 0}]}], "validities": []}', 'nope'):
        p.write_text(bad)
        with pytest.raises(engine.PorrtError):
            engine.graph_load_json(str(p))
    with pytest.raises(engine.PorrtError):
        engine.graph_load_json(str(tmp_path / "missing.json"))


def test_float_notation_is_serde_jsons(tmp_path):
    """serde_json prints f64 through ryu: shortest round-trip digits, plain decimal for 1e-5 <= |x| < 1e16, exponent form
    d.ddde-7 / 1e16 outside (documented examples of the ryu crate: 1.234, 2.71828, 1e16, 1.234e-7 ...)"""
    vals = [0.1, 1.0, -1.0, 0.30000000000000004, 1.234, 2.71828, 1e15, 1e16, 1.5e16, 123456789012345680.0, 1e-5, 9.99e-6, 1.234e-7, 5e-324,
            1.7976931348623157e308, -0.0, 0.0, 100.0, 12345.678, 0.00012, -3.2e-7]
    text = ["0.1", "1.0", "-1.0", "0.30000000000000004", "1.234", "2.71828", "1000000000000000.0", "1e16", "1.5e16", "1.2345678901234568e17", "0.00001",
            "9.99e-6", "1.234e-7", "5e-324", "1.7976931348623157e308", "-0.0", "0.0", "100.0", "12345.678", "0.00012", "-3.2e-7"]
    xy = np.array(vals + [0.0] * (len(vals) % 2)).reshape(-1, 2)
    n = len(xy)
    path = str(tmp_path / "f.json")
    empty = (np.zeros(n + 1, dtype=np.uint64), [], [])
    engine.graph_write_json(path, xy, [0] * n, empty, empty, [[True]])
    lines = [l.strip().rstrip(",") for l in open(path).read().split("\n")]
    got = [lines[i + 1] for i, l in enumerate(lines) if l == '"state": ['] + [lines[i + 2] for i, l in enumerate(lines) if l == '"state": [']
    got = [v for pair in zip(got[:n], got[n:]) for v in pair][:len(vals)]
    assert got == text
    assert engine.graph_load_json(path)["xy"].ravel()[:len(vals)].tolist() == vals          # and they read back bit for bit
