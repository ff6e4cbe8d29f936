"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/porrt_hip.h declares.  No compute call is made (there is no GPU here and no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from po_rrt_amd import build, engine
    build.build()
    return engine.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "porrt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(porrt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from po_rrt_amd import engine
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libporrt_hip.so does not export %s" % s
    assert sorted(engine.SYMBOLS) == syms


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "porrt_hip.h")).read()
    for cite in ("rrt.rs:102-174", "pto.rs:55-139", "common.rs:310-333", "sample_space.rs", "pto_c.rs:63-270"):
        assert cite in text


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    from po_rrt_amd import Engine, PorrtError
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(PorrtError):
        Engine()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pat = re.compile(r"from\s+oracle|import\s+oracle|liboracle|porrt_oracle\.h|\borc_[a-z]|oracle/")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "po_rrt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(src), "%s references the oracle" % f
    hdr = open(os.path.join(ROOT, "include", "porrt_hip.h")).read()
    assert not pat.search(hdr)


def test_null_context_is_an_error_not_a_crash(lib):
    """Every entry point of the widened rows rejects a NULL context with a negative code (no device is touched)."""
    import ctypes as C
    import numpy as np
    z = np.zeros(4)
    assert lib.porrt_build_belief_graph(None, z, 4) < 0
    assert lib.porrt_bg_compute_expected_costs(None) < 0
    assert lib.porrt_bg_extract_policy(None, None, None, None, 0, None) < 0
    assert lib.porrt_grow_prm(None, z, 0.1, 2.0, 10) < 0
    assert lib.porrt_prm_plan_path(None, z, z, None, 0) < 0
    assert lib.porrt_bg_get_expected_costs(None, z) < 0
    assert lib.porrt_bg_num_edges(None) == 0 and lib.porrt_bg_num_nodes(None) == 0 and lib.porrt_bg_num_beliefs(None) == 0
    d = C.c_double(0.0)
    assert lib.porrt_best_cost(None, C.byref(d), None) < 0
    # the explicit-graph entry validates its arrays before looking for a device
    one = np.array([0.0, 0.0])
    assert lib.porrt_conditional_dijkstra(0, 0, one, np.zeros(1, dtype=np.uint32), np.ones((1, 1)), 1, 1, np.ones(1, dtype=np.uint8),
                                          np.zeros(2, dtype=np.uint64), np.zeros(1, dtype=np.uint32), np.zeros(2, dtype=np.uint64),
                                          np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint64), 0, np.zeros(1)) < 0      # n = 0
