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
