"""Builds libporrt_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is built."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "porrt_engine.hip")
import glob
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(HERE, "..", "include", "porrt_hip.h")]
LIB = os.path.join(HERE, "libporrt_hip.so")
# -ffp-contract=off: the reference (Rust) never fuses a*b+c; parity is bit-exact only without contraction.
# -disable-promote-alloca-to-lds: the compiler moved k_kd_claim's per-thread moves (a small array) into LDS -- 77 KB per
# workgroup instead of 22 -- and a side-stream workgroup's LDS is LDS the step kernels beside it cannot have (two k_conn2
# workgroups on that CU instead of five); no other kernel is changed by it.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-mllvm", "-disable-promote-alloca-to-lds"]
# RCCL: the one exchange of a query-sharded job (csrc/porrt_exchange.hpp)
LIBS = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


SHIM_SRC = os.path.join(HERE, "csrc", "pto_c_shim.cpp")
SHIM_LIB = os.path.join(HERE, "libpo_rrt.so")


def build(force=False):
    if force or needs_build():
        cmd = [hipcc()] + FLAGS + os.environ.get("PORRT_CXXFLAGS", "").split() + ["-o", LIB, SRC] + LIBS
        subprocess.run(cmd, check=True)
    # libpo_rrt.so: the reference's own C symbols (src/pto_c.rs) on top of the C ABI above -- plain C++, no HIP
    deps = [SHIM_SRC, LIB, os.path.join(HERE, "..", "include", "po_rrt_c.h"), os.path.join(HERE, "..", "include", "porrt_hip.h")]
    if force or not os.path.exists(SHIM_LIB) or any(os.path.getmtime(d) > os.path.getmtime(SHIM_LIB) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SHIM_LIB, SHIM_SRC, "-L" + HERE, "-lporrt_hip", "-Wl,-rpath,$ORIGIN"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
