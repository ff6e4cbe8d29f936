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
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]
# RCCL: the one exchange of a query-sharded job (csrc/porrt_exchange.hpp)
LIBS = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + os.environ.get("PORRT_CXXFLAGS", "").split() + ["-o", LIB, SRC] + LIBS
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
