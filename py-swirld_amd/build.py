"""Builds csrc/libswirld_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libswirld_hip.so")
SOURCES = ["swirld_hip.hip", "synth.cpp"]
DEPS = SOURCES + ["kernels.hip.h", "order.hip.h", "crypto.hip.h", "exact.hip.h", os.path.join("..", "..", "include", "swirld_hip.h")]


def build(force=False, verbose=False):
    newest = max(os.path.getmtime(os.path.join(CSRC, d)) for d in DEPS)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result"] + SOURCES + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
