"""Builders of the HOST builds of the device sources the CPU suite checks (crypto.hip.h through
tests/crypto_host.cpp, exact.hip.h through tests/exact_host.cpp).  No pytest / conftest / oracle
imports: __graft_entry__.build() calls these, and a deployment box may have none of them."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "py-swirld_amd", "csrc")

CRYPTO_SO = os.path.join(HERE, "libswc_host.so")
CRYPTO_SRC = os.path.join(HERE, "crypto_host.cpp")
CRYPTO_HDR = os.path.join(CSRC, "crypto.hip.h")
EXACT_SO = os.path.join(HERE, "libswx_host.so")
EXACT_SO_LANES = os.path.join(HERE, "libswx_host_lanes.so")
EXACT_SRC = os.path.join(HERE, "exact_host.cpp")
EXACT_HDR = os.path.join(CSRC, "exact.hip.h")


def _stale(so, *deps):
    return not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps)


def build_crypto_host():
    if _stale(CRYPTO_SO, CRYPTO_SRC, CRYPTO_HDR):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", CRYPTO_SRC, "-o", CRYPTO_SO])
    return CRYPTO_SO


def build_exact_host(lanes=False):
    so, extra = (EXACT_SO_LANES, ["-DSW_EXACT_HOST_LANES=16"]) if lanes else (EXACT_SO, [])
    if _stale(so, EXACT_SRC, EXACT_HDR):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + extra + [EXACT_SRC, "-o", so])
    return so
