"""GPU (-m gpu): the C-ABI used from plain C (tests/c_abi_smoke.c, compiled with gcc against
include/swirld_hip.h) gives the same rounds / decided rounds / total order as the Python path."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client(pkg, tmp_path):
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-L", libdir, "-lswirld_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    n, N = 32, 20000
    out = subprocess.check_output([exe, str(n), str(N)], text=True).split()
    maxr, n_new, n_ord, digest = int(out[0]), int(out[1]), int(out[2]), int(out[3])
    h = pkg.Hashgraph(n)
    h.append_events(*pkg.synth_hashgraph(n, N, 11))
    h.divide_rounds(0, N)
    nc = h.decide_fame()
    order = h.find_order(nc)
    assert (maxr, n_new, n_ord) == (h.max_round, len(nc), len(order))
    x = 1469598103934665603
    for v in list(h.rounds()) + list(order):
        x = ((x ^ (int(v) & 0xFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert x == digest
