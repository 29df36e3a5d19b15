"""Throughput of the exact (forked-hashgraph) path, csrc/exact.hip.h: one wavefront, the reference's
statements.  Usage (GPU box): python profiles/exact_bench.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("py-swirld_amd")
from test_exact_host import add_forks  # noqa: E402  (test helper: forked stream generator)

for n, N in ((8, 20_000), (64, 20_000), (256, 8_000)):
    stream = add_forks(pkg.synth_hashgraph(n, N, 5), n, 5, 20, start=0)
    cr = stream[0]
    h = pkg.Hashgraph(n)
    t0 = time.perf_counter()
    h.append_events(*stream)
    t1 = time.perf_counter()
    h.divide_rounds(0, len(cr))
    t2 = time.perf_counter()
    nc = h.decide_fame()
    t3 = time.perf_counter()
    tx = h.find_order(nc)
    t4 = time.perf_counter()
    assert h.exact
    print("exact path, %4d members x %6d events (%d forks): append %.3f s, divide_rounds %.3f s (%.0f ev/s), decide_fame %.3f s, "
          "find_order %.3f s (%d events ordered), %d rounds" % (n, len(cr), 20, t1 - t0, t2 - t1, len(cr) / (t2 - t1), t3 - t2, t4 - t3, len(tx), h.max_round + 1))
    h.close()
