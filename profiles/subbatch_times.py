#!/usr/bin/env python3
"""Per sub-batch of a large sw_divide_rounds call, UNTRACED: how long its round loop waited for its sweep, how long the loop
took and how many iterations it ran (the library's SW_DEBUG_TIMING lines: one extra stream synchronisation per sub-batch, so
the sum is a little above an undisturbed pass; the host stage clocks are printed when the context is destroyed).
Usage: python profiles/subbatch_times.py [members events]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SW_DEBUG_TIMING"] = "1"
pkg = importlib.import_module("py-swirld_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3))
for i in range(5):
    h.rewind()
    h.synchronize()
    print("---- pass %d" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    h.divide_rounds(0, N)
    h.decide_fame()
    h.synchronize()
    print("pass %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
h.close()
