#!/usr/bin/env python3
"""Where a find_order call goes (host laps printed by the library under SW_DEBUG_TIMING=1, each after a stream sync):
256 members x 1 M events by default, three calls on one context (rewind in between).  Usage: python profiles/order_laps.py [members events]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("py-swirld_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3))
for i in range(int(os.environ.get("ORDER_CALLS", "4"))):
    h.rewind()
    h.divide_rounds(0, N)
    nc = h.decide_fame()
    h.synchronize()
    if i == 2:
        os.environ["SW_DEBUG_TIMING"] = "1"   # (read per call)
        print("---- call %d, with laps (every lap ends with a stream synchronisation)" % i, file=sys.stderr)
    else:
        os.environ.pop("SW_DEBUG_TIMING", None)
    t0 = time.perf_counter()
    tx = h.find_order(nc)
    dt = (time.perf_counter() - t0) * 1e3
    print("call %d: %d events ordered in %.3f ms" % (i, len(tx), dt), file=sys.stderr)
