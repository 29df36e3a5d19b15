#!/usr/bin/env python3
"""sw_decide_fame alone (the elections of every round of one batch call), host clock around a synchronised call.
Usage: python profiles/fame_time.py [members events [passes]]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("py-swirld_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 9
mode = int(os.environ.get("GEN_MODE", "0"))
p0, p1 = float(os.environ.get("GEN_P0", "0")), float(os.environ.get("GEN_P1", "0"))
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3, mode, p0, p1))
ts = []
for _ in range(passes + 1):
    h.rewind()
    h.divide_rounds(0, N)
    h.synchronize()
    t0 = time.perf_counter()
    nc = h.decide_fame()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
c = h.counters()
print("n=%d N=%d decide_fame: min %.1f us  med %.1f us | %d rounds decided, majority evaluations %d, coin votes %d" % (
    n, N, ts[0] * 1e6, ts[len(ts) // 2] * 1e6, len(nc), c["majority_evals"] // (passes + 1), c["coin_votes"] // (passes + 1)))
h.close()
