#!/usr/bin/env python3
"""Duration of the resolve step of k_resolve_band without the per-phase stamps (each of which drains the wave):
SW_DEBUG_CLOCKS=2 keeps the entry, band-start and end stamps only.  Usage: python profiles/resolve_time.py [members events]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SW_DEBUG_CLOCKS"] = os.environ.get("SW_DEBUG_CLOCKS", "2")
os.environ.setdefault("SW_PIPE", "1")
pkg = importlib.import_module("py-swirld_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3))
for _ in range(2):
    h.divide_rounds(0, N)
    h.decide_fame()
    h.rewind()
h.divide_rounds(0, N)
t = h.debug_clocks().astype(np.int64)
live = (t[:, 0] > 0) & (t[:, 5] > 0) & (t[:, 6] > 0)
live[:-1] &= t[1:, 0] > 0
idx = np.nonzero(live)[0][1:-1]
q = lambda x: "mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (x.mean() / 100, *(np.percentile(x, [10, 50, 90]) / 100))
print("%d members, %d events, %d iterations, stamps: %s" % (n, N, len(idx), "entry / band start / end only" if os.environ["SW_DEBUG_CLOCKS"] == "2" else "all"))
print("  resolve step (entry -> band start)   ", q(t[idx, 5] - t[idx, 0]))
print("  band phase (band start -> end)       ", q(t[idx, 6] - t[idx, 5]))
print("  iteration period                     ", q(t[idx + 1, 0] - t[idx, 0]))
