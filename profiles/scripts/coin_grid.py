#!/usr/bin/env python3
"""Which generator settings reach coin rounds (swirld.py:267-272) at 1024 members?  GPU exploration."""
import importlib, sys, time
sys.path.insert(0, ".")
pkg = importlib.import_module("py-swirld_amd")
n, N = 1024, 3_000_000
for mode, p0, p1 in [(2, 0.35, 0.02), (2, 0.40, 0.02), (2, 0.45, 0.01), (2, 0.50, 0.005), (2, 0.60, 0.01), (1, 0.002, 0), (1, 0.0005, 0), (2, 0.34, 0.002)]:
    s = pkg.synth_hashgraph(n, N, 87, mode, p0, p1)
    h = pkg.Hashgraph(n); h.reserve(N); h.append_events(*s)
    t = time.time(); h.divide_rounds(0, N); nc = h.decide_fame(); dt = time.time() - t
    c = h.counters()
    print("mode %d p0 %.4f p1 %.4f: %.2f s, rounds %d, decided %d, iterations %d, coin votes %d (flips %d), P2 %d" % (
        mode, p0, p1, dt, c["rounds"], len(nc), c["round_iterations"], c["coin_votes"], c["coin_flips"], c["majority_evals"]), flush=True)
    h.close()
