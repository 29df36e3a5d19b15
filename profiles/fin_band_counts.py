#!/usr/bin/env python3
"""How many events does the finalize check send back to their rows (counter finalize_from_rows)?  One pass per workload.
Usage: python profiles/fin_band_counts.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("py-swirld_amd")
for name, n, N, mode, p0, p1, chunk in (("uniform 256 x 1 M", 256, 1_000_000, 0, 0, 0, None), ("hot members 256 x 1 M", 256, 1_000_000, 2, 0.95, 0.002, None),
                                       ("coin stress 256 x 1 M", 256, 1_000_000, 2, 0.35, 0.02, None), ("two cliques 256 x 1 M", 256, 1_000_000, 1, 0.02, 0, None),
                                       ("uniform 64 x 100 k", 64, 100_000, 0, 0, 0, None), ("uniform 1024 x 500 k", 1024, 500_000, 0, 0, 0, None),
                                       ("uniform 256 x 200 k in calls of 256 events", 256, 200_000, 0, 0, 0, 256)):
    st = pkg.synth_hashgraph(n, N, 3, mode, p0, p1)
    h = pkg.Hashgraph(n)
    h.reserve(N)
    if chunk is None:
        h.append_events(*st)
        h.divide_rounds(0, N)
        h.decide_fame()
    else:
        for a in range(0, N, chunk):
            b = min(N, a + chunk)
            h.append_events(*[x[a:b] for x in st])
            h.divide_rounds(a, b - a)
            h.decide_fame()
    c = h.counters()
    print("%-44s events %8d  from rows %8d (%.2f %%)  band events %d" % (name, N, c["finalize_from_rows"], 100.0 * c["finalize_from_rows"] / N, c["band_events"]))
    h.close()
