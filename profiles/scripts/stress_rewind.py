#!/usr/bin/env python3
"""Stress: incremental run, rewind, one-batch divide — repeated, windowed and plain — against the oracle."""
import importlib, sys, os
sys.path.insert(0, "."); 
import numpy as np
pkg = importlib.import_module("py-swirld_amd")
from oracle.oracle import Oracle
n, N, chunk = 24, 150_000, 3_000
cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 701, 2, 0.25, 0.2)
o = Oracle(n); o.append_events(cr, sp, op, t, sig); o.divide_rounds(0, N); o.decide_fame()
exp = o.round
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for windowed in (True, False):
        for incremental in (True, False):
            h = pkg.Hashgraph(n)
            if windowed: h.set_window(True, 2)
            if incremental:
                for a in range(0, N, chunk):
                    b = min(N, a + chunk)
                    h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b]); h.divide_rounds(a, b - a)
                    h.find_order(h.decide_fame())
                ok_inc = np.array_equal(h.rounds(), exp)
                h.rewind()
            else:
                h.append_events(cr, sp, op, t, sig); ok_inc = None
            h.divide_rounds(0, N); h.decide_fame()
            r = h.rounds()
            ok = np.array_equal(r, exp)
            c = h.counters()
            print("rep %d windowed %d incremental-first %d: incremental ok %s, batch ok %s (max round %d vs %d, iterations %d)" % (
                rep, windowed, incremental, ok_inc, ok, r.max(), exp.max(), c["round_iterations"]), flush=True)
            if not ok:
                bad = np.nonzero(r != exp)[0]
                print("   first mismatch at event", bad[0], "of", len(bad), "creator", cr[bad[0]], "round", r[bad[0]], "expected", exp[bad[0]])
            h.close()
