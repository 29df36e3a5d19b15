#!/usr/bin/env python3
"""Stress for a rare mismatch seen once: windowed incremental run, getters, rejected append, rewind, batch divide."""
import importlib, sys
sys.path.insert(0, ".")
import numpy as np
pkg = importlib.import_module("py-swirld_amd")
from oracle.oracle import Oracle
n, N, chunk = 24, 150_000, 3_000
cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 701, 2, 0.25, 0.2)
o = Oracle(n); o.append_events(cr, sp, op, t, sig); o.divide_rounds(0, N); o.decide_fame()
exp, ecs = o.round, o.can_see
# dirty the allocator first
junk = [pkg.Hashgraph(200) for _ in range(3)]
for j in junk:
    s = pkg.synth_hashgraph(200, 60000, 5); j.append_events(*s); j.divide_rounds(0, 60000); j.close()
fails = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    h = pkg.Hashgraph(n); h.set_window(True, 2)
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b]); h.divide_rounds(a, b - a)
        h.find_order(h.decide_fame())
    ok_inc = np.array_equal(h.rounds(), exp)
    first, resident, ev = h.window()
    cs_ok = np.array_equal(h.can_see(first, N - first), ecs[first:])
    try: h.append_events([int(cr[N - 1])], [N - 1], [0 if cr[0] != cr[N - 1] else 1])
    except pkg.SwirldHipError: pass
    h.rewind(); h.divide_rounds(0, N); h.decide_fame()
    r = h.rounds(); ok = np.array_equal(r, exp)
    if not (ok and ok_inc and cs_ok):
        fails += 1
        cs2 = h.can_see(0, N)
        badrows = np.nonzero((cs2 != ecs).any(axis=1))[0]
        bad = np.nonzero(r != exp)[0]
        print("rep %d FAIL: inc %s cs %s batch %s | rounds differ at %d events (first %d) | can_see rows differ: %d (first %s) | iterations %d" % (
            rep, ok_inc, cs_ok, ok, len(bad), bad[0] if len(bad) else -1, len(badrows), badrows[:5], h.counters()["round_iterations"]), flush=True)
        h.rewind(); h.divide_rounds(0, N); h.decide_fame()
        print("   second rewind+divide: rounds ok %s, can_see ok %s" % (np.array_equal(h.rounds(), exp), np.array_equal(h.can_see(0, N), ecs)), flush=True)
    h.close()
print("reps done, failures:", fails)
