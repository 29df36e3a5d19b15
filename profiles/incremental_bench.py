#!/usr/bin/env python3
"""N2: the main()-style schedule — append a small batch, divide_rounds,
decide_fame, find_order — timed per call.  Informational, not the headline metric."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("py-swirld_amd")


def run(n, N, chunk, seed=5):
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed)
    h = pkg.Hashgraph(n)
    h.reserve(N)
    t0 = time.perf_counter()
    calls = 0
    split = [0.0, 0.0, 0.0, 0.0]
    pc = time.perf_counter
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        s0 = pc()
        h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        s1 = pc()
        h.divide_rounds(a, b - a)
        s2 = pc()
        nc = h.decide_fame()
        s3 = pc()
        h.find_order(nc)
        s4 = pc()
        for i, d in enumerate((s1 - s0, s2 - s1, s3 - s2, s4 - s3)):
            split[i] += d
        calls += 1
    dt = time.perf_counter() - t0
    out = {"members": n, "events": N, "batch": chunk, "calls": calls, "ms_per_call": round(dt / calls * 1e3, 3),
           "events_per_s": round(N / dt, 1), "rounds": h.max_round + 1, "ordered": int(len(h.transactions())),
           "us_per_call": {k: round(v / calls * 1e6, 1) for k, v in zip(("append", "divide_rounds", "decide_fame", "find_order"), split)}}
    h.close()
    return out


if __name__ == "__main__":
    for n, N, chunk in [(4, 4000, 4), (64, 60000, 64), (64, 60000, 1000), (256, 200000, 256), (256, 200000, 5000)]:
        print(json.dumps(run(n, N, chunk)))
