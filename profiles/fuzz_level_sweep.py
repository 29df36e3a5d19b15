#!/usr/bin/env python3
"""Extended fuzz of the level sweep beyond 256 members (GPU box): python profiles/fuzz_level_sweep.py <first seed> <last seed>.
Random member counts 257 ... 1024, generator modes, call schedules and ring depths against the oracle, every shape
three passes (the rows must not depend on how the waves of a workgroup interleave).  Round 6: seeds 0..150, 0 mismatches."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("py-swirld_amd")
from oracle.oracle import Oracle
bad = 0
t0 = time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(123000 + seed)
    n = int(rng.choice([257, 300, 320, 400, 512, 513, 600, 777, 1024]))
    N = int(rng.integers(3000, 25000))
    mode = int(rng.integers(0, 4))
    p0, p1 = float(rng.uniform(0.01, 0.7)), float(rng.uniform(0.002, 0.2))
    chunk = None if rng.random() < 0.5 else int(rng.integers(200, max(201, N // 2)))
    H = int(rng.choice([0, 1, 1, 2, 4]))
    os.environ.pop("SW_RING_H", None)
    if H: os.environ["SW_RING_H"] = str(H)
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 55000 + seed, mode, p0, p1)
    o, h = Oracle(n), pkg.Hashgraph(n)
    step = chunk or N
    ok = True
    for a in range(0, N, step):
        b = min(N, a + step)
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b]); d.divide_rounds(a, b - a)
        ok = ok and list(o.decide_fame()) == list(h.decide_fame())
    same = np.array_equal(h.can_see(), o.can_see) and np.array_equal(h.rounds(), o.round) and ok
    for _ in range(2):   # repeated passes: the rows must not depend on how the waves interleave
        h.rewind(); h.divide_rounds(0, N); same = same and np.array_equal(h.can_see(), o.can_see)
    if not same:
        bad += 1
        print("MISMATCH seed %d n=%d N=%d mode=%d chunk=%s H=%d" % (seed, n, N, mode, chunk, H), flush=True)
    h.close()
print("fuzz seeds %s..%s: %d mismatches, %.0f s" % (sys.argv[1], sys.argv[2], bad, time.time() - t0), flush=True)
