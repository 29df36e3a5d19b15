"""Candidate-window size (SW_TALLY_K) against member count: one process, one generated stream per size.
Usage (GPU box): python profiles/k_sweep.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("py-swirld_amd")

# optional second knob: python profiles/k_sweep.py SW_GALLOP 0 1 2   (every K x every value of that switch)
KNOB, VALUES = (sys.argv[1], sys.argv[2:]) if len(sys.argv) > 2 else (None, [None])
for n, N, Ks in ((1024, 2_000_000, (4, 8, 12, 16, 20, 28)), (256, 1_000_000, (12, 16, 20, 24, 28)), (64, 100_000, (8, 16, 28))):
    stream = pkg.synth_hashgraph(n, N, 3)
    for K, val in [(K, v) for K in Ks for v in VALUES]:
        os.environ["SW_TALLY_K"] = str(K)
        if KNOB:
            os.environ[KNOB] = val
        h = pkg.Hashgraph(n)
        h.reserve(N)
        h.append_events(*stream)
        h.divide_rounds(0, N); h.decide_fame()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            h.rewind(); h.divide_rounds(0, N); h.decide_fame()
            best = min(best, time.perf_counter() - t0)
        c = h.counters()
        print("n=%4d N=%7d K=%2d%s: %.3f ms per pass, %.1f M ev/s, %d iterations for %d rounds" % (
            n, N, K, " %s=%s" % (KNOB, val) if KNOB else "", best * 1e3, N / best / 1e6, c["round_iterations"] // 4, c["rounds"]), flush=True)
        h.close()
