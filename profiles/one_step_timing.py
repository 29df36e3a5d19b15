#!/usr/bin/env python3
"""Host-side timing of one bench step, sub-batch by sub-batch: how long each round loop waited
for its can_see sweep, how long it ran and at what cost per iteration (SW_DEBUG_TIMING=1 makes
sw_divide_rounds print this).  Usage: python profiles/one_step_timing.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SW_DEBUG_TIMING"] = "1"  # read when the context is created
pkg = importlib.import_module("py-swirld_amd")
n, N = 256, 1000000
st = pkg.synth_hashgraph(n, N, 3)
h = pkg.Hashgraph(n); h.reserve(N); h.append_events(*st)
for i in range(3):
    t0 = time.perf_counter(); h.divide_rounds(0, N); t1 = time.perf_counter(); h.decide_fame(); t2 = time.perf_counter()
    print("divide %.3f ms, fame %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3)); h.rewind()
