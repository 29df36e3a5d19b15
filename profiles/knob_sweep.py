#!/usr/bin/env python3
"""Environment-knob sweep in ONE process (no torch import: seconds per configuration).
Usage (GPU box): python profiles/knob_sweep.py [members events [passes]] -- CFG [CFG ...]
with CFG = "SW_PIPE=2,SW_CHUNKS=4" ("-" = defaults).  Knobs are read at sw_create, so every
configuration gets a fresh context over the same generated stream; prints min / median ms per pass
(sw_rewind + sw_divide_rounds + sw_decide_fame), events/s of the minimum, iterations, chunk counters."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("py-swirld_amd")
if os.environ.get("SWEEP_LIB"):   # A/B of two builds on one box: another libswirld_hip.so for this process (the loader reads the path at its first call)
    _l = importlib.import_module("py-swirld_amd._lib")
    _l.LIB_PATH = os.path.abspath(os.environ["SWEEP_LIB"])
    import ctypes as _C
    _have = _C.CDLL(_l.LIB_PATH)
    for _name in [k for k in _l.SIGNATURES if not hasattr(_have, k)]:   # (an older build: entries it does not export are not bound)
        del _l.SIGNATURES[_name]
    print("library:", os.environ["SWEEP_LIB"], flush=True)

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
head, cfgs = args[:split], args[split + 1:] or ["-"]
n = int(head[0]) if len(head) > 0 else 256
N = int(head[1]) if len(head) > 1 else 1_000_000
passes = int(head[2]) if len(head) > 2 else 9
mode = int(os.environ.get("GEN_MODE", "0"))
p0, p1 = float(os.environ.get("GEN_P0", "0")), float(os.environ.get("GEN_P1", "0"))
stream = pkg.synth_hashgraph(n, N, 3, mode, p0, p1)
base_env = dict(os.environ)
for cfg in cfgs:
    os.environ.clear()
    os.environ.update(base_env)
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
    h = pkg.Hashgraph(n)
    h.reserve(N)
    h.append_events(*stream)
    h.divide_rounds(0, N)
    h.decide_fame()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        h.rewind()
        h.divide_rounds(0, N)
        h.decide_fame()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    c = h.counters()
    per = passes + 1
    print("n=%d N=%d %-44s min %.3f ms  med %.3f ms  %.1f M ev/s | %d iterations, %d rounds, chunks %d prov %d resweeps %d" % (
        n, N, cfg, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, N / ts[0] / 1e6, c["round_iterations"] // per, c["rounds"],
        c["chunk_sweeps"] // per, c["chunk_provisional"] // per, c["chunk_resweeps"]), flush=True)
    h.close()
