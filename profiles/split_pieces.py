#!/usr/bin/env python3
"""What ONE part of the intra-iteration split (include/swirld_hip.h part 3) runs per iteration, alone on a GPU: a single
context plays the parts one behind the other (SW_SPLIT_EMULATE=<parts>, results unchanged), and the durations of the
band kernel and of the tally kernel of a part come from rocprofv3's kernel trace of this script.
  run:   rocprofv3 --kernel-trace -d <dir> -o kt -- python profiles/split_pieces.py <parts> [members events]
  read:  python profiles/split_pieces.py --read <results.db> <parts>"""
import importlib
import os
import sqlite3
import sys

if sys.argv[1] == "--read":
    db = sqlite3.connect(sys.argv[2])
    parts = int(sys.argv[3])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    for pat in ("k_resolve_band", "k_tally_bits"):
        rows = [b - a for (a, b) in db.execute("select d.start, d.end from %s d join %s s on d.kernel_id = s.id where s.%s like '%%%s%%'" % (disp, sym, name_col, pat))]
        live = [x for x in rows if x > 3000]   # (launches behind a finished loop return at once)
        if live:
            live.sort()
            print("parts %d  %-16s launches %6d  live %6d  avg %8.2f us  median %8.2f us  p90 %8.2f us" % (
                parts, pat, len(rows), len(live), sum(live) / len(live) / 1e3, live[len(live) // 2] / 1e3, live[int(len(live) * 0.9)] / 1e3))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
parts = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
os.environ["SW_TALLY_IMPL"] = "1"
if parts > 0:
    os.environ["SW_SPLIT_EMULATE"] = str(parts)
pkg = importlib.import_module("py-swirld_amd")
import time
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3))
for i in range(2):
    h.rewind()
    h.synchronize()
    t0 = time.perf_counter()
    h.divide_rounds(0, N)
    nc = h.decide_fame()
    h.synchronize()
    dt = time.perf_counter() - t0
c = h.counters()
print("parts %d: %d members, %d events, %d rounds decided, %d iterations, pass %.2f ms (all parts' kernels one behind the other on ONE GPU)" % (
    parts, n, N, len(nc), c["round_iterations"] // 2, dt * 1e3))
