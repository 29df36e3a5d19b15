#!/usr/bin/env python3
"""GPU-side timeline of main()-style calls from a rocprofv3 kernel trace (rocpd sqlite): for a few calls in
the middle of the run, every kernel with its start offset, duration and stream.
Usage: python profiles/small_call_timeline.py <results.db> [first_call] [n_calls]"""
import sqlite3
import sys


def main(path, first=500, ncalls=2):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    nc = "display_name" if "display_name" in scols else "kernel_name"
    rows = list(db.execute("select s.%s, d.start, d.end, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (nc, disp, sym)))
    # a call starts with k_ingest_small
    starts = [i for i, r in enumerate(rows) if "k_ingest_small" in r[0]]
    if len(starts) < first + ncalls + 1:
        first = max(0, len(starts) - ncalls - 1)
    for c in range(first, first + ncalls):
        a, b = starts[c], starts[c + 1]
        t0 = rows[a][1]
        print("---- call %d: %d kernels, %.1f us from first kernel start to the next call's first kernel" % (c, b - a, (rows[b][1] - t0) / 1e3))
        for n, s, e, st in rows[a:b]:
            print("  +%8.1f us  %7.1f us  stream %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n.split("(")[0][:60]))
    per = [(rows[starts[i + 1]][1] - rows[starts[i]][1]) / 1e3 for i in range(len(starts) - 1)]
    busy = []
    for i in range(len(starts) - 1):
        busy.append(sum((e - s) for _, s, e, _ in rows[starts[i]:starts[i + 1]]) / 1e3)
    import statistics
    print("calls %d: period median %.1f us, kernel-busy median %.1f us, kernels per call median %d" % (
        len(per), statistics.median(per), statistics.median(busy), statistics.median([starts[i + 1] - starts[i] for i in range(len(starts) - 1)])))


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:4]))
