#!/usr/bin/env python3
"""Timeline of the kernels of the LAST find_order call in a rocprofv3 kernel trace (rocpd sqlite): start and end relative to the
call's first kernel, stream, name.  Usage: python profiles/order_timeline.py <results.db>"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    rows = list(db.execute("select s.%s, d.start, d.end, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)))
    preps = [i for i, r in enumerate(rows) if r[0].startswith("k_order_prep")]
    if not preps:
        print("no find_order call in the trace")
        return
    first = preps[-1]
    t0 = rows[first][1]
    last_end = t0
    print("%10s %10s %8s  %6s  %s" % ("start_us", "end_us", "dur_us", "stream", "kernel"))
    for name, a, b, st in rows[first:]:
        print("%10.1f %10.1f %8.1f  %6s  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, st, name[:60]))
        last_end = max(last_end, b)
    print("# first kernel start -> last kernel end: %.1f us" % ((last_end - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
