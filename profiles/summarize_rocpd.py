#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel
summary table `rocprofv3 --stats` would print: calls, total, average, min, max, share.
Usage: python profiles/summarize_rocpd.py <results.db> [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    # "live" = dispatches longer than 3 us: the round-loop kernels guard on device-side state, and
    # the few iterations replayed after the loop finished return immediately (~1 us)
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
         "sum(case when d.end - d.start > 3000 then 1 else 0 end), "
         "sum(case when d.end - d.start > 3000 then d.end - d.start else 0 end) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-72s %8s %12s %10s %10s %10s %7s %8s %11s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "live", "live_avg_us"))
    for name, calls, tot, mn, mx, live, live_tot in rows:
        name = name if len(name) <= 72 else name[:69] + "..."
        print("%-72s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%% %8d %11.2f" % (
            name, calls, tot / 1e3, tot / 1e3 / calls, mn / 1e3, mx / 1e3, 100.0 * tot / total,
            live, (live_tot / 1e3 / live) if live else 0.0))
    print("# dispatches: %d, total kernel time %.3f ms; columns in %s: %s" % (
        sum(r[1] for r in rows), total / 1e6, disp.split('_0')[0], ",".join(cols[:12])))


if __name__ == "__main__":
    main(sys.argv[1])
