#!/usr/bin/env python3
"""Anatomy of ONE pass (sw_rewind + sw_divide_rounds + sw_decide_fame) from a rocprofv3 kernel trace (rocpd sqlite):
every kernel between two k_elections launches in start order, the round-loop replays folded into one line each.
Usage: python profiles/pass_timeline.py <results.db> [pass index, default 3]"""
import sqlite3
import sys


def main(path, which):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    nc = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    rows = list(db.execute("select s.%s, d.start, d.end, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (nc, disp, sym)))

    def short(n):
        return n.replace("void ", "").split("<")[0].split("(")[0]
    loopk = ("k_resolve_band", "k_tally_tree", "k_tally_bits", "k_tally_candidates")
    el = [i for i, r in enumerate(rows) if "k_elections" in r[0]]
    a, b = el[which - 1], el[which]
    t0 = rows[a][2]
    i = a + 1
    loop_us = 0.0
    while i <= b:
        n = short(rows[i][0])
        if n in loopk:
            j = i
            while j <= b and (short(rows[j][0]) in loopk or (rows[j][3] != rows[i][3])):
                j += 1
            ks = [r for r in rows[i:j] if short(r[0]) in loopk]
            others = [r for r in rows[i:j] if short(r[0]) not in loopk]
            print("%9.1f %9.1f  %8.1f us  s%d ROUND LOOP: %d kernels (%d did work)" % ((ks[0][1] - t0) / 1e3, (ks[-1][2] - t0) / 1e3, (ks[-1][2] - ks[0][1]) / 1e3, ks[0][3], len(ks), sum(1 for r in ks if r[2] - r[1] > 3000)))
            loop_us += (ks[-1][2] - ks[0][1]) / 1e3
            for r in others:
                print("%9.1f %9.1f  %8.1f us  s%d   (beside the loop) %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))
            i = j
        else:
            print("%9.1f %9.1f  %8.1f us  s%d %s" % ((rows[i][1] - t0) / 1e3, (rows[i][2] - t0) / 1e3, (rows[i][2] - rows[i][1]) / 1e3, rows[i][3], n))
            i += 1
    print("pass: %.1f us from the end of the previous elections kernel to the end of this one; round loops %.1f us" % ((rows[b][2] - t0) / 1e3, loop_us))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
