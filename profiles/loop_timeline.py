#!/usr/bin/env python3
"""Round-loop timeline from a rocprofv3 kernel trace (rocpd sqlite): per iteration, the
duration of k_resolve_band and of the tally kernel (k_tally_bits / k_tally_tree), the two dispatch gaps between them and the
iteration period.  Usage: python profiles/loop_timeline.py <results.db>"""
import sqlite3
import sys

import numpy as np


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    rows = list(db.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)))
    loop = [(("R" if "k_resolve_band" in n else "T"), s, e) for n, s, e in rows if "k_resolve_band" in n or "k_tally_bits" in n or "k_tally_tree" in n or "k_tally_candidates" in n]
    dr, dt, g_rt, g_tr, per = [], [], [], [], []
    for i in range(len(loop) - 2):
        a, b, c = loop[i], loop[i + 1], loop[i + 2]
        if a[0] == "R" and b[0] == "T" and c[0] == "R" and a[2] - a[1] > 3000 and b[2] - b[1] > 3000 and c[1] - b[2] < 50000:
            dr.append(a[2] - a[1]); dt.append(b[2] - b[1]); g_rt.append(b[1] - a[2]); g_tr.append(c[1] - b[2]); per.append(c[1] - a[1])
    def q(x):
        x = np.array(x) / 1e3
        return "n=%d mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (len(x), x.mean(), *np.percentile(x, [10, 50, 90]))
    print("k_resolve_band duration :", q(dr))
    print("gap resolve -> tally    :", q(g_rt))
    print("tally kernel duration   :", q(dt))
    print("gap tally -> resolve    :", q(g_tr))
    print("iteration period        :", q(per))


if __name__ == "__main__":
    main(sys.argv[1])
