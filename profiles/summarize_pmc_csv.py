#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from its counter_collection CSV
(--output-format csv).  Usage: summarize_pmc_csv.py <dir-or-csv> [...]"""
import glob
import os
import sys

import pandas as pd


def main(paths):
    files = []
    for p in paths:
        files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(p) else [p]
    for f in files:
        df = pd.read_csv(f)
        kn = [c for c in df.columns if c.lower() == "kernel_name"][0]
        cn = [c for c in df.columns if c.lower() == "counter_name"][0]
        cv = [c for c in df.columns if c.lower() == "counter_value"][0]
        df[kn] = df[kn].str.replace(r"\(.*", "", regex=True).str.slice(0, 48)
        g = df.groupby([kn, cn])[cv].agg(["count", "mean", "sum"]).reset_index()
        print("# %s" % os.path.basename(f))
        piv = g.pivot(index=kn, columns=cn, values="mean")
        cnt = g.groupby(kn)["count"].max()
        piv.insert(0, "dispatches", cnt)
        with pd.option_context("display.width", 250, "display.max_columns", 40, "display.float_format", "{:.4g}".format):
            print(piv.sort_values("dispatches", ascending=False).to_string())
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
