#!/usr/bin/env python3
"""Per-launch HBM traffic of every kernel family of the path from two rocprofv3 --pmc CSV passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as MI355X_MICROARCH.md §HBM prescribes) ->
profiles/traffic.json, which bench.py quotes next to the algorithmic bytes.
  FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  gfx950 correction: FETCH_SIZE counts
  64 B per 128-B request for wide coalesced streams, i.e. it under-reports such reads by 2x; for
  narrower gathers the factor is uncalibrated, so both the raw and the doubled figure are kept
  and bench.py reports the conservative (doubled) one.
  Launches that did no work (iterations replayed after the loop finished) are excluded: only
  dispatches above 5 % of the family's maximum count.
Usage: collect_traffic.py <fetch_dir> <write_dir> <out.json> <commit> <workload> <kernel-substring>..."""
import glob
import json
import os
import sys

import pandas as pd


def per_kernel(d, counter, kernel):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    df = pd.read_csv(f)
    kn = [c for c in df.columns if c.lower() == "kernel_name"][0]
    cn = [c for c in df.columns if c.lower() == "counter_name"][0]
    cv = [c for c in df.columns if c.lower() == "counter_value"][0]
    sel = df[df[kn].str.contains(kernel, regex=False) & (df[cn] == counter)][cv]
    if not len(sel):
        return None
    live = sel[sel > sel.max() * 0.05]
    return {"dispatches": int(sel.count()), "live": int(live.count()), "mean_live_KiB": float(live.mean()) if len(live) else 0.0,
            "sum_KiB": float(sel.sum())}


def main(fd, wd, out, commit, workload, kernels):
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "py-swirld_amd", "csrc", "kernels.hip.h"), "rb") as fh:
        ksha = hashlib.sha256(fh.read()).hexdigest()
    res = {"commit": commit, "kernels_sha256": ksha, "workload": workload, "kernels": {}, "detail": {},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; per kernel family the "
                     "mean over the launches that did work; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 "
                     "(FETCH_SIZE doubled per MI355X_MICROARCH.md HBM note)"}
    for k in kernels:
        f, w = per_kernel(fd, "FETCH_SIZE", k), per_kernel(wd, "WRITE_SIZE", k)
        if f is None or w is None:
            continue
        res["kernels"][k] = int((2 * f["mean_live_KiB"] + w["mean_live_KiB"]) * 1024)
        res["detail"][k] = {"fetch_KiB_per_launch_raw": round(f["mean_live_KiB"], 1),
                            "write_KiB_per_launch_raw": round(w["mean_live_KiB"], 1),
                            "launches": f["dispatches"], "live_launches": f["live"],
                            "hbm_bytes_per_launch_uncorrected": int((f["mean_live_KiB"] + w["mean_live_KiB"]) * 1024),
                            "fetch_total_KiB_raw": round(f["sum_KiB"], 1), "write_total_KiB_raw": round(w["sum_KiB"], 1)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6:])
