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
  The kernel trace that comes with the FETCH pass also gives every family's average launch duration as rocprofv3
  sees it under the counters (`avg_us`: kernels serialised).  `ms_per_pass` comes from the PLAIN kernel trace of the bench
  command (kernel_stats.txt of profiles/run_profiles.sh step 1, given with --stats): a family's total time / the passes of
  that run (= launches of the elections kernel, one per pass) — streams overlapping as in a timed step.  bench.py ranks the
  families by it (its hipEvent brackets around plain launches inflate the short loop kernels by 2-3 us each).
  The file is keyed by WORKLOAD ("<members>x<events>x<generator mode>"): an existing file is extended, entries
  measured on other kernel source (SHA-256 of kernels.hip.h + order.hip.h) are dropped.
Usage: collect_traffic.py <fetch_dir> <write_dir> <out.json> <commit> <workload-key> <command> <kernel-substring>..."""
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


def kernel_source_sha256(root):
    import hashlib
    h = hashlib.sha256()
    for name in ("kernels.hip.h", "order.hip.h"):
        with open(os.path.join(root, "py-swirld_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def avg_duration_us(d, kernel):
    """average duration of the launches that did work (> 3 us: iterations replayed behind a finished loop return at once)"""
    fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not fs:
        return None
    df = pd.read_csv(fs[0])
    kn = [c for c in df.columns if c.lower() == "kernel_name"][0]
    a = [c for c in df.columns if c.lower() in ("start_timestamp", "start")][0]
    b = [c for c in df.columns if c.lower() in ("end_timestamp", "end")][0]
    sel = df[df[kn].str.contains(kernel, regex=False)]
    if not len(sel):
        return None
    dur = (sel[b] - sel[a]) / 1e3
    live = dur[dur > 3.0]
    return round(float(live.mean()), 2) if len(live) else round(float(dur.mean()), 2)


def per_pass_from_stats(path, kernels):
    """{family: ms per pass} from a summarize_rocpd.py table of a bench.py kernel trace"""
    rows = {}
    for ln in open(path):
        parts = ln.rstrip("\n").split()
        if len(parts) < 9 or not parts[-1].replace(".", "").isdigit():
            continue
        name = " ".join(parts[:-8])
        try:
            rows[name] = (int(parts[-8]), float(parts[-7]))   # calls, total_us
        except ValueError:
            continue
    passes = sum(c for nme, (c, _t) in rows.items() if "k_elections" in nme)
    if not passes:
        return {}
    out = {}
    for k in kernels:
        tot = sum(t for nme, (_c, t) in rows.items() if k in nme)
        if tot:
            out[k] = round(tot / passes / 1e3, 4)
    return out


def main(fd, wd, out, commit, key, workload, kernels, stats=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ksha = kernel_source_sha256(root)
    allw = {"kernels_sha256": ksha, "workloads": {},
            "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; per kernel family the "
                      "mean over the launches that did work; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 "
                      "(FETCH_SIZE doubled per MI355X_MICROARCH.md HBM note); avg_us = rocprofv3's own duration of the "
                      "launches that did work, from the kernel trace of the FETCH pass"}
    try:
        old = json.load(open(out))
        if old.get("kernels_sha256") == ksha:
            allw["workloads"] = old.get("workloads", {})
    except (OSError, ValueError):
        pass
    res = {"commit": commit, "workload": workload, "kernels": {}, "avg_us": {}, "ms_per_pass": per_pass_from_stats(stats, kernels) if stats else {}, "detail": {}}
    allw["workloads"][key] = res
    for k in kernels:
        f, w = per_kernel(fd, "FETCH_SIZE", k), per_kernel(wd, "WRITE_SIZE", k)
        if f is None or w is None:
            continue
        res["kernels"][k] = int((2 * f["mean_live_KiB"] + w["mean_live_KiB"]) * 1024)
        res["avg_us"][k] = avg_duration_us(fd, k)
        res["detail"][k] = {"fetch_KiB_per_launch_raw": round(f["mean_live_KiB"], 1),
                            "write_KiB_per_launch_raw": round(w["mean_live_KiB"], 1),
                            "launches": f["dispatches"], "live_launches": f["live"],
                            "hbm_bytes_per_launch_uncorrected": int((f["mean_live_KiB"] + w["mean_live_KiB"]) * 1024),
                            "fetch_total_KiB_raw": round(f["sum_KiB"], 1), "write_total_KiB_raw": round(w["sum_KiB"], 1)}
    json.dump(allw, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    a = sys.argv[1:]
    st = None
    if "--stats" in a:
        i = a.index("--stats")
        st = a[i + 1]
        a = a[:i] + a[i + 2:]
    main(a[0], a[1], a[2], a[3], a[4], a[5], a[6:], st)
