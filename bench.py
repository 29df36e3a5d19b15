#!/usr/bin/env python3
"""bench.py — events/sec through divide_rounds + decide_fame (BASELINE.json metric).

One "step" = one pass of the hot path (sw_divide_rounds + sw_decide_fame through the
C-ABI) over one synthetic hashgraph whose events are ALREADY resident in HBM
(sw_append_events is ingest, mirrors Node.add_event, and is outside the timed region).
Workload at N=1: BASELINE.json configs[2] — 256 members, 1M events, uniform gossip
(SURVEY.md §8d generator), seed 3.  With --gpus N every rank runs an independent replica
of that workload (different seed): the path does not shard across GPUs (DESIGN.md §(e),
"replicas only"), so scaling is weak and there is no data-path collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(n, counters, n_events):
    """divide_rounds 12n + n^2/8 + 24 per non-root event;
    decide_fame V*(4n + n^2/8) + P2*n/8."""
    dr = (n_events - n) * (12 * n + n * n // 8 + 24)
    df = counters["voter_evals"] * (4 * n + n * n // 8) + counters["majority_evals"] * (n // 8)
    return dr, df


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=256)
    ap.add_argument("--events", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=300_000,
                    help="events of the same stream timed through the CPU oracle (0 = skip)")
    ap.add_argument("--contexts", type=int, default=4, help="max resident replicas of the DAG per GPU")
    ap.add_argument("--mode", type=int, default=0, help="generator mode (0 uniform gossip = the benchmark; 1 cliques, 2 slow members, 3 stale other-parents: robustness runs)")
    ap.add_argument("--p0", type=float, default=0.0)
    ap.add_argument("--p1", type=float, default=0.0)
    args = ap.parse_args()

    import torch
    rep_mod = importlib.import_module("py-swirld_amd.replicas")
    rank, local_rank, world = rep_mod.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # N > 1: independent replicas, one per GPU; RCCL only for the barrier and the max-over-ranks time
    rep = rep_mod.Replicas(backend="nccl", device=torch.device("cuda", local_rank))

    pkg = importlib.import_module("py-swirld_amd")
    n, N = args.members, args.events
    stream = pkg.synth_hashgraph(n, N, rep_mod.replica_seed(args.seed, rank), args.mode, args.p0, args.p1)  # host, untimed
    n_ctx = max(1, min(args.contexts, args.steps + args.warmup))
    ctxs = []
    t_ing0 = time.perf_counter()
    for _ in range(n_ctx):
        h = pkg.Hashgraph(n, device=local_rank)
        h.reserve(N)
        h.append_events(*stream)  # ingest (Node.add_event): untimed
        ctxs.append(h)
    ingest_s = (time.perf_counter() - t_ing0) / n_ctx
    for h in ctxs:  # set-up: every resident context builds its launch graphs once, then forgets the results
        h.divide_rounds(0, N)
        h.decide_fame()
        h.rewind()

    def one_step(i):
        h = ctxs[i % n_ctx]
        if i >= n_ctx:
            h.rewind()  # inside the timed bracket when a context is reused
        h.divide_rounds(0, N)
        return h.decide_fame()

    def barrier():
        rep.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        new_c = one_step(i)
    barrier()
    dt = rep.max_over_ranks(time.perf_counter() - t0)
    ms_per_step = dt / args.steps * 1e3
    value = world * N * args.steps / dt

    # ---- roofline of the dominant kernel (k_tally_candidates), one extra profiled pass ----
    h = ctxs[0]
    h.rewind()
    h.set_profiling(True)
    c0 = h.counters()
    h.divide_rounds(0, N)
    tm_dr = h.timings()
    new_c_prof = h.decide_fame()
    tm = h.timings()
    c1 = h.counters()
    h.set_profiling(False)
    t_fo = time.perf_counter()
    ordered = h.find_order(new_c_prof)   # N1 (outside the metric): reported for information only
    find_order_ms = (time.perf_counter() - t_fo) * 1e3
    cdelta = {k: c1[k] - c0[k] for k in c1}
    cdelta_far = cdelta.get("far_hops", 0)
    evals = c1["tally_evals"] - c0["tally_evals"]
    launches = max(1, tm_dr["tally_launches"])
    bytes_per_eval = 4 * n + n * n // 8 + 8  # one can_see row + n gathered n-bit masks + result
    avg_launch_ms = tm_dr["tally_ms"] / launches
    achieved = (evals / launches) * bytes_per_eval / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("tally_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    dr_b, df_b = algorithmic_bytes(n, cdelta, N)
    roofline = {
        "bound": "hbm", "kernel": "k_tally_bits", "achieved": round(achieved, 2), "peak": 8000.0,
        "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
        "avg_launch_us": round(avg_launch_ms * 1e3, 2), "launches": launches,
        "evals_per_launch": round(evals / launches, 1), "bytes_per_eval": bytes_per_eval,
        "far_hops": cdelta_far,
        "path_algorithmic_GBps": round((dr_b + df_b) / (ms_per_step * 1e-3) / 1e9, 2),
        "phase_ms": {k: round(v, 3) for k, v in (("can_see", tm_dr["can_see_ms"]), ("rounds", tm_dr["rounds_ms"]),
                                                 ("tally", tm_dr["tally_ms"]), ("aux_finalize_voter_span", tm_dr["finalize_ms"]),
                                                 ("fame", tm["fame_ms"]))},
        "phase_note": "can_see and aux spans run on their own streams and overlap the round loop",
    }

    # ---- CPU baseline: the oracle (C port of the reference algorithm), 1 core, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle.oracle import Oracle
        M = min(args.cpu_sample, N)
        o = Oracle(n)
        o.append_events(*[a[:M] for a in stream])
        tc0 = time.perf_counter()
        o.divide_rounds(0, M)
        o.decide_fame()
        tc = time.perf_counter() - tc0
        cpu_baseline = {"value": round(M / tc, 1), "unit": "events/s", "cores": 1, "kind": "port",
                        "host_cores": os.cpu_count(),
                        "sample": "first %d events of the same stream through oracle/swirld_oracle.c "
                                  "(sequential C restatement of swirld.py:187-277), %.1f s" % (M, tc)}

    if rank == 0:
        out = {
            "metric": "events/sec through divide_rounds+decide_fame", "value": round(value, 1),
            "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%d members, %d events, %s hashgraph, one batch "
                                   "divide_rounds + decide_fame per step" % (
                                       n, N, ["uniform-gossip", "two-clique", "slow-member", "stale-other-parent"][args.mode]),
                       "members": n, "events": N, "seed": args.seed,
                       "parallelism": "replicas x%d (no data-path collective)" % world,
                       "rounds": c1["rounds"], "ingest_s_untimed": round(ingest_s, 3),
                       "new_c_last_step": int(len(new_c)),
                       "find_order_ms_untimed": round(find_order_ms, 2), "events_ordered": int(len(ordered))},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    rep.close()


if __name__ == "__main__":
    main()
