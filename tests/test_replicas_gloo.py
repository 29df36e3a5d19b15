"""CPU, world_size 2 over gloo: the multi-GPU driver path of bench.py (replica seeds,
barrier, max-over-ranks timing, whole-job aggregation).  The hot path itself needs a GPU;
here each 'replica' is the host-side generator plus the CPU oracle on a small DAG, which
also checks that different ranks really process different hashgraphs."""
import importlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    pkg = importlib.import_module("py-swirld_amd")
    rep_mod = importlib.import_module("py-swirld_amd.replicas")
    from oracle.oracle import Oracle
    rep = rep_mod.Replicas(backend="gloo")
    assert (rep.rank, rep.world) == (rank, world)
    n, N = 8, 1500
    stream = pkg.synth_hashgraph(n, N, rep_mod.replica_seed(3, rank))
    digest = []

    def step(i):
        o = Oracle(n)
        o.append_events(*stream)
        o.divide_rounds(0, N)
        o.decide_fame()
        digest.append(int(o.round.sum()))

    import time
    t0 = time.perf_counter()
    dt = rep.timed(step, 2)
    local = time.perf_counter() - t0
    value = rep.aggregate_throughput(N, 2, dt)
    q.put((rank, dt, local, value, digest[0], int(stream[2].sum())))
    rep.close()


def test_two_replicas_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, loc0, v0, d0, s0), (r1, dt1, loc1, v1, d1, s1) = out
    assert (r0, r1) == (0, 1)
    assert dt0 == dt1, "every rank must report the max over ranks"
    assert 0 < dt0 <= max(loc0, loc1) + 1e-3   # the timed region sits inside each rank's wall time
    assert np.isclose(v0, v1) and np.isclose(v0, 2 * 1500 * 2 / dt0)   # whole-job aggregate
    assert s0 != s1, "replicas must process different hashgraphs"
