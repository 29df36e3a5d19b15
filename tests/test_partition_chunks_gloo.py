"""CPU, world_size 2 and 3 over gloo: the can_see table partitioned by EVENT RANGES
(py-swirld_amd/partition.py chunk_cuts / RowExchange) — rank k sweeps chunk k with the statement
k_cansee_chunks implements (tests/model_chunks.py: halo, unknown parents as leaves) WITHOUT any
communication, then the chunks are repaired in rank order, every rank fetching only the final rows
its provisional entries name.  The assembled table must equal the sequential rows
(swirld.py:198-205, 220); at uniform gossip with an ample halo no row moves at all."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

# n, N, seed, mode, p0, p1, halo
CASES = [(16, 3000, 21, 0, 0.0, 0.0, 40 * 16), (16, 3000, 21, 0, 0.0, 0.0, 0), (24, 4000, 22, 2, 0.4, 50.0, 200),
         (12, 2500, 23, 3, 0.3, 0.0, 30), (20, 3000, 24, 1, 0.02, 0.0, 100)]


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
    import torch.distributed as dist
    pkg = importlib.import_module("py-swirld_amd")
    part = importlib.import_module("py-swirld_amd.partition")
    import model_chunks as mc
    dist.init_process_group("gloo")
    out = []
    for n, N, seed, mode, p0, p1, halo in CASES:
        cr, sp, op = (np.asarray(x) for x in pkg.synth_hashgraph(n, N, seed, mode, p0, p1)[:3])
        cuts = part.chunk_cuts(0, N, world)
        a, b = cuts[rank], cuts[rank + 1]
        w = max(0, a - halo) if rank else 0
        L = np.full((N, n), -1, np.int32)               # this rank fills rows [a, b) only
        prov = mc.local_sweep(n, cr, sp, op, L, 0, w, a, b)      # no communication
        ex = part.RowExchange(dist, rank, world, cuts)
        for k in range(1, world):                        # repairs in rank order: lower ranks are final
            need = mc.fixup_needs(n, cr, L, 0, w, a, b) if (k == rank and prov) else []
            got = ex.fetch(need, lambda e: L[e])
            if k == rank and prov:
                mc.fixup(n, cr, L, 0, w, a, b, rows=lambda E: np.stack([got[int(e)] for e in E]))
        out.append((L[a:b].tobytes(), int(prov), int(ex.bytes_moved)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_table_partitioned_by_event_ranges_over_gloo(pkg, world):
    from model_chunks import cansee_sequential
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, (n, N, seed, mode, p0, p1, halo) in enumerate(CASES):
        cr, sp, op = (np.asarray(x) for x in pkg.synth_hashgraph(n, N, seed, mode, p0, p1)[:3])
        ref = cansee_sequential(n, cr, sp, op)
        table = b"".join(res[r][i][0] for r in range(world))
        assert table == ref.tobytes(), (n, N, mode, halo)
        moved = sum(res[r][i][2] for r in range(world))
        prov = sum(res[r][i][1] for r in range(world))
        if i == 0:   # uniform gossip, ample halo: nothing provisional, nothing moves
            assert prov == 0 and moved == 0
        if halo == 0:
            assert prov > 0 and moved > 0
        assert moved <= (world - 1) * N * n * 4      # never more than the lower ranks' tables
