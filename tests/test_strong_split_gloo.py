"""CPU, world_size 2 and 3 over gloo: py-swirld_amd/partition.py StrongSplit — ONE hashgraph over the ranks:
rank k sweeps the can_see rows of its event range from a halo (no communication), the ranges are broadcast in
ascending order as int32 tensors (async, all enqueued up front), a range with provisional entries is repaired
once the rows below it have arrived, the round loop then runs range after range on rows it finds in place,
and decide_fame is candidate-partitioned with one all-reduce.  The backend is the numpy model of the chunked
sweep + kernels (tests/model_range_backend.py); every rank must end with the state of a single-rank run,
which must equal the sequential oracle."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

# n, N, seed, mode, p0, p1, halo
CASES = [(8, 2400, 31, 0, 0.0, 0.0, 40 * 8), (12, 3000, 32, 2, 0.3, 0.05, 60), (10, 2000, 33, 3, 0.4, 0.0, 0), (16, 2600, 34, 1, 0.03, 0.0, 50)]


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
    from model_range_backend import ModelRangeBackend
    dist.init_process_group("gloo")
    out = []
    for n, N, seed, mode, p0, p1, halo in CASES:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        b = ModelRangeBackend(n, stream, halo)
        ss = part.StrongSplit(dist, rank, world)
        cuts = ss.divide_rounds(b, N)
        new_c = ss.decide_fame(b)
        out.append((cuts, new_c, b.L.tobytes(), b.model.rnd.tobytes(), b.model.famous.tobytes(), b.model.consensus.tobytes(), b.stats))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strong_split_over_gloo(pkg, world):
    from oracle.oracle import Oracle
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, (n, N, seed, mode, p0, p1, halo) in enumerate(CASES):
        for r in range(1, world):
            assert res[r][i][:6] == res[0][i][:6], "every rank ends with the same table, rounds, fame, consensus, new_c"
        cuts, new_c, L_b, rnd_b, fam_b, cons_b, _ = res[0][i]
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o = Oracle(n)
        o.append_events(*stream)
        o.divide_rounds(0, N)
        assert [int(r) for r in o.decide_fame()] == new_c and len(new_c) > 2
        assert o.can_see.tobytes() == L_b and o.round.tobytes() == rnd_b
        assert o.famous_table().tobytes() == fam_b and o.consensus().tobytes() == cons_b
        stats = [res[r][i][6] for r in range(world)]
        assert all(s["rows_imported"] == N - (cuts[r + 1] - cuts[r]) for r, s in enumerate(stats))
        if i == 0:      # uniform gossip, ample halo: nothing provisional
            assert sum(s["prov"] for s in stats) == 0
        if halo == 0:   # no halo at all: the repair does real work
            assert sum(s["prov"] for s in stats) > 0 and sum(s["fixed"] for s in stats) > 0


def test_row_exchange_padded_tensors_single_rank():
    """RowExchange's collectives are all_gathers of padded int32 tensors; world 1 exercises the packing."""
    import torch.distributed as dist
    part = importlib.import_module("py-swirld_amd.partition")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ex = part.RowExchange(dist, 0, 1, [0, 10])
        assert ex.fetch([], lambda e: np.zeros(4, np.int32)) == {}
        assert ex.fetch([3, 5], lambda e: np.full(4, e, np.int32)) == {}   # own rows are never fetched
    finally:
        dist.destroy_process_group()
