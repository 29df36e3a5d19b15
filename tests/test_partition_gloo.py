"""CPU, world_size 2 over gloo: the candidate-partitioned decide_fame of py-swirld_amd/partition.py
(elections split by candidate round, one all-reduce(MAX) of the fame table) driving the numpy
restatement of the kernels — NOT the oracle — on every rank; the merged result must be identical on
both ranks, identical to the single-rank model, and identical to the sequential oracle."""
import importlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT

CASES = [(8, 2500, 11, 0, 0.0, 0.0), (4, 1500, 12, 0, 0.0, 0.0), (12, 3000, 13, 2, 0.25, 0.05), (16, 3000, 14, 1, 0.02, 0.0)]


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
    from model_backend import ModelHashgraph
    dist.init_process_group("gloo")
    out = []
    for n, N, seed, mode, p0, p1 in CASES:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        h = ModelHashgraph(n, stream)                      # replicated divide_rounds (model of the kernels)
        pf = part.PartitionedFame(dist, rank, world)
        new_c = pf.decide_fame(h)                          # partitioned elections + all-reduce + commit
        out.append((new_c, h.famous.tobytes(), h.consensus.tobytes(), h.rnd.tobytes(), h.L.tobytes()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_fame_two_ranks_over_gloo(pkg):
    from model_backend import ModelHashgraph
    from oracle.oracle import Oracle
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1], "every rank ends with the same famous / consensus / new_c"
    for i, (n, N, seed, mode, p0, p1) in enumerate(CASES):
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        new_c, fam_b, cons_b, rnd_b, L_b = res[0][i]
        single = ModelHashgraph(n, stream)
        assert single.decide_fame() == new_c and single.famous.tobytes() == fam_b and single.consensus.tobytes() == cons_b
        o = Oracle(n)
        o.append_events(*stream)
        o.divide_rounds(0, N)
        nco = [int(r) for r in o.decide_fame()]
        assert nco == new_c
        assert o.round.tobytes() == rnd_b and o.can_see.tobytes() == L_b
        assert o.famous_table().tobytes() == fam_b and o.consensus().tobytes() == cons_b
        assert len(new_c) > 3


def test_partition_helpers(pkg):
    part = importlib.import_module("py-swirld_amd.partition")
    assert part.candidate_rounds(3, 10, 1, 3) == [4, 7]
    owned = sorted(r for p in range(4) for r in part.candidate_rounds(2, 17, p, 4))
    assert owned == list(range(2, 17))
    a = (np.array([[1, -1], [-1, -1]], np.int8), np.array([1, 0], np.uint8))
    b = (np.array([[-1, -1], [0, 1]], np.int8), np.array([0, 1], np.uint8))
    fam, dec = part.merge_fame_tables([a, b])
    assert fam.tolist() == [[1, -1], [0, 1]] and dec.tolist() == [1, 1]
    assert part.chunk_cuts(0, 10, 3) == [0, 3, 6, 10] and part.chunk_cuts(5, 9, 2) == [5, 7, 9]
