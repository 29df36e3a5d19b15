"""GPU (-m gpu): the one-DAG multi-GPU split on real hardware (include/swirld_hip.h
sw_cansee_range / sw_cansee_repair / sw_export_rows / sw_import_rows, py-swirld_amd/partition.py StrongSplit).
(1) P contexts on ONE device stand for P ranks: every context sweeps its event range from a halo, the ranges
    travel through device buffers in ascending order, provisional entries are repaired, the round loop runs
    on rows it finds in place, decide_fame is candidate-partitioned — every context must equal the oracle.
(2) The real thing over torch.distributed: two PROCESSES (gloo moving CUDA tensors, both on cuda:0 — RCCL
    refuses two ranks on one device) run StrongSplit with asynchronous broadcasts."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_emulated(pkg, n, N, seed, mode, p0, p1, parts):
    import torch
    from oracle.oracle import Oracle
    part = importlib.import_module("py-swirld_amd.partition")
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    nco = [int(r) for r in o.decide_fame()]
    dev = torch.device("cuda", 0)
    hs = [pkg.Hashgraph(n) for _ in range(parts)]
    for h in hs:
        h.append_events(*stream)
    backs = [part.HipRangeBackend(h, dev) for h in hs]
    cuts = part.emulate_strong_split(backs, N)
    assert cuts[0] == 0 and cuts[-1] == N
    tables = [h.decide_fame_partial(p, parts) for p, h in enumerate(hs)]
    fam, dec = part.merge_fame_tables(tables)
    stats = []
    for h in hs:
        assert [int(r) for r in h.commit_fame(fam, dec)] == nco
        assert np.array_equal(h.rounds(), o.round)
        assert np.array_equal(h.witnesses(), o.witnesses())
        wit = h.witnesses()
        m = wit >= 0
        assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
        for a in range(0, N, 20000):   # every can_see row, in slices
            b = min(N, a + 20000)
            assert np.array_equal(h.can_see(a, b - a), o.can_see[a:b])
        stats.append(h.range_stats())
        h.close()
    return stats


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,parts", [
    (64, 60000, 601, 0, 0, 0, 2), (256, 120000, 602, 0, 0, 0, 4), (130, 50000, 603, 3, 0.5, 0, 3), (16, 20000, 604, 0, 0, 0, 8)])
def test_event_range_split_equals_oracle(pkg, n, N, seed, mode, p0, p1, parts):
    stats = _run_emulated(pkg, n, N, seed, mode, p0, p1, parts)
    if mode == 0 and n >= 64:
        assert all(s == (0, 0, 0) for s in stats), "uniform gossip, default halo: nothing provisional, nothing repaired"


@pytest.mark.parametrize("halo,n,N,seed,mode,p0,p1,parts", [
    ("0", 64, 40000, 611, 0, 0, 0, 3),          # no halo at all: every range is repaired (or swept again) from imported rows
    ("300", 48, 40000, 612, 2, 0.3, 0.02, 4),   # slow members: entries older than the halo
    ("2000", 256, 90000, 613, 2, 0.95, 0.002, 2)])   # hot members: ranges swept a second time
def test_event_range_split_repairs(pkg, monkeypatch, halo, n, N, seed, mode, p0, p1, parts):
    monkeypatch.setenv("SW_HALO", halo)
    monkeypatch.setenv("SW_CHUNK_MIN", "4096")
    stats = _run_emulated(pkg, n, N, seed, mode, p0, p1, parts)
    assert sum(s[0] for s in stats) > 0, "provisional entries were counted"
    assert sum(s[1] for s in stats) + sum(s[2] for s in stats) > 0, "... and repaired, or their ranges swept again"


def test_range_api_errors(pkg):
    import torch
    n, N = 16, 6000
    stream = pkg.synth_hashgraph(n, N, 620)
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    buf = torch.empty(3000 * h.row_stride, dtype=torch.int32, device="cuda")
    with pytest.raises(pkg.SwirldHipError):
        h.export_rows(0, 3000, buf.data_ptr())        # nothing computed yet
    h.cansee_range(3000, 3000)
    with pytest.raises(pkg.SwirldHipError):
        h.cansee_range(4000, 1000)                      # overlaps rows present
    with pytest.raises(pkg.SwirldHipError):
        h.cansee_repair(3000, 3000)                     # rows below are not here yet
    with pytest.raises(pkg.SwirldHipError):
        h.divide_rounds(0, 4000)                        # straddles the present rows
    h.divide_rounds(0, 3000)                            # sweeps [0, 3000) itself
    h.cansee_repair(3000, 3000)
    h.divide_rounds(3000, 3000)
    from oracle.oracle import Oracle
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    assert np.array_equal(h.rounds(), o.round) and np.array_equal(h.can_see(), o.can_see)
    h.rewind()                                          # a rewind forgets the present rows
    h.divide_rounds(0, N)
    assert np.array_equal(h.rounds(), o.round)
    big = pkg.Hashgraph(300)
    big.append_events(*pkg.synth_hashgraph(300, 2000, 621))
    with pytest.raises(pkg.SwirldHipError) as ei:
        big.cansee_range(0, 2000)
    assert ei.value.code == -95
    big.close()
    h.close()


WORKER = r"""
import importlib, os, sys
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
pkg = importlib.import_module("py-swirld_amd")
part = importlib.import_module("py-swirld_amd.partition")
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:   # gloo moving device tensors (built with HIP support?): probe before relying on it
    probe = torch.full((8,), rank, dtype=torch.int32, device=dev)
    dist.broadcast(probe, src=1, async_op=True).wait()
    dist.all_reduce(probe, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    assert int(probe[0]) == 1
    on_device = True
except Exception as exc:
    print("RANK %%d: gloo cannot move device tensors here (%%r): rows staged through host tensors" %% (rank, exc), flush=True)
    on_device = False


ok = True
for n, N, seed, mode, p0, p1 in [(64, 50000, 631, 0, 0, 0), (40, 30000, 632, 2, 0.3, 0.02)]:
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    ss = part.StrongSplit(dist, rank, world, device=dev if on_device else None)
    back = part.HipRangeBackend(h, dev) if on_device else part.HostStagedRangeBackend(h, dev)
    for step in range(2):                                # twice: the staging buffers are reused
        h.rewind()
        ss.divide_rounds(back, N)
        new_c = [int(r) for r in ss.decide_fame(back)]
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    nco = [int(r) for r in o.decide_fame()]
    wit = h.witnesses(); m = wit >= 0
    ok = ok and new_c == nco and np.array_equal(h.rounds(), o.round) and np.array_equal(h.can_see(), o.can_see) \
        and np.array_equal(h.famous()[m], o.famous_by_event[wit[m]]) and np.array_equal(h.consensus(), o.consensus())
    h.close()
print("RANK %%d %%s (rows travelled as %%s tensors)" %% (rank, "OK" if ok else "MISMATCH", "device" if on_device else "host"), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_strong_split_two_processes_one_gpu(pkg, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q_ in procs:
                q_.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in out, out[-3000:]
