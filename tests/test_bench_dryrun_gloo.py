"""CPU, two processes over gloo: bench.py's multi-rank flow (`--gpus 2`) end to end — replicas figure, the one-hashgraph split
(`value_strong`) through partition.StrongSplit's real collectives, the ONE JSON line from rank 0 — with the CPU oracle standing
in for the device and the numpy model of the event-range sweep behind the range backend (tests/model_range_backend.py).  And
the failure the guard exists for: a rank that never joins a collective of the split costs the run `value_strong`, not the line."""
import importlib
import io
import json
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, hang_rank, timeout_s, q, members=12, events=3000):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import threading

    import torch
    pkg = importlib.import_module("py-swirld_amd")
    part = importlib.import_module("py-swirld_amd.partition")
    bench = importlib.import_module("bench")
    from model_range_backend import ModelRangeBackend
    from test_bench_dryrun import BenchStandIn

    class StandIn(BenchStandIn):
        def rewind(self):
            super().rewind()
            hook = getattr(self, "_on_rewind", None)
            if hook:
                hook()

        def range_stats(self):
            return 0, 0, 0

        @staticmethod
        def split_link(parts):      # (the stand-in parts each divide the whole hashgraph: the flow of bench.py is what runs here)
            assert len({id(p) for p in parts}) == len(parts) >= 2

    class DryRange(ModelRangeBackend):
        """the numpy model of the event-range sweep where bench.py would put the host-staged HIP backend"""
        def __init__(self, hs, dev):
            self._hs = hs
            hs._on_rewind = self._again
            self._again()

        def _again(self):
            ModelRangeBackend.__init__(self, self._hs.n, tuple(self._hs._stream[0]), halo=40 * self._hs.n)

        def cansee_range(self, a, K):
            if rank == hang_rank:
                threading.Event().wait()      # this rank never reaches the collectives of the split
            ModelRangeBackend.cansee_range(self, a, K)

    pkg.Hashgraph = StandIn
    part.HostStagedRangeBackend = DryRange
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    sys.argv = ["bench.py", "--gpus", str(world), "--backend", "gloo", "--one-device", "--members", str(members), "--events", str(events),
                "--steps", "2", "--warmup", "1", "--contexts", "1", "--concurrent", "0", "--cpu-sample", "0", "--e2e-steps", "1",
                "--reference-events", "0", "--strong-timeout", str(timeout_s)]
    buf = io.StringIO()
    real = sys.stdout
    sys.stdout = buf
    try:
        if hang_rank >= 0:
            # (the guard ends the process from a timer thread: hand the captured line over before that happens)
            orig_exit = os._exit

            def leave(code):
                sys.stdout = real
                q.put((rank, buf.getvalue(), code))
                time.sleep(0.5)
                orig_exit(code)
            os._exit = leave
        bench.main()
    finally:
        sys.stdout = real
    q.put((rank, buf.getvalue(), 0))


def _run(world, hang_rank, timeout_s, members=12, events=3000):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hang_rank, timeout_s, q, members, events)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, code = q.get(timeout=300)
        res[r] = (out, code)
    for p in procs:
        p.join(timeout=60)
    return res, [p.exitcode for p in procs]


def test_two_rank_bench_line_with_the_one_hashgraph_split(pkg):
    res, codes = _run(2, -1, 120)
    assert codes == [0, 0]
    lines = [ln for ln in res[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in res[1][0].splitlines() if ln.startswith("{")]   # rank 0 prints, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak-replicas" and d["value"] > 0
    assert abs(d["value"] - 2 * 3000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01    # whole-job aggregate over both ranks
    s = d["strong"]
    assert s["parts"] == 2 and s["events_per_s"] > 0 and d["value_strong"] == s["events_per_s"] and s["new_c_last_step"] > 0
    assert "replicas x2" in d["config"]["parallelism"]


def test_two_rank_bench_line_beyond_256_members_rank_0_drives_the_linked_parts(pkg):
    """more than 256 members: the split is inside the round loop's iterations (sw_split_link) and its parts are contexts of
    ONE process — rank 0 links one per rank and drives them from a thread each, rank 1 waits at the barrier; the line carries
    `value_strong` from that run"""
    res, codes = _run(2, -1, 240, members=300, events=9000)
    assert codes == [0, 0]
    lines = [ln for ln in res[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in res[1][0].splitlines() if ln.startswith("{")]
    d = json.loads(lines[0])
    s = d["strong"]
    assert d["n_gpus"] == 2 and s["parts"] == 2 and s["events_per_s"] > 0 and d["value_strong"] == s["events_per_s"]
    assert s["one_gpu_per_part"] is False and "sw_split_link" in s["what"] and s["new_c_last_step"] >= 0


def test_a_rank_that_never_joins_the_split_costs_value_strong_not_the_line(pkg):
    t0 = time.time()
    res, codes = _run(2, 1, 6)
    assert codes == [0, 0] and time.time() - t0 < 120
    lines = [ln for ln in res[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["value_strong"] is None
    assert "did not finish within 6 s" in d["strong"]["error"]
