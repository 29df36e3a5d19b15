"""CPU: bench.py's HOST logic end to end — argument handling, the timed bracket, the roofline / counters / traffic bookkeeping,
the guard around the one-hashgraph split and the ONE JSON line of the driver's contract — with the CPU oracle standing in for
the device (tests/oracle_backend.py) at a toy size.  Nothing here measures anything: it only keeps a mistake in the bench
script (an unbound name, a missing key) from costing the round's measurement."""
import importlib
import json
import sys

import numpy as np
import pytest

import oracle_backend


class BenchStandIn(oracle_backend.OracleHashgraph):
    """OracleHashgraph + the measurement entry points bench.py uses (rewind / reset / profiling / timings / counters)."""

    def __init__(self, n_members, stake=None, coin_period=6, device=0):
        super().__init__(n_members, stake, coin_period, device)
        self._args = (n_members, stake, coin_period)
        self._stream = []
        self._base = {}

    def reserve(self, n_events):
        pass

    def append_events(self, creator, self_parent, other_parent, t=None, sig=None):
        self._stream.append((creator, self_parent, other_parent, t, sig))
        super().append_events(creator, self_parent, other_parent, t, sig)

    def _fresh(self, keep_events):
        L = importlib.import_module("py-swirld_amd._lib")
        done = self.counters()
        self._base = {k: done.get(k, 0) for k, _ in L.Counters._fields_}   # counters are cumulative over rewinds
        stream = self._stream if keep_events else []
        self.__dict__.pop("_head", None)
        oracle_backend.OracleHashgraph.__init__(self, *self._args)
        self._stream = []
        for part in stream:
            self.append_events(*part)

    def rewind(self):
        self._fresh(True)

    def reset(self):
        self._fresh(False)

    def synchronize(self):
        pass

    def set_profiling(self, enable=True):
        pass

    def timings(self):
        L = importlib.import_module("py-swirld_amd._lib")
        return {k: (1 if k.endswith("_launches") else 1.0) for k, _ in L.Timings._fields_}

    def counters(self):
        L = importlib.import_module("py-swirld_amd._lib")
        c = super().counters()
        return {k: int(c.get(k, 0)) + int(self._base.get(k, 0)) for k, _ in L.Counters._fields_}

    tally_kernel = "k_tally_bits"


def test_bench_prints_the_contract_line(pkg, monkeypatch, capsys):
    import torch
    bench = importlib.import_module("bench")
    monkeypatch.setattr(pkg, "Hashgraph", BenchStandIn)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--members", "16", "--events", "6000", "--steps", "2", "--warmup", "1",
                                      "--contexts", "2", "--concurrent", "2", "--cpu-sample", "3000", "--e2e-steps", "1",
                                      "--reference-events", "0"])
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "events/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 6000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    assert "workload" in d["config"] and "timed_region" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernels", "path_frac", "hbm_bytes_per_step_pmc", "traffic_stale"):
        assert k in r, k
    assert r["traffic_stale"] is False                       # (profiles/traffic.json belongs to this tree's kernel source)
    assert r["frac"] == r["path_frac"] and r["bound"] == "hbm"   # the headline fraction is the whole path's
    assert r["traffic"] is None and "none" in r["traffic_source"]   # (no counter pass exists for a 16-member workload: nothing is quoted)
    assert r["dominant_kernel"]["served_by"].startswith("hbm")    # an L2-served family is never the kernel named under bound "hbm"
    assert {"k_resolve_band", "k_tally_bits", "k_elections"} <= {k["kernel"] for k in r["kernels"]}
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["strong"] is None and d["value_strong"] is None
    assert d["value_end_to_end"] > 0 and d["find_order_ms"] >= 0 and d["value_concurrent_contexts"]["contexts"] == 2
    assert d["roofline"]["counters"]["round_iterations"] >= 0 and "finalize_from_rows" in d["roofline"]["counters"]
