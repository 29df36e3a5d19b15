"""Test scaffolding: the event-range split of the can_see table behind the backend interface of
py-swirld_amd/partition.py StrongSplit (cansee_range / cansee_repair / export_rows / import_rows /
divide_rounds / decide_fame_partial / commit_fame), stated with the numpy model of the chunked sweep
(tests/model_chunks.py) and the numpy restatement of the kernels (tests/model_backend.py) — so that the
protocol (who sweeps, what is broadcast in which order, when a range may be repaired and divided) runs
on CPU ranks over gloo.  Not the oracle, not the product."""
import numpy as np
import torch

import model_chunks as mc
from model_backend import ModelHashgraph


class ModelRangeBackend:
    def __init__(self, n, stream, halo):
        self.n, self.stream, self.halo = n, stream, halo
        self.cr, self.sp, self.op = (np.asarray(x) for x in stream[:3])
        self.N = len(self.cr)
        self.L = np.full((self.N, n), -1, np.int32)
        self.present = np.zeros(self.N, bool)
        self.prov = {}
        self.divided = 0
        self.model = None
        self.stats = dict(prov=0, fixed=0, rows_imported=0)

    def row_buffer(self, K):
        return torch.empty(int(K) * self.n, dtype=torch.int32)

    def cansee_range(self, a, K):
        w = max(0, a - self.halo)
        assert not self.present[a:a + K].any()
        self.prov[(a, K)] = mc.local_sweep(self.n, self.cr, self.sp, self.op, self.L, 0, w, a, a + K)
        self.stats["prov"] += self.prov[(a, K)]
        self.present[a:a + K] = True

    def cansee_repair(self, a, K):
        assert self.present[:a].all(), "the rows below a range are imported before it is repaired"
        if self.prov[(a, K)]:
            self.stats["fixed"] += mc.fixup(self.n, self.cr, self.L, 0, max(0, a - self.halo), a, a + K)

    def export_rows(self, a, K, buf):
        assert self.present[a:a + K].all()
        buf.copy_(torch.from_numpy(self.L[a:a + K].reshape(-1)))

    def import_rows(self, a, K, buf):
        assert not self.present[a:a + K].any()
        self.L[a:a + K] = buf.numpy().reshape(K, self.n)
        self.present[a:a + K] = True
        self.stats["rows_imported"] += K

    def divide_rounds(self, a, K):
        assert a == self.divided and self.present[a:a + K].all(), "the round loop finds the rows of its range in place"
        self.divided = a + K
        if self.divided == self.N:   # the voting state of the whole hashgraph (the model computes it in one piece)
            self.model = ModelHashgraph(self.n, self.stream)
            assert np.array_equal(self.model.L, self.L), "the assembled table equals the rows of a single sweep"

    def decide_fame_partial(self, part, nparts):
        return self.model.decide_fame_partial(part, nparts)

    def commit_fame(self, fam, dec):
        return self.model.commit_fame(fam, dec)
