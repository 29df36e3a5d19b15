"""GPU (-m gpu): the kernel side of the candidate-partitioned decide_fame (sw_decide_fame_partial /
sw_commit_fame) on real hardware: `nparts` contexts on one device each run their share of the
elections; the element-wise MAX of their tables (what the all-reduce of py-swirld_amd/partition.py
computes) committed on every context must equal the plain sw_decide_fame and the oracle —
batch and incremental schedules, coin rounds included."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,chunk,nparts", [
    (64, 40000, 501, 0, 0, 0, None, 2), (4, 4000, 502, 0, 0, 0, 300, 3), (130, 20000, 503, 2, 0.2, 0.05, 6000, 4),
    (256, 60000, 504, 0, 0, 0, None, 8), (16, 8000, 505, 1, 0.02, 0, 1000, 2),
    (300, 30000, 506, 2, 0.35, 0.02, 8000, 3), (520, 30000, 507, 0, 0, 0, None, 2)])   # (more than 256 members: k_elections_wide)
def test_partitioned_fame_equals_decide_fame(pkg, n, N, seed, mode, p0, p1, chunk, nparts):
    from oracle.oracle import Oracle
    part = importlib.import_module("py-swirld_amd.partition")
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    parts = [pkg.Hashgraph(n) for _ in range(nparts)]
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in [o] + parts:
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco = [int(r) for r in o.decide_fame()]
        tables = [h.decide_fame_partial(p, nparts) for p, h in enumerate(parts)]
        fam, dec = part.merge_fame_tables(tables)          # = all-reduce(MAX)
        for h in parts:
            assert [int(r) for r in h.commit_fame(fam, dec)] == nco
    wit = parts[0].witnesses()
    m = wit >= 0
    for h in parts:
        assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
        assert np.array_equal(h.consensus(), o.consensus())
        h.close()
