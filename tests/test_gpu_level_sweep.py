"""GPU (-m gpu): the level-bucketed can_see sweep (k_level_hist / _scatter / _patch + k_cansee_stream) against
the oracle — the default sweep beyond 256 members, SW_CANSEE_IMPL=2 below.  Small hashgraphs with shallow rings, so that
every path runs: other-parents that leave the ring before their child's level (pinned into the side table at scatter time),
the side table wrapping round while an entry is still wanted (the child is deferred and takes both parents from memory),
rows of earlier launches read on the spot (incremental calls: every parent of the first level), both column widths
(2 columns per workgroup up to 512 members, 4 beyond), roots in later calls (members that join late)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# n, N, seed, mode, p0, p1, ring depth (None = automatic), schedule (events per call; None = one call)
CASES = [
    (300, 30000, 1, 0, 0.0, 0.0, None, None),       # 2 columns per workgroup, automatic depth
    (300, 30000, 1, 0, 0.0, 0.0, 1, None),          # depth 1: nearly every other-parent is pinned, the side table wraps
    (300, 30000, 2, 3, 0.7, 0.0, 2, 7000),          # stale other-parents + incremental calls
    (512, 40000, 3, 2, 0.4, 0.02, 4, None),         # slow members (coin rounds)
    (600, 40000, 4, 0, 0.0, 0.0, None, None),       # 4 columns per workgroup
    (600, 40000, 4, 0, 0.0, 0.0, 1, 9000),
    (1024, 30000, 5, 3, 0.6, 0.0, 2, None),
    (1024, 30000, 6, 2, 0.4, 0.02, 1, 12000),
    (700, 30000, 7, 1, 0.02, 0.0, 2, 4000),         # two cliques
    (64, 20000, 8, 0, 0.0, 0.0, 1, 333),            # SW_CANSEE_IMPL=2 below 256 members, many small calls
    (200, 30000, 9, 3, 0.5, 0.0, 2, None),
    (5, 3000, 10, 0, 0.0, 0.0, 1, 40),
]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,ring_h,sched", CASES)
def test_level_sweep_matches_oracle(pkg, monkeypatch, n, N, seed, mode, p0, p1, ring_h, sched):
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_CANSEE_IMPL", "2")
    if ring_h is not None:
        monkeypatch.setenv("SW_RING_H", str(ring_h))
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o, h = Oracle(n), pkg.Hashgraph(n)
    step = sched or N
    for a in range(0, N, step):
        b = min(N, a + step)
        for d in (o, h):
            d.append_events(*[x[a:b] for x in stream])
            d.divide_rounds(a, b - a)
        fo, fh = list(o.decide_fame()), list(h.decide_fame())
        assert fo == fh
    got = h.can_see()
    if not np.array_equal(got, o.can_see):
        bad = np.argwhere(got != o.can_see)
        e, c_ = bad[0]
        raise AssertionError("can_see differs in %d entries of %d rows; first: event %d column %d got %d expected %d"
                             % (len(bad), len(np.unique(bad[:, 0])), e, c_, got[e, c_], o.can_see[e, c_]))
    assert np.array_equal(h.rounds(), o.round)
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses())
    m = wit >= 0
    assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
    h.close()


def test_level_sweep_is_the_same_twice(pkg, monkeypatch):
    """The side table and the deferral path depend on how the waves of a workgroup interleave: the rows must not."""
    monkeypatch.setenv("SW_CANSEE_IMPL", "2")
    monkeypatch.setenv("SW_RING_H", "1")
    n, N = 400, 50000
    stream = pkg.synth_hashgraph(n, N, 21, 3, 0.6, 0.0)
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    h.divide_rounds(0, N)
    first = h.can_see().copy()
    for _ in range(4):
        h.rewind()
        h.divide_rounds(0, N)
        assert np.array_equal(h.can_see(), first)
    h.close()
