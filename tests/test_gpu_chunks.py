"""GPU (-m gpu): the chunk-parallel can_see sweep (k_cansee_chunks + k_cansee_fixup) against
the oracle.  Small hashgraphs with tiny chunks and halos, so that every path runs: nothing provisional
(ample halo), entries repaired by gathers (short or no halo), chunks swept a second time (members silent
for longer than the halo), ranges that start in the middle of the hashgraph (rows of earlier calls),
every lane / ring configuration.  The CPU statement of the same algorithm is tests/model_chunks.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# n, N, seed, mode, p0, p1, chunks, halo (None = default 32 npad), chunk_min, cfg, schedule
CASES = [
    (64, 20000, 1, 0, 0.0, 0.0, 4, None, 1024, 0, None),
    (64, 20000, 1, 0, 0.0, 0.0, 4, 0, 1024, 0, None),
    (64, 20000, 1, 0, 0.0, 0.0, 3, 64, 512, 1, None),
    (64, 20000, 1, 0, 0.0, 0.0, 8, 300, 256, 2, None),
    (256, 60000, 2, 0, 0.0, 0.0, 4, None, 4096, 0, None),
    (256, 60000, 2, 0, 0.0, 0.0, 4, 1000, 4096, 1, None),
    (200, 40000, 3, 0, 0.0, 0.0, 2, 0, 2048, 2, None),
    (130, 30000, 4, 1, 0.02, 0.0, 4, 500, 1024, 0, None),       # two cliques
    (100, 30000, 5, 2, 0.4, 50.0, 4, 2000, 1024, 0, None),      # slow members: repairs / second sweeps
    (40, 12000, 6, 3, 0.3, 0.0, 4, 100, 512, 1, None),          # stale other-parents
    (17, 9000, 7, 2, 0.5, 400.0, 3, 50, 256, 0, None),          # nearly silent members
    (5, 3000, 8, 0, 0.0, 0.0, 4, 3, 64, 0, None),
    (64, 30000, 9, 0, 0.0, 0.0, 4, 700, 1024, 0, 7000),         # incremental calls: rows of earlier calls (zone i)
    (256, 50000, 10, 2, 0.2, 30.0, 4, 3000, 2048, 0, 17000),
    (33, 8000, 11, 0, 0.0, 0.0, 4, 0, 128, 1, 1500),
]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,chunks,halo,chunk_min,cfg,sched", CASES)
def test_chunked_sweep_matches_oracle(pkg, monkeypatch, n, N, seed, mode, p0, p1, chunks, halo, chunk_min, cfg, sched):
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_CHUNKS", str(chunks))
    monkeypatch.setenv("SW_CHUNK_MIN", str(chunk_min))
    monkeypatch.setenv("SW_CHUNK_CFG", str(cfg))
    if halo is not None:
        monkeypatch.setenv("SW_HALO", str(halo))
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o, h = Oracle(n), pkg.Hashgraph(n)
    step = sched or N
    for a in range(0, N, step):
        b = min(N, a + step)
        for d in (o, h):
            d.append_events(*[x[a:b] for x in stream])
            d.divide_rounds(a, b - a)
        fo, fh = list(o.decide_fame()), list(h.decide_fame())
    got = h.can_see()
    if not np.array_equal(got, o.can_see):
        bad = np.argwhere(got != o.can_see)
        e, c_ = bad[0]
        raise AssertionError("can_see differs in %d entries of %d rows; first: event %d column %d got %d expected %d; counters %s"
                             % (len(bad), len(np.unique(bad[:, 0])), e, c_, got[e, c_], o.can_see[e, c_], h.counters()))
    assert np.array_equal(h.rounds(), o.round)
    assert fo == fh
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses())
    m = wit >= 0
    assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
    c = h.counters()
    assert c["chunk_sweeps"] >= 2, c
    if halo == 0:
        assert c["chunk_provisional"] > 0 and (c["chunk_repaired"] > 0 or c["chunk_resweeps"] > 0), c
    h.close()


def test_ample_halo_leaves_nothing_to_repair_at_uniform_gossip(pkg, monkeypatch):
    """What the scheme rests on at the benchmark's shape: with the default halo (32 npad events) no entry of a
    chunk is provisional, so neither the repair nor the second sweep runs."""
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_CHUNK_MIN", "8192")
    n, N = 256, 150000
    stream = pkg.synth_hashgraph(n, N, 21)
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(*stream)
        d.divide_rounds(0, N)
    assert np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.rounds(), o.round)
    c = h.counters()
    assert c["chunk_sweeps"] >= 8 and c["chunk_provisional"] == 0 and c["chunk_resweeps"] == 0, c
    h.close()


def test_too_many_provisional_entries_switch_the_context_back_to_the_unchunked_sweep(pkg, monkeypatch):
    """Without a halo every chunk starts with thousands of provisional stores: beyond the limit the chunks are
    swept a second time from final rows (exact), and the context stops chunking for the calls that follow."""
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_CHUNK_MIN", "2048")
    monkeypatch.setenv("SW_HALO", "0")
    n, N = 64, 40000
    stream = pkg.synth_hashgraph(n, N, 31)
    o, h = Oracle(n), pkg.Hashgraph(n)
    half = N // 2
    for a, b in ((0, half), (half, N)):
        for d in (o, h):
            d.append_events(*[x[a:b] for x in stream])
            d.divide_rounds(a, b - a)
        if a == 0:
            first = h.counters()
            assert first["chunk_resweeps"] > 0, first
    c = h.counters()
    assert c["chunk_sweeps"] == first["chunk_sweeps"], (first, c)     # the second call ran unchunked
    assert np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.rounds(), o.round)
    h.close()


def test_silent_members_are_settled_by_their_frontier_event(pkg, monkeypatch):
    """Members silent for longer than the halo: their columns hold the member's newest event (reached as a
    leaf, or through final rows), which is final although it lies outside the window — so such hashgraphs keep
    the chunk-parallel sweep, with a handful of repairs instead of second sweeps."""
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_CHUNK_MIN", "2048")
    monkeypatch.setenv("SW_HALO", "1024")
    n, N = 64, 40000
    stream = pkg.synth_hashgraph(n, N, 31, 2, 0.5, 2000.0)   # half of the members 2000x less active than the others
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(*stream)
        d.divide_rounds(0, N)
    assert np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.rounds(), o.round)
    c = h.counters()
    assert c["chunk_sweeps"] >= 4 and c["chunk_resweeps"] == 0, c
    h.close()
