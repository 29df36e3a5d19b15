"""GPU (-m gpu): randomized sweep of the HIP path against the oracle — random member counts,
generator modes, call schedules and tuning knobs (candidate width, band cap / limit, pipelining
depth, can_see kernel variant).  Small sizes, many shapes: aimed at the round-loop state machine
(cursor retries, far candidates, waits, band growth, incremental restarts)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(160))
def test_random_shapes(pkg, monkeypatch, seed):
    from oracle.oracle import Oracle
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([2, 3, 5, 8, 13, 21, 40, 64, 65, 100, 130, 257]))
    N = int(rng.integers(max(n, 50), 6000 if n <= 64 else 9000))
    mode = int(rng.integers(0, 4))
    p0, p1 = float(rng.uniform(0.01, 0.7)), float(rng.uniform(0.002, 0.2))
    chunk = None if rng.random() < 0.4 else int(rng.integers(1, max(2, N // 2)))
    monkeypatch.setenv("SW_TALLY_K", str(int(rng.choice([4, 8, 16, 28, 31, 32, 63]))))
    monkeypatch.setenv("SW_BAND", str(int(rng.choice([64, 256, 4096, 100000]))))
    if rng.random() < 0.5:
        monkeypatch.setenv("SW_BAND_MAX", str(int(rng.choice([64, 1024, 1 << 20]))))
    monkeypatch.setenv("SW_CANSEE_IMPL", str(int(rng.choice([2, 3, 3, 6, 6, 6]))))
    monkeypatch.setenv("SW_TALLY_IMPL", str(int(rng.choice([0, 1, 1, 2, 2]))))
    monkeypatch.setenv("SW_SKIP", str(int(rng.choice([0, 1, 2, 2, 3, 7]))))
    if rng.random() < 0.25:
        monkeypatch.setenv("SW_GALLOP", str(int(rng.choice([1, 2, 3]))))
    if rng.random() < 0.5:  # chunk-parallel sweep at toy sizes (only the dataflow variant 6 chunks)
        monkeypatch.setenv("SW_CHUNKS", str(int(rng.choice([2, 3, 4, 8]))))
        monkeypatch.setenv("SW_CHUNK_MIN", str(int(rng.choice([64, 300, 1000]))))
        monkeypatch.setenv("SW_HALO", str(int(rng.choice([0, 5, 100, 2000]))))
        monkeypatch.setenv("SW_CHUNK_CFG", str(int(rng.choice([0, 1, 2]))))
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 9000 + seed, mode, p0, p1)
    t = t + rng.integers(0, 3, N) * 0.5
    # (drawn after everything else, so that the shapes of the earlier rounds' sweeps stay what they were) round numbers and
    # sees-masks from the band pass + check (default) or every event from its row
    monkeypatch.setenv("SW_FIN_BAND", str(int(np.random.default_rng(8000 + seed).choice([0, 1, 1]))))
    # (round 5) the band pass's path for full groups of 8 events (default) or the generic path only
    monkeypatch.setenv("SW_BAND_FAST", str(int(np.random.default_rng(8100 + seed).choice([0, 1, 1]))))
    # (round 5) popcount bounds in front of the mask gathers of the one-wave-per-slot tally
    monkeypatch.setenv("SW_TALLY_FILTER", str(int(np.random.default_rng(8200 + seed).choice([0, 1]))))
    # (round 6) find_order through the first-descendant table whatever the size of the call, under its group / stream knobs
    org = np.random.default_rng(8400 + seed)
    if org.random() < 0.6:
        monkeypatch.setenv("SW_ORDER_BULK", "1")
        if org.random() < 0.5:
            monkeypatch.setenv("SW_ORDER_SLAB_MB", str(int(org.choice([1, 1, 2, 8]))))
        if org.random() < 0.3:
            monkeypatch.setenv("SW_ORDER_S", str(int(org.choice([1, 3, 64]))))
        if org.random() < 0.25:
            monkeypatch.setenv(str(org.choice(["SW_ORDER_ONE_STREAM", "SW_ORDER_SORT_INLINE", "SW_ORDER_LATE_COPY"])), "1")
    # (round 6) the level sweep's ring depth: shallow rings pin other-parents into its side table and defer events
    lrg = np.random.default_rng(8500 + seed)
    if lrg.random() < 0.6:
        monkeypatch.setenv("SW_RING_H", str(int(lrg.choice([1, 1, 2, 4]))))
    stake = None
    if n >= 8 and rng.random() < 0.2:  # near-unit weighted stakes (the only weighted kind that progresses)
        stake = np.ones(n, np.uint64)
        stake[rng.integers(0, n, size=max(1, n // 10))] = 2
    o, h = Oracle(n, stake), pkg.Hashgraph(n, stake)
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco, nch = o.decide_fame(), h.decide_fame()
        assert list(nco) == list(nch)
        try:
            exp = list(o.find_order(nco))
        except Exception:  # IndexError of swirld.py:305 (single seeing witness, weighted stakes)
            with pytest.raises(pkg.SwirldHipError):
                h.find_order(nch)
            return
        assert list(h.find_order(nch)) == exp
    assert np.array_equal(h.rounds(), o.round)
    assert np.array_equal(h.can_see(), o.can_see)
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses())
    m = wit >= 0
    assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
    assert np.array_equal(h.consensus(), o.consensus())
    assert np.array_equal(h.transactions(), o.transactions)
    h.close()
