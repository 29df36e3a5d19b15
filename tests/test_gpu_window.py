"""GPU (-m gpu): the windowed can_see table (sw_set_window).  A long incremental run —
append, divide_rounds, decide_fame, find_order per call — with row eviction switched on must give
exactly the reference algorithm's rounds, fame, consensus and total order while the resident part
of the table stays a small fraction of what the reference would keep; evicted rows are refused
cleanly; a rewind maps everything again."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,N,chunk,mode,p0,p1", [(64, 400_000, 20_000, 0, 0, 0), (24, 150_000, 3_000, 2, 0.25, 0.2), (256, 300_000, 25_000, 0, 0, 0),
                                                 (320, 150_000, 15_000, 0, 0, 0)])   # (beyond 256 members: the level sweep under the windowed table)
def test_windowed_run_matches_oracle(pkg, n, N, chunk, mode, p0, p1):
    from oracle.oracle import Oracle
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 701, mode, p0, p1)
    o, h = Oracle(n), pkg.Hashgraph(n)
    h.set_window(True, chunk_mb=2)
    peak = 0
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco, nch = list(o.decide_fame()), list(h.decide_fame())
        assert nco == nch
        assert list(h.find_order(nch)) == list(o.find_order(nco))
        first, resident, _ = h.window()
        peak = max(peak, resident)
    assert np.array_equal(h.rounds(), o.round)
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses())
    m = wit >= 0
    assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
    assert np.array_equal(h.transactions(), o.transactions)
    first, resident, evictions = h.window()
    full = N * ((n + 63) // 64 * 64) * 4
    assert evictions > 3 and first > N // 2, "most of the table must have been evicted"
    assert peak < full // 2, "the resident part stays well below the full table (%d of %d bytes)" % (peak, full)
    # resident rows are exact, evicted rows are refused
    assert np.array_equal(h.can_see(first, N - first), o.can_see[first:])
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.can_see(0, 1)
    assert ei.value.code == -34
    with pytest.raises(pkg.SwirldHipError) as ei:      # an event on top of an evicted parent
        h.append_events([int(cr[N - 1])], [N - 1], [0 if cr[0] != cr[N - 1] else 1])
    assert ei.value.code == -34 and h.num_events == N
    # a rewind recomputes everything from event 0: the evicted chunks are mapped again
    h.rewind()
    sweeps0 = h.counters()["chunk_sweeps"]
    h.divide_rounds(0, N)
    assert n > 256 or h.counters()["chunk_sweeps"] > sweeps0, "one large call under the windowed table sweeps in chunks (halo scratch rows mapped behind the table)"
    h.decide_fame()
    hr = h.rounds()
    if not np.array_equal(hr, o.round):   # say where the two part: the first wrong round and the wrong can_see rows
        bad = int(np.flatnonzero(hr != o.round)[0])
        hc = h.can_see(0, N)
        wrong = (hc != o.can_see).any(axis=1)
        rows = np.flatnonzero(wrong)
        edges = np.flatnonzero(np.diff(np.concatenate(([0], wrong.astype(np.int8), [0]))))
        spans = list(zip(edges[0::2].tolist(), edges[1::2].tolist()))[:12]
        r0 = int(rows[0]) if len(rows) else 0
        same = np.flatnonzero((o.can_see == hc[r0]).all(axis=1))[:4] if len(rows) else []
        const = int(sum(len(np.unique(hc[r])) == 1 for r in rows[:: max(1, len(rows) // 256)]))
        pytest.fail("after the rewind: first wrong round at event %d (%d, expected %d; %d wrong in all); can_see rows wrong: %d in spans %s; "
                    "row %d holds %s, expected %s; oracle rows equal to it: %s; constant rows in a sample of 256: %d"
                    % (bad, hr[bad], o.round[bad], int((hr != o.round).sum()), len(rows), spans, r0, hc[r0][:8], o.can_see[r0][:8], same, const))
    assert np.array_equal(h.can_see(0, 2000), o.can_see[:2000])
    h.close()


def test_window_must_be_set_before_the_first_append(pkg):
    h = pkg.Hashgraph(4)
    h.append_events([0, 1, 2, 3], [-1] * 4, [-1] * 4)
    with pytest.raises(pkg.SwirldHipError):
        h.set_window(True)
    h.close()
    h = pkg.Hashgraph(4)
    h.set_window(True)
    h.set_window(False)            # back to the plain table
    h.append_events([0, 1, 2, 3], [-1] * 4, [-1] * 4)
    h.divide_rounds(0, 4)
    assert list(h.rounds()) == [0, 0, 0, 0]
    h.close()


def test_window_off_after_reserve_keeps_the_halo_scratch_rows(pkg, monkeypatch):
    """set_window(1) -> reserve -> set_window(0) re-allocated the table WITHOUT the scratch rows the
    chunk-parallel sweep writes behind the last row; a divide of >= 2 chunks then wrote out of bounds.  One
    helper sizes the table now; the run must use the chunked sweep and equal the oracle."""
    from oracle.oracle import Oracle
    n, N = 64, 80000
    monkeypatch.setenv("SW_CHUNK_MIN", "2048")   # (sub-batches of ~19 k events: four chunks each)
    stream = pkg.synth_hashgraph(n, N, 707)
    h = pkg.Hashgraph(n)
    h.set_window(True)
    h.reserve(N)
    h.set_window(False)
    h.append_events(*stream)
    c0 = h.counters()
    h.divide_rounds(0, N)
    assert h.counters()["chunk_sweeps"] > c0["chunk_sweeps"], "the chunk-parallel sweep did not run"
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    assert np.array_equal(h.rounds(), o.round)
    assert np.array_equal(h.can_see(N - 3000, 3000), o.can_see[N - 3000:])
    h.close()


def gossip_with_a_silent_member(n, N, seed, silent, quiet_from):
    """Uniform random gossip (every event: a random member syncs from another one's latest event) in which member
    `silent` stops for good at event `quiet_from`: nobody creates for it or syncs from it afterwards."""
    rng = np.random.default_rng(seed)
    cr = np.empty(N, np.int32); sp = np.empty(N, np.int32); op = np.empty(N, np.int32)
    head = np.full(n, -1, np.int64)
    for m in range(n):                       # roots
        cr[m], sp[m], op[m] = m, -1, -1
        head[m] = m
    for e in range(n, N):
        while True:
            a, b = (int(x) for x in rng.integers(0, n, 2))
            if a != b and not (e >= quiet_from and silent in (a, b)):
                break
        cr[e], sp[e], op[e] = a, head[a], head[b]
        head[a] = e
    t = np.cumsum(rng.random(N)) + 1.0
    sig = rng.integers(0, 256, (N, 64), dtype=np.uint8)
    return cr, sp, op, t, sig, int(head[silent])


def test_a_silent_member_lapses_and_the_window_moves_on(pkg):
    """Without a lapse one silent member pins the window at its last event (the reference keeps every row anyway);
    with sw_set_window_lapse the resident part stays bounded, every result equals the oracle's, and the lapsed member's
    next event — as an event on top of any evicted row — is refused without a trace."""
    from oracle.oracle import Oracle
    n, N, chunk, silent, quiet_from = 48, 260_000, 10_000, 7, 30_000
    cr, sp, op, t, sig, last = gossip_with_a_silent_member(n, N, 11, silent, quiet_from)
    rowbytes = 64 * 4
    residents = {}
    for lapse in (0, 40_000):
        o, h = Oracle(n), pkg.Hashgraph(n)
        h.set_window(True, chunk_mb=2, lapse_events=lapse)
        peak_late = 0
        for a in range(0, N, chunk):
            b = min(N, a + chunk)
            for d in (o, h):
                d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
                d.divide_rounds(a, b - a)
            nco, nch = list(o.decide_fame()), list(h.decide_fame())
            assert nco == nch
            assert list(h.find_order(nch)) == list(o.find_order(nco))
            if a >= N // 2:
                peak_late = max(peak_late, h.window()[1])
        assert np.array_equal(h.rounds(), o.round)
        wit = h.witnesses()
        assert np.array_equal(wit, o.witnesses())
        m = wit >= 0
        assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
        assert np.array_equal(h.transactions(), o.transactions)
        first, resident, evictions = h.window()
        assert np.array_equal(h.can_see(first, N - first), o.can_see[first:])
        residents[lapse] = (first, peak_late)
        if lapse:
            assert first > last + 100_000, "the window moved past the silent member's last event (%d): first resident event %d" % (last, first)
            assert peak_late < 80_000 * rowbytes, "resident bytes stay bounded: %d" % peak_late
            with pytest.raises(pkg.SwirldHipError) as ei:      # the lapsed member wakes up: refused, nothing stored
                h.append_events([silent], [last], [N - 1])
            assert ei.value.code == -34 and h.num_events == N and "lapsed" in str(ei.value)
            with pytest.raises(pkg.SwirldHipError) as ei:      # somebody syncs from its (evicted) last event
                h.append_events([int(cr[N - 1])], [N - 1], [last])
            assert ei.value.code == -34 and h.num_events == N
            # a rewind maps every row again and forgets who had lapsed
            h.rewind()
            h.divide_rounds(0, N)
            h.decide_fame()
            assert np.array_equal(h.rounds(), o.round)
            e_new = N
            h.append_events([silent], [last], [N - 1])
            assert h.num_events == N + 1
        else:
            assert first <= last, "without a lapse the silent member's last event (%d) pins the window: first resident event %d" % (last, first)
        h.close()
    assert residents[40_000][1] < residents[0][1] // 2


def test_window_lapse_needs_the_window(pkg):
    h = pkg.Hashgraph(8)
    with pytest.raises(pkg.SwirldHipError) as ei:
        h._chk(h._L.sw_set_window_lapse(h._h, 1000))
    assert ei.value.code == -22
    h._chk(h._L.sw_set_window_lapse(h._h, 0))
    h.close()
