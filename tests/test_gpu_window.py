"""GPU (-m gpu): the windowed can_see table (sw_set_window).  A long incremental run —
append, divide_rounds, decide_fame, find_order per call — with row eviction switched on must give
exactly the reference algorithm's rounds, fame, consensus and total order while the resident part
of the table stays a small fraction of what the reference would keep; evicted rows are refused
cleanly; a rewind maps everything again."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,N,chunk,mode,p0,p1", [(64, 400_000, 20_000, 0, 0, 0), (24, 150_000, 3_000, 2, 0.25, 0.2), (256, 300_000, 25_000, 0, 0, 0)])
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
    assert h.counters()["chunk_sweeps"] > sweeps0, "one large call under the windowed table sweeps in chunks (halo scratch rows mapped behind the table)"
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
