"""GPU (-m gpu): the ingest side of the C-ABI (sw_append_events = Node.add_event, swirld.py:114-120).
Bulk appends validate on the device, build the per-member chains there and upload the payload
(timestamps, signatures) behind the call; the host mirrors (parents / heights, chain pool,
signatures) are refilled lazily.  Every path must be atomic on rejection and bit-exact against the
oracle afterwards: bulk only, bulk + small appends, small only, reset and re-ingest."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(n, stream, N):
    from oracle.oracle import Oracle
    o = Oracle(n)
    o.append_events(*[a[:N] for a in stream])
    o.divide_rounds(0, N)
    return o, list(o.decide_fame())


def test_bulk_append_is_atomic_on_rejection(pkg):
    n, N = 32, 40_000
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 201)
    o, nco = _oracle(n, (cr, sp, op, t, sig), N)
    h = pkg.Hashgraph(n)
    bad = op.copy()
    k = 31_007
    bad[k] = sp[k]                                   # other-parent by the same member, found by the device pass
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, sp, bad, t, sig)
    assert ei.value.code == -22 and str(k) in str(ei.value)
    assert h.num_events == 0
    h.set_forks(False)                               # (accepted by default: tests/test_gpu_forks.py)
    fork = sp.copy()
    j = int(np.nonzero(cr[n:] == cr[n + 5])[0][3]) + n  # a later event of that member: point it at an older self-parent
    fork[j] = sp[sp[j]]
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, fork, op, t, sig)
    assert ei.value.code == -95 and h.num_events == 0
    late = sp.copy()
    late[500] = 700                                  # parent index not earlier
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, late, op, t, sig)
    assert ei.value.code == -22 and h.num_events == 0
    # the untouched context then ingests the valid stream: bulk prefix, small appends, another bulk
    a, b = 25_000, 25_300
    h.append_events(cr[:a], sp[:a], op[:a], t[:a], sig[:a])
    for s in range(a, b, 7):
        e = min(b, s + 7)
        h.append_events(cr[s:e], sp[s:e], op[s:e], t[s:e], sig[s:e])
    with pytest.raises(pkg.SwirldHipError):          # a rejected small append changes nothing either
        h.append_events(cr[b:b + 3], sp[b:b + 3], bad[k:k + 3] * 0 + sp[b:b + 3])
    assert h.num_events == b
    h.append_events(cr[b:], sp[b:], op[b:], t[b:], sig[b:])
    assert h.num_events == N
    assert np.array_equal(h.heights(), o.height)     # lazily computed after bulk appends
    h.divide_rounds(0, N)
    nc = list(h.decide_fame())
    assert nc == nco
    assert np.array_equal(h.rounds(), o.round) and np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.witnesses(), o.witnesses())
    assert list(h.find_order(nc)) == list(o.find_order(nco))
    h.close()


@pytest.mark.parametrize("host_sort", [False, True])
def test_reset_and_reingest(pkg, monkeypatch, host_sort):
    """sw_reset forgets the events; the same context then ingests another hashgraph (the bench's
    end-to-end loop).  With the host sort forced, find_order needs the lazily fetched signatures."""
    if host_sort:
        monkeypatch.setenv("SW_ORDER_HOST", "1")
    n = 48
    h = pkg.Hashgraph(n)
    for seed, N in ((301, 30_000), (302, 12_000), (303, 30_000)):
        stream = pkg.synth_hashgraph(n, N, seed)
        o, nco = _oracle(n, stream, N)
        h.reset()
        assert h.num_events == 0
        h.append_events(*stream)
        h.divide_rounds(0, N)
        nc = list(h.decide_fame())
        assert nc == nco
        assert np.array_equal(h.rounds(), o.round)
        assert np.array_equal(h.can_see(N - 2000, 2000), o.can_see[N - 2000:])
        assert list(h.find_order(nc)) == list(o.find_order(nco))
        wit = h.witnesses()
        # a vote that needs the coin bits / voter masks after a bulk ingest
        assert h.vote(2, int(np.nonzero(wit[2] >= 0)[0][0]), 1, int(np.nonzero(wit[1] >= 0)[0][0])) in (0, 1, -1)
    h.close()


def test_small_appends_only_keep_mirrors_incremental(pkg):
    """A Node appends a handful of events per call: chains are extended in place (segments grow and
    move), mirrors stay complete without downloads, results equal the oracle's with the same schedule."""
    from oracle.oracle import Oracle
    n, N = 12, 9000
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 401)
    o, h = Oracle(n), pkg.Hashgraph(n)
    rng = np.random.default_rng(9)
    a = 0
    while a < N:
        b = min(N, a + int(rng.integers(1, 12)))
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco, nch = list(o.decide_fame()), list(h.decide_fame())
        assert nco == nch
        assert list(h.find_order(nch)) == list(o.find_order(nco))
        a = b
    assert np.array_equal(h.heights(), o.height)
    assert np.array_equal(h.rounds(), o.round) and np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.transactions(), o.transactions)
    h.close()
