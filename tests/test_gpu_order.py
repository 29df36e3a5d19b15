"""GPU (-m gpu): find_order (swirld.py:280-311) through the C-ABI against the goldens of
the unmodified reference (transactions per call, recorded schedule) and against the oracle
on fresh streams."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

FORKFREE = [n for n in golden_names() if "forks" not in n]


@pytest.mark.parametrize("bulk", ["default", "1"])   # "1": every call through the first-descendant table (k_order_firstdesc)
@pytest.mark.parametrize("name", FORKFREE)
def test_find_order_matches_reference_golden(pkg, name, bulk, monkeypatch):
    if bulk != "default":
        monkeypatch.setenv("SW_ORDER_BULK", bulk)
    g = load_golden(name)
    h = pkg.Hashgraph(g["n"], g["stake"])
    calls = 0
    for a, b in g["batches"]:
        h.append_events(g["creator"][a:b], g["self_parent"][a:b], g["other_parent"][a:b], g["t"][a:b], g["sig"][a:b])
        h.divide_rounds(a, b - a)
        nc = h.decide_fame()
        tx = h.find_order(nc)
        exp = g["transactions"][g["tx_off"][calls]:g["tx_off"][calls + 1]]
        assert list(tx) == list(exp), "find_order of call %d" % calls
        calls += 1
    assert np.array_equal(h.transactions(), g["transactions"])
    h.close()


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,chunk", [
    (4, 4000, 81, 0, 0, 0, 9), (16, 12000, 82, 2, 0.25, 0.03, 500), (64, 40000, 83, 0, 0, 0, None),
    (64, 20000, 84, 3, 0.6, 0, 3000), (130, 20000, 85, 0, 0, 0, None), (256, 30000, 86, 0, 0, 0, 10000),
    (400, 24000, 87, 0, 0, 0, None), (1024, 40000, 88, 1, 0.5, 0.02, None),   # 8 and 16 mask words: the wide forms of every find_order kernel
])
@pytest.mark.parametrize("bulk", ["default", "1", "0"])   # default threshold, always the table, always the searches
def test_find_order_matches_oracle(pkg, n, N, seed, mode, p0, p1, chunk, bulk, monkeypatch):
    from oracle.oracle import Oracle
    if bulk != "default":
        monkeypatch.setenv("SW_ORDER_BULK", bulk)
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    t = t + (np.arange(N) % 7) * 0.25  # non-monotone timestamps with ties in the medians
    o, h = Oracle(n), pkg.Hashgraph(n)
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco, nch = o.decide_fame(), h.decide_fame()
        assert list(nco) == list(nch)
        assert list(h.find_order(nch)) == list(o.find_order(nco))
    assert np.array_equal(h.transactions(), o.transactions)
    h.close()


@pytest.mark.parametrize("env", [
    {"SW_ORDER_SLAB_MB": "1"},                                  # many groups: the two slabs alternate, every group meets the next one's walk
    {"SW_ORDER_SLAB_MB": "1", "SW_ORDER_ONE_STREAM": "1"},      # ... on one stream
    {"SW_ORDER_SLAB_MB": "2", "SW_ORDER_SORT_INLINE": "1"},     # the sorts on the samples' stream
    {"SW_ORDER_SLAB_MB": "2", "SW_ORDER_LATE_COPY": "1"},       # the caller's copy in one piece at the end
    {"SW_ORDER_S": "1"}, {"SW_ORDER_S": "37", "SW_ORDER_SLAB_MB": "3"},   # one stretch per chain / more stretches than a chain has events in a group
])
@pytest.mark.parametrize("n,N,seed,mode,p0,p1", [(256, 120000, 91, 0, 0, 0), (100, 60000, 92, 2, 0.3, 0.03), (600, 90000, 93, 0, 0, 0),
                                                  (256, 200000, 94, 1, 0.5, 0.01)])   # (two cliques: rounds of more than 4096 events — the global-memory sort behind the groups)
def test_find_order_table_path_under_its_knobs(pkg, n, N, seed, mode, p0, p1, env, monkeypatch):
    """the group / slab / stream structure of the bulk path changes nothing: every variant equals the search form's
    order (which the tests above pin to the oracle), and for the first shape the oracle itself"""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    t = t + (np.arange(N) % 5) * 0.5
    def order(envs):
        for k, v in envs.items():
            monkeypatch.setenv(k, v)
        h = pkg.Hashgraph(n)
        h.append_events(cr, sp, op, t, sig)
        h.divide_rounds(0, N)
        tx = np.array(h.find_order(h.decide_fame()))
        h.close()
        for k in envs:
            monkeypatch.delenv(k)
        return tx
    ref = order({"SW_ORDER_BULK": "0"}) if n <= 512 else order({})
    got = order(dict(env, SW_ORDER_BULK="1"))
    assert len(ref) > 20000 and np.array_equal(got, ref)
    if n == 100:
        from oracle.oracle import Oracle
        o = Oracle(n)
        o.append_events(cr, sp, op, t, sig)
        o.divide_rounds(0, N)
        assert np.array_equal(ref, o.find_order(o.decide_fame()))


def test_host_sort_fallback_matches(pkg, monkeypatch):
    """The per-round device sort falls back to a host sort for oversize rounds and for
    (timestamp, 8-byte key) ties; force that path and compare."""
    from oracle.oracle import Oracle
    monkeypatch.setenv("SW_ORDER_HOST", "1")
    n, N = 32, 20000
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 99)
    t = np.floor(t / 9.0)  # heavy timestamp ties
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, N)
    nco, nch = o.decide_fame(), h.decide_fame()
    assert list(h.find_order(nch)) == list(o.find_order(nco))
    monkeypatch.delenv("SW_ORDER_HOST")
    h2 = pkg.Hashgraph(n)
    h2.append_events(cr, sp, op, t, sig)
    h2.divide_rounds(0, N)
    assert list(h2.find_order(h2.decide_fame())) == list(o.transactions)


@pytest.mark.parametrize("big_host", [False, True])
def test_rounds_larger_than_the_lds_sort(pkg, monkeypatch, big_host):
    """A hashgraph with slow members orders more than 4096 events per round at 256 members: those rounds are sorted by
    k_order_sort_big on the device (global scratch keys), not by the host — and give the oracle's order either way
    (SW_ORDER_BIG_HOST=1: the host path such rounds took before)."""
    from oracle.oracle import Oracle
    if big_host:
        monkeypatch.setenv("SW_ORDER_BIG_HOST", "1")
    n, N = 256, 70000
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 87, 2, 0.35, 0.02)
    t = t + (np.arange(N) % 5) * 0.5
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, N)
    nco, nch = o.decide_fame(), h.decide_fame()
    assert list(nco) == list(nch) and len(nch) >= 4
    tx_o = o.find_order(nco)
    tx_h = h.find_order(nch)
    assert len(tx_o) > 4096 * 3 and len(tx_o) / len(nch) > 4096   # (the rounds really are that large)
    assert list(tx_h) == list(tx_o)
    hs = h.counters()["order_rounds_host_sorted"]
    assert (hs > 0) if big_host else (hs == 0)
    h.close()
