"""GPU (-m gpu): forked hashgraphs through the C-ABI — the exact path of csrc/exact.hip.h (the
reference's divide_rounds / decide_fame / find_order statement by statement on the device-resident
state, one wavefront).  A context accepts the forked event the reference would store (README.md:84),
hands its round-synchronous state over and continues exactly: the golden fixture of the unmodified
reference on a forked DAG, random forked hashgraphs against the oracle (batch and incremental
schedules, the switch in the middle of a run), rewind / reset, the entry points that are not
available on that path, and a drop-in Node that stores forks."""
import numpy as np
import pytest

from conftest import load_golden
from test_exact_host import add_forks

pytestmark = pytest.mark.gpu


def test_forked_golden_of_the_reference(pkg):
    g = load_golden("n8_s11_forks")
    h = pkg.Hashgraph(g["n"], g["stake"])
    calls = 0
    for a, b in g["batches"]:
        h.append_events(g["creator"][a:b], g["self_parent"][a:b], g["other_parent"][a:b], g["t"][a:b], g["sig"][a:b])
        h.divide_rounds(a, b - a)
        nc = h.decide_fame()
        assert list(nc) == list(g["new_c_flat"][g["new_c_off"][calls]:g["new_c_off"][calls + 1]])
        tx = h.find_order(nc)
        assert list(tx) == list(g["transactions"][g["tx_off"][calls]:g["tx_off"][calls + 1]])
        calls += 1
    assert h.exact
    assert np.array_equal(h.heights(), g["height"])
    assert np.array_equal(h.rounds(), g["round"])
    assert np.array_equal(h.can_see(), g["can_see"])
    wit = h.witnesses()
    assert np.array_equal(wit, g["witnesses"])
    for r, order in enumerate(g["wit_order"]):
        assert np.array_equal(h.witness_order(r), order), "dict order of witnesses[%d]" % r
    fam = h.famous()
    m = wit >= 0
    assert np.array_equal(fam[m], g["famous"][wit[m]]) and (fam[~m] == -1).all()
    assert np.array_equal(h.consensus(), g["consensus"])
    assert np.array_equal(h.transactions(), g["transactions"])
    h.close()


def run_both(pkg, n, stream, chunk, stake=None):
    from oracle.oracle import Oracle
    cr, sp, op, t, sig = stream
    N = len(cr)
    o, h = Oracle(n, stake), pkg.Hashgraph(n, stake)
    chunk = chunk or N
    switched_at = None
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in (o, h):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        if switched_at is None and h.exact:
            switched_at = a
        nco, nch = list(o.decide_fame()), list(h.decide_fame())
        assert nco == nch, "new_c of the call ending at %d" % b
        assert list(o.find_order(nco)) == list(h.find_order(nch)), "find_order of the call ending at %d" % b
    return o, h, switched_at


def assert_same(o, h):
    assert np.array_equal(h.rounds(), o.round)
    assert np.array_equal(h.can_see(), o.can_see)
    assert np.array_equal(h.witnesses(), o.witnesses())
    for r in range(o.max_round + 1):
        assert np.array_equal(h.witness_order(r), o.witness_order(r)), "dict order of witnesses[%d]" % r
    assert np.array_equal(h.famous(), o.famous_table())
    assert np.array_equal(h.consensus(), o.consensus())
    assert np.array_equal(h.transactions(), o.transactions)
    co, ch = o.counters(), h.counters()
    for k in ("voter_evals", "majority_evals", "coin_votes", "coin_flips"):
        assert ch[k] == co[k], k


@pytest.mark.parametrize("n,N,seed,forks,chunk,mode,p0,p1", [
    (8, 600, 1, 10, None, 0, 0, 0), (8, 600, 2, 10, 37, 0, 0, 0), (5, 400, 3, 25, 1, 0, 0, 0),
    (16, 1500, 4, 30, 100, 2, 0.3, 0.1), (70, 3000, 5, 12, 500, 0, 0, 0), (4, 900, 6, 40, 9, 0, 0, 0),
    (12, 1000, 7, 20, None, 1, 0.02, 0), (130, 2500, 8, 6, 400, 0, 0, 0),
])
def test_forked_hashgraphs_match_oracle(pkg, n, N, seed, forks, chunk, mode, p0, p1):
    stream = add_forks(pkg.synth_hashgraph(n, N, seed, mode, p0, p1), n, seed, forks, start=N // 3 if chunk else 0)
    o, h, switched_at = run_both(pkg, n, stream, chunk)
    assert h.exact
    if chunk:
        assert switched_at is not None and switched_at + chunk > N // 3, "the calls before the first fork ran on the round-synchronous path"
    assert_same(o, h)
    h.close()


def test_stake_rewind_reset_and_unavailable_entry_points(pkg):
    n = 9
    stake = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9], np.uint64)
    base = pkg.synth_hashgraph(n, 800, 21)
    stream = add_forks(base, n, 21, 15)
    o, h, _ = run_both(pkg, n, stream, 60, stake)
    assert_same(o, h)
    for call in (lambda: h.vote(1, 0, 0, 0), lambda: h.sees_masks(0, 1), lambda: h.decide_fame_partial(0, 2),
                 lambda: h.sync_diff(5, np.zeros(n, np.int32))):
        with pytest.raises(pkg.SwirldHipError) as ei:
            call()
        assert ei.value.code == -95
    assert np.array_equal(h.known_heights(len(stream[0]) - 1),
                          np.where(o.can_see[-1] >= 0, o.height[np.maximum(o.can_see[-1], 0)], -1))
    # rewind: the events (forks included) stay, one batch pass gives the batch results
    from oracle.oracle import Oracle
    cr, sp, op, t, sig = stream
    ob = Oracle(n, stake)
    ob.append_events(cr, sp, op, t, sig)
    ob.divide_rounds(0, len(cr))
    ncb = list(ob.decide_fame())
    txb = list(ob.find_order(ncb))
    h.rewind()
    assert h.exact
    h.divide_rounds(0, len(cr))
    assert list(h.decide_fame()) == ncb and list(h.find_order(ncb)) == txb
    assert np.array_equal(h.rounds(), ob.round) and np.array_equal(h.famous(), ob.famous_table())
    # reset: a fresh, fork-free hashgraph runs on the round-synchronous path again
    h.reset()
    assert not h.exact
    cr, sp, op, t, sig = base
    of = Oracle(n, stake)
    for d in (of, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, len(cr))
    assert list(of.decide_fame()) == list(h.decide_fame()) and not h.exact
    assert np.array_equal(h.rounds(), of.round)
    # ... and a fork arriving now moves it to the exact path once more, rows of stale rounds cleared
    h.reset()
    o2, _, _ = None, None, None
    cr, sp, op, t, sig = stream
    o2 = Oracle(n, stake)
    for d in (o2, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, len(cr))
    assert list(o2.decide_fame()) == list(h.decide_fame()) and h.exact
    assert np.array_equal(h.witnesses(), o2.witnesses())
    h.close()


def test_forks_refused_on_request_and_in_windowed_mode(pkg):
    n = 6
    stream = add_forks(pkg.synth_hashgraph(n, 300, 9), n, 9, 5)
    cr, sp, op, t, sig = stream
    h = pkg.Hashgraph(n)
    h.set_forks(False)
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, sp, op, t, sig)
    assert ei.value.code == -95 and h.num_events == 0 and not h.exact
    h.close()
    h = pkg.Hashgraph(n)
    h.set_window(True, chunk_mb=2)
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, sp, op, t, sig)
    assert ei.value.code == -95 and h.num_events == 0
    h.close()


def test_node_that_stores_forks(pkg):
    """The drop-in Node with accept_forks=True: both siblings stored, results those of the reference
    algorithm (oracle on the node's own dense event order)."""
    from oracle.oracle import Oracle
    crypto = pkg.node.crypto
    kps = [crypto.sign_keypair() for _ in range(3)]
    stake = {kp[0]: 1 for kp in kps}
    a, b, c = (pkg.Node(kp, {}, 3, stake, accept_forks=True) for kp in kps)
    new = []
    for other in (b, c):
        a.add_event(other.head, other.hg[other.head]); new.append(other.head)
    b.add_event(a.head, a.hg[a.head]); b.add_event(c.head, c.hg[c.head])
    h1, e1 = b.new_event(b"one", (b.head, a.head))
    h2, e2 = b.new_event(b"two", (b.head, c.head))     # a fork of member b
    for hh, ee in ((h1, e1), (h2, e2)):
        assert a.is_valid_event(hh, ee)
        a.add_event(hh, ee); new.append(hh)
    head = a.head
    for k in range(40):                                  # a keeps building on alternating siblings
        hk, ek = a.new_event(b"x%d" % k, (head, (h1, h2)[k % 2]))
        a.add_event(hk, ek); new.append(hk); head = hk
    a.divide_rounds(new)
    a.head = head
    nc = a.decide_fame()
    a.find_order(nc)
    assert a._dev.exact
    o = Oracle(3)
    o.append_events(a._dev.creators() if hasattr(a._dev, "creators") else np.array([a._mindex[a.hg[h].c] for h in a._ids], np.int32),
                    np.array([a._index[a.hg[h].p[0]] if a.hg[h].p else -1 for h in a._ids], np.int32),
                    np.array([a._index[a.hg[h].p[1]] if a.hg[h].p else -1 for h in a._ids], np.int32),
                    np.array([a.hg[h].t for h in a._ids], np.float64),
                    np.frombuffer(b"".join(a.hg[h].s for h in a._ids), np.uint8).reshape(-1, 64))
    o.divide_rounds(0, len(a._ids))
    assert [a.round[h] for h in a._ids] == o.round.tolist()
    for r in range(o.max_round + 1):
        assert [a._mindex[pk] for pk in a.witnesses[r]] == o.witness_order(r).tolist()
        assert [a._index[w] for w in a.witnesses[r].values()] == [int(o.witnesses()[r][m]) for m in o.witness_order(r)]
