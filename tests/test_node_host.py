"""CPU: the HOST logic of the drop-in Node (py-swirld_amd/node.py) — hash <-> index maps,
lazy dict views, gossip (sync / ask_sync), the main() call protocol — with the device
backend swapped for the CPU oracle by monkeypatching `node.Hashgraph` (tests/
oracle_backend.py).  The GPU path itself is covered by tests/test_gpu_node.py."""
import contextlib
import io

import numpy as np

import oracle_backend


class _SortedKeys:
    def __init__(self, keys):
        self._k = keys

    def __sub__(self, other):
        return sorted(set(self._k) - other)


class _SortedNetwork(dict):
    """`network.keys() - {pk}` in a fixed order: Node.main picks its gossip partner from that set
    (swirld.py:322), whose iteration order otherwise depends on the process's hash seed."""

    def keys(self):
        return _SortedKeys(dict.keys(self))


def _run_simulation(pkg, n_nodes, n_turns, rng, **node_kw):
    """pkg.test() (swirld.py:331-345) with a hash-seed independent partner choice."""
    crypto = pkg.node.crypto
    kps = sorted((crypto.sign_keypair() for _ in range(n_nodes)), key=lambda kp: kp[0])
    network = _SortedNetwork()
    stake = {kp[0]: 1 for kp in kps}
    nodes = [pkg.Node(kp, network, n_nodes, stake, **node_kw) for kp in kps]
    for nd in nodes:
        network[nd.pk] = nd.ask_sync
    mains = [nd.main() for nd in nodes]
    for m in mains:
        next(m)
    for _ in range(n_turns):
        next(mains[rng.randrange(n_nodes)])
    return nodes


def test_node_main_loop_on_oracle_backend(pkg, monkeypatch):
    import random
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    rng = random.Random(20260921)  # seeded gossip: a 4-member hashgraph can also stall for a while
    monkeypatch.setattr(pkg.node.crypto, "randombytes", lambda k: bytes(rng.getrandbits(8) for _ in range(k)))
    clock = iter(range(1, 1 << 30))  # and a deterministic clock: nothing in this test depends on the wall time
    monkeypatch.setattr(pkg.node, "time", lambda: 1.0e9 + 0.001 * next(clock))
    with contextlib.redirect_stdout(io.StringIO()):
        nodes = _run_simulation(pkg, 4, 300, rng)
    assert len(nodes) == 4
    for nd in nodes:
        N = len(nd._ids)
        assert N == len(nd.hg) == len(nd.round) and N > 200
        rounds = [nd.round[h] for h in nd._ids]
        assert rounds[0] == 0 and max(rounds) == max(nd.witnesses)
        # witnesses: first event of its creator in that round, registration order ascending
        for r in nd.witnesses:
            idx = [nd._index[h] for h in nd.witnesses[r].values()]
            assert idx == sorted(idx)
            for pk, h in nd.witnesses[r].items():
                assert nd.hg[h].c == pk and nd.round[h] == r
        # can_see[head]: own entry is the event itself, entries are by the right creators
        row = nd.can_see[nd.head]
        assert row[nd.pk] == nd.head
        assert all(nd.hg[h].c == pk for pk, h in row.items())
        # famous only for decided witnesses; consensus rounds are fully decided
        for r in nd.consensus:
            assert all(h in nd.famous for h in nd.witnesses[r].values())
        # total order bookkeeping
        assert len(set(nd.transactions)) == len(nd.transactions)
        assert all(nd.idx[h] == i for i, h in enumerate(nd.transactions))
        assert nd.tbd == set(nd.hg) - set(nd.transactions)
        # unknown ids raise KeyError like the reference's dicts
        for view in (nd.round, nd.can_see):
            try:
                view[b"\\0" * 32]
                raise AssertionError("KeyError expected")
            except KeyError:
                pass
    # the partner choice no longer depends on the process's hash seed (_SortedNetwork): real progress is demanded
    assert max(len(nd.transactions) for nd in nodes) > 50


def test_divide_rounds_rejects_out_of_order(pkg, monkeypatch):
    import pytest
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    kp = pkg.node.crypto.sign_keypair()
    kp2 = pkg.node.crypto.sign_keypair()
    stake = {kp[0]: 1, kp2[0]: 1}
    a = pkg.Node(kp, {}, 2, stake)
    b = pkg.Node(kp2, {}, 2, stake)
    hb = b.head
    a.add_event(hb, b.hg[hb])
    h2, ev2 = a.new_event(None, (a.head, hb))
    a.add_event(h2, ev2)
    with pytest.raises(ValueError):
        a.divide_rounds((h2, hb))          # not the order they were added in
    with pytest.raises(KeyError):
        a.divide_rounds((b"x" * 32,))
    a.divide_rounds((hb, h2))
    assert a.round[h2] == 0 and a.witnesses[0][kp2[0]] == hb


def test_forked_events_are_dropped_not_stored(pkg, monkeypatch):
    """A Byzantine member signs two events on the same self-parent (a fork).  The reference
    stores both (no fork detection) and so does this Node by default; with accept_forks=False it
    accepts the first and drops the second and everything built on it, so that the device path
    (one self-parent chain per member) keeps running — and malformed signatures are rejected,
    not raised."""
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    crypto = pkg.node.crypto
    kpa, kpb = crypto.sign_keypair(), crypto.sign_keypair()
    stake = {kpa[0]: 1, kpb[0]: 1}
    a = pkg.Node(kpa, {}, 2, stake, accept_forks=False)
    b = pkg.Node(kpb, {}, 2, stake, accept_forks=False)
    ra, rb = a.head, b.head
    b.add_event(ra, a.hg[ra])
    a.add_event(rb, b.hg[rb])
    h1, e1 = b.new_event(b"one", (rb, ra))
    h2, e2 = b.new_event(b"two", (rb, ra))      # same self-parent: a fork of member b
    assert a.is_valid_event(h1, e1)
    a.add_event(h1, e1)
    assert not a.is_valid_event(h2, e2)         # the second child of rb is dropped
    b.add_event(h2, e2)                         # (b itself builds on the fork)
    h3, e3 = b.new_event(b"three", (h2, ra))
    assert not a.is_valid_event(h3, e3)         # parent unknown to a: dropped as well
    a.divide_rounds((rb, h1))
    assert a.round[h1] == 0
    # malformed signatures: rejected by is_valid_event instead of escaping as ctypes errors
    assert not a.is_valid_event(h1, e1._replace(s=e1.s[:10]))
    assert not a.is_valid_event(h1, e1._replace(s="not bytes"))
    # a second root of a member is a fork too
    hr, er = b.new_event(None, ())
    assert not a.is_valid_event(hr, er)


def test_forked_events_are_stored_with_accept_forks(pkg, monkeypatch):
    """accept_forks=True: the reference's behaviour (both siblings stored, the later one replaces the
    member's witness but keeps its dict position); host logic on the oracle backend."""
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    crypto = pkg.node.crypto
    kpa, kpb = crypto.sign_keypair(), crypto.sign_keypair()
    stake = {kpa[0]: 1, kpb[0]: 1}
    a = pkg.Node(kpa, {}, 2, stake, accept_forks=True)
    b = pkg.Node(kpb, {}, 2, stake, accept_forks=True)
    ra, rb = a.head, b.head
    a.add_event(rb, b.hg[rb])
    b.add_event(ra, a.hg[ra])
    h1, e1 = b.new_event(b"one", (rb, ra))
    h2, e2 = b.new_event(b"two", (rb, ra))      # a fork of member b
    assert a.is_valid_event(h1, e1) and a.is_valid_event(h2, e2)
    a.add_event(h1, e1)
    a.add_event(h2, e2)
    hr, er = b.new_event(None, ())              # a second root of b
    assert a.is_valid_event(hr, er)
    a.add_event(hr, er)
    a.divide_rounds((rb, h1, h2, hr))
    assert a._dev.exact
    assert a.round[h1] == a.round[h2] == a.round[hr] == 0
    assert list(a.witnesses[0]) == [kpa[0], kpb[0]] and a.witnesses[0][kpb[0]] == hr   # replaced value, kept position


def _seven_nodes(pkg, **kw):
    """Six honest nodes and B, about to equivocate and then fall silent.  Seven members: with one
    silent member a voter still finds more than 2n/3 hops among the OTHER members (the reference's voter
    tally leaves out the voter's own hop, swirld.py:247-254) — at 4 or 5 members nothing would ever be decided."""
    crypto = pkg.node.crypto
    kps = sorted((crypto.sign_keypair() for _ in range(7)), key=lambda kp: kp[0])
    network = _SortedNetwork()
    stake = {kp[0]: 1 for kp in kps}
    nodes = [pkg.Node(kp, network, 7, stake, **kw) for kp in kps]
    for nd in nodes:
        network[nd.pk] = nd.ask_sync
    a, b, c = nodes[:3]
    for x in nodes + nodes:
        for y in (b, nodes[(nodes.index(x) + 1) % 7]):
            if y is not x:
                x.divide_rounds(x.sync(y.pk, b"warm"))
    for y in (a, c):                      # B knows events of A and C to point at ...
        b.divide_rounds(b.sync(y.pk, b"warm"))
    for x in nodes:                       # ... and everybody ends up knowing B's head
        if x is not b:
            x.divide_rounds(x.sync(b.pk, b"warm"))
    return a, b, c, nodes[3:]


def _build_on(node, parent, payload):
    h, ev = node.new_event(payload, (node.head, parent))
    node.add_event(h, ev)
    node.head = h
    return h


def test_one_equivocator_does_not_stop_honest_nodes(pkg, monkeypatch):
    """Seven nodes, member B signs two events on one self-parent.  A has stored BOTH siblings and builds
    on the first, C has only the second and builds on it; then C syncs with A.  Default Node
    (accept_forks=True = the reference, swirld.py:104-112 stores forks): C receives the sibling it
    lacks although its height equals the one C reported (A prunes a forked member's chain at its
    trunk), ends up with both, and the honest nodes keep creating events and deciding rounds with the
    fork in its history."""
    import random

    import pytest
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    with contextlib.redirect_stdout(io.StringIO()):
        a, b, c, rest = _seven_nodes(pkg)
        assert a.accept_forks and c.accept_forks
        hb = b.head
        h1, e1 = b.new_event(b"one", (hb, b._chain_head[a.pk]))   # (other-parents B knows)
        h2, e2 = b.new_event(b"two", (hb, b._chain_head[c.pk]))   # same self-parent: B equivocates
        assert a.is_valid_event(h1, e1) and a.is_valid_event(h2, e2) and c.is_valid_event(h2, e2)
        a.add_event(h1, e1)
        a.add_event(h2, e2)
        assert b.pk in a._trunk_height
        ha = _build_on(a, h1, b"on-one")
        a.divide_rounds((h1, h2, ha))
        c.add_event(h2, e2)
        hc = _build_on(c, h2, b"on-two")
        c.divide_rounds((h2, hc))
        new = c.sync(a.pk, b"after-fork")
        assert h1 in c.hg and ha in c.hg and c.head == new[-1] and b.pk in c._trunk_height
        c.divide_rounds(new)
        assert c._dev.exact                       # a stored fork: the exact path
        c.find_order(c.decide_fame())
        rng = random.Random(5)
        honest = [a, c] + rest
        for _ in range(500):
            x, y = rng.sample(honest, 2)
            new = x.sync(y.pk, b"more")
            x.divide_rounds(new)
            x.find_order(x.decide_fame())
        assert hc in a.hg
        # ask_sync with the device diff switched on falls back to the BFS on the exact path
        a.device_sync_diff = True
        c.divide_rounds(c.sync(a.pk, b"diff"))
    assert min(a.round[a.head], c.round[c.head]) >= 3
    assert a.consensus and c.consensus
    with pytest.raises(pkg.node.VotesUnavailable):
        a.votes[a.head]


def test_siblings_nobody_can_tell_apart_cost_the_step_not_the_node(pkg, monkeypatch):
    """The same, but A has only the first sibling: neither node can know that B forked, A's
    height-pruned diff (swirld.py:154-161) leaves out the sibling of equal height, and what A built on it
    arrives at C with an unknown parent.  The reference dies there (`h` unbound, swirld.py:138-146);
    this Node stores what it can validate, creates no event on the rejected head and stays usable."""
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    with contextlib.redirect_stdout(io.StringIO()):
        a, b, c, rest = _seven_nodes(pkg)
        hb = b.head
        h1, e1 = b.new_event(b"one", (hb, b._chain_head[a.pk]))
        h2, e2 = b.new_event(b"two", (hb, b._chain_head[c.pk]))
        a.add_event(h1, e1)
        ha = _build_on(a, h1, b"on-one")
        a.divide_rounds((h1, ha))
        c.add_event(h2, e2)
        hc = _build_on(c, h2, b"on-two")
        c.divide_rounds((h2, hc))
        head_before, n_before = c.head, len(c.hg)
        new = c.sync(a.pk, b"after-fork")
        assert ha not in c.hg and c.head == head_before
        assert all(h in c.hg for h in new) and len(c.hg) == n_before + len(new)
        c.divide_rounds(new)
        c.find_order(c.decide_fame())
        c.divide_rounds(c.sync(b.pk, b"still-alive"))   # an honest-looking peer: business as usual
    assert c.head != head_before


def test_fork_rejection_mode_survives_a_rejected_remote_head(pkg, monkeypatch):
    """accept_forks=False: C drops the sibling it sees second and everything built on it — including
    A's head.  sync() then creates no event and returns only what it stored (it used to raise
    UnboundLocalError, and main() a KeyError on the rejected ids)."""
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    with contextlib.redirect_stdout(io.StringIO()):
        a, b, c, rest = _seven_nodes(pkg, accept_forks=False)
        hb = b.head
        h1, e1 = b.new_event(b"one", (hb, b._chain_head[a.pk]))   # (other-parents B knows)
        h2, e2 = b.new_event(b"two", (hb, b._chain_head[c.pk]))
        a.add_event(h1, e1)
        ha, ea = a.new_event(b"on-one", (a.head, h1))
        a.add_event(ha, ea)
        a.head = ha
        a.divide_rounds((h1, ha))
        c.add_event(h2, e2)
        c.divide_rounds((h2,))
        head_before, n_before = c.head, len(c.hg)
        new = c.sync(a.pk, b"after-fork")
    assert h1 not in c.hg and ha not in c.hg          # the second sibling and what is built on it: dropped
    assert c.head == head_before                      # no event created on a rejected remote head
    assert all(h in c.hg for h in new) and len(c.hg) == n_before + len(new)
    c.divide_rounds(new)                              # what main() does next: no KeyError
