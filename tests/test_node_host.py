"""CPU: the HOST logic of the drop-in Node (py-swirld_amd/node.py) — hash <-> index maps,
lazy dict views, gossip (sync / ask_sync), the main() call protocol — with the device
backend swapped for the CPU oracle by monkeypatching `node.Hashgraph` (tests/
oracle_backend.py).  The GPU path itself is covered by tests/test_gpu_node.py."""
import contextlib
import io

import numpy as np

import oracle_backend


def test_node_main_loop_on_oracle_backend(pkg, monkeypatch):
    import random
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    rng = random.Random(20260921)  # seeded gossip: a 4-member hashgraph can also stall for a while
    monkeypatch.setattr(pkg.node.crypto, "randombytes", lambda k: bytes(rng.getrandbits(8) for _ in range(k)))
    clock = iter(range(1, 1 << 30))  # and a deterministic clock: nothing in this test depends on the wall time
    monkeypatch.setattr(pkg.node, "time", lambda: 1.0e9 + 0.001 * next(clock))
    with contextlib.redirect_stdout(io.StringIO()):
        nodes = pkg.test(4, 300)
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
    # (how far the total order gets depends on the process's hash seed as well — the gossip partner is picked
    # from a set of bytes keys, swirld.py:322 — so only "some progress" is demanded: 39 … 150 over 20 hash seeds)
    assert max(len(nd.transactions) for nd in nodes) > 5


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
    stores both (no fork detection); this Node accepts the first and drops the second and
    everything built on it, so that the device path (one self-parent chain per member) keeps
    running — and malformed signatures are rejected, not raised."""
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    crypto = pkg.node.crypto
    kpa, kpb = crypto.sign_keypair(), crypto.sign_keypair()
    stake = {kpa[0]: 1, kpb[0]: 1}
    a = pkg.Node(kpa, {}, 2, stake)
    b = pkg.Node(kpb, {}, 2, stake)
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
