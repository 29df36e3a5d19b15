"""CPU, authoring container only (skipped where /root/reference is absent): the drop-in `Node`
(py-swirld_amd/node.py) against the UNMODIFIED reference class, both driven by their own
`test(n_nodes, n_turns)` simulation (swirld.py:331-345) with the same random bytes and the same
clock.  The two runs gossip independently — sync payloads, height-pruned `ask_sync` subsets,
toposorted insertion, the sync event — and must end with the same hashgraph in every node:
compared structurally (an event = creator + position on its creator's chain).  Event hashes
cover the pickled Event class, so both sides hash a class-free serialisation here (see
_class_free_dumps): the reference's results depend on the insertion order of concurrent events,
which follows the hashes.  The device side of the drop-in is the CPU
oracle here (tests/oracle_backend.py); the GPU twin of the call protocol is tests/test_gpu_node.py.
"""
import contextlib
import io
import pickle
import random

import pytest

import oracle_backend
import refharness

pytestmark = pytest.mark.skipif(not refharness.have_reference(), reason="needs /root/reference (authoring container)")


def _rng_bytes(seed):
    rng = random.Random(seed)
    return lambda k: bytes(rng.getrandbits(8) for _ in range(k))


def _clock():
    ticks = iter(range(1, 1 << 30))
    return lambda: 1.0e9 + 0.001 * next(ticks)


def _structure(node, members):
    """Everything a Node knows, keyed by (member index, position on that member's chain)."""
    mi = {pk: i for i, pk in enumerate(members)}
    key = {}
    for h, ev in node.hg.items():
        pos, cur = 0, ev
        while cur.p:
            pos += 1
            cur = node.hg[cur.p[0]]
        key[h] = (mi[ev.c], pos)
    parents = {key[h]: tuple(key[p] for p in ev.p) for h, ev in node.hg.items()}
    rounds = {key[h]: int(node.round[h]) for h in node.hg}
    wit = {int(r): {mi[pk]: key[h] for pk, h in d.items()} for r, d in node.witnesses.items() if d}
    famous = {key[h]: bool(v) for h, v in node.famous.items()}
    see_head = {mi[pk]: key[h] for pk, h in node.can_see[node.head].items()}
    return dict(head=key[node.head], parents=parents, rounds=rounds, witnesses=wit, famous=famous,
                consensus=sorted(int(r) for r in node.consensus), can_see_head=see_head,
                transactions=[key[h] for h in node.transactions], tbd=sorted(key[h] for h in node.tbd))


def _class_free_dumps(obj, *a, **k):
    """pickle.dumps, except that a top-level Event is serialised without its class path: the
    event hash (swirld.py:95) then is the same for swirld.Event and for the drop-in's Event, and
    with it everything that iterates containers keyed by hashes (toposort order = insertion order =
    witness registration order, which "first decider wins" depends on, swirld.py:235, 263)."""
    if hasattr(obj, "_fields"):
        obj = ("Event",) + tuple(obj)
    return pickle.dumps(obj, *a, **k)


def _run_reference(n, turns, seed):
    sw = refharness.import_reference()
    import pysodium  # the stand-in of tests/_pysodium_standin
    pysodium.set_rng(_rng_bytes(seed))
    saved = sw.time, sw.dumps
    sw.time, sw.dumps = _clock(), _class_free_dumps
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            nodes = sw.test(n, turns)
    finally:
        sw.time, sw.dumps = saved
        pysodium.set_rng(None)
    return nodes


def _run_mirror(pkg, monkeypatch, n, turns, seed):
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    monkeypatch.setattr(pkg.node.crypto, "randombytes", _rng_bytes(seed))
    monkeypatch.setattr(pkg.node, "time", _clock())
    monkeypatch.setattr(pkg.node, "dumps", _class_free_dumps)
    with contextlib.redirect_stdout(io.StringIO()):
        return pkg.test(n, turns)


@pytest.mark.parametrize("n,turns,seed", [(4, 250, 11), (5, 300, 12), (7, 900, 13), (3, 120, 14), (4, 600, 15), (6, 800, 16)])
def test_simulation_matches_reference(pkg, monkeypatch, n, turns, seed):
    ref_nodes = _run_reference(n, turns, seed)
    our_nodes = _run_mirror(pkg, monkeypatch, n, turns, seed)
    members = [nd.pk for nd in ref_nodes]
    assert members == [nd.pk for nd in our_nodes], "same random bytes must give the same key pairs"
    progressed = 0
    for ref, ours in zip(ref_nodes, our_nodes):
        a, b = _structure(ref, members), _structure(ours, members)
        for field in ("head", "parents", "rounds", "witnesses", "famous", "consensus", "can_see_head", "tbd"):
            assert a[field] == b[field], field
        assert a["transactions"] == b["transactions"]  # the total order, tie-breaks included
        progressed += len(a["transactions"])
    print("ordered events over all nodes:", progressed)
    # (how far the total order gets depends on the process's hash seed — the gossip partner is
    # picked from a set of bytes keys, swirld.py:322 — so no progress is demanded per case)
    assert all(len(nd.hg) > turns // 4 for nd in ref_nodes)


def test_is_valid_event_agrees_with_reference(pkg, monkeypatch):
    """N3, host side: the same accept / reject decisions as swirld.py:97-108 for
    well-formed events and for every way of malforming one (signature, hash, parents)."""
    sw = refharness.import_reference()
    import pysodium
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    monkeypatch.setattr(pkg.node, "dumps", _class_free_dumps)
    monkeypatch.setattr(sw, "dumps", _class_free_dumps)
    kps = [pysodium.crypto_sign_seed_keypair(bytes([i]) * 32) for i in range(3)]
    stake = {kp[0]: 1 for kp in kps}
    monkeypatch.setattr(sw, "time", _clock())        # the same clock on both sides: same root events
    monkeypatch.setattr(pkg.node, "time", _clock())
    ref = [sw.Node(kp, {}, 3, stake) for kp in kps]
    ours = [pkg.Node(kp, {}, 3, stake) for kp in kps]
    assert [nd.head for nd in ref] == [nd.head for nd in ours]
    for a, b in zip(ref, ours):  # give node 0 the roots of the others
        if a is not ref[0]:
            ref[0].add_event(a.head, a.hg[a.head])
            ours[0].add_event(b.head, b.hg[b.head])
    r0, o0 = ref[0], ours[0]
    good_h, good = r0.new_event(b"x", (r0.head, ref[1].head))
    cases = [("well-formed", good_h, good)]
    cases.append(("wrong hash", b"\0" * 32, good))
    cases.append(("tampered payload", good_h, good._replace(d=b"z")))
    cases.append(("tampered timestamp", good_h, good._replace(t=good.t + 1)))
    cases.append(("bad signature", good_h, good._replace(s=bytes(64))))
    swapped = sw.Event(b"x", (ref[1].head, r0.head), good.t, r0.pk, b"")
    swapped = swapped._replace(s=pysodium.crypto_sign_detached(_class_free_dumps(tuple(swapped[:-1])), r0.sk))
    cases.append(("parents swapped (first is not the self-parent)", sw.crypto_generichash(_class_free_dumps(swapped)), swapped))
    both_mine = sw.Event(b"x", (r0.head, r0.head), good.t, r0.pk, b"")
    both_mine = both_mine._replace(s=pysodium.crypto_sign_detached(_class_free_dumps(tuple(both_mine[:-1])), r0.sk))
    cases.append(("other-parent is mine", sw.crypto_generichash(_class_free_dumps(both_mine)), both_mine))
    unknown = sw.Event(b"x", (r0.head, b"\x07" * 32), good.t, r0.pk, b"")
    unknown = unknown._replace(s=pysodium.crypto_sign_detached(_class_free_dumps(tuple(unknown[:-1])), r0.sk))
    cases.append(("unknown parent", sw.crypto_generichash(_class_free_dumps(unknown)), unknown))
    one_parent = sw.Event(b"x", (r0.head,), good.t, r0.pk, b"")
    one_parent = one_parent._replace(s=pysodium.crypto_sign_detached(_class_free_dumps(tuple(one_parent[:-1])), r0.sk))
    cases.append(("one parent", sw.crypto_generichash(_class_free_dumps(one_parent)), one_parent))
    seen = set()
    for name, h, ev in cases:
        mine = pkg.Event(*ev)
        want = r0.is_valid_event(h, ev)
        assert o0.is_valid_event(h, mine) == want, name
        seen.add(want)
    assert seen == {True, False}


@pytest.mark.parametrize("n,steps,seed,n_forks", [(4, 260, 31, 6), (5, 400, 32, 10), (3, 150, 33, 4)])
def test_forked_hashgraph_matches_reference_node(pkg, monkeypatch, n, steps, seed, n_forks):
    """A Byzantine member forks (two events on one self-parent, later events on either sibling).  The
    unmodified reference class stores both (README.md:84); so does the drop-in with accept_forks=True —
    and one observer of each kind, fed the same events in the same order and calling
    divide_rounds / decide_fame / find_order on the same schedule, must agree on every view: rounds,
    witnesses in dict order (a sibling replaces the value, not the position), fame, consensus,
    can_see of the head, the total order."""
    sw = refharness.import_reference()
    import pysodium
    monkeypatch.setattr(pkg.node, "Hashgraph", oracle_backend.OracleHashgraph)
    monkeypatch.setattr(pkg.node, "dumps", _class_free_dumps)
    monkeypatch.setattr(sw, "dumps", _class_free_dumps)
    monkeypatch.setattr(sw, "time", _clock())
    monkeypatch.setattr(pkg.node, "time", _clock())
    kps = [pysodium.crypto_sign_seed_keypair(bytes([40 + i]) * 32) for i in range(n)]
    stake = {kp[0]: 1 for kp in kps}
    ref = [sw.Node(kp, {}, n, stake) for kp in kps]
    ours = [pkg.Node(kp, {}, n, stake, accept_forks=True) for kp in kps]
    assert [nd.head for nd in ref] == [nd.head for nd in ours]
    obs_r, obs_o = ref[0], ours[0]                       # member 0 observes; everybody creates events
    rng = random.Random(seed)
    latest = {i: [ref[i].head] for i in range(n)}        # per member: tips it may build on (two after a fork)
    pending = []
    for i in range(1, n):                                # the observer learns the other roots
        obs_r.add_event(ref[i].head, ref[i].hg[ref[i].head])
        obs_o.add_event(ours[i].head, ours[i].hg[ours[i].head])
        pending.append(ref[i].head)
    for a, b in zip(ref, ours):                          # every creator knows every root (it signs on top of them)
        for i in range(n):
            if ref[i].head not in a.hg:
                a.add_event(ref[i].head, ref[i].hg[ref[i].head])
                b.add_event(ours[i].head, ours[i].hg[ours[i].head])
    fork_steps = set(rng.sample(range(10, steps - 10), n_forks))
    byz = n - 1
    known = list(pending) + [obs_r.head]

    def create(x, sp, op, payload):
        hr, er = ref[x].new_event(payload, (sp, op))
        ho, eo = ours[x].new_event(payload, (sp, op))
        assert hr == ho, "same keys, clock and serialisation: same event ids"
        for nd, ev in [(m, er) for m in ref] + [(m, eo) for m in ours]:
            if hr not in nd.hg:
                assert nd.is_valid_event(hr, ev)
                nd.add_event(hr, ev)
        pending.append(hr)
        return hr

    for step in range(steps):
        x = byz if step in fork_steps else rng.randrange(n)
        y = rng.choice([m for m in range(n) if m != x])
        sp = rng.choice(latest[x])
        op = rng.choice(latest[y])
        h1 = create(x, sp, op, b"p%d" % step)
        if step in fork_steps:                            # the sibling: same self-parent, another other-parent
            y2 = rng.choice([m for m in range(n) if m != x])
            h2 = create(x, sp, rng.choice(latest[y2]), b"q%d" % step)
            latest[x] = [h1, h2]
        else:
            latest[x] = [h1] if rng.random() < 0.7 else (latest[x] + [h1])[-2:]
        if step % 7 == 6 or step == steps - 1:            # the observers' main()-style call
            if latest[0][-1] in obs_r.hg:
                obs_r.head = obs_o.head = latest[0][-1]
            obs_r.divide_rounds(list(pending))
            obs_o.divide_rounds(list(pending))
            pending.clear()
            with contextlib.redirect_stdout(io.StringIO()):
                ncr, nco = obs_r.decide_fame(), obs_o.decide_fame()
                assert ncr == nco, "new_c at step %d" % step
                obs_r.find_order(ncr)
                obs_o.find_order(nco)
            assert obs_r.transactions == obs_o.transactions, "total order at step %d" % step
    assert obs_o._dev.exact
    assert {h: obs_r.round[h] for h in obs_r.hg} == {h: obs_o.round[h] for h in obs_o.hg}
    assert max(obs_r.round.values()) >= 3
    rw = {r: list(d.items()) for r, d in obs_r.witnesses.items() if d}
    ow = {r: list(obs_o.witnesses[r].items()) for r in obs_o.witnesses if obs_o.witnesses[r]}
    assert rw == ow, "witnesses, dict order included"
    assert dict(obs_r.famous.items()) == dict(obs_o.famous.items()), "fame per event, replaced witnesses included"
    assert obs_r.consensus == obs_o.consensus and obs_r.tbd == obs_o.tbd
    assert dict(obs_r.can_see[obs_r.head]) == dict(obs_o.can_see[obs_o.head])
    assert obs_r.transactions == obs_o.transactions and (len(obs_r.transactions) > 0 or n == 3)
