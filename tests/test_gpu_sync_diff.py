"""GPU (-m gpu): the gossip side from device state.  sw_get_known_heights
against the reference's {member: height[can_see[head][member]]} (swirld.py:125-126), and
sw_sync_diff against a restatement of ask_sync's height-pruned BFS (swirld.py:154-161) on index
arrays: same SET of events for every (answering head, asking head) pair tried, the asker's heights
taken from its own can_see row as Node.sync does.  Then the drop-in Node with the device diff
switched on: whole gossip simulations, every node's state replayed through the oracle."""
import contextlib
import io
from collections import deque

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bfs_subset(head, sp, op, cr, ht, known):
    """swirld.py:154-161 with utils.bfs (utils.py:24-34) on dense indices."""
    seen, q = {head}, deque([head])
    while q:
        u = q.popleft()
        for p in (sp[u], op[u]):
            if p < 0:
                continue
            if (known[cr[p]] < 0 or ht[p] > known[cr[p]]) and p not in seen:
                seen.add(p)
                q.append(p)
    return seen


@pytest.mark.parametrize("n,N,seed,mode,p0,p1", [(8, 3000, 601, 0, 0, 0), (40, 12000, 602, 2, 0.3, 0.02), (130, 20000, 603, 3, 0.6, 0),
                                                 (256, 30000, 604, 1, 0.02, 0)])
def test_sync_diff_equals_reference_bfs(pkg, n, N, seed, mode, p0, p1):
    from oracle.oracle import Oracle
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, N)
    ht, cs = o.height, o.can_see
    rng = np.random.default_rng(seed)
    chains = [np.nonzero(cr == m)[0] for m in range(n)]
    for _ in range(25):
        head = int(rng.integers(n, N))
        asker = int(rng.integers(0, N))                      # the asker's head: its view = that event's can_see row
        exp_kh = np.where(cs[asker] >= 0, ht[np.maximum(cs[asker], 0)], -1)
        known = h.known_heights(asker)
        assert np.array_equal(known, exp_kh)
        first, end, tot = h.sync_diff(head, known)
        got = set()
        for m in range(n):
            ev = h.chain_events(m, int(first[m]), int(end[m]))
            assert np.array_equal(ev, chains[m][first[m]:end[m]])
            got.update(int(e) for e in ev)
        assert len(got) == tot
        assert got == bfs_subset(head, sp, op, cr, ht, known)
    # an asker that knows nobody gets every ancestor of the head
    first, end, tot = h.sync_diff(N - 1, np.full(n, -1, np.int32))
    assert tot == sum(int(np.searchsorted(chains[m], cs[N - 1][m], side="right")) for m in range(n) if cs[N - 1][m] >= 0)
    h.close()


@pytest.mark.parametrize("device_diff", [False, True])
def test_node_gossip_with_device_sync_diff(pkg, device_diff):
    from oracle.oracle import Oracle
    import random
    node_mod = pkg.node
    rng = random.Random(20260922)
    orig_rb, orig_time, orig_flag = node_mod.crypto.randombytes, node_mod.time, node_mod.Node.device_sync_diff
    node_mod.crypto.randombytes = lambda k: bytes(rng.getrandbits(8) for _ in range(k))
    clock = iter(range(1, 1 << 30))
    node_mod.time = lambda: 1.0e9 + 0.001 * next(clock)
    node_mod.Node.device_sync_diff = device_diff
    sched = {}
    orig_div = node_mod.Node.divide_rounds

    def recording(self, events):
        events = tuple(events)
        sched.setdefault(id(self), []).append(len(events))
        return orig_div(self, events)

    node_mod.Node.divide_rounds = recording
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            nodes = pkg.test(5, 300)
    finally:
        node_mod.Node.divide_rounds = orig_div
        node_mod.crypto.randombytes, node_mod.time, node_mod.Node.device_sync_diff = orig_rb, orig_time, orig_flag
    for nd in nodes:
        ids, N, index = nd._ids, len(nd._ids), nd._index
        cr = np.array([nd._mindex[nd.hg[h].c] for h in ids], np.int32)
        sp = np.array([index[nd.hg[h].p[0]] if nd.hg[h].p else -1 for h in ids], np.int32)
        op = np.array([index[nd.hg[h].p[1]] if nd.hg[h].p else -1 for h in ids], np.int32)
        t = np.array([nd.hg[h].t for h in ids], np.float64)
        sig = np.frombuffer(b"".join(nd.hg[h].s for h in ids), np.uint8).reshape(N, 64)
        o = Oracle(5)
        a = 0
        for i, k in enumerate(sched[id(nd)]):
            o.append_events(cr[a:a + k], sp[a:a + k], op[a:a + k], t[a:a + k], sig[a:a + k])
            o.divide_rounds(a, k)
            if i > 0:
                o.find_order(o.decide_fame())
            a += k
        assert [nd.round[h] for h in ids] == list(o.round)
        assert [nd.height[h] for h in ids] == list(o.height)
        assert [index[h] for h in nd.transactions] == list(o.transactions)
    assert max(len(nd._ids) for nd in nodes) > 150
