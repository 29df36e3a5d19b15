"""Test scaffolding: an object with the `Hashgraph` (py-swirld_amd/engine.py) interface that
is backed by the CPU oracle, so that the HOST logic of the drop-in Node (hash <-> index
maps, lazy views, gossip, call protocol) can be exercised on a machine without a GPU by
monkeypatching `node.Hashgraph` inside a test.  Never imported by the product."""
import numpy as np

from oracle.oracle import Oracle


class OracleHashgraph:
    def __init__(self, n_members, stake=None, coin_period=6, device=0):
        self.n = n_members
        self._o = Oracle(n_members, None if stake is None else np.asarray(stake, np.uint64), coin_period)

    def append_events(self, creator, self_parent, other_parent, t=None, sig=None):
        head = self.__dict__.setdefault("_head", {})
        base = self._o.N
        for i, (m, s_) in enumerate(zip(np.asarray(creator).tolist(), np.asarray(self_parent).tolist())):
            if head.get(m, -1) != s_:  # a fork: the real engine moves to its exact path here
                self.exact = True
            head[m] = base + i
        self._o.append_events(creator, self_parent, other_parent, t, sig)

    exact = False

    def set_forks(self, accept=True):
        self._accept_forks = bool(accept)

    def witness_order(self, r):
        return self._o.witness_order(r)

    @property
    def num_events(self):
        return self._o.N

    def divide_rounds(self, first, K):
        self._o.divide_rounds(first, K)

    def decide_fame(self):
        return self._o.decide_fame()

    def find_order(self, rounds):
        return self._o.find_order(rounds)

    @property
    def max_round(self):
        return self._o.max_round

    def rounds(self, first=0, K=None):
        r = self._o.round
        return r[first:] if K is None else r[first:first + K]

    def can_see(self, first=0, K=None):
        c = self._o.can_see
        return c[first:] if K is None else c[first:first + K]

    def witnesses(self, r0=0, r1=None):
        return self._o.witnesses(r0, r1)

    def famous(self, r0=0, r1=None):
        return self._o.famous_table(r0, r1)

    def famous_events(self, first=0, K=None):
        f = self._o.famous_by_event
        return f[first:] if K is None else f[first:first + K]

    def known_heights(self, head_event):
        row = self._o.can_see[head_event]
        ht = self._o.height
        return np.where(row >= 0, ht[np.maximum(row, 0)], -1).astype(np.int32)

    def consensus(self, r0=0, r1=None):
        return self._o.consensus(r0, r1)

    def counters(self):
        return self._o.counters()

    def close(self):
        pass
