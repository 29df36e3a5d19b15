"""Drives the UNMODIFIED reference (/root/reference/swirld.py) on a dense-index
event stream.  Test scaffolding: used by tests/golden/make_golden.py (authoring
container only) and by the differential tests that are skipped when the reference
tree is absent (it does not exist on the GPU box).
"""
import contextlib
import hashlib
import io
import os
import sys

import numpy as np

REF_DIR = os.environ.get("SWIRLD_REFERENCE_PATH", "/root/reference")
_STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pysodium_standin")


def have_reference():
    return os.path.exists(os.path.join(REF_DIR, "swirld.py"))


def import_reference():
    for p in (REF_DIR, _STANDIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    import swirld  # noqa: the reference module itself
    return swirld


def member_pk(c):
    return hashlib.blake2b(b"member-%d" % c, digest_size=32).digest()


def event_id(i):
    return hashlib.blake2b(b"event-%d" % i, digest_size=32).digest()


class RefRun:
    """A blank reference Node (fields of swirld.py:39-72, without the root event of
    75-80) fed with a dense-index stream through Node.add_event."""

    def __init__(self, n, stake=None):
        self.sw = import_reference()
        self.n = n
        self.pks = [member_pk(c) for c in range(n)]
        self.pk_index = {pk: c for c, pk in enumerate(self.pks)}
        stake = [1] * n if stake is None else [int(s) for s in stake]
        node = object.__new__(self.sw.Node)
        node.pk, node.sk = self.pks[0], b"\0" * 64
        node.network = {}
        node.n = n
        node.stake = {pk: s for pk, s in zip(self.pks, stake)}
        node.tot_stake = sum(stake)
        node.min_s = 2 * node.tot_stake / 3
        node.hg = {}
        node.head = None
        node.round = {}
        node.tbd = set()
        node.transactions = []
        node.idx = {}
        node.consensus = set()
        from collections import defaultdict
        node.votes = defaultdict(dict)
        node.witnesses = defaultdict(dict)
        node.famous = {}
        node.height = {}
        node.can_see = {}
        self.node = node
        self.ids = []
        self.id_index = {}

    def append(self, creator, sp, op, t, sig):
        Event = self.sw.Event
        base = len(self.ids)
        for i in range(len(creator)):
            e = base + i
            h = event_id(e)
            p = () if sp[i] < 0 else (self.ids[sp[i]], self.ids[op[i]])
            ev = Event(None, p, float(t[i]), self.pks[int(creator[i])], bytes(sig[i]))
            self.ids.append(h)
            self.id_index[h] = e
            self.node.add_event(h, ev)

    def divide_rounds(self, first, K):
        self.node.divide_rounds(self.ids[first:first + K])

    def decide_fame(self):
        return sorted(self.node.decide_fame())

    def find_order(self, new_c):
        before = len(self.node.transactions)
        with contextlib.redirect_stdout(io.StringIO()):
            self.node.find_order(set(new_c))
        return [self.id_index[h] for h in self.node.transactions[before:]]

    # ---- extraction into dense arrays ----
    def extract(self):
        nd, n, N = self.node, self.n, len(self.ids)
        ix, pk_ix = self.id_index, self.pk_index
        rnd = np.full(N, -1, np.int32)
        for h, r in nd.round.items():
            rnd[ix[h]] = r
        height = np.array([nd.height[h] for h in self.ids], np.int32)
        cs = np.full((N, n), -1, np.int32)
        for h, row in nd.can_see.items():
            e = ix[h]
            for pk, k in row.items():
                cs[e, pk_ix[pk]] = ix[k]
        R = (max(nd.witnesses) + 1) if nd.witnesses else 0
        wit = np.full((R, n), -1, np.int32)
        wit_order = []  # per round: members in dict order
        for r in range(R):
            d = nd.witnesses.get(r, {})
            wit_order.append(np.array([pk_ix[pk] for pk in d.keys()], np.int32))
            for pk, h in d.items():
                wit[r, pk_ix[pk]] = ix[h]
        fam = np.full(N, -1, np.int8)
        for h, v in nd.famous.items():
            fam[ix[h]] = 1 if v else 0
        cons = np.zeros(R, np.uint8)
        for r in nd.consensus:
            cons[r] = 1
        votes = []
        for y, dct in nd.votes.items():
            for x, v in dct.items():
                votes.append((ix[y], ix[x], 1 if v else 0))
        votes = np.array(sorted(votes), np.int32).reshape(-1, 3)
        tx = np.array([ix[h] for h in nd.transactions], np.int32)
        tbd = np.zeros(N, np.uint8)
        for h in nd.tbd:
            tbd[ix[h]] = 1
        return dict(round=rnd, height=height, can_see=cs, witnesses=wit, wit_order=wit_order,
                    famous=fam, consensus=cons, votes=votes, transactions=tx, tbd=tbd)
