"""Drop-in `Node` / `Event` with the reference's API (swirld.py:30, 36-328) whose virtual
voting — divide_rounds / decide_fame / find_order and the can_see reachability table they
sit on — runs on the GPU through the C-ABI (include/swirld_hip.h).

Same constructor, attribute names, method names, return values and call protocol as the
reference class, so code written against `swirld.Node` (its `main()` loop, `test()`, the
viz app that reads `round` / `famous` / `idx` / `height` / `hg`) keeps working:

    Node(kp, network, n_nodes, stake)               swirld.py:38
    new_event / is_valid_event / add_event           swirld.py:82-120   (host, unchanged role)
    sync / ask_sync                                  swirld.py:122-161  (host, unchanged role)
    divide_rounds(events) -> None                    swirld.py:187      -> sw_divide_rounds
    decide_fame() -> set of rounds                   swirld.py:224      -> sw_decide_fame
    find_order(new_c) -> None                        swirld.py:280      -> sw_find_order
    main() generator                                 swirld.py:315

State that the reference keeps in dicts keyed by 32-byte hashes lives in HBM keyed by dense
indices; `round`, `can_see`, `witnesses`, `famous` are lazy read-only Mapping views over it
(host glue keeps hash -> index and pk -> member maps).  The hot path has no CPU fallback.
"""
from collections import namedtuple
from collections.abc import Mapping
from pickle import dumps, loads
from time import time

import numpy as np

from . import crypto
from ._lib import SwirldHipError
from .engine import Hashgraph
from .hgutils import bfs, randrange, toposort

C = 6  # coin-round period, swirld.py:17


def majority(it):
    """Stake-weighted vote tally; a tie counts as True (swirld.py:20-27)."""
    no = yes = 0
    for stake, vote in it:
        if vote:
            yes += stake
        else:
            no += stake
    return (False, no) if no > yes else (True, yes)


Event = namedtuple("Event", "d p t c s")  # payload, parents, time, creator pk, signature


class _RoundView(Mapping):
    """Node.round: {event hash -> round number} (swirld.py:51-52)."""

    def __init__(self, node):
        self._n = node

    def __getitem__(self, h):
        i = self._n._index[h]
        if i >= self._n._divided:
            raise KeyError(h)
        return int(self._n._rounds()[i])

    def __iter__(self):
        return iter(self._n._ids[: self._n._divided])

    def __len__(self):
        return self._n._divided


class _CanSeeView(Mapping):
    """Node.can_see: {event -> {member pk -> latest event of that member it sees}}
    (swirld.py:69-72); rows are fetched from HBM on demand."""

    def __init__(self, node):
        self._n = node

    def __getitem__(self, h):
        nd = self._n
        i = nd._index[h]
        if i >= nd._divided:
            raise KeyError(h)
        row = nd._dev.can_see(i, 1)[0]
        return {nd._members[c]: nd._ids[k] for c, k in enumerate(row) if k >= 0}

    def __iter__(self):
        return iter(self._n._ids[: self._n._divided])

    def __len__(self):
        return self._n._divided


class _WitnessView(Mapping):
    """Node.witnesses: {round -> {member pk -> witness hash}}; inner dicts are in
    registration order (ascending event index), which decide_fame relies on."""

    def __init__(self, node):
        self._n = node

    def _table(self):
        return self._n._witness_table()

    def __getitem__(self, r):
        tab = self._table()
        if not 0 <= r < tab.shape[0]:
            return {}  # the reference's defaultdict would create an empty dict
        row = tab[r]
        if self._n._dev.exact:  # forks: a sibling replaces a member's witness but keeps its dict position
            order = [int(c) for c in self._n._dev.witness_order(r)]
        else:
            order = [c for c in np.argsort(np.where(row >= 0, row, np.iinfo(np.int32).max), kind="stable") if row[c] >= 0]
        return {self._n._members[c]: self._n._ids[row[c]] for c in order}

    def __iter__(self):
        return iter(range(self._table().shape[0]))

    def __len__(self):
        return self._table().shape[0]


class _FamousView(Mapping):
    """Node.famous: {witness hash -> bool}, decided witnesses only (swirld.py:64, 263)."""

    def __init__(self, node):
        self._n = node

    def _dict(self):
        return self._n._famous_dict()

    def __getitem__(self, h):
        return self._dict()[h]

    def __iter__(self):
        return iter(self._dict())

    def __len__(self):
        return len(self._dict())


class VotesUnavailable(LookupError):
    """Node.votes is not kept in this state: the context stored a forked event (the exact path keeps two
    vote layers, not the history: sw_get_vote -> SW_ENOTSUP) or fame was committed from a partitioned
    decide_fame (sw_commit_fame).  Deliberately NOT a KeyError: `votes.get(...)` / `in` must not read
    "no such entry" where the answer is unknown.  round / witnesses / famous / consensus are unaffected."""


class _VoterVotes(Mapping):
    """Node.votes[y]: {candidate witness hash -> bool} of one voter."""

    def __init__(self, node, y):
        self._n, self._y = node, y

    def _slot(self, h):
        nd = self._n
        e = nd._index[h]
        if e >= nd._divided:  # known event, not yet through divide_rounds: no round, hence no votes
            raise KeyError(h)
        return int(nd._rounds()[e]), nd._mindex[nd.hg[h].c]

    def _vote(self, rv, mv, rc, mc):
        # SW_ENOTSUP from sw_get_vote (exact path; fame committed from a partitioned decide_fame, whose deciding
        # voters are not recorded): "unknown", not "no such entry" 
        try:
            return self._n._dev.vote(rv, mv, rc, mc)
        except SwirldHipError as exc:
            if exc.code == -95:
                raise VotesUnavailable(str(exc)) from None
            raise

    def __getitem__(self, x):
        nd = self._n
        (rv, mv), (rc, mc) = self._slot(self._y), self._slot(x)
        wit = nd._witness_table()
        if wit[rv, mv] != nd._index[self._y] or wit[rc, mc] != nd._index[x]:
            raise KeyError(x)
        v = self._vote(rv, mv, rc, mc)
        if v < 0:
            raise KeyError(x)
        return bool(v)

    def __iter__(self):
        nd = self._n
        rv, mv = self._slot(self._y)
        wit = nd._witness_table()
        for rc in range(rv):
            for mc in np.argsort(np.where(wit[rc] >= 0, wit[rc], np.iinfo(np.int32).max), kind="stable"):
                if wit[rc, mc] >= 0 and self._vote(rv, mv, rc, int(mc)) >= 0:
                    yield nd._ids[wit[rc, mc]]

    def __len__(self):
        return sum(1 for _ in self)


class _VotesView(Mapping):
    """Node.votes: {voter witness -> {candidate witness -> bool}} (swirld.py:60-61).  The GPU
    elections keep votes as per-round member bitmasks; entries are recomputed on demand
    (sw_get_vote) with the reference's full semantics, the decide_fame() call schedule included: a
    voter has an entry for every candidate it evaluated, in whichever call, before that candidate
    was decided (Appendix A Q8/Q9)."""

    def __init__(self, node):
        self._n = node

    def _check(self):
        if self._n._dev.exact:
            raise VotesUnavailable("votes are not kept once a forked event was stored (exact path)")

    def __getitem__(self, y):
        self._check()
        if y not in self._n._index:
            raise KeyError(y)
        return _VoterVotes(self._n, y)

    def __iter__(self):
        self._check()
        nd = self._n
        wit = nd._witness_table()
        for r in range(1, wit.shape[0]):
            for m in np.argsort(np.where(wit[r] >= 0, wit[r], np.iinfo(np.int32).max), kind="stable"):
                if wit[r, m] >= 0:
                    yield nd._ids[wit[r, m]]

    def __len__(self):
        return sum(1 for _ in self)


class Node:
    # ask_sync from the device-resident state (sw_sync_diff) instead of the reference's
    # height-pruned BFS over Python dicts.  Same SET of events for honest askers; the reply dict is then
    # filled member by member instead of in BFS order, which changes nothing for the receiver except the
    # iteration order of a Python set (and with it the order concurrent events are added in) — so the
    # default keeps the BFS, which makes whole simulations reproduce the reference event for event.
    device_sync_diff = False
    # Batched is_valid_event crypto on the GPU (sw_crypto_verify_batch / sw_crypto_hash_batch, SURVEY.md
    # §8f N3) for sync payloads of at least this many events; smaller payloads go through libsodium on
    # the host (one signature costs a GPU thread ~1-2 ms of latency, a CPU core ~60 us).  None = never.
    device_crypto_threshold = 512

    def __init__(self, kp, network, n_nodes, stake, device=0, accept_forks=True):
        self.pk, self.sk = kp
        # Forked events (two events of a member on one self-parent; the reference stores them, README.md:84).
        # True [default]: stored as in the reference (swirld.py:104-112 has no fork detection); the device
        # context then moves to the exact path (csrc/exact.hip.h: identical results, one wavefront — a
        # Byzantine fork slows this node down, it does not stop it).  False: a forked event and everything
        # built on it is dropped in is_valid_event, which keeps the round-synchronous device path but cuts
        # this node off from every honest member that did accept the sibling (their later events have an
        # ancestor this node refuses) — only for closed simulations without equivocation.
        self.accept_forks = bool(accept_forks)
        self.network = network  # {pk -> Node.ask_sync}
        self.n = n_nodes
        self.stake = stake
        self.tot_stake = sum(stake.values())
        self.min_s = 2 * self.tot_stake / 3

        self.hg = {}          # {event hash -> Event}
        self.head = None
        self.tbd = set()      # events whose final order is not decided yet
        self.transactions = []
        self.idx = {}
        self.consensus = set()
        self.height = {}

        # dense indices for the device
        self._members = list(stake.keys())
        if len(self._members) != n_nodes:
            raise ValueError("stake must have one entry per member")
        self._mindex = {pk: i for i, pk in enumerate(self._members)}
        self._ids = []
        self._index = {}
        self._chain_head = {}  # {member pk -> its newest event in this view} (fork detection)
        self._trunk_height = {}  # {member pk that forked -> height of its earliest forked self-parent (-1: second root)}
        self._chains = [[] for _ in range(n_nodes)]  # per member: its events in self-parent order (hashes)
        self._pending = []    # (creator, self_parent, other_parent, t, sig) not yet uploaded
        self._uploaded = 0
        self._divided = 0
        self._device = device
        self._dev = Hashgraph(n_nodes, [stake[pk] for pk in self._members], coin_period=C, device=device)
        self._dev.set_forks(self.accept_forks)
        self._round_cache = np.zeros(0, np.int32)
        self._wit_cache = None
        self._fam_cache = None

        self.round = _RoundView(self)
        self.can_see = _CanSeeView(self)
        self.witnesses = _WitnessView(self)
        self.famous = _FamousView(self)
        self.votes = _VotesView(self)

        # the node's own root event (swirld.py:75-80)
        h, ev = self.new_event(None, ())
        self.add_event(h, ev)
        self.divide_rounds((h,))
        self.head = h

    # ------------------------------------------------------------------ gossip side (host)
    def new_event(self, d, p):
        """Create, sign and hash a new event of this node (swirld.py:82-95)."""
        assert p == () or len(p) == 2
        assert p == () or self.hg[p[0]].c == self.pk   # first parent is the self-parent
        assert p == () or self.hg[p[1]].c != self.pk   # second parent is someone else's
        t = time()
        s = crypto.sign_detached(dumps((d, p, t, self.pk)), self.sk)
        ev = Event(d, p, t, self.pk, s)
        return crypto.generichash(dumps(ev)), ev

    def _signature_ok(self, ev):
        try:
            crypto.verify_detached(ev.s, dumps(ev[:-1]), ev.c)
            return True
        except ValueError:
            return False

    def _parents_ok(self, ev):
        if ev.p == ():
            return True
        if len(ev.p) != 2 or any(p not in self.hg for p in ev.p):
            return False
        mine, theirs = (self.hg[p].c for p in ev.p)
        return mine == ev.c and theirs != ev.c

    def _not_a_fork(self, h, ev):
        """DEVIATION from the reference (which has no fork detection, swirld.py:88-89, 110-112,
        README.md:84): an event whose self-parent is not its creator's newest event in this view
        (or a second root) is rejected, and with it everything built on top of it.  The device path
        needs one self-parent chain per member; dropping the fork keeps an honest node running
        where storing it would make every later divide_rounds fail."""
        if self.accept_forks or h in self.hg:  # (already accepted: sync re-validates the remote head, swirld.py:138)
            return True
        return self._chain_head.get(ev.c) == (ev.p[0] if ev.p else None)

    def is_valid_event(self, h, ev, _crypto=None):
        """Signature, hash and parent checks (swirld.py:97-108).  `_crypto` = (signature ok, event id)
        when both were precomputed by a device batch."""
        if _crypto is None:
            sig_ok, hid = self._signature_ok(ev), crypto.generichash(dumps(ev))
        else:
            sig_ok, hid = _crypto
        return bool(sig_ok) and hid == h and self._parents_ok(ev) and self._not_a_fork(h, ev)

    def _batch_crypto(self, eids, events):
        """{event id -> (signature ok, BLAKE2b-256 of the pickled event)} for a whole sync payload, on
        the GPU.  Malformed signatures / keys (wrong type or length) are simply invalid."""
        from .engine import hash_batch, verify_batch
        msgs, sigs, pks, whole, good = [], [], [], [], []
        for eid in eids:
            ev = events[eid]
            wf = isinstance(ev.s, (bytes, bytearray)) and len(ev.s) == 64 and isinstance(ev.c, (bytes, bytearray)) and len(ev.c) == 32
            good.append(wf)
            msgs.append(dumps(ev[:-1]))
            sigs.append(bytes(ev.s) if wf else b"\0" * 64)
            pks.append(bytes(ev.c) if wf else b"\0" * 32)
            whole.append(dumps(ev))
        ok = verify_batch(msgs, sigs, pks, device=self._device)
        ids = hash_batch(whole, device=self._device)
        return {eid: (bool(ok[i]) and good[i], ids[i]) for i, eid in enumerate(eids)}

    def add_event(self, h, ev):
        """Store an event (swirld.py:114-120); it is uploaded with the next divide_rounds."""
        self.hg[h] = ev
        if ev.c in self._chain_head and self._chain_head[ev.c] != (ev.p[0] if ev.p else None):
            # a fork of member ev.c: everything of it above the forked self-parent may be on a branch a peer lacks
            th = self.height[ev.p[0]] if ev.p else -1
            self._trunk_height[ev.c] = min(self._trunk_height.get(ev.c, th), th)
        self._chain_head[ev.c] = h
        self.tbd.add(h)
        self.height[h] = 1 + max(self.height[p] for p in ev.p) if ev.p else 0
        self._index[h] = len(self._ids)
        self._ids.append(h)
        self._chains[self._mindex[ev.c]].append(h)
        sp, op = (self._index[ev.p[0]], self._index[ev.p[1]]) if ev.p else (-1, -1)
        self._pending.append((self._mindex[ev.c], sp, op, float(ev.t), ev.s))

    def _known_heights(self):
        """{member pk -> height of the newest event of that member my head can see}: what a
        peer needs to know to send only what I am missing (swirld.py:125-126)."""
        hd = self._index[self.head]
        if hd >= self._divided:
            raise KeyError(self.head)
        kh = self._dev.known_heights(hd)  # one device call: heights of the can_see[head] entries
        return {self._members[c]: int(v) for c, v in enumerate(kh) if v >= 0}

    def sync(self, pk, payload):
        """Pull-sync with `pk`; returns the new event ids in topological order
        (swirld.py:122-146)."""
        request = crypto.sign(dumps(self._known_heights()), self.sk)
        reply = crypto.sign_open(self.network[pk](self.pk, request), pk)
        remote_head, remote_hg = loads(reply)
        unknown = remote_hg.keys() - self.hg.keys()
        new = tuple(toposort(unknown, lambda u: remote_hg[u].p))
        thr = self.device_crypto_threshold
        if not crypto.HAVE_SODIUM:  # stand-in signatures are keyed hashes: the device verifier (real Ed25519) would refuse all of them
            thr = None
        pre = self._batch_crypto(new, remote_hg) if thr is not None and len(new) >= thr else {}
        # Only what was actually stored is returned (main() hands it to divide_rounds): the reference
        # returns the rejected ids too and then fails on them (swirld.py:134-146, 326), and references an
        # unbound `h` when the remote head itself is rejected; here a bad payload costs the step, not the node.
        added = []
        for eid in new:
            if self.is_valid_event(eid, remote_hg[eid], pre.get(eid)):
                self.add_event(eid, remote_hg[eid])
                added.append(eid)
        if remote_head in remote_hg and self.is_valid_event(remote_head, remote_hg[remote_head]):
            h, ev = self.new_event(payload, (self.head, remote_head))
            assert self.is_valid_event(h, ev)
            self.add_event(h, ev)
            self.head = h
            added.append(h)
        return tuple(added)

    def ask_sync(self, pk, info):
        """Answer a sync request with every event the asker cannot know yet: walk back from
        my head, not descending below what the asker reported per member (swirld.py:148-161)."""
        asker_heights = loads(crypto.sign_open(info, pk))
        # (a context on the exact path — it has stored a fork — has no sw_sync_diff and `_chains` is no longer
        # one chain per member: the BFS below serves it, so a peer cannot break ask_sync by delivering a fork)
        if self.device_sync_diff and not self._dev.exact and self._index[self.head] < self._divided:
            known = np.full(self.n, -1, np.int32)
            for creator, hgt in asker_heights.items():
                c = self._mindex.get(creator)
                if c is not None:
                    known[c] = min(int(hgt), 0x7FFFFFFF) if hgt >= 0 else -1
            first, end, _ = self._dev.sync_diff(self._index[self.head], known)
            subset = {self.head: self.hg[self.head]}
            for c in range(self.n):
                for eid in self._chains[c][first[c]:end[c]]:
                    subset[eid] = self.hg[eid]
            return crypto.sign(dumps((self.head, subset)), self.sk)

        # EXTENSION of the reference's height-pruned diff (swirld.py:154-161), active only for members this
        # node has seen fork: one reported height cannot tell which BRANCH the asker has (fork siblings
        # can even have equal heights), so for such a member the walk is pruned at its trunk — the part
        # below its earliest fork point, which every branch contains — instead of at the reported height.
        # Without it two honest nodes holding different siblings can never exchange them, and each
        # refuses everything the other builds afterwards (unknown parent).
        # COST: from the first fork of member c on, every answer re-sends c's events above the fork point to every
        # asker, for good (a second root: c's whole chain) — the asker drops what it already has, but pickling, signing and
        # transfer grow with c's history.  One equivocator therefore makes sync payloads O(history of that member); bounding
        # it needs the asker to report its branch tips for forked members (a protocol change the reference does not have).
        trunk = self._trunk_height

        def missing_parents(u):
            for p in self.hg[u].p:
                creator = self.hg[p].c
                known = asker_heights.get(creator)
                if known is not None and creator in trunk:
                    known = min(known, trunk[creator])
                if known is None or self.height[p] > known:
                    yield p

        subset = {}
        for eid in bfs((self.head,), missing_parents):
            subset[eid] = self.hg[eid]
        return crypto.sign(dumps((self.head, subset)), self.sk)

    def ancestors(self, c):
        """Self-parent chain of c, newest first (swirld.py:163-168)."""
        while True:
            yield c
            if not self.hg[c].p:
                return
            c = self.hg[c].p[0]

    def higher(self, a, b):
        return a is not None and (b is None or self.height[a] >= self.height[b])

    def maxi(self, a, b):
        return a if self.higher(a, b) else b

    # ------------------------------------------------------------------ virtual voting (GPU)
    def _flush(self):
        if self._pending:
            cr, sp, op, t, sig = zip(*self._pending)
            self._dev.append_events(np.array(cr, np.int32), np.array(sp, np.int32), np.array(op, np.int32),
                                    np.array(t, np.float64),
                                    np.frombuffer(b"".join(sig), np.uint8).reshape(len(sig), 64))
            self._uploaded += len(self._pending)
            self._pending = []

    def divide_rounds(self, events):
        """Assign rounds / witnesses / can_see rows to the new events (swirld.py:187-222).
        `events` must be the not-yet-divided events in the order they were added (which is
        what sync() returns and main() passes)."""
        events = tuple(events)
        if not events:
            return
        first = self._divided
        expect = self._ids[first:first + len(events)]
        if list(events) != expect:
            for h in events:
                if h not in self.hg:
                    raise KeyError(h)
            raise ValueError("divide_rounds expects the undivided events in the order they were added")
        self._flush()
        self._dev.divide_rounds(first, len(events))
        self._divided = first + len(events)
        self._wit_cache = None

    def decide_fame(self):
        """Run the witness elections; returns the set of newly decided rounds
        (swirld.py:224-277)."""
        new_c = {int(r) for r in self._dev.decide_fame()}
        self.consensus |= new_c
        self._fam_cache = None
        return new_c

    def find_order(self, new_c):
        """Extend the total order with the events the newly decided rounds receive
        (swirld.py:280-311)."""
        order = self._dev.find_order(new_c)
        final = [self._ids[i] for i in order]
        for i, x in enumerate(final):
            self.idx[x] = i + len(self.transactions)
        self.tbd.difference_update(final)
        self.transactions += final
        if self.consensus:
            print(self.consensus)

    def main(self):
        """Main working loop: `payload = yield new_event_ids` (swirld.py:315-328)."""
        new = ()
        while True:
            payload = (yield new)
            c = tuple(self.network.keys() - {self.pk})[randrange(self.n - 1)]
            new = self.sync(c, payload)
            self.divide_rounds(new)
            new_c = self.decide_fame()
            self.find_order(new_c)

    # ------------------------------------------------------------------ view plumbing
    def _rounds(self):
        have = self._round_cache.shape[0]
        if have < self._divided:
            self._round_cache = np.concatenate([self._round_cache, self._dev.rounds(have, self._divided - have)])
        return self._round_cache

    def _witness_table(self):
        if self._wit_cache is None:
            self._wit_cache = self._dev.witnesses()
        return self._wit_cache

    def _famous_dict(self):
        if self._fam_cache is None and self._dev.exact:
            # forks: a replaced witness keeps its entry (swirld.py:64 is keyed by event), so the per-event view
            fe = self._dev.famous_events(0, self._divided)
            self._fam_cache = {self._ids[int(e)]: bool(fe[e]) for e in np.flatnonzero(fe >= 0)}
        if self._fam_cache is None:
            wit, fam = self._witness_table(), self._dev.famous()
            d = {}
            for r in range(wit.shape[0]):
                row = wit[r]
                for c in np.argsort(np.where(row >= 0, row, np.iinfo(np.int32).max), kind="stable"):
                    if row[c] >= 0 and fam[r, c] >= 0:
                        d[self._ids[row[c]]] = bool(fam[r, c])
            self._fam_cache = d
        return self._fam_cache


def test(n_nodes, n_turns, device=0):
    """The reference's simulation driver (swirld.py:331-345): n_nodes nodes sharing an
    in-process 'network', stepped at random."""
    kps = [crypto.sign_keypair() for _ in range(n_nodes)]
    network = {}
    stake = {kp[0]: 1 for kp in kps}
    nodes = [Node(kp, network, n_nodes, stake, device=device) for kp in kps]
    for n in nodes:
        network[n.pk] = n.ask_sync
    mains = [n.main() for n in nodes]
    for m in mains:
        next(m)
    for i in range(n_turns):
        r = randrange(n_nodes)
        print("working node: %i, event number: %i" % (r, i))
        next(mains[r])
    return nodes
