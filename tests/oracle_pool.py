"""Test scaffolding: the CPU oracle runs of the HEAVY parity cases (BASELINE.json configs[2..4]
sizes, 1024 members, hot-member streams), started in background threads when the GPU session
begins so that their single-core minutes overlap the rest of the suite (the oracle's C calls
release the GIL; the GPU box has hundreds of host cores).  Never imported by the product."""
import importlib
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# name -> (members, events, seed, generator mode, p0, p1, chunk or None)
HEAVY = {
    # BASELINE.json configs[2]: the headline config, in full
    "c3_256x1M": (256, 1_000_000, 3, 0, 0.0, 0.0, None),
    # 1024 members (16 mask words): ~8 rounds of uniform gossip, batch and chunked schedules
    "n1024_uniform": (1024, 100_000, 81, 0, 0.0, 0.0, None),
    "n1024_uniform_chunked": (1024, 60_000, 82, 0, 0.0, 0.0, 17_000),
    # 1024 members, two cliques with 2 % cross traffic (the C5-style stress shape)
    "n1024_cliques": (1024, 100_000, 83, 1, 0.02, 0.0, None),
    # 13 of 256 members create 96 % of the events (long chains per round: gallop territory)
    "hot_256x400k": (256, 400_000, 84, 2, 0.95, 0.002, None),
    # BASELINE.json configs[4]-style coin-round stress at a size the oracle can follow: 35 % of the
    # members are 50x less active, elections run to distance 27 with ~46 k coin-round votes
    # (swirld.py:267-272); batch and chunked call schedules
    "coin_256x200k": (256, 200_000, 85, 2, 0.35, 0.02, None),
    "coin_256x200k_chunked": (256, 200_000, 85, 2, 0.35, 0.02, 23_000),
    # configs[4]'s WIDTH with coin rounds reached (VERDICT r3 missing #2): 1024 members, 40 % of them 50x less active —
    # the prefix of the 1.5 M-event stress stream of test_1024_members_coin_stress_properties that the oracle can follow
    # (~4 minutes of one core): elections to distance 6, 76 k coin-round votes (swirld.py:267-272 through k_elections_wide)
    "n1024_coin_200k": (1024, 200_000, 86, 2, 0.40, 0.02, None),
}


if os.environ.get("SW_DRYRUN") == "1":
    # tests/dryrun_heavy.py: the LOGIC of the heavy GPU tests on the CPU (oracle-backed Hashgraph), tiny sizes
    HEAVY = {
        "c3_256x1M": (16, 20000, 3, 0, 0.0, 0.0, None),
        "n1024_uniform": (40, 9000, 81, 0, 0.0, 0.0, None),
        "n1024_uniform_chunked": (40, 6000, 82, 0, 0.0, 0.0, 1700),
        "n1024_cliques": (40, 9000, 83, 1, 0.02, 0.0, None),
        "hot_256x400k": (32, 20000, 84, 2, 0.8, 0.01, None),
        "coin_256x200k": (24, 12000, 85, 2, 0.35, 0.02, None),
        "coin_256x200k_chunked": (24, 12000, 85, 2, 0.35, 0.02, 2300),
        "n1024_coin_200k": (24, 12000, 86, 2, 0.40, 0.02, None),
    }


class OracleRun:
    def __init__(self, name):
        self.name = name
        self.n, self.N, self.seed, self.mode, self.p0, self.p1, self.chunk = HEAVY[name]
        self.oracle = None
        self.new_c = None   # list of new_c per decide_fame call
        self.seconds = 0.0
        self.stream = None

    def run(self):
        from oracle.oracle import Oracle
        pkg = importlib.import_module("py-swirld_amd")
        t0 = time.time()
        self.stream = pkg.synth_hashgraph(self.n, self.N, self.seed, self.mode, self.p0, self.p1)
        cr, sp, op, t, sig = self.stream
        o = Oracle(self.n)
        ncs = []
        chunk = self.chunk or self.N
        for a in range(0, self.N, chunk):
            b = min(self.N, a + chunk)
            o.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            o.divide_rounds(a, b - a)
            ncs.append([int(r) for r in o.decide_fame()])
        self.oracle, self.new_c = o, ncs
        self.seconds = time.time() - t0
        return self


class OraclePool:
    def __init__(self, names=None):
        # build / load both libraries once, before any thread touches them
        importlib.import_module("py-swirld_amd.build").build()
        from oracle import oracle as _o
        _o.lib()
        self._ex = ThreadPoolExecutor(max_workers=8, thread_name_prefix="oracle")
        self._fut = {}
        self._lock = threading.Lock()
        for nm in (HEAVY if names is None else names):
            self.start(nm)

    def start(self, name):
        with self._lock:
            if name not in self._fut:
                self._fut[name] = self._ex.submit(OracleRun(name).run)

    def get(self, name, timeout=1500):
        self.start(name)
        return self._fut[name].result(timeout=timeout)

    def drop(self, name):
        """Release a finished run (its can_see table can be gigabytes)."""
        with self._lock:
            self._fut.pop(name, None)

    def shutdown(self):
        self._ex.shutdown(wait=False, cancel_futures=True)


def compare_state(h, o, n_events, can_see_step=100_000, can_see_rows=None):
    """Every piece of divide_rounds / decide_fame state of the HIP context `h` against the oracle
    `o` (bit-exact): round, witness table, famous, consensus, can_see rows, V / P2 counters."""
    assert np.array_equal(h.rounds(), o.round), "round"
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses()), "witness table"
    m = wit >= 0
    fam = h.famous()
    assert np.array_equal(fam[m], o.famous_by_event[wit[m]]), "famous"
    assert (fam[~m] == -1).all()
    assert np.array_equal(h.consensus(), o.consensus()), "consensus"
    ocs = o.can_see
    if can_see_rows is None:
        for a in range(0, n_events, can_see_step):
            k = min(can_see_step, n_events - a)
            assert np.array_equal(h.can_see(a, k), ocs[a:a + k]), "can_see rows %d.." % a
    else:
        for a, k in can_see_rows:
            assert np.array_equal(h.can_see(a, k), ocs[a:a + k]), "can_see rows %d.." % a
