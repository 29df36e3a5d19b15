"""GPU (-m gpu): every single-GPU BASELINE.json configuration under bit-exact parity with the
sequential CPU oracle, collected by default (no opt-in switch, nothing skipped):

  configs[2]  256 members / 1 M events in full — every round, every can_see row, witness table,
              famous, consensus, new_c, V / P2 counters and the total order; default path and
              SW_GALLOP=2 (strided candidate windows);
  configs[3]  256 members / 10 M events on one GPU: the first 1 M events against the oracle run of
              configs[2] (same generator seed => same stream prefix; round / can_see / witnesses of a
              prefix depend on the prefix only), the other 9 M through invariants of
              swirld.py:187-222 and schedule independence (pipelined vs unpipelined sweeps);
  1024 members (16 mask words: the widest kernels) — uniform gossip and two cliques with 2 % cross
              traffic, batch and chunked call schedules;
  hot members 13 of 256 members create 96 % of the events (thousands of chain positions per round),
              default path and SW_GALLOP=2.

The oracle runs take about a minute each on one core; tests/oracle_pool.py computes them in
background threads from the start of the session, and these tests run last."""
import os

import numpy as np
import pytest

from oracle_pool import HEAVY, compare_state

pytestmark = pytest.mark.gpu

WIDE_STRESS = (1024, 1_500_000, 300_000)  # members, events, prefix of test_1024_members_coin_stress_properties
if os.environ.get("SW_DRYRUN") == "1":
    WIDE_STRESS = (24, 20000, 6000)


def hip_run(pkg, run, monkeypatch=None, env=None):
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    cr, sp, op, t, sig = run.stream
    h = pkg.Hashgraph(run.n)
    ncs = []
    chunk = run.chunk or run.N
    for a in range(0, run.N, chunk):
        b = min(run.N, a + chunk)
        h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        h.divide_rounds(a, b - a)
        ncs.append([int(r) for r in h.decide_fame()])
    return h, ncs


@pytest.mark.heavy("c3_256x1M")
@pytest.mark.parametrize("gallop", ["0", "2"])
def test_config3_one_million_events_bit_exact(pkg, oracle_pool, monkeypatch, gallop):
    run = oracle_pool.get("c3_256x1M")
    o = run.oracle
    # (the default run takes round[] and the sees-masks from the band pass; the gallop variant finalizes every event from its
    # row, with the early finalize of the last sub-batch — events below the band of the round in progress, beside the rest of
    # its loop — in the middle of that loop and the unthrottled launch shape)
    h, ncs = hip_run(pkg, run, monkeypatch, {"SW_GALLOP": gallop} if gallop == "0" else
                     {"SW_GALLOP": gallop, "SW_FIN_BAND": "0", "SW_MID_PCT": "50", "SW_FIN_BLOCKS": "8192", "SW_ELECT_CG": "64"})
    assert ncs == run.new_c
    if gallop == "0":
        compare_state(h, o, run.N)
        tx_h = h.find_order(ncs[0])
        tx_o = o.find_order(run.new_c[0])  # (the oracle orders once: the gallop variant skips it)
        assert np.array_equal(tx_h, tx_o)
    else:
        compare_state(h, o, run.N, can_see_rows=[(0, 4096), (run.N - 4096, 4096)])
    c, co = h.counters(), o.counters()
    assert c["rounds"] == co["rounds"]
    assert c["voter_evals"] == co["voter_evals"] and c["majority_evals"] == co["majority_evals"]
    h.close()


@pytest.mark.heavy("n1024_uniform", "n1024_uniform_chunked", "n1024_cliques")
@pytest.mark.parametrize("name", ["n1024_uniform", "n1024_uniform_chunked", "n1024_cliques"])
def test_1024_members_bit_exact(pkg, oracle_pool, name):
    run = oracle_pool.get(name)
    o = run.oracle
    assert o.max_round >= 3, "the case must span several rounds"
    h, ncs = hip_run(pkg, run)
    assert ncs == run.new_c
    compare_state(h, o, run.N, can_see_step=20_000)
    if run.chunk is None:
        c, co = h.counters(), o.counters()
        assert c["voter_evals"] == co["voter_evals"] and c["majority_evals"] == co["majority_evals"]
        assert np.array_equal(h.find_order(ncs[0]), o.find_order(run.new_c[0]))
    h.close()
    oracle_pool.drop(name)


@pytest.mark.heavy("hot_256x400k")
@pytest.mark.parametrize("gallop", ["0", "2"])
def test_hot_members_bit_exact(pkg, oracle_pool, monkeypatch, gallop):
    run = oracle_pool.get("hot_256x400k")
    o = run.oracle
    assert o.max_round >= 4
    h, ncs = hip_run(pkg, run, monkeypatch, {"SW_GALLOP": gallop})
    assert ncs == run.new_c
    compare_state(h, o, run.N, can_see_rows=[(0, 50_000), (run.N - 50_000, 50_000)])
    h.close()


@pytest.mark.heavy("coin_256x200k", "coin_256x200k_chunked")
@pytest.mark.parametrize("name", ["coin_256x200k", "coin_256x200k_chunked"])
def test_coin_round_stress_bit_exact(pkg, oracle_pool, name):
    """configs[4]-style fame stress (a third of the members nearly silent): elections that run over
    many rounds, coin rounds included (swirld.py:267-272) — the counts of coin-round votes and of
    votes taken from the signature bit must equal the reference algorithm's."""
    run = oracle_pool.get(name)
    o = run.oracle
    co = o.counters()
    assert co["coin_flips"] > (10_000 if run.N >= 100_000 else 0) and co["max_vote_distance"] >= (12 if run.N >= 100_000 else 6)
    h, ncs = hip_run(pkg, run)
    assert ncs == run.new_c
    compare_state(h, o, run.N, can_see_rows=[(0, 20_000), (run.N - 20_000, 20_000)])
    if run.chunk is None:
        c = h.counters()
        for k in ("voter_evals", "majority_evals", "coin_votes", "coin_flips"):
            assert c[k] == co[k], k
        assert np.array_equal(h.find_order(ncs[0]), o.find_order(run.new_c[0]))
    h.close()
    oracle_pool.drop(name)


@pytest.mark.heavy("n1024_coin_200k")
def test_1024_members_coin_rounds_bit_exact(pkg, oracle_pool):
    """Coin rounds (swirld.py:267-272) at the width of configs[4], against the ORACLE: 1024 members, 40 % of them
    nearly silent, 200 k events — every vote counter, the fame table and the rounds must be the reference
    algorithm's (the wide elections kernel with coin rounds was oracle-checked up to 700 members before)."""
    run = oracle_pool.get("n1024_coin_200k", timeout=2400)
    o = run.oracle
    co = o.counters()
    if run.N >= 100_000:
        assert co["coin_votes"] > 10_000 and co["max_vote_distance"] >= 6, "the case must reach coin rounds"
    h, ncs = hip_run(pkg, run)
    assert ncs == run.new_c
    compare_state(h, o, run.N, can_see_rows=[(0, 4096), (run.N // 2, 2048), (run.N - 4096, 4096)])
    c = h.counters()
    for k in ("voter_evals", "majority_evals", "coin_votes", "coin_flips"):
        assert c[k] == co[k], k
    h.close()
    oracle_pool.drop("n1024_coin_200k")


def test_1024_members_coin_stress_properties(pkg):
    """1024 members with 40 % of them nearly silent (the shape of configs[4]; the oracle needs hours at
    this width): invariants of swirld.py:195-222, a prefix re-run, and coin rounds actually reached."""
    n, N, M = WIDE_STRESS
    stream = pkg.synth_hashgraph(n, N, 86, 2, 0.40, 0.02)
    cr, sp, op, t, sig = stream
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    h.divide_rounds(0, N)
    nc = h.decide_fame()
    c = h.counters()
    assert c["coin_votes"] > 0, "the stress stream must reach coin rounds"
    rnd = h.rounds()
    pr = np.maximum(rnd[sp[n:]], rnd[op[n:]])
    assert (rnd[:n] == 0).all() and ((rnd[n:] == pr) | (rnd[n:] == pr + 1)).all()
    wit = h.witnesses()
    is_wit = np.zeros(N, bool)
    is_wit[wit[wit >= 0]] = True
    exp_wit = np.ones(N, bool)
    exp_wit[n:] = rnd[n:] > rnd[sp[n:]]
    assert np.array_equal(is_wit, exp_wit)
    for e in (N - 1, N // 2 + 7, M + 3):
        exp = np.maximum(h.can_see(sp[e], 1)[0], h.can_see(op[e], 1)[0])
        exp[cr[e]] = e
        assert np.array_equal(h.can_see(e, 1)[0], exp)
    h2 = pkg.Hashgraph(n)                               # prefix stability
    h2.append_events(*[a[:M] for a in stream])
    h2.divide_rounds(0, M)
    assert np.array_equal(h2.rounds(), rnd[:M])
    assert np.array_equal(h2.can_see(M - 3000, 3000), h.can_see(M - 3000, 3000))
    w2 = h2.witnesses()   # a prefix knows a witness iff the witness lies inside it (slow members arrive late)
    full = wit[: w2.shape[0]]
    assert np.array_equal(w2, np.where(full < M, full, -1))
    fam = h.famous()
    cons = h.consensus()
    assert len(nc) == int(cons.sum())
    assert ((fam >= 0) | (wit < 0))[cons.astype(bool)].all(), "every witness of a consensus round is decided"
    h.close(); h2.close()


def _digest(h):
    return (int(h.rounds().astype(np.int64).sum()), h.witnesses().tobytes(), h.famous().tobytes(),
            h.consensus().tobytes())


@pytest.mark.heavy("c3_256x1M")
def test_config4_ten_million_events_prefix_and_invariants(pkg, oracle_pool, monkeypatch):
    n, M = HEAVY["c3_256x1M"][:2]
    N = 10 * M  # 256 members, 10 M events
    run = oracle_pool.get("c3_256x1M")
    o = run.oracle
    stream = pkg.synth_hashgraph(n, N, HEAVY["c3_256x1M"][2], 0, 0, 0)
    cr, sp, op, t, sig = stream
    for a, b in zip(stream, run.stream):
        assert np.array_equal(a[:M], b), "the 10 M stream must extend the 1 M stream"
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    h.divide_rounds(0, N)
    nc = h.decide_fame()
    rnd = h.rounds()
    # ---- the 1 M prefix against the oracle
    assert np.array_equal(rnd[:M], o.round)
    ocs = o.can_see
    for a in range(0, M, M // 4):
        k = min(20_000, M - a)
        assert np.array_equal(h.can_see(a, k), ocs[a:a + k])
    Ro = o.max_round + 1
    wit = h.witnesses()
    wo = o.witnesses()
    assert np.array_equal(wo, np.where(wit[:Ro] < M, wit[:Ro], -1))  # a prefix knows exactly the witnesses inside it
    fam, fo, co = h.famous(), o.famous_table(), o.consensus()
    for r in range(Ro - 1):
        if co[r]:  # decided inside the prefix: the longer run decides it the same way
            m = wo[r] >= 0
            assert np.array_equal(fam[r][m], fo[r][m]), "famous of round %d" % r
    # ---- the other 9 M events: invariants of swirld.py:195-222
    assert (rnd[:n] == 0).all()
    pr = np.maximum(rnd[sp[n:]], rnd[op[n:]])
    assert ((rnd[n:] == pr) | (rnd[n:] == pr + 1)).all()
    R = wit.shape[0]
    assert R == h.max_round + 1 and R > 8 * Ro
    is_wit = np.zeros(N, bool)
    is_wit[wit[wit >= 0]] = True
    exp_wit = np.ones(N, bool)
    exp_wit[n:] = rnd[n:] > rnd[sp[n:]]
    assert np.array_equal(is_wit, exp_wit)
    wr, wc = np.nonzero(wit >= 0)
    assert np.array_equal(rnd[wit[wr, wc]], wr) and np.array_equal(cr[wit[wr, wc]], wc)
    cons = h.consensus()
    assert cons[: R - 12].all() and len(nc) == int(cons.sum())
    assert ((fam >= 0) == (wit >= 0))[: R - 12].all()
    # a can_see row far beyond the prefix, recomputed on the host from its parents' rows
    for e in (N - 1, N - 12345, (N // 4) * 3 + 54_321 % M):
        rows = h.can_see(e, 1)[0]
        exp = np.maximum(h.can_see(sp[e], 1)[0], h.can_see(op[e], 1)[0])
        exp[cr[e]] = e
        assert np.array_equal(rows, exp)
    d0 = _digest(h)
    h.close()
    # ---- schedule independence: one unpipelined sweep + round loop gives the same state
    monkeypatch.setenv("SW_PIPE", "1")
    h2 = pkg.Hashgraph(n)
    h2.append_events(*stream)
    h2.divide_rounds(0, N)
    nc2 = h2.decide_fame()
    assert list(nc2) == list(nc) and _digest(h2) == d0
    h2.close()
    oracle_pool.drop("c3_256x1M")
