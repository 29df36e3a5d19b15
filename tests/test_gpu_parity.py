"""GPU (-m gpu): the HIP path, called through the C-ABI, against
 (1) every fork-free golden fixture of the unmodified reference (bit-exact), with the
     recorded call schedule (batch and incremental);
 (2) the CPU oracle on fresh seeded streams at sizes it finishes in seconds, including
     n = 64 / 100k (BASELINE.json configs[1]) and a 256-member prefix;
 (3) size-independent properties at the full 256-member / 1M-event size.
Bit-exact: round, can_see, witness table, famous, consensus, new_c, sees-masks."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

FORKFREE = [n for n in golden_names() if "forks" not in n]


def run_schedule(h, g):
    calls = 0
    for a, b in g["batches"]:
        h.append_events(g["creator"][a:b], g["self_parent"][a:b], g["other_parent"][a:b],
                        g["t"][a:b], g["sig"][a:b])
        h.divide_rounds(a, b - a)
        nc = h.decide_fame()
        exp_nc = g["new_c_flat"][g["new_c_off"][calls]:g["new_c_off"][calls + 1]]
        assert list(nc) == list(exp_nc), "new_c of call %d" % calls
        calls += 1


def assert_state_equal(h, exp_round, exp_cs, exp_wit, exp_fam_by_event, exp_cons):
    assert np.array_equal(h.rounds(), exp_round)
    assert np.array_equal(h.can_see(), exp_cs)
    wit = h.witnesses()
    assert np.array_equal(wit, exp_wit)
    fam = h.famous()
    m = wit >= 0
    assert np.array_equal(fam[m], exp_fam_by_event[wit[m]])
    assert (fam[~m] == -1).all()
    assert np.array_equal(h.consensus(), exp_cons)


@pytest.mark.parametrize("name", FORKFREE)
def test_hip_matches_reference_golden(pkg, name):
    g = load_golden(name)
    h = pkg.Hashgraph(g["n"], g["stake"])
    run_schedule(h, g)
    assert np.array_equal(h.heights(), g["height"])
    assert_state_equal(h, g["round"], g["can_see"], g["witnesses"], g["famous"], g["consensus"])
    for r, order in enumerate(g["wit_order"]):   # iteration order of self.witnesses[r] (swirld.py:234, 240)
        assert np.array_equal(h.witness_order(r), order), "dict order of witnesses[%d]" % r
    assert not h.exact
    h.close()


@pytest.mark.parametrize("name", ["n4_s1_batch", "n4_s3_batch", "n16_s3_batch", "n10_s1_stake", "n64_s1_batch"])
def test_votes_match_reference(pkg, name):
    """Node.votes entries (every one the reference recorded, plus absent ones) for batch
    schedules, through sw_get_vote."""
    g = load_golden(name)
    assert len(g["batches"]) == 1
    h = pkg.Hashgraph(g["n"], g["stake"])
    run_schedule(h, g)
    rnd, cr = g["round"], g["creator"]
    votes = g["votes"]
    step = max(1, len(votes) // 1500)
    have = {(int(y), int(x)) for y, x, _ in votes}
    for y, x, v in votes[::step]:
        assert h.vote(rnd[y], cr[y], rnd[x], cr[x]) == v
    # pairs without an entry in the reference: decided candidates, later voters
    wit = g["witnesses"]
    rng = np.random.default_rng(0)
    R = wit.shape[0]
    checked = 0
    for _ in range(4000):
        rv = int(rng.integers(1, R)); rc = int(rng.integers(0, rv))
        mv = int(rng.integers(0, g["n"])); mc = int(rng.integers(0, g["n"]))
        y, x = wit[rv, mv], wit[rc, mc]
        if y < 0 or x < 0 or (int(y), int(x)) in have:
            continue
        assert h.vote(rv, mv, rc, mc) == -1
        checked += 1
    assert checked > 20
    h.close()


@pytest.mark.parametrize("name", ["n4_s5_chunk1", "n4_s6_chunk7", "n7_s2_chunk13", "n16_s4_chunk50", "n16_s4_chunk250",
                                  "n64_s2_slow_chunk1000", "n4_mainloop_node0"])
def test_votes_match_reference_incremental_schedules(pkg, name):
    """Node.votes after INCREMENTAL call schedules: the reference's dict also holds the entries of
    voters that evaluated a candidate in an earlier decide_fame() call, before a later-arriving
    voter decided it (Appendix A Q8/Q9).  Every entry of the reference and the absence of every
    other (voter, candidate) pair — exhaustively where the table is small."""
    g = load_golden(name)
    assert len(g["batches"]) > 1
    h = pkg.Hashgraph(g["n"], g["stake"])
    run_schedule(h, g)
    rnd, cr, wit = g["round"], g["creator"], g["witnesses"]
    have = {(int(y), int(x)): int(v) for y, x, v in g["votes"]}
    assert len(have) == len(g["votes"])
    R, n = wit.shape
    slots = [(r, m) for r in range(R) for m in range(n) if wit[r, m] >= 0]
    pairs = [(sy, sx) for sy in slots for sx in slots if sx[0] < sy[0]]
    if len(pairs) > 30000:
        rng = np.random.default_rng(1)
        keep = {(int(rnd[y]), int(cr[y]), int(rnd[x]), int(cr[x])) for y, x in have}
        pick = rng.choice(len(pairs), 12000, replace=False)
        sample = [pairs[i] for i in pick] + [((a, b), (c_, d)) for a, b, c_, d in list(keep)[:12000]]
    else:
        sample = pairs
    n_entries = 0
    for (rv, mv), (rc, mc) in sample:
        y, x = int(wit[rv, mv]), int(wit[rc, mc])
        exp = have.get((y, x), -1)
        got = h.vote(rv, mv, rc, mc)
        assert got == exp, "votes[%d][%d]: got %d, reference %d" % (y, x, got, exp)
        n_entries += exp >= 0
    if len(pairs) <= 30000:
        assert n_entries == len(have), "every entry of the reference is a (witness, earlier witness) pair"
    h.close()


def test_fork_is_refused(pkg):
    """sw_set_forks(0): what a Node that drops forked events runs with (default: accepted, tests/test_gpu_forks.py)."""
    g = load_golden("n8_s11_forks")
    h = pkg.Hashgraph(g["n"])
    h.set_forks(False)
    cr, sp, op = g["creator"], g["self_parent"], g["other_parent"]
    with pytest.raises(pkg.SwirldHipError) as ei:
        h.append_events(cr, sp, op, g["t"], g["sig"])
    assert ei.value.code == -95
    assert h.num_events == 0, "a refused append stores nothing"
    # the fork-free prefix is accepted and processed; the context stays usable afterwards
    head, k = {}, 0
    while k < len(cr) and head.get(int(cr[k]), -1) == sp[k]:
        head[int(cr[k])] = k
        k += 1
    assert 0 < k < len(cr)
    h.append_events(cr[:k], sp[:k], op[:k], g["t"][:k], g["sig"][:k])
    h.divide_rounds(0, k)
    with pytest.raises(pkg.SwirldHipError):
        h.append_events(cr[k:k + 1], sp[k:k + 1], op[k:k + 1])
    assert h.num_events == k
    h.decide_fame()
    assert np.array_equal(h.rounds(), g["round"][:k])


def oracle_run(n, stream, stake=None, chunk=None):
    from oracle.oracle import Oracle
    cr, sp, op, t, sig = stream
    N = len(cr)
    o = Oracle(n, stake)
    ncs = []
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        o.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        o.divide_rounds(a, b - a)
        ncs.append(list(o.decide_fame()))
    return o, ncs


def hip_run(pkg, n, stream, stake=None, chunk=None):
    cr, sp, op, t, sig = stream
    N = len(cr)
    h = pkg.Hashgraph(n, stake)
    ncs = []
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        h.divide_rounds(a, b - a)
        ncs.append(list(h.decide_fame()))
    return h, ncs


CASES = [
    # n, N, seed, mode, p0, p1, chunk
    (4, 5000, 31, 0, 0, 0, None),          # coin rounds
    (4, 3000, 32, 0, 0, 0, 17),            # incremental, coin rounds
    (3, 2000, 33, 0, 0, 0, None),
    (2, 500, 34, 0, 0, 0, None),
    (16, 20000, 35, 0, 0, 0, None),
    (16, 20000, 36, 2, 0.25, 0.02, 700),   # slow members, incremental (laggard restarts)
    (40, 20000, 37, 3, 0.6, 0, None),      # stale other-parents
    (64, 100000, 2, 0, 0, 0, None),        # BASELINE.json configs[1]
    (64, 30000, 38, 1, 0.01, 0, 5000),     # two cliques, incremental
    (64, 24000, 44, 0, 0, 0, 25),          # ~1000 tiny appends: chain segments grow and relocate
    (65, 20000, 39, 0, 0, 0, None),        # first size with two mask words
    (128, 30000, 40, 0, 0, 0, None),
    (200, 30000, 41, 2, 0.1, 0.05, None),
    (256, 40000, 3, 0, 0, 0, None),        # prefix of BASELINE.json configs[2]
    (300, 30000, 42, 0, 0, 0, None),       # 8 mask words
    (600, 30000, 43, 0, 0, 0, None),       # 16 mask words
    (300, 40000, 45, 2, 0.35, 0.02, 9000), # 8 mask words, coin-round stress, incremental: the wide elections kernel (k_elections_wide)
    (520, 40000, 46, 2, 0.4, 0.02, None),  # 16 mask words, coin-round stress
    (700, 30000, 47, 1, 0.02, 0, 7000),    # 16 mask words, two cliques, incremental
]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,chunk", CASES)
def test_hip_matches_oracle(pkg, n, N, seed, mode, p0, p1, chunk):
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o, ncs_o = oracle_run(n, stream, chunk=chunk)
    h, ncs_h = hip_run(pkg, n, stream, chunk=chunk)
    assert ncs_h == ncs_o
    assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
    c, co = h.counters(), o.counters()
    assert c["rounds"] == co["rounds"]
    if chunk is None:
        assert c["voter_evals"] == co["voter_evals"]
        assert c["majority_evals"] == co["majority_evals"]
        assert c["coin_votes"] == co["coin_votes"] and c["coin_flips"] == co["coin_flips"]
        if n <= 4:
            assert co["coin_flips"] > 0, "the 4-member cases must exercise coin rounds (swirld.py:267-272)"
    h.close()


@pytest.mark.parametrize("stake", [[1] * 9 + [2], [2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2], [3, 1, 4, 1, 5, 2]])
def test_hip_weighted_stake(pkg, stake):
    n = len(stake)
    stream = pkg.synth_hashgraph(n, 4000, 50 + n, 0, 0, 0)
    st = np.array(stake, np.uint64)
    o, ncs_o = oracle_run(n, stream, stake=st)
    h, ncs_h = hip_run(pkg, n, stream, stake=st)
    assert ncs_h == ncs_o
    assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())


def test_tuning_knobs_do_not_change_results(pkg, monkeypatch):
    """Candidate-list width, band cap and its growth limit only change the schedule, never the
    results.  Small caps exercise cursor retries, FAR candidates decided by inheritance, waiting
    members, band doubling, and (cap == limit) forced tallies with on-the-fly hop masks."""
    cases = [(48, 15000, 60, 2, 0.2, 0.03), (24, 9000, 61, 2, 0.3, 0.004), (16, 6000, 62, 1, 0.01, 0)]
    oracles = {}
    from_rows = 0
    for k, band, band_max in [("4", "64", None), ("4", "100000", None), ("32", "128", None),
                              ("8", "64", "64"), ("16", "256", "512"),
                              ("31", "4096", None), ("63", "256", None), ("1", "4096", None)]:  # candidate-table widths
        monkeypatch.setenv("SW_TALLY_K", k)
        monkeypatch.setenv("SW_BAND", band)
        if band_max:
            monkeypatch.setenv("SW_BAND_MAX", band_max)
        else:
            monkeypatch.delenv("SW_BAND_MAX", raising=False)
        for case in cases:
            n, N, seed, mode, p0, p1 = case
            stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
            if case not in oracles:
                oracles[case] = oracle_run(n, stream)
            o, ncs_o = oracles[case]
            h, ncs_h = hip_run(pkg, n, stream)
            assert ncs_h == ncs_o
            assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
            if band == "64":   # a band capped at 64 events does not cover every event in the pass of its own round:
                from_rows += h.counters()["finalize_from_rows"]   # the finalize check sends those back to their rows
            h.close()
    assert from_rows > 0


@pytest.mark.parametrize("fin_band,band_fast", [("0", "1"), ("1", "1"), ("1", "0"), ("0", "0")])
def test_round_numbers_from_the_band_pass_or_from_the_rows(pkg, monkeypatch, fin_band, band_fast):
    """round[] and the sees-masks (swirld.py:217-219; the voters' hop masks of decide_fame) written by the round loop's band
    pass and checked afterwards (SW_FIN_BAND=1, the default) against every event finalized from its row (0): batch and
    incremental schedules, slow members, two cliques, stale other-parents."""
    monkeypatch.setenv("SW_FIN_BAND", fin_band)
    monkeypatch.setenv("SW_BAND_FAST", band_fast)   # (round 5) full groups of 8 band events through the fixed-index path, or the generic path only
    for n, N, seed, mode, p0, p1, chunk in [(130, 14000, 175, 0, 0, 0, None), (256, 30000, 176, 2, 0.2, 0.05, 7000),
                                            (100, 9000, 177, 1, 0.02, 0, 1500), (200, 20000, 178, 3, 0.6, 0, 333), (40, 8000, 179, 2, 0.5, 0.01, 1)]:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o, ncs_o = oracle_run(n, stream, chunk=chunk)
        h, ncs_h = hip_run(pkg, n, stream, chunk=chunk)
        assert ncs_h == ncs_o
        assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
        if fin_band == "0":
            assert h.counters()["finalize_from_rows"] == 0
        h.close()


@pytest.mark.parametrize("cansee,tally,ring_h", [("2", "1", None), ("2", "1", "1"), ("2", "0", "4"), ("3", "1", "2"), ("3", "0", None),
                                                 ("6", "1", None), ("6", "0", None), ("6", "2", None), ("3", "2", None)])
def test_kernel_variants_agree(pkg, monkeypatch, cansee, tally, ring_h):
    """Both can_see sweeps (level-bucketed with its LDS ring at several depths / dataflow) and both
    tally kernels (column-lane / bit-sliced) against the oracle, on inputs that stress the
    ring (stale other-parents miss it; incremental batches read rows of earlier kernels)."""
    monkeypatch.setenv("SW_CANSEE_IMPL", cansee)
    monkeypatch.setenv("SW_TALLY_IMPL", tally)
    if ring_h:
        monkeypatch.setenv("SW_RING_H", ring_h)
    for n, N, seed, mode, p0, p1, chunk in [(48, 12000, 70, 3, 0.85, 0, None), (256, 20000, 71, 3, 0.7, 0, 6000),
                                            (20, 8000, 72, 2, 0.3, 0.01, 900), (130, 14000, 73, 0, 0, 0, None),
                                            (5, 3000, 74, 0, 0, 0, 40)]:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o, ncs_o = oracle_run(n, stream, chunk=chunk)
        h, ncs_h = hip_run(pkg, n, stream, chunk=chunk)
        assert ncs_h == ncs_o
        assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
        h.close()


@pytest.mark.parametrize("elect,cg", [("0", None), ("1", None), ("1", "64"), ("1", "256")])
def test_election_kernel_variants_agree(pkg, monkeypatch, elect, cg):
    """One thread per candidate (0) and NW threads per candidate in tiles of CG candidates (1, k_elections_tiled: the
    default from 128 members on; at 256 members a round is 4, 2 or 1 workgroups) against the oracle: decisions, consensus
    rounds and the V / P2 counters, batch and incremental schedules, four generator modes."""
    monkeypatch.setenv("SW_ELECT_IMPL", elect)
    if cg:
        monkeypatch.setenv("SW_ELECT_CG", cg)
    for n, N, seed, mode, p0, p1, chunk in [(130, 14000, 75, 0, 0, 0, None), (256, 30000, 76, 2, 0.2, 0.05, 7000),
                                            (100, 9000, 77, 1, 0.02, 0, 1500), (200, 20000, 78, 3, 0.6, 0, None)]:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o, ncs_o = oracle_run(n, stream, chunk=chunk)
        h, ncs_h = hip_run(pkg, n, stream, chunk=chunk)
        assert ncs_h == ncs_o
        assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
        if chunk is None:  # the counters are compared for batch schedules (as in test_hip_matches_oracle)
            ch, co = h.counters(), o.counters()
            assert ch["voter_evals"] == co["voter_evals"] and ch["majority_evals"] == co["majority_evals"]
        h.close()


@pytest.mark.parametrize("skip,gallop,k", [("0", "0", "28"), ("1", "0", "8"), ("2", "2", "4"), ("5", "0", "4"), ("3", "1", "16"),
                                           ("9", "0", "28"), ("2", "0", "63")])
def test_window_offset_and_gallop_agree(pkg, monkeypatch, skip, gallop, k):
    """Window offset of fresh rounds (SW_SKIP) and strided windows (SW_GALLOP) only change which
    candidates a launch looks at, never the results: slow members, cliques, hot members, tiny
    hashgraphs (offset larger than a chain), batch and incremental schedules."""
    monkeypatch.setenv("SW_SKIP", skip)
    monkeypatch.setenv("SW_GALLOP", gallop)
    monkeypatch.setenv("SW_TALLY_K", k)
    for n, N, seed, mode, p0, p1, chunk in [(48, 15000, 95, 2, 0.2, 0.03, None), (16, 6000, 96, 1, 0.01, 0, 700),
                                            (64, 30000, 97, 2, 0.9, 0.004, None), (130, 14000, 98, 0, 0, 0, 3000),
                                            (4, 1500, 99, 0, 0, 0, 9), (256, 30000, 100, 0, 0, 0, None)]:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o, ncs_o = oracle_run(n, stream, chunk=chunk)
        h, ncs_h = hip_run(pkg, n, stream, chunk=chunk)
        assert ncs_h == ncs_o
        assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
        h.close()


@pytest.mark.parametrize("pipe", ["1", "3", "8"])
def test_pipelined_subbatches_match_oracle(pkg, monkeypatch, pipe):
    """One big divide_rounds call is internally split into sub-batches whose can_see sweeps
    overlap the round loop of the previous sub-batch; the split must not change anything."""
    monkeypatch.setenv("SW_PIPE", pipe)
    for n, N, seed, mode, p0, p1 in [(64, 150000, 90, 0, 0, 0), (200, 90000, 91, 2, 0.1, 0.05), (16, 70000, 92, 3, 0.5, 0)]:
        stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
        o, ncs_o = oracle_run(n, stream)
        h, ncs_h = hip_run(pkg, n, stream)
        assert ncs_h == ncs_o
        assert_state_equal(h, o.round, o.can_see, o.witnesses(), o.famous_by_event, o.consensus())
        assert list(h.find_order(ncs_h[0])) == list(o.find_order(ncs_o[0]))
        h.close()


def test_full_size_properties(pkg):
    """256 members / 1M events (BASELINE.json configs[2]): properties that need no oracle.
    (a) sees-mask self bit; (b) rounds are monotone along every self-parent chain and
    round <= max(parent rounds)+1; (c) witness table = first event of each member per
    round; (d) the 40k-event prefix equals a separate 40k-event run (prefix stability of
    round / can_see / witnesses); (e) re-running gives identical results."""
    n, N = 256, 1_000_000
    stream = pkg.synth_hashgraph(n, N, 3, 0, 0, 0)
    cr, sp, op, t, sig = stream
    h, ncs = hip_run(pkg, n, stream)
    rnd = h.rounds()
    nr = n
    assert (rnd[:nr] == 0).all()
    pr = np.maximum(rnd[sp[nr:]], rnd[op[nr:]])
    assert ((rnd[nr:] == pr) | (rnd[nr:] == pr + 1)).all()
    wit = h.witnesses()
    R = wit.shape[0]
    assert R == h.max_round + 1 and R > 250
    # witness = first event of its creator with that round, and round increased vs self-parent
    for r in (0, 1, R // 2, R - 1):
        for c in range(0, n, 37):
            w = wit[r, c]
            if w >= 0:
                assert cr[w] == c and rnd[w] == r
                assert sp[w] < 0 or rnd[sp[w]] < r
    is_wit = np.zeros(N, bool)
    is_wit[wit[wit >= 0]] = True
    exp_wit = np.ones(N, bool)
    exp_wit[nr:] = rnd[nr:] > rnd[sp[nr:]]
    assert np.array_equal(is_wit, exp_wit)
    masks = h.sees_masks(0, 5000)
    own = (masks[np.arange(5000), cr[:5000] // 64] >> (cr[:5000] % 64).astype(np.uint64)) & np.uint64(1)
    assert (own == 1).all()
    # prefix stability
    M = 40000
    h2, _ = hip_run(pkg, n, tuple(a[:M] for a in stream))
    assert np.array_equal(h2.rounds(), rnd[:M])
    assert np.array_equal(h2.can_see(), h.can_see(0, M))
    R2 = h2.max_round  # rounds below the prefix's last round have the same witnesses
    assert np.array_equal(h2.witnesses(0, R2), wit[:R2])
    fam = h.famous()
    cons = h.consensus()
    assert cons[: R - 12].all(), "all but the last few rounds must be decided"
    assert ((fam >= 0) == (wit >= 0))[: R - 12].all()
    # determinism
    h3, ncs3 = hip_run(pkg, n, stream)
    assert ncs3 == ncs and np.array_equal(h3.rounds(), rnd) and np.array_equal(h3.famous(), fam)

