"""CPU: the round-synchronous / candidate-major reformulation the HIP kernels implement
(tests/model_bulk.py, numpy) against the sequential oracle, on fork-free goldens and on
fresh seeded streams.  Validates the algorithm independent of any GPU."""
import numpy as np
import pytest

import model_bulk as mb
from conftest import golden_names, load_golden
from oracle.oracle import Oracle

BATCH = [n for n in golden_names() if n.endswith("batch") or n.endswith("cliques")
         or n.endswith("slow") or n.endswith("stale") or n.endswith("stake")]
BATCH = [n for n in BATCH if not n.startswith("n130") and not n.startswith("n70")]


def check(n, cr, sp, op, sig, stake, exp, K, MCAP):
    L, lo, _ = mb.bulk_rounds(n, cr, sp, op, stake, K=K, MCAP=MCAP)
    rnd, S, wit = mb.finalize(n, cr, L, lo)
    assert np.array_equal(L, exp["can_see"])
    assert np.array_equal(rnd, exp["round"])
    assert np.array_equal(wit, exp["witnesses"])
    Sw = mb.voter_masks(n, L, rnd, S, wit, stake)
    fam = np.full(wit.shape, -1, np.int8)
    cons = np.zeros(wit.shape[0], np.uint8)
    new_c, p2 = mb.elections(n, wit, Sw, stake, sig[:, 0] >= 128, fam, cons)
    m = wit >= 0
    assert np.array_equal(fam[m], exp["famous"][wit[m]])
    assert np.array_equal(cons, exp["consensus"])
    return new_c, p2


@pytest.mark.parametrize("name", BATCH)
def test_model_matches_reference_golden(name):
    g = load_golden(name)
    assert len(g["batches"]) == 1
    new_c, _ = check(g["n"], g["creator"], g["self_parent"], g["other_parent"], g["sig"],
                     g["stake"].astype(np.int64), g, K=3, MCAP=4 * g["n"])
    assert list(new_c) == list(g["new_c_flat"])


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,K,MCAP", [
    (4, 1500, 21, 0, 0, 0, 1, 8),
    (9, 1500, 22, 2, 0.3, 0.02, 2, 20),
    (20, 2500, 23, 3, 0.6, 0, 4, 50),
    (24, 2500, 24, 1, 0.02, 0, 8, 400),
])
def test_model_matches_oracle(pkg, n, N, seed, mode, p0, p1, K, MCAP):
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    nc = o.decide_fame()
    exp = dict(can_see=o.can_see, round=o.round, witnesses=o.witnesses(),
               famous=o.famous_by_event, consensus=o.consensus())
    new_c, p2 = check(n, cr, sp, op, sig, np.ones(n, np.int64), exp, K, MCAP)
    assert list(new_c) == list(nc)
    assert p2 == o.counters()["majority_evals"]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,K,cap", [
    (4, 600, 1, 0, 0, 0, 2, 8), (16, 3000, 3, 2, 0.25, 0.01, 4, 32), (16, 3000, 5, 1, 0.01, 0, 4, 24),
    (24, 4000, 7, 2, 0.2, 0.002, 8, 64), (12, 3000, 9, 1, 0.002, 0, 4, 16),
])
def test_far_candidate_inheritance_model(pkg, n, N, seed, mode, p0, p1, K, cap):
    """The far-candidate rule of the round loop (a candidate with a parent beyond the band is
    decided by inheritance from its other-parent, waits, or doubles the band) vs the oracle."""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    L, lo, stats = mb.bulk_rounds_v2(n, cr, sp, op, np.ones(n, np.int64), K=K, NEARCAP=cap)
    rnd, S, wit = mb.finalize(n, cr, L, lo)
    assert np.array_equal(rnd, o.round)
    assert np.array_equal(wit, o.witnesses())
    assert stats["waits"] > 0 and stats["grows"] > 0


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,K,cap", [
    (16, 6000, 31, 2, 0.2, 50.0, 4, 64),     # hot members: 3 of 16 create ~90 % of the events
    (20, 8000, 34, 2, 0.15, 100.0, 8, 128),
    (12, 4000, 33, 0, 0, 0, 3, 24),          # uniform gossip with a tiny window: galloping misfires
    (16, 3000, 3, 2, 0.25, 0.01, 4, 32),     # slow members: far candidates inside strided windows
])
def test_galloping_window_model(pkg, n, N, seed, mode, p0, p1, K, cap):
    """Next kernel change, validated here first: after two windows without a
    passing candidate a member's window becomes strided; exactness vs the oracle, and the
    iteration count on hot-member DAGs (the weak spot of the current round loop)."""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    one = np.ones(n, np.int64)
    L3, lo3, st3 = mb.bulk_rounds_v3(n, cr, sp, op, one, K=K, NEARCAP=cap)
    rnd, S, wit = mb.finalize(n, cr, L3, lo3)
    assert np.array_equal(rnd, o.round)
    assert np.array_equal(wit, o.witnesses())
    _, lo2, st2 = mb.bulk_rounds_v2(n, cr, sp, op, one, K=K, NEARCAP=cap)
    assert np.array_equal(lo2, lo3)
    assert st3["strided"] > 0 and st3["refines"] > 0
    if p1 > 1:  # hot members: far fewer iterations and tallies
        assert 2 * st3["iters"] < st2["iters"] and 2 * st3["evals"] < st2["evals"]
    else:       # elsewhere galloping must not cost more than a few percent
        assert st3["iters"] <= st2["iters"] * 1.1 + 2


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,K,cap", [
    (12, 4000, 33, 0, 0, 0, 6, 48), (24, 5000, 40, 0, 0, 0, 10, 200), (20, 5000, 41, 3, 0.6, 0, 8, 100),
    (24, 4000, 7, 2, 0.2, 0.002, 8, 64), (16, 6000, 31, 2, 0.2, 50.0, 4, 64),
])
@pytest.mark.parametrize("skip", [1, 2, 3])
def test_window_offset_model(pkg, n, N, seed, mode, p0, p1, K, cap, skip):
    """Next kernel change, validated here first: the window of a fresh round
    starts `skip` positions after the cursor (at 256 members the first passing position is
    13.8 +- 3.7 after it, never before position 1), a passing or far slot 0 sends the member back
    to the cursor.  Exactness vs the oracle, with far candidates, waits and galloping in play."""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    L3, lo3, st3 = mb.bulk_rounds_v3(n, cr, sp, op, np.ones(n, np.int64), K=K, NEARCAP=cap, skip=skip)
    rnd, S, wit = mb.finalize(n, cr, L3, lo3)
    assert np.array_equal(rnd, o.round)
    assert np.array_equal(wit, o.witnesses())


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,K,cap", [
    (12, 4000, 33, 0, 0, 0, 6, None), (24, 5000, 40, 0, 0, 0, 10, None), (20, 5000, 41, 3, 0.6, 0, 8, None),
    (24, 4000, 7, 2, 0.2, 0.002, 8, None), (16, 6000, 31, 2, 0.2, 50.0, 4, None), (9, 3000, 5, 1, 0.05, 0, 5, None),
    (12, 4000, 33, 0, 0, 0, 6, 48), (24, 4000, 7, 2, 0.2, 0.002, 8, 64), (20, 5000, 41, 3, 0.6, 0, 8, 100),
])
def test_round_numbers_and_sees_masks_from_the_band_pass_model(pkg, n, N, seed, mode, p0, p1, K, cap):
    """The rule k_resolve_band applies since round 4: every band pass writes round = r and the pass's mask for
    the band events at or after their creator's round-r witness; rounds only go up, so what is left behind is the event's own
    round and sees-mask wherever a band of that round covered the event.  (1) Where the band's round equals the searched round
    — what k_finalize_check tests — the band's mask IS the sees-mask computed from the row; (2) with an unbounded band every
    event is covered; with a capped band the uncovered ones are exactly those the check sends back to their rows."""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    band = {}
    L3, lo3, st3 = mb.bulk_rounds_v3(n, cr, sp, op, np.ones(n, np.int64), K=K, NEARCAP=cap or N, skip=1, band_out=band)
    rnd, S, wit = mb.finalize(n, cr, L3, lo3)
    covered = band["round"] == rnd
    assert np.array_equal(band["S"][covered], S[covered])
    assert (band["round"] <= rnd).all()          # a pass of a later round never writes an event of an earlier one
    if cap is None:
        assert covered.all()
    else:
        assert covered.mean() > 0.5              # (the capped band still covers most events; the rest go back to their rows)


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,parts", [
    (9, 2400, 61, 0, 0, 0, 4), (12, 3000, 62, 2, 0.3, 0.02, 5), (16, 3200, 63, 1, 0.03, 0, 4), (20, 3000, 64, 3, 0.6, 0, 3),
    (10, 3000, 65, 2, 0.5, 0.01, 6),
])
def test_elections_of_closed_rounds_before_the_last_events_model(pkg, n, N, seed, mode, p0, p1, parts):
    """Prototype of an overlap the device does not have yet (mb.elections_closed): after every prefix of the call's events the
    rounds every member has left are CLOSED — their witness rows and voter masks equal the final ones — and their elections,
    committed all or nothing per round, leave the real decide_fame nothing to redo: decisions, consensus rounds and the P2
    counter of the ONE decide_fame call the reference makes come out identical."""
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    stake = np.ones(n, np.int64)
    coin = sig[:, 0] >= 128
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    nc = o.decide_fame()
    L, lo, _ = mb.bulk_rounds_v3(n, cr, sp, op, stake, K=4, skip=1)
    rnd_f, S_f, wit_f = mb.finalize(n, cr, L, lo)
    Sw_f = mb.voter_masks(n, L, rnd_f, S_f, wit_f, stake)
    R = wit_f.shape[0]
    fam = np.full((R, n), -1, np.int8)
    cons = np.zeros(R, np.uint8)
    pre_rounds, p2_total, closed_seen = [], 0, 0
    for k in range(1, parts):
        cut = N * k // parts
        Lp, lop, _ = mb.bulk_rounds_v3(n, cr[:cut], sp[:cut], op[:cut], stake, K=4, skip=1)
        rnd, S, wit = mb.finalize(n, cr[:cut], Lp, lop)
        front = np.full(n, -1, np.int64)
        np.maximum.at(front, cr[:cut], rnd)
        r_closed = int(front.min())
        if r_closed < 1:
            continue
        Sw = mb.voter_masks(n, Lp, rnd, S, wit, stake)
        # closure: what the prefix knows of the rounds <= r_closed is what the whole call knows
        assert np.array_equal(wit[:r_closed + 1], wit_f[:r_closed + 1])
        assert np.array_equal(Sw[1:r_closed + 1], Sw_f[1:r_closed + 1])
        closed_seen = max(closed_seen, r_closed)
        Rp = wit.shape[0]
        done, p2 = mb.elections_closed(n, wit, Sw, stake, coin, fam[:Rp], cons[:Rp], r_closed)
        pre_rounds += done
        p2_total += p2
    # (the prototype did decide rounds early — except where half of the members are 100 x slower: the rounds they have all left
    # are few, and deciding them takes more levels than are closed)
    assert closed_seen >= 2 and (len(pre_rounds) >= 1 or p0 >= 0.5)
    new_c, p2 = mb.elections(n, wit_f, Sw_f, stake, coin, fam, cons)
    p2_total += p2
    m = wit_f >= 0
    assert np.array_equal(fam[m], o.famous_by_event[wit_f[m]])
    assert np.array_equal(cons, o.consensus())
    assert sorted(pre_rounds + list(new_c)) == list(nc)
    assert p2_total == o.counters()["majority_evals"]


def test_sixteen_bit_band_rows_model():
    """Prototype for 1024 members (profiles/NOTES_next_round.md, experiment 3; at 256 members the band phase is bound by
    requests and the table lost): band rows kept as 16-bit distances L16[k][c] = min(k - L[k][c], 65535).  The band's
    compare L[k][c] >= lo[r][c] becomes L16 <= k - lo[r][c] — exact, saturated entries included, as long as the band is
    shorter than 65535 events (thresholds are at or after its first event); longer bands take the 32-bit rows."""
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(2, 40))
        mlo = int(rng.integers(0, 3_000_000))
        blen = int(rng.integers(1, 65535))
        ks = mlo + rng.integers(0, blen, size=64)
        # thresholds: inside the band, at its start, or INF (no round-r witness)
        thr = mlo + rng.integers(0, blen, size=n).astype(np.int64)
        thr[rng.random(n) < 0.2] = int(mb.INF)
        thr[rng.random(n) < 0.1] = mlo
        for k in ks:
            k = int(k)
            # entries: recent, far in the past (saturating), never seen (-1), the event itself
            L = k - rng.integers(0, 200_000, size=n).astype(np.int64)
            L[rng.random(n) < 0.1] = -1
            L[rng.random(n) < 0.05] = k
            L = np.clip(L, -1, k)
            L16 = np.minimum(k - L, 65535).astype(np.uint16)
            want = L >= thr
            d = k - thr                                   # negative: the threshold lies after the event
            got = (d >= 0) & (L16.astype(np.int64) <= d)
            assert np.array_equal(got, want), (trial, k)
