"""CPU check of the two bounds the one-wave-per-slot tally takes from the POPCOUNTS of the hop masks before it gathers the
masks themselves (k_tally_bits<., FILT>, round 5).  For a candidate e of round r with V valid hops, threshold
t = floor(2 T / 3) (unit stakes, T = n) and S = sum over the valid hops k of popcount{c_ : can_see[k][c_] >= lo[r][c_]}
(= the sum over the columns of hits[c_], swirld.py:208-216):
    S <  (t + 1)^2                 =>  the tally FAILS  (a pass needs t + 1 columns with more than t hits each)
    S >  t V + (n - t) t           =>  the tally PASSES (a failure has at most t columns above t, each at most V)
Every (round, member, chain position) of seeded hashgraphs is evaluated exactly, from the oracle's rounds and can_see table, and
compared with both bounds; the test also reports how many slots the bounds leave undecided."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import Oracle  # noqa: E402
from synth_util import synth  # noqa: E402

INF = 1 << 30


def bounds_vs_exact(n, N, seed, mode=0, p0=0.0, p1=0.0, positions=40):
    cr, sp, op, t, sig = synth(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    o.append_events(cr, sp, op, t, sig)
    o.divide_rounds(0, N)
    rnd = np.array(o.round)
    L = np.array(o.can_see).reshape(N, -1)[:, :n]
    R = int(rnd.max()) + 1
    chains = [np.nonzero(cr == c)[0] for c in range(n)]
    lo = np.full((R + 1, n), INF, dtype=np.int64)
    for c in range(n):
        rr = rnd[chains[c]]
        for r in range(R):
            k = int(np.searchsorted(rr, r))
            if k < len(chains[c]):
                lo[r, c] = chains[c][k]
    thr3 = (2 * n) // 3
    checked = undecided = 0
    for r in range(R - 1):
        thr = lo[r]
        for c in range(n):
            if thr[c] == INF:
                continue
            start = int(np.searchsorted(chains[c], thr[c]))
            for j in range(positions):
                if start + j >= len(chains[c]):
                    break
                e = chains[c][start + j]
                P = L[e].copy()
                P[c] = sp[e]          # the row BEFORE the self overwrite (swirld.py:203-205 vs 220)
                valid = P >= thr      # round[hop] == r  <=>  hop >= lo[r][creator] for an ancestor of a round-r event's successor
                V = int(valid.sum())
                hits = (L[P[valid]] >= thr[None, :]).sum(0) if V else np.zeros(n, dtype=np.int64)
                S = int(hits.sum())
                exact = int((hits > thr3).sum()) > thr3
                # the exact predicate IS the promotion test: round[e] >= r + 1
                assert exact == (rnd[e] >= r + 1), (n, seed, r, c, j)
                fails_for_sure = V <= thr3 or S < (thr3 + 1) ** 2
                passes_for_sure = V > thr3 and S > thr3 * V + (n - thr3) * thr3
                assert not (fails_for_sure and exact), (n, seed, r, c, j, V, S)
                assert not (passes_for_sure and not exact), (n, seed, r, c, j, V, S)
                checked += 1
                undecided += int(not fails_for_sure and not passes_for_sure)
                if exact and passes_for_sure:
                    break             # (monotone along the chain: everything later passes as well)
    return checked, undecided


@pytest.mark.parametrize("n,N,seed,mode,p0,p1", [(4, 600, 1, 0, 0, 0), (7, 1500, 2, 0, 0, 0), (16, 4000, 3, 0, 0, 0), (33, 6000, 4, 0, 0, 0),
                                                 (64, 12000, 5, 0, 0, 0), (20, 5000, 6, 1, 0.05, 0), (24, 6000, 7, 2, 0.3, 0.02),
                                                 (30, 6000, 8, 3, 0.7, 0), (130, 16000, 9, 0, 0, 0)])
def test_popcount_bounds_never_contradict_the_exact_tally(n, N, seed, mode, p0, p1):
    checked, undecided = bounds_vs_exact(n, N, seed, mode, p0, p1)
    assert checked > 100
    # the bounds are worth having only if they decide most slots
    assert undecided < checked
