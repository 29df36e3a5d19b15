"""Executable specification (numpy, CPU) of the round-synchronous bulk algorithm the
HIP kernels implement — TEST INFRASTRUCTURE.

It mirrors the kernel structure one to one (level-ordered can_see rows; per-round
loop with per-member candidate cursors, band mask table + on-the-fly far hops;
finalize; candidate-major elections) so that the *reformulation itself* can be checked
against the sequential oracle on the CPU, independent of any GPU.  The product never
imports this file.

Reference semantics: swirld.py:187-277; index form and the validity argument of the
round-synchronous form: SURVEY.md Appendix A.
"""
import numpy as np

INF = np.int32(0x7FFFFFFF)


def can_see_rows(n, cr, sp, op):
    """L[e][c] = latest event of member c among the ancestors-or-self of e (fork-free:
    'higher' by height == larger index on one creator's chain, swirld.py:170-184)."""
    N = len(cr)
    L = np.full((N, n), -1, np.int32)
    for e in range(N):
        if sp[e] >= 0:
            np.maximum(L[sp[e]], L[op[e]], out=L[e])
        L[e, cr[e]] = e
    return L


def tally(L, e, cr, sp, lo_r, stake, T, masks, mlo, mhi, own_is_self):
    """Monotone strongly-sees predicate SS_r(e) (Appendix A, round-synchronous form)."""
    P = L[e].copy()
    if not own_is_self:
        P[cr[e]] = sp[e]  # row BEFORE the self overwrite (swirld.py:204 vs 220, Q4)
    valid = (P >= lo_r)  # hop has round >= r   (P == -1 fails since lo_r >= 0)
    hits = np.zeros(len(lo_r), np.int64)
    for c in np.nonzero(valid)[0]:
        k = P[c]
        m = masks[k - mlo] if mlo <= k < mhi else (L[k] >= lo_r)
        hits += stake[c] * m
    cnt = int(np.count_nonzero(3 * hits > 2 * T))
    return 3 * cnt > 2 * T


def bulk_rounds(n, cr, sp, op, stake, K=4, MCAP=None):
    """Returns (L, lo) with lo[r][c] = first event of c whose round >= r (INF if none)."""
    N = len(cr)
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    MCAP = MCAP or 8 * n
    L = can_see_rows(n, cr, sp, op)
    chains = [np.nonzero(cr == c)[0].astype(np.int32) for c in range(n)]
    lo = [np.array([ch[0] if len(ch) else INF for ch in chains], np.int32)]
    pos = np.zeros(n, np.int64)
    stats = dict(iters=0, evals=0, far=0)
    while True:
        lo_r = lo[-1]
        active = lo_r < INF
        if not active.any():
            lo.pop()  # the all-INF row is not a round
            break
        mlo = int(lo_r[active].min())
        mhi = min(N, mlo + MCAP)
        masks = np.zeros((mhi - mlo, n), bool)
        for k in range(mlo, mhi):
            if k >= lo_r[cr[k]]:
                masks[k - mlo] = L[k] >= lo_r
        lo_next = np.full(n, INF, np.int32)
        pos_next = pos.copy()
        unresolved = [c for c in range(n) if active[c]]
        cur = pos.copy()
        while unresolved:
            stats["iters"] += 1
            still = []
            for c in unresolved:
                found = False
                for j in range(K):
                    p = cur[c] + j
                    if p >= len(chains[c]):
                        break
                    e = int(chains[c][p])
                    stats["evals"] += 1
                    if tally(L, e, cr, sp, lo_r, stake, T, masks, mlo, mhi, False):
                        lo_next[c] = e
                        pos_next[c] = p
                        found = True
                        break
                if not found:
                    if cur[c] + K >= len(chains[c]):
                        pass  # chain exhausted: no event of c reaches round r+1 (yet)
                    else:
                        cur[c] += K
                        still.append(c)
            unresolved = still
        lo.append(lo_next)
        pos = pos_next
    return L, np.array(lo, np.int32), stats


def finalize(n, cr, L, lo):
    """round[e], sees-mask S[e] (bool n) and witness table from the lo table."""
    N = len(cr)
    R = lo.shape[0]
    rnd = np.zeros(N, np.int32)
    for e in range(N):
        col = lo[:, cr[e]]
        rnd[e] = np.searchsorted(col, e, side="right") - 1  # max r with lo[r][c] <= e
    S = np.zeros((N, n), bool)
    for e in range(N):
        S[e] = L[e] >= lo[rnd[e]]
    wit = np.full((R, n), -1, np.int32)
    for r in range(R):
        nxt = lo[r + 1] if r + 1 < R else np.full(n, INF, np.int32)
        ok = (lo[r] < INF) & (nxt > lo[r])
        wit[r, ok] = lo[r, ok]
    return rnd, S, wit


def voter_masks(n, L, rnd, S, wit, stake):
    """Sw[r][c] = members whose round-(r-1) witness the witness (r, c) strongly sees
    (swirld.py:247-254; final row incl. self, Q4/Q6)."""
    R = wit.shape[0]
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    Sw = np.zeros((R, n, n), bool)
    for r in range(1, R):
        for c in range(n):
            y = wit[r, c]
            if y < 0:
                continue
            hits = np.zeros(n, np.int64)
            for c2 in range(n):
                k = L[y, c2]
                if k >= 0 and rnd[k] == r - 1:
                    hits += stake[c2] * S[k]
            Sw[r, c] = (3 * hits > 2 * T) & (wit[r - 1] >= 0)
    return Sw


def elections(n, wit, Sw, stake, coin, famous, consensus, coin_period=6):
    """Candidate-major decide_fame (swirld.py:224-277).  famous: int8 [R][n] table
    (-1 undecided), consensus: uint8 [R]; both updated in place.  Returns sorted new_c."""
    R = wit.shape[0]
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    max_c = 0
    while max_c < R and consensus[max_c]:
        max_c += 1
    done = set()
    p2 = 0
    for r in range(max_c, R):
        if consensus[r]:
            continue
        for cx in range(n):
            x = wit[r, cx]
            if x < 0 or famous[r, cx] >= 0:
                continue
            V = None
            for d in range(1, R - r):
                rv = r + d
                voters = wit[rv] >= 0
                if d == 1:
                    V = Sw[rv][:, cx] & voters
                    continue
                yes = (Sw[rv] & V[None, :]) @ stake
                tot = Sw[rv] @ stake
                no = tot - yes
                v = ~(no > yes)
                t = np.where(v, yes, no)
                sm = (3 * t > 2 * T) & voters
                if d % coin_period != 0:
                    if sm.any():
                        # first decider in dict order == smallest event index; voters
                        # after it never evaluate x (swirld.py:235)
                        idx = np.where(sm, wit[rv], INF)
                        first = int(np.argmin(idx))
                        famous[r, cx] = 1 if v[first] else 0
                        p2 += int((voters & (wit[rv] <= wit[rv][first])).sum())
                        done.add(r)
                        break
                    V = v & voters
                else:
                    V = np.where(sm, v, coin[np.maximum(wit[rv], 0)].astype(bool)) & voters
                p2 += int(voters.sum())
    new_c = sorted(r for r in done if all(famous[r, c] >= 0 for c in range(n) if wit[r, c] >= 0))
    for r in new_c:
        consensus[r] = 1
    return new_c, p2
