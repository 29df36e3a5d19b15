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


def elections_closed(n, wit, Sw, stake, coin, famous, consensus, r_closed, coin_period=6):
    """PROTOTYPE (profiles/NOTES_next_round.md, experiment 2): elections that may run BEFORE every event of the call has been
    divided — beside the last round loop — without changing anything decide_fame reports.  Only voters of CLOSED rounds take
    part (rv <= r_closed: every member already has an event of round >= rv, so no witness of a round <= rv can appear later
    and the voters' masks are final), and a round is committed ALL OR NOTHING: decisions, consensus flag and the P2 count of
    a round are kept only if every witness of the round got decided; otherwise the round is left exactly as it was, for the
    real decide_fame.  Returns (rounds committed, P2 of those rounds)."""
    R = wit.shape[0]
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    committed, p2_total = [], 0
    for r in range(0, min(R, r_closed)):
        if consensus[r]:
            continue
        row = famous[r].copy()
        p2 = 0
        any_dec = False
        for cx in range(n):
            x = wit[r, cx]
            if x < 0 or row[cx] >= 0:
                continue
            V = None
            for d in range(1, r_closed - r + 1):
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
                        idx = np.where(sm, wit[rv], INF)
                        first = int(np.argmin(idx))
                        row[cx] = 1 if v[first] else 0
                        p2 += int((voters & (wit[rv] <= wit[rv][first])).sum())
                        any_dec = True
                        break
                    V = v & voters
                else:
                    V = np.where(sm, v, coin[np.maximum(wit[rv], 0)].astype(bool)) & voters
                p2 += int(voters.sum())
        if any_dec and all(row[c] >= 0 for c in range(n) if wit[r, c] >= 0):
            famous[r] = row
            consensus[r] = 1
            committed.append(r)
            p2_total += p2
    return committed, p2_total


def bulk_rounds_v2(n, cr, sp, op, stake, K=4, NEARCAP=None, CAPMAX=None):
    """Round loop with the inheritance shortcut for FAR candidates (kernel structure of
    k_resolve_band / k_tally_* since the slow-member fix):
      * a candidate e is FAR when max(sp[e], op[e]) lies beyond the band; far candidates are
        never tallied;
      * all earlier positions of e's creator being false, round[e] >= r+1 iff the other-parent
        q = op[e] has round >= r+1, i.e. q >= lo[r+1][creator(q)]; that is known once creator(q)
        is resolved for this round, and definitely false when q lies before that creator's
        cursor; otherwise the member waits for the next iteration;
      * a far candidate whose parents both have round <= r needs a real tally: the band is
        doubled (up to CAPMAX) so that it becomes near.
    `band_out` (a dict): FINALIZE FROM THE BAND (k_resolve_band since round 4): every band pass leaves round_band[k] = r and
    S_band[k] = its mask for the band events at or after their creator's round-r witness; the pass of an event's own round is
    the last to write it.  Filled with {"round": int array (-1: never written), "S": bool (N, n)}."""
    N = len(cr)
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    NEARCAP = NEARCAP or 8 * n
    CAPMAX = CAPMAX or max(N, NEARCAP)
    L = can_see_rows(n, cr, sp, op)
    chains = [np.nonzero(cr == c)[0].astype(np.int64) for c in range(n)]
    lo = [np.array([ch[0] if len(ch) else INF for ch in chains], np.int64)]
    pos = np.zeros(n, np.int64)
    stats = dict(iters=0, evals=0, waits=0, grows=0)
    while True:
        lo_r = lo[-1]
        active = lo_r < INF
        if not active.any():
            lo.pop()
            break
        mlo = int(lo_r[active].min())
        ncap = NEARCAP
        lo_next = np.full(n, INF, np.int64)
        pos_next = pos.copy()
        resolved = ~active            # inactive members: lo[r+1] = INF, known
        cur = pos.copy()
        mhi_done = mlo
        masks = {}
        while not resolved.all():
            stats["iters"] += 1
            unres = [c for c in range(n) if not resolved[c]]
            maxc = max(int(chains[c][min(cur[c] + K, len(chains[c])) - 1]) for c in unres
                       if cur[c] < len(chains[c])) if any(cur[c] < len(chains[c]) for c in unres) else mlo
            mhi = min(N, min(maxc + 1, mlo + ncap))
            for k in range(mhi_done, mhi):            # band (extension)
                if k >= lo_r[cr[k]]:
                    masks[k] = L[k] >= lo_r
            mhi_done = max(mhi_done, mhi)
            found = {c: None for c in unres}
            farslot = {c: None for c in unres}
            nslots = {c: 0 for c in unres}
            for c in unres:                            # ---- tally kernel
                for j in range(K):
                    p = cur[c] + j
                    if p >= len(chains[c]):
                        break
                    nslots[c] = j + 1
                    e = int(chains[c][p])
                    reach = max(int(sp[e]), int(op[e]))
                    if reach >= mhi_done:              # far: not tallied
                        if farslot[c] is None:
                            farslot[c] = j
                        continue
                    stats["evals"] += 1
                    P = L[e].copy()
                    P[cr[e]] = sp[e]
                    hits = np.zeros(n, np.int64)
                    for c2 in np.nonzero(P >= lo_r)[0]:
                        hits += stake[c2] * masks[int(P[c2])]
                    if 3 * int(np.count_nonzero(3 * hits > 2 * T)) > 2 * T and found[c] is None:
                        found[c] = j
            # ---- resolve
            newly = {}
            grow = False
            for c in unres:
                f, jf = found[c], farslot[c]
                if f is not None and (jf is None or f < jf):
                    newly[c] = (int(chains[c][cur[c] + f]), cur[c] + f)
                elif jf is None:
                    if cur[c] + K >= len(chains[c]):
                        newly[c] = (INF, None)         # exhausted
                    else:
                        cur[c] += K
                else:
                    cur[c] += jf                       # near slots before jf are false
            for c in unres:                            # far candidates (uses this iteration's results)
                if c in newly or farslot[c] is None or (found[c] is not None and found[c] < farslot[c]):
                    continue
                e = int(chains[c][cur[c]])
                q = int(op[e])
                b = int(cr[q])
                if resolved[b] or b in newly:
                    ln = lo_next[b] if resolved[b] else newly[b][0]
                    if q >= ln:
                        newly[c] = (e, cur[c])
                    else:
                        grow = True
                else:
                    lb = int(chains[b][cur[b]]) if cur[b] < len(chains[b]) else INF
                    if q < lb:
                        grow = True
                    else:
                        stats["waits"] += 1
            for c, (ev, p) in newly.items():
                resolved[c] = True
                if ev != INF:
                    lo_next[c] = ev
                    pos_next[c] = p
            if grow:
                assert ncap < CAPMAX, "model: band cap exhausted"
                ncap = min(2 * ncap, CAPMAX)
                stats["grows"] += 1
        lo.append(lo_next)
        pos = pos_next
    return L, np.array(lo, np.int64).astype(np.int32), stats


def bulk_rounds_v3(n, cr, sp, op, stake, K=4, NEARCAP=None, CAPMAX=None, gallop_after=2, skip=0, band_out=None):
    """bulk_rounds_v2 plus GALLOPING for long runs of false candidates (a few members creating
    most of the events: thousands of chain positions per round, K per iteration).  After
    `gallop_after` consecutive windows without a passing candidate a member's window becomes
    STRIDED (positions cur, cur+K, cur+2K, ...): the predicate is monotone along the chain, so a
    failing slot rules out everything before it, and a passing slot f > 0 brackets the first
    passing position in (slot f-1, slot f], which one ordinary window covers.  A strided window
    falls back to an ordinary one whenever it meets a far candidate or the end of the chain.
    `skip` > 0 additionally starts the window of a fresh round `skip` positions after the cursor
    (the first positions after a witness practically never pass): a passing slot 0 then brackets
    the first passing position in [cursor, cursor + skip] and the member looks again from the
    cursor; so does a far slot 0.
    Prototype of the next kernel change; the rest is bulk_rounds_v2:
    Round loop with the inheritance shortcut for FAR candidates (kernel structure of
    k_resolve_band / k_tally_* since the slow-member fix):
      * a candidate e is FAR when max(sp[e], op[e]) lies beyond the band; far candidates are
        never tallied;
      * all earlier positions of e's creator being false, round[e] >= r+1 iff the other-parent
        q = op[e] has round >= r+1, i.e. q >= lo[r+1][creator(q)]; that is known once creator(q)
        is resolved for this round, and definitely false when q lies before that creator's
        cursor; otherwise the member waits for the next iteration;
      * a far candidate whose parents both have round <= r needs a real tally: the band is
        doubled (up to CAPMAX) so that it becomes near.
    `band_out` (a dict): FINALIZE FROM THE BAND (k_resolve_band since round 4): every band pass leaves round_band[k] = r and
    S_band[k] = its mask for the band events at or after their creator's round-r witness; the pass of an event's own round is
    the last to write it.  Filled with {"round": int array (-1: never written), "S": bool (N, n)}."""
    N = len(cr)
    stake = np.asarray(stake, np.int64)
    T = int(stake.sum())
    NEARCAP = NEARCAP or 8 * n
    CAPMAX = CAPMAX or max(N, NEARCAP)
    L = can_see_rows(n, cr, sp, op)
    chains = [np.nonzero(cr == c)[0].astype(np.int64) for c in range(n)]
    lo = [np.array([ch[0] if len(ch) else INF for ch in chains], np.int64)]
    pos = np.zeros(n, np.int64)
    stats = dict(iters=0, evals=0, waits=0, grows=0, strided=0, refines=0)
    if band_out is not None:
        band_out["round"] = np.full(N, -1, np.int32)
        band_out["S"] = np.zeros((N, n), bool)
    while True:
        lo_r = lo[-1]
        active = lo_r < INF
        if not active.any():
            lo.pop()
            break
        mlo = int(lo_r[active].min())
        ncap = NEARCAP
        lo_next = np.full(n, INF, np.int64)
        pos_next = pos.copy()
        resolved = ~active            # inactive members: lo[r+1] = INF, known
        cur = pos.copy()
        stride = np.ones(n, np.int64)
        misses = np.zeros(n, np.int64)
        skipped = np.zeros(n, np.int64)   # positions [cur - skipped, cur) have not been looked at
        if skip:
            for c in range(n):
                if active[c] and cur[c] + skip < len(chains[c]):
                    cur[c] += skip
                    skipped[c] = skip
        mhi_done = mlo
        masks = {}
        while not resolved.all():
            stats["iters"] += 1
            unres = [c for c in range(n) if not resolved[c]]
            def last_slot(c):  # chain position of the last slot of c's window
                return min(cur[c] + (K - 1) * stride[c], len(chains[c]) - 1)
            maxc = max(int(chains[c][last_slot(c)]) for c in unres
                       if cur[c] < len(chains[c])) if any(cur[c] < len(chains[c]) for c in unres) else mlo
            mhi = min(N, min(maxc + 1, mlo + ncap))
            for k in range(mhi_done, mhi):            # band (extension)
                if k >= lo_r[cr[k]]:
                    masks[k] = L[k] >= lo_r
                    if band_out is not None:
                        band_out["round"][k] = len(lo) - 1
                        band_out["S"][k] = masks[k]
            mhi_done = max(mhi_done, mhi)
            found = {c: None for c in unres}
            farslot = {c: None for c in unres}
            nslots = {c: 0 for c in unres}
            for c in unres:                            # ---- tally kernel
                if stride[c] > 1:
                    stats["strided"] += 1
                for j in range(K):
                    p = cur[c] + j * stride[c]
                    if p >= len(chains[c]):
                        break
                    nslots[c] = j + 1
                    e = int(chains[c][p])
                    reach = max(int(sp[e]), int(op[e]))
                    if reach >= mhi_done:              # far: not tallied
                        if farslot[c] is None:
                            farslot[c] = j
                        continue
                    stats["evals"] += 1
                    P = L[e].copy()
                    P[cr[e]] = sp[e]
                    hits = np.zeros(n, np.int64)
                    for c2 in np.nonzero(P >= lo_r)[0]:
                        hits += stake[c2] * masks[int(P[c2])]
                    if 3 * int(np.count_nonzero(3 * hits > 2 * T)) > 2 * T and found[c] is None:
                        found[c] = j
            # ---- resolve
            newly = {}
            grow = False
            refined = set()
            for c in unres:
                f, jf = found[c], farslot[c]
                s_ = int(stride[c])
                if skipped[c]:
                    if (f == 0 and (jf is None or jf > 0)) or jf == 0:
                        cur[c] -= skipped[c]           # slot 0 passed or is far: look again from the cursor
                        skipped[c] = 0
                        refined.add(c)
                        stats["refines"] += 1
                        continue
                    skipped[c] = 0                     # slot 0 failed: everything before it fails too
                if f is not None and (jf is None or f < jf):
                    if s_ == 1 or f == 0:
                        newly[c] = (int(chains[c][cur[c] + f * s_]), cur[c] + f * s_)
                    else:                              # bracketed: (slot f-1, slot f]
                        cur[c] += (f - 1) * s_ + 1
                        stride[c] = 1
                        refined.add(c)
                        stats["refines"] += 1
                elif jf is None:
                    last = cur[c] + (nslots[c] - 1) * s_   # last evaluated position: false
                    if s_ == 1 and cur[c] + K >= len(chains[c]):
                        newly[c] = (INF, None)         # exhausted
                    elif s_ > 1 and cur[c] + K * s_ >= len(chains[c]):
                        cur[c] = last + 1              # the tail of the chain: ordinary windows
                        stride[c] = 1
                        refined.add(c)
                    else:
                        cur[c] = last + 1
                        misses[c] += 1
                        if misses[c] >= gallop_after:
                            stride[c] = K
                elif s_ > 1 and jf > 0:                # a far slot inside a strided window
                    cur[c] += (jf - 1) * s_ + 1
                    stride[c] = 1
                    refined.add(c)
                else:
                    cur[c] += jf * s_                  # near slots before jf are false (jf == 0 if strided)
                    stride[c] = 1
            for c in unres:                            # far candidates (uses this iteration's results)
                if c in newly or c in refined or farslot[c] is None or (found[c] is not None and found[c] < farslot[c]):
                    continue
                e = int(chains[c][cur[c]])
                q = int(op[e])
                b = int(cr[q])
                if resolved[b] or b in newly:
                    ln = lo_next[b] if resolved[b] else newly[b][0]
                    if q >= ln:
                        newly[c] = (e, cur[c])
                    else:
                        grow = True
                else:
                    lb = int(chains[b][cur[b]]) if cur[b] < len(chains[b]) else INF
                    if q < lb:
                        grow = True
                    else:
                        stats["waits"] += 1
            for c, (ev, p) in newly.items():
                resolved[c] = True
                if ev != INF:
                    lo_next[c] = ev
                    pos_next[c] = p
            if grow:
                assert ncap < CAPMAX, "model: band cap exhausted"
                ncap = min(2 * ncap, CAPMAX)
                stats["grows"] += 1
        lo.append(lo_next)
        pos = pos_next
    return L, np.array(lo, np.int64).astype(np.int32), stats
