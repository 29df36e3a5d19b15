"""Executable specification (numpy, CPU) of the CHUNK-PARALLEL can_see sweep — TEST INFRASTRUCTURE.

The reference fills can_see event by event (swirld.py:198-205, 220): a chain of dependent steps as
deep as the DAG.  The chunked sweep (k_cansee_chunks) cuts a range of events
[a_0, a_G) into G chunks that are swept CONCURRENTLY, each from `halo` events before its own
start, and repairs afterwards what a chunk could not know.  This file states that algorithm on the
CPU so that its exactness can be checked against the sequential rows without a GPU; the same
statement is what a multi-GPU partition of the table by event ranges runs (one chunk per rank,
tests/test_partition_chunks_gloo.py).  The product never imports this file.

Zones of a parent x of an event of chunk k (window start w_k = max(a_0, a_k - halo)):
  (i)   x <  a_0   rows of earlier calls: final in memory, read as they are;
  (ii)  a_0 <= x < w_k   rows another chunk is computing right now: UNKNOWN.  x is treated as a
        LEAF: the row {creator(x): x}, everything else absent;
  (iii) x >= w_k  inside the window: computed by this chunk (halo rows are recomputed, not stored).
A value v = V[e][c] of the local sweep is FINAL if v >= w_k: an ancestor by c inside the window
exists, every path to an in-window ancestor stays inside the window (ancestors have smaller
indices), so the local maximum is the true one.  It is also final if v == F_c, the last event of c
before w_k: no larger value below w_k exists (what settles the columns of members silent for longer
than the halo as soon as their newest event was reached as a leaf).  Otherwise it is PROVISIONAL and
    T[e][c] = max(V[e][c], max over members m with a zone-(ii) entry of T[E_m(e)][c]),
    E_m(e) = F_m = last event of m before w_k   if V[e][m] >= w_k  (e reaches m's chain inside the
                                                  window, hence F_m through self-parents)
           = V[e][m]                             if a_0 <= V[e][m] < w_k  (the latest leaf of m)
where every E_m(e) lies in an earlier chunk, whose rows are final once the chunks are repaired in
ascending order.  Fork-free DAGs only (one self-parent chain per member), as the whole fast path.
"""
import numpy as np


def cansee_sequential(n, cr, sp, op):
    N = len(cr)
    L = np.full((N, n), -1, np.int32)
    for e in range(N):
        if sp[e] >= 0:
            np.maximum(L[sp[e]], L[op[e]], out=L[e])
        L[e, cr[e]] = e
    return L


def _leaf(n, x, cr):
    row = np.full(n, -1, np.int32)
    row[cr[x]] = x
    return row


def local_sweep(n, cr, sp, op, L, a0, w, a, b):
    """Rows of the window [w, b) with zone-(ii) parents as leaves; rows of [a, b) are stored into L
    (halo rows [w, a) live in a scratch table).  Returns the number of provisional entries stored."""
    halo = {}
    F = frontier(n, cr, w) if w > a0 else None

    def row_of(x):
        if x < a0:
            return L[x]            # zone (i)
        if x < w:
            return _leaf(n, x, cr)  # zone (ii)
        return halo[x] if x < a else L[x]

    prov = 0
    for e in range(w, b):
        if sp[e] >= 0:
            r = np.maximum(row_of(sp[e]), row_of(op[e]))
        else:
            r = np.full(n, -1, np.int32)
        r[cr[e]] = e
        if e < a:
            halo[e] = r
        else:
            L[e] = r
            if w > a0:
                prov += int(np.count_nonzero((r < w) & (r != F)))
    return prov


def frontier(n, cr, w):
    """F_m = last event of member m before w (-1: none)."""
    F = np.full(n, -1, np.int64)
    for e in range(w):
        F[cr[e]] = e
    return F


def fixup_needs(n, cr, L, a0, w, a, b, F=None):
    """The events whose final rows the repair of the stored rows [a, b) reads (all below w)."""
    if w <= a0:
        return []
    F = frontier(n, cr, w) if F is None else F
    need = set()
    for e in range(a, b):
        V = L[e]
        if ((V < w) & (V != F)).any():
            E = np.where(V >= w, F, V.astype(np.int64))
            need.update(int(x) for x in E[E >= a0])
    return sorted(need)


def fixup(n, cr, L, a0, w, a, b, rows=None):
    """Repairs the provisional entries of the stored rows [a, b); rows below w must be final.  `rows(E)`
    returns the final rows of the events E (default: this table — a rank of a partitioned table passes
    the rows it fetched from the owners)."""
    if w <= a0:
        return 0
    if rows is None:
        rows = lambda E: L[E]
    F = frontier(n, cr, w)
    fixed = 0
    for e in range(a, b):
        V = L[e]
        pcols = np.nonzero((V < w) & (V != F))[0]
        if not len(pcols):
            continue
        E = np.where(V >= w, F, V.astype(np.int64))
        E = E[E >= a0]
        if len(E):
            T = rows(E)[:, pcols].max(axis=0)
            new = np.maximum(V[pcols], T)
            fixed += int(np.count_nonzero(new != V[pcols]))
            L[e, pcols] = new
    return fixed


def cansee_chunked(n, cr, sp, op, a0, cuts, halo, L=None, resweep_limit=None):
    """can_see rows of the events [cuts[0], cuts[-1]) by chunks [cuts[k], cuts[k+1]), given the final
    rows below cuts[0] (in L).  Rows below a0 are READ by the sweeps (zone i); rows in [a0, cuts[0]) are
    final too but the sweeps treat them as unknown (leaves) and only the repairs read them — a0 = 0 is
    what the device does (no sweep ever reads a row from memory: chunk 0 gets a halo like the others).
    Every local sweep only reads zone-(i) rows and its own window, so the G sweeps are independent (the
    loop below could run them in any order or concurrently); the repairs run in ascending chunk order.
    `resweep_limit`: a chunk with more provisional entries than this is swept again sequentially
    instead of repaired entry by entry (what the device does when gathers would cost more than the
    dependent sweep).  Returns (L, stats)."""
    N = len(cr)
    assert cuts[0] >= a0 and all(x < y for x, y in zip(cuts, cuts[1:])) and cuts[-1] <= N
    if L is None:
        L = np.full((N, n), -1, np.int32)
    stats = dict(prov=[], fixed=[], resweeps=0, halo_events=0)
    windows = []
    for k in range(len(cuts) - 1):
        a, b = cuts[k], cuts[k + 1]
        w = max(a0, a - halo)
        windows.append((w, a, b))
    for w, a, b in reversed(windows):            # any order: the sweeps do not depend on each other
        stats["prov"].insert(0, local_sweep(n, cr, sp, op, L, a0, w, a, b))
        stats["halo_events"] += a - w
    for k, (w, a, b) in enumerate(windows):
        if stats["prov"][k] == 0:
            stats["fixed"].append(0)
        elif resweep_limit is not None and stats["prov"][k] > resweep_limit:
            local_sweep(n, cr, sp, op, L, a, a, a, b)   # everything below a is final: an exact sweep
            stats["resweeps"] += 1
            stats["fixed"].append(-1)
        else:
            stats["fixed"].append(fixup(n, cr, L, a0, w, a, b))
    return L, stats
