#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/swirld.py, imported through tests/refharness.py with the pysodium
stand-in) on seeded synthetic event streams.  Runs in the authoring container only
(the reference tree does not travel to the GPU box); the resulting small fixtures are
committed and pin both the CPU oracle (tests/test_oracle_golden.py) and the HIP path
(tests/test_gpu_parity.py).

Each fixture stores the input stream (so it does not depend on the generator staying
bit-stable), the call schedule, and every piece of Node state the hot path produces:
round, height, can_see, witnesses (+ dict order), famous, consensus, votes,
transactions, tbd, and the per-call return values of decide_fame / find_order.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from refharness import RefRun  # noqa: E402
from synth_util import synth  # noqa: E402

# name, n, N, seed, mode, p0, p1, stake, chunk (None = one batch call)
CASES = [
    ("n4_s1_batch", 4, 400, 1, 0, 0, 0, None, None),
    ("n4_s2_batch", 4, 400, 2, 0, 0, 0, None, None),
    ("n4_s3_batch", 4, 600, 3, 0, 0, 0, None, None),
    ("n4_s5_chunk1", 4, 400, 5, 0, 0, 0, None, 1),
    ("n4_s6_chunk7", 4, 400, 6, 0, 0, 0, None, 7),
    ("n5_s1_batch", 5, 500, 1, 0, 0, 0, None, None),
    ("n7_s2_chunk13", 7, 700, 2, 0, 0, 0, None, 13),
    ("n16_s3_batch", 16, 2000, 3, 0, 0, 0, None, None),
    ("n16_s4_chunk50", 16, 2000, 4, 0, 0, 0, None, 50),
    ("n16_s4_chunk250", 16, 2000, 4, 0, 0, 0, None, 250),
    ("n33_s5_batch", 33, 3000, 5, 0, 0, 0, None, None),
    ("n16_s7_cliques", 16, 2000, 7, 1, 0.05, 0, None, None),
    ("n16_s8_slow", 16, 2000, 8, 2, 0.25, 0.05, None, None),
    ("n16_s9_stale", 16, 2000, 9, 3, 0.5, 0, None, None),
    ("n10_s1_stake", 10, 1500, 1, 0, 0, 0, [1] * 9 + [2], None),
    ("n12_s2_stake", 12, 1500, 2, 0, 0, 0, [1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1], None),
    ("n6_s3_stuck_stake", 6, 600, 3, 0, 0, 0, [3, 1, 4, 1, 5, 2], None),
    ("n64_s1_batch", 64, 6000, 1, 0, 0, 0, None, None),
    ("n64_s2_slow_chunk1000", 64, 5000, 2, 2, 0.2, 0.1, None, 1000),
    ("n70_s3_batch", 70, 4000, 3, 0, 0, 0, None, None),
    ("n130_s4_batch", 130, 14000, 4, 0, 0, 0, None, None),
]


def run_case(n, stream, stake, chunk):
    cr, sp, op, t, sig = stream
    N = len(cr)
    ref = RefRun(n, stake)
    chunk = chunk or N
    new_c_flat, new_c_off, tx_off = [], [0], [0]
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        ref.append(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        ref.divide_rounds(a, b - a)
        nc = ref.decide_fame()
        ref.find_order(nc)
        new_c_flat += list(nc)
        new_c_off.append(len(new_c_flat))
        tx_off.append(len(ref.node.transactions))
    ex = ref.extract()
    wo_flat = np.concatenate(ex["wit_order"]) if ex["wit_order"] else np.zeros(0, np.int32)
    wo_off = np.cumsum([0] + [len(o) for o in ex["wit_order"]]).astype(np.int32)
    return dict(
        n=np.int32(n), stake=np.array([1] * n if stake is None else stake, np.uint64),
        chunk=np.int64(chunk), creator=cr, self_parent=sp, other_parent=op, t=t, sig=sig,
        round=ex["round"], height=ex["height"], can_see=ex["can_see"],
        witnesses=ex["witnesses"], wit_order_flat=wo_flat.astype(np.int32), wit_order_off=wo_off,
        famous=ex["famous"], consensus=ex["consensus"], votes=ex["votes"],
        transactions=ex["transactions"], tbd=ex["tbd"],
        new_c_flat=np.array(new_c_flat, np.int32), new_c_off=np.array(new_c_off, np.int32),
        tx_off=np.array(tx_off, np.int64))


def run_mainloop_case(n_nodes, n_turns, seed):
    """BASELINE.json configs[0]: the reference's own simulation swirld.test(n_nodes, n_turns)
    (swirld.py:331-345) with its real gossip, Ed25519 signatures and main() loop, made
    deterministic (seeded libsodium RNG stand-in, counter clock).  Records, for node 0, the
    event stream in the order it was added, the batch sizes its main loop passed to
    divide_rounds, and all resulting state."""
    import contextlib
    import io
    import random
    import refharness
    sw = refharness.import_reference()
    import pysodium
    import utils as ref_utils
    rng = random.Random(seed)
    pysodium.set_rng(lambda k: bytes(rng.getrandbits(8) for _ in range(k)))
    clock = [0.0]

    def fake_time():
        clock[0] += 1.0
        return clock[0]
    old_time, old_rb = sw.time, ref_utils.randombytes
    sw.time = fake_time
    ref_utils.randombytes = pysodium.randombytes
    sched = {}
    orig_dr = sw.Node.divide_rounds
    orig_df = sw.Node.decide_fame
    orig_fo = sw.Node.find_order
    newc_log, tx_log = {}, {}

    def dr(self, events):
        events = tuple(events)
        sched.setdefault(id(self), []).append(len(events))
        return orig_dr(self, events)

    def df(self):
        out = orig_df(self)
        newc_log.setdefault(id(self), []).append(sorted(out))
        return out

    def fo(self, new_c):
        orig_fo(self, new_c)
        tx_log.setdefault(id(self), []).append(len(self.transactions))
    sw.Node.divide_rounds, sw.Node.decide_fame, sw.Node.find_order = dr, df, fo
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            nodes = sw.test(n_nodes, n_turns)
    finally:
        sw.Node.divide_rounds, sw.Node.decide_fame, sw.Node.find_order = orig_dr, orig_df, orig_fo
        sw.time, ref_utils.randombytes = old_time, old_rb
        pysodium.set_rng(None)
    node = nodes[0]
    ids = list(node.hg.keys())            # dict order == add order
    index = {h: i for i, h in enumerate(ids)}
    members = list(node.stake.keys())
    mindex = {pk: i for i, pk in enumerate(members)}
    N = len(ids)
    cr = np.array([mindex[node.hg[h].c] for h in ids], np.int32)
    sp = np.array([index[node.hg[h].p[0]] if node.hg[h].p else -1 for h in ids], np.int32)
    op = np.array([index[node.hg[h].p[1]] if node.hg[h].p else -1 for h in ids], np.int32)
    t = np.array([node.hg[h].t for h in ids], np.float64)
    sig = np.frombuffer(b"".join(node.hg[h].s for h in ids), np.uint8).reshape(N, 64).copy()
    # re-express the state through a RefRun-like extraction
    rr = RefRun.__new__(RefRun)
    rr.node, rr.n, rr.ids, rr.id_index, rr.pk_index = node, n_nodes, ids, index, mindex
    ex = rr.extract()
    batches = [1] + sched[id(node)]       # the root is set up by __init__ (swirld.py:75-80)
    assert sum(batches) == N
    new_c_flat, new_c_off, tx_off = [], [0, 0], [0, 0]  # call 0 = the root, no fame/order call
    for nc, ntx in zip(newc_log[id(node)], tx_log[id(node)]):
        new_c_flat += nc
        new_c_off.append(len(new_c_flat))
        tx_off.append(ntx)
    wo_flat = np.concatenate(ex["wit_order"]) if ex["wit_order"] else np.zeros(0, np.int32)
    wo_off = np.cumsum([0] + [len(o) for o in ex["wit_order"]]).astype(np.int32)
    return dict(
        n=np.int32(n_nodes), stake=np.ones(n_nodes, np.uint64), chunk=np.int64(0),
        sched=np.array(batches, np.int64), creator=cr, self_parent=sp, other_parent=op, t=t, sig=sig,
        round=ex["round"], height=ex["height"], can_see=ex["can_see"], witnesses=ex["witnesses"],
        wit_order_flat=wo_flat.astype(np.int32), wit_order_off=wo_off, famous=ex["famous"],
        consensus=ex["consensus"], votes=ex["votes"], transactions=ex["transactions"], tbd=ex["tbd"],
        new_c_flat=np.array(new_c_flat, np.int32), new_c_off=np.array(new_c_off, np.int32),
        tx_off=np.array(tx_off, np.int64))


def add_forks(stream, n, seed, n_forks):
    """Append forked events: same creator and self-parent as an existing event, other
    other-parent.  The reference accepts these (no fork detection, README.md:84)."""
    cr, sp, op, t, sig = [a.copy() for a in stream]
    rng = np.random.default_rng(seed)
    cr, sp, op, t = list(cr), list(sp), list(op), list(t)
    sig = list(sig)
    N0 = len(cr)
    for _ in range(n_forks):
        i = int(rng.integers(n, N0))
        cands = [k for k in range(N0) if cr[k] != cr[i] and k != op[i]]
        o = int(rng.choice(cands))
        cr.append(cr[i]); sp.append(sp[i]); op.append(o); t.append(float(len(t)))
        sig.append(rng.integers(0, 256, 64, dtype=np.uint8))
        # a descendant of the fork, created by a third member, so that the fork is seen
        third = int(rng.choice([c for c in range(n) if c != cr[i]]))
        heads = [k for k in range(len(cr) - 1) if cr[k] == third]
        cr.append(third); sp.append(max(heads)); op.append(len(cr) - 2); t.append(float(len(t)))
        sig.append(rng.integers(0, 256, 64, dtype=np.uint8))
    return (np.array(cr, np.int32), np.array(sp, np.int32), np.array(op, np.int32),
            np.array(t, np.float64), np.array(sig, np.uint8).reshape(-1, 64))


def main():
    for name, n, N, seed, mode, p0, p1, stake, chunk in CASES:
        stream = synth(n, N, seed, mode, p0, p1)
        out = run_case(n, stream, stake, chunk)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s R=%3d consensus=%3d tx=%5d votes=%6d  %6.1f KB" % (
            name, out["witnesses"].shape[0], int(out["consensus"].sum()),
            len(out["transactions"]), len(out["votes"]), os.path.getsize(path) / 1024))
    # configs[0]: the reference's own main loop (4 members, 1000 turns), node 0's view
    out = run_mainloop_case(4, 1000, 12345)
    path = os.path.join(HERE, "n4_mainloop_node0.npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%d R=%3d consensus=%3d tx=%5d calls=%d" % (
        "n4_mainloop_node0", len(out["creator"]), out["witnesses"].shape[0], int(out["consensus"].sum()),
        len(out["transactions"]), len(out["sched"])))
    # forked DAG, batch (forks appended at the end of the stream)
    base = synth(8, 500, 11, 0, 0, 0)
    stream = add_forks(base, 8, 11, 6)
    out = run_case(8, stream, None, None)
    path = os.path.join(HERE, "n8_s11_forks.npz")
    np.savez_compressed(path, **out)
    print("%-28s R=%3d consensus=%3d tx=%5d" % ("n8_s11_forks", out["witnesses"].shape[0],
                                               int(out["consensus"].sum()), len(out["transactions"])))
    main_forks_incremental()


# forked DAGs under INCREMENTAL call schedules: forks spread over the whole stream (the generator of
# tests/test_exact_host.py), so that siblings replace witnesses between decide_fame() calls
FORK_CASES = [
    ("n8_s12_forks_chunk9", 8, 600, 12, 10, None, 9),
    ("n5_s13_forks_chunk1", 5, 300, 13, 12, None, 1),
    ("n12_s14_forks_stake_chunk40", 12, 1200, 14, 16, [1, 2, 1, 1, 3, 1, 1, 2, 1, 1, 1, 2], 40),
]


def main_forks_incremental():
    from test_exact_host import add_forks as add_forks_spread
    for name, n, N, seed, forks, stake, chunk in FORK_CASES:
        stream = add_forks_spread(synth(n, N, seed, 0, 0, 0), n, seed, forks)
        out = run_case(n, stream, stake, chunk)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("%-28s R=%3d consensus=%3d tx=%5d calls=%d" % (name, out["witnesses"].shape[0], int(out["consensus"].sum()),
                                                            len(out["transactions"]), len(out["new_c_off"]) - 1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--forks-incremental":   # only the fixtures added last
        main_forks_incremental()
    else:
        main()
