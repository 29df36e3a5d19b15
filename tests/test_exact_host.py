"""CPU: the functions the exact (forked-hashgraph) kernels execute (py-swirld_amd/csrc/exact.hip.h:
divide / decide_fame / find_order, the reference's statements on the device-resident layout), compiled
for the HOST by g++ (tests/exact_host.cpp, one "lane") and run against every golden fixture of the
unmodified reference — the forked DAG included — and against the oracle on random forked hashgraphs,
batch and incremental schedules.  Two host builds: one "lane" (the plain sequential statement), and 16
cooperative fibers that hand over at every sync — the opposite extreme of a wavefront's lockstep
execution — so that the lane-parallel structure the kernels really run (lane-owned columns, wave
sums, lane-0 sections) is exercised on the CPU and needs nothing but its syncs to be right.
The GPU run of the same cases is tests/test_gpu_forks.py."""
import ctypes as C
import os

import numpy as np
import pytest

import hostlibs
from conftest import ROOT, golden_names, load_golden
from oracle.oracle import Oracle
from synth_util import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def build_host_lib(lanes=False):
    return hostlibs.build_exact_host(lanes)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ExactHost:
    """Oracle-like driver over the host build of exact.hip.h."""

    def __init__(self, n, stake=None, coin_period=6, lanes=False):
        L = C.CDLL(build_host_lib(lanes))
        L.swx_host_create.restype = C.c_void_p
        L.swx_host_create.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.swx_host_destroy.argtypes = [C.c_void_p]
        L.swx_host_append.argtypes = [C.c_void_p, C.c_longlong] + [C.c_void_p] * 5
        L.swx_host_divide.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong]
        L.swx_host_fame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.swx_host_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.swx_host_R.argtypes = [C.c_void_p]
        L.swx_host_get.argtypes = [C.c_void_p] * 10
        L.swx_host_import_check.argtypes = [C.c_void_p]
        self.L = L
        self.n = n
        st = np.ones(n, np.uint32) if stake is None else np.ascontiguousarray(stake, np.uint32)
        self.h = L.swx_host_create(n, _p(st), coin_period)
        self.N = 0
        self.tx = []

    def __del__(self):
        if getattr(self, "h", None):
            self.L.swx_host_destroy(self.h)
            self.h = None

    def append_events(self, cr, sp, op, t=None, sig=None):
        cr = np.ascontiguousarray(cr, np.int32); sp = np.ascontiguousarray(sp, np.int32); op = np.ascontiguousarray(op, np.int32)
        t = None if t is None else np.ascontiguousarray(t, np.float64)
        sig = None if sig is None else np.ascontiguousarray(sig, np.uint8)
        assert self.L.swx_host_append(self.h, len(cr), _p(cr), _p(sp), _p(op), _p(t), _p(sig)) == 0
        self.N += len(cr)

    def divide_rounds(self, first, K):
        rc = self.L.swx_host_divide(self.h, first, K)
        assert rc == 0, rc

    def decide_fame(self):
        out = np.zeros(self.L.swx_host_R(self.h) + 1, np.int32)
        n_new = C.c_int(0)
        rc = self.L.swx_host_fame(self.h, _p(out), C.byref(n_new))
        assert rc == 0, rc
        return out[:n_new.value].copy()

    def find_order(self, rounds):
        rounds = np.ascontiguousarray(list(rounds), np.int32)
        out = np.zeros(self.N + 1, np.int32)
        n_out = C.c_longlong(0)
        rc = self.L.swx_host_order(self.h, _p(rounds), len(rounds), _p(out), C.byref(n_out))
        if rc != 0:
            raise IndexError(rc)
        self.tx.extend(out[:n_out.value].tolist())
        return out[:n_out.value].copy()

    def state(self):
        R = self.L.swx_host_R(self.h)
        n, N = self.n, self.N
        rnd = np.zeros(N, np.int32); Lt = np.zeros((N, n), np.int32)
        wit = np.zeros((R, n), np.int32); wo = np.zeros((R, n), np.int32); wc = np.zeros(R, np.int32)
        cons = np.zeros(R, np.uint8); fam = np.zeros(N, np.int8); tbd = np.zeros(N, np.uint8); fs = np.zeros((R, n), np.int8)
        self.L.swx_host_get(self.h, _p(rnd), _p(Lt), _p(wit), _p(wo), _p(wc), _p(cons), _p(fam), _p(tbd), _p(fs))
        return dict(round=rnd, can_see=Lt, wit=wit, worder=[wo[r, :wc[r]] for r in range(R)], cons=cons, fam=fam, tbd=tbd, fam_slot=fs)

    def reimport(self):
        self.L.swx_host_import_check(self.h)


def add_forks(stream, n, seed, n_forks, start=0):
    """Forked events (same creator and self-parent as an existing event, another other-parent), each
    followed by an event of a third member that builds on the fork — spread over the whole stream, with
    every index kept topological."""
    cr, sp, op, t, sig = [list(a) if a.ndim == 1 else [row for row in a] for a in stream]
    rng = np.random.default_rng(seed)
    N0 = len(cr)
    out = ([], [], [], [], [])
    remap = {}

    def push(c, s, o, tt, sg):
        out[0].append(c); out[1].append(s); out[2].append(o); out[3].append(tt); out[4].append(sg)
        return len(out[0]) - 1

    fork_at = set(int(x) for x in rng.integers(max(2 * n, start), N0, n_forks))
    last_of = {}
    for i in range(N0):
        s = remap[sp[i]] if sp[i] >= 0 else -1
        o = remap[op[i]] if op[i] >= 0 else -1
        remap[i] = push(cr[i], s, o, t[i], sig[i])
        last_of[cr[i]] = remap[i]
        if i in fork_at and s >= 0:
            cands = [k for k in range(len(out[0]) - 1) if out[0][k] != cr[i] and k != o]
            o2 = int(rng.choice(cands))
            f = push(cr[i], s, o2, t[i] + 0.25, rng.integers(0, 256, 64, dtype=np.uint8))
            third = int(rng.choice([c for c in last_of if c != cr[i]]))
            k = push(third, last_of[third], f, t[i] + 0.5, rng.integers(0, 256, 64, dtype=np.uint8))
            last_of[third] = k
            # the stream's later events of `third` must build on its new head: rewrite through remap
            for j in range(i, -1, -1):
                if cr[j] == third:
                    remap[j] = k
                    break
    return (np.array(out[0], np.int32), np.array(out[1], np.int32), np.array(out[2], np.int32),
            np.array(out[3], np.float64), np.array(out[4], np.uint8).reshape(-1, 64))


def run_both(n, stream, chunk, stake=None, with_order=True, lanes=False):
    cr, sp, op, t, sig = stream
    N = len(cr)
    o, x = Oracle(n, stake), ExactHost(n, stake, lanes=lanes)
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for d in (o, x):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nco, ncx = list(o.decide_fame()), list(x.decide_fame())
        assert nco == ncx, "new_c of the call ending at %d" % b
        if with_order:
            assert list(o.find_order(nco)) == list(x.find_order(ncx)), "find_order of the call ending at %d" % b
    return o, x


def assert_same(o, x):
    st = x.state()
    assert np.array_equal(st["round"], o.round)
    assert np.array_equal(st["can_see"], o.can_see)
    assert np.array_equal(st["wit"], o.witnesses())
    for r, order in enumerate(st["worder"]):
        assert np.array_equal(order, o.witness_order(r)), "dict order of witnesses[%d]" % r
    assert np.array_equal(st["fam"], o.famous_by_event)
    assert np.array_equal(st["fam_slot"], o.famous_table())
    assert np.array_equal(st["cons"], o.consensus())
    assert np.array_equal(st["tbd"], o.tbd)
    assert np.array_equal(np.array(x.tx, np.int32), o.transactions)


def _small_goldens():  # (the large fixtures are the fast path's business; this path is sequential)
    return [name for name in golden_names() if len(load_golden(name)["creator"]) <= 4000]


def _golden_cases():  # (a fiber switch per lane and sync: the many-call schedules stay with the 1-lane build)
    return [pytest.param(name, lanes, id="%s-%s" % (name, "16lanes" if lanes else "1lane"))
            for name in _small_goldens() for lanes in (False, True)
            if not (lanes and len(load_golden(name)["batches"]) > 60)]


@pytest.mark.parametrize("name,lanes", _golden_cases())
def test_exact_path_matches_reference_golden(name, lanes):
    g = load_golden(name)
    x = ExactHost(g["n"], g["stake"], lanes=lanes)
    calls = 0
    for a, b in g["batches"]:
        x.append_events(g["creator"][a:b], g["self_parent"][a:b], g["other_parent"][a:b], g["t"][a:b], g["sig"][a:b])
        x.divide_rounds(a, b - a)
        nc = x.decide_fame()
        assert list(nc) == list(g["new_c_flat"][g["new_c_off"][calls]:g["new_c_off"][calls + 1]])
        tx = x.find_order(nc)
        assert list(tx) == list(g["transactions"][g["tx_off"][calls]:g["tx_off"][calls + 1]])
        calls += 1
    st = x.state()
    assert np.array_equal(st["round"], g["round"])
    assert np.array_equal(st["can_see"], g["can_see"])
    assert np.array_equal(st["wit"], g["witnesses"])
    for r, order in enumerate(g["wit_order"]):
        assert np.array_equal(st["worder"][r], order)
    assert np.array_equal(st["fam"], g["famous"])
    assert np.array_equal(st["cons"], g["consensus"])
    assert np.array_equal(st["tbd"], g["tbd"])


FORK_CASES = [
    (8, 600, 1, 10, None, 0, 0, 0), (8, 600, 2, 10, 37, 0, 0, 0), (5, 400, 3, 25, 1, 0, 0, 0),
    (16, 1500, 4, 30, 100, 2, 0.3, 0.1), (70, 5000, 5, 12, 1000, 0, 0, 0), (4, 900, 6, 40, 9, 0, 0, 0),
    (12, 1000, 7, 20, None, 1, 0.02, 0),
]


@pytest.mark.parametrize("n,N,seed,forks,chunk,mode,p0,p1,lanes",
                         [c + (lanes,) for c in FORK_CASES for lanes in (False, True)
                          if not (lanes and c[4] is not None and c[4] < 30)])
def test_exact_path_matches_oracle_on_forked_hashgraphs(n, N, seed, forks, chunk, mode, p0, p1, lanes):
    stream = add_forks(synth(n, N, seed, mode, p0, p1), n, seed, forks)
    o, x = run_both(n, stream, chunk, lanes=lanes)
    assert_same(o, x)
    assert o.max_round >= 3


@pytest.mark.parametrize("lanes", [False, True], ids=["1lane", "16lanes"])
def test_exact_path_with_stake(lanes):
    n = 9
    stake = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9], np.uint64)
    stream = add_forks(synth(n, 800, 21), n, 21, 15)
    o, x = run_both(n, stream, 60, stake, lanes=lanes)
    assert_same(o, x)


@pytest.mark.parametrize("lanes", [False, True], ids=["1lane", "16lanes"])
def test_import_of_a_fork_free_state_rebuilds_order_fame_and_tbd(lanes):
    """What a context that ran on the fast path hands over at its first fork: per-slot tables only."""
    n = 10
    stream = synth(n, 900, 33)
    o, x = run_both(n, stream, 45, lanes=lanes)
    before = x.state()
    x.reimport()
    after = x.state()
    for k in ("fam", "tbd", "cons", "wit"):
        assert np.array_equal(before[k], after[k]), k
    for a, b in zip(before["worder"], after["worder"]):
        assert np.array_equal(a, b)
