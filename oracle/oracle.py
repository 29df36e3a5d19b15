"""ctypes wrapper of oracle/swirld_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The oracle restates /root/reference/swirld.py:187-311 sequentially on the CPU and is
pinned against outputs of the unmodified reference (tests/golden/*.npz).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libswirld_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "swirld_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libswirld_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.or_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.or_destroy.argtypes = [C.c_void_p]
        L.or_destroy.restype = None
        L.or_append_events.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.or_divide_rounds.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.or_decide_fame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.or_find_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64)]
        L.or_num_events.argtypes = [C.c_void_p]
        L.or_num_events.restype = C.c_int64
        L.or_max_round.argtypes = [C.c_void_p]
        for name in ("round", "height", "cansee", "famous", "tbd", "transactions"):
            f = getattr(L, "or_%s_ptr" % name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        L.or_num_ordered.argtypes = [C.c_void_p]
        L.or_num_ordered.restype = C.c_int64
        L.or_get_witnesses.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.or_get_witness_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.or_get_consensus.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.or_get_vote.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.or_num_votes.argtypes = [C.c_void_p]
        L.or_num_votes.restype = C.c_int64
        L.or_votes_digest.argtypes = [C.c_void_p]
        L.or_votes_digest.restype = C.c_uint64
        L.or_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.or_get_counters.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


_ERR = {-2: "KeyError in the reference", -3: "IndexError in the reference (swirld.py:305)",
        -12: "out of memory", -22: "invalid argument"}


def _chk(rc):
    if rc != 0:
        raise OracleError("oracle rc=%d (%s)" % (rc, _ERR.get(rc, "?")))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Sequential CPU restatement of Node's voting state (dense indices)."""

    def __init__(self, n, stake=None, coin_period=6):
        self.n = int(n)
        st = np.ones(n, np.uint64) if stake is None else np.ascontiguousarray(stake, np.uint64)
        assert st.shape == (n,)
        self.stake = st
        self._h = C.c_void_p()
        _chk(lib().or_create(self.n, _p(st), int(coin_period), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().or_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def N(self):
        return int(lib().or_num_events(self._h))

    def append_events(self, creator, sp, op, t=None, sig=None):
        creator = np.ascontiguousarray(creator, np.int32)
        sp = np.ascontiguousarray(sp, np.int32)
        op = np.ascontiguousarray(op, np.int32)
        K = creator.shape[0]
        t = None if t is None else np.ascontiguousarray(t, np.float64)
        sig = None if sig is None else np.ascontiguousarray(sig, np.uint8).reshape(K, 64)
        _chk(lib().or_append_events(self._h, K, _p(creator), _p(sp), _p(op), _p(t), _p(sig)))

    def divide_rounds(self, first, K):
        _chk(lib().or_divide_rounds(self._h, int(first), int(K)))

    def decide_fame(self):
        cap = max(1, self.max_round + 2)
        out = np.empty(cap, np.int32)
        n_new = C.c_int()
        _chk(lib().or_decide_fame(self._h, _p(out), cap, C.byref(n_new)))
        return out[: n_new.value].copy()

    def find_order(self, rounds):
        rounds = np.ascontiguousarray(sorted(int(r) for r in rounds), np.int32)
        cap = self.N
        out = np.empty(max(cap, 1), np.int32)
        n_out = C.c_int64()
        _chk(lib().or_find_order(self._h, _p(rounds), len(rounds), _p(out), cap, C.byref(n_out)))
        return out[: n_out.value].copy()

    def _arr(self, name, dtype, shape):
        ptr = getattr(lib(), "or_%s_ptr" % name)(self._h)
        if not ptr or int(np.prod(shape)) == 0:
            return np.zeros(shape, dtype)
        buf = (C.c_char * (int(np.prod(shape)) * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    @property
    def round(self):
        return self._arr("round", np.int32, (self.N,))

    @property
    def height(self):
        return self._arr("height", np.int32, (self.N,))

    @property
    def can_see(self):
        return self._arr("cansee", np.int32, (self.N, self.n))

    @property
    def famous_by_event(self):
        return self._arr("famous", np.int8, (self.N,))

    @property
    def tbd(self):
        return self._arr("tbd", np.uint8, (self.N,))

    @property
    def transactions(self):
        return self._arr("transactions", np.int32, (int(lib().or_num_ordered(self._h)),))

    @property
    def max_round(self):
        return int(lib().or_max_round(self._h))

    def witnesses(self, r0=0, r1=None):
        r1 = self.max_round + 1 if r1 is None else r1
        out = np.empty((max(r1 - r0, 0), self.n), np.int32)
        lib().or_get_witnesses(self._h, r0, r1, _p(out))
        return out

    def witness_order(self, r):
        out = np.empty(self.n, np.int32)
        k = lib().or_get_witness_order(self._h, int(r), _p(out))
        return out[:k].copy()

    def famous_table(self, r0=0, r1=None):
        """[-1 undecided / 0 / 1] per witness slot, -1 where there is no witness."""
        w = self.witnesses(r0, r1)
        f = self.famous_by_event
        out = np.full(w.shape, -1, np.int8)
        m = w >= 0
        out[m] = f[w[m]]
        return out

    def consensus(self, r0=0, r1=None):
        r1 = self.max_round + 1 if r1 is None else r1
        out = np.empty(max(r1 - r0, 0), np.uint8)
        lib().or_get_consensus(self._h, r0, r1, _p(out))
        return out

    def vote(self, voter_event, cand_event):
        return int(lib().or_get_vote(self._h, int(voter_event), int(cand_event)))

    @property
    def num_votes(self):
        return int(lib().or_num_votes(self._h))

    @property
    def votes_digest(self):
        return int(lib().or_votes_digest(self._h))

    def counters(self):
        out = np.zeros(8, np.int64)
        lib().or_get_counters(self._h, _p(out))
        return {"voter_evals": int(out[0]), "majority_evals": int(out[1]),
                "tally_inner": int(out[2]), "rounds": int(out[3]),
                "coin_votes": int(out[4]), "coin_flips": int(out[5]), "max_vote_distance": int(out[6])}
