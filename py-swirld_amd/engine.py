"""Array-in / array-out front end of the C-ABI: one `Hashgraph` = the voting state of
one Node view, resident in HBM.  Mirrors the reference's hot-path methods
(swirld.py:187-311) on dense event / member indices."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Counters, SwirldHipError, Timings


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Hashgraph:
    def __init__(self, n_members, stake=None, coin_period=6, device=0):
        self._L = _lib.load()
        self.n = int(n_members)
        if stake is None:
            st = np.ones(self.n, np.uint64)
        else:
            raw = np.asarray(stake)
            if raw.shape != (self.n,):
                raise ValueError("stake must have one entry per member")
            # the reference compares sums of stakes with 2*tot/3 in Python arithmetic (swirld.py:41-44); the
            # device tallies are integer: non-integer, negative or non-finite stakes are refused, not truncated
            if raw.dtype.kind not in "iu":
                as_f = raw.astype(np.float64)
                if not np.all(np.isfinite(as_f)) or np.any(as_f != np.floor(as_f)):
                    raise ValueError("stakes must be integers (got %r)" % (raw.tolist(),))
            if np.any(raw.astype(np.float64) < 0):
                raise ValueError("stakes must be non-negative")
            st = np.ascontiguousarray(raw, np.uint64)
        self.stake = st
        self._h = C.c_void_p()
        rc = self._L.sw_create(self.n, _p(st), int(coin_period), int(device), C.byref(self._h))
        if rc != 0:
            raise SwirldHipError(rc, (self._L.sw_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SwirldHipError(rc, (self._L.sw_last_error(self._h) or b"").decode())

    # ---- ingest (Node.add_event, swirld.py:114-120) ----
    def reserve(self, n_events):
        self._chk(self._L.sw_reserve(self._h, int(n_events)))

    def append_events(self, creator, self_parent, other_parent, t=None, sig=None):
        creator = np.ascontiguousarray(creator, np.int32)
        sp = np.ascontiguousarray(self_parent, np.int32)
        op = np.ascontiguousarray(other_parent, np.int32)
        K = creator.shape[0]
        if sp.shape != (K,) or op.shape != (K,):
            raise ValueError("creator / parents must have equal length")
        t = None if t is None else np.ascontiguousarray(t, np.float64)
        sig = None if sig is None else np.ascontiguousarray(sig, np.uint8).reshape(K, 64)
        self._chk(self._L.sw_append_events(self._h, K, _p(creator), _p(sp), _p(op), _p(t), _p(sig)))

    @property
    def num_events(self):
        return int(self._L.sw_num_events(self._h))

    # ---- hot path ----
    def divide_rounds(self, first, K):
        self._chk(self._L.sw_divide_rounds(self._h, int(first), int(K)))

    def decide_fame(self):
        cap = self.max_round + 2
        out = np.empty(max(cap, 1), np.int32)
        n_new = C.c_int()
        self._chk(self._L.sw_decide_fame(self._h, _p(out), int(out.shape[0]), C.byref(n_new)))
        return out[: n_new.value].copy()

    def decide_fame_partial(self, part, nparts):
        """This part's share of the elections (candidate rounds max_c + part, + nparts, ...):
        (famous[R][n] int8 with -1 = undecided or not owned, decided[R] uint8); nothing committed."""
        R = self.max_round + 1
        fam = np.empty((max(R, 1), self.n), np.int8)
        dec = np.empty(max(R, 1), np.uint8)
        r_out = C.c_int()
        self._chk(self._L.sw_decide_fame_partial(self._h, int(part), int(nparts), _p(fam), _p(dec), R, C.byref(r_out)))
        return fam[:R], dec[:R]

    def commit_fame(self, famous, decided):
        """Commit the element-wise MAX of all parts' tables: the context is then exactly as after
        decide_fame(); returns new_c."""
        fam = np.ascontiguousarray(famous, np.int8)
        dec = np.ascontiguousarray(decided, np.uint8)
        R = dec.shape[0]
        out = np.empty(max(R, 1), np.int32)
        n_new = C.c_int()
        self._chk(self._L.sw_commit_fame(self._h, _p(fam), _p(dec), R, _p(out), int(out.shape[0]), C.byref(n_new)))
        return out[: n_new.value].copy()

    @staticmethod
    def split_link(parts):
        """Link the contexts `parts` (2 .. 8 Hashgraph objects holding the same events, one per GPU or several on one): from now
        on ONE round loop runs over all of them, split inside its iterations (include/swirld_hip.h, part 3).  Every part then
        calls divide_rounds with the same arguments from its own thread: `split_divide_rounds` does that."""
        arr = (C.c_void_p * len(parts))(*[p._h for p in parts])
        rc = parts[0]._L.sw_split_link(arr, len(parts))
        if rc != 0:
            msg = next((m for m in (p._L.sw_last_error(p._h).decode() for p in parts) if m), "") or parts[0]._L.sw_last_error(None).decode()
            raise SwirldHipError(rc, msg)

    def split_unlink(self):
        self._chk(self._L.sw_split_unlink(self._h))

    @staticmethod
    def split_rewind(parts):
        """rewind every linked part and wait for all of them: none may divide again before every one has rewound
        (include/swirld_hip.h, part 3)"""
        for p in parts:
            p.rewind()
        for p in parts:
            p.synchronize()

    @staticmethod
    def split_divide_rounds(parts, first, K):
        """divide_rounds(first, K) on every linked context at once, one host thread per part (the C calls release the GIL and meet
        iteration by iteration on the device); raises the first part's error."""
        import threading
        errs = [None] * len(parts)

        def work(i):
            try:
                parts[i].divide_rounds(first, K)
            except Exception as exc:  # noqa: BLE001
                errs[i] = exc
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for e in errs:
            if e is not None:
                raise e

    def find_order(self, rounds):
        rounds = np.ascontiguousarray(sorted(int(r) for r in rounds), np.int32)
        cap = self.num_events
        out = np.empty(max(cap, 1), np.int32)
        n_out = C.c_int64()
        self._chk(self._L.sw_find_order(self._h, _p(rounds), len(rounds), _p(out), cap, C.byref(n_out)))
        return out[: n_out.value].copy()

    # ---- state views ----
    @property
    def max_round(self):
        v = C.c_int()
        self._chk(self._L.sw_max_round(self._h, C.byref(v)))
        return v.value

    def rounds(self, first=0, K=None):
        K = self.num_events - first if K is None else K
        out = np.empty(K, np.int32)
        self._chk(self._L.sw_get_round(self._h, first, K, _p(out)))
        return out

    def heights(self, first=0, K=None):
        K = self.num_events - first if K is None else K
        out = np.empty(K, np.int32)
        self._chk(self._L.sw_get_height(self._h, first, K, _p(out)))
        return out

    def can_see(self, first=0, K=None):
        K = self.num_events - first if K is None else K
        out = np.empty((K, self.n), np.int32)
        self._chk(self._L.sw_get_can_see(self._h, first, K, _p(out)))
        return out

    def sees_masks(self, first=0, K=None):
        K = self.num_events - first if K is None else K
        out = np.empty((K, (self.n + 63) // 64), np.uint64)
        self._chk(self._L.sw_get_sees_mask(self._h, first, K, _p(out)))
        return out

    def witnesses(self, r0=0, r1=None):
        r1 = self.max_round + 1 if r1 is None else r1
        out = np.empty((max(r1 - r0, 0), self.n), np.int32)
        self._chk(self._L.sw_get_witnesses(self._h, r0, r1, _p(out)))
        return out

    def witness_order(self, r):
        """Members of witnesses[r] in dict insertion order (swirld.py:234, 240)."""
        out = np.empty(self.n, np.int32)
        k = C.c_int(0)
        self._chk(self._L.sw_get_witness_order(self._h, int(r), _p(out), C.byref(k)))
        return out[:k.value].copy()

    def famous(self, r0=0, r1=None):
        r1 = self.max_round + 1 if r1 is None else r1
        out = np.empty((max(r1 - r0, 0), self.n), np.int8)
        self._chk(self._L.sw_get_famous(self._h, r0, r1, _p(out)))
        return out

    def famous_events(self, first=0, K=None):
        """Node.famous keyed by event: -1 undecided / not a witness, 0 / 1 (swirld.py:64)."""
        K = self.num_events - first if K is None else K
        out = np.empty(max(K, 0), np.int8)
        self._chk(self._L.sw_get_famous_events(self._h, first, K, _p(out)))
        return out

    def consensus(self, r0=0, r1=None):
        r1 = self.max_round + 1 if r1 is None else r1
        out = np.empty(max(r1 - r0, 0), np.uint8)
        self._chk(self._L.sw_get_consensus(self._h, r0, r1, _p(out)))
        return out

    def vote(self, rv, mv, rc, mc):
        """Node.votes[witness (rv, mv)][witness (rc, mc)]: 0 / 1, or -1 when there is no entry."""
        v = C.c_int8()
        self._chk(self._L.sw_get_vote(self._h, int(rv), int(mv), int(rc), int(mc), C.byref(v)))
        return int(v.value)

    # ---- gossip side from device state (N4) ----
    def known_heights(self, head_event):
        """{member -> height of the newest event of that member `head_event` sees} as an array, -1 absent
        (what Node.sync reports, swirld.py:125-126)."""
        out = np.empty(self.n, np.int32)
        self._chk(self._L.sw_get_known_heights(self._h, int(head_event), _p(out)))
        return out

    def sync_diff(self, head_event, known_height):
        """Chain position ranges (first[m], end[m]) of the events a peer with these heights is missing
        (what Node.ask_sync sends, swirld.py:154-161)."""
        kn = np.ascontiguousarray(known_height, np.int32)
        first = np.empty(self.n, np.int32)
        end = np.empty(self.n, np.int32)
        tot = C.c_int64()
        self._chk(self._L.sw_sync_diff(self._h, int(head_event), _p(kn), _p(first), _p(end), C.byref(tot)))
        return first, end, int(tot.value)

    def chain_events(self, member, p0, p1):
        out = np.empty(max(p1 - p0, 0), np.int32)
        self._chk(self._L.sw_get_chain_events(self._h, int(member), int(p0), int(p1), _p(out)))
        return out

    def transactions(self):
        n = C.c_int64()
        self._chk(self._L.sw_num_ordered(self._h, C.byref(n)))
        out = np.empty(n.value, np.int32)
        self._chk(self._L.sw_get_transactions(self._h, 0, n.value, _p(out)))
        return out

    # ---- measurement ----
    def counters(self):
        c = Counters()
        # (size-aware call: a library whose struct has grown cannot overrun this mirror)
        self._chk(self._L.sw_get_counters_sized(self._h, C.byref(c), C.sizeof(c)))
        return {k: int(getattr(c, k)) for k, _ in Counters._fields_}

    TALLY_KERNELS = ("k_tally", "k_tally_bits", "k_tally_tree")

    @property
    def tally_kernel(self):
        """Name of the step-3 kernel of the round loop the most recent divide_rounds used (sw_get_tally_impl)."""
        i = int(self._L.sw_get_tally_impl(self._h))
        if not 0 <= i < len(self.TALLY_KERNELS):
            raise SwirldHipError(i, "sw_get_tally_impl")
        return self.TALLY_KERNELS[i]

    def set_profiling(self, enable=True):
        self._chk(self._L.sw_set_profiling(self._h, 1 if enable else 0))

    def timings(self):
        t = Timings()
        self._chk(self._L.sw_get_timings(self._h, C.byref(t)))
        return {k: (int(getattr(t, k)) if k.endswith("_launches") else float(getattr(t, k)))
                for k, _ in Timings._fields_}

    def debug_clocks(self):
        """[4096][32] uint64 phase stamps of the round-loop kernels (needs SW_DEBUG_CLOCKS=1 at creation)."""
        out = np.zeros((4096, 32), np.uint64)
        self._chk(self._L.sw_debug_clocks(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def debug_block_clocks(self, iters=1024):
        """[iters][2][2048] uint64: when every workgroup of k_resolve_band / the tally kernel was done (SW_DEBUG_CLOCKS=3)."""
        out = np.zeros((int(iters), 2, 2048), np.uint64)
        self._chk(self._L.sw_debug_block_clocks(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_forks(self, accept=True):
        """Forked events: accepted (the context moves to the exact path, csrc/exact.hip.h — the reference's
        statements on the device, identical results, far slower) or refused with SW_ENOTSUP."""
        self._chk(self._L.sw_set_forks(self._h, 1 if accept else 0))

    @property
    def exact(self):
        """True once a forked event moved the context to the exact path."""
        v = C.c_int(0)
        self._chk(self._L.sw_get_exact(self._h, C.byref(v)))
        return bool(v.value)

    def set_window(self, enable=True, chunk_mb=0, lapse_events=0):
        """Windowed can_see table (before the first append): rows no later call can read are evicted
        after every find_order (HIP virtual memory management, include/swirld_hip.h).  lapse_events > 0:
        a member silent for more than that many events stops holding the window back and its further
        events are refused (sw_set_window_lapse; the reference keeps every row)."""
        self._chk(self._L.sw_set_window(self._h, 1 if enable else 0, int(chunk_mb)))
        if enable and lapse_events:
            self._chk(self._L.sw_set_window_lapse(self._h, int(lapse_events)))

    def window(self):
        """(first resident event, bytes of the can_see table currently mapped, evictions so far)."""
        a, b, e = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.sw_get_window(self._h, C.byref(a), C.byref(b), C.byref(e)))
        return int(a.value), int(b.value), int(e.value)

    # ---- multi-GPU split of the can_see table by event ranges (include/swirld_hip.h) ----
    @property
    def row_stride(self):
        """int32 elements per can_see row on the device (members padded to a multiple of 64)."""
        return int(self._L.sw_row_stride(self._h))

    def cansee_range(self, first, K):
        """Sweep the can_see rows of [first, first + K) from a halo (asynchronous); rows below the halo are leaves."""
        self._chk(self._L.sw_cansee_range(self._h, int(first), int(K)))

    def cansee_repair(self, first, K):
        """Repair the provisional entries of a swept range from the (imported) final rows below it; device-gated."""
        self._chk(self._L.sw_cansee_repair(self._h, int(first), int(K)))

    def export_rows(self, first, K, dst_ptr, stream=0):
        """Copy rows [first, first + K) to DEVICE memory at `dst_ptr` (K * row_stride int32); `stream` (a raw
        hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) is made to wait for the copy."""
        self._chk(self._L.sw_export_rows(self._h, int(first), int(K), C.c_void_p(int(dst_ptr)), C.c_void_p(int(stream))))

    def import_rows(self, first, K, src_ptr, stream=0):
        """Copy rows [first, first + K) from DEVICE memory at `src_ptr` into the table, after what `stream` has
        enqueued so far; sw_divide_rounds will not sweep them."""
        self._chk(self._L.sw_import_rows(self._h, int(first), int(K), C.c_void_p(int(src_ptr)), C.c_void_p(int(stream))))

    def range_stats(self):
        """(provisional entries counted by the range sweeps, entries changed by the repairs, ranges swept twice)."""
        a, b, e = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.sw_get_range_stats(self._h, C.byref(a), C.byref(b), C.byref(e)))
        return int(a.value), int(b.value), int(e.value)

    def rewind(self):
        """Forget all voting state; the ingested events stay resident (bench utility)."""
        self._chk(self._L.sw_rewind(self._h))

    def reset(self):
        """rewind() and forget the events too; device storage stays allocated (bench utility)."""
        self._chk(self._L.sw_reset(self._h))

    def synchronize(self):
        self._chk(self._L.sw_synchronize(self._h))


def _pack_messages(msgs):
    off = np.zeros(len(msgs) + 1, np.int64)
    np.cumsum([len(m) for m in msgs], out=off[1:])
    return np.frombuffer(b"".join(msgs), np.uint8) if off[-1] else np.zeros(1, np.uint8), off


def verify_batch(msgs, sigs, pks, device=0):
    """Batch Ed25519 verification on the GPU (sw_crypto_verify_batch): one bool per (message,
    64-byte signature, 32-byte public key), equal to libsodium's crypto_sign_verify_detached
    accepting it (swirld.py:99-100)."""
    L = _lib.load()
    K = len(msgs)
    if not (len(sigs) == len(pks) == K):
        raise ValueError("msgs, sigs and pks must have equal length")
    if any(len(s) != 64 for s in sigs) or any(len(p) != 32 for p in pks):
        raise ValueError("signatures are 64 bytes, public keys 32 bytes")
    data, off = _pack_messages(msgs)
    sg = np.frombuffer(b"".join(sigs), np.uint8) if K else np.zeros(1, np.uint8)
    pk = np.frombuffer(b"".join(pks), np.uint8) if K else np.zeros(1, np.uint8)
    ok = np.zeros(max(K, 1), np.uint8)
    rc = L.sw_crypto_verify_batch(int(device), K, _p(data), _p(off), _p(sg), _p(pk), _p(ok))
    if rc != 0:
        raise SwirldHipError(rc, (L.sw_last_error(None) or b"").decode())
    return ok[:K].astype(bool)


def hash_batch(msgs, device=0):
    """Batch BLAKE2b-256 on the GPU (sw_crypto_hash_batch): the event ids of swirld.py:95, 103."""
    L = _lib.load()
    K = len(msgs)
    data, off = _pack_messages(msgs)
    out = np.zeros((max(K, 1), 32), np.uint8)
    rc = L.sw_crypto_hash_batch(int(device), K, _p(data), _p(off), _p(out))
    if rc != 0:
        raise SwirldHipError(rc, (L.sw_last_error(None) or b"").decode())
    return [bytes(out[i]) for i in range(K)]


def synth_hashgraph(n, N, seed, mode=0, p0=0.0, p1=0.0, with_sig=True):
    """Host-side synthetic gossip hashgraph (csrc/synth.cpp): returns
    (creator, self_parent, other_parent, t, sig) as numpy arrays."""
    L = _lib.load()
    cr = np.empty(N, np.int32)
    sp = np.empty(N, np.int32)
    op = np.empty(N, np.int32)
    t = np.empty(N, np.float64)
    sig = np.empty((N, 64), np.uint8) if with_sig else None
    rc = L.sw_synth_hashgraph(int(n), int(N), int(seed), int(mode), float(p0), float(p1),
                              _p(cr), _p(sp), _p(op), _p(t), _p(sig))
    if rc != 0:
        raise SwirldHipError(rc, "sw_synth_hashgraph: invalid arguments")
    return cr, sp, op, t, sig
