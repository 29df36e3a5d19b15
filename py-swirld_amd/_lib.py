"""ctypes binding of the C-ABI (include/swirld_hip.h).  The shared library is built
in-tree by `__graft_entry__.build()` / `py-swirld_amd/build.py`; there is NO fallback: if
the library is missing, or no GPU is present when a context is created, this fails
loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libswirld_hip.so")

SW_OK = 0
ERRNO_NAMES = {-5: "SW_EIO", -12: "SW_ENOMEM", -19: "SW_ENODEV", -22: "SW_EINVAL",
               -34: "SW_ERANGE", -75: "SW_EOVERFLOW", -95: "SW_ENOTSUP"}


class SwirldHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRNO_NAMES.get(code, "error"), code, msg))
        self.code = code


class Counters(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "events_divided", "rounds", "tally_evals", "round_iterations", "voter_evals",
        "majority_evals", "levels", "kernel_launches", "far_hops", "band_events", "coin_votes", "coin_flips",
        "chunk_sweeps", "chunk_provisional", "chunk_repaired", "chunk_resweeps", "finalize_from_rows", "order_rounds_host_sorted")]


class Timings(C.Structure):
    _fields_ = [("can_see_ms", C.c_float), ("rounds_ms", C.c_float), ("tally_ms", C.c_float),
                ("tally_launches", C.c_int32), ("finalize_ms", C.c_float), ("fame_ms", C.c_float),
                ("total_ms", C.c_float), ("cansee_kernel_ms", C.c_float), ("cansee_launches", C.c_int32),
                ("resolve_ms", C.c_float), ("resolve_launches", C.c_int32), ("elections_ms", C.c_float)]


# name -> (restype, argtypes): every symbol include/swirld_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "sw_version": (C.c_int, []),
    "sw_create": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "sw_destroy": (C.c_int, [_P]),
    "sw_last_error": (C.c_char_p, [_P]),
    "sw_reserve": (C.c_int, [_P, C.c_int64]),
    "sw_append_events": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "sw_num_events": (C.c_int64, [_P]),
    "sw_divide_rounds": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "sw_decide_fame": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "sw_decide_fame_partial": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "sw_commit_fame": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "sw_row_stride": (C.c_int, [_P]),
    "sw_get_tally_impl": (C.c_int, [_P]),
    "sw_cansee_range": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "sw_split_link": (C.c_int, [_P, C.c_int]),
    "sw_split_unlink": (C.c_int, [_P]),
    "sw_cansee_repair": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "sw_export_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "sw_import_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "sw_get_range_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sw_find_order": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "sw_get_height": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_get_round": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_get_can_see": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_max_round": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "sw_get_witnesses": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "sw_get_famous": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "sw_get_famous_events": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_get_consensus": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "sw_get_sees_mask": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_get_vote": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int8)]),
    "sw_get_known_heights": (C.c_int, [_P, C.c_int64, _P]),
    "sw_sync_diff": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.POINTER(C.c_int64)]),
    "sw_get_chain_events": (C.c_int, [_P, C.c_int, C.c_int32, C.c_int32, _P]),
    "sw_crypto_verify_batch": (C.c_int, [C.c_int, C.c_int64, _P, _P, _P, _P, _P]),
    "sw_crypto_hash_batch": (C.c_int, [C.c_int, C.c_int64, _P, _P, _P]),
    "sw_num_ordered": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sw_get_transactions": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "sw_get_counters": (C.c_int, [_P, C.POINTER(Counters)]),
    "sw_get_counters_sized": (C.c_int, [_P, _P, C.c_size_t]),
    "sw_set_profiling": (C.c_int, [_P, C.c_int]),
    "sw_get_timings": (C.c_int, [_P, C.POINTER(Timings)]),
    "sw_debug_clocks": (C.c_int, [_P, _P, C.c_int64]),
    "sw_debug_block_clocks": (C.c_int, [_P, _P, C.c_int64]),
    "sw_set_forks": (C.c_int, [_P, C.c_int]),
    "sw_get_exact": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "sw_get_witness_order": (C.c_int, [_P, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "sw_set_window": (C.c_int, [_P, C.c_int, C.c_int]),
    "sw_get_window": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sw_set_window_lapse": (C.c_int, [_P, C.c_int64]),
    "sw_rewind": (C.c_int, [_P]),
    "sw_reset": (C.c_int, [_P]),
    "sw_synchronize": (C.c_int, [_P]),
    "sw_synth_hashgraph": (C.c_int, [C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_double,
                                     C.c_double, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libswirld_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: the HIP extension has not been built (run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`python py-swirld_amd/build.py`).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def hip_runtime_paths():
    """The HIP runtime libraries mapped into this process (paths of libamdhip64*).  PyTorch wheels carry their own copy with
    the SONAME of the system one: loaded FIRST, it also serves this library (one runtime, stream handles and events can be
    shared); loaded AFTER this library it becomes a SECOND runtime — its streams mean nothing to the C-ABI, and it may not even
    find the GPU once a windowed table has reserved its address range.  Import torch before this package when both are used."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return sorted(paths)


def require_single_hip_runtime(what):
    p = hip_runtime_paths()
    if len(p) > 1:
        raise RuntimeError("%s: two HIP runtimes are loaded in this process (%s) — torch was imported after py-swirld_amd; "
                           "import torch first, so that both use one runtime and stream handles can cross the C-ABI" % (what, ", ".join(p)))

