"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/swirld_hip.h declares; host-only entry points work; creating a context without a
GPU fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "swirld_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    import ctypes
    lib = ctypes.CDLL(pkg.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "%s declared in swirld_hip.h but not exported" % s
    import importlib
    L = importlib.import_module("py-swirld_amd._lib")
    assert sorted(L.SIGNATURES) == syms, "ctypes table out of sync with the header"
    assert lib.sw_version() == 7


def test_integration_stub_matches_the_abi(pkg):
    """The ctypes binding shown in INTEGRATION.md (what a maintainer of the reference would add)
    is executed against the real library: every entry point it binds exists, and its argument
    lists have the arity of the header prototypes and of the package's own ctypes table."""
    import ctypes as ct
    import importlib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("# --- add to swirld.py"):]
    block = block[:block.index("```")]
    lines = [ln for ln in block.splitlines() if re.match(r"_sw\.sw_[a-z_]+\.(argtypes|restype)\s*=", ln) or ln.startswith("_P =")]
    assert len(lines) >= 9
    env = {"ct": ct, "_sw": ct.CDLL(pkg.LIB_PATH)}
    exec("\n".join(lines), env)
    L = importlib.import_module("py-swirld_amd._lib")
    header = open(os.path.join(ROOT, "include", "swirld_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for ln in lines:
        m = re.match(r"_sw\.(sw_[a-z_]+)\.argtypes", ln)
        if not m:
            continue
        fn = m.group(1)
        bound = getattr(env["_sw"], fn).argtypes
        proto = re.search(r"\b%s\s*\(([^)]*)\)" % fn, header).group(1)
        assert len(bound) == len([a for a in proto.split(",") if a.strip()]), fn
        assert len(bound) == len(L.SIGNATURES[fn][1]), fn
    for fn in re.findall(r"_sw\.(sw_[a-z_]+)\(", block):  # every call made by the stub is a real entry point
        assert fn in L.SIGNATURES, fn


def test_synth_is_a_valid_forkfree_dag(pkg):
    for mode, p0, p1 in [(0, 0, 0), (1, 0.05, 0), (2, 0.25, 0.05), (3, 0.5, 0)]:
        n, N = 12, 3000
        cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 7, mode, p0, p1)
        assert (cr[:n] == np.arange(n)).all() and (sp[:n] == -1).all() and (op[:n] == -1).all()
        idx = np.arange(n, N)
        assert (sp[n:] < idx).all() and (op[n:] < idx).all() and (sp[n:] >= 0).all()
        assert (cr[sp[n:]] == cr[n:]).all() and (cr[op[n:]] != cr[n:]).all()
        head = {}
        for e in range(N):  # each member's events form one self-parent chain
            assert head.get(cr[e], -1) == sp[e]
            head[cr[e]] = e
        cr2 = pkg.synth_hashgraph(n, N, 7, mode, p0, p1)[0]
        assert (cr == cr2).all()
        assert sig.shape == (N, 64) and (t == np.arange(N)).all()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback(pkg):
    with pytest.raises(pkg.SwirldHipError) as ei:
        pkg.Hashgraph(4)
    assert ei.value.code == -19  # SW_ENODEV


def test_stakes_are_validated_before_the_device_is_touched(pkg):
    """Non-integer / negative stakes are refused (the device tallies are integer), not truncated."""
    for bad in ([1, 2.5, 1, 1], [1, -1, 1, 1], [1, float("nan"), 1, 1], [1, 1, 1]):
        with pytest.raises(ValueError):
            pkg.Hashgraph(4, bad)


def test_committed_measurement_fixtures_bench_reads():
    """bench.py copies two committed measurements into its JSON line: the per-kernel HBM traffic of a
    same-code rocprofv3 --pmc run and the timing of the unmodified Python reference (authoring container)."""
    import json
    import os
    from conftest import ROOT
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        tr = json.load(f)
    # keyed by workload (the 256 x 1 M figures were quoted for every workload): the default workload of
    # bench.py must be there, with the sweep's and the loop kernels' bytes and rocprofv3's own launch durations
    w = tr["workloads"]["256x1000000x0"]
    assert w["commit"] and w["kernels"]["k_cansee_chunks"] > 0 and w["kernels"]["k_tally_tree"] > 0 and w["avg_us"]["k_cansee_chunks"] > 0
    # the counter passes must have been taken on THIS tree's kernels: the file carries the SHA-256 of
    # csrc/kernels.hip.h + csrc/order.hip.h it was measured on; bench.py quotes nothing from a stale file, and this test says so loudly
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_sha", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert tr.get("kernels_sha256") == bench.kernels_sha256(), ("profiles/traffic.json was measured on other kernel source (commit %s): "
                                                               "run profiles/run_profiles.sh on the GPU box and copy its traffic.json" % w["commit"])
    assert bench.load_traffic("256x1000000x0")[0] and not bench.load_traffic("1024x123x0")[0], "a workload without a counter pass gets no figures"
    with open(os.path.join(ROOT, "profiles", "reference_python_timing.json")) as f:
        rp = json.load(f)
    assert rp["reference_equals_oracle_on_this_prefix"] is True and rp["members"] == 256 and rp["events_per_s"] > 0


def test_ctypes_structs_mirror_the_header_structs(pkg):
    """sw_counters / sw_timings are filled by value through a caller-provided pointer: the ctypes mirrors in
    py-swirld_amd/_lib.py must have the header's fields, in the header's order, with the header's types —
    a field appended on one side only reads (or overruns) the wrong bytes silently."""
    import ctypes as C
    import importlib
    L = importlib.import_module("py-swirld_amd._lib")
    header = open(os.path.join(ROOT, "include", "swirld_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    ctype = {"int64_t": C.c_int64, "int32_t": C.c_int32, "float": C.c_float}
    for name, mirror in (("sw_counters", L.Counters), ("sw_timings", L.Timings)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, flags=re.S).group(1)
        fields = [(m.group(2), ctype[m.group(1)]) for m in re.finditer(r"\b(int64_t|int32_t|float)\s+([a-z_0-9]+)\s*;", body)]
        assert fields == list(mirror._fields_), name
        assert C.sizeof(mirror) == sum(C.sizeof(t) for _, t in fields), name


def test_hip_runtime_listing(pkg):
    """_lib.hip_runtime_paths(): at least the runtime the library itself is linked against; the guard used where torch
    streams cross the C-ABI names both copies when there are two."""
    import importlib
    L = importlib.import_module("py-swirld_amd._lib")
    L.load()
    paths = L.hip_runtime_paths()
    assert len(paths) >= 1 and all("libamdhip64" in p for p in paths)
    if len(paths) > 1:
        with pytest.raises(RuntimeError) as ei:
            L.require_single_hip_runtime("test")
        assert all(p in str(ei.value) for p in paths)
    else:
        L.require_single_hip_runtime("test")
