import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "heavy: GPU parity case whose CPU oracle run takes about a minute "
                                       "(computed in a background thread, see tests/oracle_pool.py)")


_POOL = None


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    # heavy cases last: their oracle runs overlap everything that comes before them
    items.sort(key=lambda it: 1 if it.get_closest_marker("heavy") else 0)


def pytest_collection_finish(session):
    global _POOL
    names = []
    for it in session.items:
        mk = it.get_closest_marker("heavy")
        if mk is not None:
            names.extend(a for a in mk.args if a not in names)
    # torch ships its own copy of the HIP runtime under the system library's SONAME: imported BEFORE py-swirld_amd's library it
    # serves both (one runtime: torch streams can cross the C-ABI); imported after, it is a second runtime — and reports "No HIP
    # GPUs are available" once a windowed context has reserved its address range (profiles/r04r_pytest.log).  So: torch first,
    # whatever the order of the files.  Only when tests that use torch on the GPU were selected.
    uses_torch = any(it.get_closest_marker("gpu") is not None and
                     os.path.basename(str(it.fspath)) in ("test_gpu_strong_split.py", "test_gpu_partition.py") for it in session.items)
    if uses_torch and not session.config.option.collectonly:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # (the tests themselves report what is missing)
            pass
    # (the oracle pool AFTER the torch import: its threads call the library's host-side generator at once, which loads
    # libswirld_hip.so — and with it the system HIP runtime — while this function is still running; with torch imported behind
    # that, the process had two runtimes and the first sw_create of the session found no device: profiles/r05o_pytest.log)
    if names and _POOL is None and not session.config.option.collectonly:
        from oracle_pool import OraclePool
        _POOL = OraclePool(names)


def pytest_sessionfinish(session, exitstatus):
    global _POOL
    if _POOL is not None:
        _POOL.shutdown()
        _POOL = None


@pytest.fixture(scope="session")
def oracle_pool():
    global _POOL
    if _POOL is None:
        from oracle_pool import OraclePool
        _POOL = OraclePool([])
    return _POOL


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["n"] = int(g["n"])
    g["chunk"] = int(g["chunk"])
    N = len(g["creator"])
    if "sched" in g:  # variable batch sizes (the reference's own main loop)
        ends = np.cumsum(g["sched"])
        g["batches"] = [(int(e - k), int(e)) for k, e in zip(g["sched"], ends)]
    else:
        g["batches"] = [(a, min(N, a + g["chunk"])) for a in range(0, N, g["chunk"])]
    off = g["wit_order_off"]
    g["wit_order"] = [g["wit_order_flat"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    return g


@pytest.fixture(scope="session")
def pkg():
    import importlib
    build = importlib.import_module("py-swirld_amd.build")
    build.build()
    return importlib.import_module("py-swirld_amd")
