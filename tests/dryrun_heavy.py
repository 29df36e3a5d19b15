"""CPU dry run of the HEAVY GPU parity tests (tests/test_gpu_baseline_configs.py): their LOGIC —
oracle pool, comparisons, prefix arguments — executed at tiny sizes with the CPU oracle standing
in for the device (tests/oracle_backend.py), so that a mistake in the test code does not cost a
GPU run.  Usage: python tests/dryrun_heavy.py"""
import importlib
import os
import sys

os.environ["SW_DRYRUN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pytest  # noqa: E402

import oracle_backend  # noqa: E402

pkg = importlib.import_module("py-swirld_amd")
pkg.Hashgraph = oracle_backend.OracleHashgraph
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_baseline_configs.py"),
                      "-p", "no:cacheprovider"] + sys.argv[1:]))
