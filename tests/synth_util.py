"""Test helper: numpy front-end of the library's host-side synthetic generator
(sw_synth_hashgraph, py-swirld_amd/csrc/synth.cpp)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def synth(n, N, seed, mode=0, p0=0.0, p1=0.0):
    pkg = importlib.import_module("py-swirld_amd")
    return pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
