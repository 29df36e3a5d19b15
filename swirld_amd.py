"""Importable alias of the package directory `py-swirld_amd/` (whose name, taken from the
reference repository, is not a valid Python identifier)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("py-swirld_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
__all__ = list(_pkg.__all__)
