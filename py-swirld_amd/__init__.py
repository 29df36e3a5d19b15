"""py-swirld_amd — MI355X-native virtual-voting hot path of py-swirld
(Node.divide_rounds / decide_fame / find_order, swirld.py:187-311) behind the
reference's own Node/Event API.  Python host code -> ctypes C-ABI
(include/swirld_hip.h) -> hand-written HIP kernels for gfx950.

The directory name contains a hyphen (it follows the reference repo's name); import it
with `importlib.import_module("py-swirld_amd")` or through the alias module
`swirld_amd` at the repo root.
"""
from ._lib import SwirldHipError, LIB_PATH  # noqa: F401
from .engine import Hashgraph, hash_batch, synth_hashgraph, verify_batch  # noqa: F401
from .node import C, Event, Node, VotesUnavailable, majority, test  # noqa: F401

__all__ = ["Hashgraph", "synth_hashgraph", "verify_batch", "hash_batch", "SwirldHipError", "LIB_PATH",
           "Node", "Event", "C", "majority", "test", "VotesUnavailable"]
