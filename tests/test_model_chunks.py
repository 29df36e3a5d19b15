"""CPU: the chunk-parallel can_see sweep (tests/model_chunks.py, the statement k_cansee_chunks
implements) against the sequential rows of swirld.py:198-205, 220 — uniform gossip, slow and silent
members, cliques, stale other-parents; halos from 0 (everything at a chunk start is provisional) to
ample; ranges that start in the middle of the hashgraph (zone i); the re-sweep fallback."""
import numpy as np
import pytest

from model_chunks import cansee_chunked, cansee_sequential
from synth_util import synth

CASES = [
    # n, N, seed, mode, p0, p1
    (4, 600, 1, 0, 0.0, 0.0),
    (16, 4000, 2, 0, 0.0, 0.0),
    (64, 12000, 3, 0, 0.0, 0.0),
    (33, 6000, 4, 1, 0.02, 0.0),    # two cliques
    (40, 8000, 5, 2, 0.4, 50.0),    # 40 % of the members 50x less active
    (24, 5000, 6, 3, 0.3, 0.0),     # stale other-parents
    (10, 3000, 7, 2, 0.5, 400.0),   # nearly silent members
]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1", CASES)
def test_chunked_rows_equal_sequential_rows(n, N, seed, mode, p0, p1):
    cr, sp, op = (np.asarray(x) for x in synth(n, N, seed, mode, p0, p1)[:3])
    ref = cansee_sequential(n, cr, sp, op)
    rng = np.random.default_rng(seed)
    for halo in (0, 3, 5 * n, 40 * n):
        for G in (2, 3, 5):
            for a0 in (0, N // 3):
                inner = np.sort(rng.choice(np.arange(a0 + 1, N), size=G - 1, replace=False))
                cuts = [a0] + [int(x) for x in inner] + [N]
                L = np.full((N, n), -1, np.int32)
                L[:a0] = ref[:a0]
                L, st = cansee_chunked(n, cr, sp, op, a0, cuts, halo, L)
                assert np.array_equal(L, ref), (halo, G, a0, cuts, st)
                L = np.full((N, n), -1, np.int32)
                L[:a0] = ref[:a0]
                L, st2 = cansee_chunked(n, cr, sp, op, a0, cuts, halo, L, resweep_limit=n)
                assert np.array_equal(L, ref), (halo, G, a0, cuts, st2)
                # the device's form: no sweep reads a row from memory (leaf boundary 0), the first chunk has a
                # halo like the others, the final rows below the range only serve the repairs
                L = np.full((N, n), -1, np.int32)
                L[:a0] = ref[:a0]
                L, st3 = cansee_chunked(n, cr, sp, op, 0, cuts, halo, L)
                assert np.array_equal(L, ref), (halo, G, a0, cuts, st3)


def test_uniform_gossip_needs_no_repair_with_an_ample_halo():
    """What makes the scheme pay: at uniform gossip an event reaches an in-window event of every member
    after a few thousand events, so with the halo the kernels use nothing is provisional."""
    n, N = 64, 30000
    cr, sp, op = (np.asarray(x) for x in synth(n, N, 11)[:3])
    L, st = cansee_chunked(n, cr, sp, op, 0, [0, 10000, 20000, N], 32 * n)
    assert np.array_equal(L, cansee_sequential(n, cr, sp, op))
    assert st["prov"] == [0, 0, 0]
    L, st = cansee_chunked(n, cr, sp, op, 0, [0, 10000, 20000, N], 0)
    assert st["prov"][1] > 0 and st["prov"][2] > 0      # without a halo every chunk start is provisional
