"""CPU, authoring container only (skipped where /root/reference is absent): randomized
differential test of the oracle against the UNMODIFIED reference — random member counts, stakes,
generator modes and call schedules.  The committed goldens are the portable
subset of exactly this comparison."""
import numpy as np
import pytest

from refharness import RefRun, have_reference
from oracle.oracle import Oracle, OracleError

pytestmark = pytest.mark.skipif(not have_reference(), reason="the reference tree is not present on this machine")


def random_case(rng):
    n = int(rng.integers(2, 28))
    N = int(rng.integers(n, 900))
    mode = int(rng.integers(0, 4))
    p0 = float(rng.uniform(0.01, 0.6))
    p1 = float(rng.uniform(0.005, 0.2))
    stake = None
    if rng.random() < 0.3:  # near-unit stakes still make progress (Appendix A Q2)
        stake = np.ones(n, np.int64)
        stake[rng.integers(0, n, size=max(1, n // 8))] = 2
    chunk = None if rng.random() < 0.5 else int(rng.integers(1, max(2, N // 3)))
    return n, N, mode, p0, p1, stake, chunk


@pytest.mark.parametrize("seed", range(24))
def test_random_streams(pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    n, N, mode, p0, p1, stake, chunk = random_case(rng)
    cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 5000 + seed, mode, p0, p1)
    t = t + rng.integers(0, 3, N) * 0.5  # ties and inversions in the timestamps
    ref, orc = RefRun(n, stake), Oracle(n, None if stake is None else stake.astype(np.uint64))
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        ref.append(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        orc.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        ref.divide_rounds(a, b - a)
        orc.divide_rounds(a, b - a)
        assert list(ref.decide_fame()) == list((nc := orc.decide_fame()))
        try:
            tx_ref = ref.find_order(nc)
        except IndexError:  # swirld.py:305, single seeing witness (unequal stakes only)
            with pytest.raises(OracleError):
                orc.find_order(nc)
            return
        assert list(tx_ref) == list(orc.find_order(nc))
    ex = ref.extract()
    assert np.array_equal(ex["round"], orc.round)
    assert np.array_equal(ex["can_see"], orc.can_see)
    assert np.array_equal(ex["witnesses"], orc.witnesses())
    assert np.array_equal(ex["famous"], orc.famous_by_event)
    assert np.array_equal(ex["consensus"], orc.consensus())
    assert np.array_equal(ex["transactions"], orc.transactions)
    assert len(ex["votes"]) == orc.num_votes
