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


@pytest.mark.parametrize("seed", range(16))
def test_random_forked_streams(pkg, seed):
    """The same comparison on FORKED hashgraphs (the reference stores forks, README.md:84): height-based
    maxi with ties, witnesses replaced by fork siblings between calls, fame keyed by event, and — with
    them — the exact path's host builds next to the oracle."""
    from test_exact_host import ExactHost, add_forks
    rng = np.random.default_rng(7000 + seed)
    n, N, mode, p0, p1, stake, chunk = random_case(rng)
    n, N = max(n, 3), max(N, 6 * n)
    base = pkg.synth_hashgraph(n, N, 9000 + seed, mode, p0, p1)
    cr, sp, op, t, sig = add_forks(base, n, seed, int(rng.integers(2, 30)))
    N = len(cr)
    ref, orc = RefRun(n, stake), Oracle(n, None if stake is None else stake.astype(np.uint64))
    exh = ExactHost(n, None if stake is None else stake.astype(np.uint32), lanes=bool(seed % 2) and (chunk is None or chunk > 30))
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        ref.append(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        ref.divide_rounds(a, b - a)
        for d in (orc, exh):
            d.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
            d.divide_rounds(a, b - a)
        nc_ref = list(ref.decide_fame())
        assert nc_ref == list((nc := orc.decide_fame())) == list(exh.decide_fame())
        try:
            tx_ref = ref.find_order(nc)
        except IndexError:  # swirld.py:305
            with pytest.raises(OracleError):
                orc.find_order(nc)
            with pytest.raises(IndexError):
                exh.find_order(nc)
            return
        assert list(tx_ref) == list(orc.find_order(nc)) == list(exh.find_order(nc))
    ex = ref.extract()
    st = exh.state()
    assert np.array_equal(ex["round"], orc.round) and np.array_equal(ex["round"], st["round"])
    assert np.array_equal(ex["can_see"], orc.can_see) and np.array_equal(ex["can_see"], st["can_see"])
    assert np.array_equal(ex["witnesses"], orc.witnesses()) and np.array_equal(ex["witnesses"], st["wit"])
    for r, order in enumerate(ex["wit_order"]):
        assert np.array_equal(order, orc.witness_order(r)) and np.array_equal(order, st["worder"][r])
    assert np.array_equal(ex["famous"], orc.famous_by_event) and np.array_equal(ex["famous"], st["fam"])
    assert np.array_equal(ex["consensus"], orc.consensus()) and np.array_equal(ex["consensus"], st["cons"])
    assert np.array_equal(ex["transactions"], orc.transactions)
    assert np.array_equal(ex["tbd"], orc.tbd) and np.array_equal(ex["tbd"], st["tbd"])
    assert len(ex["votes"]) == orc.num_votes
