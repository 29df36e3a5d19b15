"""CPU: the oracle (oracle/swirld_oracle.c) replayed against every committed golden
fixture — outputs of the UNMODIFIED reference (tests/golden/make_golden.py).  This is
what pins the oracle; it covers unit and mixed stakes, coin rounds (n=4), skipped-round
witnesses, incremental call schedules (Q9/Q10), stale other-parents, slow members and a
forked DAG."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle.oracle import Oracle


def replay(driver, g, with_order=True):
    """Feeds fixture g to `driver` (Oracle-like API) with the recorded call schedule and
    checks the per-call return values of decide_fame / find_order."""
    calls = 0
    for a, b in g["batches"]:
        driver.append_events(g["creator"][a:b], g["self_parent"][a:b], g["other_parent"][a:b],
                             g["t"][a:b], g["sig"][a:b])
        driver.divide_rounds(a, b - a)
        nc = driver.decide_fame()
        exp_nc = g["new_c_flat"][g["new_c_off"][calls]:g["new_c_off"][calls + 1]]
        assert list(nc) == list(exp_nc), "new_c of call %d" % calls
        if with_order:
            tx = driver.find_order(nc)
            exp_tx = g["transactions"][g["tx_off"][calls]:g["tx_off"][calls + 1]]
            assert list(tx) == list(exp_tx), "find_order of call %d" % calls
        calls += 1


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name):
    g = load_golden(name)
    o = Oracle(g["n"], g["stake"])
    replay(o, g)
    assert np.array_equal(o.round, g["round"])
    assert np.array_equal(o.height, g["height"])
    assert np.array_equal(o.can_see, g["can_see"])
    assert np.array_equal(o.witnesses(), g["witnesses"])
    for r, order in enumerate(g["wit_order"]):
        assert np.array_equal(o.witness_order(r), order), "dict order of witnesses[%d]" % r
    assert np.array_equal(o.famous_by_event, g["famous"])
    assert np.array_equal(o.consensus(), g["consensus"])
    assert np.array_equal(o.transactions, g["transactions"])
    assert np.array_equal(o.tbd, g["tbd"])
    assert o.num_votes == len(g["votes"])
    for y, x, v in g["votes"]:
        assert o.vote(y, x) == v


def test_oracle_rejects_bad_input():
    from oracle.oracle import OracleError
    o = Oracle(3)
    with pytest.raises(OracleError):
        o.append_events([5], [-1], [-1])
    with pytest.raises(OracleError):
        o.append_events([0], [0], [-1])
