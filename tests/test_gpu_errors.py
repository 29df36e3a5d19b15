"""GPU (-m gpu): error behaviour and edge cases of the C-ABI ("Error
behaviour"): every misuse returns an errno-style code with a message, never crashes, and
leaves the context usable."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def code(pkg, fn, *a):
    with pytest.raises(pkg.SwirldHipError) as ei:
        fn(*a)
    assert str(ei.value)
    return ei.value.code


def test_argument_errors(pkg):
    assert code(pkg, pkg.Hashgraph, 0) == -22
    assert code(pkg, pkg.Hashgraph, 2000) == -22
    assert code(pkg, pkg.Hashgraph, 4, [1, 1, 1, 2 ** 31]) == -75
    h = pkg.Hashgraph(3)
    assert code(pkg, h.decide_fame) == -22                       # max() of an empty dict
    assert code(pkg, h.append_events, [5], [-1], [-1]) == -22     # creator out of range
    assert code(pkg, h.append_events, [0], [0], [-1]) == -22      # one parent only
    assert code(pkg, h.append_events, [0], [3], [4]) == -22       # parents not earlier
    h.append_events([0, 1, 2], [-1, -1, -1], [-1, -1, -1])
    assert code(pkg, h.append_events, [0], [1], [2]) == -22       # self-parent by another member
    assert code(pkg, h.append_events, [0], [0], [0]) == -22       # other-parent by the same member
    assert h.num_events == 3
    assert code(pkg, h.divide_rounds, 1, 2) == -22                # must continue at event 0
    assert code(pkg, h.divide_rounds, 0, 9) == -34
    assert code(pkg, h.rounds, 0, 1) == -34                       # nothing divided yet
    h.divide_rounds(0, 3)
    h.divide_rounds(3, 0)                                         # empty batch is a no-op
    assert list(h.rounds()) == [0, 0, 0]
    assert list(h.decide_fame()) == []
    assert code(pkg, h.find_order, [0]) == -22                    # round 0 undecided (KeyError)
    assert code(pkg, h.find_order, [7]) == -34
    assert list(h.find_order([])) == []
    assert h.vote(0, 0, 0, 1) == -1
    # the context is still usable
    h.append_events([0, 1], [0, 1], [1, 2])
    h.divide_rounds(3, 2)
    assert list(h.rounds()) == [0, 0, 0, 0, 0]
    assert h.can_see(3, 1)[0].tolist() == [3, 1, -1]
    h.close()


def test_single_member_and_tiny_graphs(pkg):
    h = pkg.Hashgraph(1)
    h.append_events([0], [-1], [-1])
    h.divide_rounds(0, 1)
    assert h.max_round == 0 and list(h.rounds()) == [0] and h.witnesses().tolist() == [[0]]
    assert list(h.decide_fame()) == []
    h.close()
    # two members ping-pong: compare with the oracle
    from oracle.oracle import Oracle
    n, N = 2, 400
    cr = np.array([0, 1] + [i % 2 for i in range(N - 2)], np.int32)
    sp = np.array([-1, -1] + [max(i - 2 + 2 - 2, 0) for i in range(N - 2)], np.int32)
    sp[2:] = np.arange(0, N - 2)
    op = np.array([-1, -1] + [0] * (N - 2), np.int32)
    op[2:] = np.arange(1, N - 1)
    sig = np.random.default_rng(5).integers(0, 256, (N, 64), dtype=np.uint8)
    t = np.arange(N, dtype=np.float64)
    o, h = Oracle(n), pkg.Hashgraph(n)
    for d in (o, h):
        d.append_events(cr, sp, op, t, sig)
        d.divide_rounds(0, N)
    nco, nch = o.decide_fame(), h.decide_fame()
    assert list(nco) == list(nch)
    assert np.array_equal(h.rounds(), o.round) and np.array_equal(h.can_see(), o.can_see)
    assert list(h.find_order(nch)) == list(o.find_order(nco))
    h.close()


def test_rewind_and_reuse(pkg):
    from oracle.oracle import Oracle
    n, N = 20, 6000
    stream = pkg.synth_hashgraph(n, N, 77)
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    nco = list(o.decide_fame())
    h = pkg.Hashgraph(n)
    h.append_events(*stream)
    for _ in range(3):
        h.divide_rounds(0, N)
        assert list(h.decide_fame()) == nco
        assert np.array_equal(h.rounds(), o.round)
        h.rewind()
        assert h.max_round == -1
    h.close()


@pytest.mark.parametrize("name,value", [("SW_TALLY_K", "0"), ("SW_TALLY_K", "abc"), ("SW_CANSEE_IMPL", "5"), ("SW_CANSEE_IMPL", "7"),
                                        ("SW_SKIP", "33"), ("SW_PIPE", "2x"), ("SW_CHUNKS", "9"), ("SW_GRAPH", "2"), ("SW_HALO", "-1"),
                                        ("SW_ELECT_CG", "96"), ("SW_MID_PCT", "100"), ("SW_FIN_BLOCKS", "8"), ("SW_FIN_BAND", "2"), ("SW_GRAPH_BIG", "513"),
                                        ("SW_BAND_FAST", "2"), ("SW_CHUNK_CFG", "3"), ("SW_TALLY_FILTER", "2"), ("SW_SHOT_PCT", "5"), ("SW_TALLY_IMPL", "3")])
def test_tuning_knobs_are_validated(pkg, monkeypatch, name, value):
    """the SW_* switches were read with atoi and silently clamped.  A value that is not an
    integer inside the documented range now fails sw_create with SW_EINVAL naming the variable."""
    monkeypatch.setenv(name, value)
    with pytest.raises(pkg.SwirldHipError) as ei:
        pkg.Hashgraph(4)
    assert ei.value.code == -22 and name in str(ei.value)
    monkeypatch.delenv(name)
    pkg.Hashgraph(4).close()
