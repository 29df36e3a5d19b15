"""GPU (-m gpu): ONE hashgraph's round loop split inside its iterations over several linked contexts (include/swirld_hip.h
part 3, SURVEY.md §8e last bullet; swirld.py:208-216 once per candidate event, dealt to the parts).  P contexts on ONE device
stand for P GPUs: every part builds the band masks of its share of the band events and tallies its share of the members,
stores what it produces into the tables of every part, and meets the others at both kernel boundaries of every iteration.
Every part must end with exactly the state the reference algorithm (oracle) gives — rounds, witnesses, can_see, fame,
consensus, total order — for batch and incremental call schedules, uniform and skewed hashgraphs, 1 to 16 mask words."""
import numpy as np
import pytest

from oracle_pool import compare_state

pytestmark = pytest.mark.gpu


def run_split(pkg, n, stream, parts, chunk=None):
    cr, sp, op, t, sig = stream
    N = len(cr)
    hs = [pkg.Hashgraph(n) for _ in range(parts)]
    for h in hs:
        h.reserve(N)
    ncs = []
    linked = False
    chunk = chunk or N
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        for h in hs:
            h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
        if not linked:
            pkg.Hashgraph.split_link(hs)
            linked = True
        pkg.Hashgraph.split_divide_rounds(hs, a, b - a)
        nc = [[int(r) for r in h.decide_fame()] for h in hs]
        assert all(x == nc[0] for x in nc), "every part decides the same rounds"
        ncs.append(nc[0])
    return hs, ncs


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,parts,chunk", [
    (8, 6000, 11, 0, 0, 0, 2, None), (64, 40000, 12, 0, 0, 0, 3, None), (64, 30000, 13, 2, 0.3, 0.03, 2, 7000),
    (130, 30000, 14, 0, 0, 0, 4, None), (256, 60000, 15, 0, 0, 0, 8, None), (256, 50000, 16, 1, 0.5, 0.02, 2, 17000),
    (400, 40000, 17, 0, 0, 0, 3, None),
])
def test_split_round_loop_matches_oracle(pkg, n, N, seed, mode, p0, p1, parts, chunk):
    from oracle.oracle import Oracle
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    o = Oracle(n)
    onc = []
    c = chunk or N
    for a in range(0, N, c):
        b = min(N, a + c)
        o.append_events(*[x[a:b] for x in stream])
        o.divide_rounds(a, b - a)
        onc.append([int(r) for r in o.decide_fame()])
    hs, ncs = run_split(pkg, n, stream, parts, chunk)
    assert ncs == onc
    assert o.max_round >= 3, "the case must span several rounds"
    for h in hs:
        compare_state(h, o, N, can_see_step=20_000)
        assert h.counters()["round_iterations"] == hs[0].counters()["round_iterations"]
    if chunk is None:   # the total order from one of the parts (find_order is per context)
        assert np.array_equal(hs[-1].find_order(ncs[0]), o.find_order(onc[0]))
    for h in hs:
        h.close()


def test_split_equals_an_unlinked_context_after_rewind(pkg):
    """the same contexts, first linked, then unlinked and rewound: both passes leave the same state"""
    n, N = 200, 50000
    stream = pkg.synth_hashgraph(n, N, 31)
    hs, ncs = run_split(pkg, n, stream, 2)
    r_split, w_split, f_split = hs[0].rounds(), hs[0].witnesses(), hs[0].famous()
    hs[0].split_unlink()
    for h in hs:
        h.rewind()
        h.divide_rounds(0, N)
        assert [int(r) for r in h.decide_fame()] == ncs[0]
        assert np.array_equal(h.rounds(), r_split) and np.array_equal(h.witnesses(), w_split) and np.array_equal(h.famous(), f_split)
        h.close()


@pytest.mark.heavy("n1024_uniform")
@pytest.mark.parametrize("parts", [2, 4])
def test_split_round_loop_1024_members(pkg, oracle_pool, parts):
    """BASELINE.json configs[4]'s member count: the shape the split is for (an iteration is ~105 us on one GPU there,
    ~90 us of it parallel over band events / members)"""
    run = oracle_pool.get("n1024_uniform")
    o = run.oracle
    hs, ncs = run_split(pkg, run.n, run.stream, parts)
    assert ncs == run.new_c
    for h in hs:
        compare_state(h, o, run.N, can_see_step=50_000)
        h.close()


def test_split_rewind_and_divide_again(pkg):
    """linked contexts rewound together divide again to the same state (the measured step of bench.py --split strong)"""
    from oracle.oracle import Oracle
    n, N = 96, 30000
    stream = pkg.synth_hashgraph(n, N, 41, 2, 0.3, 0.05)
    o = Oracle(n)
    o.append_events(*stream)
    o.divide_rounds(0, N)
    onc = [int(r) for r in o.decide_fame()]
    hs, ncs = run_split(pkg, n, stream, 3)
    for _ in range(2):
        pkg.Hashgraph.split_rewind(hs)
        pkg.Hashgraph.split_divide_rounds(hs, 0, N)
        for h in hs:
            assert [int(r) for r in h.decide_fame()] == onc
    for h in hs:
        compare_state(h, o, N, can_see_step=15_000)
        h.close()


def test_split_link_refuses_what_it_cannot_do(pkg):
    a, b = pkg.Hashgraph(16), pkg.Hashgraph(16)
    st = pkg.synth_hashgraph(16, 2000, 5)
    a.append_events(*st)
    with pytest.raises(pkg.SwirldHipError) as ei:      # not the same hashgraph
        pkg.Hashgraph.split_link([a, b])
    assert ei.value.code == -22
    with pytest.raises(pkg.SwirldHipError):            # one context twice
        pkg.Hashgraph.split_link([a, a])
    b.append_events(*st)
    w1, w2 = pkg.Hashgraph(4, [1, 2, 1, 1]), pkg.Hashgraph(4, [1, 2, 1, 1])
    with pytest.raises(pkg.SwirldHipError) as ei:      # weighted stake: the split kernels are the unit-stake tally
        pkg.Hashgraph.split_link([w1, w2])
    assert ei.value.code == -95
    pkg.Hashgraph.split_link([a, b])
    with pytest.raises(pkg.SwirldHipError):            # linked already
        pkg.Hashgraph.split_link([a, b])
    pkg.Hashgraph.split_divide_rounds([a, b], 0, 2000)
    assert np.array_equal(a.rounds(), b.rounds())
    a.close()                                          # closing one part dissolves the group: the other works on alone
    b.rewind()
    b.divide_rounds(0, 2000)
    for h in (b, w1, w2):
        h.close()
