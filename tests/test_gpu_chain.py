"""GPU (-m gpu): chained round loops (round 5, DESIGN.md §4 "Pipelining").  A large sw_divide_rounds call runs one round loop per
sub-batch; with SW_CHAIN=1 (default) the loop of sub-batch i + 1 — k_loop_init in its chained form, which finds its start round
on the device — is enqueued behind the first shot of loop i, and the host reads loop i's state one loop late.  Checked against
the oracle, and against the loop-by-loop path (SW_CHAIN=0), with first shots that are too short (SW_SHOT_PCT=50: the chained
start REFUSES, the iterations enqueued for the next loop go on with the old one, the host starts the next loop again), too
long (no-op iterations), several cut schedules, bridge shots of 2 ... 512 iterations (SW_BRIDGE: a chained start is followed by a
short shot, the rest of the prediction once the previous loop's state was read), members whose first event arrives late (a sub-batch that holds a root is
never chained), and a second large call on the same context."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def late_joiners(n, N, seed, late_frac=0.5, join_at=0.45):
    """A fork-free stream in which the last `late_frac` of the members create their first event only after `join_at` of the
    events: roots in the middle of a large call."""
    rng = np.random.default_rng(seed)
    early = max(2, int(n * (1 - late_frac)))
    cr = np.empty(N, np.int32); sp = np.full(N, -1, np.int32); op = np.full(N, -1, np.int32)
    head = np.full(n, -1, np.int64)
    active = list(range(early))
    i = 0
    for m in range(early):
        cr[i] = m; head[m] = i; i += 1
    join_from = int(N * join_at)
    nxt = early
    while i < N:
        if nxt < n and i >= join_from and rng.random() < 0.02:
            cr[i] = nxt; head[nxt] = i; active.append(nxt); nxt += 1; i += 1   # a root
            continue
        a = active[int(rng.integers(len(active)))]
        b = a
        while b == a:
            b = active[int(rng.integers(len(active)))]
        cr[i] = a; sp[i] = head[a]; op[i] = head[b]; head[a] = i; i += 1
    t = np.arange(N, dtype=np.float64)
    sig = rng.integers(0, 256, size=(N, 64), dtype=np.uint8)
    return cr, sp, op, t, sig


def run_both(pkg, n, stream, calls):
    from oracle.oracle import Oracle
    o, h = Oracle(n), pkg.Hashgraph(n)
    N = len(stream[0])
    edges = [int(N * f) for f in calls] + [N]
    a = 0
    for b in edges:
        if b <= a:
            continue
        for d in (o, h):
            d.append_events(*[x[a:b] for x in stream])
            d.divide_rounds(a, b - a)
        assert list(o.decide_fame()) == list(h.decide_fame())
        a = b
    assert np.array_equal(h.rounds(), o.round)
    wit = h.witnesses()
    assert np.array_equal(wit, o.witnesses())
    m = wit >= 0
    assert np.array_equal(h.famous()[m], o.famous_by_event[wit[m]])
    assert np.array_equal(h.consensus(), o.consensus())
    assert np.array_equal(h.can_see(), o.can_see)
    c = h.counters()
    h.close()
    return c


CASES = [
    # n, N, seed, mode, p0, p1, calls (fractions where a new call starts)
    (64, 140000, 2, 0, 0.0, 0.0, [0.5]),            # two large calls on one context (exhaustion marks carried over)
    (130, 90000, 3, 2, 0.3, 0.02, []),              # slow members: a loop that re-enters old rounds
    (20, 70000, 4, 1, 0.02, 0.0, []),               # two cliques
    (256, 80000, 5, 0, 0.0, 0.0, []),
]


@pytest.mark.parametrize("n,N,seed,mode,p0,p1,calls", CASES)
@pytest.mark.parametrize("env", [{}, {"SW_SHOT_PCT": "50"}, {"SW_SHOT_PCT": "300", "SW_PIPE": "8"},
                                 {"SW_CUTS": "0.01;0.03;0.1;0.3;0.6", "SW_SHOT_PCT": "70"}, {"SW_GRAPH": "0", "SW_SHOT_PCT": "50"},
                                 {"SW_BRIDGE": "2"}, {"SW_BRIDGE": "4", "SW_SHOT_PCT": "50"}, {"SW_BRIDGE": "512"}])
def test_chained_loops_match_the_oracle_and_the_loop_by_loop_path(pkg, monkeypatch, n, N, seed, mode, p0, p1, calls, env):
    stream = pkg.synth_hashgraph(n, N, seed, mode, p0, p1)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SW_CHAIN", "1")
    c1 = run_both(pkg, n, stream, calls)
    monkeypatch.setenv("SW_CHAIN", "0")
    c0 = run_both(pkg, n, stream, calls)
    # the iterations a loop EXECUTES do not depend on how they were enqueued
    assert c1["round_iterations"] == c0["round_iterations"]
    assert c1["rounds"] == c0["rounds"]


@pytest.mark.parametrize("env", [{}, {"SW_SHOT_PCT": "50", "SW_PIPE": "8"}])
def test_a_sub_batch_that_holds_a_root_is_started_by_the_host(pkg, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SW_CHAIN", "1")
    for n, N, seed in [(40, 90000, 11), (100, 120000, 12)]:
        stream = late_joiners(n, N, seed)
        run_both(pkg, n, stream, [])
        run_both(pkg, n, stream, [0.3])
