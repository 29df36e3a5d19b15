import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("py-swirld_amd")
def run(n, N, mode, p0, bulk, rounds=None):
    os.environ["SW_ORDER_BULK"] = bulk
    st = pkg.synth_hashgraph(n, N, 83, mode, p0, 0.0)
    h = pkg.Hashgraph(n); h.append_events(*st); h.divide_rounds(0, N); nc = list(h.decide_fame())
    if rounds: nc = nc[:rounds]
    got = np.array(h.find_order(nc)); h.close()
    return nc, got
for n in (512, 1024):
    for N in (30000, 100000):
        for p0 in (0.02, 0.5):
            for rounds in (2, None):
                nc, a = run(n, N, 1, p0, "1", rounds)
                _, b = run(n, N, 1, p0, "0", rounds)
                _, b2 = run(n, N, 1, p0, "0", rounds)
                print(n, N, p0, "rounds", len(nc), "ordered", len(a), "small==bulk", np.array_equal(a, b), "small deterministic", np.array_equal(b, b2), flush=True)
