import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("py-swirld_amd")
from oracle.oracle import Oracle
n, N = 1024, 100_000
st = pkg.synth_hashgraph(n, N, 83, 1, 0.02, 0.0)
o = Oracle(n); o.append_events(*st); o.divide_rounds(0, N); nco = list(o.decide_fame()); exp = np.array(o.find_order(nco))
print("oracle new_c", nco, len(exp))
for env in ({}, {"SW_ORDER_BIG_HOST": "1"}, {"SW_ORDER_HOST": "1"}, {"SW_ORDER_BULK": "1"}, {}):
    for k in ("SW_ORDER_BIG_HOST", "SW_ORDER_HOST", "SW_ORDER_BULK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    h = pkg.Hashgraph(n); h.append_events(*st); h.divide_rounds(0, N); nc = list(h.decide_fame())
    got = np.array(h.find_order(nc))
    ok = np.array_equal(got, exp)
    nbad = int((got != exp).sum()) if len(got) == len(exp) else -1
    print(env, "equal" if ok else "DIFFERENT (%d positions), first %s vs %s" % (nbad, got[:5], exp[:5]), "host-sorted rounds", h.counters()["order_rounds_host_sorted"])
    h.close()
