import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("py-swirld_amd")
n, N, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cr, sp, op, t, sig = pkg.synth_hashgraph(n, N, 82, 0, 0.0, 0.0)
h = pkg.Hashgraph(n)
for a in range(0, N, chunk):
    b = min(N, a + chunk)
    h.append_events(cr[a:b], sp[a:b], op[a:b], t[a:b], sig[a:b])
    h.divide_rounds(a, b - a)
    nc = list(h.decide_fame())
    print("call", a, "new_c", nc, flush=True)
    os.environ["SW_DEBUG_TIMING"] = "1"
    tx = h.find_order(nc)
    print("   ordered", len(tx), flush=True)
