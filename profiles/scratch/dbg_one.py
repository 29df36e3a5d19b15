import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("py-swirld_amd")
n, N, mode, p0, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
st = pkg.synth_hashgraph(n, N, 83, mode, p0, 0.0)
h = pkg.Hashgraph(n); h.append_events(*st); h.divide_rounds(0, N); nc = list(h.decide_fame())
if rounds: nc = nc[:rounds]
print("rounds", nc, flush=True)
os.environ["SW_DEBUG_TIMING"] = "1"
got = np.array(h.find_order(nc))
print("ordered", len(got), flush=True)
h.close()
print("closed", flush=True)
