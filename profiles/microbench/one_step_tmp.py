import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("py-swirld_amd")
n, N = 256, 1000000
st = pkg.synth_hashgraph(n, N, 3)
h = pkg.Hashgraph(n); h.reserve(N); h.append_events(*st)
for i in range(3):
    if i == 2: os.environ["SW_DEBUG_TIMING"] = "1"
    t0 = time.perf_counter(); h.divide_rounds(0, N); t1 = time.perf_counter(); h.decide_fame(); t2 = time.perf_counter()
    print("divide %.3f ms, fame %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3)); h.rewind()
