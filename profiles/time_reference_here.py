"""The UNMODIFIED reference (/root/reference/swirld.py, pure Python) timed in the authoring container on
a prefix of the very stream bench.py uses (256 members, seed 3), next to the C oracle on the same
prefix, with the two results compared — so that the `cpu_baseline` of the bench line (the C port,
which is what can travel to the GPU box) is tied to a measured figure of the reference itself.

Protocol of BASELINE.md §3: blank Node, add_event for every event (untimed), then one batch call of
divide_rounds + decide_fame (timed), stdout suppressed, 1 core.
Usage (authoring container only; needs /root/reference):  python profiles/time_reference_here.py [events]
Writes profiles/reference_python_timing.json (read by bench.py) and prints a summary."""
import importlib
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import refharness  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

pkg = importlib.import_module("py-swirld_amd")
N_BENCH, n, seed = 1_000_000, 256, 3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000
stream = pkg.synth_hashgraph(n, N_BENCH, seed)          # the bench stream; the prefix is what gets timed
pre = [a[:M] for a in stream]

ref = refharness.RefRun(n)
ref.append(*pre)
t0 = time.perf_counter()
ref.divide_rounds(0, M)
t1 = time.perf_counter()
nc_ref = ref.decide_fame()
t2 = time.perf_counter()

o = Oracle(n)
o.append_events(*pre)
c0 = time.perf_counter()
o.divide_rounds(0, M)
nc_o = list(o.decide_fame())
c1 = time.perf_counter()

ex = ref.extract()
same = (np.array_equal(ex["round"], o.round) and np.array_equal(ex["can_see"], o.can_see)
        and np.array_equal(ex["witnesses"], o.witnesses()) and np.array_equal(ex["famous"], o.famous_by_event)
        and list(nc_ref) == nc_o)
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
out = {
    "what": "unmodified /root/reference/swirld.py, Node.divide_rounds + Node.decide_fame, one batch call each",
    "members": n, "events": M, "stream": "first %d events of bench.py's stream (sw_synth_hashgraph(256, 1000000, seed 3))" % M,
    "divide_rounds_s": round(t1 - t0, 3), "decide_fame_s": round(t2 - t1, 3),
    "events_per_s": round(M / (t2 - t0), 1), "divide_rounds_events_per_s": round(M / (t1 - t0), 1),
    "cores": 1, "python": platform.python_version(), "cpu": cpu, "where": "authoring container (no GPU; the reference cannot travel to the GPU box)",
    "c_oracle_same_prefix_events_per_s": round(M / (c1 - c0), 1),
    "reference_equals_oracle_on_this_prefix": bool(same),
    "rounds": int(o.max_round + 1),
}
assert same, "the reference and the oracle disagree on this prefix"
with open(os.path.join(ROOT, "profiles", "reference_python_timing.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
