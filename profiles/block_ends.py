#!/usr/bin/env python3
"""When does every workgroup of the two round-loop kernels finish?  (SW_DEBUG_CLOCKS=3: each workgroup stamps its end.)
Prints, per kernel, the spread of the workgroups' end times inside an iteration: median, 90 %, 99 %, last — relative to the
first workgroup done — and which workgroups are last.  Usage: python profiles/block_ends.py [members events]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SW_DEBUG_CLOCKS"] = "3"
os.environ.setdefault("SW_PIPE", "1")
pkg = importlib.import_module("py-swirld_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*pkg.synth_hashgraph(n, N, 3))
for _ in range(2):
    h.divide_rounds(0, N)
    h.decide_fame()
    h.rewind()
h.divide_rounds(0, N)
t = h.debug_block_clocks(512).astype(np.int64)
ent = h.debug_clocks().astype(np.int64)
for kk, name in ((0, "k_resolve_band"), (1, "tally kernel")):
    rows = []
    last_ids = []
    for it in range(20, 300):
        x = t[it, kk]
        m = x > 0
        if m.sum() < 16:
            continue
        v = x[m]
        base = v.min()
        rows.append([np.median(v) - base, np.percentile(v, 90) - base, np.percentile(v, 99) - base, v.max() - base, m.sum()])
        last_ids.append(int(np.flatnonzero(m)[np.argmax(v)]))
    if not rows:
        print("%s: no workgroup stamps (k_tally_tree does not stamp: pin SW_TALLY_IMPL=1 to see k_tally_bits)" % name)
        continue
    r = np.array(rows, float)
    print("%s: %d iterations, %d workgroups stamped; end times after the FIRST workgroup done (us): median %.2f  p90 %.2f  p99 %.2f  last %.2f"
          % (name, len(r), int(r[:, 4].mean()), *(r[:, :4].mean(axis=0) / 100)))
    ids, cnt = np.unique(last_ids, return_counts=True)
    top = np.argsort(-cnt)[:8]
    print("   workgroups that finish last most often:", [(int(ids[i]), int(cnt[i])) for i in top])
# resolve kernel: entry of block 1 -> last block done
ok = (ent[:300, 0] > 0)
d = [(t[it, 0][t[it, 0] > 0].max() - ent[it, 0]) / 100 for it in range(20, 300) if ok[it] and (t[it, 0] > 0).any()]
print("k_resolve_band: entry of workgroup 1 -> last workgroup done: mean %.2f us" % np.mean(d))
