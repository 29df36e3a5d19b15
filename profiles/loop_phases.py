#!/usr/bin/env python3
"""Where the time of one round-loop iteration goes: phase stamps written by k_resolve_band and
k_tally_bits themselves (SW_DEBUG_CLOCKS=1, 100 MHz clock, each stamp after a full s_waitcnt).
Usage: SW_DEBUG_CLOCKS=1 SW_PIPE=1 python profiles/loop_phases.py [members events]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SW_DEBUG_CLOCKS", "1")
pkg = importlib.import_module("py-swirld_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
stream = pkg.synth_hashgraph(n, N, 3)
h = pkg.Hashgraph(n)
h.reserve(N)
h.append_events(*stream)
for _ in range(2):
    h.divide_rounds(0, N)
    h.decide_fame()
    h.rewind()
h.divide_rounds(0, N)
t = h.debug_clocks().astype(np.int64)
live = (t[:, 0] > 0) & (t[:, 6] > 0)
live[:-1] &= t[1:, 0] > 0
idx = np.nonzero(live)[0][1:-1]
print("%d members, %d events: %d stamped iterations" % (n, N, len(idx)))


def q(name, d):
    if not len(d):
        return
    d = d / 100.0  # 100 MHz -> us
    print("  %-52s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90])))


A, T, Tl = 0, 16, 24
q("iteration period (resolve entry -> next resolve entry)", t[idx + 1, A] - t[idx, A])
print(" k_resolve_band, block 1 (a block that does not publish the state):")
order = [0, 1, 2, 3, 8, 10, 4, 12, 13, 14, 5, 6]
names = ["first loads back", "cursors advanced (chain_ev round trip)", "inheritance, count(unresolved)",
         "round committed", "count(active), count(unresolved), min threshold", "next round entered",
         "last candidate known", "max(last candidate), sum(evaluated)", "band range", "(state stores: writer only)",
         "band masks built (this block's share)"]
for a, b, nm in zip(order[:-1], order[1:], names):
    ok = (t[idx, a] > 0) & (t[idx, b] > 0)
    q("-> " + nm, t[idx[ok], b] - t[idx[ok], a])
q("end - entry", t[idx, 6] - t[idx, 0])
print("  band events per iteration: mean %.0f" % t[idx, A + 7].mean())
kend = t[idx, A + 6]
for name, b in (("member 0, slot K/2", T), ("last member, slot K/2", Tl)):
    ok = (t[idx, b] > 0) & (t[idx, b + 1] > 0)
    i1 = idx[ok]
    print(" k_tally_bits, wave of %s (%d iterations reached it):" % (name, len(i1)))
    if not len(i1):
        continue
    q("resolve end (stamped blocks) -> entry", t[i1, b] - kend[ok])
    q("entry -> first loads back", t[i1, b + 1] - t[i1, b])
    for k, lab in ((2, "-> candidate + self-parent ids"), (3, "-> can_see row + other-parent"),
                   (4, "-> hop masks gathered + counted"), (5, "-> compared, result published")):
        okk = (t[i1, b + k] > 0) & (t[i1, b + k - 1] > 0)
        if okk.any():
            q(lab + " (%d)" % okk.sum(), t[i1[okk], b + k] - t[i1[okk], b + k - 1])
    okk = t[i1, b + 5] > 0
    if okk.any():
        q("end - entry", t[i1[okk], b + 5] - t[i1[okk], b])
        q("end -> next resolve entry", t[i1[okk] + 1, A] - t[i1[okk], b + 5])
# ---- which iterations are the slow ones?  (period against the band built in that iteration, the kernel's own length and
# the gaps on either side of the tally)
per = (t[idx + 1, A] - t[idx, A]) / 100.0
band = t[idx, A + 7]
res_len = (t[idx, 6] - t[idx, 0]) / 100.0
slow = per > np.percentile(per, 50) * 1.25
print(" slow iterations (period > 1.25 x median): %d of %d, mean period %.1f us (the others %.1f us)" % (slow.sum(), len(per), per[slow].mean() if slow.any() else 0, per[~slow].mean()))
if slow.any():
    print("   band events built: slow %.0f, others %.0f;  resolve kernel entry -> end: slow %.2f us, others %.2f us" % (band[slow].mean(), band[~slow].mean(), res_len[slow].mean(), res_len[~slow].mean()))
    okT = t[idx, T + 5] > 0
    for lab, m in (("slow", slow & okT), ("others", ~slow & okT)):
        if m.any():
            i2 = idx[m]
            print("   %-6s resolve end -> tally wave entry %.2f us, tally wave %.2f us, its end -> next resolve entry %.2f us" % (
                lab, ((t[i2, T] - t[i2, 6]) / 100.0).mean(), ((t[i2, T + 5] - t[i2, T]) / 100.0).mean(), ((t[i2 + 1, A] - t[i2, T + 5]) / 100.0).mean()))
    print("   iteration index of the slow ones (mod 24):", np.bincount(idx[slow] % 24, minlength=24).tolist())
    print("   first 40 slow iteration indices:", idx[slow][:40].tolist())
# ---- the same phases along the pass (round 5: the stamps are indexed by the iteration of the context, so the loops of all
# sub-batches are there): groups of consecutive iterations, to see what the sweeps running beside the first loops cost and where
G = int(os.environ.get("LOOP_PHASES_GROUP", "24"))
last = idx[idx >= idx.max() - int(os.environ.get("LOOP_PHASES_LAST", "330"))]
print(" along the pass (groups of %d iterations; medians, us): period | resolve step | band | resolve end -> tally entry | tally wave | tally end -> next entry" % G)
for a0 in range(0, len(last), G):
    g = last[a0:a0 + G]
    g = g[(t[g, T] > 0) & (t[g, T + 5] > 0) & (t[g, 5] > 0)]
    if len(g) < 4:
        continue
    med = lambda x: float(np.median(x)) / 100.0
    print("   iterations %5d..%5d: %6.2f | %5.2f | %5.2f | %5.2f | %5.2f | %5.2f" % (
        g[0], g[-1], med(t[g + 1, A] - t[g, A]), med(t[g, 5] - t[g, 0]), med(t[g, 6] - t[g, 5]), med(t[g, T] - t[g, 6]),
        med(t[g, T + 5] - t[g, T]), med(t[g + 1, A] - t[g, T + 5])))
