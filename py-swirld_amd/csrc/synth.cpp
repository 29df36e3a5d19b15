// Synthetic hashgraph generator (host only, no GPU).
//
// The reference has no generator; its only event source is the random gossip
// stepping of swirld.test() (swirld.py:331-345: pick a random node, it syncs
// with a random *other* node and creates one event whose self-parent is its own
// head and whose other-parent is the peer's head, swirld.py:139-144, 323).  This
// file produces the same DAG shape directly as dense-index SoA arrays so that
// bench.py / tests can feed 10^5..10^7 events without running Ed25519.
//
// Index order is a valid topological order; the DAG is fork-free by
// construction (each member's events form one self-parent chain).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/swirld_hip.h"

namespace {
struct Xoshiro {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Xoshiro(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // uniform in [0, n)
    uint32_t below(uint32_t n) { return (uint32_t)(((unsigned __int128)next() * n) >> 64); }
    double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};
}  // namespace

extern "C" int sw_synth_hashgraph(int n, int64_t N, uint64_t seed, int mode, double p0, double p1,
                                  int32_t* creator, int32_t* self_parent, int32_t* other_parent,
                                  double* t, uint8_t* sig64) {
    if (n < 2 || N < n || !creator || !self_parent || !other_parent) return SW_EINVAL;
    if (mode < 0 || mode > 3) return SW_EINVAL;
    Xoshiro rng(seed * 0x2545F4914F6CDD1Dull + 0x1234567ull + (uint64_t)mode);
    std::vector<int32_t> head(n);
    for (int c = 0; c < n; ++c) {  // one root per member (swirld.py:75-80)
        creator[c] = c; self_parent[c] = -1; other_parent[c] = -1; head[c] = c;
    }
    // mode 2: cumulative activity weights
    std::vector<double> cum;
    if (mode == 2) {
        int n_slow = (int)(p0 * n);
        if (n_slow >= n) n_slow = n - 1;
        cum.resize(n);
        double acc = 0;
        for (int c = 0; c < n; ++c) { acc += (c >= n - n_slow) ? p1 : 1.0; cum[c] = acc; }
    }
    const int half = n / 2;
    for (int64_t i = n; i < N; ++i) {
        int a, b;
        if (mode == 2) {
            double u = rng.unit() * cum[n - 1];
            int lo = 0, hi = n - 1;
            while (lo < hi) { int mid = (lo + hi) / 2; if (cum[mid] > u) hi = mid; else lo = mid + 1; }
            a = lo;
        } else {
            a = (int)rng.below((uint32_t)n);
        }
        if (mode == 1 && half >= 2 && n - half >= 2) {
            // two cliques; cross-clique other-parent with probability p0
            bool cross = rng.unit() < p0;
            bool a_low = a < half;
            bool pick_low = cross ? !a_low : a_low;
            int base = pick_low ? 0 : half, cnt = pick_low ? half : n - half;
            do { b = base + (int)rng.below((uint32_t)cnt); } while (b == a);
        } else {
            b = (int)rng.below((uint32_t)(n - 1));
            if (b >= a) ++b;
        }
        int32_t o = head[b];
        if (mode == 3) {
            // stale other-parent: walk back the peer's self-parent chain with prob p0 per step
            while (self_parent[o] >= 0 && rng.unit() < p0) o = self_parent[o];
        }
        creator[i] = a; self_parent[i] = head[a]; other_parent[i] = o; head[a] = (int32_t)i;
    }
    if (t) for (int64_t i = 0; i < N; ++i) t[i] = (double)i;
    if (sig64) {
        Xoshiro srng(seed ^ 0xA5A5A5A5DEADBEEFull);
        uint64_t* w = reinterpret_cast<uint64_t*>(sig64);
        for (int64_t i = 0; i < N * 8; ++i) w[i] = srng.next();
    }
    return SW_OK;
}
