// Exact path for FORKED hashgraphs (SURVEY.md §8 a2 / VERDICT r1 item 6).
//
// The round-synchronous kernels of kernels.hip.h rest on one self-parent chain per member (index order
// == height order on a chain, per-round thresholds lo[r][c], chain binary searches).  A fork — two
// events of one member on the same self-parent, which the reference stores without complaint
// (README.md:84) — breaks every one of those, while the reference keeps producing results: `maxi`
// picks by HEIGHT with ties going to the self-parent's entry (swirld.py:170-184), a later fork sibling
// OVERWRITES the member's witness of a round but keeps its dict position (swirld.py:221-222), `famous`
// is keyed by event, and find_order's `higher` no longer implies ancestry.
//
// This file is the reference algorithm itself, statement by statement, on the device-resident state
// (same can_see table, round array, witness / fame / consensus tables as the fast path), run by ONE
// wavefront: the event / voter / candidate / BFS loops are sequential exactly as in swirld.py, the inner
// per-member loops are spread over the 64 lanes (lane c_ owns column c_ of the strongly-sees tally),
// sums go through wave reductions.  It is slow by construction (one wave, dependent gathers) and exact
// by construction; a context switches to it at its first forked event and stays there.
//
//   divide   swirld.py:187-222      fame   swirld.py:224-277      order   swirld.py:280-311
//
// The same source compiles for the host (g++ -DSW_EXACT_HOST, one "lane"): the CPU suite runs these very
// functions against the oracle on forked DAGs (tests/test_exact_host.py); the product only ever
// launches the kernels (k_exact_* in kernels.hip.h's translation unit).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__) && !defined(SW_EXACT_HOST)
#define SWX_HD __device__
#define SWX_DEVICE 1
#else
#define SWX_HD
#define SWX_DEVICE 0
#endif

namespace swx {

typedef uint64_t u64;

enum { X_OK = 0, X_EKEY = -2, X_EINDEX = -3, X_EINVAL = -22 };

// hdr[] slots (int64): written by lane 0, read by everyone after a sync, copied back by the host
enum { H_R = 0, H_RC, H_NNEW, H_NOUT, H_VOTER_EVALS, H_MAJ_EVALS, H_COIN_VOTES, H_COIN_FLIPS, H_QH, H_QT, H_NITEMS, H_TMP, H_ERR_AT, H_WORDS = 16 };

struct State {
    int n, npad, coin_period;
    u64 tot;                      // total stake (swirld.py:42); min_s = 2*tot/3 handled as 3x > 2*tot
    const uint32_t* stake;        // [npad]
    const int *cr, *sp, *op, *ht; // per event
    const double* t;
    const unsigned char* sig;     // 64 B per event
    int* round;                   // -1 = not divided
    int* L;                       // can_see [N][npad], -1 absent (swirld.py:69-72)
    unsigned char* tbd;           // swirld.py:53-54
    signed char* fam_ev;          // per EVENT: -1 undecided, 0 / 1 (swirld.py:64)
    int Rcap;
    int* wit;                     // [Rcap][npad] member -> witness event (last registered)
    int* worder;                  // [Rcap][npad] members in dict insertion order
    int* wcnt;                    // [Rcap]
    unsigned char* cons;          // [Rcap]
    signed char* fam_slot;        // [Rcap][npad] fame of the table entry (what sw_get_famous returns)
    long long* hdr;               // [H_WORDS]
};

// ---- the one wavefront -------------------------------------------------------------------------
#if SWX_DEVICE
struct Wave {
    static __device__ int lane() { return (int)threadIdx.x; }
    static constexpr int nl = 64;
    static __device__ void sync() { __syncthreads(); }   // one wave per workgroup: a barrier + workgroup-scope fence
    static __device__ u64 sum(u64 v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
            v += ((u64)hi << 32) | lo;
        }
        return v;
    }
};
#elif defined(SW_EXACT_HOST_LANES)
// CPU emulation of the wavefront for the tests (tests/exact_host.cpp): SW_EXACT_HOST_LANES cooperative
// fibers, each running to its next sync() before the next one starts — the opposite extreme of the
// lockstep execution of a real wavefront, so results that match under both need nothing but the syncs.
extern "C" int swx_emul_lane();
extern "C" void swx_emul_sync();
extern "C" unsigned long long swx_emul_sum(unsigned long long v);
struct Wave {
    static int lane() { return swx_emul_lane(); }
    static constexpr int nl = SW_EXACT_HOST_LANES;
    static void sync() { swx_emul_sync(); }
    static u64 sum(u64 v) { return swx_emul_sum(v); }
};
#else
struct Wave {
    static int lane() { return 0; }
    static constexpr int nl = 1;
    static void sync() {}
    static u64 sum(u64 v) { return v; }
};
#endif

// Node.higher, swirld.py:183-184
SWX_HD inline bool higher(const State& s, int a, int b) { return a >= 0 && (b < 0 || s.ht[a] >= s.ht[b]); }

// self.witnesses[r][c] = e (swirld.py:197, 222): a new key goes to the end of the round's dict, an
// existing key keeps its position and gets the new value.  Lane 0 only.
SWX_HD inline void register_witness(const State& s, int r, int c, int e) {
    if (r + 1 > (int)s.hdr[H_R]) s.hdr[H_R] = r + 1;
    int* slot = &s.wit[(size_t)r * s.npad + c];
    if (*slot < 0) s.worder[(size_t)r * s.npad + s.wcnt[r]++] = c;
    *slot = e;
}

// Node.divide_rounds(events), swirld.py:187-222, for the events [first, first + K) in index order.
SWX_HD inline int divide(const State& s, long long first, long long K) {
    const int n = s.n, np = s.npad, lane = Wave::lane();
    for (long long e = first; e < first + K; ++e) {
        int* row = s.L + (size_t)e * np;
        const int ce = s.cr[e], sp = s.sp[e], op = s.op[e];
        if (sp < 0) {  // root, swirld.py:195-198
            for (int c = lane; c < np; c += Wave::nl) row[c] = c == ce ? (int)e : -1;
            if (lane == 0) { s.round[e] = 0; register_witness(s, 0, ce, (int)e); }
            Wave::sync();
            continue;
        }
        const int rs = s.round[sp], rp = s.round[op];
        if (rs < 0 || rp < 0) { if (lane == 0) { s.hdr[H_RC] = X_EKEY; s.hdr[H_ERR_AT] = e; } return X_EKEY; }
        const int r = rs > rp ? rs : rp;  // :200
        const int* p0 = s.L + (size_t)sp * np;
        const int* p1 = s.L + (size_t)op * np;
        for (int c = lane; c < np; c += Wave::nl) {  // :203-205, maxi = swirld.py:170-174 (ties: the self-parent's entry)
            int v = -1;
            if (c < n) { const int a = p0[c], b = p1[c]; v = higher(s, a, b) ? a : b; }
            row[c] = v;
        }
        Wave::sync();
        u64 cnt = 0;  // :208-216: lane c_ owns hits[c_]
        for (int c_ = lane; c_ < n; c_ += Wave::nl) {
            u64 h = 0;
            for (int c = 0; c < n; ++c) {
                const int k = row[c];
                if (k >= 0 && s.round[k] == r) {
                    const int k_ = s.L[(size_t)k * np + c_];
                    if (k_ >= 0 && s.round[k_] == r) h += s.stake[c];
                }
            }
            if (3 * h > 2 * s.tot) ++cnt;
        }
        cnt = Wave::sum(cnt);
        Wave::sync();  // every lane has read the row before its own entry is overwritten
        if (lane == 0) {
            const int re = (3 * cnt > 2 * s.tot) ? r + 1 : r;  // :216-219: a COUNT against the stake threshold
            s.round[e] = re;
            row[ce] = (int)e;                                   // :220
            if (re > rs) register_witness(s, re, ce, (int)e);   // :221-222
        }
        Wave::sync();
    }
    return X_OK;
}

// Scratch of decide_fame: two layers of votes (the voters of round r_-1 and of round r_), indexed
// [layer][voter member][(r - max_c) * n + candidate member]; -1 = no entry (KeyError).
struct FameScratch {
    signed char* votes;     // [2][n][win * n]
    size_t layer;           // n * win * n
    int win;                // candidate rounds max_c .. max_c + win - 1
    unsigned char* s_m;     // [npad]
    unsigned char* done;    // [Rcap]
    int* new_rounds;        // [Rcap] sorted(new_c)
};

// Node.decide_fame(), swirld.py:224-277.  hdr[H_NNEW] = len(new_c).
SWX_HD inline int decide_fame(const State& s, const FameScratch& x) {
    const int n = s.n, np = s.npad, lane = Wave::lane();
    const int R = (int)s.hdr[H_R];
    if (R == 0) return X_EINVAL;  // max() of an empty dict
    const int max_r = R - 1;      // :225
    int max_c = 0;                // :226-228
    while (max_c < R && s.cons[max_c]) ++max_c;
    for (int r = lane; r < R; r += Wave::nl) x.done[r] = 0;
    for (size_t i = lane; i < 2 * x.layer; i += Wave::nl) x.votes[i] = -1;
    Wave::sync();
    const size_t vrow = (size_t)x.win * n;
    for (int r_ = max_c + 1; r_ <= max_r; ++r_) {  // iter_voters, :238-241
        signed char* cur = x.votes + (size_t)(r_ & 1) * x.layer;
        const signed char* prev = x.votes + (size_t)((r_ & 1) ^ 1) * x.layer;
        for (size_t i = lane; i < x.layer; i += Wave::nl) cur[i] = -1;
        Wave::sync();
        const int ny = s.wcnt[r_];
        for (int iy = 0; iy < ny; ++iy) {
            const int cy = s.worder[(size_t)r_ * np + iy];
            const int y = s.wit[(size_t)r_ * np + cy];
            const int* ry = s.L + (size_t)y * np;
            u64 bad = 0;
            for (int c_ = lane; c_ < n; c_ += Wave::nl) {  // :247-254
                u64 h = 0;
                for (int c = 0; c < n; ++c) {
                    const int k = ry[c];
                    if (k >= 0 && s.round[k] == r_ - 1) {
                        const int k_ = s.L[(size_t)k * np + c_];
                        if (k_ >= 0 && s.round[k_] == r_ - 1) h += s.stake[c];
                    }
                }
                const bool in_s = 3 * h > 2 * s.tot;
                x.s_m[c_] = in_s;
                if (in_s && s.wit[(size_t)(r_ - 1) * np + c_] < 0) bad = 1;  // KeyError on self.witnesses[r_-1][c]
            }
            bad = Wave::sum(bad);
            if (bad) { if (lane == 0) s.hdr[H_RC] = X_EKEY; return X_EKEY; }
            if (lane == 0) s.hdr[H_VOTER_EVALS]++;
            Wave::sync();
            for (int r = max_c; r < r_; ++r) {  // iter_undetermined(r_), :231-236
                if (s.cons[r]) continue;
                const int nx = s.wcnt[r];
                for (int ix = 0; ix < nx; ++ix) {
                    const int cx = s.worder[(size_t)r * np + ix];
                    const int xev = s.wit[(size_t)r * np + cx];
                    if (s.fam_ev[xev] >= 0) continue;  // :235
                    const int d = r_ - r;
                    signed char* slot = cur + (size_t)cy * vrow + (size_t)(r - max_c) * n + cx;
                    if (d == 1) {  // :257-258: x in s
                        if (lane == 0) *slot = x.s_m[cx] && s.wit[(size_t)(r_ - 1) * np + cx] == xev;
                    } else {
                        u64 h0 = 0, h1 = 0, miss = 0;  // majority(), swirld.py:20-27
                        for (int c = lane; c < n; c += Wave::nl) {
                            if (!x.s_m[c]) continue;
                            const int w = s.wit[(size_t)(r_ - 1) * np + c];
                            const signed char vw = prev[(size_t)c * vrow + (size_t)(r - max_c) * n + cx];
                            if (vw < 0) miss = 1;
                            else if (vw) h1 += s.stake[s.cr[w]];
                            else h0 += s.stake[s.cr[w]];
                        }
                        h0 = Wave::sum(h0); h1 = Wave::sum(h1); miss = Wave::sum(miss);
                        if (miss) { if (lane == 0) s.hdr[H_RC] = X_EKEY; return X_EKEY; }
                        const int v = !(h0 > h1);  // tie -> True
                        const u64 tt = v ? h1 : h0;
                        const bool sm = 3 * tt > 2 * s.tot;
                        if (lane == 0) {
                            s.hdr[H_MAJ_EVALS]++;
                            if (d % s.coin_period != 0) {  // :261-266
                                if (sm) { s.fam_ev[xev] = (signed char)v; x.done[r] = 1; }
                                else *slot = (signed char)v;
                            } else {                       // :267-272
                                s.hdr[H_COIN_VOTES]++;
                                if (sm) *slot = (signed char)v;
                                else { s.hdr[H_COIN_FLIPS]++; *slot = (signed char)(s.sig[64 * (size_t)y] / 128); }
                            }
                        }
                    }
                    Wave::sync();  // fam_ev / votes written by lane 0 are read by every lane next
                }
            }
        }
    }
    if (lane == 0) {  // :274-277
        int cnt = 0;
        for (int r = 0; r < R; ++r) {
            if (!x.done[r]) continue;
            bool all = true;
            for (int i = 0; i < s.wcnt[r]; ++i)
                if (s.fam_ev[s.wit[(size_t)r * np + s.worder[(size_t)r * np + i]]] < 0) { all = false; break; }
            if (all) x.new_rounds[cnt++] = r;
        }
        for (int i = 0; i < cnt; ++i) s.cons[x.new_rounds[i]] = 1;
        s.hdr[H_NNEW] = cnt;
    }
    Wave::sync();
    for (size_t i = lane; i < (size_t)R * np; i += Wave::nl) {  // the per-slot view the getters return
        const int w = s.wit[i];
        s.fam_slot[i] = w >= 0 ? s.fam_ev[w] : (signed char)-1;
    }
    return X_OK;
}

struct OrderScratch {
    int* queue;              // [N]
    unsigned char* visited;  // [N], all zero between calls
    int* fw;                 // [npad] famous witnesses of the round
    unsigned char* sflag;    // [npad]
    double* times;           // [npad]
    double* tsort;           // [npad]
    unsigned char* white;    // [64]
    int* items_ev;           // [N] received events, sorted per round, concatenated
    double* items_ts;        // [N]
};

// (ts, white ^ sig) order of swirld.py:306: big-endian 512-bit integers = bytewise comparison
SWX_HD inline bool item_less(const State& s, const OrderScratch& x, int ea, double ta, int eb, double tb) {
    if (ta < tb) return true;
    if (ta > tb) return false;
    const unsigned char* a = s.sig + 64 * (size_t)ea;
    const unsigned char* b = s.sig + 64 * (size_t)eb;
    for (int i = 0; i < 64; ++i) {
        const unsigned char ka = a[i] ^ x.white[i], kb = b[i] ^ x.white[i];
        if (ka != kb) return ka < kb;
    }
    return false;
}

// heap sort of items [base, base + m) by item_less (lane 0)
SWX_HD inline void sort_items(const State& s, const OrderScratch& x, long long base, long long m) {
    int* ev = x.items_ev + base;
    double* ts = x.items_ts + base;
    auto sift = [&](long long root, long long end) {
        for (;;) {
            long long child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && item_less(s, x, ev[child], ts[child], ev[child + 1], ts[child + 1])) ++child;
            if (!item_less(s, x, ev[root], ts[root], ev[child], ts[child])) break;
            const int te = ev[root]; ev[root] = ev[child]; ev[child] = te;
            const double tt = ts[root]; ts[root] = ts[child]; ts[child] = tt;
            root = child;
        }
    };
    for (long long i = m / 2 - 1; i >= 0; --i) sift(i, m);
    for (long long end = m - 1; end > 0; --end) {
        const int te = ev[0]; ev[0] = ev[end]; ev[end] = te;
        const double tt = ts[0]; ts[0] = ts[end]; ts[end] = tt;
        sift(0, end);
    }
}

// Node.find_order(new_c), swirld.py:280-311; `rounds` sorted by the host (sorted() at :283).
// hdr[H_NOUT] = number of events received, items_ev[0 .. H_NOUT) = their final order.
SWX_HD inline int find_order(const State& s, const OrderScratch& x, const int* rounds, int n_rounds) {
    const int np = s.npad, lane = Wave::lane();
    const int R = (int)s.hdr[H_R];
    long long produced = 0;
    for (int ir = 0; ir < n_rounds; ++ir) {
        const int r = rounds[ir];
        if (r < 0 || r >= R) { if (lane == 0) s.hdr[H_RC] = X_EKEY; return X_EKEY; }
        if (lane == 0) {  // f_w, :284
            int nfw = 0, rc = X_OK;
            for (int i = 0; i < s.wcnt[r]; ++i) {
                const int w = s.wit[(size_t)r * np + s.worder[(size_t)r * np + i]];
                if (s.fam_ev[w] < 0) { rc = X_EKEY; break; }
                if (s.fam_ev[w]) x.fw[nfw++] = w;
            }
            s.hdr[H_TMP] = rc ? -1 : nfw;
        }
        Wave::sync();
        if (s.hdr[H_TMP] < 0) { if (lane == 0) s.hdr[H_RC] = X_EKEY; return X_EKEY; }
        const int nfw = (int)s.hdr[H_TMP];
        for (int b = lane; b < 64; b += Wave::nl) {  // white, :285
            unsigned char w = 0;
            for (int i = 0; i < nfw; ++i) w ^= s.sig[64 * (size_t)x.fw[i] + b];
            x.white[b] = w;
        }
        if (lane == 0) {  // bfs over the tbd ancestors of the famous witnesses, :288-289 / utils.py:24-34
            long long qt = 0;
            for (int i = 0; i < nfw; ++i)
                if (s.tbd[x.fw[i]] && !x.visited[x.fw[i]]) { x.visited[x.fw[i]] = 1; x.queue[qt++] = x.fw[i]; }
            s.hdr[H_QH] = 0; s.hdr[H_QT] = qt; s.hdr[H_NITEMS] = 0;
        }
        Wave::sync();
        for (;;) {
            const long long qh = s.hdr[H_QH], qt = s.hdr[H_QT];
            if (qh >= qt) break;
            const int ev = x.queue[qh];
            const int c = s.cr[ev];
            u64 stake_sum = 0;  // :291-292
            for (int i = lane; i < nfw; i += Wave::nl) {
                const int k = s.L[(size_t)x.fw[i] * np + c];
                const bool f = k >= 0 && higher(s, k, ev);
                x.sflag[i] = f;
                if (f) stake_sum += s.stake[s.cr[x.fw[i]]];
            }
            stake_sum = Wave::sum(stake_sum);
            Wave::sync();
            const bool received = 2 * stake_sum > s.tot;  // :293
            if (received) {
                for (int i = lane; i < nfw; i += Wave::nl) {  // :298-303, one famous witness per lane
                    if (!x.sflag[i]) continue;
                    int a = x.fw[i];
                    for (;;) {
                        const int k = s.L[(size_t)a * np + c];
                        if (!(k >= 0 && higher(s, k, ev) && s.sp[a] >= 0)) break;
                        a = s.sp[a];
                    }
                    x.times[i] = s.t[a];
                }
                Wave::sync();
            }
            if (lane == 0) {
                if (received) {
                    s.tbd[ev] = 0;  // :294
                    int ns = 0;
                    for (int i = 0; i < nfw; ++i)
                        if (x.sflag[i]) {  // insertion sort = times.sort()
                            const double tv = x.times[i];
                            int j = ns++;
                            while (j > 0 && x.tsort[j - 1] > tv) { x.tsort[j] = x.tsort[j - 1]; --j; }
                            x.tsort[j] = tv;
                        }
                    if ((ns + 1) / 2 >= ns) { s.hdr[H_RC] = X_EINDEX; s.hdr[H_ERR_AT] = ev; }  // :305 IndexError when len == 1
                    else {
                        const long long at = produced + s.hdr[H_NITEMS]++;
                        x.items_ev[at] = ev;
                        x.items_ts[at] = .5 * (x.tsort[ns / 2] + x.tsort[(ns + 1) / 2]);  // :305
                    }
                }
                // successors: parents still in tbd, evaluated after the body ran (the lazy generator of utils.py:31)
                long long q = s.hdr[H_QT];
                if (s.sp[ev] >= 0) {
                    const int ps[2] = {s.sp[ev], s.op[ev]};
                    for (int j = 0; j < 2; ++j)
                        if (s.tbd[ps[j]] && !x.visited[ps[j]]) { x.visited[ps[j]] = 1; x.queue[q++] = ps[j]; }
                }
                s.hdr[H_QT] = q;
                s.hdr[H_QH] = qh + 1;
            }
            Wave::sync();
            if (s.hdr[H_RC] != X_OK) break;
        }
        {   // leave `visited` all zero
            const long long qt = s.hdr[H_QT];
            for (long long i = lane; i < qt; i += Wave::nl) x.visited[x.queue[i]] = 0;
        }
        Wave::sync();
        if (s.hdr[H_RC] != X_OK) return (int)s.hdr[H_RC];
        const long long nitems = s.hdr[H_NITEMS];
        if (lane == 0) sort_items(s, x, produced, nitems);  // :306
        produced += nitems;
        Wave::sync();
    }
    if (lane == 0) s.hdr[H_NOUT] = produced;
    return X_OK;
}

// State of a context that ran on the fast path so far, completed for this one (the first fork arrives):
// dict insertion order of every round's witnesses (ascending event index: without forks a member
// registers one witness per round, in processing order), fame per event, tbd.
SWX_HD inline void import_fast_state(const State& s, long long n_events, const int* ordered, long long n_ordered) {
    const int np = s.npad, lane = Wave::lane();
    const int R = (int)s.hdr[H_R];
    for (long long e = lane; e < n_events; e += Wave::nl) { s.tbd[e] = 1; s.fam_ev[e] = -1; }
    Wave::sync();
    for (long long i = lane; i < n_ordered; i += Wave::nl) s.tbd[ordered[i]] = 0;
    for (int r = lane; r < R; r += Wave::nl) {
        int cnt = 0;
        int* ord = s.worder + (size_t)r * np;
        for (int c = 0; c < s.n; ++c) {
            const int w = s.wit[(size_t)r * np + c];
            if (w < 0) continue;
            int j = cnt++;
            while (j > 0 && s.wit[(size_t)r * np + ord[j - 1]] > w) { ord[j] = ord[j - 1]; --j; }
            ord[j] = c;
            s.fam_ev[w] = s.fam_slot[(size_t)r * np + c];
        }
        s.wcnt[r] = cnt;
    }
    Wave::sync();
}

}  // namespace swx

#if SWX_DEVICE
// ---- kernels: one workgroup of one wavefront each ------------------------------------------------
__global__ void __launch_bounds__(64) k_exact_divide(swx::State s, long long first, long long K) {
    const int rc = swx::divide(s, first, K);
    if (rc != swx::X_OK && threadIdx.x == 0) s.hdr[swx::H_RC] = rc;
}
__global__ void __launch_bounds__(64) k_exact_fame(swx::State s, swx::FameScratch x) {
    const int rc = swx::decide_fame(s, x);
    if (rc != swx::X_OK && threadIdx.x == 0) s.hdr[swx::H_RC] = rc;
}
__global__ void __launch_bounds__(64) k_exact_order(swx::State s, swx::OrderScratch x, const int* rounds, int n_rounds) {
    const int rc = swx::find_order(s, x, rounds, n_rounds);
    if (rc != swx::X_OK && threadIdx.x == 0) s.hdr[swx::H_RC] = rc;
}
__global__ void __launch_bounds__(64) k_exact_import(swx::State s, long long n_events, const int* ordered, long long n_ordered) {
    swx::import_fast_state(s, n_events, ordered, n_ordered);
}
#endif
