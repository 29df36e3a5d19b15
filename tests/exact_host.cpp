// Host build of exact.hip.h (g++ -DSW_EXACT_HOST) — TEST INFRASTRUCTURE: lets the CPU suite run the very
// functions the exact (forked-hashgraph) kernels execute against the oracle (tests/test_exact_host.py).
// The product never loads this library; it launches the kernels of swirld_hip.hip.
#define SW_EXACT_HOST 1
#include "../py-swirld_amd/csrc/exact.hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

// ---- run(body): the body once (one lane), or — built with -DSW_EXACT_HOST_LANES=64 — once per lane on
// cooperative fibers that hand over at every sync(): lane 0 runs to its first sync, then lane 1, ...;
// uniform control flow makes "yield to the next lane" a barrier.  A lane that ends while another
// still waits in a sync is a non-uniform sync count: reported and aborted.
#ifdef SW_EXACT_HOST_LANES
#include <ucontext.h>
namespace emul {
constexpr int NL = SW_EXACT_HOST_LANES;
ucontext_t main_ctx, ctx[NL];
std::vector<char> stacks;
bool done[NL];
int cur = 0, syncs[NL];
unsigned long long slot[NL];
const std::function<void()>* body = nullptr;
void entry() { (*body)(); done[cur] = true; swapcontext(&ctx[cur], &main_ctx); }
void run(const std::function<void()>& f) {
    body = &f;
    if (stacks.empty()) stacks.resize((size_t)NL * (256 << 10));
    for (int i = 0; i < NL; ++i) {
        getcontext(&ctx[i]);
        ctx[i].uc_stack.ss_sp = stacks.data() + (size_t)i * (256 << 10);
        ctx[i].uc_stack.ss_size = 256 << 10;
        ctx[i].uc_link = &main_ctx;
        makecontext(&ctx[i], entry, 0);
        done[i] = false; syncs[i] = 0;
    }
    for (;;) {  // one sweep = every live lane runs to its next sync (or to its end)
        int live = 0;
        for (int i = 0; i < NL; ++i) {
            if (done[i]) continue;
            cur = i;
            swapcontext(&main_ctx, &ctx[i]);
            live += !done[i];
        }
        if (!live) break;
        for (int i = 0; i < NL; ++i)
            if (done[i] != done[0] || syncs[i] != syncs[0]) { fprintf(stderr, "lanes disagree on the number of syncs (lane %d)\n", i); abort(); }
    }
}
}  // namespace emul
extern "C" int swx_emul_lane() { return emul::cur; }
extern "C" void swx_emul_sync() { emul::syncs[emul::cur]++; swapcontext(&emul::ctx[emul::cur], &emul::main_ctx); }
extern "C" unsigned long long swx_emul_sum(unsigned long long v) {
    emul::slot[emul::cur] = v;
    swx_emul_sync();
    unsigned long long t = 0;
    for (int i = 0; i < emul::NL; ++i) t += emul::slot[i];
    swx_emul_sync();
    return t;
}
static void run(const std::function<void()>& f) { emul::run(f); }
#else
static void run(const std::function<void()>& f) { f(); }
#endif

namespace {
struct Ctx {
    int n, np, C;
    std::vector<uint32_t> stake;
    uint64_t tot = 0;
    std::vector<int> cr, sp, op, ht, round, L, wit, worder, wcnt, tx;
    std::vector<double> t;
    std::vector<unsigned char> sig, tbd, cons;
    std::vector<signed char> fam_ev, fam_slot;
    long long hdr[swx::H_WORDS];
    int Rcap = 0;
    long long N = 0;
    swx::State state() {
        swx::State s{};
        s.n = n; s.npad = np; s.coin_period = C; s.tot = tot; s.stake = stake.data();
        s.cr = cr.data(); s.sp = sp.data(); s.op = op.data(); s.ht = ht.data(); s.t = t.data(); s.sig = sig.data();
        s.round = round.data(); s.L = L.data(); s.tbd = tbd.data(); s.fam_ev = fam_ev.data();
        s.Rcap = Rcap; s.wit = wit.data(); s.worder = worder.data(); s.wcnt = wcnt.data(); s.cons = cons.data();
        s.fam_slot = fam_slot.data(); s.hdr = hdr;
        return s;
    }
    void rounds(int need) {
        if (need <= Rcap) return;
        wit.resize((size_t)need * np, -1); worder.resize((size_t)need * np, 0); wcnt.resize(need, 0);
        cons.resize(need, 0); fam_slot.resize((size_t)need * np, -1);
        Rcap = need;
    }
};
}  // namespace

extern "C" {
void* swx_host_create(int n, const uint32_t* stake, int coin_period) {
    Ctx* c = new Ctx;
    c->n = n; c->np = (n + 63) / 64 * 64; c->C = coin_period;
    c->stake.assign(c->np, 0);
    for (int i = 0; i < n; ++i) { c->stake[i] = stake[i]; c->tot += stake[i]; }
    memset(c->hdr, 0, sizeof c->hdr);
    return c;
}
void swx_host_destroy(void* p) { delete (Ctx*)p; }
int swx_host_append(void* p, long long K, const int* cr, const int* sp, const int* op, const double* t, const unsigned char* sig) {
    Ctx* c = (Ctx*)p;
    int hmax = 0;
    for (long long i = 0; i < K; ++i) {
        c->cr.push_back(cr[i]); c->sp.push_back(sp[i]); c->op.push_back(op[i]);
        c->ht.push_back(sp[i] < 0 ? 0 : std::max(c->ht[sp[i]], c->ht[op[i]]) + 1);
        c->t.push_back(t ? t[i] : 0.0);
        for (int b = 0; b < 64; ++b) c->sig.push_back(sig ? sig[64 * i + b] : 0);
        c->round.push_back(-1); c->tbd.push_back(1); c->fam_ev.push_back(-1);
    }
    c->N += K;
    c->L.resize((size_t)c->N * c->np, -1);
    for (int h : c->ht) hmax = std::max(hmax, h);
    c->rounds(hmax + 3);
    return 0;
}
int swx_host_divide(void* p, long long first, long long K) {
    Ctx* c = (Ctx*)p;
    c->hdr[swx::H_RC] = 0;
    int rc = 0;
    const swx::State st = c->state();
    run([&] { const int r = swx::divide(st, first, K); if (r) rc = r; });
    return rc;
}
int swx_host_fame(void* p, int* new_rounds, int* n_new) {
    Ctx* c = (Ctx*)p;
    c->hdr[swx::H_RC] = 0; c->hdr[swx::H_NNEW] = 0;
    const int R = (int)c->hdr[swx::H_R];
    int max_c = 0;
    while (max_c < R && c->cons[max_c]) ++max_c;
    swx::FameScratch x{};
    x.win = std::max(1, R - max_c);
    x.layer = (size_t)c->n * x.win * c->n;
    std::vector<signed char> votes(2 * x.layer);
    std::vector<unsigned char> s_m(c->np), done(c->Rcap);
    std::vector<int> nr(c->Rcap);
    x.votes = votes.data(); x.s_m = s_m.data(); x.done = done.data(); x.new_rounds = nr.data();
    int rc = 0;
    const swx::State st = c->state();
    run([&] { const int r = swx::decide_fame(st, x); if (r) rc = r; });
    *n_new = (int)c->hdr[swx::H_NNEW];
    for (int i = 0; i < *n_new; ++i) new_rounds[i] = nr[i];
    return rc;
}
int swx_host_order(void* p, const int* rounds, int n_rounds, int* out, long long* n_out) {
    Ctx* c = (Ctx*)p;
    c->hdr[swx::H_RC] = 0; c->hdr[swx::H_NOUT] = 0;
    std::vector<int> rs(rounds, rounds + n_rounds);
    std::sort(rs.begin(), rs.end());
    swx::OrderScratch x{};
    std::vector<int> queue(c->N + 1), fw(c->np), items_ev(c->N + 1);
    std::vector<unsigned char> visited(c->N + 1, 0), sflag(c->np), white(64);
    std::vector<double> times(c->np), tsort(c->np), items_ts(c->N + 1);
    x.queue = queue.data(); x.visited = visited.data(); x.fw = fw.data(); x.sflag = sflag.data(); x.times = times.data();
    x.tsort = tsort.data(); x.white = white.data(); x.items_ev = items_ev.data(); x.items_ts = items_ts.data();
    int rc = 0;
    const swx::State st = c->state();
    run([&] { const int r = swx::find_order(st, x, rs.data(), n_rounds); if (r) rc = r; });
    *n_out = c->hdr[swx::H_NOUT];
    for (long long i = 0; i < *n_out; ++i) { out[i] = items_ev[i]; c->tx.push_back(items_ev[i]); }
    return rc;
}
// getters
int swx_host_R(void* p) { return (int)((Ctx*)p)->hdr[swx::H_R]; }
void swx_host_get(void* p, int* round, int* L, int* wit, int* worder, int* wcnt, unsigned char* cons, signed char* fam_ev, unsigned char* tbd, signed char* fam_slot) {
    Ctx* c = (Ctx*)p;
    const int R = (int)c->hdr[swx::H_R];
    memcpy(round, c->round.data(), c->N * sizeof(int));
    for (long long e = 0; e < c->N; ++e) memcpy(L + e * c->n, c->L.data() + e * c->np, c->n * sizeof(int));
    for (int r = 0; r < R; ++r) {
        memcpy(wit + (size_t)r * c->n, c->wit.data() + (size_t)r * c->np, c->n * sizeof(int));
        memcpy(worder + (size_t)r * c->n, c->worder.data() + (size_t)r * c->np, c->n * sizeof(int));
        memcpy(fam_slot + (size_t)r * c->n, c->fam_slot.data() + (size_t)r * c->np, c->n);
        wcnt[r] = c->wcnt[r]; cons[r] = c->cons[r];
    }
    memcpy(fam_ev, c->fam_ev.data(), c->N);
    memcpy(tbd, c->tbd.data(), c->N);
}
// a context that ran fork-free so far: witness order / fame per event / tbd rebuilt from the per-slot tables
void swx_host_import_check(void* p) {
    Ctx* c = (Ctx*)p;
    std::vector<int> worder0(c->worder), wcnt0(c->wcnt);
    std::vector<signed char> fam0(c->fam_ev);
    std::vector<unsigned char> tbd0(c->tbd);
    const swx::State st = c->state();
    run([&] { swx::import_fast_state(st, c->N, c->tx.data(), (long long)c->tx.size()); });
}
}
