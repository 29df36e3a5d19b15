/*
 * swirld_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, single-threaded CPU restatement of the virtual-voting hot path of
 * the reference (Lapin0t/py-swirld, /root/reference/swirld.py), in dense-index form.
 * It follows the reference statement by statement (each function cites the lines it
 * restates) and deliberately uses none of the reformulations the HIP path relies on
 * (no round-synchronous peeling, no bitmasks, no candidate-major elections), so that
 * it can serve as the checker for them.
 *
 * Pinning: the reference ships no tests or golden vectors.  This oracle
 * is pinned against outputs of the UNMODIFIED reference itself, generated in the
 * authoring container by tests/golden/make_golden.py (which imports
 * /root/reference/swirld.py) and committed under tests/golden/ (npz files); see
 * tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OR_OK 0
#define OR_EINVAL (-22)
#define OR_ENOMEM (-12)
#define OR_EKEY (-2)    /* the reference would raise KeyError here   */
#define OR_EINDEX (-3)  /* the reference would raise IndexError here */

typedef struct {
    uint64_t key;  /* (voter_event << 32) | candidate_event, +1 so that 0 = empty */
    int8_t val;
} vote_slot;

typedef struct or_ctx {
    int n;
    int coin_period;  /* C, swirld.py:17 */
    uint64_t* stake;  /* per member, swirld.py:41 */
    uint64_t tot;     /* tot_stake, swirld.py:42; min_s = 2*tot/3 handled as 3x > 2*tot */
    int64_t N, cap;
    int32_t *cr, *sp, *op, *ht;
    double* t;
    uint8_t* sig; /* 64 B per event */
    int32_t* round;   /* -1 = not yet divided */
    int32_t* cansee;  /* [cap][n], -1 absent (swirld.py:69-72) */
    uint8_t* tbd;     /* swirld.py:53-54 */
    int8_t* famous;   /* per EVENT: -1 undecided, 0/1 (swirld.py:64) */
    int64_t divided;
    /* witnesses: {round -> ordered {member -> event}} (swirld.py:62-63) */
    int R, Rcap;
    int32_t* wit;        /* [Rcap][n] */
    int32_t* wit_order;  /* [Rcap][n] members in dict insertion order */
    int32_t* wit_cnt;    /* [Rcap] */
    uint8_t* consensus;  /* [Rcap] (swirld.py:58-59) */
    /* votes {voter -> {candidate -> bool}} (swirld.py:60-61) */
    vote_slot* votes;
    uint64_t votes_cap, votes_cnt;
    /* transactions / idx (swirld.py:55-57) */
    int32_t* transactions;
    int64_t n_tx;
    /* counters */
    int64_t voter_evals, majority_evals, tally_inner;
    int64_t coin_votes, coin_flips;  /* votes cast in coin rounds (swirld.py:267-272), of which by the signature bit (:272) */
    int max_dist;                    /* largest voter-candidate round distance evaluated */
} or_ctx;

static int grow_rounds(or_ctx* o, int need) {
    if (need <= o->Rcap) return OR_OK;
    int nc = o->Rcap ? o->Rcap : 64;
    while (nc < need) nc *= 2;
    int32_t* w = realloc(o->wit, (size_t)nc * o->n * sizeof(int32_t));
    if (!w) return OR_ENOMEM;
    o->wit = w;
    int32_t* wo = realloc(o->wit_order, (size_t)nc * o->n * sizeof(int32_t));
    if (!wo) return OR_ENOMEM;
    o->wit_order = wo;
    int32_t* wc = realloc(o->wit_cnt, (size_t)nc * sizeof(int32_t));
    if (!wc) return OR_ENOMEM;
    o->wit_cnt = wc;
    uint8_t* cs = realloc(o->consensus, (size_t)nc);
    if (!cs) return OR_ENOMEM;
    o->consensus = cs;
    for (int r = o->Rcap; r < nc; ++r) {
        for (int c = 0; c < o->n; ++c) o->wit[(size_t)r * o->n + c] = -1;
        o->wit_cnt[r] = 0;
        o->consensus[r] = 0;
    }
    o->Rcap = nc;
    return OR_OK;
}

static int grow_events(or_ctx* o, int64_t need) {
    if (need <= o->cap) return OR_OK;
    int64_t nc = o->cap ? o->cap : 1024;
    while (nc < need) nc *= 2;
#define GROW(p, T, per)                                        \
    do {                                                       \
        T* q = realloc(o->p, (size_t)nc * (per) * sizeof(T)); \
        if (!q) return OR_ENOMEM;                              \
        o->p = q;                                              \
    } while (0)
    GROW(cr, int32_t, 1);
    GROW(sp, int32_t, 1);
    GROW(op, int32_t, 1);
    GROW(ht, int32_t, 1);
    GROW(t, double, 1);
    GROW(sig, uint8_t, 64);
    GROW(round, int32_t, 1);
    GROW(cansee, int32_t, o->n);
    GROW(tbd, uint8_t, 1);
    GROW(famous, int8_t, 1);
    GROW(transactions, int32_t, 1);
#undef GROW
    o->cap = nc;
    return OR_OK;
}

int or_create(int n, const uint64_t* stake, int coin_period, or_ctx** out) {
    if (n < 1 || !stake || coin_period < 1 || !out) return OR_EINVAL;
    or_ctx* o = calloc(1, sizeof(or_ctx));
    if (!o) return OR_ENOMEM;
    o->n = n;
    o->coin_period = coin_period;
    o->stake = malloc(sizeof(uint64_t) * n);
    for (int c = 0; c < n; ++c) {
        o->stake[c] = stake[c];
        o->tot += stake[c];
    }
    o->votes_cap = 1u << 16;
    o->votes = calloc(o->votes_cap, sizeof(vote_slot));
    *out = o;
    return OR_OK;
}

void or_destroy(or_ctx* o) {
    if (!o) return;
    free(o->stake); free(o->cr); free(o->sp); free(o->op); free(o->ht); free(o->t);
    free(o->sig); free(o->round); free(o->cansee); free(o->tbd); free(o->famous);
    free(o->wit); free(o->wit_order); free(o->wit_cnt); free(o->consensus);
    free(o->votes); free(o->transactions);
    free(o);
}

/* Node.add_event, swirld.py:114-120 (hg insert, tbd.add, height). */
int or_append_events(or_ctx* o, int64_t K, const int32_t* creator, const int32_t* sp,
                     const int32_t* op, const double* t, const uint8_t* sig64) {
    int rc = grow_events(o, o->N + K);
    if (rc) return rc;
    for (int64_t i = 0; i < K; ++i) {
        int64_t e = o->N + i;
        if (creator[i] < 0 || creator[i] >= o->n) return OR_EINVAL;
        if ((sp[i] < 0) != (op[i] < 0)) return OR_EINVAL;
        if (sp[i] >= e || op[i] >= e) return OR_EINVAL;
        o->cr[e] = creator[i];
        o->sp[e] = sp[i];
        o->op[e] = op[i];
        if (sp[i] < 0) {
            o->ht[e] = 0; /* swirld.py:117-118 */
        } else {
            int32_t a = o->ht[sp[i]], b = o->ht[op[i]];
            o->ht[e] = (a > b ? a : b) + 1; /* swirld.py:119-120 */
        }
        o->t[e] = t ? t[i] : 0.0;
        if (sig64) memcpy(o->sig + 64 * e, sig64 + 64 * i, 64);
        else memset(o->sig + 64 * e, 0, 64);
        o->round[e] = -1;
        o->tbd[e] = 1; /* swirld.py:116 */
        o->famous[e] = -1;
    }
    o->N += K;
    return OR_OK;
}

/* Node.higher, swirld.py:183-184: a is not None and (b is None or height[a] >= height[b]) */
static inline int higher(const or_ctx* o, int32_t a, int32_t b) {
    return a >= 0 && (b < 0 || o->ht[a] >= o->ht[b]);
}

static int register_witness(or_ctx* o, int r, int c, int32_t e) {
    int rc = grow_rounds(o, r + 1);
    if (rc) return rc;
    if (r + 1 > o->R) o->R = r + 1;
    int32_t* slot = &o->wit[(size_t)r * o->n + c];
    if (*slot < 0) o->wit_order[(size_t)r * o->n + o->wit_cnt[r]++] = c; /* new dict key */
    *slot = e; /* overwrite keeps dict position */
    return OR_OK;
}

/* Node.divide_rounds(events), swirld.py:187-222, for events [first, first+K). */
int or_divide_rounds(or_ctx* o, int64_t first, int64_t K) {
    const int n = o->n;
    if (first < 0 || K < 0 || first + K > o->N) return OR_EINVAL;
    uint64_t* hits = malloc(sizeof(uint64_t) * n);
    if (!hits) return OR_ENOMEM;
    for (int64_t e = first; e < first + K; ++e) {
        int32_t* row = o->cansee + (size_t)e * n;
        const int c_e = o->cr[e];
        if (o->sp[e] < 0) { /* root, swirld.py:195-198 */
            o->round[e] = 0;
            int rc = register_witness(o, 0, c_e, (int32_t)e);
            if (rc) { free(hits); return rc; }
            for (int c = 0; c < n; ++c) row[c] = -1;
            row[c_e] = (int32_t)e;
            continue;
        }
        const int32_t s = o->sp[e], p = o->op[e];
        if (o->round[s] < 0 || o->round[p] < 0) { free(hits); return OR_EKEY; }
        const int r = o->round[s] > o->round[p] ? o->round[s] : o->round[p]; /* :200 */
        const int32_t* p0 = o->cansee + (size_t)s * n;
        const int32_t* p1 = o->cansee + (size_t)p * n;
        for (int c = 0; c < n; ++c) /* :203-205, maxi = swirld.py:170-174 */
            row[c] = higher(o, p0[c], p1[c]) ? p0[c] : p1[c];
        memset(hits, 0, sizeof(uint64_t) * n);
        for (int c = 0; c < n; ++c) { /* :208-214 */
            const int32_t k = row[c];
            if (k >= 0 && o->round[k] == r) {
                const int32_t* rk = o->cansee + (size_t)k * n;
                for (int c_ = 0; c_ < n; ++c_) {
                    const int32_t k_ = rk[c_];
                    if (k_ >= 0 && o->round[k_] == r) { hits[c_] += o->stake[c]; o->tally_inner++; }
                }
            }
        }
        uint64_t cnt = 0; /* :216: a COUNT of members compared with the STAKE threshold (Q2) */
        for (int c_ = 0; c_ < n; ++c_)
            if (3 * hits[c_] > 2 * o->tot) ++cnt;
        o->round[e] = (3 * cnt > 2 * o->tot) ? r + 1 : r; /* :216-219 */
        row[c_e] = (int32_t)e;                            /* :220 */
        if (o->round[e] > o->round[s]) {                  /* :221-222 */
            int rc = register_witness(o, o->round[e], c_e, (int32_t)e);
            if (rc) { free(hits); return rc; }
        }
    }
    o->divided = first + K;
    free(hits);
    return OR_OK;
}

/* ---- votes dict ---- */
static uint64_t vhash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
static int votes_put(or_ctx* o, int32_t y, int32_t x, int v);
static int votes_grow(or_ctx* o) {
    vote_slot* old = o->votes;
    uint64_t oc = o->votes_cap;
    o->votes_cap *= 2;
    o->votes = calloc(o->votes_cap, sizeof(vote_slot));
    if (!o->votes) { o->votes = old; o->votes_cap = oc; return OR_ENOMEM; }
    o->votes_cnt = 0;
    for (uint64_t i = 0; i < oc; ++i)
        if (old[i].key) {
            uint64_t k = old[i].key - 1;
            votes_put(o, (int32_t)(k >> 32), (int32_t)(k & 0xffffffffu), old[i].val);
        }
    free(old);
    return OR_OK;
}
static int votes_put(or_ctx* o, int32_t y, int32_t x, int v) {
    if ((o->votes_cnt + 1) * 2 > o->votes_cap) {
        int rc = votes_grow(o);
        if (rc) return rc;
    }
    uint64_t key = (((uint64_t)(uint32_t)y << 32) | (uint32_t)x) + 1;
    uint64_t m = o->votes_cap - 1, i = vhash(key) & m;
    while (o->votes[i].key && o->votes[i].key != key) i = (i + 1) & m;
    if (!o->votes[i].key) { o->votes[i].key = key; o->votes_cnt++; }
    o->votes[i].val = (int8_t)v;
    return OR_OK;
}
static int votes_get(const or_ctx* o, int32_t y, int32_t x) { /* -1 = KeyError */
    uint64_t key = (((uint64_t)(uint32_t)y << 32) | (uint32_t)x) + 1;
    uint64_t m = o->votes_cap - 1, i = vhash(key) & m;
    while (o->votes[i].key) {
        if (o->votes[i].key == key) return o->votes[i].val;
        i = (i + 1) & m;
    }
    return -1;
}

/* Node.decide_fame(), swirld.py:224-277.  new_rounds receives sorted(new_c). */
int or_decide_fame(or_ctx* o, int32_t* new_rounds, int cap, int* n_new) {
    const int n = o->n;
    if (o->R == 0) return OR_EINVAL;   /* max() of empty dict */
    const int max_r = o->R - 1;        /* :225 */
    int max_c = 0;                     /* :226-228 */
    while (max_c < o->R && o->consensus[max_c]) ++max_c;
    uint8_t* done = calloc((size_t)o->R + 1, 1);
    uint64_t* hits = malloc(sizeof(uint64_t) * n);
    uint8_t* s_m = malloc(n);
    if (!done || !hits || !s_m) { free(done); free(hits); free(s_m); return OR_ENOMEM; }
    int rc = OR_OK;
    for (int r_ = max_c + 1; r_ <= max_r && !rc; ++r_) { /* iter_voters, :238-241 */
        for (int iy = 0; iy < o->wit_cnt[r_] && !rc; ++iy) {
            const int32_t y = o->wit[(size_t)r_ * n + o->wit_order[(size_t)r_ * n + iy]];
            const int32_t* ry = o->cansee + (size_t)y * n;
            memset(hits, 0, sizeof(uint64_t) * n);
            for (int c = 0; c < n; ++c) { /* :247-252 */
                const int32_t k = ry[c];
                if (k >= 0 && o->round[k] == r_ - 1) {
                    const int32_t* rk = o->cansee + (size_t)k * n;
                    for (int c_ = 0; c_ < n; ++c_) {
                        const int32_t k_ = rk[c_];
                        if (k_ >= 0 && o->round[k_] == r_ - 1) hits[c_] += o->stake[c];
                    }
                }
            }
            o->voter_evals++;
            for (int c = 0; c < n; ++c) { /* :253-254 */
                s_m[c] = (3 * hits[c] > 2 * o->tot);
                if (s_m[c] && o->wit[(size_t)(r_ - 1) * n + c] < 0) { rc = OR_EKEY; break; }
            }
            if (rc) break;
            for (int r = max_c; r < r_ && !rc; ++r) { /* iter_undetermined(r_), :231-236 */
                if (o->consensus[r]) continue;
                for (int ix = 0; ix < o->wit_cnt[r] && !rc; ++ix) {
                    const int cx = o->wit_order[(size_t)r * n + ix];
                    const int32_t x = o->wit[(size_t)r * n + cx];
                    if (o->famous[x] >= 0) continue; /* :235 */
                    const int d = r_ - r;
                    if (d == 1) { /* :257-258: x in s */
                        rc = votes_put(o, y, x, s_m[cx] && o->wit[(size_t)(r_ - 1) * n + cx] == x);
                    } else {
                        uint64_t h0 = 0, h1 = 0; /* majority(), swirld.py:20-27 */
                        for (int c = 0; c < n; ++c) {
                            if (!s_m[c]) continue;
                            const int32_t w = o->wit[(size_t)(r_ - 1) * n + c];
                            const int vw = votes_get(o, w, x);
                            if (vw < 0) { rc = OR_EKEY; break; }
                            if (vw) h1 += o->stake[o->cr[w]]; else h0 += o->stake[o->cr[w]];
                        }
                        if (rc) break;
                        o->majority_evals++;
                        const int v = !(h0 > h1);           /* tie -> True */
                        const uint64_t tt = v ? h1 : h0;
                        const int sm = (3 * tt > 2 * o->tot);
                        if (d % o->coin_period != 0) { /* :261-266 */
                            if (sm) { o->famous[x] = (int8_t)v; done[r] = 1; }
                            else rc = votes_put(o, y, x, v);
                        } else { /* :267-272 */
                            o->coin_votes++;
                            if (sm) rc = votes_put(o, y, x, v);
                            else { o->coin_flips++; rc = votes_put(o, y, x, o->sig[64 * (size_t)y] / 128); }
                        }
                        if (d > o->max_dist) o->max_dist = d;
                    }
                }
            }
        }
    }
    int cnt = 0;
    if (!rc) { /* :274-277 */
        for (int r = 0; r < o->R; ++r) {
            if (!done[r]) continue;
            int all = 1;
            for (int i = 0; i < o->wit_cnt[r]; ++i)
                if (o->famous[o->wit[(size_t)r * n + o->wit_order[(size_t)r * n + i]]] < 0) { all = 0; break; }
            if (all) {
                if (cnt < cap && new_rounds) new_rounds[cnt] = r;
                ++cnt;
            }
        }
        for (int r = 0; r < o->R; ++r) {
            if (!done[r]) continue;
            int all = 1;
            for (int i = 0; i < o->wit_cnt[r]; ++i)
                if (o->famous[o->wit[(size_t)r * n + o->wit_order[(size_t)r * n + i]]] < 0) { all = 0; break; }
            if (all) o->consensus[r] = 1;
        }
    }
    if (n_new) *n_new = cnt;
    free(done); free(hits); free(s_m);
    return rc;
}

/* ---- find_order ---- */
typedef struct { double ts; uint8_t key[64]; int32_t ev; } order_item;
static int order_cmp(const void* a, const void* b) {
    const order_item *x = a, *y = b;
    if (x->ts < y->ts) return -1;
    if (x->ts > y->ts) return 1;
    return memcmp(x->key, y->key, 64); /* big-endian 512-bit integers, swirld.py:281, 306 */
}
static int dbl_cmp(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* Node.find_order(new_c), swirld.py:280-311; rounds need not be sorted (sorted() at :283). */
int or_find_order(or_ctx* o, const int32_t* rounds_in, int n_rounds, int32_t* out_events,
                  int64_t cap, int64_t* n_out) {
    const int n = o->n;
    int32_t* rounds = malloc(sizeof(int32_t) * (n_rounds > 0 ? n_rounds : 1));
    int32_t* queue = malloc(sizeof(int32_t) * (size_t)(o->N > 0 ? o->N : 1));
    uint8_t* visited = calloc((size_t)(o->N > 0 ? o->N : 1), 1);
    order_item* items = malloc(sizeof(order_item) * (size_t)(o->N > 0 ? o->N : 1));
    double* times = malloc(sizeof(double) * n);
    int32_t* fw = malloc(sizeof(int32_t) * n);
    int32_t* sset = malloc(sizeof(int32_t) * n);
    if (!rounds || !queue || !visited || !items || !times || !fw || !sset) {
        free(rounds); free(queue); free(visited); free(items); free(times); free(fw); free(sset);
        return OR_ENOMEM;
    }
    memcpy(rounds, rounds_in, sizeof(int32_t) * n_rounds);
    for (int i = 1; i < n_rounds; ++i) { /* sorted(new_c) */
        int32_t v = rounds[i]; int j = i - 1;
        while (j >= 0 && rounds[j] > v) { rounds[j + 1] = rounds[j]; --j; }
        rounds[j + 1] = v;
    }
    int rc = OR_OK;
    int64_t produced = 0;
    for (int ir = 0; ir < n_rounds && !rc; ++ir) {
        const int r = rounds[ir];
        if (r < 0 || r >= o->R) { rc = OR_EKEY; break; }
        int nfw = 0; /* f_w, :284 */
        for (int i = 0; i < o->wit_cnt[r]; ++i) {
            int32_t w = o->wit[(size_t)r * n + o->wit_order[(size_t)r * n + i]];
            if (o->famous[w] < 0) { rc = OR_EKEY; break; }
            if (o->famous[w]) fw[nfw++] = w;
        }
        if (rc) break;
        uint8_t white[64]; /* :285 */
        memset(white, 0, 64);
        for (int i = 0; i < nfw; ++i)
            for (int b = 0; b < 64; ++b) white[b] ^= o->sig[64 * (size_t)fw[i] + b];
        /* bfs over tbd ancestors of the famous witnesses, :288-289 / utils.py:24-34 */
        int64_t qh = 0, qt = 0, nitems = 0;
        for (int i = 0; i < nfw; ++i)
            if (o->tbd[fw[i]] && !visited[fw[i]]) { visited[fw[i]] = 1; queue[qt++] = fw[i]; }
        const int64_t q0 = 0;
        while (qh < qt && !rc) {
            const int32_t x = queue[qh++];
            const int c = o->cr[x];
            int ns = 0; /* :291-292 */
            uint64_t stake_sum = 0;
            for (int i = 0; i < nfw; ++i) {
                const int32_t k = o->cansee[(size_t)fw[i] * n + c];
                if (k >= 0 && higher(o, k, x)) { sset[ns++] = fw[i]; stake_sum += o->stake[o->cr[fw[i]]]; }
            }
            if (2 * stake_sum > o->tot) { /* :293 */
                o->tbd[x] = 0;            /* :294 */
                for (int i = 0; i < ns; ++i) { /* :298-303 */
                    int32_t a = sset[i];
                    for (;;) {
                        const int32_t k = o->cansee[(size_t)a * n + c];
                        if (!(k >= 0 && higher(o, k, x) && o->sp[a] >= 0)) break;
                        a = o->sp[a];
                    }
                    times[i] = o->t[a];
                }
                qsort(times, ns, sizeof(double), dbl_cmp);
                if ((ns + 1) / 2 >= ns) { rc = OR_EINDEX; break; } /* :305 IndexError when len==1 */
                order_item* it = &items[nitems++];
                it->ts = .5 * (times[ns / 2] + times[(ns + 1) / 2]); /* :305 */
                it->ev = x;
                for (int b = 0; b < 64; ++b) it->key[b] = white[b] ^ o->sig[64 * (size_t)x + b];
            }
            /* successors: parents still in tbd (evaluated after the body ran, as the lazy
             * generator of utils.py:31 does) */
            if (o->sp[x] >= 0) {
                const int32_t ps[2] = {o->sp[x], o->op[x]};
                for (int j = 0; j < 2; ++j)
                    if (o->tbd[ps[j]] && !visited[ps[j]]) { visited[ps[j]] = 1; queue[qt++] = ps[j]; }
            }
        }
        for (int64_t i = q0; i < qt; ++i) visited[queue[i]] = 0;
        if (rc) break;
        qsort(items, nitems, sizeof(order_item), order_cmp); /* :306 */
        for (int64_t i = 0; i < nitems; ++i) {               /* :307-309 */
            o->transactions[o->n_tx++] = items[i].ev;
            if (out_events && produced < cap) out_events[produced] = items[i].ev;
            ++produced;
        }
    }
    if (n_out) *n_out = produced;
    free(rounds); free(queue); free(visited); free(items); free(times); free(fw); free(sset);
    return rc;
}

/* ---- getters ---- */
int64_t or_num_events(const or_ctx* o) { return o->N; }
int or_max_round(const or_ctx* o) { return o->R - 1; }
const int32_t* or_round_ptr(const or_ctx* o) { return o->round; }
const int32_t* or_height_ptr(const or_ctx* o) { return o->ht; }
const int32_t* or_cansee_ptr(const or_ctx* o) { return o->cansee; }
const int8_t* or_famous_ptr(const or_ctx* o) { return o->famous; }
const uint8_t* or_tbd_ptr(const or_ctx* o) { return o->tbd; }
const int32_t* or_transactions_ptr(const or_ctx* o) { return o->transactions; }
int64_t or_num_ordered(const or_ctx* o) { return o->n_tx; }
int or_get_witnesses(const or_ctx* o, int r0, int r1, int32_t* out) {
    for (int r = r0; r < r1; ++r)
        for (int c = 0; c < o->n; ++c)
            out[(size_t)(r - r0) * o->n + c] = (r >= 0 && r < o->R) ? o->wit[(size_t)r * o->n + c] : -1;
    return OR_OK;
}
/* members of round r in dict insertion order; returns the count */
int or_get_witness_order(const or_ctx* o, int r, int32_t* out_members) {
    if (r < 0 || r >= o->R) return 0;
    for (int i = 0; i < o->wit_cnt[r]; ++i) out_members[i] = o->wit_order[(size_t)r * o->n + i];
    return o->wit_cnt[r];
}
int or_get_consensus(const or_ctx* o, int r0, int r1, uint8_t* out) {
    for (int r = r0; r < r1; ++r) out[r - r0] = (r >= 0 && r < o->R) ? o->consensus[r] : 0;
    return OR_OK;
}
int or_get_vote(const or_ctx* o, int32_t voter_event, int32_t cand_event) {
    return votes_get(o, voter_event, cand_event);
}
int64_t or_num_votes(const or_ctx* o) { return (int64_t)o->votes_cnt; }
/* order-independent digest of the votes dict: sum over entries of hash(key, val) */
uint64_t or_votes_digest(const or_ctx* o) {
    uint64_t acc = 0;
    for (uint64_t i = 0; i < o->votes_cap; ++i)
        if (o->votes[i].key) acc += vhash(o->votes[i].key * 2 + (uint64_t)o->votes[i].val);
    return acc;
}
void or_get_counters(const or_ctx* o, int64_t* out8) {
    out8[0] = o->voter_evals; out8[1] = o->majority_evals; out8[2] = o->tally_inner; out8[3] = o->R;
    out8[4] = o->coin_votes; out8[5] = o->coin_flips; out8[6] = o->max_dist; out8[7] = 0;
}
