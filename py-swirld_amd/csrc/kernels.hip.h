// Device kernels of the MI355X-native virtual-voting hot path (gfx950 / CDNA4, wave64).
//
// All kernels are integer / bitmask work (no MFMA).  Layout in HBM:
//   L     [N][npad]  int32   can_see rows (swirld.py:69-72): latest event of member c
//                            among the ancestors-or-self of e, -1 = absent; npad = 64*NW
//   S     [N][NW]    u64     per event the member bitmask {c_ : round[L[e][c_]] == round[e]}
//                            (the inner test of swirld.py:211-214 / 250-252)
//   lo    [R][npad]  int32   lo[r][c] = first event of member c with round >= r (INF none)
//   wit   [R][npad]  int32   Node.witnesses[r][c] (-1 none)
//   Sw    [R][npad][NW] u64  per witness: members whose round r-1 witness it strongly sees
//   Mb    [MCAP][NW] u64     per-round scratch: threshold masks of the events in the band
//
// One wave (64 lanes) handles one event row: lane l owns columns l, l+64, ... (NW of them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
#define SW_INF 0x7fffffff

struct RState {
    int r;          // round whose promotion predicate is being evaluated
    int done;       // no member has an event of round >= r: the loop is finished
    int need_mask;  // the band mask table must be (re)built for round r
    int mlo, mhi;   // band of event indices covered by the mask table
    int iter;       // iterations executed in this call
    int n_unres;    // members still searching their first round-(r+1) event
    int max_round;  // valid when done
    int err;        // 1 = lo table capacity exceeded
    int pad_;
    u64 evals;      // tallies evaluated (live candidates)
    u64 far_hops;   // hop masks computed on the fly (outside the band)
};

struct FameCounters {
    u64 voter_evals;     // V  (swirld.py:247-254)
    u64 majority_evals;  // P2 (swirld.py:260)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------------
// Level buckets: events of one divide_rounds batch sorted by DAG height, so that all
// events of one level are independent (parents have strictly smaller height,
// swirld.py:117-120).
// ---------------------------------------------------------------------------------
__global__ void k_level_hist(const int* __restrict__ ht, int first, int K, int hmin, int* cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) atomicAdd(&cnt[ht[first + i] - hmin], 1);
}

// single block exclusive scan, cnt[0..n) -> start[0..n], cursor zeroed
__global__ void k_level_scan(const int* __restrict__ cnt, int n, int* start, int* cursor) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += nt) {
        int i = base + tid;
        int v = (i < n) ? cnt[i] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < nt; off <<= 1) {  // Hillis-Steele inclusive scan
            int t = (tid >= off) ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        int incl = s_part[tid];
        int carry = s_carry;
        if (i < n) { start[i] = carry + incl - v; cursor[i] = 0; }
        __syncthreads();
        if (tid == nt - 1) s_carry = carry + incl;
        __syncthreads();
    }
    if (tid == 0) start[n] = s_carry;
}

__global__ void k_level_scatter(const int* __restrict__ ht, const int* __restrict__ cr,
                                const int* __restrict__ sp, const int* __restrict__ op, int first, int K,
                                int hmin, const int* __restrict__ start, int* cursor, int4* desc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    int e = first + i;
    int lv = ht[e] - hmin;
    int slot = start[lv] + atomicAdd(&cursor[lv], 1);
    desc[slot] = make_int4(e, sp[e], op[e], cr[e]);
}

// ---------------------------------------------------------------------------------
// can_see rows (swirld.py:198, 203-205, 220).  Fork-free DAG: maxi()/higher() by height
// (swirld.py:170-184) equals max() of the dense indices on one creator's chain.
// Each workgroup owns CB columns of every row and sweeps the levels in order; levels are
// separated by a workgroup barrier only (no inter-workgroup dependency at all).
// ---------------------------------------------------------------------------------
template <int CB>
__global__ void __launch_bounds__(1024)
k_cansee_levels(const int4* __restrict__ desc, const int* __restrict__ lev_start, int nlev,
                int* L, int npad) {
    const int tid = threadIdx.x;
    const int col = blockIdx.x * CB + (tid % CB);
    const int sub = tid / CB;
    constexpr int EPB = 1024 / CB;  // events per pass
    int s = lev_start[0];
    for (int lv = 0; lv < nlev; ++lv) {
        const int t = lev_start[lv + 1];
        for (int i = s + sub; i < t; i += EPB) {
            const int4 d = desc[i];  // {e, sp, op, cr}
            int v = -1;
            if (d.y >= 0) {
                const int a = L[(size_t)d.y * npad + col];
                const int b = L[(size_t)d.z * npad + col];
                v = a > b ? a : b;
            }
            if (col == d.w) v = d.x;
            L[(size_t)d.x * npad + col] = v;
        }
        s = t;
        __syncthreads();  // rows of this level visible to the whole workgroup
    }
}

// ---------------------------------------------------------------------------------
// Round loop, step 1 (one workgroup, one thread per member): consume the results of the
// previous tally launch, advance the per-member cursors, commit lo[r+1] when every member
// is resolved, enter the next round that has work, emit the next candidate list.
// Round-synchronous form: round[e] >= r+1  <=>  SS_r(e), evaluated with the thresholds
// lo[r][.] (checked on the CPU in tests/model_bulk.py).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_resolve(RState* st, int npad, int K, int N, int MCAP, int Rcap,
          const int* __restrict__ chain_start, const int* __restrict__ chain_ev,
          int* lo, int* lopos, int* evalround, int* evalpos,
          int* lo_r, int* cur, int* unres, int* lo_next, int* pos_next,
          int* cand, const unsigned char* __restrict__ res) {
    __shared__ int s_min;
    const int c = threadIdx.x;
    if (st->done) return;
    int r = st->r;
    const int iter = st->iter;
    const int cs = chain_start[c];
    const int clen = chain_start[c + 1] - cs;
    int un = unres[c];
    if (iter > 0 && un) {
        int found = -1;
        for (int j = 0; j < K; ++j)
            if (res[c * K + j]) { found = j; break; }
        if (found >= 0) {
            lo_next[c] = cand[c * K + found];
            pos_next[c] = cur[c] + found;
            un = 0;
        } else if (cur[c] + K >= clen) {  // chain exhausted: no round-(r+1) event of c (yet)
            un = 0;
            evalround[c] = r;
            evalpos[c] = clen;
        } else {
            cur[c] += K;
        }
    }
    int nun = __syncthreads_count(un);
    int need_mask = 0, done = 0, err = 0, max_round = 0;
    int mlo = st->mlo, mhi = st->mhi;
    if (nun == 0) {
        if (iter > 0) {  // commit round r
            if (lo_next[c] != SW_INF) {
                lo[(size_t)(r + 1) * npad + c] = lo_next[c];
                lopos[(size_t)(r + 1) * npad + c] = pos_next[c];
            }
            r = r + 1;
        }
        for (;;) {  // enter the next round that has unresolved members
            if (r + 1 >= Rcap) { err = 1; done = 1; break; }
            const int lr = lo[(size_t)r * npad + c];
            const int nx = lo[(size_t)(r + 1) * npad + c];
            const int act = lr != SW_INF;
            un = 0;
            if (act && nx == SW_INF) {
                int start = lopos[(size_t)r * npad + c];
                if (evalround[c] == r && evalpos[c] > start) start = evalpos[c];
                cur[c] = start;
                un = start < clen;
            }
            const int nact = __syncthreads_count(act);
            if (nact == 0) { done = 1; max_round = r - 1; break; }
            nun = __syncthreads_count(un);
            if (nun > 0) {
                if (c == 0) s_min = SW_INF;
                __syncthreads();
                if (act) atomicMin(&s_min, lr);
                __syncthreads();
                mlo = s_min;
                mhi = (N - mlo > MCAP) ? mlo + MCAP : N;
                lo_r[c] = lr;
                lo_next[c] = SW_INF;
                need_mask = 1;
                break;
            }
            ++r;
        }
    }
    unres[c] = un;
    for (int j = 0; j < K; ++j) {
        const int p = cur[c] + j;
        cand[c * K + j] = (un && !done && p < clen) ? chain_ev[cs + p] : -1;
    }
    if (c == 0) {
        st->r = r; st->done = done; st->need_mask = need_mask; st->mlo = mlo; st->mhi = mhi;
        st->iter = iter + 1; st->n_unres = nun;
        if (done) st->max_round = max_round;
        if (err) st->err = 1;
    }
}

// ---------------------------------------------------------------------------------
// Round loop, step 2: threshold masks of the band events.  Mb[k-mlo] bit c_ =
// (L[k][c_] >= lo[r][c_]), i.e. "the latest event of c_ that k sees has round >= r".
// One wave per band event, NW ballots.
// ---------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(256)
k_band_masks(const RState* __restrict__ st, const int* __restrict__ L, const int* __restrict__ cr,
             const int* __restrict__ lo_r, u64* Mb, int npad) {
    if (st->done || !st->need_mask) return;
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int mlo = st->mlo, mhi = st->mhi;
    int thr[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) thr[j] = lo_r[j * 64 + lane];
    for (int k = mlo + wave; k < mhi; k += nwaves) {
        if (k < lo_r[cr[k]]) continue;  // round[k] < r: never a valid hop this round
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int v = L[(size_t)k * npad + j * 64 + lane];
            const u64 b = __ballot(v >= thr[j]);
            if (lane == 0) Mb[(size_t)(k - mlo) * NW + j] = b;
        }
    }
}

// ---------------------------------------------------------------------------------
// The strongly-sees tally (swirld.py:208-216 and 247-254), one wave per evaluated event.
//   hits[c_] = sum over hops c of stake[c] * mask(hop_c)[c_],  then compare with 2T/3.
// Lane l accumulates columns l, l+64, ...; hop masks are staged per 64-hop chunk in LDS
// (word-major, so the staging writes are conflict-free and the reads are broadcasts).
// ---------------------------------------------------------------------------------
template <int NW, bool UNIT>
__device__ __forceinline__ void tally_chunk(const u64 vm, const int j, const uint32_t* hm32,
                                            const uint32_t* __restrict__ stake, uint32_t (&hits)[NW],
                                            const int lane) {
    u64 m = vm;
    const int half = lane >> 5, bit = lane & 31;
    while (m) {
        const int h = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t w = UNIT ? 1u : stake[j * 64 + h];
#pragma unroll
        for (int jj = 0; jj < NW; ++jj) {
            const uint32_t word = hm32[(jj * 64 + h) * 2 + half];
            hits[jj] += ((word >> bit) & 1u) * w;
        }
    }
}

// Round loop, step 3: evaluate SS_r(e) for the candidate list.
template <int NW, bool UNIT>
__global__ void __launch_bounds__(256)
k_tally_candidates(RState* st, const int* __restrict__ cand, const int* __restrict__ L,
                   const int* __restrict__ cr, const int* __restrict__ sp,
                   const int* __restrict__ lo_r, const u64* __restrict__ Mb,
                   const uint32_t* __restrict__ stake, uint32_t tot2, unsigned char* res, int npad) {
    __shared__ u64 s_hm[4][NW * 64];
    if (st->done) return;
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wib;
    const int e = cand[w];
    if (e < 0) {
        if (lane == 0) res[w] = 0;
        return;
    }
    const int mlo = st->mlo, mhi = st->mhi;
    u64* hm = s_hm[wib];
    const int ce = cr[e], spe = sp[e];
    int thr[NW];
    int P[NW];
    uint32_t hits[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        thr[j] = lo_r[j * 64 + lane];
        int v = L[(size_t)e * npad + j * 64 + lane];
        if (j * 64 + lane == ce) v = spe;  // the row BEFORE the self overwrite (Q4)
        P[j] = v;
        hits[j] = 0;
    }
    u64 nfar = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int k = P[j];
        const bool valid = k >= thr[j];  // hop has round >= r (k == -1 fails: thr >= 0)
        const bool inband = valid && k < mhi;  // k >= mlo holds for every valid hop
        if (inband) {
            const u64* src = Mb + (size_t)(k - mlo) * NW;
#pragma unroll
            for (int jj = 0; jj < NW; ++jj) hm[jj * 64 + lane] = src[jj];
        }
        u64 far = __ballot(valid && !inband);
        nfar += __popcll(far);
        while (far) {  // rare: hop outside the band, build its mask from its row
            const int h = __ffsll((long long)far) - 1;
            far &= far - 1;
            const int kf = __shfl(k, h);
#pragma unroll
            for (int jj = 0; jj < NW; ++jj) {
                const int v = L[(size_t)kf * npad + jj * 64 + lane];
                const u64 b = __ballot(v >= thr[jj]);
                if (lane == 0) hm[jj * 64 + h] = b;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tally_chunk<NW, UNIT>(__ballot(valid), j, (const uint32_t*)hm, stake, hits, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) cnt += __popcll(__ballot(3u * hits[j] > tot2));
    if (lane == 0) {
        res[w] = (3u * cnt > tot2) ? 1 : 0;  // count of members vs the STAKE threshold (Q2)
        atomicAdd(&st->evals, 1ull);
        if (nfar) atomicAdd(&st->far_hops, nfar);
    }
}

// ---------------------------------------------------------------------------------
// Finalize: round numbers (swirld.py:217-219) and sees-masks for the events of the batch.
// round[e] = max r with lo[r][creator(e)] <= e  (binary search of one column of lo).
// ---------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(256)
k_finalize_events(const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ lo,
                  int R, int first, int K, int* round, u64* S, int npad) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < K; i += nwaves) {
        const int e = first + i;
        const int c = cr[e];
        int a = 0, b = R - 1;  // lo[0][c] <= e always (chain start)
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (lo[(size_t)mid * npad + c] <= e) a = mid; else b = mid - 1;
        }
        if (lane == 0) round[e] = a;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int v = L[(size_t)e * npad + j * 64 + lane];
            const u64 bm = __ballot(v >= lo[(size_t)a * npad + j * 64 + lane]);
            if (lane == 0) S[(size_t)e * NW + j] = bm;
        }
    }
}

// Witness table (swirld.py:197, 221-222): member c has a witness in round r iff its first
// event of round >= r has round exactly r, i.e. lo[r+1][c] > lo[r][c].
__global__ void k_witness_table(const int* __restrict__ lo, int R, int r0, int npad, int* wit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = (R - r0) * npad;
    if (i >= total) return;
    const int r = r0 + i / npad, c = i % npad;
    const int a = lo[(size_t)r * npad + c];
    const int b = lo[(size_t)(r + 1) * npad + c];  // row R exists and is INF
    wit[(size_t)r * npad + c] = (a != SW_INF && b > a) ? a : -1;
}

// ---------------------------------------------------------------------------------
// decide_fame, part 1 (swirld.py:247-254): for every witness y of round r >= 1 the set s of
// round r-1 witnesses it strongly sees, as a member bitmask.  Hops are the entries of the
// FINAL row of y (own entry = y itself, round r, hence excluded: Q4/Q6) whose round is
// exactly r-1; their masks are the stored S rows.
// ---------------------------------------------------------------------------------
template <int NW, bool UNIT>
__global__ void __launch_bounds__(256)
k_voter_masks(const int* __restrict__ wit, const int* __restrict__ L, const int* __restrict__ lo,
              const u64* __restrict__ S, const uint32_t* __restrict__ stake, uint32_t tot2,
              int r0, int R, int npad, u64* Sw, FameCounters* fc) {
    __shared__ u64 s_hm[4][NW * 64];
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wib;
    const int total = (R - r0) * npad;
    if (w >= total) return;
    const int r = r0 + w / npad, c = w % npad;
    const int y = wit[(size_t)r * npad + c];
    u64* out = Sw + ((size_t)r * npad + c) * NW;
    if (y < 0 || r < 1) {
        if (lane < NW) out[lane] = 0;
        return;
    }
    u64* hm = s_hm[wib];
    uint32_t hits[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) hits[j] = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int col = j * 64 + lane;
        const int k = L[(size_t)y * npad + col];
        // round[k] == r-1  <=>  lo[r-1][col] <= k < lo[r][col]
        const bool valid = k >= lo[(size_t)(r - 1) * npad + col] && k < lo[(size_t)r * npad + col];
        if (valid) {
            const u64* src = S + (size_t)k * NW;
#pragma unroll
            for (int jj = 0; jj < NW; ++jj) hm[jj * 64 + lane] = src[jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tally_chunk<NW, UNIT>(__ballot(valid), j, (const uint32_t*)hm, stake, hits, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const bool in_s = 3u * hits[j] > tot2 && wit[(size_t)(r - 1) * npad + j * 64 + lane] >= 0;
        const u64 b = __ballot(in_s);
        if (lane == 0) out[j] = b;
    }
    if (lane == 0) atomicAdd(&fc->voter_evals, 1ull);
}

// ---------------------------------------------------------------------------------
// decide_fame, part 2 (swirld.py:256-277), candidate-major: one workgroup per candidate
// round r, one thread per candidate witness x = wit[r][cx].  For d = 1, 2, ... the votes of
// the round r+d witnesses on x form a member bitmask V; a voter's majority() over the
// witnesses it strongly sees is popcount(Sw & V) vs popcount(Sw & ~V).  "First decider
// wins" (swirld.py:235, 263) = the deciding voter with the smallest event index (dict
// order of Node.witnesses[r] is registration order = ascending index).
// ---------------------------------------------------------------------------------
template <int NW, bool UNIT>
__global__ void __launch_bounds__(1024)
k_elections(const int* __restrict__ wit, const u64* __restrict__ Sw, const unsigned char* __restrict__ coin,
            const uint32_t* __restrict__ stake, uint32_t tot2, int coin_period, int max_c, int R,
            int npad, signed char* fam, unsigned char* cons, unsigned char* newc, FameCounters* fc) {
    const int r = max_c + blockIdx.x;
    const int cx = threadIdx.x;
    if (cons[r]) return;
    const int x = wit[(size_t)r * npad + cx];
    bool active = x >= 0 && fam[(size_t)r * npad + cx] < 0;
    u64 V[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) V[j] = 0;
    int any_decided = 0;
    u64 p2 = 0;
    for (int d = 1; r + d < R; ++d) {
        if (!__syncthreads_or(active)) break;
        const int rv = r + d;
        const int* wv_row = wit + (size_t)rv * npad;
        const u64* sw_row = Sw + (size_t)rv * npad * NW;
        u64 Vn[NW];
        int best_idx = SW_INF, best_v = 0, nvoters = 0;
        const bool coin_round = (d % coin_period) == 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u64 acc = 0;
            for (int ci = 0; ci < 64; ++ci) {
                const int c = j * 64 + ci;
                const int wv = wv_row[c];  // uniform
                if (wv < 0) continue;
                ++nvoters;
                const u64* m = sw_row + (size_t)c * NW;
                int bitv;
                if (d == 1) {
                    bitv = (int)((m[cx >> 6] >> (cx & 63)) & 1ull);  // x in s (swirld.py:258)
                } else {
                    uint32_t yes = 0, tot = 0;
                    if (UNIT) {
#pragma unroll
                        for (int jj = 0; jj < NW; ++jj) {
                            const u64 mm = m[jj];
                            yes += __popcll(mm & V[jj]);
                            tot += __popcll(mm);
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < NW; ++jj) {
                            u64 mm = m[jj];
                            while (mm) {
                                const int b = __ffsll((long long)mm) - 1;
                                mm &= mm - 1;
                                const uint32_t sk = stake[jj * 64 + b];
                                tot += sk;
                                if ((V[jj] >> b) & 1ull) yes += sk;
                            }
                        }
                    }
                    const uint32_t no = tot - yes;
                    const int v = !(no > yes);  // majority(): tie -> True (swirld.py:24-27)
                    const uint32_t t = v ? yes : no;
                    const bool sm = 3u * t > tot2;
                    if (!coin_round) {
                        if (sm && wv < best_idx) { best_idx = wv; best_v = v; }
                        bitv = v;
                    } else {
                        bitv = sm ? v : (int)coin[wv];  // swirld.py:267-272
                    }
                }
                acc |= (u64)bitv << ci;
            }
            Vn[j] = acc;
        }
        if (active && d >= 2) {
            if (!coin_round && best_idx != SW_INF) {
                fam[(size_t)r * npad + cx] = (signed char)best_v;  // swirld.py:263
                active = false;
                any_decided = 1;
                int le = 0;  // voters after the first decider never evaluate x
                for (int c = 0; c < npad; ++c) {
                    const int wv = wv_row[c];
                    le += (wv >= 0 && wv <= best_idx);
                }
                p2 += le;
            } else {
                p2 += nvoters;
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) V[j] = Vn[j];
    }
    const int x_open = x >= 0 && fam[(size_t)r * npad + cx] < 0;
    const int n_open = __syncthreads_count(x_open);
    const int dec = __syncthreads_or(any_decided);
    if (cx == 0 && dec && n_open == 0) {  // swirld.py:274-277
        newc[r] = 1;
        cons[r] = 1;
    }
    if (p2) atomicAdd(&fc->majority_evals, p2);
}

__global__ void k_fill_i32(int* p, size_t n, int v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
