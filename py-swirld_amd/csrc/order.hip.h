// find_order (swirld.py:280-311), fork-free form, on the device (included by kernels.hip.h).
//
// Because "w sees x" (swirld.py:291-292: can_see[w][c] is at least as high as x on creator c's chain) is
// inherited by every ancestor of x, the set of already ordered events is ancestor-closed, i.e. a PREFIX of
// every member's self-parent chain, and the events a decided round r newly orders on chain c are the chain
// positions [ordered[c], q[r][c]) where q is the first position whose event is no longer seen by famous
// witnesses holding more than half of the stake (:293).
//
// Round 6: the whole call is driven from device tables — the host uploads the list of rounds and reads back
// one small block (events per round entry, error words, the members' new ordered prefix) before it sizes the
// bulk kernels; nothing else of the call visits it until the sorted order comes back.
//   k_order_prep      fwm[ri][m]   = the famous witness of member m in round entry ri (-1 none)      :284
//   k_order_bounds    q[ri][c]     = end of the positions of chain c the entry's witnesses accept     :288-293
//   k_order_runmax    ordat[ri][c] = positions of c ordered before entry ri (running maximum of q), segment lengths
//   k_order_rowscan / k_order_offsets   offsets of the segments in the round-major list of ordered events
//   k_order_segments  the ordered events themselves, round-major
//   bulk calls:  k_order_group (per group of entries: tiles per chain, walk ranges) -> k_order_walk (first-descendant
//                table, written chain by chain) -> k_order_median (samples + pseudo-median per ordered event)
//   small calls: k_order_times (binary searches)
//   k_order_white, k_order_sort, k_order_sort_big   whitening key, final order inside a round                 :285, 306
#pragma once

struct OrderInfo {
    long long n_acc;      // events the call orders
    int undecided_ri;     // smallest round entry with an undecided witness (KeyError at swirld.py:284), INT_MAX none
    int index_err;        // an ordered event has a single sample (IndexError at swirld.py:305)
    int pad_[4];
};

// ---------------------------------------------------------------------------------
// :284  f_w of every round entry, as a member-indexed row (the creator of a famous witness is its column)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_order_prep(const int* __restrict__ rounds, const int* __restrict__ wit, const signed char* __restrict__ fam,
             const int* __restrict__ seq, int n, int npad, int* __restrict__ fwm, int* __restrict__ fwseq, OrderInfo* info) {
    const int ri = blockIdx.x, m = threadIdx.x;
    const int r = rounds[ri];
    int out = -1;
    if (m < n) {
        const int w = wit[(size_t)r * npad + m];
        const int f = fam[(size_t)r * npad + m];
        if (w >= 0 && f < 0) atomicMin(&info->undecided_ri, ri);
        if (w >= 0 && f > 0) out = w;
    }
    fwm[(size_t)ri * npad + m] = out;
    fwseq[(size_t)ri * npad + m] = out >= 0 ? seq[out] : -1;   // its position on m's chain: what k_order_median compares with
}

// ---------------------------------------------------------------------------------
// q[ri][c]: whether position p of chain c is accepted depends on its event x only through "how much stake of
// the famous witnesses has L[w][c] >= x": every famous witness accepts the positions up to its own latest-seen
// event of c, so the boundary lies between the smallest and the largest of those entries — about a round of c's
// chain, 4 probes.  Workgroup = (round entry, 64 columns); the witnesses are dealt to G = npad / 64 waves, every
// thread keeps its <= 64 entries in REGISTERS (one pass of independent row reads; the probes of the search are
// register compares whose partial sums meet in LDS).  Round 5: one thread per column walked ~190 rows five
// times, 0.4-0.65 ms per 280 rounds on 280 workgroups.
// ---------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(64 * G)
k_order_bounds(const int* __restrict__ fwm, const int* __restrict__ L, const int* __restrict__ seq,
               const uint32_t* __restrict__ stake, uint32_t tot, const int* __restrict__ chain_start,
               const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev, int n, int safe_row, int* __restrict__ q) {
    constexpr int npad = 64 * G;
    __shared__ int s_fw[npad];
    __shared__ uint32_t s_stk[npad];   // stake of the member's famous witness, 0 where it has none
    __shared__ int s_min[G][64], s_max[G][64];
    __shared__ uint32_t s_sum[G][64];
    const int ri = blockIdx.x, tid = threadIdx.x;
    const int g = tid >> 6, cl = tid & 63;
    const int c = blockIdx.y * 64 + cl;
    {
        const int w = fwm[(size_t)ri * npad + tid];
        s_fw[tid] = w;
        s_stk[tid] = w >= 0 ? stake[tid] : 0u;
    }
    __syncthreads();
    int val[64];            // L[w][c] of this thread's witnesses (members g, g + G, ...), -2 where the member has none
    int vmin = 0x7fffffff, vmax = -1;
    uint32_t s_all = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const int w = s_fw[g + k * G];
        val[k] = L[(size_t)(w < 0 ? safe_row : w) * npad + c];   // (unconditional: the loads of the pass are independent; a member without a famous witness reads a row that is resident under the windowed table)
        val[k] = w >= 0 ? val[k] : -2;
    }
#pragma unroll
    for (int k = 0; k < 64; ++k)
        if (val[k] > -2) {
            vmin = val[k] < vmin ? val[k] : vmin;
            vmax = val[k] > vmax ? val[k] : vmax;
            s_all += s_stk[g + k * G];
        }
    s_min[g][cl] = vmin; s_max[g][cl] = vmax; s_sum[g][cl] = s_all;
    __syncthreads();
    vmin = 0x7fffffff; vmax = -1; s_all = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        vmin = s_min[k][cl] < vmin ? s_min[k][cl] : vmin;
        vmax = s_max[k][cl] > vmax ? s_max[k][cl] : vmax;
        s_all += s_sum[k][cl];
    }
    const int cs = chain_start[c], clen = chain_cnt[c];
    // invariant: positions < a are accepted, positions >= b are not
    int a = 0, b = 0;
    if (2u * s_all > tot && vmax >= 0 && clen > 0) {   // (else: not even the first event of c is accepted)
        a = vmin >= 0 ? seq[vmin] + 1 : 0;   // seen by every famous witness
        b = seq[vmax] + 1;                   // beyond the latest one any of them sees: by none
        if (b > clen) b = clen;
        if (a > b) a = b;
    }
    while (true) {
        const bool act = a < b;
        if (!__syncthreads_or(act)) break;
        const int mid = (a + b) >> 1;
        const int x = act ? chain_ev[cs + mid] : 0x7fffffff;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) sum += val[k] >= x ? s_stk[g + k * G] : 0u;
        s_sum[g][cl] = sum;
        __syncthreads();
        sum = 0;
#pragma unroll
        for (int k = 0; k < G; ++k) sum += s_sum[k][cl];
        if (act) { if (2u * sum > tot) a = mid + 1; else b = mid; }
    }
    if (g == 0) q[(size_t)ri * npad + c] = a;
}

// ---------------------------------------------------------------------------------
// tbd (swirld.py:53-54, 288-289) as chain prefixes: entry ri orders the positions [ordat[ri][c], q[ri][c]) of chain c when
// q exceeds what the entries before it (and earlier calls: ordpos) have ordered.  One thread per member walks the entries.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_order_runmax(const int* __restrict__ q, const int* __restrict__ ordpos, int nr, int npad,
               int* __restrict__ ordat, int* __restrict__ seg_start, int* __restrict__ seg_len, int* __restrict__ ord_new) {
    const int m = threadIdx.x;
    int ord = ordpos[m];
#pragma unroll 8
    for (int ri = 0; ri < nr; ++ri) {
        const int hi = q[(size_t)ri * npad + m];
        ordat[(size_t)ri * npad + m] = ord;
        const bool grows = hi > ord;
        seg_start[(size_t)ri * npad + m] = grows ? ord : -1;
        seg_len[(size_t)ri * npad + m] = grows ? hi - ord : 0;
        ord = grows ? hi : ord;
    }
    ordat[(size_t)nr * npad + m] = ord;
    ord_new[m] = ord;
}

// exclusive scan of one row of npad <= 1024 values held one per thread (every thread of the workgroup calls it)
__device__ __forceinline__ int block_excl_scan(int v, int* s_buf /*[1024]*/, int npad, int* total) {
    const int tid = threadIdx.x;
    s_buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < npad; off <<= 1) {
        const int add = tid >= off ? s_buf[tid - off] : 0;
        __syncthreads();
        s_buf[tid] += add;
        __syncthreads();
    }
    const int incl = s_buf[tid];
    *total = s_buf[npad - 1];
    __syncthreads();
    return incl - v;
}

// offsets of the segments inside their round entry (member order) and the entry's total
__global__ void __launch_bounds__(1024)
k_order_rowscan(const int* __restrict__ seg_len, int npad, int* __restrict__ seg_off, int* __restrict__ rowsum) {
    __shared__ int s_buf[1024];
    const int ri = blockIdx.x, m = threadIdx.x;
    int total;
    const int ex = block_excl_scan(seg_len[(size_t)ri * npad + m], s_buf, npad, &total);
    seg_off[(size_t)ri * npad + m] = ex;
    if (m == 0) rowsum[ri] = total;
}

// acc_off[ri] = events ordered by the entries before ri (round-major list), acc_off[nr] = all of them
__global__ void __launch_bounds__(1024)
k_order_offsets(const int* __restrict__ rowsum, int nr, long long* __restrict__ acc_off, OrderInfo* info) {
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nr + 1023) / 1024;
    const int a = tid * per, b = a + per < nr ? a + per : nr;
    long long s = 0;
    for (int i = a; i < b; ++i) s += rowsum[i];
    s_part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long add = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    long long run = s_part[tid] - s;
    for (int i = a; i < b; ++i) { acc_off[i] = run; run += rowsum[i]; }
    if (tid == 1023) { acc_off[nr] = s_part[1023]; info->n_acc = s_part[1023]; }
}

// The events a call newly orders, round by round (swirld.py:288-293 as chain segments): segment (round entry i,
// member m) = chain positions [start, start + len) of m, written at acc_off[i] + seg_off[i][m] of the round-major list.
__global__ void k_order_segments(const int* __restrict__ seg_start, const int* __restrict__ seg_len, const int* __restrict__ seg_off,
                                 const long long* __restrict__ acc_off, const int* __restrict__ chain_start,
                                 const int* __restrict__ chain_ev, int npad, int total, int* __restrict__ acc_ev, int* __restrict__ acc_ri) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int st = seg_start[i];
    if (st < 0) return;
    const int m = i % npad, ri = i / npad;
    const int len = seg_len[i];
    const long long off = acc_off[ri] + seg_off[i];
    const int* src = chain_ev + chain_start[m] + st;
    for (int k = 0; k < len; ++k) { acc_ev[off + k] = src[k]; acc_ri[off + k] = ri; }
}

// ---------------------------------------------------------------------------------
// Order statistics len/2 and (len+1)/2 of a wave's samples (swirld.py:304-305), E per lane, `valid[r]` = the lanes whose
// v[r] is a sample.  Selection by counting instead of sorting (round 5: a bitonic sort in registers, 21 cross-lane stages of
// ds_bpermute pairs per event): a pivot taken from the candidates splits them by two ballots per register; the counts are
// scalar popcounts, the candidate sets scalar masks, so an iteration is 2 E compares and a handful of scalar instructions,
// and ~2 ln(len) iterations find the statistic.  The second statistic is the first one or its successor.
// Every lane of the wave must be active; len >= 2.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const long long b = __double_as_longlong(x);
    const int lo_ = __builtin_amdgcn_readlane((int)b, l), hi_ = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi_ << 32) | (unsigned)lo_);
}

__device__ __forceinline__ double wave_min_f64(double v) {
#define SW_MIN64_STEP(CTRL)                                                                                      \
    {                                                                                                            \
        const long long b_ = __double_as_longlong(v);                                                            \
        const int lo_ = dpp_i32<CTRL>((int)b_), hi_ = dpp_i32<CTRL>((int)(b_ >> 32));                            \
        const double o_ = __longlong_as_double(((long long)hi_ << 32) | (unsigned)lo_);                          \
        v = o_ < v ? o_ : v;                                                                                     \
    }
    SW_MIN64_STEP(DPP_XOR1) SW_MIN64_STEP(DPP_XOR2) SW_MIN64_STEP(DPP_HALF_MIRROR) SW_MIN64_STEP(DPP_MIRROR)
#undef SW_MIN64_STEP
    {
        const long long b_ = __double_as_longlong(v);
        int al, bl, ah, bh;
        rows_pair16((int)b_, al, bl); rows_pair16((int)(b_ >> 32), ah, bh);
        const double x = __longlong_as_double(((long long)ah << 32) | (unsigned)al), y = __longlong_as_double(((long long)bh << 32) | (unsigned)bl);
        v = y < x ? y : x;
    }
    {
        const long long b_ = __double_as_longlong(v);
        int al, bl, ah, bh;
        rows_pair32((int)b_, al, bl); rows_pair32((int)(b_ >> 32), ah, bh);
        const double x = __longlong_as_double(((long long)ah << 32) | (unsigned)al), y = __longlong_as_double(((long long)bh << 32) | (unsigned)bl);
        v = y < x ? y : x;
    }
    return v;
}

// popcount(mask) + acc on the VALU (the same value in every lane): the scalar unit, one per CU, is what a wave-wide
// selection by ballots runs out of first (measured: 716 scalar against 320 vector instructions per event with s_bcnt1)
__device__ __forceinline__ int bcnt_acc(const u64 mask, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "s"((int)(unsigned)mask), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "s"((int)(unsigned)(mask >> 32)), "v"(r));
    return r;
}

// the value lane ^ J holds, or (J = 16, 32: v_permlane16/32_swap) the pair {own, partner} in an order that depends on the lane
template <int J>
__device__ __forceinline__ void xor_pair_f64(const double v, double& a, double& b) {
    const long long bits = __double_as_longlong(v);
    const int lo_ = (int)bits, hi_ = (int)(bits >> 32);
    int al = lo_, ah = hi_, bl, bh;
    if constexpr (J == 1) { bl = dpp_i32<DPP_XOR1>(lo_); bh = dpp_i32<DPP_XOR1>(hi_); }
    else if constexpr (J == 2) { bl = dpp_i32<DPP_XOR2>(lo_); bh = dpp_i32<DPP_XOR2>(hi_); }
    else if constexpr (J == 4) {   // i -> 7 - i inside its 8, then reversed inside its 4: i ^ 4
        bl = dpp_i32<0x1B>(dpp_i32<DPP_HALF_MIRROR>(lo_)); bh = dpp_i32<0x1B>(dpp_i32<DPP_HALF_MIRROR>(hi_));
    } else if constexpr (J == 8) { bl = dpp_i32<DPP_ROR8>(lo_); bh = dpp_i32<DPP_ROR8>(hi_); }
    else if constexpr (J == 16) { rows_pair16(lo_, al, bl); rows_pair16(hi_, ah, bh); }
    else { rows_pair32(lo_, al, bl); rows_pair32(hi_, ah, bh); }
    a = __longlong_as_double(((long long)ah << 32) | (unsigned)al);
    b = __longlong_as_double(((long long)bh << 32) | (unsigned)bl);
}

template <int K, int J>
__device__ __forceinline__ void bitonic_step(double& v, const int lane) {
    double a, b;
    xor_pair_f64<J>(v, a, b);
    const bool keepmin = ((lane & J) == 0) == ((lane & K) == 0);
    const double mn = a < b ? a : b, mx = a < b ? b : a;
    v = keepmin ? mn : mx;
    if constexpr (J > 1) bitonic_step<K, J / 2>(v, lane);
}

// 64 doubles, one per lane (no NaNs), ascending by lane: 21 compare-exchange steps, VALU only (DPP inside a row of 16 lanes,
// v_permlane16/32_swap across the rows)
__device__ __forceinline__ double wave_sort64_f64(double v, const int lane) {
    bitonic_step<2, 1>(v, lane);
    bitonic_step<4, 2>(v, lane);
    bitonic_step<8, 4>(v, lane);
    bitonic_step<16, 8>(v, lane);
    bitonic_step<32, 16>(v, lane);
    bitonic_step<64, 32>(v, lane);
    return v;
}

// v[r]: the samples, NaN where lane / register r holds none (a NaN compares false both ways: no mask is needed on the
// ballots).  Pivots from the candidates narrow them down until they fit ONE register (<= 64: 2-4 full-width steps for ~190
// samples); those are compacted through `scratch` (64 doubles of LDS per wave) and sorted across the lanes — the remaining
// ~9 full-width steps of a plain quickselect were most of the kernel (profiles/r06p_order_pmc.txt: 610 scalar + 510 vector
// instructions per event).
template <int E>
__device__ __forceinline__ double wave_pseudo_median(const double (&v)[E], const u64 (&valid)[E], const int len, double* scratch) {
    const int k1 = len / 2, k2 = (len + 1) / 2;
    const int lane = lane_id();
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    u64 cand[E];
#pragma unroll
    for (int r = 0; r < E; ++r) cand[r] = valid[r];
    int below = 0, hi_cnt = len;   // samples below the candidates; samples below the candidates' upper bound
    double hi_val = inf;           // the smallest sample above the candidates
    double first = 0.0, second = 0.0;
    bool done = false;
    while (hi_cnt - below > 64) {
        // the pivot: the first candidate in (register, lane) order — samples come in member order, unrelated to their values
        int rr = 0;
        u64 cm = cand[0];
#pragma unroll
        for (int r = 1; r < E; ++r)
            if (cm == 0) { cm = cand[r]; rr = r; }
        const int pl = __ffsll((long long)cm) - 1;
        double pv = 0.0;
#pragma unroll
        for (int r = 0; r < E; ++r)
            if (rr == r) pv = readlane_f64(v[r], pl);
        u64 mlt[E], mle[E];
        int a_lt = 0, a_le = 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            mlt[r] = __ballot(v[r] < pv);
            mle[r] = __ballot(v[r] <= pv);
            a_lt = bcnt_acc(mlt[r], a_lt);
            a_le = bcnt_acc(mle[r], a_le);
        }
        const int c_lt = __builtin_amdgcn_readfirstlane(a_lt), c_le = __builtin_amdgcn_readfirstlane(a_le);
        if (k1 < c_lt) {
#pragma unroll
            for (int r = 0; r < E; ++r) cand[r] &= mlt[r];
            hi_cnt = c_lt;
            hi_val = pv;
        } else if (k1 < c_le) {   // the pivot is the first statistic; the second is the pivot or its successor
            first = pv;
            second = pv;
            if (k2 >= c_le) {
                double mn = inf;
#pragma unroll
                for (int r = 0; r < E; ++r) mn = (v[r] > pv && v[r] < mn) ? v[r] : mn;
                second = wave_min_f64(mn);
            }
            done = true;
            break;
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r) cand[r] &= ~mle[r];
            below = c_le;
        }
    }
    if (!done) {
        const int ncand = hi_cnt - below;
        int base = 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int rank = base + __builtin_amdgcn_mbcnt_hi((unsigned)(cand[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cand[r], 0));
            if ((cand[r] >> lane) & 1ull) scratch[rank] = v[r];
            base += __popcll(cand[r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double x = lane < ncand ? scratch[lane] : inf;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        x = wave_sort64_f64(x, lane);
        first = readlane_f64(x, k1 - below);
        second = k2 - below < ncand ? readlane_f64(x, k2 - below) : hi_val;
    }
    return .5 * (first + second);
}

// ---------------------------------------------------------------------------------
// consensus timestamp of every newly ordered event (swirld.py:295-305), SMALL calls: one wave per event.
// For each famous witness w that sees x, the sample is the timestamp of the first self-ancestor of w that does NOT
// see x, or of w's creator's root (Q11) = the predecessor, on the chain of w's creator m, of the first event of m
// that sees x (binary search: the latest-seen entry for x's creator is monotone along a chain).
// ---------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(256)
k_order_times(const int* __restrict__ acc_ev, const int* __restrict__ acc_ri, int n_acc,
              const int* __restrict__ fwm, const int* __restrict__ L,
              const int* __restrict__ cr, const int* __restrict__ seq, const double* __restrict__ t,
              const int* __restrict__ chain_start, const int* __restrict__ chain_ev, const int* __restrict__ ord_pos,
              double* __restrict__ ts, OrderInfo* info) {
    constexpr int npad = 64 * NW;
    __shared__ double s_scratch[4][64];
    const int lane = lane_id();
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n_acc) return;
    const int x = acc_ev[idx];
    const int ri = acc_ri[idx];
    const int c = cr[x];
    double v[NW];
    unsigned seesbits = 0;   // (per lane, in a vector register: wave masks kept across the divergent searches below were spilled
                             // by the compiler at 16 mask words and came back wrong — the ballots are taken behind the searches)
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int m = lane + 64 * r;
        const int w = fwm[(size_t)ri * npad + m];
        bool sees = false;
        double sample = 0.0;
        if (w >= 0 && L[(size_t)w * npad + c] >= x) {
            sees = true;
            const int cs = chain_start[m];
            // first position p in [ord_pos[m], seq[w]] with L[chain_m[p]][c] >= x: the ordered prefix of m's chain
            // cannot see an unordered x (the ordered set is ancestor-closed), so the search never touches its
            // rows — which is what lets old can_see rows be evicted (windowed table)
            int lo_ = ord_pos[m], hi = seq[w];
            if (lo_ > hi) lo_ = hi;
            while (lo_ < hi) {
                const int mid = (lo_ + hi) >> 1;
                if (L[(size_t)chain_ev[cs + mid] * npad + c] >= x) hi = mid; else lo_ = mid + 1;
            }
            sample = t[chain_ev[cs + (lo_ > 0 ? lo_ - 1 : 0)]];
        }
        v[r] = sees ? sample : __longlong_as_double(0x7ff8000000000000ll);
        seesbits |= sees ? 1u << r : 0u;
    }
    u64 valid[NW];
    int len = 0;
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        valid[r] = __ballot((seesbits >> r) & 1u);
        len += __popcll(valid[r]);
    }
    if ((len + 1) / 2 >= len) {  // IndexError in the reference (len == 1, only with unequal stakes)
        if (lane == 0) { atomicExch(&info->index_err, 1); ts[idx] = 0.0; }
        return;
    }
    const double med = wave_pseudo_median<NW>(v, valid, len, s_scratch[threadIdx.x >> 6]);
    if (lane == 0) ts[idx] = med;
}

// ---------------------------------------------------------------------------------
// BULK calls (a call that orders many events): instead of one binary search of ~10 scattered 4-byte gathers per
// (event, famous witness) pair, ONE streaming pass over the can_see rows answers the transposed question "which is
// the first event of member m that sees x?" for every x being ordered:
//   FD[m][x] = min { y on m's chain : can_see[y][creator(x)] >= x }   (stored as y's position on m's chain).
// An event y newly sees, of member c's chain, exactly the positions (seq[L[sp(y)][c]], seq[L[y][c]]] — what its row
// has beyond its self-parent's row.  Round 6 walks the CHAINS: a workgroup takes a stretch of member m's chain, one
// lane per column c; the previous row's entry stays in a register, so every row is read ONCE (round 4-5: thread (y, c)
// read L[y][c], sp[y], L[sp][c], seq[..] in three dependent trips with a 32-byte footprint per row and workgroup, every
// row twice: 0.84 TB/s), D rows in flight per lane; and the table is laid out [member m][chain c][position]: what lane c
// writes for consecutive events of m is a run of consecutive addresses — its lines are completed by ONE lane of ONE
// workgroup, nothing has to meet in L2 and nothing has to be cleared first (the last stretch of a chain fills the tail
// of its runs with "not seen").
// A GROUP of consecutive round entries [i0, i1) shares one table: chain c contributes its positions [ordat[i0][c],
// ordat[i1][c]), padded to whole tiles of P positions: dense index of (c, p) = toff[c] * P + p - ordat[i0][c].
// What is stored: the position on m's chain of the first event of m that sees x, -1 when no event of m below the
// group's last famous witness sees x.
// ---------------------------------------------------------------------------------
// timestamps in chain order, for the positions a call can sample: tch[chain_start[m] + p] = t[p-th event of m], p from the
// position in front of m's first unordered event (the first event of m that sees an unordered x is itself unordered — the
// ordered set is ancestor-closed — and the sample is the event in front of it).  One gather per sample instead of two.
__global__ void __launch_bounds__(256)
k_order_tchain(const int* __restrict__ chain_start, const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev,
               const int* __restrict__ ordpos, const double* __restrict__ t, double* __restrict__ tch) {
    const int m = blockIdx.x;
    const int cs = chain_start[m], cnt = chain_cnt[m];
    int p0 = ordpos[m] - 1;
    p0 = p0 < 0 ? 0 : p0;
    for (int p = p0 + blockIdx.y * 256 + threadIdx.x; p < cnt; p += gridDim.y * 256) tch[cs + p] = t[chain_ev[cs + p]];
}

struct OrderGroup {        // per group, on the device: [toff: npad + 1][j0: npad][j1: npad]
    static __host__ __device__ size_t ints(int npad) { return (size_t)3 * npad + 64; }
};

// one workgroup of npad threads per group (all groups of a call in ONE launch: two chains of ~12 dependent probes each)
__global__ void __launch_bounds__(1024)
k_order_group(const int* __restrict__ ordat, const int* __restrict__ fwm, const int* __restrict__ chain_start,
              const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev, const int* __restrict__ gbounds, int n, int npad, int P,
              int* __restrict__ grp_all) {
    __shared__ int s_buf[1024];
    __shared__ int s_x0, s_y1;
    const int c = threadIdx.x;
    const int i0 = gbounds[2 * blockIdx.x], i1 = gbounds[2 * blockIdx.x + 1];
    int* grp = grp_all + OrderGroup::ints(npad) * blockIdx.x;
    int* toff = grp;
    int* J0 = grp + npad + 1;
    int* J1 = J0 + npad;
    if (c == 0) { s_x0 = 0x7fffffff; s_y1 = 0; }
    const int plo = ordat[(size_t)i0 * npad + c], phi = ordat[(size_t)i1 * npad + c];
    const int len = phi - plo;
    int total;
    const int ex = block_excl_scan((len + P - 1) / P, s_buf, npad, &total);   // (its barriers publish s_x0 / s_y1)
    toff[c] = ex;
    if (c == 0) toff[npad] = total;
    const int cs = chain_start[c], clen = chain_cnt[c];
    if (len > 0) atomicMin(&s_x0, chain_ev[cs + plo]);
    int wmax = -1;
    if (c < n) {
#pragma unroll 8
        for (int i = i0; i < i1; ++i) { const int w = fwm[(size_t)i * npad + c]; wmax = w > wmax ? w : wmax; }
    }
    if (wmax >= 0) atomicMax(&s_y1, wmax + 1);
    __syncthreads();
    const int x0 = s_x0, y1 = s_y1;
    // the stretch of this member's chain the walk covers: its events in [x0, y1)
    int a = 0, b = clen;
    while (a < b) { const int mid = (a + b) >> 1; if (chain_ev[cs + mid] < x0) a = mid + 1; else b = mid; }
    const int j0 = a;
    b = clen;
    while (a < b) { const int mid = (a + b) >> 1; if (chain_ev[cs + mid] < y1) a = mid + 1; else b = mid; }
    J0[c] = j0;
    J1[c] = a;
}

// grid = n chains x (npad / CW) column groups x S stretches; CW threads, one per column.
//
// Stores.  A lane's entries are consecutive addresses, but ~1 entry of 4 bytes per step: left to L2 they leave as partial lines
// (measured: 2.9 x the table's bytes written, 74 % of the wave cycles stalled at issue, profiles/r06e_order_pmc.txt).  Every lane
// assembles its entries in a ring of 32 in LDS (slot-major: a wave's writes never conflict) and stores whole 64-byte chunks,
// four 16-byte stores each; the first and the last chunk of a stretch, shared with the neighbouring stretch or chain, go out
// entry by entry.
// Instructions.  With a loop "for every step: for every new entry" a wave runs max-over-lanes(entries of the step) trips per
// step with ~10 of 64 lanes active: ~150 instructions per step, and the kernel was bound by issuing them
// (profiles/r06g_order_pmc.txt).  The steps are taken 8 at a time instead: every lane first turns its 8 counts into a bitmap —
// count zeros, then a one, per step — and then writes its entries one per trip, the step of an entry = the ones below its zero
// (one v_ffbl per trip); a wave runs max-over-lanes(entries of 8 steps) trips with about half of the lanes active, and
// completed chunks are looked for once per four trips.  An entry is the CHAIN POSITION of the first event of m that sees x
// (the position of the batch's first step + the entry's step: arithmetic, nothing to look up); the consumer turns it into
// the sample.
// Latency.  The rows of the next 8 steps are requested before the current 8 are consumed, their chain entries one batch
// earlier still: one exposed round trip per batch (the positions of the entries the rows hold) instead of three.
template <int CW>
__global__ void __launch_bounds__(CW)
k_order_walk(const int* __restrict__ L, const int* __restrict__ seq, const int* __restrict__ chain_start,
             const int* __restrict__ chain_ev, const int* __restrict__ ordat, int i0, int i1, const int* __restrict__ grp,
             int npad, int S, int P, long long stride, int* __restrict__ FDT) {
    constexpr int D = 8;        // steps per batch
    constexpr int CH = 16;      // entries per chunk
    constexpr int RING = 32;    // entries of a lane's ring (two chunks); slot RING takes the writes of lanes that have none
    constexpr int LS = 36;      // dwords between two lanes' rings (16-byte aligned: the chunks are read as four b128)
    __shared__ __attribute__((aligned(16))) int s_ring[LS * CW];
    __shared__ int s_list[CW / 64][2][16];
    const int ncg = npad / CW;
    const int s = blockIdx.x % S;
    const int cg = (blockIdx.x / S) % ncg;
    const int m = blockIdx.x / (S * ncg);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int c = cg * CW + tid;
    const int* toff = grp;
    const int j0 = grp[npad + 1 + m], j1 = grp[2 * npad + 1 + m];
    const int span = j1 > j0 ? j1 - j0 : 0;
    const int seglen = (span + S - 1) / S;
    int ja = j0 + s * seglen;
    ja = ja < j0 + span ? ja : j0 + span;
    int jb = ja + seglen;
    jb = jb < j0 + span ? jb : j0 + span;
    const bool last = s == S - 1;
    if (ja >= jb && !last) return;
    const int plo = ordat[(size_t)i0 * npad + c], phi = ordat[(size_t)i1 * npad + c];
    const bool dead = phi <= plo;      // nothing of this column is ordered by the group (padding columns among them)
    const int* const chain = chain_ev + chain_start[m];
    int* const plane = FDT + (size_t)m * stride;
    const int dbase = toff[c] * P - plo;   // dense index of chain position p = dbase + p
    int* const ring = s_ring + tid * LS;   // this lane's ring
    int* const wring = s_ring + (tid - lane) * LS;   // ... and the ring of lane 0 of this wave
    int (*const list)[16] = s_list[tid >> 6];
    int nextp = plo;      // positions below are written (by this stretch or the ones before it)
    if (ja > j0) {        // what the event in front of the stretch saw: written by the stretch before this one
        const int pv = L[(size_t)chain[ja - 1] * npad + c];
        if (pv >= 0 && !dead) { const int ps = seq[pv] + 1; nextp = ps > nextp ? ps : nextp; }
    }
    int w = dbase + nextp;   // dense index of the next entry
    int f = w;               // dense index of the first entry still in the ring
    // The complete chunks of the wave's rings go out TOGETHER (every lane of the wave calls this): a chunk is 64 contiguous
    // bytes of ONE lane's plane, so a lane storing its own chunk issues four 16-byte stores that nothing coalesces (measured:
    // 77 of the kernel's 183 us, with the table in L2 or not).  The lanes with a complete chunk are compacted into a list of
    // <= 16; lanes 4 g .. 4 g + 3 then move the chunk of the g-th of them, 16 bytes each: one store instruction = up to 16
    // chunks, every one of them a whole 64-byte request.
    auto flush_complete = [&]() {
        int fal = f & ~(CH - 1);
        bool rdy = w - fal >= CH;
        if (rdy && f != fal) {   // (the first chunk of a stretch that starts inside it: shared with the stretch before)
            for (int k = f - fal; k < CH; ++k) plane[fal + k] = ring[(fal & (RING - 1)) + k];
            f = fal + CH;
            rdy = false;
        }
        u64 mask = __ballot(rdy);
        while (mask) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            const bool mine = rdy && rank < 16;
            if (mine) { list[0][rank] = lane; list[1][rank] = fal; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int nrdy = __popcll(mask);
            const int g = lane >> 2, q = lane & 3;
            if (g < nrdy) {
                const int src = list[0][g], fs = list[1][g];
                const int4 e = *reinterpret_cast<const int4*>(wring + src * LS + (fs & (RING - 1)) + 4 * q);
                *reinterpret_cast<int4*>(plane + fs + 4 * q) = e;
            }
            if (mine) { f = fal + CH; rdy = false; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mask = __ballot(rdy);
        }
    };
    // entries [nextp, up] = val, one at a time (long runs; the tail); every lane of the wave calls this
    auto emit_slow = [&](const int up, const int val) {
        while (__any(nextp <= up)) {
            const bool act = nextp <= up;
            ring[act ? (w & (RING - 1)) : RING] = val;
            w += act ? 1 : 0;
            nextp += act ? 1 : 0;
            flush_complete();
        }
    };
    int yn[D], vn[D], y2[D];   // events and rows of the next batch, events of the one behind it
#pragma unroll
    for (int d = 0; d < D; ++d) yn[d] = chain[ja + d < jb ? ja + d : (jb > ja ? jb - 1 : ja)];
#pragma unroll
    for (int d = 0; d < D; ++d) vn[d] = jb > ja ? L[(size_t)yn[d] * npad + c] : -1;
#pragma unroll
    for (int d = 0; d < D; ++d) y2[d] = chain[ja + D + d < jb ? ja + D + d : (jb > ja ? jb - 1 : ja)];
    for (int j = ja; j < jb; j += D) {
        int v[D], u[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = vn[d];
#pragma unroll
        for (int d = 0; d < D; ++d) u[d] = seq[(v[d] < 0 || dead) ? 0 : v[d]];
#pragma unroll
        for (int d = 0; d < D; ++d) yn[d] = y2[d];
        if (j + D < jb) {
#pragma unroll
            for (int d = 0; d < D; ++d) vn[d] = L[(size_t)yn[d] * npad + c];
#pragma unroll
            for (int d = 0; d < D; ++d) y2[d] = chain[j + 2 * D + d < jb ? j + 2 * D + d : jb - 1];
        }
        // last position each of the 8 steps sees (-1: none yet), the steps' counts as a bitmap
        int up[D];
        unsigned B = 0;
        int cur = nextp, pos = 0;
#pragma unroll
        for (int e = 0; e < D; ++e) {
            int q = v[e] >= 0 ? u[e] : -1;
            q = q < phi - 1 ? q : phi - 1;
            up[e] = j + e < jb ? q : -1;
            int cnt = up[e] - cur + 1;
            cnt = cnt > 0 ? cnt : 0;
            cur += cnt;
            pos += cnt;
            B |= pos < 32 ? 1u << pos : 0u;
            ++pos;
        }
        const int T = cur - nextp;
        if (__any(T > 24)) {   // (a bitmap of 32 bits holds 8 ones and 24 zeros: long runs take the plain loop)
#pragma unroll
            for (int e = 0; e < D; ++e) emit_slow(up[e], j + e);
        } else {
            int val = j;   // the entry's step as a position on m's chain
            for (int i = 0; __any(i < T); i += 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = __ffs(~B) - 1;   // ones below the next zero: steps without (further) entries
                    val += t;
                    B >>= t + 1;
                    const bool act = i + k < T;
                    ring[act ? (w & (RING - 1)) : RING] = val;
                    w += act ? 1 : 0;
                }
                flush_complete();
            }
            nextp = cur;
        }
    }
    if (last) emit_slow(phi - 1, -1);   // no event of m below the group's last famous witness sees these
    for (int d = f; d < w; ++d) plane[d] = ring[d & (RING - 1)];   // the chunk the next stretch (or the next chain's columns) continues
}

// One workgroup per tile of P consecutive positions of ONE chain c: the tile's column of every member's plane is read in
// runs of P entries (whole 64-byte sectors) and transposed through LDS; then one wave per ordered event x takes the sample of
// every famous witness that sees x (swirld.py:291-303) and the pseudo-median (:304-305).  An entry jj of member m's plane is
// the position, on m's chain, of the first event of m that sees x (-1 none): the famous witness w of m sees x iff jj <= seq[w]
// (swirld.py:291-292), and the sample is the timestamp of the event in front of position jj — the first self-ancestor that
// does NOT see x — or of the root itself at jj = 0 (Q11).
template <int NW, int P>
__global__ void __launch_bounds__(256)
k_order_median(const int* __restrict__ FDT, long long stride, const int* __restrict__ grp, const int* __restrict__ ordat,
               int i0, int i1, int n, const int* __restrict__ fwseq, const long long* __restrict__ acc_off,
               const int* __restrict__ seg_off, const int* __restrict__ chain_start, const double* __restrict__ tch,
               double* __restrict__ ts, OrderInfo* info) {
    constexpr int npad = 64 * NW;
    constexpr int LD = npad + 1;
    __shared__ int s_a[P * LD];
    __shared__ int s_ri[P], s_idx[P];
    __shared__ double s_scratch[4][64];
    const int T = blockIdx.x;
    const int* toff = grp;
    if (T >= toff[npad]) return;
    int lo_ = 0, hi_ = npad;   // the chain of this tile: the last c with toff[c] <= T
    while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (toff[mid] <= T) lo_ = mid; else hi_ = mid; }
    const int c = lo_;
    const int plo = ordat[(size_t)i0 * npad + c], phi = ordat[(size_t)i1 * npad + c];
    const int p0 = plo + (T - toff[c]) * P;
    const int cnt = phi - p0 < P ? phi - p0 : P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // the tile, transposed: s_a[k][m]
        constexpr int MPL = 256 / P;            // members per pass of the workgroup
        constexpr int NP = npad / MPL;          // passes
        const int k = tid % P, mm = tid / P;
        const int* src = FDT + (size_t)T * P + k;
        int e[NP];   // (all requests of a thread in flight together: a loop with the LDS write inside waited for every one of them in turn)
#pragma unroll
        for (int i = 0; i < NP; ++i) { const int m = mm + i * MPL; e[i] = src[(size_t)(m < n ? m : 0) * stride]; }
#pragma unroll
        for (int i = 0; i < NP; ++i) { const int m = mm + i * MPL; if (m < n) s_a[k * LD + m] = e[i]; }
    }
    if (tid < cnt) {   // the round entry of every position, and where its timestamp goes
        const int p = p0 + tid;
        int a = i0, b = i1;   // the last entry ri with ordat[ri][c] <= p
        while (b - a > 1) { const int mid = (a + b) >> 1; if (ordat[(size_t)mid * npad + c] <= p) a = mid; else b = mid; }
        s_ri[tid] = a;
        s_idx[tid] = (int)(acc_off[a] + seg_off[(size_t)a * npad + c] + (p - ordat[(size_t)a * npad + c]));
    }
    int cs[NW];
#pragma unroll
    for (int r = 0; r < NW; ++r) cs[r] = chain_start[lane + 64 * r];
    __syncthreads();
    constexpr int KP = P / 4;
    int fw[NW];
    int ri_have = -1;
    for (int k = wave * KP; k < (wave + 1) * KP && k < cnt; ++k) {
        const int ri = s_ri[k];
        if (ri != ri_have) {
#pragma unroll
            for (int r = 0; r < NW; ++r) fw[r] = fwseq[(size_t)ri * npad + lane + 64 * r];
            ri_have = ri;
        }
        double v[NW];
        u64 valid[NW];
        int len = 0;
        int jj[NW];
        // (every read and every gather unconditional, from indices that exist: a load inside a conditional block is waited for
        // at the end of the block, four blocks were four trips in a row)
#pragma unroll
        for (int r = 0; r < NW; ++r) jj[r] = s_a[k * LD + lane + 64 * r];   // (columns >= n hold nothing: their fw is -1)
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const bool sees = fw[r] >= 0 && jj[r] >= 0 && jj[r] <= fw[r];
            v[r] = tch[cs[r] + (sees && jj[r] > 0 ? jj[r] - 1 : 0)];
            asm volatile("" : "+v"(v[r]));   // (the load stays in front of the selects)
            valid[r] = __ballot(sees);
            len += __popcll(valid[r]);
        }
#pragma unroll
        for (int r = 0; r < NW; ++r) v[r] = ((valid[r] >> lane) & 1ull) ? v[r] : __longlong_as_double(0x7ff8000000000000ll);
        if ((len + 1) / 2 >= len) {  // IndexError in the reference (len == 1, only with unequal stakes)
            if (lane == 0) { atomicExch(&info->index_err, 1); ts[s_idx[k]] = 0.0; }
            continue;
        }
        const double med = wave_pseudo_median<NW>(v, valid, len, s_scratch[wave]);
        if (lane == 0) ts[s_idx[k]] = med;
    }
}

// whitening key of a decided round (swirld.py:285): XOR of its famous witnesses' signatures; 16 groups of members x 64 bytes
__global__ void __launch_bounds__(1024)
k_order_white(const int* __restrict__ fwm, int n, int npad,
              const unsigned char* __restrict__ sig, unsigned char* __restrict__ white) {
    __shared__ unsigned char s_w[16][64];
    const int ri = blockIdx.x, b = threadIdx.x & 63, g = threadIdx.x >> 6;
    unsigned char w = 0;
#pragma unroll 4
    for (int m = g; m < n; m += 16) {
        const int e = fwm[(size_t)ri * npad + m];
        const unsigned char x = sig[(size_t)(e < 0 ? 0 : e) * 64 + b];
        w ^= e >= 0 ? x : (unsigned char)0;
    }
    s_w[g][b] = w;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 1; k < 16; ++k) w ^= s_w[k][b];
        white[(size_t)ri * 64 + b] = w;
    }
}

// final order inside a round (swirld.py:306): sort by (consensus timestamp, whitened signature).
// One workgroup per round, bitonic sort in LDS on (ts, first 8 key bytes as a big-endian
// integer); a round with more than SORT_CAP events, or with two events equal in both (the
// remaining 56 key bytes would have to decide), is flagged and sorted by the host instead.
constexpr int SORT_CAP = 4096;
__global__ void __launch_bounds__(1024)
k_order_sort(const int* __restrict__ acc_ev, const long long* __restrict__ acc_off,
             const double* __restrict__ ts, const unsigned char* __restrict__ sig,
             const unsigned char* __restrict__ white, int ri0, int* out_ev, int* host_flag) {
    __shared__ double s_ts[SORT_CAP];
    __shared__ u64 s_k8[SORT_CAP];
    __shared__ int s_ev[SORT_CAP];
    const int ri = ri0 + blockIdx.x, tid = threadIdx.x;
    const long long a0 = acc_off[ri];
    const int cnt = (int)(acc_off[ri + 1] - a0);
    if (cnt > SORT_CAP) {
        if (tid == 0) host_flag[ri] = 1;
        return;
    }
    int m = 1;
    while (m < cnt) m <<= 1;
    u64 wk = 0;
    for (int b = 0; b < 8; ++b) wk = (wk << 8) | white[(size_t)ri * 64 + b];
    for (int i = tid; i < m; i += 1024) {
        if (i < cnt) {
            const int e = acc_ev[a0 + i];
            u64 k = 0;
            for (int b = 0; b < 8; ++b) k = (k << 8) | sig[(size_t)e * 64 + b];
            s_ts[i] = ts[a0 + i];
            s_k8[i] = k ^ wk;
            s_ev[i] = e;
        } else {
            s_ts[i] = __longlong_as_double(0x7ff0000000000000ll);  // +inf padding sorts last
            s_k8[i] = ~0ull;
            s_ev[i] = 0x7fffffff;
        }
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (m >> 1); p += 1024) {   // thread p owns the pair (i, i | j)
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const double ta = s_ts[i], tb = s_ts[l];
                const u64 ka = s_k8[i], kb = s_k8[l];
                const int ea = s_ev[i], eb = s_ev[l];
                const bool gt = ta > tb || (ta == tb && (ka > kb || (ka == kb && ea > eb)));
                if (gt == up) {
                    s_ts[i] = tb; s_ts[l] = ta;
                    s_k8[i] = kb; s_k8[l] = ka;
                    s_ev[i] = eb; s_ev[l] = ea;
                }
            }
            __syncthreads();
        }
    }
    int tie = 0;
    for (int i = tid; i < cnt; i += 1024) {
        out_ev[a0 + i] = s_ev[i];
        if (i + 1 < cnt && s_ts[i] == s_ts[i + 1] && s_k8[i] == s_k8[i + 1]) tie = 1;
    }
    if (__syncthreads_or(tie) && tid == 0) host_flag[ri] = 1;
}

// ... and the rounds with more than SORT_CAP events (non-uniform hashgraphs: a round of two cliques or of a hashgraph with
// slow members orders 4-6 k events at 256 members; the host sorted those: 88-107 ms per 1 M events, profiles/r04_final3_*):
// the same bitonic network over a scratch copy of the keys in global memory (20 B per event, L2-resident), one workgroup per
// such round, thread p of a stage owns the pair (i, i | j).  big_ri[b] = round-list entry, big_off[b] .. big_off[b + 1] =
// its slice of the scratch arrays (length = the next power of two).  Ties stay with the host (flag), as above.
__global__ void __launch_bounds__(1024)
k_order_sort_big(const int* __restrict__ big_ri, const long long* __restrict__ big_off,
                 const int* __restrict__ acc_ev, const long long* __restrict__ acc_off,
                 const double* __restrict__ ts, const unsigned char* __restrict__ sig,
                 const unsigned char* __restrict__ white, double* k_ts, u64* k_k8, int* k_ev, int* out_ev, int* host_flag) {
    const int ri = big_ri[blockIdx.x], tid = threadIdx.x;
    const long long s0 = big_off[blockIdx.x];
    const int m = (int)(big_off[blockIdx.x + 1] - s0);
    const long long a0 = acc_off[ri];
    const int cnt = (int)(acc_off[ri + 1] - a0);
    double* T = k_ts + s0;
    u64* K8 = k_k8 + s0;
    int* E = k_ev + s0;
    u64 wk = 0;
    for (int b = 0; b < 8; ++b) wk = (wk << 8) | white[(size_t)ri * 64 + b];
    for (int i = tid; i < m; i += 1024) {
        if (i < cnt) {
            const int e = acc_ev[a0 + i];
            u64 k = 0;
            for (int b = 0; b < 8; ++b) k = (k << 8) | sig[(size_t)e * 64 + b];
            T[i] = ts[a0 + i];
            K8[i] = k ^ wk;
            E[i] = e;
        } else {
            T[i] = __longlong_as_double(0x7ff0000000000000ll);  // +inf padding sorts last
            K8[i] = ~0ull;
            E[i] = 0x7fffffff;
        }
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (m >> 1); p += 1024) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const double ta = T[i], tb = T[l];
                const u64 ka = K8[i], kb = K8[l];
                const int ea = E[i], eb = E[l];
                const bool gt = ta > tb || (ta == tb && (ka > kb || (ka == kb && ea > eb)));
                if (gt == up) {
                    T[i] = tb; T[l] = ta;
                    K8[i] = kb; K8[l] = ka;
                    E[i] = eb; E[l] = ea;
                }
            }
            __syncthreads();
        }
    }
    int tie = 0;
    for (int i = tid; i < cnt; i += 1024) {
        out_ev[a0 + i] = E[i];
        if (i + 1 < cnt && T[i] == T[i + 1] && K8[i] == K8[i + 1]) tie = 1;
    }
    const int any = __syncthreads_or(tie);
    if (tid == 0) host_flag[ri] = any;   // (k_order_sort, launched before, flagged the round as oversize)
}
