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
    int mask_from;  // first band event whose mask must be (re)built in this iteration
    int N;          // events visible to this round-loop run (rows below N are complete)
    int ncap;       // current cap of the band length (doubles when a far candidate needs a real tally)
    int iter;       // iterations executed in this call
    int n_unres;    // members still searching their first round-(r+1) event
    int max_round;  // valid when done
    int err;        // 1 = lo table capacity exceeded
    u64 evals;      // tallies evaluated (live candidates)
    u64 far_hops;   // hop masks computed on the fly (outside the band)
    u64 band_events;  // band events whose threshold mask was (re)built (work of k_resolve_band's step 2)
    int fin_from;     // first event of THIS run's sub-batch: band events from here on get their round and sees-mask from the band pass
    int pad_;         // diagnostics (SW_DEBUG_CLOCKS): iterations of the context before this loop — the index base of the phase stamps
};

struct FameCounters {
    u64 voter_evals;     // V  (swirld.py:247-254)
    u64 majority_evals;  // P2 (swirld.py:260)
    u64 coin_votes;      // votes cast in coin rounds (swirld.py:267-272)
    u64 coin_flips;      // ... of which taken from the signature bit (swirld.py:272)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Wave and workgroup reductions for the resolve step.  LDS atomics on one address with a
// different value per lane are expanded by the compiler into a 64-trip scalar loop (~2 us on the
// critical path of every iteration): butterflies + one LDS slot per wave + ONE barrier instead.
// Round 5: the butterflies no longer go through ds_bpermute (`__shfl_xor`: six DEPENDENT trips through the LDS
// crossbar per reduction, ~0.3 us each time on the critical path of both loop kernels) — four DPP steps inside a
// row of 16 lanes, then v_permlane16_swap / v_permlane32_swap (gfx950) across the rows: VALU only.
// EVERY lane of the wave must be active at the call (all call sites are in wave-uniform control flow).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <- lane 7 - i of its 8
constexpr int DPP_MIRROR = 0x140;      // lane i <- lane 15 - i of its row
constexpr int DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;   // rotation inside a row of 16 lanes

// the two addends of a butterfly step across rows: {own, partner} up to order, the same pair in both partner lanes
__device__ __forceinline__ void rows_pair16(int v, int& a, int& b) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0]; b = (int)r[1];
}
__device__ __forceinline__ void rows_pair32(int v, int& a, int& b) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0]; b = (int)r[1];
}
#define SW_WAVE_REDUCE(v, OP)                                                                   \
    do {                                                                                        \
        int o_, a_, b_;                                                                         \
        o_ = dpp_i32<DPP_XOR1>(v); v = OP(v, o_);                                               \
        o_ = dpp_i32<DPP_XOR2>(v); v = OP(v, o_);                                               \
        o_ = dpp_i32<DPP_HALF_MIRROR>(v); v = OP(v, o_);                                        \
        o_ = dpp_i32<DPP_MIRROR>(v); v = OP(v, o_);                                             \
        rows_pair16(v, a_, b_); v = OP(a_, b_);                                                 \
        rows_pair32(v, a_, b_); v = OP(a_, b_);                                                 \
    } while (0)
#define SW_OP_MIN(x, y) ((y) < (x) ? (y) : (x))
#define SW_OP_MAX(x, y) ((y) > (x) ? (y) : (x))
#define SW_OP_ADD(x, y) ((x) + (y))
__device__ __forceinline__ int wave_min_i32(int v) { SW_WAVE_REDUCE(v, SW_OP_MIN); return v; }
__device__ __forceinline__ int wave_max_i32(int v) { SW_WAVE_REDUCE(v, SW_OP_MAX); return v; }
__device__ __forceinline__ int wave_sum_i32(int v) { SW_WAVE_REDUCE(v, SW_OP_ADD); return v; }

// the two addends of the bit-sliced adders' step across lane distance OFF (a power of two): lanes (g, w) and (g', w) of one
// mask word w.  Inside a row the partner comes by DPP — an xor for OFF 1 / 2, a ROTATION for 4 / 8 (the steps are taken in
// ascending order up to 8, so the rotations by 4 and by 8 together visit every group of the row: each lane ends with the sum
// over all of them, like the xor butterfly) — across rows by the permlane swaps.
template <int OFF>
__device__ __forceinline__ void lanes_pair(uint32_t x, uint32_t& a, uint32_t& b) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "lane distance");
    if constexpr (OFF == 32) { int a_, b_; rows_pair32((int)x, a_, b_); a = (uint32_t)a_; b = (uint32_t)b_; }
    else if constexpr (OFF == 16) { int a_, b_; rows_pair16((int)x, a_, b_); a = (uint32_t)a_; b = (uint32_t)b_; }
    else if constexpr (OFF == 8) { a = x; b = (uint32_t)dpp_i32<DPP_ROR8>((int)x); }
    else if constexpr (OFF == 4) { a = x; b = (uint32_t)dpp_i32<DPP_ROR4>((int)x); }
    else if constexpr (OFF == 2) { a = x; b = (uint32_t)dpp_i32<DPP_XOR2>((int)x); }
    else { a = x; b = (uint32_t)dpp_i32<DPP_XOR1>((int)x); }
}


// ---------------------------------------------------------------------------------
// Level buckets: events of one divide_rounds batch sorted by DAG height, so that all
// events of one level are independent (parents have strictly smaller height,
// swirld.py:117-120).
// ---------------------------------------------------------------------------------
// Will event e find its other-parent o in the level kernel's ring (depth H per member)?  o's slot is taken by its creator's
// event H chain positions later — x — as soon as x's level is done; levels grow along a chain, so o has left the ring (or is
// leaving it in e's very level) iff x belongs to this sweep and ht[x] <= ht[e].  Such an o is PINNED: its row slice also goes to
// a small side table of the level kernel (slot o % SW_LEVEL_SIDE behind the ring), which is where e looks for it.
#define SW_LEVEL_SIDE 512
__device__ __forceinline__ bool level_op_pinned(int e, int o, int first, int K, int H, const int* __restrict__ ht, const int* __restrict__ cr,
                                                const int* __restrict__ seq, const int* __restrict__ chain_start,
                                                const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev) {
    if (o < first) return false;               // a row of an earlier launch: read from memory
    const int co = cr[o];
    const int p = seq[o] + H;
    if (p >= chain_cnt[co]) return false;
    const int x = chain_ev[chain_start[co] + p];
    return x < first + K && ht[x] <= ht[e];
}

__global__ void k_level_hist(const int* __restrict__ ht, int first, int K, int hmin, int* cnt, const int* __restrict__ cr,
                             const int* __restrict__ op, const int* __restrict__ seq, const int* __restrict__ chain_start,
                             const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev, int H, unsigned char* pin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int e = first + i;
    atomicAdd(&cnt[ht[e] - hmin], 1);
    const int o = op[e];
    if (o >= 0 && level_op_pinned(e, o, first, K, H, ht, cr, seq, chain_start, chain_cnt, chain_ev)) pin[o - first] = 1;
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

// desc = {event, self-parent, other-parent, w}; w packs the ring indices of the level kernel (k_cansee_stream, ring depth H):
//   own slot cr(e)·H + seq(e) % H [14 bits]
//   | where the other-parent is looked for << 14 [15 bits]: its ring slot cr(op)·H + seq(op) % H; its side table slot
//     npad·H + 1 + rank(op) % SW_LEVEL_SIDE if it will have left the ring (level_op_pinned; filled in by k_level_patch);
//     npad·H = the all-absent slot of a root
//   | (seq(e) % H == 0) << 29 (the self-parent's slot is own slot - 1, + H behind a wrap) | (e itself is pinned) << 30,
// seq = position on the creator's self-parent chain.  The PINNED events of a level stand at the head of its descriptors
// (they are dealt from the front, the others from the back), so the pinned event at position k of level lv has
// rank = pinbase[lv] + k among all pinned events of the sweep, level by level: the side table is written round-robin
// and an entry lives until SW_LEVEL_SIDE later pinned events have been written — hundreds of levels.
__global__ void k_level_scatter(const int* __restrict__ ht, const int* __restrict__ cr,
                                const int* __restrict__ sp, const int* __restrict__ op,
                                const int* __restrict__ seq, int first, int K,
                                int hmin, const int* __restrict__ start, int* cfront, int* cback, int4* desc, int H, int npad,
                                const unsigned char* __restrict__ pin, int* __restrict__ pos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    int e = first + i;
    int lv = ht[e] - hmin;
    const int pinned = pin[i];
    const int slot = pinned ? start[lv] + atomicAdd(&cfront[lv], 1) : start[lv + 1] - 1 - atomicAdd(&cback[lv], 1);
    pos[i] = slot;
    const int o = op[e];
    const int se = seq[e] % H;
    int w = (cr[e] * H + se) | ((se == 0 ? 1 : 0) << 29) | (pinned << 30);
    w |= (o >= 0 ? cr[o] * H + seq[o] % H : npad * H) << 14;
    desc[slot] = make_int4(e, sp[e], o, w);
}

// the children of pinned events look for them in the side table
__global__ void k_level_patch(const int* __restrict__ ht, const int* __restrict__ cr, const int* __restrict__ op,
                              const int* __restrict__ seq, int first, int K, int hmin, const int* __restrict__ start,
                              const int* __restrict__ pinbase, int4* desc, int H, int npad,
                              const int* __restrict__ chain_start, const int* __restrict__ chain_cnt, const int* __restrict__ chain_ev,
                              const int* __restrict__ pos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int e = first + i;
    const int o = op[e];
    if (o < 0 || !level_op_pinned(e, o, first, K, H, ht, cr, seq, chain_start, chain_cnt, chain_ev)) return;
    const int lvo = ht[o] - hmin;
    const int rank = pinbase[lvo] + pos[o - first] - start[lvo];
    int* w = &desc[pos[i]].w;
    *w = (*w & ~(0x7fff << 14)) | ((npad * H + 1 + (rank & (SW_LEVEL_SIDE - 1))) << 14);
}



// LDS-only workgroup barrier of the level-bucketed sweep (never waits for global memory)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}



// A load the compiler's wait-count pass does not see: issued and waited for inside one asm
// statement.  A visible load inside the worker loop would make the pass put `s_waitcnt vmcnt(0)`
// in front of every later use of the (reused) destination register, i.e. drain the wave's stores
// on every trip.  sc1: served by L2, bypassing this CU's vector L1.
__device__ __forceinline__ int load_sc1_and_wait(const int* ptr) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ptr) : "memory");
    return v;
}

// helpers of the level-bucketed sweep (k_cansee_stream, below the column vectors)
__device__ __forceinline__ void wait_vm_at_most(int n) {   // n = a LOWER bound of the VMEM instructions issued behind the one waited for
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one wave's 1 KB piece of a staging chunk: lane i's 16 bytes land at lds_dst + 16 i (lds_dst wave-uniform, in an SGPR)
__device__ __forceinline__ void lds_dma_16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}


// ---------------------------------------------------------------------------------
// DATAFLOW sweep — no DAG levels, no heights, no barriers.
//
// The level-synchronous kernel above pays one workgroup barrier (~0.45 us) per DAG level and
// need the events bucketed by height first.  Here every member's self-parent chain is walked by
// one lane at its own pace: the lane's next event needs (a) the member's previous row value,
// which stays in a register, and (b) the other-parent's value, which the lane POLLS for in an LDS
// ring of {event id, value} pairs (slot = chain position mod H, one 8-byte read).  A lane whose
// dependency is not there yet simply tries again in the next trip of its wave's loop; lanes of
// one wave progress independently (one loop, per-lane state), waves never wait for each other.
//  * tag == wanted: hit.  tag < wanted: not produced yet (event ids grow along a chain): poll
//    again.  tag > wanted: the slot was reused, i.e. the producer lane has advanced >= H
//    positions since, each with its own store instruction, and every producer wave keeps at most
//    H - 4 store instructions in flight (`s_waitcnt vmcnt(H - 4)` after each) — so the row reached L2
//    long ago and is re-read from there with an sc1 load.  Events before `first_event` were
//    written by earlier kernels and are read from memory directly.
//  * Deadlock-free: a lane only waits for events with a smaller index, so the lowest unprocessed
//    event of the sub-batch can always proceed; all waves of the workgroup are resident.
//  * Descriptors {event, other-parent, creator(op) | seq(op) << 10} come per member from a
//    pool-indexed array (same indexing as chain_ev), streamed by one LOADER wave into per-member
//    LDS FIFOs (loads and stores share the in-order vmcnt counter of a wave, so workers only store
//    and the loader only loads, as in the level kernels); `filled` / `taken` are per-member LDS
//    counters, DS operations of one wave are processed in order, and every polled location is
//    re-read behind a compiler barrier.
//  * One workgroup per column (XCD-aware mapping as above), npad / MPL worker lanes each walking
//    MPL chains + 64 loader lanes; 1024 members run as 512 lanes x 2 chains.
// ---------------------------------------------------------------------------------
#define SW_CBAR() asm volatile("" ::: "memory")

template <int NW, int MPL, int F, int H, bool WIDE, bool DBG>
__global__ void __launch_bounds__(64 * NW / MPL + 64)
k_cansee_flow(const int4* __restrict__ cdesc, const int* __restrict__ chain_start,
              const int* __restrict__ pos0, const int* __restrict__ pos1,
              const int* __restrict__ chain_ev, int first_event, int* L, int* err, u64* dbg) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr int npad = 64 * NW;
    constexpr int NT = npad / MPL;       // worker lanes
    constexpr int SPIN_LIMIT = 1 << 27;  // trips of a polling loop before it gives up (a protocol bug, never work)
    constexpr int hm = H - 1, fm = F - 1;
    static_assert((H & hm) == 0 && (F & fm) == 0 && H <= 64 && H >= 8, "ring / FIFO depths are powers of two; H > 6 store instructions");
    int4* fifo = (int4*)smem;                                  // [F][npad]
    u64* ring = (u64*)(fifo + (size_t)F * npad);               // [H][npad] {value << 32 | event id}
    int* filled = (int*)(ring + (size_t)H * npad);             // [npad] chain positions loaded so far
    int* taken = filled + npad;                                // [npad] chain positions taken by the worker
    const int tid = threadIdx.x;
    const bool loader = tid >= NT;
    const int ll = tid - NT;              // loader lane
    const int nblk = gridDim.x;
    const int col = (nblk % 8 == 0) ? (blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8 : blockIdx.x;
    for (int i = tid; i < npad * H; i += blockDim.x) ring[i] = 0xffffffffffffffffull;  // id -1: empty
    for (int i = tid; i < npad; i += blockDim.x) { const int p = pos0[i]; filled[i] = p; taken[i] = p; }
    __syncthreads();
    if (loader) {
        // members ll, ll + 64, ...: refill a member's FIFO with B descriptors whenever B slots are free;
        // G members per memory round trip (all their loads in flight together)
        constexpr int B = F / 2;
        constexpr int MAXBUF = NW >= 8 ? 16 : 32;  // descriptors staged in registers per round trip
        constexpr int G = (NW * B <= MAXBUF) ? NW : MAXBUF / B;
        static_assert(G >= 1 && NW % G == 0, "loader groups");
        int fl[NW], pe[NW], cs[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int k = ll + 64 * j;
            fl[j] = pos0[k];
            pe[j] = pos1[k];
            cs[j] = chain_start[k];
        }
        int passes = 0, idle_passes = 0;
        for (int spins = 0;; ++spins) {
            bool more = false;
            unsigned need = 0;
            if (spins > SPIN_LIMIT) { if (ll == 0) atomicExch(err, 2); break; }
            SW_CBAR();
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (fl[j] < pe[j]) {
                    more = true;
                    const int tk = taken[ll + 64 * j];
                    if (fl[j] - tk <= F - B) need |= 1u << j;
                }
            }
            if (!__ballot(more)) break;
            if (!__ballot(need != 0)) { __builtin_amdgcn_s_sleep(2); ++idle_passes; continue; }
            ++passes;
#pragma unroll
            for (int g0 = 0; g0 < NW; g0 += G) {
                // Branch-free loads with clamped positions: a load inside an exec-masked block makes
                // the compiler wait for it at the end of the block, which serialised all loads of a pass
                // (14 us per pass instead of one memory round trip).  Lanes that do not need a refill
                // load a valid neighbouring entry and drop it.
                int bx[G * B], by[G * B], bz[G * B];  // (scalar arrays: promoted to registers after unrolling)
#pragma unroll
                for (int jj = 0; jj < G; ++jj) {
                    const int j = g0 + jj;
                    const int last = pe[j] > 0 ? pe[j] - 1 : 0;
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        const int pos = fl[j] + u < pe[j] ? fl[j] + u : last;
                        const int4 t = cdesc[(size_t)cs[j] + pos];
                        bx[jj * B + u] = t.x; by[jj * B + u] = t.y; bz[jj * B + u] = t.z;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < G; ++jj) {
                    const int j = g0 + jj;
                    if ((need >> j) & 1u) {
                        const int k = ll + 64 * j;
                        int nf = fl[j];
#pragma unroll
                        for (int u = 0; u < B; ++u)
                            if (fl[j] + u < pe[j]) {
                                fifo[(size_t)((fl[j] + u) & fm) * npad + k] = make_int4(bx[jj * B + u], by[jj * B + u], bz[jj * B + u], 0);
                                nf = fl[j] + u + 1;
                            }
                        SW_CBAR();  // the entries are written before the count that publishes them (DS ops stay in order)
                        filled[k] = nf;
                        SW_CBAR();
                        fl[j] = nf;
                    }
                }
            }
        }
        if (DBG && dbg && ll == 0 && blockIdx.x == 0) { atomicAdd(&dbg[4], (u64)passes); atomicAdd(&dbg[5], (u64)idle_passes); }
        return;
    }
    // ---- workers: one branch-light trip = every LDS read of the trip issued together (ring poll,
    // FIFO count, the descriptor FOLLOWING the one held), one wait, then the decision.  A lane that
    // completes an event takes the next descriptor in the same trip and polls for it in the next.
    // All addressing is 32-bit (LDS indices; byte offsets against the uniform column base — WIDE
    // selects 64-bit row offsets for tables beyond 4 GB).
    int p[MPL], pend[MPL], mine[MPL], ev[MPL], opar[MPL], ridx[MPL];
    bool have[MPL];
    char* const Lcol = reinterpret_cast<char*>(L + col);
#pragma unroll
    for (int q = 0; q < MPL; ++q) {
        const int m = tid + q * NT;
        p[q] = pos0[m];
        pend[q] = pos1[m];
        mine[q] = -1;
        have[q] = false;
        ev[q] = -1; opar[q] = -1; ridx[q] = 0;
        if (p[q] > 0 && p[q] < pend[q]) {  // the member's latest event of an earlier kernel: its row is the self-parent's row
            const int prev = chain_ev[chain_start[m] + p[q] - 1];
            mine[q] = L[(size_t)prev * npad + col];
        }
    }
#pragma unroll
    for (int q = 0; q < MPL; ++q) asm volatile("" : "+v"(mine[q]));  // the prologue loads are complete before the loop
    int n_mem = 0, n_done = 0, n_starved = 0, n_iter = 0;
    for (int spins = 0;; ++spins) {
        if (spins > SPIN_LIMIT) { if ((tid & 63) == 0) atomicExch(err, 1); break; }
        if (DBG) ++n_iter;
        SW_CBAR();
        u64 pr[MPL];
        int fcnt[MPL];
        int4 nd[MPL];
#pragma unroll
        for (int q = 0; q < MPL; ++q) {
            const int m = tid + q * NT;
            pr[q] = ring[ridx[q]];
            fcnt[q] = filled[m];
            SW_CBAR();  // the count is read before the entry it publishes (DS operations stay in order)
            nd[q] = fifo[((p[q] + (have[q] ? 1 : 0)) & fm) * npad + m];
        }
        SW_CBAR();
        bool busy = false;
#pragma unroll
        for (int q = 0; q < MPL; ++q) {
            const int m = tid + q * NT;
            const int o = opar[q];
            const int tag = (int)(unsigned)pr[q];
            const bool hit = tag == o;
            int other = hit ? (int)(pr[q] >> 32) : -1;
            bool ready = have[q] && (o < 0 || hit);
            // rare: the other-parent's row must come from memory (an earlier kernel's event, or a ring
            // slot that was reused: the row is then >= H store instructions old)
            const bool from_mem = have[q] && o >= 0 && !hit && (o < first_event || tag > o);
            if (__ballot(from_mem)) {
                if (from_mem) {
                    other = load_sc1_and_wait(WIDE ? &L[(size_t)o * npad + col]
                                                   : reinterpret_cast<const int*>(Lcol + (unsigned)o * (unsigned)(npad * 4)));
                    ready = true;
                    if (DBG) ++n_mem;
                }
            }
            if (ready) {
                const int e = ev[q];
                int v = mine[q] > other ? mine[q] : other;  // maxi(): index order == height order on one chain
                if (col == m) v = e;                          // own entry (swirld.py:220)
                mine[q] = v;
                if (WIDE) L[(size_t)e * npad + col] = v;
                else *reinterpret_cast<int*>(Lcol + (unsigned)e * (unsigned)(npad * 4)) = v;
                ring[(p[q] & hm) * npad + m] = ((u64)(unsigned)v << 32) | (unsigned)e;
                ++p[q];
                have[q] = false;
                if (DBG) ++n_done;
            }
            const bool want = !have[q] && p[q] < pend[q];
            if (DBG && want && p[q] >= fcnt[q]) ++n_starved;
            if (want && p[q] < fcnt[q]) {  // nd[q] is the descriptor of position p[q]
                ev[q] = nd[q].x;
                opar[q] = nd[q].y;
                ridx[q] = ((nd[q].z >> 10) & hm) * npad + (nd[q].z & 1023);
                have[q] = true;
                taken[m] = p[q] + 1;  // the slot may be refilled from here on
            }
            busy = busy || p[q] < pend[q];
        }
        // at most H - 4 store instructions of this wave in flight (see the reuse argument above): every
        // store instruction, however few lanes it carries, holds a slot of the wave's in-order counter
        // until L2 acknowledges it
        if constexpr (H >= 32) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if constexpr (H >= 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (!__ballot(busy)) break;
    }
    if (DBG && dbg && blockIdx.x == 0) {  // diagnostics (SW_DEBUG_TIMING): column 0 only
        atomicAdd(&dbg[0], (u64)n_done);
        atomicAdd(&dbg[1], (u64)n_mem);
        atomicAdd(&dbg[2], (u64)n_starved);
        if ((tid & 63) == 0) { atomicAdd(&dbg[3], (u64)n_iter); atomicAdd(&dbg[6], 1ull); }
    }
}

// ---------------------------------------------------------------------------------
// Seventh version: CHUNK-PARALLEL dataflow sweep (tests/model_chunks.py is its executable statement).
//
// The dataflow sweep above is bound by the DEPTH of the hashgraph: one dependent LDS hop per DAG
// level (~0.4 us), 13.2 k levels per million events at 256 members, whatever the number of
// columns.  Here the events of one launch [a_0, a_G) are cut into G chunks that are swept
// CONCURRENTLY, chunk k from w_k = max(a_0, a_k - halo) on; a parent in [a_0, w_k) — a row another
// chunk is computing right now — is treated as a LEAF (the row {creator(x): x}).  A computed value
// >= w_k is final (an in-window ancestor by that member exists and every path to it stays inside
// the window); a smaller one is PROVISIONAL, counted, and repaired afterwards by k_cansee_fixup —
// or, when a chunk has too many of them (members silent for longer than the halo), the chunk is
// swept again from final rows (the same kernel, `exact_chunk` >= 0, gated on the count).  At uniform
// gossip the oldest entry of a row is ~13.4 n events old (max 6.4 k at 256 members), so a halo of
// 32 n events leaves nothing to repair and the sweep is G times shallower.
//   * C columns per LANE: the dependency structure (which event waits for which) is the same for
//     every column, so one poll / one descriptor / one 4C-byte store serve C columns, and a launch
//     needs npad / C workgroups per chunk: G = C chunks occupy the chip exactly like one unchunked
//     sweep did.  Ring entries are {value, tag} pairs, two per 16-byte plane; every tag is checked.
//   * Halo rows [w_k, a_k) are recomputed for the dataflow only and go to a scratch table (chunk
//     k-1 stores the real ones), which is also what a reused ring slot is re-read from.
//   * Same protocol and the same safety argument as k_cansee_flow (one store instruction per
//     completed event, at most H - 4 in flight per wave); MPL = 1 (n <= 256).
// ---------------------------------------------------------------------------------
#define SW_MAX_CHUNKS 8
struct ChunkEv {
    int w[SW_MAX_CHUNKS];  // window start of chunk k (first recomputed event)
    int a[SW_MAX_CHUNKS];  // first event whose row chunk k stores
};

template <int C> struct ColVec;
template <> struct ColVec<2> { typedef int2 T; };
template <> struct ColVec<4> { typedef int4 T; };

template <int C>
__device__ __forceinline__ void load_cols_sc1_and_wait(const int* ptr, int (&v)[C]) {
    if constexpr (C == 4) {
        int4 t;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(ptr) : "memory");
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        int2 t;
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(ptr) : "memory");
        v[0] = t.x; v[1] = t.y;
    }
}

template <int C>
__device__ __forceinline__ void store_cols(int* ptr, const int (&v)[C]) {
    // (round 5, measured and dropped — profiles/r05a_knobs_256x1M.log: streaming (`nt`) stores for the rows, to keep the sweep out of
    // the L2 the round loop gathers from: 7.00 -> 7.78 ms per pass — the rows of one chain are assembled in L2 from 16-byte pieces;
    // profiles/r05i_*, r05k_*: write-through (`sc0 sc1`) stores, so that the loop's kernel boundaries find no dirty lines of the
    // sweep to write back: the sweeps end ~25 iterations earlier, the iterations beside them take 26-27 us instead of 20.5 —
    // boundaries AND kernels: 6.14 -> 6.42 ms.  What the sweep costs the loop beside it is its memory traffic, whatever its form.)
    if constexpr (C == 4) *reinterpret_cast<int4*>(ptr) = make_int4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<int2*>(ptr) = make_int2(v[0], v[1]);
}

// ---------------------------------------------------------------------------------
// Level-bucketed can_see sweep (serves more than 256 members): the oracle's own loop, one DAG level at a time.
// One workgroup = CB adjacent columns of every row, one THREAD per event of a level (a level holds at most one
// event per member: equal heights imply different creators), the CB values of a row slice as one vector: one ring
// read per parent, one 4·CB-byte store per event.  Per member an LDS ring of its H most recent row slices (slot =
// chain position mod H, tag = event id; slot npad·H = "absent", what a root's parents point at); the level
// descriptors stream through an LDS staging ring (NS chunks of CH >= npad descriptors: a level spans at most two).
//
// Round 6 rewrite — the kernel is exposed beyond 512 members since the sweep no longer hides behind the round loop
// there.  The previous form (one thread per (event, column), 4 passes of 256 events, two barriers) spent 136 VALU
// instructions per wave and level on 16 waves, and `s_waitcnt vmcnt(0)` stood in front of every store (the compiler
// cannot tell whether a parent value came from the miss path's global load, and a descriptor chunk prefetched into
// registers is waited for through the same in-order counter as the stores): 1.9 us per level, 10.4 ms per 2 M
// events at 1024 members (profiles/r06_final_1024_pmc_summary.txt).  Now a level is
//     barrier -> ONE LDS round trip (tags and values of both parents, the miss flag of the previous level)
//             -> a dozen VALU instructions -> ring write + row store + next descriptor -> barrier:
//   * decode, tag compares and addresses once per event; the ring indices come precomputed in the descriptor;
//     waves whose slice of the level is empty skip it through scalar branches;
//   * no drain of the global stores on the per-level path: staging chunks arrive by LDS-DMA
//     (`global_load_lds_dwordx4`, 1 KB pieces, no registers, invisible to the compiler's wait-count pass); a wave
//     counts the store instructions it has certainly issued since its last piece and waits, when the chunk is first
//     needed, for `vmcnt(that count)` — loads and stores retire in order, so normally it does not wait at all;
//   * ONE barrier per level: a ring entry is written as {tag := -1, value, tag := event} and an other-parent is read
//     as {tag, value, tag} (DS operations of a wave are performed in order), so a reader that races with a writer of
//     the same level (the slot's member has an event H positions later in this very level) sees a tag mismatch — a
//     miss; a self-parent's slot cannot be overwritten before the event itself is done;
//   * misses are settled before the sweep starts: whether an other-parent will still be in the ring when its child's
//     level comes is a property of the DAG (level_op_pinned, one look at the chain), so the level kernels mark such
//     parents, their row slices also go to a side table of SW_LEVEL_SIDE slots behind the ring — written round-robin
//     in the order the pinned events come (k_level_scatter) — and the child's descriptor points there; a parent of
//     an earlier launch is read from memory on the spot (its row is complete);
//   * what is left — a pinned parent still wanted SW_LEVEL_SIDE pinned events later — defers the event: the thread raises a flag every wave
//     reads with the next level's ring reads; in that case all waves drain their stores and meet, the deferred events
//     take both parents from L2, and the level's ring reads are repeated.
// ---------------------------------------------------------------------------------
template <int CB> struct RingVec;
template <> struct RingVec<2> { typedef int T __attribute__((ext_vector_type(2))); };
template <> struct RingVec<4> { typedef int T __attribute__((ext_vector_type(4))); };

template <int CB>
__device__ __forceinline__ void ring_read_parents(unsigned tag_a, unsigned val_a, unsigned tag_b, unsigned val_b, unsigned flag,
                                                  int& ta, typename RingVec<CB>::T& va, int& tb1, typename RingVec<CB>::T& vb, int& tb2, int& fl) {
    // program order = LDS order: the second tag of the other-parent is read BEHIND its value
    if constexpr (CB == 4)
        asm volatile("ds_read_b32 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b32 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b32 %4, %8\n\tds_read_b32 %5, %10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ta), "=&v"(va), "=&v"(tb1), "=&v"(vb), "=&v"(tb2), "=&v"(fl)
                     : "v"(tag_a), "v"(val_a), "v"(tag_b), "v"(val_b), "v"(flag) : "memory");
    else
        asm volatile("ds_read_b32 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b32 %2, %8\n\tds_read_b64 %3, %9\n\tds_read_b32 %4, %8\n\tds_read_b32 %5, %10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ta), "=&v"(va), "=&v"(tb1), "=&v"(vb), "=&v"(tb2), "=&v"(fl)
                     : "v"(tag_a), "v"(val_a), "v"(tag_b), "v"(val_b), "v"(flag) : "memory");
}

// (LDS words by byte address: a `volatile int*` into LDS becomes a FLAT access with `s_waitcnt vmcnt(0)` behind it)
__device__ __forceinline__ void lds_set(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ int lds_get(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

// a side slot takes ONE writer per phase (the {tag, value, tag} protocol is single-writer): the first claimant of a stamp wins
__device__ __forceinline__ bool lds_claim(unsigned addr, int stamp) {
    int old;
    asm volatile("ds_max_rtn_i32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(addr), "v"(stamp) : "memory");
    return old < stamp;
}

template <int CB>
__device__ __forceinline__ void ring_write(unsigned tag_addr, unsigned val_addr, typename RingVec<CB>::T v, int e) {
    const int none = -1;
    if constexpr (CB == 4)
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b128 %2, %3\n\tds_write_b32 %0, %4" :: "v"(tag_addr), "v"(none), "v"(val_addr), "v"(v), "v"(e) : "memory");
    else
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b64 %2, %3\n\tds_write_b32 %0, %4" :: "v"(tag_addr), "v"(none), "v"(val_addr), "v"(v), "v"(e) : "memory");
}

template <int CB>
__global__ void __launch_bounds__(1024)
k_cansee_stream(const int4* __restrict__ desc, const int* __restrict__ lev_start, const int* __restrict__ lev_pinbase, int nlev,
                int* L, int npad, int H, int chs, int first_event) {
    typedef typename RingVec<CB>::T V;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr int NS = 4;
    const int CH = 1 << chs;  // descriptors per staging chunk (power of two, >= npad = blockDim.x)
    const int smask = NS * CH - 1;
    const int nslot = npad * H;                              // ring slots; slot nslot = absent; behind it the side table
    const int nall = nslot + 1 + SW_LEVEL_SIDE;
    int4* dstage = (int4*)smem;                              // [NS * CH], descriptor i at i & smask
    V* vals = (V*)(dstage + (size_t)NS * CH);                // [nall]
    int* tags = (int*)(vals + (size_t)nall);                 // [nall]
    int* side_lock = tags + (size_t)nall;                    // [SW_LEVEL_SIDE]: the last phase that wrote the side slot (lds_claim)
    int* s_flag = side_lock + SW_LEVEL_SIDE;                 // [1]: 1 + the last level an event of which was deferred; by byte address only
    const int tid = threadIdx.x, BT = blockDim.x;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid & ~63);   // first event slot of this wave
    // XCD-aware column groups: workgroup b runs on XCD b % 8 (observed); give each XCD a
    // contiguous run of column groups so that its L2 assembles whole 128-byte lines of a row
    const int nblk = gridDim.x;
    const int grp = (nblk % 8 == 0) ? (blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8 : blockIdx.x;
    const int gcol0 = grp * CB;
    char* const Lcol = reinterpret_cast<char*>(L + gcol0);
    const unsigned rowb = (unsigned)npad * 4u;
    const int total = lev_start[nlev];
    const int last_desc = total > 0 ? total - 1 : 0;
    const unsigned stage_lds = (unsigned)(size_t)dstage;     // (the low half of a generic LDS address is the LDS byte address)
    const unsigned vals_lds = (unsigned)(size_t)vals, tags_lds = (unsigned)(size_t)tags, flag_lds = (unsigned)(size_t)s_flag;
    const unsigned lock_lds = (unsigned)(size_t)side_lock;
    const int own_lo = gcol0 * H;                            // own slots of the members whose columns these are: [own_lo, own_lo + CB·H)
    // this wave's pieces of chunk q -> slot q % NS (entries beyond the last descriptor hold a copy of it: never read)
    auto issue_chunk = [&](int q) {
        for (int base = wave0; base < CH; base += BT) {
            const long long gi = (long long)q * CH + base + (tid & 63);
            const unsigned dst = stage_lds + ((unsigned)(q % NS) * (unsigned)CH + (unsigned)base) * 16u;
            lds_dma_16(desc + (gi < total ? gi : last_desc), (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
        }
    };
    V none;
    none.x = -1; none.y = -1;
    if constexpr (CB == 4) { none.z = -1; none.w = -1; }
    for (int i = tid; i < nall; i += BT) { tags[i] = -1; vals[i] = none; }
    if (tid == 0) s_flag[0] = -1;
    for (int i = tid; i < SW_LEVEL_SIDE; i += BT) side_lock[i] = 0;
    // descriptor chunks 0 .. 2 resident, chunk 3 in flight
    for (int i = tid; i < 3 * CH; i += BT) dstage[i] = i < total ? desc[i] : make_int4(-1, -1, -1, 0);
    int pend_q = 3;      // the chunk whose pieces are in flight
    int n_since = 0;     // store instructions this wave has certainly issued behind its last piece of chunk pend_q
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue_chunk(pend_q);
    int s_cur = lev_start[0];
    int t_cur = lev_start[1];
    int t_nxt = nlev > 1 ? lev_start[2] : t_cur;
    int t_nn = nlev > 2 ? lev_start[3] : t_nxt;
    int4 d = s_cur + tid < t_cur ? dstage[(s_cur + tid) & smask] : make_int4(-1, -1, -1, 0);
    int pb_prev = 0, pb_cur = lev_pinbase[0];   // pinned events of the sweep in front of levels lv - 1, lv (their first side-table ranks)
    int4 dprev = make_int4(-1, -1, -1, 0);   // my event of the previous level, kept for the case it was deferred
    bool deferred = false;
    // levels 0 .. nlev - 1, then one empty level that serves the deferred events of the last one
    for (int lv = 0; lv <= nlev; ++lv) {
        const int t_n3 = lv + 4 <= nlev ? lev_start[lv + 4] : t_nn;   // (end of level lv + 3: the next iteration's t_nn)
        const int pb_nxt = lev_pinbase[lv + 1 <= nlev ? lv + 1 : nlev];
        const int n_cur = t_cur - s_cur;   // events in this level (uniform)
        const int n_nxt = t_nxt - t_cur;   // events in the next level
        // Level lv + 1 fetches the descriptors of level lv + 2: the chunk that level ends in must have landed — every wave's
        // pieces of it — behind THIS level's barrier, so each wave makes sure of its own pieces here: they have landed once at
        // most `n_since` memory instructions of the wave are outstanding.  The slot the following chunk goes to held chunk
        // need_q - 3: this level's fetches (level lv + 1's descriptors) touch chunks >= need_q - 2 (two levels span at most
        // 2 CH descriptors), and every wave is past the previous level's barrier, i.e. past its fetches.
        const int need_q = t_nn > 0 ? (t_nn - 1) >> chs : 0;
        while (need_q >= pend_q) {
            wait_vm_at_most(n_since);
            ++pend_q;
            issue_chunk(pend_q);
            n_since = 0;
        }
        const bool mine = wave0 < n_cur;   // (scalar) some event of the level falls to this wave
        const int slot = d.w & 0x3fff;
        const int ib = (d.w >> 14) & 0x7fff;
        const int ia = slot - 1 + ((d.w >> 29) & 1) * H;
        int ta = 0, tb1 = 0, tb2 = 0, fl;
        V va = none, vb = none;
        if (mine) ring_read_parents<CB>(tags_lds + 4u * (unsigned)ia, vals_lds + (unsigned)sizeof(V) * (unsigned)ia, tags_lds + 4u * (unsigned)ib,
                                        vals_lds + (unsigned)sizeof(V) * (unsigned)ib, flag_lds, ta, va, tb1, vb, tb2, fl);
        else fl = lds_get(flag_lds);
        // (the flag word only grows; it equals lv iff level lv - 1 raised it, and then no wave gets past the meetings below — to
        // where this level could raise it — before every wave has read it)
        if (__builtin_amdgcn_readfirstlane(fl) == lv) {
            // rare: events of level lv - 1 were deferred.  Every wave drains its stores, then the deferred events take
            // both parents from L2 (their rows were stored before this point, by whichever wave), then the ring is read again.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (deferred) {
                int pa[CB], pb[CB], v[CB];
                load_cols_sc1_and_wait<CB>(reinterpret_cast<const int*>(Lcol + (size_t)(unsigned)dprev.y * rowb), pa);
                load_cols_sc1_and_wait<CB>(reinterpret_cast<const int*>(Lcol + (size_t)(unsigned)dprev.z * rowb), pb);
                const int pslot = dprev.w & 0x3fff;
                const int rel = pslot - own_lo;
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int t = pa[c] > pb[c] ? pa[c] : pb[c];
                    v[c] = (rel >= c * H && rel < (c + 1) * H) ? dprev.x : t;   // own entry (swirld.py:220)
                }
                store_cols<CB>(reinterpret_cast<int*>(Lcol + (size_t)(unsigned)dprev.x * rowb), v);
                V vv;
                vv.x = v[0]; vv.y = v[1];
                if constexpr (CB == 4) { vv.z = v[2]; vv.w = v[3]; }
                ring_write<CB>(tags_lds + 4u * (unsigned)pslot, vals_lds + (unsigned)sizeof(V) * (unsigned)pslot, vv, dprev.x);
                if ((dprev.w >> 30) & 1) {
                    const unsigned sd = (unsigned)((pb_prev + tid) & (SW_LEVEL_SIDE - 1)), ps = (unsigned)(nslot + 1) + sd;
                    if (lds_claim(lock_lds + 4u * sd, 2 * lv + 1)) ring_write<CB>(tags_lds + 4u * ps, vals_lds + (unsigned)sizeof(V) * ps, vv, dprev.x);
                }
            }
            n_since = 0;   // (the drain: nothing of mine is outstanding but these stores)
            lds_barrier();
            if (mine) ring_read_parents<CB>(tags_lds + 4u * (unsigned)ia, vals_lds + (unsigned)sizeof(V) * (unsigned)ia, tags_lds + 4u * (unsigned)ib,
                                            vals_lds + (unsigned)sizeof(V) * (unsigned)ib, flag_lds, ta, va, tb1, vb, tb2, fl);
        }
        int4 dn = make_int4(-1, -1, -1, 0);
        if (wave0 < n_nxt && t_cur + tid < t_nxt) dn = dstage[(t_cur + tid) & smask];
        deferred = false;
        if (mine) {
            const bool act = d.x >= 0;
            bool ma = act & (ta != d.y), mb = act & !((tb1 == d.z) & (tb2 == d.z));
            if (__ballot(ma | mb)) {   // a parent of an earlier launch: its row is complete in memory
                if (ma & (d.y < first_event)) {
                    int t[CB];
                    load_cols_sc1_and_wait<CB>(reinterpret_cast<const int*>(Lcol + (size_t)(unsigned)d.y * rowb), t);
                    va.x = t[0]; va.y = t[1];
                    if constexpr (CB == 4) { va.z = t[2]; va.w = t[3]; }
                    ma = false;
                }
                if (mb & (d.z < first_event)) {
                    int t[CB];
                    load_cols_sc1_and_wait<CB>(reinterpret_cast<const int*>(Lcol + (size_t)(unsigned)d.z * rowb), t);
                    vb.x = t[0]; vb.y = t[1];
                    if constexpr (CB == 4) { vb.z = t[2]; vb.w = t[3]; }
                    mb = false;
                }
                n_since = 0;   // (the loads wait for everything of this wave)
            }
            bool hit = !(ma | mb);
            deferred = act & !hit;
            if (deferred) lds_set(flag_lds, lv + 1);
            const bool go = act & hit;
            if (go) {
                V vv;
                vv.x = va.x > vb.x ? va.x : vb.x;
                vv.y = va.y > vb.y ? va.y : vb.y;
                if constexpr (CB == 4) { vv.z = va.z > vb.z ? va.z : vb.z; vv.w = va.w > vb.w ? va.w : vb.w; }
                const int rel = slot - own_lo;
                if (__ballot((unsigned)rel < (unsigned)(CB * H))) {   // own entry (swirld.py:220): a member of these columns
                    if (rel >= 0 && rel < H) vv.x = d.x;
                    if (rel >= H && rel < 2 * H) vv.y = d.x;
                    if constexpr (CB == 4) {
                        if (rel >= 2 * H && rel < 3 * H) vv.z = d.x;
                        if (rel >= 3 * H && rel < 4 * H) vv.w = d.x;
                    }
                }
                *reinterpret_cast<V*>(Lcol + (size_t)(unsigned)d.x * rowb) = vv;
                ring_write<CB>(tags_lds + 4u * (unsigned)slot, vals_lds + (unsigned)sizeof(V) * (unsigned)slot, vv, d.x);
                if (__ballot((d.w >> 30) & 1)) {   // pinned: a child will look for this row slice after it has left the ring
                    if ((d.w >> 30) & 1) {   // (more than SW_LEVEL_SIDE pinned events in one level: the later claimant's child defers)
                        const unsigned sd = (unsigned)((pb_cur + tid) & (SW_LEVEL_SIDE - 1)), ps = (unsigned)(nslot + 1) + sd;   // (rank: pinned events head their level)
                        if (lds_claim(lock_lds + 4u * sd, 2 * lv + 2)) ring_write<CB>(tags_lds + 4u * ps, vals_lds + (unsigned)sizeof(V) * ps, vv, d.x);
                    }
                }
            }
            n_since += __ballot(go) != 0 ? 1 : 0;   // (a store instruction with at least one lane has certainly been issued: a LOWER bound)
        }
        lds_barrier();
        s_cur = t_cur; t_cur = t_nxt; t_nxt = t_nn; t_nn = t_n3;
        dprev = d;
        d = dn;
        pb_prev = pb_cur; pb_cur = pb_nxt;
    }
}

template <int NW, int C, int F, int H, bool WIDE>
__global__ void __launch_bounds__(64 * NW + 64)
k_cansee_chunks(const int4* __restrict__ cdesc, const int* __restrict__ chain_start, const int* __restrict__ chain_ev,
                const int* __restrict__ bnd, ChunkEv ce, int a0_all, int exact_chunk, int n_members,
                int* L, int halo_row0, int halo_cap, unsigned* prov, unsigned gate_limit, int* err) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr int npad = 64 * NW;
    constexpr int NT = npad;             // worker lanes = members
    constexpr int NCG = npad / C;        // column groups = workgroups per chunk
    constexpr int PL = C / 2;            // 16-byte ring planes: two {value, tag} pairs each
    constexpr int SPIN_LIMIT = 1 << 27;
    constexpr int hm = H - 1, fm = F - 1;
    static_assert(C == 2 || C == 4, "columns per lane");
    static_assert((H & hm) == 0 && (F & fm) == 0 && H <= 64 && H >= 8, "ring / FIFO depths are powers of two; H > 6 store instructions");
    int4* fifo = (int4*)smem;                                  // [F][npad]
    int4* ring = fifo + (size_t)F * npad;                      // [PL][H][npad] {v, tag, v', tag}
    int* filled = (int*)(ring + (size_t)PL * H * npad);        // [npad]
    int* taken = filled + npad;                                // [npad]
    const int tid = threadIdx.x;
    const bool loader = tid >= NT;
    const int ll = tid - NT;
    // workgroup b runs on XCD b % 8 (observed, used for locality only): each XCD keeps NCG / 8 consecutive
    // column groups = npad / 8 consecutive columns of every chunk, i.e. whole 128-byte lines of a row
    int k, cg;
    if (exact_chunk >= 0) {
        k = exact_chunk;
        cg = (NCG % 8 == 0) ? (blockIdx.x % 8) * (NCG / 8) + blockIdx.x / 8 : blockIdx.x;
        // second sweep of a chunk from final rows: only when the repair by gathers would cost more
        if (prov[k] <= gate_limit) return;
    } else if constexpr (NCG % 8 == 0) {
        const int slot = blockIdx.x / 8;
        k = slot / (NCG / 8);
        cg = (blockIdx.x % 8) * (NCG / 8) + slot % (NCG / 8);
    } else {
        k = blockIdx.x / NCG;
        cg = blockIdx.x % NCG;
    }
    const int col0 = cg * C;
    const bool exact = exact_chunk >= 0;
    const int a_k = ce.a[k];
    const int w_k = exact ? a_k : ce.w[k];
    const int a0 = exact ? a_k : a0_all;          // rows below a0 are final in memory
    const int* pw = bnd + (size_t)(exact ? 2 * k + 1 : 2 * k) * npad;   // chain positions of the window start
    const int* pa = bnd + (size_t)(2 * k + 1) * npad;                   // ... of the first stored row
    const int* pe = bnd + (size_t)(2 * k + 3) * npad;                   // ... of the end of the chunk
    // the scratch rows of the halo live in the SAME allocation as the table, behind its last row (one base
    // pointer: a row select, global — not flat — stores, 32-bit offsets while the table stays below 4 GB)
    const int halo_k0 = halo_row0 + k * halo_cap - w_k;   // halo event e -> row halo_k0 + e
    for (int i = tid; i < PL * npad * H; i += blockDim.x) ring[i] = make_int4(-1, -1, -1, -1);  // tag -1: empty
    for (int i = tid; i < npad; i += blockDim.x) { const int p = pw[i]; filled[i] = p; taken[i] = p; }
    __syncthreads();
    if (loader) {
        constexpr int B = F / 2;
        constexpr int MAXBUF = 32;
        constexpr int G = (NW * B <= MAXBUF) ? NW : MAXBUF / B;
        static_assert(G >= 1 && NW % G == 0, "loader groups");
        int fl[NW], pend[NW], cs[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int kk = ll + 64 * j;
            fl[j] = pw[kk];
            pend[j] = pe[kk];
            cs[j] = chain_start[kk];
        }
        for (int spins = 0;; ++spins) {
            bool more = false;
            unsigned need = 0;
            if (spins > SPIN_LIMIT) { if (ll == 0) atomicExch(err, 2); break; }
            SW_CBAR();
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (fl[j] < pend[j]) {
                    more = true;
                    const int tk = taken[ll + 64 * j];
                    if (fl[j] - tk <= F - B) need |= 1u << j;
                }
            }
            if (!__ballot(more)) break;
            if (!__ballot(need != 0)) { __builtin_amdgcn_s_sleep(2); continue; }
#pragma unroll
            for (int g0 = 0; g0 < NW; g0 += G) {
                // branch-free loads with clamped positions (see k_cansee_flow)
                int bx[G * B], by[G * B], bz[G * B];
#pragma unroll
                for (int jj = 0; jj < G; ++jj) {
                    const int j = g0 + jj;
                    const int last = pend[j] > 0 ? pend[j] - 1 : 0;
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        const int pos = fl[j] + u < pend[j] ? fl[j] + u : last;
                        const int4 t = cdesc[(size_t)cs[j] + pos];
                        bx[jj * B + u] = t.x; by[jj * B + u] = t.y; bz[jj * B + u] = t.z;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < G; ++jj) {
                    const int j = g0 + jj;
                    if ((need >> j) & 1u) {
                        const int kk = ll + 64 * j;
                        int nf = fl[j];
#pragma unroll
                        for (int u = 0; u < B; ++u)
                            if (fl[j] + u < pend[j]) {
                                fifo[(size_t)((fl[j] + u) & fm) * npad + kk] = make_int4(bx[jj * B + u], by[jj * B + u], bz[jj * B + u], 0);
                                nf = fl[j] + u + 1;
                            }
                        SW_CBAR();  // the entries are written before the count that publishes them (DS ops stay in order)
                        filled[kk] = nf;
                        SW_CBAR();
                        fl[j] = nf;
                    }
                }
            }
        }
        return;
    }
    // ---- workers: lane = member m, columns col0 .. col0 + C - 1.  The trip is kept branch-light: the
    // classification of an other-parent, the own-column overwrite and the provisional test are selects.
    const int m = tid;
    const int own = m - col0;            // index of the member's own column among mine (outside [0, C): none)
    int p = pw[m];
    const int pend = pe[m];
    int mine[C], oth[C], nm[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { mine[c] = -1; oth[c] = -1; nm[c] = (c == own) ? 0 : -1; }   // e | nm[c] = e in the own column, -1 elsewhere
    int ev = -1, opar = -1, ridx = 0;
    int mode = 2;                        // 0: poll the ring, 1: final row in memory, 2: `oth` already holds the other-parent's values (leaf, root)
    bool have = false;
    unsigned n_prov = 0;
    const bool count_prov = !exact && w_k > a0;
    // F_c = the last event of member c before the window: the largest value column c can hold below w_k.  A
    // computed value equal to it is FINAL although it lies outside the window (nothing of c in between can
    // be missing) — what keeps the columns of members silent for longer than the halo from being provisional
    // whenever a leaf (their newest event as somebody's other-parent) was reached.
    int Fc[C], wthr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int pc = pw[col0 + c];
        Fc[c] = (count_prov && pc > 0) ? chain_ev[chain_start[col0 + c] + pc - 1] : -1;
        wthr[c] = col0 + c < n_members ? w_k : (int)0x80000000;   // padded columns are never provisional
    }
    // where the values of event e for my columns live: its row of the table, or a scratch row (w_k <= e < a_k)
    char* const Lcol = reinterpret_cast<char*>(L + col0);
    auto row_ptr = [&](int e) -> int* {
        const int row = ((e >= a_k) | (e < a0)) ? e : halo_k0 + e;
        if (WIDE) return reinterpret_cast<int*>(Lcol + (size_t)row * (size_t)(npad * 4));
        return reinterpret_cast<int*>(Lcol + (unsigned)row * (unsigned)(npad * 4));
    };
    if (p > 0 && p < pend) {
        // the member's last event before the window: a final row (zone i) or a leaf (zone ii)
        const int prev = chain_ev[chain_start[m] + p - 1];
        if (prev < a0) {
            const typename ColVec<C>::T t = *reinterpret_cast<const typename ColVec<C>::T*>(L + (size_t)prev * npad + col0);
            mine[0] = t.x; mine[1] = t.y;
            if constexpr (C == 4) { mine[2] = t.z; mine[3] = t.w; }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) mine[c] = prev | nm[c];
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(mine[c]), "+v"(Fc[c]));  // the prologue loads are complete before the loop
    for (int spins = 0;; ++spins) {
        if (spins > SPIN_LIMIT) { if ((tid & 63) == 0) atomicExch(err, 1); break; }
        SW_CBAR();
        int4 pr[PL];
#pragma unroll
        for (int h = 0; h < PL; ++h) pr[h] = ring[(size_t)h * H * npad + ridx];
        const int fcnt = filled[m];
        SW_CBAR();  // the count is read before the entry it publishes (DS operations stay in order)
        const int4 nd = fifo[((p + (have ? 1 : 0)) & fm) * npad + m];
        SW_CBAR();
        const int o = opar;
        int diff = 0;                    // every tag of every plane must be the wanted event
#pragma unroll
        for (int h = 0; h < PL; ++h) diff |= (pr[h].y ^ o) | (pr[h].w ^ o);
        // (bitwise operators on purpose: short-circuit ones become branches)
        const bool hit = (mode == 0) & (diff == 0);   // (an event outside the window is never in the ring)
        int other[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int rv = (c & 1) ? pr[c >> 1].z : pr[c >> 1].x;
            other[c] = hit ? rv : oth[c];            // (oth is -1 in every column while the ring is polled)
        }
        bool ready = have & ((mode == 2) | hit);
        // rare: the other-parent's row comes from memory (a final row of an earlier launch, or a ring slot
        // that was reused — plane 0 is written first, so its tag is the newest: the row, real or halo, is
        // then >= H store instructions old)
        const bool from_mem = have & !ready & ((mode == 1) | (pr[0].y > o));
        if (__ballot(from_mem)) {
            if (from_mem) {
                load_cols_sc1_and_wait<C>(row_ptr(o), other);
                ready = true;
            }
        }
        if (ready) {
            const int e = ev;
            int v[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // maxi(): index order == height order on one chain; the own entry (swirld.py:220) is e itself,
                // larger than every ancestor
                const int oe = e | nm[c];
                int t = mine[c] > other[c] ? mine[c] : other[c];
                t = t > oe ? t : oe;
                v[c] = t;
                mine[c] = t;
            }
            store_cols<C>(row_ptr(e), v);
            const int slot = (p & hm) * npad + m;
#pragma unroll
            for (int h = 0; h < PL; ++h) ring[(size_t)h * H * npad + slot] = make_int4(v[2 * h], e, v[2 * h + 1], e);
            if (count_prov) {   // a stored row of a chunk with unknown parents: count the stores that hold something to repair
                int pv = 0;   // (bitwise on purpose: short-circuit operators become branches in this loop)
#pragma unroll
                for (int c = 0; c < C; ++c) pv |= (int)(v[c] < wthr[c]) & (int)(v[c] != Fc[c]);
                n_prov += (unsigned)(pv & (int)(e >= a_k));
            }
            ++p;
            have = false;
        }
        if (!have && p < pend && p < fcnt) {  // nd is the descriptor of position p
            ev = nd.x;
            opar = nd.y;
            ridx = ((nd.z >> 10) & hm) * npad + (nd.z & 1023);
            have = true;
            taken[m] = p + 1;  // the slot may be refilled from here on
            // zone of the other-parent: (i) final row in memory, (ii) a leaf {creator: event}, (iii) in the window; a root has none
            const int y = nd.y;
            const bool leaf = y >= a0 && y < w_k;
            mode = y < 0 ? 2 : (y < a0 ? 1 : (leaf ? 2 : 0));
            const int oc = (nd.z & 1023) - col0;   // the other-parent's own column among mine
            const int lv = leaf ? y : -1;
#pragma unroll
            for (int c = 0; c < C; ++c) oth[c] = (c == oc) ? lv : -1;
        }
        // at most H - 4 store instructions of this wave in flight (the reuse argument of k_cansee_flow)
        if constexpr (H >= 32) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if constexpr (H >= 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (!__ballot(p < pend)) break;
        // (measured and dropped in round 4, profiles/r04g_*: starved waves sleeping 128 cycles per unproductive trip to leave the
        // CU to the round-loop kernels: the pass gets 0.4 % slower, hot-member hashgraphs 3 %)
    }
    if (!exact && w_k > a0) {
        unsigned tot = n_prov;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += (unsigned)__shfl_xor((int)tot, off);
        if ((tid & 63) == 0 && tot) atomicAdd(&prov[k], tot);   // (counted in STORES: up to C entries each)
    }
}

// Repair of the provisional entries of chunk k (tests/model_chunks.py fixup()): rows [a, b), window start w.
//   T[e][c] = max(V[e][c], max over members m of T[E_m(e)][c]),  E_m(e) = F_m if V[e][m] >= w else V[e][m],
// F_m = last event of m before w; only entries E_m(e) >= a0 matter (rows below a0 were read in full by the
// sweep).  Every E_m(e) < w lies in an earlier chunk, repaired before this one (stream order).  One wave
// per event; runs only when the sweep counted 0 < provisional entries <= limit.
template <int NW>
__global__ void __launch_bounds__(256)
k_cansee_fixup(const int* __restrict__ chain_start, const int* __restrict__ chain_ev, const int* __restrict__ pw,
               int a0, int w, int a, int b, int n_members, int* L, const unsigned* __restrict__ prov_k, unsigned limit,
               unsigned* fixed_out) {
    constexpr int npad = 64 * NW;
    __shared__ int s_F[npad];
    const unsigned cnt = *prov_k;
    if (cnt == 0 || cnt > limit) return;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int p = pw[i];
        s_F[i] = p > 0 ? chain_ev[chain_start[i] + p - 1] : -1;
    }
    __syncthreads();
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned fixed = 0;
    for (int e = a + wave; e < b; e += nwaves) {
        int V[NW], E[NW];
        u64 pm[NW];
        bool any = false;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            V[j] = L[(size_t)e * npad + j * 64 + lane];
            pm[j] = __ballot(V[j] < w && V[j] != s_F[j * 64 + lane] && j * 64 + lane < n_members);   // (== F_c: final, see the sweep)
            any = any || pm[j] != 0;
            E[j] = V[j] >= w ? s_F[j * 64 + lane] : V[j];
            if (E[j] < a0) E[j] = -1;
        }
        if (!any) continue;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u64 rest = pm[j];
            while (rest) {
                const int cl = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const int c = j * 64 + cl;
                int t = -1;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj) {
                    const int x = E[jj] >= 0 ? L[(size_t)E[jj] * npad + c] : -1;
                    t = x > t ? x : t;
                }
                t = wave_max_i32(t);
                if (lane == cl && t > V[j]) { V[j] = t; L[(size_t)e * npad + c] = t; ++fixed; }
            }
        }
    }
    if (fixed_out && fixed) atomicAdd(fixed_out, fixed);
}

// chain positions of the sub-batch cuts: out[i][m] = number of member m's events with index < cut[i]
__global__ void __launch_bounds__(1024)
k_chain_bounds(const int* __restrict__ chain_start, const int* __restrict__ chain_cnt,
               const int* __restrict__ chain_ev, const long long* __restrict__ cuts, int npad, int* out) {
    const int m = threadIdx.x, i = blockIdx.x;
    const int x = (int)cuts[i];
    const int cs = chain_start[m];
    int a = 0, b = chain_cnt[m];
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (chain_ev[cs + mid] < x) a = mid + 1; else b = mid;
    }
    out[(size_t)i * npad + m] = a;
}

// Ingest: chain pool entry and pool-indexed chain descriptor {event, other-parent,
// creator(op) | (seq(op) & 63) << 10, self-parent} of the events [first, first + K).
__global__ void k_chain_scatter(const int* __restrict__ cr, const int* __restrict__ sp, const int* __restrict__ op,
                                const int* __restrict__ seq, const int* __restrict__ chain_start, int first, int K,
                                int* chain_ev, int4* cdesc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int e = first + i;
    const int o = op[e];
    int w = 0;
    if (o >= 0) w = cr[o] | ((seq[o] & 63) << 10);
    const size_t at = (size_t)chain_start[cr[e]] + seq[e];
    chain_ev[at] = e;
    cdesc[at] = make_int4(e, o, w, sp[e]);
}

// Ingest of a SMALL append (a Node's gossip step: a handful of events): one packed record per event,
// one host-to-device copy, this one kernel — parent arrays, height, chain position, timestamp,
// signature, coin bit, round = "not divided", chain pool entry, chain descriptor, chain length.
struct SmallRec {
    int cr, sp, op, seq, ht, at, w, pad;  // at = pool index of the event; w = creator(op) | (seq(op) & 63) << 10
    double t;
    unsigned char sig[64];
};
static_assert(sizeof(SmallRec) == 104, "packed ingest record");

__global__ void k_ingest_small(const SmallRec* __restrict__ rec, int first, int K, int* cr, int* sp, int* op, int* seq, int* ht,
                               double* t, unsigned char* sig, unsigned char* coin, int* round, int* chain_ev, int4* cdesc,
                               int* chain_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const SmallRec r = rec[i];
    const int e = first + i;
    cr[e] = r.cr; sp[e] = r.sp; op[e] = r.op; seq[e] = r.seq; ht[e] = r.ht;
    t[e] = r.t;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rec[i].sig);
    uint32_t* dst = reinterpret_cast<uint32_t*>(sig + (size_t)e * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[k] = src[k];
    coin[e] = r.sig[0] >> 7;  // swirld.py:272
    round[e] = -1;
    chain_ev[r.at] = e;
    cdesc[r.at] = make_int4(e, r.op, r.w, r.sp);
    atomicMax(&chain_cnt[r.cr], r.seq + 1);
}

// Ingest: the part of is_valid_event's parent check (swirld.py:104-108) that needs a lookup —
// "the other-parent is by another member" — for a bulk append, on the device; err = smallest
// offending event index (INT_MAX: none).  (Arity, order and the fork / same-creator test of the
// self-parent only need per-member tables and run in the host pass.)
__global__ void k_validate_other_parent(const int* __restrict__ cr, const int* __restrict__ op, int first, int K, int* err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int e = first + i;
    const int o = op[e];
    if (o >= 0 && cr[o] == cr[e]) atomicMin(err, e);
}

// Ingest: coin bit of every event = top bit of the first signature byte (swirld.py:272)
__global__ void k_coin_bits(const unsigned char* __restrict__ sig, int first, int K, unsigned char* coin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) coin[first + i] = sig[(size_t)(first + i) * 64] >> 7;
}

// ---------------------------------------------------------------------------------
// Round loop.  One iteration = k_resolve_band -> k_tally_*; ~1 iteration per round.
// Round-synchronous form: round[e] >= r+1  <=>  SS_r(e), evaluated with the thresholds
// lo[r][.] (checked on the CPU in tests/model_bulk.py).
//
// All loop state is double-buffered by iteration parity (`par`): a kernel reads buffer `par`
// and writes buffer `1 - par`, so the resolve step can be REPLICATED in every workgroup of
// the band kernel (each needs its results) while only workgroup 0 stores them — this removes
// a serial single-workgroup launch from every iteration.
// ---------------------------------------------------------------------------------
struct LoopBufs {
    RState* st;       // [2]
    int* lo_r;        // [2][npad] thresholds of the round being resolved
    int* cur;         // [2][npad] chain position of the member's next candidate window
    int* unres;       // [2][npad] member still searching its first round-(r+1) event
    int* lo_next;     // [2][npad] lo[r+1][c] found so far in this round
    int* pos_next;    // [2][npad] ... and its chain position
    int* evalround;   // [2][npad] round in which the member's chain was last exhausted
    int* evalpos;     // [2][npad] ... and up to which position
    u64* found64;     // [2][npad] the member's first candidate whose tally passed, as ONE key an atomicMin can order:
                      // {event << 32 | slot << 26 | (last candidate of the window that would follow it) - event}
                      // (low field 0x3ffffff: not known; ~0: no candidate passed).  The minimum over events is the
                      // minimum over slots (one chain), and k_resolve_band gets the event and the look-ahead without
                      // two dependent look-ups (candidate table, chain index) at its head.
    int* farslot;     // [2][npad] smallest candidate slot that was FAR (not tallied; INF: none)
    int* force;       // [2][npad] tally the member's cursor candidate even though it is far
    int* cand;        // [2][npad][64] candidate table of the next tally launch: entry 1 + j = event of
                      // slot j (chain position cursor + j * stride); -1 = none
    int* gallop;      // [2][npad] window stride (bits 0-7; 1 = contiguous) and consecutive windows without
                      // a passing candidate (bits 8+) of the member in the current round
    int* treecnt;     // [npad] tallies evaluated for the member in this run (k_tally_tree: one workgroup owns a member; read back with the loop state)
    int* front;       // [npad] per member the last round r with lo[r][member] finite (-1 none): where the next call resumes
    u64* dbg;         // diagnostics (SW_DEBUG_CLOCKS=1): [iteration][32] wall-clock stamps, else null
    u64* dbg_blk;     // diagnostics (SW_DEBUG_CLOCKS=3): [iteration][2][2048] end time of every workgroup of the two loop kernels, else null
    int dbg_minor;    // ... 1: every phase of the resolve step is stamped (each stamp drains the wave: SW_DEBUG_CLOCKS=2 keeps only entry / band start / end)
};

// ONE hashgraph's round loop over `parts` linked contexts (one per GPU): the replicated resolve step runs in
// every context, the band events and the members are DEALT to the parts, and what a part produces — the band events' mask
// rows and popcounts, their round numbers and sees-masks, its members' verdict words — is stored into the tables of EVERY
// part (peer-mapped device memory: plain stores and atomics over xGMI, no collective inside an iteration).  The contexts'
// streams meet at the two kernel boundaries of an iteration (events; swirld_hip.hip, enqueue_iteration).
#define SW_MAX_PARTS 8
struct SplitDst {
    int part, parts;              // this part's share: band groups / members part, part + parts, ...
    int ndst;                     // tables stored to: `parts` (1 when ONE context plays the parts one behind the other: SW_SPLIT_EMULATE)
    u64* Mb[SW_MAX_PARTS];        // row 1 of every part's band-mask table
    int* Pc[SW_MAX_PARTS];        // ... and of its popcounts
    u64* S[SW_MAX_PARTS];         // sees-masks
    int* round[SW_MAX_PARTS];     // round numbers
    u64* found64[SW_MAX_PARTS];   // verdict words of the tally
    int* farslot[SW_MAX_PARTS];
};

// Kernel arguments are fetched lazily by the compiler (an s_load right before the first use, one
// per 64-byte line of the kernarg segment), and every such fetch is a dependent scalar-cache miss
// in the middle of a latency-bound kernel.  Pinning makes all of them arrive with the first one.
__device__ __forceinline__ void pin_arg(const void* p) { asm volatile("" ::"s"((unsigned long long)p)); }
__device__ __forceinline__ void pin_arg(int v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ void pin_arg(uint32_t v) { asm volatile("" ::"s"(v)); }

// phase stamps of the round-loop kernels (100 MHz constant clock); only with SW_DEBUG_CLOCKS=1.
// The wait makes the stamp mean "everything issued so far has completed".
#define SW_DBG_MAX_ITERS 4096
#define SW_STAMP(cond, it, slot)                                                          \
    do {                                                                                  \
        if (B.dbg && (cond) && (it) < SW_DBG_MAX_ITERS) {                                 \
            __builtin_amdgcn_s_waitcnt(0);                                                \
            B.dbg[(size_t)(it) * 32 + (slot)] = wall_clock64();                           \
        }                                                                                 \
    } while (0)

// Start of a round-loop run: the loop state and the per-member buffers in ONE launch (a small
// call would otherwise pay five separate copies / fills, ~10 us each).
__global__ void __launch_bounds__(1024)
k_loop_init(LoopBufs B, int npad, int r_start, int N, int ncap, const int* __restrict__ visible_len, int* chain_len, int eval_src, int fin_from,
            int iter_base) {
    // chain lengths visible to this run = the sub-batch's row of the cut table (already on the device)
    for (int i = threadIdx.x; i < npad; i += blockDim.x) { chain_len[i] = visible_len[i]; B.treecnt[i] = 0; }
    if (eval_src)   // the previous run ended on an odd iteration: its exhaustion marks are in half 1, this run reads half 0
        for (int i = threadIdx.x; i < npad; i += blockDim.x) { B.evalround[i] = B.evalround[npad + i]; B.evalpos[i] = B.evalpos[npad + i]; }
    if (threadIdx.x == 0) {
        RState t{};
        t.r = r_start;
        t.N = N;
        t.ncap = ncap;
        t.fin_from = fin_from;
        t.pad_ = iter_base;   // (diagnostics only: iterations of this context before this loop — the index base of the phase stamps; the host counts them and resets the count on a rewind)
        B.st[0] = t;
    }
    for (int i = threadIdx.x; i < 2 * npad; i += blockDim.x) {
        B.unres[i] = 0;
        B.found64[i] = ~0ull;
        B.farslot[i] = SW_INF;
        B.force[i] = 0;
        B.gallop[i] = 1;
    }
}

// Step 1 (replicated, one thread per member): consume the tally results, advance the
// per-member cursors, commit lo[r+1] when every member is resolved, enter the next round that
// has work, derive the band.  Step 2: threshold masks of the band events, Mb[k-mlo] bit c_ =
// (L[k][c_] >= lo[r][c_]) = "the latest event of c_ that k sees has round >= r" (one wave per
// band event, NW ballots).
template <int NW, bool FAST, bool SPLIT = false>
__global__ void __launch_bounds__(1024)
k_resolve_band(LoopBufs B, int par, int npad, int K, int gallop_after, int skip, int NEARCAP, int MCAP, int Rcap,
               const int* __restrict__ chain_start, const int* __restrict__ chain_len,
               const int* __restrict__ chain_ev, int* lo, int* lopos,
               const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ op, u64* Mb,
               int* __restrict__ round_out, u64* __restrict__ S_out, int* __restrict__ Pc, const SplitDst* __restrict__ sdp) {
    static_assert(!(FAST && SPLIT), "the split form takes the generic band path");
    // (the parts' tables come through a POINTER: by value they were 400 bytes of kernel arguments on every launch of the plain
    // loop too, and the host, which enqueues an iteration about as fast as the device runs it, got 1.5 % slower)
    const SplitDst& sd = *(SPLIT ? sdp : reinterpret_cast<const SplitDst*>(B.st));
    __shared__ int s_red[2][4][16];  // [parity][quantity][wave]: per-wave partial results
    __shared__ int s_thr[1024];
    __shared__ int s_ln[1024];    // lo[r+1][b] when member b is resolved for this round
    __shared__ int s_res[1024];   // ... and whether it is
    __shared__ int s_cp[1024];    // chain_ev index of b's cursor candidate (-1: chain exhausted)
    __shared__ int s_ce[1024];    // chain_ev index one past b's last visible event
    __builtin_amdgcn_s_setprio(3);  // critical path: win issue arbitration against the can_see sweep
    pin_arg(B.st); pin_arg(B.lo_r); pin_arg(B.cur); pin_arg(B.unres); pin_arg(B.lo_next); pin_arg(B.pos_next);
    pin_arg(B.evalround); pin_arg(B.evalpos); pin_arg(B.found64); pin_arg(B.farslot); pin_arg(B.force); pin_arg(B.dbg);
    pin_arg(par); pin_arg(npad); pin_arg(K); pin_arg(skip); pin_arg(NEARCAP); pin_arg(MCAP); pin_arg(Rcap);
    pin_arg(chain_start); pin_arg(chain_len); pin_arg(chain_ev); pin_arg(lo); pin_arg(lopos);
    pin_arg(L); pin_arg(cr); pin_arg(op); pin_arg(Mb); pin_arg(Pc); pin_arg((int)blockDim.x); pin_arg((int)gridDim.x);
    // workgroups of max(npad, 256) threads (enqueue_iteration): a COMPILE-TIME wave count — with a run-time one the three
    // reductions below were general loops whose remainder form (4 waves < the unroll factor of 8) read LDS one dependent word
    // at a time (round 5, from the ISA)
    constexpr int nthr = NW * 64 < 256 ? 256 : NW * 64;
    // (measured and dropped in round 4, profiles/r04d_*: an extra wave per workgroup that touches the rows of the previous
    // band + one round while the member threads resolve.  The stamped block's band phase went from 3.4 to 2.1 us and the
    // pass did not move: the band phase is bound by the bytes of the whole band, not by the latency of a wave's loads.)
    const RState* si = B.st + par;
    RState* so = B.st + (1 - par);
    const bool writer = blockIdx.x == 0;
    const int c = threadIdx.x;
    const unsigned in = (unsigned)(par * npad), out = (unsigned)((1 - par) * npad);   // (32-bit offsets: scalar base + vector offset addressing)
    const bool member = (NW >= 4) || c < npad;   // (workgroups of max(npad, 256) threads: every thread is a member from 256 members on)
    // First memory round trip: the loop state and every per-member value whose address does not
    // depend on it, issued together BEFORE the first branch (a load behind an early return cannot
    // be hoisted by the compiler and would cost a dependent round trip of its own).  The per-member loads are
    // UNCONDITIONAL, from a clamped index: fourteen `member ? x[c] : d` are fourteen exec-masked blocks with a taken
    // branch each (round 5, from the ISA), and the threads beyond npad take the defaults by a select afterwards.
    const int s_done = si->done;
    int r = si->r;
    const int iter = si->iter;
    const int N = si->N;
    const int s_mlo = si->mlo, s_mhi = si->mhi, s_ncap = si->ncap;
    const int fin_from = si->fin_from;
    const int dbg_it = iter + si->pad_;   // (diagnostics: stamps are indexed by the iteration of the CONTEXT — the loops of a call's sub-batches one behind the other)
    const int cq = member ? c : 0;
    const int cs = chain_start[cq];
    const int clen_ld = chain_len[cq];  // events of member c visible to this run
    const int un_ld = B.unres[in + cq];
    int curc = B.cur[in + cq];
    const u64 fev_ld = B.found64[in + cq];
    const int jf_ld = B.farslot[in + cq];
    const int frc_ld = B.force[in + cq];
    const int gsv_ld = B.gallop[in + cq];
    const int evr_ld = B.evalround[in + cq], evp_ld = B.evalpos[in + cq];
    const int in_lo_next = B.lo_next[in + cq];
    const int in_pos_next = B.pos_next[in + cq];
    const int thr_ld = B.lo_r[in + cq];
    const int clen = member ? clen_ld : 0;
    int un = member ? un_ld : 0;
    const u64 fev = member ? fev_ld : ~0ull;
    const int fnd = fev == ~0ull ? SW_INF : (int)((fev >> 26) & 63);   // smallest candidate slot whose tally passed
    const int jf = member ? jf_ld : SW_INF;
    int frc = member ? frc_ld : 0;
    const int gsv = member ? gsv_ld : 1;
    int evr_now = member ? evr_ld : -1, evp_now = member ? evp_ld : 0;
    int thr = member ? thr_ld : SW_INF;
    const bool stamp = c == 0 && blockIdx.x == 1;  // a block that does not publish the state
    const int sb = 0;
    if (B.dbg && stamp && !s_done && dbg_it < SW_DBG_MAX_ITERS) B.dbg[(size_t)dbg_it * 32 + sb] = wall_clock64();
    SW_STAMP(stamp && !s_done, dbg_it, sb + 1);
    if (s_done) {
        if (writer && c == 0) *so = *si;
        return;
    }
    // (handing these three over with the loop state — stored by the writer block, read with the first round
    // trip — was measured: the writer's extra dependent loads lengthen the kernel by more than the round trip
    // saved here, 128.1 -> 126.3 M events/s)
    // (clamped rows, unconditional loads, selects afterwards: see above)
    const int rq1 = r + 1 < Rcap ? r + 1 : Rcap - 1, rq2 = r + 2 < Rcap ? r + 2 : Rcap - 1;
    const int lo_r1_ld = lo[(unsigned)(rq1 * npad + cq)];
    const int lo_r2_ld = lo[(unsigned)(rq2 * npad + cq)];
    const int lopos_r1_ld = lopos[(unsigned)(rq1 * npad + cq)];
    const int lo_r1 = (member && r + 1 < Rcap) ? lo_r1_ld : SW_INF;
    const int lo_r2 = (member && r + 2 < Rcap) ? lo_r2_ld : SW_INF;
    const int lopos_r1 = (member && r + 1 < Rcap) ? lopos_r1_ld : 0;
    int my_lo_next = (iter > 0 && member) ? in_lo_next : SW_INF;
    int my_pos_next = (iter > 0 && member) ? in_pos_next : 0;
    int mlo = s_mlo, mhi = s_mhi;
    int ncap = s_ncap;
    // tallies evaluated by the previous launch for this member: the slots before its first far one
    int evaluated = 0;
    int spec_cur = -1, spec_last = -1;
    int far_wait = 0;  // the cursor candidate is FAR: decide it by inheritance below
    // GALLOPING (SW_GALLOP, tests/model_bulk.py bulk_rounds_v3): after `gallop_after` consecutive
    // windows without a passing candidate the member's window is strided (positions cursor,
    // cursor + K, ...).  The predicate is monotone along the chain: a failing slot rules out
    // everything before it, a passing slot f > 0 brackets the first passing position in
    // (slot f-1, slot f], which one contiguous window covers.  A strided window falls back to a
    // contiguous one at a far candidate and at the end of the chain.
    // WINDOW OFFSET (SW_SKIP, tests/model_bulk.py bulk_rounds_v3(skip=...)): the window of a fresh
    // round starts `skip` positions after the cursor (the first positions after a witness practically
    // never pass; the first passing position lies 13.8 +- 3.7 after the cursor at 256 members), which
    // spares most of the rounds that need a second look with K = 28.  A passing or far slot 0 of such
    // a window only brackets the first passing position in [cursor, cursor + skip]: the member looks
    // again from the cursor; a failing slot 0 rules out the skipped positions as well.
    int strd = gsv & 0xff, miss = (gsv >> 8) & 0xff, skp = (gsv >> 16) & 0xff;
    bool refined = false;
    if (iter > 0 && un && skp) {
        if ((fnd == 0 && jf != 0) || jf == 0) {  // look again from the cursor
            evaluated = jf != SW_INF ? jf : (clen - curc < K ? clen - curc : K);  // (tallies of the offset window)
            curc -= skp;
            refined = true;
        }
        skp = 0;
    }
    if (iter > 0 && un && !refined) {
        // slots offered by the previous launch: positions curc + j * strd < clen, j < K
        const int offered = strd == 1 ? (clen - curc < K ? clen - curc : K)
                                      : ((clen - 1 - curc) / strd + 1 < K ? (clen - 1 - curc) / strd + 1 : K);
        evaluated = jf != SW_INF ? jf : offered;
        if (fnd != SW_INF && fnd < jf) {
            if (strd == 1 || fnd == 0) {
                my_pos_next = curc + fnd * strd;
                // = chain_ev[cs + my_pos_next]: the passing tally published the event itself (the minimum
                // over events is the minimum over slots: one chain) — no look-up behind `fnd`
                my_lo_next = (int)(fev >> 32);
                // the next round's window of this member starts here; its last candidate (the band range,
                // if the round is entered right away) came with the event when the published window reached
                // that far, else it is fetched here (end of the visible chain, strided windows)
                {   // (the window of the next round: offset by `skip` when that leaves a candidate)
                    const int w0 = my_pos_next + skip < clen ? my_pos_next + skip : my_pos_next;
                    spec_cur = w0;
                    const int dl = (int)(fev & 0x3ffffffull);
                    spec_last = (dl != 0x3ffffff && strd == 1) ? my_lo_next + dl : chain_ev[cs + (clen - w0 < K ? clen : w0 + K) - 1];
                }
                un = 0;
            } else {  // bracketed by a strided window: look at (slot fnd-1, slot fnd] next
                curc += (fnd - 1) * strd + 1;
                strd = 1;
            }
        } else if (jf != SW_INF) {
            if (strd > 1 && jf > 0) {  // a far slot inside a strided window: back to contiguous after the last false slot
                curc += (jf - 1) * strd + 1;
                strd = 1;
            } else {
                curc += jf * strd;  // the near slots before the first far one are false
                strd = 1;
                far_wait = 1;
                frc = 0;
            }
        } else if (strd == 1 && curc + K >= clen) {  // chain exhausted: no round-(r+1) event of c (yet)
            un = 0;
            evr_now = r;
            evp_now = clen;
        } else if (strd > 1 && curc + K * strd >= clen) {  // the tail of the chain: contiguous windows
            curc += (offered - 1) * strd + 1;
            strd = 1;
        } else {
            curc += (offered - 1) * strd + 1;  // = curc + K for a contiguous window
            if (miss < 255) ++miss;
            if (gallop_after > 0 && miss >= gallop_after) strd = K < 255 ? K : 255;
        }
    }
    // ---- far candidates (inheritance): every earlier position of c being false, the cursor
    // candidate e has round >= r+1 iff its other-parent q has, i.e. q >= lo[r+1][creator(q)].
    // That is known once creator(q) is resolved for this round; it is definitely false when q
    // lies before that creator's cursor; otherwise c waits for the next iteration.  When both
    // parents have round <= r the candidate needs a real tally: the band cap is doubled.
    int grow = 0;
    SW_STAMP(stamp && B.dbg_minor, dbg_it, sb + 2);
    if (iter > 0) {
        if (member) {
            // a member is "resolved for round r" unless it is still searching
            s_res[c] = !un;
            s_ln[c] = my_lo_next != SW_INF ? my_lo_next : lo_r1;
            s_cp[c] = curc < clen ? cs + curc : -1;
        }
        __syncthreads();
        if (far_wait) {
            const int e = chain_ev[cs + curc];
            const int q = op[e];
            const int b = cr[q];
            if (s_res[b]) {
                if (q >= s_ln[b]) { my_lo_next = e; my_pos_next = curc; un = 0; }
                else grow = 1;
            } else {
                const int lb = s_cp[b] >= 0 ? chain_ev[s_cp[b]] : SW_INF;
                if (q < lb) grow = 1;  // q precedes b's first possible round-(r+1) event
            }
            if (grow) {
                if (ncap >= MCAP) frc = 1;  // cap exhausted: tally it with on-the-fly hop masks
            }
        }
    }
    const int rl = c & 63, rw = c >> 6;
    constexpr int nwv = nthr >> 6;
    int nun = 0;
    {   // count(un) and any(grow) with one barrier
        // (round 4, measured and dropped — profiles/r04i_*: the next round's entry counts computed speculatively and reduced on
        // THIS barrier, and the inheritance pass skipped unless a tally flagged a FAR candidate: the resolve step went from
        // 3.44 to 3.13 us, the band phase behind it from 2.28 to 2.59 us, the iteration stayed at 18.1 us)
        const u64 bu = __ballot(un != 0), bg = __ballot(grow != 0);
        if (rl == 0) { s_red[0][0][rw] = __popcll(bu); s_red[0][1][rw] = bg != 0; }
        __syncthreads();
        int anyg = 0;
#pragma unroll
        for (int w = 0; w < nwv; ++w) { nun += s_red[0][0][w]; anyg |= s_red[0][1][w]; }
        if (anyg && ncap < MCAP) ncap = ncap * 2 < MCAP ? ncap * 2 : MCAP;
    }
    SW_STAMP(stamp && B.dbg_minor, dbg_it, sb + 3);
    int need_mask = 0, done = 0, err = 0, max_round = 0;
    if (nun == 0) {
        int lr, nx, start;
        if (iter > 0) {  // commit round r, then look at round r+1
            if (my_lo_next != SW_INF) {
                if (writer) {
                    lo[(unsigned)((r + 1) * npad + c)] = my_lo_next;
                    lopos[(unsigned)((r + 1) * npad + c)] = my_pos_next;
                    B.front[c] = r + 1;
                }
                lr = my_lo_next;
                start = my_pos_next;
            } else {
                lr = lo_r1;
                start = lopos_r1;
            }
            nx = lo_r2;
            r = r + 1;
        } else {
            lr = (member && r < Rcap) ? lo[(unsigned)(r * npad + c)] : SW_INF;
            start = (member && r < Rcap) ? lopos[(unsigned)(r * npad + c)] : 0;
            nx = lo_r1;
        }
        SW_STAMP(stamp && B.dbg_minor, dbg_it, 8);
        int lp = 1;  // s_red[0] was used by the count above
        for (;;) {  // enter the next round that has unresolved members
            if (r + 1 >= Rcap) { err = 1; done = 1; break; }
            const int act = lr != SW_INF;
            un = 0;
            skp = 0;
            if (act && nx == SW_INF) {
                bool resumed = false;
                if (evr_now == r && evp_now > start) { start = evp_now; resumed = true; }
                curc = start;
                un = start < clen;
                if (un && !resumed && skip > 0 && start + skip < clen) { curc = start + skip; skp = skip; }
            }
            // count(act), count(un), min(lr over the active members) with one barrier; the LDS
            // slots alternate between passes of this loop (a wave is at most one barrier ahead)
            int nact = 0, minlr = SW_INF;
            {
                const u64 ba = __ballot(act != 0), bu = __ballot(un != 0);
                const int wm = wave_min_i32(act ? lr : SW_INF);
                if (rl == 0) { s_red[lp][0][rw] = __popcll(ba); s_red[lp][1][rw] = __popcll(bu); s_red[lp][2][rw] = wm; }
                __syncthreads();
                nun = 0;
#pragma unroll
                for (int w = 0; w < nwv; ++w) {
                    nact += s_red[lp][0][w];
                    nun += s_red[lp][1][w];
                    const int m = s_red[lp][2][w];
                    minlr = m < minlr ? m : minlr;
                }
                lp ^= 1;
            }
            SW_STAMP(stamp && B.dbg_minor, dbg_it, 10);
            if (nact == 0) { done = 1; max_round = r - 1; break; }
            if (nun > 0) {
                SW_STAMP(stamp && B.dbg_minor, dbg_it, 11);
                mlo = minlr;
                thr = lr;
                my_lo_next = SW_INF;
                need_mask = 1;
                ncap = NEARCAP;
                frc = 0;
                strd = 1;
                miss = 0;
                break;
            }
            ++r;  // nothing to do in this round: step to the next one (rare, incremental calls)
            lr = nx;
            start = (member && r < Rcap) ? lopos[(unsigned)(r * npad + c)] : 0;
            nx = (member && r + 1 < Rcap) ? lo[(unsigned)((r + 1) * npad + c)] : SW_INF;
        }
    }
    if (done) un = 0;
    SW_STAMP(stamp && B.dbg_minor, dbg_it, sb + 4);
    // candidates of member c in the next tally launch: chain positions [curc, curc + K)
    const int live = !un ? 0 : strd == 1 ? (clen - curc < K ? clen - curc : K)
                                         : ((clen - 1 - curc) / strd + 1 < K ? (clen - 1 - curc) / strd + 1 : K);
    const int maxc = !live ? -1 : (curc == spec_cur && strd == 1 ? spec_last : chain_ev[cs + curc + (live - 1) * strd]);
    SW_STAMP(stamp && B.dbg_minor, dbg_it, 12);
    int s_max = -1, s_cnt = 0;
    {   // max(last candidate), sum(evaluated) and the thresholds for the band, one barrier
        // (the [.][3] slots are written only here, once per launch)
        const int wmx = wave_max_i32(live ? maxc : -1);
        const int wsm = wave_sum_i32(evaluated);
        if (rl == 0) { s_red[0][3][rw] = wmx; s_red[1][3][rw] = wsm; }
        if (member) {
            s_thr[c] = thr;
            s_cp[c] = live ? cs + curc : -1;  // (the three arrays are free again after the inheritance step)
            s_ce[c] = cs + clen;
            s_ln[c] = live;
            s_res[c] = strd;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < nwv; ++w) {
            const int m = s_red[0][3][w];
            s_max = m > s_max ? m : s_max;
            s_cnt += s_red[1][3][w];
        }
    }
    SW_STAMP(stamp && B.dbg_minor, dbg_it, 13);
    // Candidate table for the tally (saves it a dependent round trip): workgroup b publishes the
    // window of member b.  The load is issued here and the store deferred behind the band rows, so
    // that it costs this kernel no round trip of its own.
    // Entries 1 .. K are the candidates; a contiguous window is published up to 63 positions far (never tallied:
    // the tally of slot j hands entry j + skip + K — the last candidate of the window that would follow slot
    // j — on to the next resolve step together with its verdict).
    const int KPS = (NW <= 4 || K >= 32) ? 64 : 32;   // (wider member counts: no look-ahead — their iterations are throughput-bound)
    int cand_v = -1;
    const bool cand_mine = (int)blockIdx.x < npad && (int)threadIdx.x < KPS;
    {
        auto window = [&](int m) -> int {
            const int j = (int)threadIdx.x - 1;
            const int base = s_cp[m], lv = s_ln[m], st_ = s_res[m];
            if (base < 0 || j < 0) return -1;
            if (j < lv) return chain_ev[base + j * st_];
            return (st_ == 1 && base + j < s_ce[m]) ? chain_ev[base + j] : -1;   // look-ahead (visible events only)
        };
        if (cand_mine) cand_v = window(blockIdx.x);
        if ((int)threadIdx.x < KPS)  // fewer workgroups than members (tuning runs): the rest right away
            for (int m = blockIdx.x + gridDim.x; m < npad; m += gridDim.x)
                B.cand[((size_t)(1 - par) * npad + m) * 64 + threadIdx.x] = window(m);
    }
    auto flush_cand = [&]() {
        if (cand_mine) B.cand[((size_t)(1 - par) * npad + blockIdx.x) * 64 + threadIdx.x] = cand_v;
        if (B.dbg_blk && dbg_it < SW_DBG_MAX_ITERS) {   // (diagnostics: when this workgroup was done)
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (threadIdx.x == 0 && blockIdx.x < 2048) B.dbg_blk[((size_t)dbg_it * 2 + 0) * 2048 + blockIdx.x] = wall_clock64();
        }
    };
    // band = every event a candidate can have as a hop: [mlo, max candidate], capped at MCAP
    // (hops beyond the cap are rebuilt from their rows by the tally kernel)
    int mask_from = mlo;
    {
        int want = s_max + 1;
        if (want - mlo > ncap) want = mlo + ncap;
        if (want > N) want = N;
        if (need_mask) {
            mhi = want > mlo ? want : mlo;
        } else if (!done && want > mhi) {  // retry iteration reaching further: extend the table
            mask_from = mhi;
            mhi = want;
            need_mask = 1;
        }
    }
    SW_STAMP(stamp && B.dbg_minor, dbg_it, 14);
    if (writer) {
        if (member) {
            B.unres[out + c] = un;
            B.cur[out + c] = curc;
            B.lo_next[out + c] = my_lo_next;
            B.pos_next[out + c] = my_pos_next;
            B.evalround[out + c] = evr_now;
            B.evalpos[out + c] = evp_now;
            B.lo_r[out + c] = thr;
            B.found64[out + c] = ~0ull;
            B.farslot[out + c] = SW_INF;
            B.force[out + c] = frc;
            B.gallop[out + c] = strd | (miss << 8) | (skp << 16);
        }
        if (c == 0) {
            RState t = *si;
            t.r = r; t.done = done; t.need_mask = need_mask; t.mlo = mlo; t.mhi = mhi;
            t.mask_from = mask_from;
            t.ncap = ncap;
            t.iter = iter + 1; t.n_unres = nun;
            t.evals = si->evals + (u64)s_cnt;
            t.band_events = si->band_events + (u64)((need_mask && !done) ? mhi - mask_from : 0);
            if (done) t.max_round = max_round;
            if (err) t.err = 1;
            *so = t;
        }
    }
    // ---- band masks
    SW_STAMP(stamp, dbg_it, sb + 5);
    if (B.dbg && stamp && dbg_it < SW_DBG_MAX_ITERS) B.dbg[(size_t)dbg_it * 32 + sb + 7] = (u64)(need_mask ? mhi - mask_from : 0);
    if (done || !need_mask) { flush_cand(); return; }
    const int lane = lane_id();
    constexpr int wpb = nthr >> 6;
    // the writer block finishes later than the others (it publishes the state): it takes no
    // share of the band, so that the kernel ends with the band and not with its stores
    const int skipw = gridDim.x > 1 ? 1 : 0;
    if (skipw && writer) { flush_cand(); return; }
    // (split: the groups of band events are dealt to the parts — wave w of part p is wave w * parts + p of the whole)
    const int wave = SPLIT ? ((blockIdx.x - skipw) * wpb + (threadIdx.x >> 6)) * sd.parts + sd.part : (blockIdx.x - skipw) * wpb + (threadIdx.x >> 6);
    const int nwaves = SPLIT ? (gridDim.x - skipw) * wpb * sd.parts : (gridDim.x - skipw) * wpb;
    int t_[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) t_[j] = s_thr[j * 64 + lane];
    // a wave takes the groups of 8 consecutive events g = wave (mod nwaves), by ABSOLUTE event index, the rows of up to
    // RIF events in flight at once.  Measured and dropped in round 4 (profiles/r04e_*, r04f_*): the rows as ONE vector load
    // per lane (dwordx4: 8.10 against 7.63 ms per pass), and a 16-bit side table of the rows relative to the event index
    // (half the bytes: 8.34 ms with ushort loads, 8.77 ms with vector loads) — four dword loads per row it stays.
    // Groups of GE consecutive events: 8 up to 256 members; 4 beyond (round 5: a wave keeps the rows of 4 events in flight there, so a
    // group of 8 was two passes — and with ~1.1 groups per wave the kernel lasted as long as the waves that drew TWO groups:
    // profiles/r05o_loop_phases_1024.txt, the stamped workgroup's share done after 14 us, the kernel after 32)
    constexpr int GE = NW <= 4 ? 8 : 4;
    const int g_first = mask_from / GE;
    for (int g = g_first + ((wave - g_first) % nwaves + nwaves) % nwaves; g * GE < mhi; g += nwaves) {
        // (masks are built for every band event: testing "can it be a hop at all" first costs a dependent load, an unused mask the
        // traffic of its row.  Round 5, measured at 1024 members — profiles/r05o_ab_1024x2M.log: rows below their creator's
        // threshold skipped, 32.76 against 32.77 ms: uniform gossip has hardly any such event inside the band)
        const int base = g * GE;
        const int kk = base + (lane & (GE - 1));
        const bool mine = lane < GE && kk < mhi && kk >= mask_from;
        u64 vm = __ballot(mine);
        // FINALIZE FROM THE BAND (round 4): a band event of this run's sub-batch that lies at or after its
        // creator's round-r witness has round >= r, and its mask against lo[r] — the ballots below — is its sees-mask if its round
        // IS r.  The pass of the event's true round is the last one to write it (rounds only go up), so after the loop
        // round[e] / S[e] are final for every event a band of its own round covered; k_finalize_check finds the others.
        // (the creator is fetched with a CLAMPED index, unconditionally: a load inside an exec-masked block is waited for at the
        // block's end — one dependent round trip in front of the rows of every group)
        // The `asm` after the row loads pins the first USE of the creator there: the compiler would otherwise wait for it (and
        // for the LDS look-up behind it) before it issues the rows — two dependent round trips per group instead of one.
        int kc = kk < mask_from ? mask_from : (kk < mhi ? kk : mhi - 1);
        kc = kc < N ? kc : N - 1;   // (an empty band — mhi == mask_from — clamps to mhi: stay inside the events, the value is not used then)
        kc = kc < 0 ? 0 : kc;
        int crk = cr[kc];
        auto fin_mask = [&]() -> u64 {
            asm volatile("" : "+v"(crk));
            const int fin_thr = s_thr[crk];   // lo[r][creator]: INF when the creator has no round-r witness
            return __ballot(mine && kk >= fin_thr && kk >= fin_from);
        };
        constexpr int RIF = GE;  // rows in flight per wave = a whole group (one memory round trip per group)
        if constexpr (FAST && RIF == 8) {
            // a FULL group — eight consecutive events, all of them band events (the common case): fixed indices, ONE 64-bit
            // row base per group and compile-time offsets (npad = 64 NW) instead of the ffs / mask bookkeeping and a
            // multiply-add per load (ISA: ~400 instead of ~1 260 instructions per group at NW = 4)
            if (vm == 0xffull) {
                constexpr int NP = 64 * NW;
                const int* row0 = L + (size_t)base * NP + lane;
                int v[8][NW];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < NW; ++j) v[u][j] = row0[u * NP + j * 64];
                const u64 fin_m = fin_mask();
                u64* mb0 = Mb + (size_t)(base - mlo) * NW + lane;
                u64* s0 = S_out + (size_t)base * NW + lane;
                int* pc0 = Pc + (base - mlo);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    u64 word = 0;
                    int pcs = 0;   // popcount of the whole mask: the tally's cheap bounds (k_tally_bits, FILT) add these up
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const u64 bm = __ballot(v[u][j] >= t_[j]);
                        word = lane == j ? bm : word;
                        pcs += __popcll(bm);
                    }
                    if (lane < NW) mb0[u * NW] = word;
                    if (lane == NW + 1) pc0[u] = pcs;
                    if ((fin_m >> u) & 1ull) {
                        if (lane < NW) s0[u * NW] = word;
                        if (lane == NW) round_out[base + u] = r;
                    }
                }
                continue;
            }
        }
        while (vm) {
            int ks[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                ks[u] = -1;
                if (vm) { ks[u] = base + __ffsll((long long)vm) - 1; vm &= vm - 1; }
            }
            int v[RIF][NW];
#pragma unroll
            for (int u = 0; u < RIF; ++u)
#pragma unroll
                for (int j = 0; j < NW; ++j)
                    v[u][j] = ks[u] >= 0 ? L[(size_t)ks[u] * npad + j * 64 + lane] : -1;
                const u64 fin_m = fin_mask();   // (RIF covers a whole group: this loop makes one pass)
#pragma unroll
            for (int u = 0; u < RIF; ++u)
                if (ks[u] >= 0) {
                    u64 word = 0;   // lane j < NW keeps mask word j: the NW words of an event leave in ONE store instruction
                    int pcs = 0;
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const u64 bm = __ballot(v[u][j] >= t_[j]);
                        word = lane == j ? bm : word;
                        pcs += __popcll(bm);
                    }
                    if constexpr (SPLIT) {   // into the tables of every part
                        const bool fin = (fin_m >> (ks[u] - base)) & 1ull;
                        for (int q = 0; q < sd.ndst; ++q) {
                            if (lane < NW) sd.Mb[q][(size_t)(ks[u] - mlo) * NW + lane] = word;
                            if (lane == NW + 1) sd.Pc[q][ks[u] - mlo] = pcs;
                            if (fin) {
                                if (lane < NW) sd.S[q][(size_t)ks[u] * NW + lane] = word;
                                if (lane == NW) sd.round[q][ks[u]] = r;
                            }
                        }
                    } else {
                    if (lane < NW) Mb[(size_t)(ks[u] - mlo) * NW + lane] = word;
                    if (lane == NW + 1) Pc[ks[u] - mlo] = pcs;
                    if ((fin_m >> (ks[u] - base)) & 1ull) {
                        if (lane < NW) S_out[(size_t)ks[u] * NW + lane] = word;
                        if (lane == NW) round_out[ks[u]] = r;
                    }
                    }
                }
        }
    }
    flush_cand();
    SW_STAMP(stamp, dbg_it, sb + 6);
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
k_tally_candidates(LoopBufs B, int par, int K,
                   const int* __restrict__ chain_start, const int* __restrict__ chain_len,
                   const int* __restrict__ chain_ev,
                   const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ sp,
                   const int* __restrict__ op, const u64* __restrict__ Mb,
                   const uint32_t* __restrict__ stake, uint32_t tot2, int npad) {
    __shared__ u64 s_hm[4][NW * 64];
    __shared__ u64 s_key[4];
    __shared__ int s_fark[4], s_cm[4];
    RState* st = B.st + (1 - par);  // written by k_resolve_band of this iteration
    const size_t pb = (size_t)(1 - par) * npad;
    const int* unres = B.unres + pb;
    const int* lo_r = B.lo_r + pb;
    if (st->done) return;   // (uniform over the grid)
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wib;
    const int cm = w / K, cj = w - cm * K;  // member, candidate slot
    // verdicts reduced per workgroup before the atomics, as in k_tally_bits
    u64 key = ~0ull;
    int fark = SW_INF;
    u64 nfar = 0;
    const int e = unres[cm] ? B.cand[((size_t)(1 - par) * npad + cm) * 64 + cj + 1] : -1;  // slot cj of the member's window
    if (e >= 0) do {
    const int mlo = st->mlo, mhi = st->mhi;
    u64* hm = s_hm[wib];
    const int ce = cr[e], spe = sp[e];
    {   // FAR candidate (a parent beyond the band): decided by inheritance in k_resolve_band
        const int ope = op[e];
        if ((spe > ope ? spe : ope) >= mhi && !(cj == 0 && B.force[pb + cm])) { fark = cj; break; }
    }
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
    if (3u * cnt > tot2)  // count of members vs the STAKE threshold (Q2)
        key = ((u64)(uint32_t)e << 32) | ((u64)cj << 26) | 0x3ffffffull;
    } while (0);
    if (lane == 0) { s_key[wib] = key; s_fark[wib] = fark; s_cm[wib] = cm; }
    __syncthreads();
    if (lane == 0) {
        if (wib == 0 || s_cm[wib - 1] != cm) {   // the first wave of a member in this workgroup speaks for the member's waves
            for (int w2 = wib + 1; w2 < 4 && s_cm[w2] == cm; ++w2) {
                key = s_key[w2] < key ? s_key[w2] : key;
                fark = s_fark[w2] < fark ? s_fark[w2] : fark;
            }
            if (key != ~0ull) atomicMin(reinterpret_cast<unsigned long long*>(&B.found64[pb + cm]), key);
            if (fark != SW_INF) atomicMin(&B.farslot[pb + cm], fark);
        }
        if (nfar) atomicAdd(&st->far_hops, nfar);
    }
}


// ---------------------------------------------------------------------------------
// Unit-stake fast path of the same tally: bit-sliced ("vertical") counters.
// A hop mask is W32 = 2*NW 32-bit words; lane l = (group g = l / W32, word w = l % W32)
// accumulates word w of the masks of the hops h == g (mod G), G = 64 / W32, into bit-plane
// counters with carry-save adders (one 32-bit logic op handles 32 members at once), the G
// groups are then added with bit-sliced full adders across lanes, and "hits > 2T/3" is a
// bit-sliced comparison whose popcount is the number of strongly-seen members.
// W32 consecutive lanes read one 8*NW-byte mask: coalesced gathers from the L2-resident
// band table.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void csa(uint32_t& hi, uint32_t& lo_, const uint32_t a, const uint32_t b, const uint32_t c) {
    const uint32_t u = a ^ b;
    hi = (a & b) | (u & c);
    lo_ = u ^ c;
}

template <int PLT>
__device__ __forceinline__ void ripple_add(uint32_t (&b)[PLT], uint32_t x, const int from) {
#pragma unroll
    for (int p = 0; p < PLT; ++p) {
        if (p >= from) {
            const uint32_t t = b[p] & x;
            b[p] ^= x;
            x = t;
        }
    }
}

// adds eight 1-bit-per-member words (Harley-Seal block)
template <int PLT>
__device__ __forceinline__ void add8(uint32_t (&b)[PLT], const uint32_t (&x)[8]) {
    static_assert(PLT >= 4, "add8 needs at least 4 planes");
    uint32_t twoA, twoB, twoC, twoD, fourA, fourB, eight;
    csa(twoA, b[0], b[0], x[0], x[1]);
    csa(twoB, b[0], b[0], x[2], x[3]);
    csa(fourA, b[1], b[1], twoA, twoB);
    csa(twoC, b[0], b[0], x[4], x[5]);
    csa(twoD, b[0], b[0], x[6], x[7]);
    csa(fourB, b[1], b[1], twoC, twoD);
    csa(eight, b[2], b[2], fourA, fourB);
    ripple_add<PLT>(b, eight, 3);
}

constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }


// Bit-sliced tally, part 1: lane (g, w) adds word w of the masks of its hops (hop h belongs to
// group h % G; pk[h] = row of hop h in `table32`, -1 = not a hop) into bit-plane counters.
template <int NW>
__device__ __forceinline__ void bits_accumulate(const int* pk, const uint32_t* __restrict__ table32,
                                                uint32_t (&b)[ilog2_c(64 * NW) + 1], const int lane) {
    constexpr int W32 = 2 * NW, G = 64 / W32, HPL = (64 * NW) / G, PLT = ilog2_c(64 * NW) + 1;
    const int w = lane % W32, g = lane / W32;
#pragma unroll
    for (int p = 0; p < PLT; ++p) b[p] = 0;
    if constexpr (HPL >= 8) {
#pragma unroll 4
        for (int i0 = 0; i0 < HPL; i0 += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = pk[g + G * (i0 + u)];
                x[u] = kk >= 0 ? table32[(size_t)kk * W32 + w] : 0u;
            }
            add8<PLT>(b, x);
        }
    } else {
#pragma unroll
        for (int i = 0; i < HPL; ++i) {
            const int kk = pk[g + G * i];
            const uint32_t x = kk >= 0 ? table32[(size_t)kk * W32 + w] : 0u;
            ripple_add<PLT>(b, x, 0);
        }
    }
}

// The same for the round-loop tally, with the dependent chain taken out of the gathers: `pk` holds
// table rows (row 0 = all-zero = "not a hop": no branch around the load), laid out so that the
// hops of lane group g are contiguous (pk[g * PKS + i] = hop g + G*i; wide LDS reads, issued ahead
// of the loads), and the offsets are 32-bit (scalar base + vector offset addressing).
// Only the planes a per-lane count can reach (<= HPL) are touched.
template <int PLT, int TOP>
__device__ __forceinline__ void ripple_add_to(uint32_t (&b)[PLT], uint32_t x, const int from) {
#pragma unroll
    for (int p = 0; p < TOP; ++p) {
        if (p >= from) {
            const uint32_t t = b[p] & x;
            b[p] ^= x;
            x = t;
        }
    }
}

template <int NW>
struct BitsGeom {
    static constexpr int W32 = 2 * NW, G = 64 / W32, HPL = (64 * NW) / G, PLT = ilog2_c(64 * NW) + 1;
    static constexpr int PKS = HPL + 4;              // padded row of pk (keeps 16-byte alignment)
    static constexpr int PA = ilog2_c(HPL) + 1;      // planes of a per-lane count
    static constexpr int PK_INTS = G * PKS;
    __device__ static __forceinline__ int slot(int h) { return (h % G) * PKS + h / G; }
};

template <int NW>
__device__ __forceinline__ void bits_accumulate_z(const int* pk, const uint32_t* __restrict__ table32,
                                                  uint32_t (&b)[ilog2_c(64 * NW) + 1], const int lane) {
    using Gm = BitsGeom<NW>;
    constexpr int W32 = Gm::W32, HPL = Gm::HPL, PLT = Gm::PLT, PA = Gm::PA;
    const uint32_t w = lane % W32;
    const int g = lane / W32;
    const int* row = pk + g * Gm::PKS;
    // uniform base + 32-bit byte offset (the table is at most (MCAP + 1) * 8 * NW bytes << 4 GB)
    const char* tb = reinterpret_cast<const char*>(table32);
    const uint32_t wb = w * 4u;
    auto ld = [&](int k) -> uint32_t {
        return *reinterpret_cast<const uint32_t*>(tb + ((uint32_t)k * (uint32_t)(W32 * 4) + wb));
    };
#pragma unroll
    for (int p = 0; p < PLT; ++p) b[p] = 0;
    if constexpr (HPL >= 8) {
#pragma unroll 4
        for (int i0 = 0; i0 < HPL; i0 += 8) {
            const int4 k0 = *reinterpret_cast<const int4*>(row + i0);
            const int4 k1 = *reinterpret_cast<const int4*>(row + i0 + 4);
            uint32_t x[8];
            x[0] = ld(k0.x); x[1] = ld(k0.y); x[2] = ld(k0.z); x[3] = ld(k0.w);
            x[4] = ld(k1.x); x[5] = ld(k1.y); x[6] = ld(k1.z); x[7] = ld(k1.w);
            uint32_t twoA, twoB, twoC, twoD, fourA, fourB, eight;
            csa(twoA, b[0], b[0], x[0], x[1]);
            csa(twoB, b[0], b[0], x[2], x[3]);
            csa(fourA, b[1], b[1], twoA, twoB);
            csa(twoC, b[0], b[0], x[4], x[5]);
            csa(twoD, b[0], b[0], x[6], x[7]);
            csa(fourB, b[1], b[1], twoC, twoD);
            csa(eight, b[2], b[2], fourA, fourB);
            ripple_add_to<PLT, PA>(b, eight, 3);
        }
    } else {
#pragma unroll
        for (int i = 0; i < HPL; ++i) {
            const uint32_t x = ld(row[i]);
            ripple_add_to<PLT, PA>(b, x, 0);
        }
    }
}

// cross-lane part for bits_accumulate_z: level k adds two counts of PA + k planes
template <int NW, int OFF, int TOP>
__device__ __forceinline__ void bits_level_z(uint32_t (&b)[ilog2_c(64 * NW) + 1]) {
    constexpr int PLT = ilog2_c(64 * NW) + 1;
    if constexpr (OFF < 64) {
        uint32_t carry = 0;
#pragma unroll
        for (int p = 0; p < PLT; ++p) {
            if (p < TOP) {
                uint32_t x, y;
                lanes_pair<OFF>(b[p], x, y);
                const uint32_t u = x ^ y;
                const uint32_t nc = (x & y) | (u & carry);
                b[p] = u ^ carry;
                carry = nc;
            }
        }
        if constexpr (TOP < PLT) b[TOP] = carry;
        bits_level_z<NW, OFF * 2, TOP + 1>(b);
    }
}

template <int NW>
__device__ __forceinline__ uint32_t bits_finish_z(uint32_t (&b)[ilog2_c(64 * NW) + 1], const uint32_t t23) {
    using Gm = BitsGeom<NW>;
    constexpr int W32 = Gm::W32, PLT = Gm::PLT;
    bits_level_z<NW, W32, Gm::PA>(b);
    uint32_t gt = 0, eq = 0xffffffffu;  // most significant plane first
#pragma unroll
    for (int p = PLT - 1; p >= 0; --p) {
        const uint32_t tb = ((t23 >> p) & 1u) ? 0xffffffffu : 0u;
        gt |= eq & b[p] & ~tb;
        eq &= ~(b[p] ^ tb);
    }
    if ((t23 >> PLT) != 0) gt = 0;  // threshold beyond any possible count
    return gt;
}

// part 2: add the G hop groups across lanes (bit-sliced full adders) and compare every
// member's count with t23 = floor(2T/3); returns, in every lane (g, w), the 32-member word w
// of "hits > 2T/3".
template <int NW, int OFF>
__device__ __forceinline__ void bits_level(uint32_t (&b)[ilog2_c(64 * NW) + 1]) {
    constexpr int PLT = ilog2_c(64 * NW) + 1;
    if constexpr (OFF < 64) {
        uint32_t carry = 0;
#pragma unroll
        for (int p = 0; p < PLT; ++p) {
            uint32_t x, y;
            lanes_pair<OFF>(b[p], x, y);
            const uint32_t u = x ^ y;
            const uint32_t nc = (x & y) | (u & carry);
            b[p] = u ^ carry;
            carry = nc;
        }
        bits_level<NW, OFF * 2>(b);
    }
}

template <int NW>
__device__ __forceinline__ uint32_t bits_finish(uint32_t (&b)[ilog2_c(64 * NW) + 1], const uint32_t t23) {
    constexpr int W32 = 2 * NW, PLT = ilog2_c(64 * NW) + 1;
    bits_level<NW, W32>(b);
    uint32_t gt = 0, eq = 0xffffffffu;  // most significant plane first
#pragma unroll
    for (int p = PLT - 1; p >= 0; --p) {
        const uint32_t tb = ((t23 >> p) & 1u) ? 0xffffffffu : 0u;
        gt |= eq & b[p] & ~tb;
        eq &= ~(b[p] ^ tb);
    }
    if ((t23 >> PLT) != 0) gt = 0;  // threshold beyond any possible count
    return gt;
}

// (round 5, measured and dropped — profiles/r05a_knobs_1024x2M.log: at 1024 members the kernel takes 99 VGPRs = 4 waves per SIMD; asking
// the allocator for 5 waves (96 VGPRs, 8 B of scratch) changes nothing, for 6 (80 VGPRs, 18 spilled dwords) costs 6 %)
// FILT (round 5): before the n masks of a candidate are gathered, the POPCOUNTS of those masks (Pc[], left by the band pass: 4 bytes
// per hop instead of n / 8) bound the verdict from both sides.  With V valid hops, t = floor(2T/3) and S = sum of the hops'
// popcounts = sum over the columns of hits[c_]:  a passing tally has more than t columns with hits > t, hence S >= (t + 1)^2 —
// below that the candidate FAILS without a gather; a failing one has at most t columns above t, each at most V, the others at
// most t, hence S <= t V + (n - t) t — above that it PASSES without a gather.  On uniform gossip the two bounds leave ~4 of a
// member's slots for the exact count (tests/model_bulk.py `popcount_bounds`); every verdict is still exact.
template <int NW, bool FILT, bool SPLIT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW <= 4 ? 8 : 2, 8)))
k_tally_bits(LoopBufs B, int par, int K, int skip, int mb_prefetch,
             const int* __restrict__ chain_start, const int* __restrict__ chain_len,
             const int* __restrict__ chain_ev,
             const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ sp,
             const int* __restrict__ op, const uint32_t* __restrict__ Mb32, uint32_t tot2, int npad, const int* __restrict__ Pc, const SplitDst* __restrict__ sdp) {
    const SplitDst& sd = *(SPLIT ? sdp : reinterpret_cast<const SplitDst*>(B.st));   // (only the split form looks at it)
    constexpr int W32 = 2 * NW;          // 32-bit words per mask
    constexpr int G = 64 / W32;          // hop groups
    constexpr int PLT = ilog2_c(64 * NW) + 1;  // planes for counts up to npad
    using Gm = BitsGeom<NW>;
    __shared__ __attribute__((aligned(16))) int s_pk[4][Gm::PK_INTS];
    __shared__ u64 s_key[4];
    __shared__ int s_fark[4], s_cm[4];
    __builtin_amdgcn_s_setprio(3);  // critical path: win issue arbitration against the can_see sweep
    pin_arg(B.st); pin_arg(B.lo_r); pin_arg(B.cur); pin_arg(B.unres); pin_arg(B.farslot);
    pin_arg(B.force); pin_arg(B.dbg); pin_arg(B.cand); pin_arg(B.found64); pin_arg(B.gallop); pin_arg(par); pin_arg(K); pin_arg(skip); pin_arg(mb_prefetch);
    pin_arg(L); pin_arg(sp); pin_arg(op); pin_arg(Mb32); pin_arg(tot2); pin_arg(npad); pin_arg((int)gridDim.x);
    RState* st = B.st + (1 - par);  // written by k_resolve_band of this iteration
    const size_t pb = (size_t)(1 - par) * npad;
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int wv = blockIdx.x * 4 + wib;
    // member, candidate slot (split: this part tallies the members part, part + parts, ...; a wave beyond the last of them
    // takes the path of a member that is not searching)
    const int cm_raw = SPLIT ? sd.part + sd.parts * (wv / K) : wv / K;
    const bool cm_ok = !SPLIT || cm_raw < npad;
    const int cm = cm_ok ? cm_raw : npad - 1, cj = wv - (wv / K) * K;
    // The verdict of this wave: `key` (its tally passed) or `fark` (a FAR candidate).  Verdicts go to the
    // member's words by atomicMin — ~7000 waves on ~24 cache lines serialise at the memory side (3-5 us
    // of kernel tail, measured with the phase stamps); the four waves of a workgroup are (mostly) four
    // consecutive slots of ONE member, so they reduce in LDS first and one lane speaks for them.
    u64 key = ~0ull;
    int fark = SW_INF;
    u64 nfar = 0;
    // Round trip 1: everything addressed by the launch parameters alone, issued before the first
    // branch — including the candidate, from the table k_resolve_band published (creator(e) = cm
    // by construction).  Round trip 2: its can_see row and both parents.  Round trip 3: the
    // gathered hop masks.
    const int s_done = st->done, mlo = st->mlo, mhi = st->mhi;
    // The band-mask table was written by the other XCDs' workgroups; the first waves of every XCD miss on it in
    // their gathers (2.9 us against 0.9 us for late waves, phase stamps).  The first 64 waves of each XCD touch
    // one 128-byte line per lane while they wait for their own first round trips (SW_TALLY_PF=0 switches it
    // off): 7.76 -> 7.69 ms per pass at 256 members / 1 M events.
    // (the member's words are requested BEFORE the touch below: see k_tally_tree)
    const int un = cm_ok ? B.unres[pb + cm] : 0, frc = B.force[pb + cm];
    const int* cand = B.cand + ((size_t)(1 - par) * npad + cm) * 64;
    const int e = cand[cj + 1];  // published by k_resolve_band (-1: no such candidate)
    // what the next resolve step will want if this slot is the member's first passing one: the last candidate of
    // the window that would follow it (entry j + skip + K of a contiguous window's look-ahead; -1: not published)
    const int la_i = cj + skip + K;
    const bool use_la = NW <= 4 || K >= 32;   // (the table is published 63 positions far only then, see k_resolve_band)
    const int la_ld = cand[(use_la && la_i < 64) ? la_i : 0];
    const int la = (use_la && la_i < 64) ? la_ld : -1;
    const int gsv = B.gallop[pb + cm];
    int thr[NW], P[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) thr[j] = B.lo_r[pb + j * 64 + lane];
    int pf_dummy = 0;
    if (mb_prefetch) {
        const int wx = ((int)blockIdx.x >> 3) * 4 + wib;               // wave index within its XCD (block b runs on XCD b mod 8: locality only)
        const int nlines = ((mhi - mlo + 1) * NW * 8 + 127) >> 7;
        const int line = wx * 64 + lane;
        if (wx < 64 && line < nlines) {
            const char* q = reinterpret_cast<const char*>(Mb32) + (size_t)line * 128;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf_dummy) : "v"(q) : "memory");
        }
    }
    // stamped waves: candidate slot K/2 of the first and of the last member
    const bool stamp = B.dbg && lane == 0 && cj == (K >> 1) && (cm == 0 || cm == (int)(gridDim.x * 4 / K) - 1);
    const int sb = cm == 0 ? 16 : 24;
    const int it_ = stamp ? st->iter - 1 + st->pad_ : 0;
    if (stamp && !s_done && it_ < SW_DBG_MAX_ITERS) B.dbg[(size_t)it_ * 32 + sb] = wall_clock64();
    SW_STAMP(stamp && !s_done, it_, sb + 1);
    if (s_done) return;   // (uniform over the grid: no wave of this workgroup reaches the barrier below)
    if (un && e >= 0) do {
    const int ce = cm;
    int* pk = s_pk[wib];
    SW_STAMP(stamp, it_, sb + 2);
    const int ope = op[e], spe = sp[e];
#pragma unroll
    for (int j = 0; j < NW; ++j) P[j] = L[(size_t)e * npad + j * 64 + lane];
    SW_STAMP(stamp, it_, sb + 3);
    // FAR candidate (a parent beyond the band): decided by inheritance in k_resolve_band
    if ((spe > ope ? spe : ope) >= mhi && !(cj == 0 && frc)) { fark = cj; break; }
    u64 farm[NW];
    uint32_t nvalid = 0;
    int prow[NW];   // this lane's hops as rows of the band tables (0: not a hop)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        int v = P[j];
        if (j * 64 + lane == ce) v = spe;  // the row BEFORE the self overwrite (Q4)
        P[j] = v;
        const bool valid = v >= thr[j];
        const bool inband = valid && v < mhi;
        prow[j] = inband ? v - mlo + 1 : 0;  // row 0 of the table is all-zero
        pk[Gm::slot(j * 64 + lane)] = prow[j];
        farm[j] = __ballot(valid && !inband);
        nfar += __popcll(farm[j]);
        nvalid += __popcll(__ballot(valid));
    }
    // necessary condition: a member is strongly seen only through more than 2T/3 (unit-stake)
    // hops, so with fewer valid hops no column can pass — skip the gathers altogether
    if (3u * nvalid <= tot2) break;
    bool sure = false;
    if constexpr (FILT) {
        if (!nfar) {   // (a hop beyond the band has no popcount in the table: the exact count decides)
            int sum = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) sum += Pc[prow[j]];
            const uint32_t S = (uint32_t)wave_sum_i32(sum);
            const uint32_t t = tot2 / 3u, nmem = tot2 >> 1;
            if (S < (t + 1u) * (t + 1u)) break;                   // fewer hits than t + 1 columns above t need
            sure = S > t * nvalid + (nmem - t) * t;               // more hits than t columns at V and the rest at t hold
        }
    }
    const int w = lane % W32, g = lane / W32;
    uint32_t cnt = tot2;   // (a sure pass: any count above 2T/3)
    if (!sure) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t b[PLT];
    bits_accumulate_z<NW>(pk, Mb32, b, lane);
    SW_STAMP(stamp, it_, sb + 4);
    if (nfar) {  // rare: hops outside the band, masks built from their rows on the fly
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u64 far = farm[j];
            while (far) {
                const int h = __ffsll((long long)far) - 1;
                far &= far - 1;
                const int kf = __shfl(P[j], h);
                uint32_t x = 0;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj) {
                    const int v = L[(size_t)kf * npad + jj * 64 + lane];
                    const u64 bal = __ballot(v >= thr[jj]);
                    if ((w >> 1) == jj) x = (uint32_t)(bal >> (32 * (w & 1)));
                }
                if (g == ((j * 64 + h) % G)) ripple_add<PLT>(b, x, 0);
            }
        }
    }
    const uint32_t gt = bits_finish_z<NW>(b, tot2 / 3u);
    cnt = (uint32_t)wave_sum_i32((g == 0) ? __popc(gt) : 0);
    }
    if (3u * cnt > tot2) {  // count of members vs the STAKE threshold (Q2)
        const int dl = ((gsv & 0xff) == 1 && la >= 0 && la - e < 0x3ffffff) ? la - e : 0x3ffffff;
        key = ((u64)(uint32_t)e << 32) | ((u64)cj << 26) | (u64)dl;
    }
    } while (0);
    if (lane == 0) { s_key[wib] = key; s_fark[wib] = fark; s_cm[wib] = cm; }
    __syncthreads();
    if (lane == 0) {
        if (wib == 0 || s_cm[wib - 1] != cm) {   // the first wave of a member in this workgroup speaks for the member's waves
            for (int w2 = wib + 1; w2 < 4 && s_cm[w2] == cm; ++w2) {
                key = s_key[w2] < key ? s_key[w2] : key;
                fark = s_fark[w2] < fark ? s_fark[w2] : fark;
            }
            if constexpr (SPLIT) {   // into the verdict words of every part
                for (int q = 0; q < sd.ndst; ++q) {   // (system scope: the words of the other parts live on other GPUs)
                    if (key != ~0ull) __hip_atomic_fetch_min(reinterpret_cast<unsigned long long*>(&sd.found64[q][pb + cm]), (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (fark != SW_INF) __hip_atomic_fetch_min(&sd.farslot[q][pb + cm], fark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else {
            if (key != ~0ull) atomicMin(reinterpret_cast<unsigned long long*>(&B.found64[pb + cm]), key);
            if (fark != SW_INF) atomicMin(&B.farslot[pb + cm], fark);
            }
        }
        if (nfar) atomicAdd(&st->far_hops, nfar);
    }
    asm volatile("" ::"v"(pf_dummy));   // (the touch's destination register stays reserved to the end)
    SW_STAMP(stamp, it_, sb + 5);
    if (B.dbg_blk) {   // (diagnostics: when this workgroup was done)
        const int itb = st->iter - 1 + st->pad_;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < 2048 && itb >= 0 && itb < SW_DBG_MAX_ITERS) B.dbg_blk[((size_t)itb * 2 + 1) * 2048 + blockIdx.x] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------------
// Round loop, step 3, TREE form (SW_TALLY_IMPL=2): ONE workgroup of 8 waves per member searches the member's
// window of K candidates in two dependent levels instead of evaluating every slot.
// SS_r is monotone along a chain, so a FAILING slot rules out every slot before it:
//   level 1   wave w evaluates slot (w + 1) s - 1, s = ceil(K / 8) (the last one clipped to the window);
//   level 2   the first level-1 slot q that did not fail brackets the answer in (q - s, q]: the <= s - 1 slots in
//             between are evaluated by as many waves; the first of them that does not fail — else q — is the
//             member's verdict (PASS: its first event of round r + 1 in this window; FAR: decided by inheritance
//             in k_resolve_band), and every slot before it is known to be false.
// Against k_tally_bits (one wave per slot, K = 28: 7168 waves): 8 + 3 evaluations per member instead of 28, 2 waves per
// SIMD instead of 7 (room for the concurrent sweep, 2.5x less L2 gather traffic), a 256-workgroup launch, and
// the verdict is ONE plain store per member — no atomics at all (they serialise at ~12 ns each per cache line:
// the tail of k_tally_bits).  The price is a second dependent evaluation on the critical path.
// The verdict words mean the same to k_resolve_band: slots before the published one are false.  A forced cursor
// candidate (band cap exhausted) is evaluated first, on its own.
// ---------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ int tree_eval(const int e, const bool forced, const int ce, const int mlo, const int mhi, const int (&thr)[NW],
                                         int* pk, const int* __restrict__ L, const int* __restrict__ sp, const int* __restrict__ op,
                                         const uint32_t* __restrict__ Mb32, const uint32_t tot2, const int npad, const int lane, u64& nfar_acc) {
    using Gm = BitsGeom<NW>;
    constexpr int W32 = 2 * NW, G = 64 / W32, PLT = ilog2_c(64 * NW) + 1;
    const int ope = op[e], spe = sp[e];
    int P[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) P[j] = L[(size_t)e * npad + j * 64 + lane];
    if ((spe > ope ? spe : ope) >= mhi && !forced) return 2;   // FAR: a parent beyond the band
    u64 farm[NW];
    u64 nfar = 0;
    uint32_t nvalid = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (a second evaluation of this wave reuses its hop list)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        int v = P[j];
        if (j * 64 + lane == ce) v = spe;  // the row BEFORE the self overwrite (Q4)
        P[j] = v;
        const bool valid = v >= thr[j];
        const bool inband = valid && v < mhi;
        pk[Gm::slot(j * 64 + lane)] = inband ? v - mlo + 1 : 0;  // row 0 of the table is all-zero
        farm[j] = __ballot(valid && !inband);
        nfar += __popcll(farm[j]);
        nvalid += __popcll(__ballot(valid));
    }
    if (3u * nvalid <= tot2) return 0;   // fewer valid hops than any column needs
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int w = lane % W32, g = lane / W32;
    uint32_t b[PLT];
    bits_accumulate_z<NW>(pk, Mb32, b, lane);
    if (nfar) {  // rare: hops outside the band, masks built from their rows on the fly
        nfar_acc += nfar;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u64 far = farm[j];
            while (far) {
                const int h = __ffsll((long long)far) - 1;
                far &= far - 1;
                const int kf = __shfl(P[j], h);
                uint32_t x = 0;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj) {
                    const int v = L[(size_t)kf * npad + jj * 64 + lane];
                    const u64 bal = __ballot(v >= thr[jj]);
                    if ((w >> 1) == jj) x = (uint32_t)(bal >> (32 * (w & 1)));
                }
                if (g == ((j * 64 + h) % G)) ripple_add<PLT>(b, x, 0);
            }
        }
    }
    const uint32_t gt = bits_finish_z<NW>(b, tot2 / 3u);
    const uint32_t cnt = (uint32_t)wave_sum_i32((g == 0) ? __popc(gt) : 0);
    return 3u * cnt > tot2 ? 1 : 0;   // count of members vs the STAKE threshold (Q2)
}

template <int NW>
__global__ void __launch_bounds__(512)
k_tally_tree(LoopBufs B, int par, int K, int skip, int mb_prefetch,
             const int* __restrict__ L, const int* __restrict__ sp, const int* __restrict__ op,
             const uint32_t* __restrict__ Mb32, uint32_t tot2, int npad) {
    using Gm = BitsGeom<NW>;
    __shared__ __attribute__((aligned(16))) int s_pk[8][Gm::PK_INTS];
    __shared__ __attribute__((aligned(16))) int s_k1[8];
    __shared__ __attribute__((aligned(16))) int s_k2[8];
    __builtin_amdgcn_s_setprio(3);  // critical path: win issue arbitration against the can_see sweep
    pin_arg(B.st); pin_arg(B.lo_r); pin_arg(B.unres); pin_arg(B.farslot); pin_arg(B.force); pin_arg(B.dbg); pin_arg(B.cand);
    pin_arg(B.found64); pin_arg(B.gallop); pin_arg(B.treecnt); pin_arg(par); pin_arg(K); pin_arg(skip); pin_arg(mb_prefetch);
    pin_arg(L); pin_arg(sp); pin_arg(op); pin_arg(Mb32); pin_arg(tot2); pin_arg(npad);
    RState* st = B.st + (1 - par);  // written by k_resolve_band of this iteration
    const size_t pb = (size_t)(1 - par) * npad;
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int cm = blockIdx.x;
    const int s_done = st->done, mlo = st->mlo, mhi = st->mhi;
    // (the member's words are requested BEFORE the touch below: the touch sits in an exec-masked block that needs the band range,
    // i.e. a wait for the loop state — behind it these loads were a second dependent round trip at the head of the kernel)
    const int un = B.unres[pb + cm], frc = B.force[pb + cm], gsv = B.gallop[pb + cm];
    const int ce_lane = B.cand[((size_t)(1 - par) * npad + cm) * 64 + lane];   // entry 1 + j = the event of slot j
    int thr[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) thr[j] = B.lo_r[pb + j * 64 + lane];
    int pf_dummy = 0;
    if (mb_prefetch) {   // the first 64 waves of every XCD touch the band-mask table (written by the other XCDs), see k_tally_bits
        const int wx = ((int)blockIdx.x >> 3) * 8 + wib;
        const int nlines = ((mhi - mlo + 1) * NW * 8 + 127) >> 7;
        const int line = wx * 64 + lane;
        if (wx < 64 && line < nlines) {
            const char* q = reinterpret_cast<const char*>(Mb32) + (size_t)line * 128;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf_dummy) : "v"(q) : "memory");
        }
    }
    const bool stamp = B.dbg && lane == 0 && wib == 0 && (cm == 0 || cm == (int)gridDim.x - 1);
    const int sb = cm == 0 ? 16 : 24;
    const int it_ = stamp ? st->iter - 1 + st->pad_ : 0;
    if (stamp && !s_done && it_ < SW_DBG_MAX_ITERS) B.dbg[(size_t)it_ * 32 + sb] = wall_clock64();
    SW_STAMP(stamp && !s_done, it_, sb + 1);
    if (s_done || !un) return;   // (uniform over the workgroup: nobody reaches a barrier)
    const int live = __popcll(__ballot(lane >= 1 && lane <= K && ce_lane >= 0));   // candidates are contiguous from slot 0
    if (live == 0) return;
    // (every slot index below is uniform over the wave: v_readlane instead of a ds_bpermute round trip in front of the row address)
    auto slot_ev = [&](int j) -> int { return __builtin_amdgcn_readlane(ce_lane, __builtin_amdgcn_readfirstlane(j + 1)); };
    int* pk = s_pk[wib];
    u64 nfar = 0;
    int evals = 0;
    int p = -1, kind = 0;   // the verdict: slot and what it is (1 PASS, 2 FAR); 0: every candidate of the window is false
    if (frc) {   // rare: the cursor candidate is far and the band cannot grow — it is tallied with on-the-fly masks
        int k0 = 0;
        if (wib == 0) k0 = tree_eval<NW>(slot_ev(0), true, cm, mlo, mhi, thr, pk, L, sp, op, Mb32, tot2, npad, lane, nfar);
        if (threadIdx.x == 0) s_k2[0] = k0;
        __syncthreads();
        if (s_k2[0] == 1) { p = 0; kind = 1; }
        evals += 1;
        __syncthreads();
    }
    SW_STAMP(stamp, it_, sb + 2);
    if (kind == 0) {
        const int s = (K + 7) >> 3;
        const int nprobe = (live + s - 1) / s;   // <= 8
        int k1 = 0;
        // (a forced cursor candidate that failed above is known: evaluating slot 0 again, unforced, would call it FAR)
        // (round 4, measured and dropped — profiles/r04m_*: the rows of a wave's bracket loaded with its probe's row and their hop
        // lists staged in LDS before the barrier, so that level 2 starts at the gathers: level 2 went from 2.4 to 2.0 us, level 1
        // from 2.4 to 3.7 us)
        // (round 5, measured and dropped — profiles/r05c_ab_256x1M.log: every kernel starts with a cold L2, so the rows of level 2 are a
        // second trip to memory behind the barrier; each probing wave requesting the rows of its own bracket behind its probe's row
        // (values unused, 4 x the row traffic) made the pass 0.4 % slower: 6.21 -> 6.235 ms)
        if (wib < nprobe) {
            const int j1 = (wib + 1) * s - 1 < live - 1 ? (wib + 1) * s - 1 : live - 1;
            if (!(frc && j1 == 0)) k1 = tree_eval<NW>(slot_ev(j1), false, cm, mlo, mhi, thr, pk, L, sp, op, Mb32, tot2, npad, lane, nfar);
        }
        if (lane == 0) s_k1[wib] = k1;
        __syncthreads();
        SW_STAMP(stamp, it_, sb + 3);
        int qw = -1;
        int k1v[8];   // (all eight words in two wide LDS reads, THEN the selection: a conditional read per word is a dependent trip each)
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) k1v[w2] = s_k1[w2];
#pragma unroll
        for (int w2 = 7; w2 >= 0; --w2) if (w2 < nprobe && k1v[w2] != 0) qw = w2;
        evals += nprobe;
        if (qw >= 0) {
            const int prev = qw * s - 1;
            const int q = (qw + 1) * s - 1 < live - 1 ? (qw + 1) * s - 1 : live - 1;
            const int cnt2 = q - prev - 1;        // <= s - 1 <= 7 slots in between
            int k2 = 0;
            if (wib < cnt2 && !(frc && prev + 1 + wib == 0))
                k2 = tree_eval<NW>(slot_ev(prev + 1 + wib), false, cm, mlo, mhi, thr, pk, L, sp, op, Mb32, tot2, npad, lane, nfar);
            if (lane == 0) s_k2[wib] = k2;
            __syncthreads();
            p = q;
            kind = 0;
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) kind = w2 == qw ? k1v[w2] : kind;
            int k2v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) k2v[i] = s_k2[i];
#pragma unroll
            for (int i = 7; i >= 0; --i) if (i < cnt2 && k2v[i] != 0) { p = prev + 1 + i; kind = k2v[i]; }
            evals += cnt2;
        }
    }
    SW_STAMP(stamp, it_, sb + 4);
    // (wave shuffles outside the one-thread branch: p is uniform over the workgroup)
    const int pe = p >= 0 ? p : 0;
    const int e = slot_ev(pe);
    const bool use_la = NW <= 4 || K >= 32;   // (the table is published 63 positions far only then, see k_resolve_band)
    const int la_i = pe + skip + K;
    const int la_v = slot_ev(la_i < 64 ? la_i - 1 : 0);
    const int la = (use_la && la_i < 64) ? la_v : -1;
    if (threadIdx.x == 0) {
        if (kind == 1) {
            const int dl = ((gsv & 0xff) == 1 && la >= 0 && la - e < 0x3ffffff) ? la - e : 0x3ffffff;
            B.found64[pb + cm] = ((u64)(uint32_t)e << 32) | ((u64)p << 26) | (u64)dl;
        } else if (kind == 2) {
            B.farslot[pb + cm] = p;
        }
        atomicAdd(&B.treecnt[cm], evals);   // (no return value: the read-modify-write does not hold the kernel's end for a round trip)
    }
    if (lane == 0 && nfar) atomicAdd(&st->far_hops, nfar);
    asm volatile("" ::"v"(pf_dummy));   // (the touch's destination register stays reserved to the end)
    SW_STAMP(stamp, it_, sb + 5);
}

// (Round 5, measured and dropped — profiles/r05s_*, r05t_*, r05u_*: a SEARCH form for 512 members and more, where an exact tally gathers
// 128 KB: one workgroup of 4 waves per member CLASSIFIES every slot of the window by the popcount bounds of k_tally_bits' FILT, then
// BISECTS the undecided run with exact tallies — 2.7 instead of 4.1 of them per member and round.  Parity-green at 300 ... 1024
// members, and never faster: with the exact tally by one wave the bisection is three dependent 15-20 us evaluations (1024 members /
// 2 M events 36.6 against 32.3 ms); with all four waves of the workgroup gathering a quarter of the hops each and meeting in LDS it
// EQUALS the one-wave-per-slot kernel (32.36 against 32.41 ms; coin-round stress +1.8 %); with the rows of a wave's three slots in one
// batch of loads, at 128 VGPRs, it spills and loses 5 %.  The 1024-member tally is not bound by the bytes of its exact
// evaluations alone: every slot's row (4 KB) and a wave launch per slot cost as much.)
// (Round 4, measured and dropped — profiles/r04j_*, r04k_*: the workgroups of k_tally_bits finish in dispatch order over 5.7 us
// although a wave lives 2 us on average, which reads like a launch-rate bound.  A kernel with a quarter of the waves — one
// workgroup per member, every wave evaluating four slots at once with interleaved gathers — does start and end within 1.9 us,
// but its waves then take 10 us: the same 58 MB of mask gathers go through L2 at the same ~13 TB/s, and the four counts are a
// serial instruction stream nothing hides.  The tally is bound by the L2 gather rate; 7 waves per SIMD is how it is hidden.)

// ---------------------------------------------------------------------------------
// Finalize: round numbers (swirld.py:217-219) and sees-masks for the events of the batch.
// round[e] = max r with lo[r][creator(e)] <= e  (search of one column of lo).
// Since round 4 the round loop's band pass writes both for every band event it covers in the pass of the event's own round
// (k_resolve_band): what is left is a CHECK — one thread per event, a binary search of its creator's lo column (L2-resident),
// no row read: round[e] equal to the searched round means the pass of that round wrote S[e] as well — and the recomputation
// from the row of the few events no band of their own round covered (band caps, the end of a chain), listed by the check.
// ---------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void finalize_one(const int e, const int lane, const int* __restrict__ L, const int* __restrict__ cr,
                                             const int* __restrict__ lo, const int R, int* round, u64* S, const int npad) {
    const int c = cr[e];
    // 64-ary search of the (non-decreasing) column lo[.][c]: 2 dependent probes for R <= 4096
    int a = 0, len = R;  // the answer lies in [a, a + len); lo[0][c] <= e always (chain start)
    while (len > 1) {
        const int stride = (len + 63) >> 6;
        const int row = a + lane * stride;
        const bool ok = row < a + len && lo[(size_t)row * npad + c] <= e;
        const u64 bal = __ballot(ok);
        const int hi = 63 - __clzll((long long)bal);
        const int end = a + len;
        a += hi * stride;
        len = end - a < stride ? end - a : stride;
    }
    if (lane == 0) round[e] = a;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int v = L[(size_t)e * npad + j * 64 + lane];
        const u64 bm = __ballot(v >= lo[(size_t)a * npad + j * 64 + lane]);
        if (lane == 0) S[(size_t)e * NW + j] = bm;
    }
}

// every event of [first, first + K) from its row (SW_FIN_BAND=0, and callers that have no band pass behind them)
template <int NW>
__global__ void __launch_bounds__(256)
k_finalize_events(const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ lo,
                  int R, int first, int K, int* round, u64* S, int npad) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < K; i += nwaves) finalize_one<NW>(first + i, lane, L, cr, lo, R, round, S, npad);
}

// the check: events whose round[] is not the round their creator's lo column gives go on the list.  fin[0] = length of the
// list (zeroed by the host before the launch), fin[1] = running total (read by sw_get_counters)
__global__ void __launch_bounds__(256)
k_finalize_check(const int* __restrict__ cr, const int* __restrict__ lo, int R, int first, int K,
                 const int* __restrict__ round, int npad, int* __restrict__ list, unsigned* __restrict__ fin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int e = first + i;
    const int c = cr[e];
    int a = 0, b = R;   // the round lies in [a, b): lo[a][c] <= e (lo[0][c] = the chain start), lo[b][c] > e or b = R
    while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (lo[(size_t)mid * npad + c] <= e) a = mid; else b = mid;
    }
    if (round[e] != a) list[atomicAdd(&fin[0], 1u)] = e;
}

// ... and the listed events from their rows, one wave per event
template <int NW>
__global__ void __launch_bounds__(256)
k_finalize_listed(const int* __restrict__ L, const int* __restrict__ cr, const int* __restrict__ lo, int R,
                  const int* __restrict__ list, unsigned* __restrict__ fin, int* round, u64* S, int npad) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int cnt = (int)fin[0];
    for (int i = wave; i < cnt; i += nwaves) finalize_one<NW>(list[i], lane, L, cr, lo, R, round, S, npad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(fin + 2), (unsigned long long)cnt);
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
}


// Unit-stake voter masks with the bit-sliced tally (same as k_voter_masks<NW, true>).
template <int NW>
__global__ void __launch_bounds__(256)
k_voter_masks_bits(const int* __restrict__ wit, const int* __restrict__ L, const int* __restrict__ lo,
                   const uint32_t* __restrict__ S32, uint32_t tot2, int r0, int R, int npad,
                   uint32_t* Sw32, FameCounters* fc) {
    constexpr int W32 = 2 * NW, PLT = ilog2_c(64 * NW) + 1;
    __shared__ int s_pk[4][64 * NW];
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int wv = blockIdx.x * 4 + wib;
    const int total = (R - r0) * npad;
    if (wv >= total) return;
    const int r = r0 + wv / npad, c = wv % npad;
    const int y = wit[(size_t)r * npad + c];
    uint32_t* out = Sw32 + ((size_t)r * npad + c) * W32;
    if (y < 0 || r < 1) {
        if (lane < W32) out[lane] = 0;
        return;
    }
    int* pk = s_pk[wib];
    u64 ex[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int col = j * 64 + lane;
        const int k = L[(size_t)y * npad + col];
        // round[k] == r-1  <=>  lo[r-1][col] <= k < lo[r][col]
        const bool valid = k >= lo[(size_t)(r - 1) * npad + col] && k < lo[(size_t)r * npad + col];
        pk[col] = valid ? k : -1;
        ex[j] = __ballot(wit[(size_t)(r - 1) * npad + col] >= 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t b[PLT];
    bits_accumulate<NW>(pk, S32, b, lane);
    const uint32_t gt = bits_finish<NW>(b, tot2 / 3u);
    const int w = lane % W32;
    uint32_t exw = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j)
        if ((w >> 1) == j) exw = (uint32_t)(ex[j] >> (32 * (w & 1)));
    if (lane < W32) out[w] = gt & exw;
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
            int npad, signed char* fam, unsigned char* cons, unsigned char* newc, FameCounters* fc, int* dec_call, int* dec_by, int call_idx, int part, int nparts) {
    const int r = max_c + part + nparts * (int)blockIdx.x;  // candidate rounds of this part (1 part: all of them)
    const int cx = threadIdx.x;
    const int x = wit[(size_t)r * npad + cx];
    {   // V of SURVEY.md §8d: the reference re-evaluates every witness of rounds max_c+1..max_r as a
        // voter in each decide_fame() call (swirld.py:238-254), decided rounds included
        const int nw_r = __syncthreads_count(x >= 0);
        if (cx == 0 && r > max_c) atomicAdd(&fc->voter_evals, (u64)nw_r);
    }
    if (cons[r]) return;
    // the voters of a round are staged through LDS 64 at a time (64 voters x NW words = one word
    // per thread, one coalesced load): reading them straight from memory is a chain of dependent
    // uniform loads, ~0.3 us per voter
    __shared__ int s_wv[64];
    __shared__ u64 s_m[64 * NW];
    __shared__ u64 s_p2[16];
    bool active = x >= 0 && fam[(size_t)r * npad + cx] < 0;
    u64 V[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) V[j] = 0;
    int any_decided = 0;
    u64 p2 = 0, cvotes = 0, cflips = 0;
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
            __syncthreads();  // the previous chunk has been consumed
            s_m[cx] = sw_row[(size_t)j * 64 * NW + cx];
            if (cx < 64) s_wv[cx] = wv_row[j * 64 + cx];
            __syncthreads();
            u64 acc = 0;
            for (int ci = 0; ci < 64; ++ci) {
                const int wv = s_wv[ci];  // uniform
                if (wv < 0) continue;
                ++nvoters;
                const u64* m = s_m + ci * NW;
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
                        if (active) { ++cvotes; cflips += !sm; }
                    }
                }
                acc |= (u64)bitv << ci;
            }
            Vn[j] = acc;
        }
        if (active && d >= 2) {
            if (!coin_round && best_idx != SW_INF) {
                fam[(size_t)r * npad + cx] = (signed char)best_v;  // swirld.py:263
                dec_call[(size_t)r * npad + cx] = call_idx;        // which decide_fame() call decided x, and which voter
                dec_by[(size_t)r * npad + cx] = best_idx;
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
    {   // one atomic per workgroup and counter
        u64 t = p2, tv = cvotes, tf = cflips;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            t += (u64)__shfl_xor((long long)t, off);
            tv += (u64)__shfl_xor((long long)tv, off);
            tf += (u64)__shfl_xor((long long)tf, off);
        }
        __shared__ u64 s_cv[16], s_cf[16];
        if ((cx & 63) == 0) { s_p2[cx >> 6] = t; s_cv[cx >> 6] = tv; s_cf[cx >> 6] = tf; }
        __syncthreads();
        if (cx == 0) {
            u64 tot = 0, totv = 0, totf = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tot += s_p2[w]; totv += s_cv[w]; totf += s_cf[w]; }
            if (tot) atomicAdd(&fc->majority_evals, tot);
            if (totv) atomicAdd(&fc->coin_votes, totv);
            if (totf) atomicAdd(&fc->coin_flips, totf);
        }
    }
}


// The same elections with NW threads per candidate (SW_ELECT_IMPL=1, the default for 128 members and more): thread (j, cx)
// tallies the 64 voters of mask word j for candidate cx, so a level is 1/NW of the dependent chain and NW times the waves.
// A workgroup holds CG candidates of ONE round (CG * NW threads), a round is npad / CG workgroups: at 256 members two of 512
// threads (323 rounds x 1024 threads left 67 CUs with twice the work of the others), at 1024 members sixteen of 1024.
//   * the voters' masks of a level are staged in LDS up to 256 members (8 KB), read from global memory beyond (a round of
//     them is 128 KB at 1024 members; the word a wave reads is the same for all its lanes);
//   * what depends on the voter only is computed once per level, not once per (voter, candidate): its total `tot` (the
//     popcount / stake of its mask), kept as the two thresholds the vote is compared with —
//         v  = !(no > yes)        <=>  yes >= ceil(tot / 2)
//         3 * max(yes, no) > 2T   <=>  yes >= thr3  or  yes <= tot - thr3,   thr3 = floor(2T / 3) + 1
//     (swirld.py:24-27, 260-262) — which leaves NW and-popcounts and four compares per pair;
//   * the first decider (smallest event index among the voters with a supermajority, swirld.py:263) is a min over the
//     packed key (event << 1 | vote); the vote words and keys of the NW threads of a candidate are exchanged through LDS and
//     every one of them takes the same decision (thread j = 0 records it);
//   * what concerns the ROUND — "every witness decided, at least one of them in this call" (swirld.py:274-277) — is agreed
//     by the workgroups of the round through three words in `rsc` (zeroed by the host): open witnesses, decided flag,
//     arrival ticket; the last one to arrive publishes it.
// Round 4 (profiles/r04o_*): 230 us -> see DESIGN.md §4 for the 256 x 1 M elections; one thread per candidate (k_elections)
// took 6.7 ms per pass at 1024 members / 2 M events.
template <int NW, bool UNIT, int CG>
__global__ void __launch_bounds__(CG * NW)
k_elections_tiled(const int* __restrict__ wit, const u64* __restrict__ Sw, const unsigned char* __restrict__ coin,
                  const uint32_t* __restrict__ stake, uint32_t tot2, int coin_period, int max_c, int R,
                  int npad, signed char* fam, unsigned char* cons, unsigned char* newc, FameCounters* fc, int* dec_call, int* dec_by,
                  int call_idx, int part, int nparts, int* rsc) {
    constexpr int NT = CG * NW;
    constexpr bool STAGE = NW <= 4;          // voters' masks of a level in LDS
    static_assert(CG % 64 == 0 && NT <= 1024 && (64 * NW) % CG == 0, "a wave has one j; a round is a whole number of workgroups");
    constexpr int GB = 64 * NW / CG;         // workgroups per round (npad = 64 NW)
    const int rb = (int)blockIdx.x / GB, g = (int)blockIdx.x - rb * GB;
    const int r = max_c + part + nparts * rb;
    const int tid = threadIdx.x;
    const int j = tid / CG, cxl = tid - j * CG;   // CG is a multiple of 64: j is uniform in a wave
    const int cx = g * CG + cxl;
    const int x = wit[(size_t)r * npad + cx];
    {
        const int nw_r = __syncthreads_count(j == 0 && x >= 0);
        if (tid == 0 && r > max_c && nw_r) atomicAdd(&fc->voter_evals, (u64)nw_r);
    }
    if (cons[r]) return;
    __shared__ int s_wv[64 * NW];
    __shared__ int s_half[64 * NW];   // ceil(tot / 2) of the voter
    __shared__ int s_lot[64 * NW];    // tot - thr3 (signed)
    __shared__ int4 s_vt[(STAGE && UNIT) ? 64 * NW : 1];   // {witness, ceil(tot / 2), tot - thr3, 0} of the voter: ONE read per voter in the branch-free loop
    __shared__ u64 s_m[STAGE ? 64 * NW * NW : 1];
    __shared__ u64 s_V[NW][CG];
    __shared__ uint32_t s_key[NW][CG];
    __shared__ int s_nv[NW];
    __shared__ u64 s_p2[16], s_cv[16], s_cf[16];
    const uint32_t thr3 = tot2 / 3u + 1u;
    bool active = x >= 0 && fam[(size_t)r * npad + cx] < 0;
    u64 V[NW];
#pragma unroll
    for (int jj = 0; jj < NW; ++jj) V[jj] = 0;
    int any_decided = 0;
    u64 p2 = 0, cvotes = 0, cflips = 0;
    for (int d = 1; r + d < R; ++d) {
        if (!__syncthreads_or(active)) break;  // also: every read of the previous level is done
        const int rv = r + d;
        const int* wv_row = wit + (size_t)rv * npad;
        const u64* sw_row = Sw + (size_t)rv * npad * NW;
        const bool coin_round = (d % coin_period) == 0;
        for (int i = tid; i < 64 * NW; i += NT) s_wv[i] = wv_row[i];
        if (STAGE)
            for (int i = tid; i < 64 * NW * NW; i += NT) s_m[i] = sw_row[i];
        __syncthreads();
        if (d >= 2) {   // per voter, once: the total of its mask as the two thresholds
            for (int i = tid; i < 64 * NW; i += NT) {
                uint32_t tot = 0;
                if (s_wv[i] >= 0) {   // (the mask row of a member without a witness in this round is never read)
                    const u64* m = STAGE ? s_m + i * NW : sw_row + (size_t)i * NW;
#pragma unroll
                    for (int jj = 0; jj < NW; ++jj) {
                        u64 mm = m[jj];
                        if (UNIT) {
                            tot += __popcll(mm);
                        } else {
                            while (mm) {
                                const int b = __ffsll((long long)mm) - 1;
                                mm &= mm - 1;
                                tot += stake[jj * 64 + b];
                            }
                        }
                    }
                }
                s_half[i] = (int)((tot + 1u) >> 1);
                s_lot[i] = (int)tot - (int)thr3;
                if constexpr (STAGE && UNIT) s_vt[i] = make_int4(s_wv[i], (int)((tot + 1u) >> 1), (int)tot - (int)thr3, 0);
            }
            __syncthreads();
        }
        u64 acc = 0;
        uint32_t key = 0xffffffffu;
        int nv = 0;
        if (STAGE && UNIT && d >= 2 && !coin_round) {
            // the bulk of the work — an ordinary round at distance >= 2, unit stakes — BRANCH-FREE and eight voters at a time (round 5,
            // from the ISA: the general loop below is one voter per trip with three dependent LDS round trips and a dozen branches
            // in it; the kernel was bound by those latencies, not by its and-popcounts).  A member without a witness in the voters'
            // round is masked out instead of skipped: its bit stays 0, its key stays "none", it is not counted.
            for (int c0 = 0; c0 < 64; c0 += 8) {
                uint32_t bits = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = j * 64 + c0 + u;
                    const int4 vt = s_vt[c];
                    const u64* m = s_m + c * NW;
                    uint32_t yes = 0;
#pragma unroll
                    for (int jj = 0; jj < NW; ++jj) yes += __popcll(m[jj] & V[jj]);
                    const uint32_t valid = vt.x >= 0 ? 1u : 0u;
                    const uint32_t v = yes >= (uint32_t)vt.y ? 1u : 0u;                    // majority(): tie -> True (swirld.py:24-27)
                    const bool sm = (yes >= thr3) | ((int)yes <= vt.z);                    // the winning side holds more than 2/3 of the stake
                    const uint32_t k = (valid & (uint32_t)sm) ? (((uint32_t)vt.x << 1) | v) : 0xffffffffu;
                    key = k < key ? k : key;
                    bits |= (valid & v) << u;
                    nv += (int)valid;
                }
                acc |= (u64)bits << c0;
            }
        } else
        for (int ci = 0; ci < 64; ++ci) {
            const int c = j * 64 + ci;
            const int wv = s_wv[c];  // uniform in the wave
            if (wv < 0) continue;
            ++nv;
            const u64* m = STAGE ? s_m + c * NW : sw_row + (size_t)c * NW;   // (the same words for every lane of the wave)
            int bitv;
            if (d == 1) {
                bitv = (int)((m[cx >> 6] >> (cx & 63)) & 1ull);  // x in s (swirld.py:258)
            } else {
                uint32_t yes = 0;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj) {
                    u64 mm = m[jj] & V[jj];
                    if (UNIT) {
                        yes += __popcll(mm);
                    } else {
                        while (mm) {
                            const int b = __ffsll((long long)mm) - 1;
                            mm &= mm - 1;
                            yes += stake[jj * 64 + b];
                        }
                    }
                }
                const int v = yes >= (uint32_t)s_half[c];                 // majority(): tie -> True (swirld.py:24-27)
                const bool sm = yes >= thr3 || (int)yes <= s_lot[c];      // the winning side holds more than 2/3 of the stake
                if (!coin_round) {
                    const uint32_t k = sm ? (((uint32_t)wv << 1) | (uint32_t)v) : 0xffffffffu;
                    key = k < key ? k : key;
                    bitv = v;
                } else {
                    bitv = sm ? v : (int)coin[wv];  // swirld.py:267-272
                    if (active) { ++cvotes; cflips += !sm; }
                }
            }
            acc |= (u64)bitv << ci;
        }
        s_V[j][cxl] = acc;
        s_key[j][cxl] = key;
        if (cxl == 0) s_nv[j] = nv;
        __syncthreads();
        key = 0xffffffffu;
#pragma unroll
        for (int jj = 0; jj < NW; ++jj) {
            V[jj] = s_V[jj][cxl];
            const uint32_t k = s_key[jj][cxl];
            key = k < key ? k : key;
        }
        if (active && d >= 2) {
            if (!coin_round && key != 0xffffffffu) {
                const int best_idx = (int)(key >> 1);
                active = false;
                any_decided = 1;
                if (j == 0) {
                    fam[(size_t)r * npad + cx] = (signed char)(key & 1u);  // swirld.py:263
                    dec_call[(size_t)r * npad + cx] = call_idx;            // which decide_fame() call decided x, and which voter
                    dec_by[(size_t)r * npad + cx] = best_idx;
                }
                int le = 0;  // voters after the first decider never evaluate x (each of the NW threads counts its own word)
                for (int ci = 0; ci < 64; ++ci) {
                    const int wv = s_wv[j * 64 + ci];
                    le += (wv >= 0 && wv <= best_idx);
                }
                p2 += le;
            } else {
                p2 += nv;
            }
        }
    }
    const int x_open = j == 0 && x >= 0 && active;   // (`active` mirrors fam < 0 for this thread's candidate)
    const int n_open = __syncthreads_count(x_open);
    const int dec = __syncthreads_or(any_decided);
    if (tid == 0) {  // swirld.py:274-277
        if (GB == 1) {
            if (dec && n_open == 0) { newc[r] = 1; cons[r] = 1; }
        } else {     // ... over the workgroups of the round
            if (n_open) atomicAdd(&rsc[3 * r], n_open);
            if (dec) atomicOr(&rsc[3 * r + 1], 1);
            __threadfence();
            if (atomicAdd(&rsc[3 * r + 2], 1) == GB - 1) {
                __threadfence();
                if (atomicAdd(&rsc[3 * r], 0) == 0 && atomicAdd(&rsc[3 * r + 1], 0) != 0) {
                    newc[r] = 1;
                    cons[r] = 1;
                }
            }
        }
    }
    {   // one atomic per workgroup and counter
        u64 t = p2, tv = cvotes, tf = cflips;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            t += (u64)__shfl_xor((long long)t, off);
            tv += (u64)__shfl_xor((long long)tv, off);
            tf += (u64)__shfl_xor((long long)tf, off);
        }
        if ((tid & 63) == 0) { s_p2[tid >> 6] = t; s_cv[tid >> 6] = tv; s_cf[tid >> 6] = tf; }
        __syncthreads();
        if (tid == 0) {
            u64 tot = 0, totv = 0, totf = 0;
            for (int w = 0; w < NT / 64; ++w) { tot += s_p2[w]; totv += s_cv[w]; totf += s_cf[w]; }
            if (tot) atomicAdd(&fc->majority_evals, tot);
            if (totv) atomicAdd(&fc->coin_votes, totv);
            if (totf) atomicAdd(&fc->coin_flips, totf);
        }
    }
}

// find_order (swirld.py:280-311): the kernels live in order.hip.h
#include "order.hip.h"


// ---------------------------------------------------------------------------------
// Gossip side: what a peer needs from this node, from the device-resident state.
// k_known_heights: {member -> height of the newest event of that member my head can see}
// (swirld.py:125-126: {c: height[h] for c, h in can_see[head].items()}); -1 = absent.
// k_sync_diff: the events ask_sync sends (swirld.py:154-161) as per-member chain position ranges.
// The reference walks back from the head, not descending into parents the asker reported as known
// (height <= its reported height for that creator).  For heights reported from a real can_see row a
// known event has only known ancestors, so the walk reaches exactly the ancestors-or-self of the
// head that are unknown: on member m's chain the positions after the last one with height <=
// known[m], up to the newest event of m the head sees; the head itself is always sent.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_known_heights(const int* __restrict__ L, const int* __restrict__ ht, int head, int npad, int* out) {
    const int m = threadIdx.x;
    const int k = L[(size_t)head * npad + m];
    out[m] = k >= 0 ? ht[k] : -1;
}

__global__ void __launch_bounds__(1024)
k_sync_diff(const int* __restrict__ L, const int* __restrict__ ht, const int* __restrict__ seq, const int* __restrict__ cr,
            const int* __restrict__ chain_start, const int* __restrict__ chain_ev, const int* __restrict__ known,
            int head, int npad, int* pos_first, int* pos_end) {
    const int m = threadIdx.x;
    const int seen = L[(size_t)head * npad + m];
    int p0 = 0, p1 = 0;
    if (seen >= 0) {
        p1 = seq[seen] + 1;
        const int h = known[m];
        const int cs = chain_start[m];
        int a = 0, b = p1;  // first position whose height exceeds the asker's (heights grow along a chain)
        if (h >= 0)
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (ht[chain_ev[cs + mid]] > h) b = mid; else a = mid + 1;
            }
        p0 = a;
        if (m == cr[head] && p0 >= p1) p0 = p1 - 1;  // the walk starts at the head whatever the asker knows
    }
    pos_first[m] = p0;
    pos_end[m] = p1 > p0 ? p1 : p0;
}

// sw_rewind: every table of the voting state back to its initial value in ONE launch (eleven fills and memsets took
// ~0.1 ms of every measured pass): blockIdx.y = the table, 32-bit words, grid-stride.
// decide_fame: the new_c flags and the round-level agreement words of the elections, zeroed in ONE launch (two memsets were
// three fill kernels of ~5 us each in front of the elections, at the end of every pass)
__global__ void __launch_bounds__(256)
k_fame_prep(unsigned char* newc, int R, int* rsc, int n_rsc) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) newc[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rsc; i += gridDim.x * blockDim.x) rsc[i] = 0;
}

struct RewindJob { int* p[12]; unsigned long long n[12]; int v[12]; };
__global__ void k_rewind_fill(RewindJob J) {
    int* p = J.p[blockIdx.y];
    const size_t n = J.n[blockIdx.y];
    const int v = J.v[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void k_fill_i32(int* p, size_t n, int v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
