// Host side of the C-ABI (include/swirld_hip.h): owns the HBM-resident hashgraph state of one
// Node view and drives the kernels of kernels.hip.h on one HIP stream.
//
// Path (reference file:line):  Node.add_event swirld.py:114-120 -> sw_append_events;
// Node.divide_rounds swirld.py:187-222 -> sw_divide_rounds; Node.decide_fame
// swirld.py:224-277 -> sw_decide_fame; Node.find_order swirld.py:280-311 -> sw_find_order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/swirld_hip.h"
#include "kernels.hip.h"
#define SW_PROV_ROWS 16   // sub-batches of one divide_rounds call that can be swept in chunks
#define SW_RANGE_SLOTS 16 // event ranges swept by sw_cansee_range between two rewinds
#include "crypto.hip.h"
#include "exact.hip.h"

namespace {

std::string g_create_error;

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
};

// growable int32 array in pinned host memory: the ordered events (Node.transactions) — device copies land in it directly
struct PinnedVec {
    int32_t* p = nullptr;
    size_t n = 0, cap = 0;
    size_t size() const { return n; }
    int32_t* data() { return p; }
    int32_t& operator[](size_t i) { return p[i]; }
    void clear() { n = 0; }
    bool resize(size_t m) {   // (only while nothing is in flight into the array)
        if (m > cap) {
            const size_t nc = std::max(m, cap + cap / 2 + 4096);
            int32_t* q = nullptr;
            if (hipHostMalloc((void**)&q, nc * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (n) memcpy(q, p, n * sizeof(int32_t));
            if (p) (void)hipHostFree(p);
            p = q;
            cap = nc;
        }
        n = m;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
};

}  // namespace

// can_see table under HIP virtual memory management (windowed mode, sw_set_window): ONE reserved address
// range for the whole table, physical chunks mapped as events arrive and recycled from below the
// eviction horizon — every kernel keeps the same base pointer and the same row arithmetic.
struct VmTable {
    bool active = false;
    char* base = nullptr;
    size_t va_bytes = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handle;  // per chunk slot of the address range
    std::vector<char> mapped;
    std::vector<hipMemGenericAllocationHandle_t> pool;    // physical chunks unmapped by evictions, ready for reuse
    size_t hi = 0;        // chunk slots [lo, hi) are mapped
    size_t lo = 0;
    int64_t evictions = 0;
};

// ONE hashgraph's round loop over several linked contexts (sw_split_link; kernels.hip.h SplitDst): what the host threads of
// the parts share.  Every part runs its own sw_divide_rounds (same events, same call); inside an iteration of the round loop
// its stream waits, at both kernel boundaries, for the other parts' kernels of that boundary — an event per (part, boundary,
// iteration), waited for only after its owner has RECORDED it (`rec`: the producer's packet is then in front of the
// waiter's wherever the runtime maps the streams).
struct SplitGroup {
    int parts = 0;
    sw_ctx* ctx[SW_MAX_PARTS] = {nullptr};
    std::atomic<long long> rec[2][SW_MAX_PARTS];   // iterations whose band / tally event the part has recorded
    std::atomic<int> abort{0};                     // a part failed: the others stop waiting for it
    hipEvent_t ev[2][SW_MAX_PARTS][64] = {};
};

struct sw_ctx {
    int n = 0, nw = 0, npad = 0, coin_period = 6, device = 0;
    SplitGroup* split = nullptr;   // linked with other contexts (sw_split_link)
    int split_part = 0;
    long long split_iter = 0;      // iterations of the round loop enqueued since the link
    bool split_failed = false;     // a meeting of the parts failed (a part gave up): the call reports it
    SplitDst* d_split = nullptr;   // device copies of the SplitDst the split kernels read: [0] this part's (linked), [q] part q's (SW_SPLIT_EMULATE)
    SplitDst split_up[SW_MAX_PARTS] = {};   // ... and what was uploaded last
    int split_saved[4] = {0, 0, 0, 0};   // tally choice of the context before the link pinned the one-wave-per-slot tally (restored by the unlink)
    int split_emulate = 0;         // SW_SPLIT_EMULATE (measurement): this many parts played by this one context, one behind the other
    VmTable vm;
    int64_t first_resident = 0;   // can_see rows below this event index have been evicted (windowed mode)
    int64_t window_lapse = 0;     // sw_set_window_lapse: a member silent for more than this many events no longer holds the window back (0 = never)
    std::vector<char> lapsed;     // ... and is marked here: its further events are refused (SW_ERANGE) until a rewind / reset
    bool vm_scratch_ok = false;   // windowed mode: the halo scratch rows behind row `cap` are mapped (the sweep may run in chunks)
    DBuf<int32_t> d_ordpos;       // per member: chain positions already ordered (find_order's search bound)
    // exact path for forked hashgraphs (exact.hip.h): entered at the first forked event, left by sw_reset
    bool exact = false;
    int forks_mode = 1;           // 1: accept forks (exact path), 0: refuse them (SW_ENOTSUP, nothing stored)
    DBuf<int32_t> x_worder, x_wcnt, x_newr, x_queue, x_fw, x_items_ev, x_rounds;
    DBuf<signed char> x_fam_ev, x_votes;
    DBuf<unsigned char> x_tbd, x_sm, x_done, x_visited, x_sflag, x_white;
    DBuf<double> x_times, x_tsort, x_items_ts;
    DBuf<long long> x_hdr;
    int x_Rcap = 0;               // rows of x_worder / x_wcnt / x_done / x_newr
    int64_t x_cap = 0;            // events of x_fam_ev / x_tbd / x_queue / x_visited / x_items_*
    FameCounters x_fc_base{};     // fame counters of the fast path at the switch (the exact path counts from zero)
    bool unit_stake = true;
    uint32_t tot = 0;
    std::vector<uint32_t> stake_h;
    hipStream_t stream = nullptr;
    hipStream_t stream_cs = nullptr;   // can_see sweeps run here, ahead of the round loop
    hipStream_t stream_aux = nullptr;  // per-sub-batch finalize + voter masks run here, behind the round loop
    FameCounters fc_seen{};            // device fame counters already added to ctr
    std::vector<hipEvent_t> cs_events;
    int pipe = 5;                       // sub-batches per divide_rounds call behind the short first one (pipelining depth)
    std::string err;

    // host mirror of the DAG (validation, height, chains)
    std::vector<int32_t> cr, sp, op, ht;
    std::vector<int32_t> head;      // latest event per member (-1 none)
    std::vector<int32_t> first_ev;  // first event (root) per member (-1 none)
    std::vector<int32_t> nev;       // events per member so far (next chain position)
    bool pool_h_valid = true;       // chain_ev_h mirrors the device pool (bulk appends invalidate it)
    bool poisoned = false;          // a HIP failure left the context half-updated: every later call fails
    hipStream_t stream_io = nullptr;  // payload (timestamps, signatures) uploads of bulk appends
    hipEvent_t ev_payload = nullptr;  // ... and their completion (find_order / coin bits wait for it)
    bool payload_pending = false;
    std::vector<int32_t> seq_h;     // chain position of every event (host mirror: small appends pack creator(op) | seq(op))
    SmallRec* h_small = nullptr;    // pinned staging of small appends (one packed record per event)
    size_t h_small_cap = 0;
    hipEvent_t ev_small = nullptr;  // ... and the completion of its last upload
    bool small_pending = false;
    char* h_pin = nullptr;          // pinned staging of bulk appends
    size_t h_pin_cap = 0;
    int max_height = 0;
    int64_t N = 0, cap = 0, divided = 0;

    // device: events
    DBuf<int32_t> d_cr, d_sp, d_op, d_ht, d_seq, d_round, d_L, d_chain_ev;
    DBuf<unsigned char> d_coin, d_sig;
    DBuf<double> d_t;
    DBuf<u64> d_S;
    DBuf<int32_t> d_finlist;   // events k_finalize_check found without a round from the band pass (one entry per event at most)
    unsigned* d_fin = nullptr; // [0] length of that list for the launch in flight, [2..3] running total (u64)
    DBuf<int32_t> d_chain_start;  // npad: offset of each member's segment in the chain pool
    DBuf<int32_t> d_chain_cnt;    // npad: events per member (segments have slack: geometric growth)
    DBuf<int4> d_cdesc;           // pool-indexed chain descriptors of the dataflow can_see sweep (same indexing as chain_ev)
    DBuf<int32_t> d_bounds;       // [cuts][npad] chain positions of the sub-batch cuts of the running divide_rounds call
    DBuf<long long> d_cuts;
    // chunk-parallel can_see sweep (k_cansee_chunks): per chunked sub-batch 2G + 2 rows of chain positions
    // {w_0, a_0, w_1, a_1, ..., end, end}, the halo rows' scratch table, provisional-entry counters (inside d_rb)
    DBuf<int32_t> d_cbnd;
    DBuf<long long> d_ccuts;
    unsigned* d_prov = nullptr;   // [SW_PROV_ROWS][SW_MAX_CHUNKS] provisional entries per chunk, then [SW_PROV_ROWS] repaired entries
    int chunks = 4;               // SW_CHUNKS: chunks swept concurrently per sub-batch (1 = the unchunked k_cansee_flow)
    int chunk_cfg = 0;            // SW_CHUNK_CFG: 0 = 4 columns per lane, FIFO 8, ring 8; 1 = 2 columns, 8 / 16; 2 = 4 columns, 4 / 8
    int64_t halo = 0;             // SW_HALO: events recomputed in front of a chunk (default 32 x npad: ~2.4x the age of a row's oldest entry at uniform gossip)
    int64_t chunk_min = 8192;     // SW_CHUNK_MIN: smallest chunk worth a halo (round 5: 16384 -> 8192, the short first sub-batch — the one sweep nothing hides — goes out as 4 chunks instead of 3: 6.03 -> 5.98 ms, profiles/r05n_knobs_256x1M.log)
    bool chunks_off = false;      // set when a call had to sweep chunks twice (members silent for longer than the halo): unchunked from then on
    struct ChunkPlan { int G = 0; int row0 = 0; int64_t a[SW_MAX_CHUNKS + 1]; int64_t w[SW_MAX_CHUNKS]; };
    std::vector<ChunkPlan> chunk_plan;   // per sub-batch of the running call
    std::vector<long long> ccuts_stage;  // host staging of the chunk cuts (persistent: uploaded without a sync)
    // multi-GPU split of the can_see table by event ranges (sw_cansee_range / sw_import_rows): the ranges swept here,
    // the rows present in the table without having been divided yet, their cut tables and counters
    struct RangeRec { int64_t first = 0, K = 0; ChunkPlan pl; int row0 = 0; int slot = 0; bool repaired = false; };
    std::vector<RangeRec> ranges;
    std::vector<std::pair<int64_t, int64_t>> present;   // [a, b) rows present, sorted, merged
    DBuf<int32_t> d_rbnd;
    DBuf<long long> d_rcuts;
    std::vector<long long> rcuts_stage;
    unsigned* d_rprov = nullptr;      // [SW_RANGE_SLOTS][SW_MAX_CHUNKS + 1]: provisional entries per chunk, repaired entries
    hipEvent_t ev_user = nullptr;
    std::vector<int32_t> chain_cap;   // per member: capacity of its segment
    int64_t pool_used = 0;             // ints of the chain pool handed out
    DBuf<int32_t> d_chain_len;    // npad: events per member visible to the running round loop
    std::vector<int32_t> blk_hmin, blk_hmax;  // height span per 4096-event block (ingest-time index)
    std::vector<int32_t> divided_head;
    DBuf<uint32_t> d_stake;       // npad

    // device: levels
    DBuf<int32_t> d_lev_cnt, d_lev_start, d_lev_cursor;
    DBuf<int32_t> d_lev_cback, d_lev_pinbase, d_lev_pos;   // level sweep: the back cursors, pinned events in front of each level, descriptor position per event
    DBuf<uint8_t> d_lev_pin;     // level sweep: events whose row slice a child looks for after it has left the ring (k_level_hist)
    DBuf<int4> d_desc;

    // device: rounds
    int Rcap = 0;  // rows allocated in lo/lopos/wit/fam/cons/newc
    int R = 0;     // max round + 1
    DBuf<int32_t> d_lo, d_lopos, d_wit;
    DBuf<signed char> d_fam;
    DBuf<int32_t> d_dec_call, d_dec_by;  // per witness slot: the decide_fame() call that decided it and the deciding voter (Node.votes bookkeeping)
    struct FameCall { int max_c, R; int64_t divided; };
    std::vector<FameCall> fame_calls;    // one record per decide_fame() call
    bool votes_partial = false;          // a partitioned commit left dec_call / dec_by of other parts' witnesses unset: no Node.votes
    std::vector<int32_t> cons_call;      // per round: the call that added it to `consensus` (-1: not yet)
    DBuf<unsigned char> d_cons, d_newc;
    DBuf<u64> d_Sw;
    int Sw_rows = 0;
    bool debug_timing = false;   // SW_DEBUG_TIMING=1
    double stage_us[8] = {0};    // sw_divide_rounds host stages: sweeps enqueued, loop set-up, round loop, front rows, aux launches, final syncs
    int64_t stage_calls = 0;
    unsigned long long* d_dbg = nullptr;  // SW_DEBUG_CLOCKS=1: phase stamps of the round-loop kernels
    unsigned long long* d_dbg_blk = nullptr;   // SW_DEBUG_CLOCKS=3: end time of every workgroup of the loop kernels, per iteration
    int64_t dbg_iter_base = 0;            // round_iterations at the last rewind: the phase stamps are indexed by the iterations since
    int dbg_minor = 1;                    // SW_DEBUG_CLOCKS=2: only the entry / band start / end stamps (the others drain the wave)
    struct { int32_t* p = nullptr; } d_front;   // inside d_rb
    int32_t* d_treecnt = nullptr;                // inside d_rb: tallies evaluated per member in the running loop (k_tally_tree)
    DBuf<unsigned char> d_small;   // device copy of the packed records of the current small append
    hipEvent_t ev_aux_done = nullptr, ev_cs_done = nullptr, ev_main_mark = nullptr, ev_bounds = nullptr, ev_loop_done = nullptr;
    std::vector<int32_t> divided_cnt;   // per member: events already divided (chain positions below `divided`)
    std::vector<int32_t> bounds_stage;  // host staging of the cut table (persistent: uploaded without a sync)
    DBuf<int32_t> d_evalround, d_evalpos, d_lo_r, d_cur, d_unres, d_lo_next, d_pos_next, d_farslot, d_force, d_cand, d_gallop;
    DBuf<u64> d_Mb;
    DBuf<int32_t> d_Pc;    // [MCAP + 1] popcounts of the band masks (row 0 = 0, like d_Mb): the cheap bounds of k_tally_bits<., FILT>.
                           // INVARIANT: whoever writes a row of d_Mb writes its popcount here (k_resolve_band's band pass is the only writer of both,
                           // in its plain and its split form): a `sure` verdict of the filtered tally never looks at the mask itself
    DBuf<int32_t> d_rsc;   // [R][3] round-level agreement of k_elections_tiled: open witnesses, decided flag, arrival ticket
    DBuf<u64> d_found64;   // [2][npad] {event << 32 | slot << 26 | look-ahead} of the members' first passing candidates (LoopBufs::found64)
    // one device block read back with ONE copy per round-loop shot: loop state (x2), sweep error flag, per-member
    // front rounds; and its pinned host mirror
    unsigned char* d_rb = nullptr;
    unsigned char* h_rb = nullptr;        // the read-back slot that was read last (one of h_rb_all's)
    unsigned char* h_rb_all = nullptr;    // SW_PROV_ROWS pinned slots: the loops of a call's sub-batches are read back one behind the other
    std::vector<hipEvent_t> rb_events, shot_events;   // per slot: read-back complete / last iteration enqueued so far
    int shot_pct = 100, shot_extra = 2;   // SW_SHOT_PCT / SW_SHOT_EXTRA: a loop's first shot = predicted iterations x pct / 100 + extra (tests: 50 makes every loop top up)
    size_t rb_bytes = 0;
    RState* d_state = nullptr;
    FameCounters* d_fc = nullptr;   // header of d_newc: the fame counters travel with the new_c flags in one copy
    unsigned char* h_fame = nullptr;  // pinned: [FameCounters][newc flags]
    size_t h_fame_cap = 0;
    std::vector<int32_t> front;       // per member: max r with lo[r][c] finite (-1 none); the device keeps it (d_front), the host mirror follows
    std::vector<int32_t> front_dev;   // read-back buffer of d_front
    std::vector<int32_t> lo0_h;       // host copy of lo[0][.] (chain starts)
    std::vector<unsigned char> cons_h;  // host mirror of consensus
    int sw_dirty_from = 1;            // voter masks of rounds >= this must be (re)built

    // tuning
    int skip = 1;         // SW_SKIP: window offset of a fresh round (0 = off); 1 measured best at 256 members (310 vs 326 iterations)
    int gallop_after = 2; // SW_GALLOP: strided candidate windows after this many windows without a passing candidate (0 = never); 2 costs uniform gossip nothing (the first window passes there) and cuts hot-member hashgraphs from 2756 to ~200 iterations
    int elect_impl = 1; // 1: NW threads per candidate (k_elections_tiled, 128 members and more), 0: one thread per candidate
    int elect_cg = 128; // SW_ELECT_CG (256 members): candidates per workgroup of k_elections_tiled — 64, 128 or 256
    int K = 28;        // candidates per member per tally launch: 7 waves per SIMD (the 8th slot is
                       // taken by the concurrent can_see sweep; 29+ costs a second wave generation)
    int MCAP = 0;      // largest band (events) the mask table can hold
    int NEARCAP = 0;   // band cap at round entry (doubles up to MCAP when a far candidate needs a tally)
    int BATCH = 24;    // loop iterations between host checks
    int cansee_impl = 6;  // 6 = dataflow sweep (no levels, no barriers: k_cansee_chunks / k_cansee_flow); 2 / 3 = level-bucketed sweep (k_cansee_stream: the default beyond 256 members; one kernel since round 6, both values select it)
    int ring_H_req = 0;   // SW_RING_H override (0 = automatic)
    int flow_cfg = 1;     // SW_FLOW_CFG: FIFO / ring depths of the dataflow sweep: 0 = 16/32, 1 = 8/16, 2 = 16/16, 3 = 8/32
    int tally_impl = 1;   // 0 = column-lane tally, 1 = bit-sliced, one wave per slot (unit stake only), 2 = bit-sliced two-level search (k_tally_tree)
    bool tally_auto = true;   // SW_TALLY_IMPL not set: large calls of ~256-member hashgraphs without strongly skewed activity use 2
    bool K_auto = true;       // SW_TALLY_K not set: 28 slots for the flat tally, 32 for the tree
    int K_flat = 28;
    int eval_src = 0;      // which half of d_evalround / d_evalpos the last round-loop run left the members' exhaustion marks in
    int band_blocks = 512; // workgroups of the resolve+band kernel
    int fin_blocks = 1024; // SW_FIN_BLOCKS: workgroups of a k_finalize_events launch that runs beside a round loop (profiles/r04q_*: 8192 of them
                           // slow the loop's gathers down; 1024 with the early finalize below: 7.30 -> 7.16 ms per pass at 256 x 1 M, 69.7 -> 66.9 ms at 10 M)
    int mid_pct = 0;       // SW_MID_PCT: where the last sub-batch's round loop is interrupted once for an early finalize (0 = never: with
                           // SW_FIN_BAND=1 there is nothing left to hide; 88 was the default of the row-reading finalize)
    int fin_band = 1;      // SW_FIN_BAND: round[] and the sees-masks come from the round loop's band pass, the finalize launch only checks (1) / every event from its row (0)
    int band_fast = 1;     // SW_BAND_FAST: k_resolve_band takes full groups of 8 band events through a path with fixed indices and compile-time offsets
                           // (round 5, profiles/r05a_*: band phase 2.66 -> 1.90 us, 7.00 -> 6.72 ms per pass at 256 members / 1 M events; 0 = the generic path only)
    int tally_filter = 0;  // SW_TALLY_FILTER: the one-wave-per-slot tally bounds a verdict by the popcounts of the hop masks before it gathers the masks
                           // (round 5, profiles/r05e_*: default beyond 256 members, where the tally is bound by the bytes it gathers — 1024 members / 2 M
                           // events 34.8 -> 32.7 ms, 57.4 -> 61.2 M events/s; at 256 members the one-wave-per-slot kernel is bound by its 8 192 wave launches,
                           // with or without the gathers: 6.58 -> 6.34 ms, the two-level search 6.08)
    int tally_pf = 1;      // SW_TALLY_PF: the first waves of every XCD touch the band-mask table at the head of k_tally_bits (+1 %)

    // round-loop graph
    struct GraphKey { int Rcap; int64_t N; void* lo; void* L; void* chain; void* S; void* round; int K, tally_impl, BATCH, MCAP, variant; };
    bool use_graph = true;
    hipGraph_t loop_graph[4] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t loop_exec[4] = {nullptr, nullptr, nullptr, nullptr};
    int graph_big = 64;    // SW_GRAPH_BIG: iterations of the largest replayable graph (0 = none beyond 24).  Every graph boundary costs
                           // the loop ≈ 8 us (profiles/r04y_loop_phases_256x1M.txt: iterations 23, 47, 71, ... are the slow ones)
    GraphKey loop_key{};
    int64_t stat_iters = 0, stat_events = 0;  // iterations-per-event history (first-shot sizing)

    // profiling
    bool profiling = false;
    sw_timings tm{};
    sw_counters ctr{};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // find_order state (swirld.py:53-57): ordered events are a prefix of every member's chain
    PinnedVec transactions;
    std::vector<int32_t> ord_pos;            // per member: chain positions already ordered
    std::vector<unsigned char> sig_h;        // host copy of the signatures (whitening, sort key)
    std::vector<int32_t> chain_start_h, chain_ev_h;
    DBuf<int32_t> d_q, d_acc_ev, d_acc_ri, d_sorted, d_hostflag;
    // find_order, tables of a call (order.hip.h): the rounds asked for, f_w per entry as a member-indexed row, ordered prefixes per entry,
    // [start | length | offset] of the newly ordered chain segments per (entry, member), events per entry, the block that is read back
    // ([OrderInfo][new ordered prefixes: npad][acc_off: entries + 1]), per-group tables of the bulk kernels
    DBuf<int32_t> d_ord_rounds, d_fwm, d_ordat, d_rowsum, d_grp;
    DBuf<long long> d_oblk;
    char* h_ord = nullptr;           // pinned staging of a find_order call (rounds and ordered prefixes up, the block and the host-sort flags down)
    size_t h_ord_cap = 0;
    hipStream_t stream_ord = nullptr;     // bulk find_order: the samples of group g run here beside the walk of group g + 1 ...
    hipStream_t stream_srt[2] = {nullptr, nullptr};   // ... and its sort (one workgroup per round: ~0.2 ms whatever the number of rounds) and read-back here, groups alternating
    std::vector<hipEvent_t> ord_events;   // per group: table walked / table consumed; then the ends of the three side streams
    int32_t* h_ord_stage = nullptr;  // pinned: the [i0, i1) round-entry bounds of the groups of a bulk call
    size_t h_ord_stage_cap = 0;
    DBuf<int32_t> d_big_ri, d_sk_ev;     // find_order: rounds too large for the LDS sort and their scratch keys (k_order_sort_big)
    DBuf<long long> d_big_off;
    DBuf<double> d_sk_ts;
    DBuf<u64> d_sk_k8;
    DBuf<int32_t> d_seg;
    DBuf<int32_t> d_fd;    // bulk find_order: first-descendant table of one group of round entries, [member][chain][position] (k_order_walk)
    DBuf<unsigned char> d_white;
    DBuf<double> d_ts;
    DBuf<double> d_tch;    // bulk find_order: timestamps in chain-pool order (k_order_tchain)
    int* d_err = nullptr;
    int* d_flow_err = nullptr;   // set by k_cansee_flow when a polling loop gives up (protocol bug)
    u64* d_flow_dbg = nullptr;   // SW_DEBUG_TIMING: counters of the dataflow sweep (column 0)
};

namespace {

int fail(sw_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(c, SW_EIO, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                             \
    } while (0)

#define CHK(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != SW_OK) return rc_; \
    } while (0)

// debugging aid (SW_POISON=<byte>): fresh device memory is filled with that byte, so that a read of
// memory no kernel has written shows up in any test instead of depending on what the heap held before
int poison_byte() {
    static const int v = [] { const char* e = getenv("SW_POISON"); return e && *e ? (int)(strtol(e, nullptr, 0) & 0xff) : -1; }();
    return v;
}
void poison(void* p, size_t bytes) {
    if (poison_byte() >= 0 && p && bytes) { (void)hipMemset(p, poison_byte(), bytes); (void)hipDeviceSynchronize(); }
}

// grow a device buffer to >= need elements, preserving the first `keep` elements
template <class T>
int dgrow(sw_ctx* c, DBuf<T>& b, size_t need, size_t keep) {
    if (need <= b.cap) return SW_OK;
    size_t nc = b.cap ? b.cap : 1;
    while (nc < need) nc = nc + nc / 2 + 64;
    if (b.cap == 0) nc = need;  // first allocation: exact (reserve() gives the final size)
    T* q = nullptr;
    hipError_t e = hipMalloc((void**)&q, nc * sizeof(T));
    if (e != hipSuccess) return fail(c, SW_ENOMEM, "hipMalloc(%zu bytes) failed: %s", nc * sizeof(T), hipGetErrorString(e));
    poison(q, nc * sizeof(T));
    if (b.p) {
        // the old buffer may still be in use on any of the context's streams (calls return without a
        // host synchronisation): drain the device before it is copied and freed — reallocation is rare
        HIPCHK(c, hipDeviceSynchronize());
        if (keep) HIPCHK(c, hipMemcpy(q, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice));
        (void)hipFree(b.p);
    }
    b.p = q;
    b.cap = nc;
    return SW_OK;
}

template <class T>
void dfree(DBuf<T>& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int fill_i32(sw_ctx* c, int32_t* p, size_t n, int v) {
    if (!n) return SW_OK;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks), dim3(256), 0, c->stream, p, n, v);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

// ---- windowed can_see table (VmTable) ----------------------------------------------------------
int vm_map_slot(sw_ctx* c, size_t slot) {
    VmTable& v = c->vm;
    if (v.mapped[slot]) return SW_OK;
    hipMemGenericAllocationHandle_t h;
    if (!v.pool.empty()) { h = v.pool.back(); v.pool.pop_back(); }
    else {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = c->device;
        hipError_t e = hipMemCreate(&h, v.chunk, &prop, 0);
        if (e != hipSuccess) return fail(c, SW_ENOMEM, "hipMemCreate(%zu bytes) failed: %s", v.chunk, hipGetErrorString(e));
    }
    hipError_t e = hipMemMap(v.base + slot * v.chunk, v.chunk, 0, h, 0);
    if (e != hipSuccess) { v.pool.push_back(h); return fail(c, SW_EIO, "hipMemMap failed: %s", hipGetErrorString(e)); }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIPCHK(c, hipMemSetAccess(v.base + slot * v.chunk, v.chunk, &acc, 1));
    poison(v.base + slot * v.chunk, v.chunk);
    v.handle[slot] = h;
    v.mapped[slot] = 1;
    return SW_OK;
}

// After hipMemUnmap the GPU's translation caches may still hold the old address -> chunk translation
// (measured: profiles/microbench/vmm_remap.hip; the unmap of the virtual-memory-management path does
// not invalidate them, the driver's map / unmap of an ordinary allocation does).  A recycled chunk
// or a re-used address would then be written through the stale entry.  Every batch of unmaps is
// therefore followed by one ordinary 4 MB allocation + free (above the runtime's sub-allocator
// threshold, so it reaches the driver).  SW_VM_FLUSH=0 switches this off (experiments only).
int vm_flush_translations(sw_ctx* c) {
    static const bool on = [] { const char* e = getenv("SW_VM_FLUSH"); return !(e && atoi(e) == 0); }();
    if (!on) return SW_OK;
    void* t = nullptr;
    HIPCHK(c, hipMalloc(&t, (size_t)4 << 20));
    HIPCHK(c, hipFree(t));
    return SW_OK;
}

// make the table resident up to `bytes` (from the first resident chunk)
int vm_ensure(sw_ctx* c, size_t bytes) {
    VmTable& v = c->vm;
    if (bytes > v.va_bytes) return fail(c, SW_ERANGE, "the windowed can_see table is limited to %zu GB of address space", v.va_bytes >> 30);
    const size_t need_hi = (bytes + v.chunk - 1) / v.chunk;
    for (size_t s_ = std::max(v.hi, v.lo); s_ < need_hi; ++s_) CHK(vm_map_slot(c, s_));
    v.hi = std::max(v.hi, need_hi);
    return SW_OK;
}

// unmap every chunk entirely below `bytes`; the physical chunks go to the pool
int vm_evict_below(sw_ctx* c, size_t bytes) {
    VmTable& v = c->vm;
    const size_t new_lo = std::min(bytes / v.chunk, v.hi);
    if (new_lo <= v.lo) return SW_OK;
    HIPCHK(c, hipDeviceSynchronize());  // nothing in flight may still read those rows
    for (size_t s_ = v.lo; s_ < new_lo; ++s_) {
        if (!v.mapped[s_]) continue;
        HIPCHK(c, hipMemUnmap(v.base + s_ * v.chunk, v.chunk));
        v.pool.push_back(v.handle[s_]);
        v.mapped[s_] = 0;
    }
    v.lo = new_lo;
    v.evictions++;
    return vm_flush_translations(c);
}

void vm_destroy(sw_ctx* c) {
    VmTable& v = c->vm;
    if (!v.active) return;
    for (size_t s_ = 0; s_ < v.mapped.size(); ++s_)
        if (v.mapped[s_]) { (void)hipMemUnmap(v.base + s_ * v.chunk, v.chunk); (void)hipMemRelease(v.handle[s_]); }
    for (auto h : v.pool) (void)hipMemRelease(h);
    if (v.base) (void)hipMemAddressFree(v.base, v.va_bytes);
    v = VmTable{};
    (void)vm_flush_translations(c);
}

// elements of the (non-windowed) can_see table for `cap` events: the rows, and behind the last row the scratch rows
// of the chunk-parallel sweep's halos (k_cansee_chunks: SW_MAX_CHUNKS x halo rows).  ONE place decides the size:
// ensure_events and sw_set_window(0) both allocate through it, and the chunk planner checks it.
size_t table_elems(const sw_ctx* c, int64_t cap) {
    return (size_t)(cap + (c->npad <= 256 ? SW_MAX_CHUNKS * c->halo : 0)) * c->npad;
}

// Windowed table: the halo scratch rows of the chunk-parallel sweep live behind row `cap` of the SAME address range
// (one base pointer, as in the plain table): the chunks that hold rows [cap, cap + SW_MAX_CHUNKS * halo) are mapped
// by the first call that is large enough to sweep in chunks (evictions work from the bottom and never reach them; when
// the capacity grows the old scratch chunks simply become table rows and the new ones are mapped on demand).
int vm_map_scratch(sw_ctx* c) {
    VmTable& v = c->vm;
    c->vm_scratch_ok = false;
    if (!v.active || c->npad > 256) return SW_OK;
    const size_t rowbytes = (size_t)c->npad * sizeof(int32_t);
    const size_t a = (size_t)c->cap * rowbytes, b = ((size_t)c->cap + (size_t)c->chunks * (size_t)c->halo) * rowbytes;   // (chunk k uses block k)
    if (b > v.va_bytes) return SW_OK;   // (no room in the reservation: the sweep stays unchunked)
    for (size_t s_ = a / v.chunk; s_ <= (b - 1) / v.chunk; ++s_) CHK(vm_map_slot(c, s_));
    c->vm_scratch_ok = true;
    return SW_OK;
}

int ensure_events(sw_ctx* c, int64_t need) {
    if (need <= c->cap) return SW_OK;
    int64_t nc = c->cap ? c->cap : 0;
    if (nc == 0) nc = need;
    while (nc < need) nc = nc + nc / 2 + 1024;
    const size_t keep = (size_t)c->N;
    CHK(dgrow(c, c->d_cr, nc, keep));
    CHK(dgrow(c, c->d_sp, nc, keep));
    CHK(dgrow(c, c->d_op, nc, keep));
    CHK(dgrow(c, c->d_ht, nc, keep));
    CHK(dgrow(c, c->d_seq, nc, keep));
    CHK(dgrow(c, c->d_round, nc, keep));
    CHK(dgrow(c, c->d_finlist, nc, 0));
    CHK(dgrow(c, c->d_coin, nc, keep));
    CHK(dgrow(c, c->d_t, nc, keep));
    CHK(dgrow(c, c->d_sig, (size_t)nc * 64, keep * 64));
    CHK(dgrow(c, c->d_S, (size_t)nc * c->nw, keep * c->nw));
    // (windowed table: chunks are mapped per append.)  Behind the last row: the scratch rows of the chunk-parallel
    // sweep's halos (k_cansee_chunks), SW_MAX_CHUNKS x halo rows
    if (!c->vm.active) CHK(dgrow(c, c->d_L, table_elems(c, nc), keep * c->npad));
    c->cap = nc;
    c->vm_scratch_ok = false;   // (windowed table: the scratch rows sit behind row `cap`: mapped again on demand)
    return SW_OK;
}

// rows of the per-round tables; new rows are initialised (lo = INF, wit = -1, fam = -1)
int ensure_rounds(sw_ctx* c, int need) {
    if (need <= c->Rcap) return SW_OK;
    int nc = c->Rcap ? c->Rcap : 256;
    while (nc < need) nc *= 2;
    // (the loop kernels address lo / lopos with 32-bit offsets)
    if ((int64_t)nc * c->npad >= (1ll << 31)) return fail(c, SW_ERANGE, "round table of %d rounds x %d columns exceeds 2^31 entries", nc, c->npad);
    // the per-round tables move: nothing may still be writing the old ones (the aux stream fills
    // witness rows behind the round loop)
    if (c->stream_aux) HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    if (c->stream_cs) HIPCHK(c, hipStreamSynchronize(c->stream_cs));
    const size_t np = c->npad;
    const size_t keep = (size_t)c->Rcap * np;
    CHK(dgrow(c, c->d_lo, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_lopos, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_wit, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_fam, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_dec_call, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_dec_by, (size_t)nc * np, keep));
    CHK(dgrow(c, c->d_cons, nc, c->Rcap));
    CHK(dgrow(c, c->d_newc, (size_t)nc + sizeof(FameCounters), c->d_newc.p ? sizeof(FameCounters) : 0));
    if (!c->d_fc) HIPCHK(c, hipMemset(c->d_newc.p, 0, sizeof(FameCounters)));
    c->d_fc = reinterpret_cast<FameCounters*>(c->d_newc.p);
    if ((size_t)nc + sizeof(FameCounters) > c->h_fame_cap) {
        if (c->h_fame) (void)hipHostFree(c->h_fame);
        c->h_fame = nullptr;
        c->h_fame_cap = (size_t)nc + sizeof(FameCounters);
        if (hipHostMalloc((void**)&c->h_fame, c->h_fame_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->h_fame_cap = 0; return fail(c, SW_ENOMEM, "hipHostMalloc for the fame read-back failed"); }
    }
    const size_t fresh = (size_t)(nc - c->Rcap) * np;
    CHK(fill_i32(c, c->d_lo.p + keep, fresh, SW_INF));
    CHK(fill_i32(c, c->d_lopos.p + keep, fresh, 0));
    CHK(fill_i32(c, c->d_wit.p + keep, fresh, -1));
    CHK(fill_i32(c, c->d_dec_call.p + keep, fresh, -1));
    CHK(fill_i32(c, c->d_dec_by.p + keep, fresh, -1));
    HIPCHK(c, hipMemsetAsync(c->d_fam.p + keep, 0xff, fresh, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_cons.p + c->Rcap, 0, nc - c->Rcap, c->stream));
    c->cons_h.resize(nc, 0);
    c->cons_call.resize(nc, -1);
    c->Rcap = nc;
    return SW_OK;
}

hipEvent_t next_event(sw_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

struct Span {
    hipEvent_t a = nullptr, b = nullptr;
};
Span span_begin(sw_ctx* c, hipStream_t strm = nullptr) {
    Span s;
    if (c->profiling) {
        s.a = next_event(c);
        s.b = next_event(c);
        if (s.a) (void)hipEventRecord(s.a, strm ? strm : c->stream);
    }
    return s;
}
void span_end(sw_ctx* c, Span& s, hipStream_t strm = nullptr) {
    if (c->profiling && s.b) (void)hipEventRecord(s.b, strm ? strm : c->stream);
}
float span_ms(const Span& s) {
    float ms = 0.f;
    if (s.a && s.b) (void)hipEventElapsedTime(&ms, s.a, s.b);
    return ms;
}

// Per-member self-parent chains (index order == chain order for a fork-free DAG): every member
// owns a segment of a pool, chain_ev[chain_start[m] + p] = its p-th event.  Segments have slack
// and grow geometrically (a full segment moves to the end of the pool), so appending events
// costs O(new events), not O(all events); bulk appends rebuild the pool compactly.
// Chain pool entries and chain descriptors of the events [first, first + K): one device pass over
// cr / sp / op / seq with the CURRENT segment offsets (no host loop, no pool upload).
int scatter_chains(sw_ctx* c, int64_t first, int64_t K) {
    if (K <= 0) return SW_OK;
    hipLaunchKernelGGL(k_chain_scatter, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, c->stream, (const int*)c->d_cr.p,
                       (const int*)c->d_sp.p, (const int*)c->d_op.p, (const int*)c->d_seq.p, (const int*)c->d_chain_start.p,
                       (int)first, (int)K, c->d_chain_ev.p, c->d_cdesc.p);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

// Bulk layout: every member gets a fresh segment sized from its event count `cnt` (with slack), and
// the whole pool is re-scattered on the device.  The host copy of the pool becomes stale.
int rebuild_chains(sw_ctx* c, const std::vector<int32_t>& cnt, int64_t n_events) {
    const int np = c->npad, n = c->n;
    std::vector<int32_t> start(np, 0), cap(n, 0);
    int64_t off = 0;
    for (int m = 0; m < n; ++m) {
        start[m] = (int32_t)off;
        cap[m] = cnt[m] + cnt[m] / 4 + 16;
        off += cap[m];
    }
    if (off > 0x7fffffff) return fail(c, SW_ERANGE, "chain pool exceeds 2^31 entries");
    CHK(dgrow(c, c->d_chain_ev, (size_t)off + (size_t)off / 2 + 1024, 0));
    CHK(dgrow(c, c->d_cdesc, c->d_chain_ev.cap, 0));
    c->chain_start_h.swap(start);
    c->chain_cap.swap(cap);
    c->pool_used = off;
    c->pool_h_valid = false;
    std::vector<int32_t> cntp(np, 0);
    std::copy(cnt.begin(), cnt.end(), cntp.begin());
    HIPCHK(c, hipMemcpyAsync(c->d_chain_start.p, c->chain_start_h.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_chain_cnt.p, cntp.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the staging vectors above are locals)
    return scatter_chains(c, 0, n_events);
}

// ---- lazily fetched host mirrors: the device arrays are the source of truth -------------------
// self / other parents and heights (swirld.py:117-120) of all events: needed by sw_get_height and
// by the level-bucketed can_see kernels only
int ensure_dag_h(sw_ctx* c) {
    const int64_t have = (int64_t)c->sp.size();
    if (have >= c->N) return SW_OK;
    const int64_t K = c->N - have;
    c->sp.resize(c->N); c->op.resize(c->N); c->ht.resize(c->N);
    HIPCHK(c, hipMemcpyAsync(c->sp.data() + have, c->d_sp.p + have, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->op.data() + have, c->d_op.p + have, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->blk_hmin.resize((size_t)((c->N + 4095) >> 12), 0x7fffffff);
    c->blk_hmax.resize((size_t)((c->N + 4095) >> 12), -1);
    for (int64_t e = have; e < c->N; ++e) {
        const int32_t s_ = c->sp[e], o_ = c->op[e];
        const int32_t h = s_ < 0 ? 0 : std::max(c->ht[s_], c->ht[o_]) + 1;  // swirld.py:117-120
        c->ht[e] = h;
        c->max_height = std::max(c->max_height, h);
        c->blk_hmin[e >> 12] = std::min(c->blk_hmin[e >> 12], h);
        c->blk_hmax[e >> 12] = std::max(c->blk_hmax[e >> 12], h);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ht.p + have, c->ht.data() + have, (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

// host copy of the chain pool (find_order's segment lists)
int ensure_pool_h(sw_ctx* c) {
    if (c->pool_h_valid) return SW_OK;
    c->chain_ev_h.resize((size_t)c->pool_used);
    if (c->pool_used) {
        HIPCHK(c, hipMemcpyAsync(c->chain_ev_h.data(), c->d_chain_ev.p, (size_t)c->pool_used * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->pool_h_valid = true;
    return SW_OK;
}

// host copy of the signatures (find_order's host-sort fallback, sw_get_vote's coin bits)
int ensure_sig_h(sw_ctx* c) {
    const int64_t have = (int64_t)(c->sig_h.size() / 64);
    if (have >= c->N) return SW_OK;
    if (c->payload_pending) { HIPCHK(c, hipEventSynchronize(c->ev_payload)); c->payload_pending = false; }
    c->sig_h.resize((size_t)c->N * 64);
    HIPCHK(c, hipMemcpyAsync(c->sig_h.data() + (size_t)have * 64, c->d_sig.p + (size_t)have * 64, (size_t)(c->N - have) * 64,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

// geometry of the level-bucketed can_see kernel for this member count: CB columns per workgroup (one thread per event of a
// level, the CB values of a row slice as one vector), npad threads, the deepest per-member ring that fits beside the staging ring
struct CanseeCfg { int CB, H, chs; size_t lds; };
CanseeCfg cansee_cfg(int npad, int want_H) {
    CanseeCfg g{};
    g.CB = npad <= 512 ? 2 : 4;     // >= 160 workgroups up to 512 members, 256 at 1024
    int ch = 256;
    g.chs = 8;
    while (ch < npad) { ch <<= 1; g.chs++; }
    int H = 16;
    auto bytes = [&](int h) { return (size_t)4 * ch * 16 + ((size_t)npad * h + 1 + SW_LEVEL_SIDE) * (g.CB + 1) * sizeof(int) + SW_LEVEL_SIDE * sizeof(int) + 16; };
    while (H > 1 && bytes(H) > 158u * 1024u) H >>= 1;
    if (want_H >= 1 && want_H <= H && (want_H & (want_H - 1)) == 0) H = want_H;
    g.H = H;
    g.lds = bytes(H);
    return g;
}

template <int CB>
int launch_cansee_stream(sw_ctx* c, int nlev, const CanseeCfg& g, int64_t first_event) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_cansee_stream<CB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    hipLaunchKernelGGL((k_cansee_stream<CB>), dim3(c->npad / CB), dim3(c->npad), g.lds, c->stream_cs,
                       (const int4*)c->d_desc.p, (const int*)c->d_lev_start.p, (const int*)c->d_lev_pinbase.p, nlev, c->d_L.p, c->npad, g.H, g.chs, (int)first_event);
    return SW_OK;
}

template <int NW>
int launch_cansee(sw_ctx* c, int nlev, int64_t first_event) {
    // the level-bucketed kernel (the default beyond 256 members; SW_CANSEE_IMPL = 2 / 3 selects it below)
    const CanseeCfg g = cansee_cfg(c->npad, c->ring_H_req);
    if (g.CB == 2) CHK((launch_cansee_stream<2>(c, nlev, g, first_event)));
    else CHK((launch_cansee_stream<4>(c, nlev, g, first_event)));
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

// dataflow can_see sweep of the sub-batch whose chain positions are bounds rows i and i + 1
template <int NW, int MPL, int F, int H, bool WIDE, bool DBG>
int launch_cansee_flow_t(sw_ctx* c, int i, int64_t first_event) {
    constexpr int npad = 64 * NW;
    const size_t lds = (size_t)npad * ((size_t)F * 16 + (size_t)H * 8 + 8);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_cansee_flow<NW, MPL, F, H, WIDE, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    hipLaunchKernelGGL((k_cansee_flow<NW, MPL, F, H, WIDE, DBG>), dim3(npad), dim3(npad / MPL + 64), lds, c->stream_cs,
                       (const int4*)c->d_cdesc.p, (const int*)c->d_chain_start.p,
                       (const int*)c->d_bounds.p + (size_t)i * npad, (const int*)c->d_bounds.p + (size_t)(i + 1) * npad,
                       (const int*)c->d_chain_ev.p, (int)first_event, c->d_L.p, c->d_flow_err, c->d_flow_dbg);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

// chunk-parallel dataflow sweep (k_cansee_chunks) of one plan: all chunks in ONE launch (`sweep_all`), then per
// chunk k >= k_from the repair of its provisional entries and the second sweep — both always enqueued, both gated
// on the device by the count the sweep left (nothing to read back, no host decision on the sweep stream).
// `a0_all`: rows below it are final in memory; a parent in [a0_all, w_k) is a leaf.  A sub-batch of
// sw_divide_rounds passes its first event (chunk 0 has no halo, repairs start at chunk 1); an event range of the
// multi-GPU split (sw_cansee_range) passes 0: nothing below its halo is on this device yet, and its repair
// (sw_cansee_repair, chunk 0 included) runs once the rows below it have been imported.
template <int NW, int C, int F, int H>
int launch_cansee_chunks_t(sw_ctx* c, const sw_ctx::ChunkPlan& pl, const int* bnd, unsigned* prov, unsigned* fixed,
                           int a0_all, bool sweep_all, int k_from, int k_to) {
    constexpr int npad = 64 * NW;
    constexpr int NCG = npad / C;
    const size_t lds = (size_t)npad * ((size_t)F * 16 + (size_t)(C / 2) * H * 16 + 8);
    const bool wide = ((size_t)c->cap + (size_t)SW_MAX_CHUNKS * (size_t)c->halo) * (size_t)npad * sizeof(int32_t) >= (1ull << 32);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_cansee_chunks<NW, C, F, H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        if (hipFuncSetAttribute((const void*)k_cansee_chunks<NW, C, F, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        attr_set = true;
    }
    ChunkEv ce{};
    for (int k = 0; k < pl.G; ++k) { ce.w[k] = (int)pl.w[k]; ce.a[k] = (int)pl.a[k]; }
    hipStream_t cs = c->stream_cs;
    auto sweep = [&](int grid, int exact_chunk, unsigned limit) {
        if (wide)
            hipLaunchKernelGGL((k_cansee_chunks<NW, C, F, H, true>), dim3(grid), dim3(npad + 64), lds, cs,
                               (const int4*)c->d_cdesc.p, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p, bnd, ce,
                               a0_all, exact_chunk, c->n, c->d_L.p, (int)c->cap, (int)c->halo, prov, limit, c->d_flow_err);
        else
            hipLaunchKernelGGL((k_cansee_chunks<NW, C, F, H, false>), dim3(grid), dim3(npad + 64), lds, cs,
                               (const int4*)c->d_cdesc.p, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p, bnd, ce,
                               a0_all, exact_chunk, c->n, c->d_L.p, (int)c->cap, (int)c->halo, prov, limit, c->d_flow_err);
        c->ctr.kernel_launches++;
    };
    if (sweep_all) {
        sweep(pl.G * NCG, -1, 0u);
        c->ctr.chunk_sweeps += pl.G;
    }
    for (int k = k_from; k < k_to; ++k) {
        const int64_t len = pl.a[k + 1] - pl.a[k];
        // repair by gathers up to 1/32 of the chunk's entries, a second (dependent) sweep beyond (the sweep counts
        // provisional STORES of C columns each)
        const unsigned limit = (unsigned)std::min<int64_t>((len * c->n) / (32 * C), 0x7fffffff);
        const int blocks = (int)std::min<int64_t>((len + 3) / 4, 2048);
        hipLaunchKernelGGL(k_cansee_fixup<NW>, dim3(blocks), dim3(256), 0, cs, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p,
                           bnd + (size_t)(2 * k) * npad, a0_all, (int)pl.w[k], (int)pl.a[k], (int)pl.a[k + 1], c->n, c->d_L.p,
                           (const unsigned*)(prov + k), limit, fixed);
        c->ctr.kernel_launches++;
        sweep(NCG, k, limit);
    }
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

template <int NW>
int launch_cansee_plan(sw_ctx* c, const sw_ctx::ChunkPlan& pl, const int* bnd, unsigned* prov, unsigned* fixed,
                       int a0_all, bool sweep_all, int k_from, int k_to) {
    if constexpr (NW <= 4) {
        switch (c->chunk_cfg) {
            case 1: return launch_cansee_chunks_t<NW, 2, 8, 16>(c, pl, bnd, prov, fixed, a0_all, sweep_all, k_from, k_to);
            case 2: return launch_cansee_chunks_t<NW, 4, 4, 8>(c, pl, bnd, prov, fixed, a0_all, sweep_all, k_from, k_to);
            default: return launch_cansee_chunks_t<NW, 4, 8, 8>(c, pl, bnd, prov, fixed, a0_all, sweep_all, k_from, k_to);
        }
    }
    return fail(c, SW_EINVAL, "chunked sweep: more than 256 members");
}

// the chunked sweep of sub-batch i of the running sw_divide_rounds call
template <int NW>
int launch_cansee_chunks(sw_ctx* c, int i) {
    const sw_ctx::ChunkPlan& pl = c->chunk_plan[i];
    return launch_cansee_plan<NW>(c, pl, (const int*)c->d_cbnd.p + (size_t)pl.row0 * c->npad, c->d_prov + (size_t)i * SW_MAX_CHUNKS,
                                  c->d_prov + (size_t)SW_PROV_ROWS * SW_MAX_CHUNKS + i, (int)pl.a[0], true, 1, pl.G);
}

template <int NW, int MPL, int F, int H>
int launch_cansee_flow_g(sw_ctx* c, int i, int64_t first_event) {
    // 32-bit row offsets while the can_see table stays below 4 GB
    const bool wide = (size_t)c->cap * (size_t)(64 * NW) * sizeof(int32_t) >= (1ull << 32);
    if (c->d_flow_dbg) return wide ? launch_cansee_flow_t<NW, MPL, F, H, true, true>(c, i, first_event)
                                   : launch_cansee_flow_t<NW, MPL, F, H, false, true>(c, i, first_event);
    return wide ? launch_cansee_flow_t<NW, MPL, F, H, true, false>(c, i, first_event)
                : launch_cansee_flow_t<NW, MPL, F, H, false, false>(c, i, first_event);
}

template <int NW>
int launch_cansee_flow(sw_ctx* c, int i, int64_t first_event) {
    if constexpr (NW <= 4) {
        // LDS footprint matters beyond this kernel: the round-loop kernels share the CU with it, and a
        // 133 KB workgroup leaves room for fewer of their workgroups (SW_FLOW_CFG: FIFO / ring depths)
        switch (c->flow_cfg) {
            case 0: return launch_cansee_flow_g<NW, 1, 16, 32>(c, i, first_event);
            case 2: return launch_cansee_flow_g<NW, 1, 16, 16>(c, i, first_event);
            case 3: return launch_cansee_flow_g<NW, 1, 8, 32>(c, i, first_event);
            default: return launch_cansee_flow_g<NW, 1, 8, 16>(c, i, first_event);
        }
    }
    else if constexpr (NW == 8) return launch_cansee_flow_g<8, 2, 8, 16>(c, i, first_event);   // 256 lanes x 2 chains
    else return launch_cansee_flow_g<16, 4, 4, 8>(c, i, first_event);                          // 256 lanes x 4 chains
}

LoopBufs loop_bufs(sw_ctx* c) {
    LoopBufs B;
    B.st = c->d_state;
    B.lo_r = c->d_lo_r.p; B.cur = c->d_cur.p; B.unres = c->d_unres.p; B.lo_next = c->d_lo_next.p;
    B.pos_next = c->d_pos_next.p; B.evalround = c->d_evalround.p; B.evalpos = c->d_evalpos.p;
    B.found64 = c->d_found64.p;
    B.farslot = c->d_farslot.p;
    B.force = c->d_force.p;
    B.cand = c->d_cand.p;
    B.gallop = c->d_gallop.p;
    B.front = c->d_front.p;
    B.treecnt = c->d_treecnt;
    B.dbg = c->d_dbg;
    B.dbg_minor = c->dbg_minor;
    B.dbg_blk = c->d_dbg_blk;
    return B;
}

// one iteration of the round loop = (resolve + band masks) -> tally; both kernels guard on the
// device-side state, so extra iterations after `done` are no-ops.  `par` = iteration parity
// (which half of the double-buffered loop state is read / written).
// the tables of every part of a split, as the split kernels take them
SplitDst split_dst(const sw_ctx* c) {
    SplitDst d{};
    const SplitGroup* g = c->split;
    d.part = c->split_part;
    d.parts = g->parts;
    d.ndst = g->parts;
    for (int q = 0; q < g->parts; ++q) {
        const sw_ctx* o = g->ctx[q];
        d.Mb[q] = o->d_Mb.p + o->nw;
        d.Pc[q] = o->d_Pc.p + 1;
        d.S[q] = o->d_S.p;
        d.round[q] = o->d_round.p;
        d.found64[q] = o->d_found64.p;
        d.farslot[q] = o->d_farslot.p;
    }
    return d;
}

// the device copy of a SplitDst (slot 0: the linked part's; slot q: part q of an emulated split), uploaded when it changed
const SplitDst* split_dev(sw_ctx* c, int slot, const SplitDst& d) {
    if (!c->d_split) {
        if (hipMalloc((void**)&c->d_split, sizeof(SplitDst) * SW_MAX_PARTS) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        memset((void*)c->split_up, 0xff, sizeof c->split_up);
    }
    if (memcmp(&c->split_up[slot], &d, sizeof d) != 0) {
        c->split_up[slot] = d;
        if (hipMemcpyAsync(c->d_split + slot, &c->split_up[slot], sizeof d, hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
    }
    return c->d_split + slot;
}

// One boundary of a split iteration: this part's kernel of the boundary is enqueued — record its event, tell the others, and
// make this part's stream wait for theirs (each waited for only once its owner has recorded it).  false: a part gave up.
bool split_meet(sw_ctx* c, int which, long long it) {
    SplitGroup* g = c->split;
    const int p = c->split_part, slot = (int)(it & 63);
    if (hipEventRecord(g->ev[which][p][slot], c->stream) != hipSuccess) { g->abort.store(1); return false; }
    g->rec[which][p].store(it + 1, std::memory_order_release);
    for (int q = 0; q < g->parts; ++q) {
        if (q == p) continue;
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (g->rec[which][q].load(std::memory_order_acquire) < it + 1) {
            if (g->abort.load()) return false;
            if ((++spins & 1023) == 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { g->abort.store(1); return false; }   // (a part that never entered the call)
                std::this_thread::yield();
            }
        }
        if (hipStreamWaitEvent(c->stream, g->ev[which][q][slot], 0) != hipSuccess) { g->abort.store(1); return false; }
    }
    return true;
}

// an iteration of a linked context (sw_split_link): its share of the band events, the meeting, its share of the members, the meeting
template <int NW>
void enqueue_iteration_split(sw_ctx* c, int par) {
    const int np = c->npad, K = c->K;
    const int bt = std::max(np, 256);
    const uint32_t tot2 = 2u * c->tot;
    const LoopBufs B = loop_bufs(c);
    const SplitDst sd = split_dst(c);
    const SplitDst* sdp = split_dev(c, 0, sd);
    if (!sdp) { c->split->abort.store(1); c->split_failed = true; return; }
    const long long it = c->split_iter++;
    hipLaunchKernelGGL((k_resolve_band<NW, false, true>), dim3(c->band_blocks), dim3(bt), 0, c->stream, B, par, np, K, c->gallop_after, c->skip,
                       c->NEARCAP, c->MCAP, c->Rcap, (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p,
                       (const int*)c->d_chain_ev.p, c->d_lo.p, c->d_lopos.p,
                       (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_op.p, c->d_Mb.p + NW, c->d_round.p, c->d_S.p, c->d_Pc.p + 1, sdp);
    if (!split_meet(c, 0, it)) { c->split_failed = true; return; }
    const int members_here = (np - sd.part + sd.parts - 1) / sd.parts;
    const int tally_blocks = (members_here * K + 3) / 4;
    auto tally_bits = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K, c->skip, 0,
                           (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                           (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                           (const uint32_t*)c->d_Mb.p, tot2, np, (const int*)c->d_Pc.p, sdp);
    };
    if (c->tally_filter) tally_bits(k_tally_bits<NW, true, true>);
    else tally_bits(k_tally_bits<NW, false, true>);
    if (!split_meet(c, 1, it)) { c->split_failed = true; return; }
    c->ctr.kernel_launches += 2;
}

// SW_SPLIT_EMULATE=<parts> (measurement): ONE unlinked context plays the parts of a split one behind the other — the band
// kernel of every part, then the tally kernel of every part, all storing into its own tables.  Results are those of the plain
// loop; every launch is exactly what one GPU of a `parts`-GPU node runs per iteration, alone on the chip
// (profiles/split_pieces.py reads their durations from the kernel trace).
template <int NW>
void enqueue_iteration_emulated(sw_ctx* c, int par, int parts) {
    const int np = c->npad, K = c->K;
    const int bt = std::max(np, 256);
    const uint32_t tot2 = 2u * c->tot;
    const LoopBufs B = loop_bufs(c);
    SplitDst sd{};
    sd.parts = parts; sd.ndst = 1;
    sd.Mb[0] = c->d_Mb.p + NW; sd.Pc[0] = c->d_Pc.p + 1; sd.S[0] = c->d_S.p; sd.round[0] = c->d_round.p;
    sd.found64[0] = c->d_found64.p; sd.farslot[0] = c->d_farslot.p;
    const SplitDst* sdq[SW_MAX_PARTS];
    for (int q = 0; q < parts; ++q) { sd.part = q; sdq[q] = split_dev(c, q, sd); if (!sdq[q]) { c->split_failed = true; return; } }
    for (int q = 0; q < parts; ++q) {
        hipLaunchKernelGGL((k_resolve_band<NW, false, true>), dim3(c->band_blocks), dim3(bt), 0, c->stream, B, par, np, K, c->gallop_after, c->skip,
                           c->NEARCAP, c->MCAP, c->Rcap, (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p,
                           (const int*)c->d_chain_ev.p, c->d_lo.p, c->d_lopos.p,
                           (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_op.p, c->d_Mb.p + NW, c->d_round.p, c->d_S.p, c->d_Pc.p + 1, sdq[q]);
    }
    for (int q = 0; q < parts; ++q) {
        const int members_here = (np - q + parts - 1) / parts;
        const int tally_blocks = (members_here * K + 3) / 4;
        if (c->tally_filter)
            hipLaunchKernelGGL((k_tally_bits<NW, true, true>), dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K, c->skip, 0,
                               (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                               (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                               (const uint32_t*)c->d_Mb.p, tot2, np, (const int*)c->d_Pc.p, sdq[q]);
        else
            hipLaunchKernelGGL((k_tally_bits<NW, false, true>), dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K, c->skip, 0,
                               (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                               (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                               (const uint32_t*)c->d_Mb.p, tot2, np, (const int*)c->d_Pc.p, sdq[q]);
    }
    c->ctr.kernel_launches += 2 * parts;
}

template <int NW>
void enqueue_iteration(sw_ctx* c, int par, std::vector<Span>* tally_spans, std::vector<Span>* resolve_spans = nullptr) {
    if (c->split) { enqueue_iteration_split<NW>(c, par); return; }
    if (c->split_emulate > 0 && c->unit_stake && c->tally_impl == 1) { enqueue_iteration_emulated<NW>(c, par, c->split_emulate); return; }
    const int np = c->npad, K = c->K;
    const int tally_blocks = np * K / 4;
    const int bt = std::max(np, 256);
    const int band_blocks = c->band_blocks;
    const uint32_t tot2 = 2u * c->tot;
    const LoopBufs B = loop_bufs(c);
    Span sr{};
    if (resolve_spans) sr = span_begin(c);
    auto resolve_band = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(band_blocks), dim3(bt), 0, c->stream, B, par, np, K, c->gallop_after, c->skip,
                           c->NEARCAP, c->MCAP, c->Rcap, (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p,
                           (const int*)c->d_chain_ev.p, c->d_lo.p, c->d_lopos.p,
                           (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_op.p, c->d_Mb.p + NW, c->d_round.p, c->d_S.p, c->d_Pc.p + 1, (const SplitDst*)nullptr);
    };
    if constexpr (NW <= 4) {
        if (c->band_fast) resolve_band(k_resolve_band<NW, true>);
        else resolve_band(k_resolve_band<NW, false>);
    } else resolve_band(k_resolve_band<NW, false>);
    if (resolve_spans) { span_end(c, sr); resolve_spans->push_back(sr); }
    Span s{};
    if (tally_spans) s = span_begin(c);
    if (c->unit_stake && c->tally_impl == 2)
        hipLaunchKernelGGL(k_tally_tree<NW>, dim3(np), dim3(512), 0, c->stream, B, par, K, c->skip, c->tally_pf,
                           (const int*)c->d_L.p, (const int*)c->d_sp.p, (const int*)c->d_op.p, (const uint32_t*)c->d_Mb.p, tot2, np);
    else if (c->unit_stake && c->tally_impl >= 1) {
        auto tally_bits = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K, c->skip, c->tally_pf,
                               (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                               (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                               (const uint32_t*)c->d_Mb.p, tot2, np, (const int*)c->d_Pc.p, (const SplitDst*)nullptr);
        };
        if (c->tally_filter) tally_bits(k_tally_bits<NW, true>);
        else tally_bits(k_tally_bits<NW, false>);
    }
    else if (c->unit_stake)
        hipLaunchKernelGGL((k_tally_candidates<NW, true>), dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K,
                           (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                           (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                           (const u64*)(c->d_Mb.p + NW), (const uint32_t*)c->d_stake.p, tot2, np);
    else
        hipLaunchKernelGGL((k_tally_candidates<NW, false>), dim3(tally_blocks), dim3(256), 0, c->stream, B, par, K,
                           (const int*)c->d_chain_start.p, (const int*)c->d_chain_len.p, (const int*)c->d_chain_ev.p,
                           (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_sp.p, (const int*)c->d_op.p,
                           (const u64*)(c->d_Mb.p + NW), (const uint32_t*)c->d_stake.p, tot2, np);
    if (tally_spans) { span_end(c, s); tally_spans->push_back(s); }
    c->ctr.kernel_launches += 2;
}

// the loop body as replayable hipGraphs of [SW_GRAPH_BIG /] 24 / 8 / 2 iterations (2 kernel nodes each); the
// kernel arguments are frozen at capture, so the graphs are rebuilt when any of them changes.
// Iteration counts are even because the loop state is double-buffered by iteration parity.
constexpr int kGraphSizesBase[4] = {0, 24, 8, 2};

template <int NW>
int launch_iterations(sw_ctx* c, int n_iters, std::vector<Span>* tally_spans, std::vector<Span>* resolve_spans = nullptr) {
    // short shots (a Node's small calls): plain launches — the first hipGraphLaunch of a call costs ~200 us
    if (!c->use_graph || tally_spans || n_iters <= 8 || c->split || c->split_emulate) {   // (split: the meetings of the parts are host calls between the kernels)
        for (int it = 0; it < n_iters; ++it) enqueue_iteration<NW>(c, it & 1, tally_spans, resolve_spans);
        return SW_OK;
    }
    sw_ctx::GraphKey key;
    memset(&key, 0, sizeof key);
    key.Rcap = c->Rcap; key.N = c->N; key.lo = (void*)c->d_lo.p; key.L = (void*)c->d_L.p;
    key.chain = (void*)c->d_chain_ev.p; key.K = c->K; key.tally_impl = c->tally_impl;
    key.S = (void*)c->d_S.p; key.round = (void*)c->d_round.p;
    key.BATCH = c->band_blocks + 4096 * c->skip; key.MCAP = c->MCAP + 7 * c->NEARCAP; key.variant = c->band_fast + 2 * c->tally_filter;
    if (memcmp(&key, &c->loop_key, sizeof key) != 0) {
        for (int g = 0; g < 4; ++g) {
            if (c->loop_exec[g]) { (void)hipGraphExecDestroy(c->loop_exec[g]); c->loop_exec[g] = nullptr; }
            if (c->loop_graph[g]) { (void)hipGraphDestroy(c->loop_graph[g]); c->loop_graph[g] = nullptr; }
        }
        c->loop_key = key;
    }
    int left = n_iters;
    const int kGraphSizes[4] = {c->graph_big, kGraphSizesBase[1], kGraphSizesBase[2], kGraphSizesBase[3]};
    for (int g = 0; g < 4; ++g) {
        if (kGraphSizes[g] <= 0) continue;
        while (left >= kGraphSizes[g]) {
            if (!c->loop_exec[g]) {
                HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int64_t launches_before = c->ctr.kernel_launches;
                for (int it = 0; it < kGraphSizes[g]; ++it) enqueue_iteration<NW>(c, it & 1, nullptr);
                c->ctr.kernel_launches = launches_before;
                HIPCHK(c, hipStreamEndCapture(c->stream, &c->loop_graph[g]));
                HIPCHK(c, hipGraphInstantiate(&c->loop_exec[g], c->loop_graph[g], nullptr, nullptr, 0));
            }
            HIPCHK(c, hipGraphLaunch(c->loop_exec[g], c->stream));
            c->ctr.kernel_launches += 2 * kGraphSizes[g];
            left -= kGraphSizes[g];
        }
    }
    return SW_OK;
}

// round numbers and sees-masks of the events [first, first + K) on `ax`, table rows 0 .. R-1 final: with the band pass
// writing them (SW_FIN_BAND=1) a check of one thread per event + the listed leftovers from their rows, else every event
// from its row.  `blocks` = workgroups of the row-reading launch.
template <int NW>
int launch_finalize(sw_ctx* c, hipStream_t ax, int64_t first, int64_t K, int R, int blocks) {
    const int np = c->npad;
    if (K <= 0) return SW_OK;
    if (!c->fin_band) {
        hipLaunchKernelGGL(k_finalize_events<NW>, dim3((unsigned)std::min<int64_t>((K + 3) / 4, blocks)), dim3(256), 0, ax, (const int*)c->d_L.p,
                           (const int*)c->d_cr.p, (const int*)c->d_lo.p, R, (int)first, (int)K, c->d_round.p, c->d_S.p, np);
        c->ctr.kernel_launches++;
        return SW_OK;
    }
    if (!c->d_fin) {
        HIPCHK(c, hipMalloc((void**)&c->d_fin, 16));
        HIPCHK(c, hipMemset(c->d_fin, 0, 16));
    }
    HIPCHK(c, hipMemsetAsync(c->d_fin, 0, sizeof(unsigned), ax));
    hipLaunchKernelGGL(k_finalize_check, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ax, (const int*)c->d_cr.p, (const int*)c->d_lo.p, R,
                       (int)first, (int)K, (const int*)c->d_round.p, np, c->d_finlist.p, c->d_fin);
    hipLaunchKernelGGL(k_finalize_listed<NW>, dim3((unsigned)std::min<int64_t>((K + 3) / 4, std::min(blocks, 1024))), dim3(256), 0, ax, (const int*)c->d_L.p,
                       (const int*)c->d_cr.p, (const int*)c->d_lo.p, R, (const int*)c->d_finlist.p, c->d_fin, c->d_round.p, c->d_S.p, np);
    c->ctr.kernel_launches += 2;
    return SW_OK;
}

// first shot of a round loop: the predicted number of iterations for this many events (from the iterations-per-event rate of
// earlier runs); without a measured rate about one round per 11.7 n events (SURVEY.md §8 probe) plus the round in progress —
// a small call must not pay for a long first shot
inline int predict_shot(const sw_ctx* c, int64_t n_new_events) {
    int shot = std::min<int64_t>(c->BATCH, 2 + (n_new_events / (12 * (int64_t)c->n) + 1) * 2);
    if (c->stat_iters > 0 && c->stat_events > 0) {
        const double pred = (double)c->stat_iters / (double)c->stat_events * (double)n_new_events;
        shot = std::max(2, (int)(pred * c->shot_pct / 100.0) + c->shot_extra);
    }
    return std::min(shot, 4096) & ~1;
}

template <int NW>
int run_round_loop(sw_ctx* c, int r_start, int64_t limit, int64_t n_new_events, const int32_t* visible_len, float* tally_ms_out, int* tally_launches_out,
                   const std::function<int()>* after_first_shot = nullptr, const std::function<int(const RState&)>* mid_loop = nullptr,
                   int64_t fin_from = 0x7fffffff) {
    const int np = c->npad, K = c->K;
    hipLaunchKernelGGL(k_loop_init, dim3(1), dim3(std::min(2 * np, 1024)), 0, c->stream, loop_bufs(c), np, r_start,
                       (int)limit, c->NEARCAP, (const int*)visible_len, c->d_chain_len.p, c->eval_src,
                       (int)std::min<int64_t>(fin_from, 0x7fffffff), (int)std::min<int64_t>(c->ctr.round_iterations - c->dbg_iter_base, 0x7fffffff));
    c->eval_src = 0;
    c->ctr.kernel_launches++;
    std::vector<Span> tally_spans, resolve_spans;
    RState st{};
    int launched = 0;
    // first shot: the predicted number of iterations (predict_shot), then short top-ups until the loop reports done
    int shot = predict_shot(c, n_new_events);
    // `mid_loop` (the last sub-batch of a large call): the first shot stops `mid_pct` % of the way, the host looks at the loop
    // state once — every event below the band of the round in progress has its final round by then — and hands it to the
    // caller, which finalizes those events beside the rest of the loop instead of behind it; the rest of the prediction follows
    int rest = 0;
    if (mid_loop && shot >= 64 && c->mid_pct > 0) {
        const int head = std::max(2, (int)((int64_t)shot * c->mid_pct / 100) & ~1);
        rest = std::max(2, (shot - head) & ~1);
        shot = head;
    }
    for (;;) {
        CHK(ensure_rounds(c, c->R + launched + shot + 4));
        CHK(launch_iterations<NW>(c, shot, c->profiling && !c->split ? &tally_spans : nullptr, c->profiling && !c->split ? &resolve_spans : nullptr));
        if (c->split_failed) { c->poisoned = true; return fail(c, SW_EIO, "split round loop: a linked context did not arrive at iteration %lld (sw_split_link: every part calls sw_divide_rounds)", c->split_iter); }
        // host work that is off the critical path (the finalize / witness / voter-mask launches of the PREVIOUS sub-batch, on
        // their own stream) goes here: the GPU is already busy with this sub-batch's first shot
        if (launched == 0 && after_first_shot) CHK((*after_first_shot)());
        launched += shot;
        HIPCHK(c, hipGetLastError());
        // loop state, sweep error flag (the sweep of this sub-batch is complete) and the members' front
        // rounds: ONE copy into pinned memory
        HIPCHK(c, hipMemcpyAsync(c->h_rb, c->d_rb, c->rb_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(&st, c->h_rb, sizeof st);
        int ferr = 0;
        memcpy(&ferr, c->h_rb + 2 * sizeof(RState), sizeof ferr);
        memcpy(c->front_dev.data(), c->h_rb + 256, np * sizeof(int32_t));
        if (ferr) return fail(c, SW_EIO, "can_see sweep gave up polling (code %d): internal protocol error", ferr);
        if (st.err) return fail(c, SW_ERANGE, "round table capacity exceeded (internal)");
        if (st.done) break;
        // rounds <= DAG height + 1, retries <= N / K: anything beyond that is a bug, not work
        if ((int64_t)launched > (int64_t)c->max_height + 2 + c->N / K + 4096)
            return fail(c, SW_EIO, "round loop did not terminate after %d iterations (r=%d)", launched, st.r);
        if (rest) {
            CHK((*mid_loop)(st));
            shot = rest;
            rest = 0;
        } else
        shot = launched < 8 ? 2 : (launched < 48 ? 8 : 4);
    }
    if (n_new_events >= 4096) {  // keep the rate estimate to runs where it means something
        c->stat_iters += st.iter;
        c->stat_events += n_new_events;
    }
    // the per-member exhaustion marks persist across runs; the next run reads half 0: its k_loop_init moves them there
    // (two small device copies here sat in the gap between two sub-batches' loops)
    c->eval_src = st.iter & 1;
    c->R = st.max_round + 1;
    if (c->unit_stake && c->tally_impl == 2) {   // the tree search counts the tallies it really evaluated (per member, read back with the state)
        const int32_t* tc = reinterpret_cast<const int32_t*>(c->h_rb + ((unsigned char*)c->d_treecnt - c->d_rb));
        for (int m = 0; m < np; ++m) c->ctr.tally_evals += tc[m];
    } else
    c->ctr.tally_evals += (int64_t)st.evals;
    c->ctr.far_hops += (int64_t)st.far_hops;
    c->ctr.round_iterations += st.iter;
    if (c->profiling) {
        float ms = 0.f;
        int cnt = 0;
        // only the launches that did work (iterations before `done`)
        for (size_t i = 0; i < tally_spans.size() && (int)i < st.iter - 1; ++i) { ms += span_ms(tally_spans[i]); ++cnt; }
        *tally_ms_out += ms;
        *tally_launches_out += cnt;
        for (size_t i = 0; i < resolve_spans.size() && (int)i < st.iter; ++i) { c->tm.resolve_ms += span_ms(resolve_spans[i]); c->tm.resolve_launches++; }
    }
    c->ctr.band_events += (int64_t)st.band_events;
    return SW_OK;
}


template <int NW>
int launch_voter_masks(sw_ctx* c, int r0, int R, hipStream_t strm) {
    const int np = c->npad;
    if (R > c->Sw_rows) {
        int nr = c->Sw_rows ? c->Sw_rows : 256;
        while (nr < R) nr *= 2;
        u64* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, (size_t)nr * np * NW * sizeof(u64));
        if (e != hipSuccess) return fail(c, SW_ENOMEM, "hipMalloc(Sw) failed: %s", hipGetErrorString(e));
        poison(q, (size_t)nr * np * NW * sizeof(u64));
        HIPCHK(c, hipStreamSynchronize(c->stream_aux));
        if (c->Sw_rows && c->d_Sw.p)
            HIPCHK(c, hipMemcpy(q, c->d_Sw.p, (size_t)c->Sw_rows * np * NW * sizeof(u64), hipMemcpyDeviceToDevice));
        if (c->d_Sw.p) (void)hipFree(c->d_Sw.p);
        c->d_Sw.p = q;
        c->d_Sw.cap = (size_t)nr * np * NW;
        c->Sw_rows = nr;
    }
    r0 = std::max(1, r0);
    if (R <= r0) return SW_OK;
    const uint32_t tot2 = 2u * c->tot;
    const int total = (R - r0) * np;
    if (c->unit_stake && c->tally_impl >= 1)
        hipLaunchKernelGGL(k_voter_masks_bits<NW>, dim3((total + 3) / 4), dim3(256), 0, strm,
                           (const int*)c->d_wit.p, (const int*)c->d_L.p, (const int*)c->d_lo.p, (const uint32_t*)c->d_S.p,
                           tot2, r0, R, np, (uint32_t*)c->d_Sw.p, c->d_fc);
    else if (c->unit_stake)
        hipLaunchKernelGGL((k_voter_masks<NW, true>), dim3((total + 3) / 4), dim3(256), 0, strm,
                           (const int*)c->d_wit.p, (const int*)c->d_L.p, (const int*)c->d_lo.p, (const u64*)c->d_S.p,
                           (const uint32_t*)c->d_stake.p, tot2, r0, R, np, c->d_Sw.p, c->d_fc);
    else
        hipLaunchKernelGGL((k_voter_masks<NW, false>), dim3((total + 3) / 4), dim3(256), 0, strm,
                           (const int*)c->d_wit.p, (const int*)c->d_L.p, (const int*)c->d_lo.p, (const u64*)c->d_S.p,
                           (const uint32_t*)c->d_stake.p, tot2, r0, R, np, c->d_Sw.p, c->d_fc);
    c->ctr.kernel_launches++;
    return SW_OK;
}

// height span of the events [a, b): from the ingest-time block index, edges by scanning
void height_span(const sw_ctx* c, int64_t a, int64_t b, int* hmin, int* hmax) {
    int lo = 0x7fffffff, hi = -1;
    int64_t e = a;
    while (e < b) {
        if ((e & 4095) == 0 && e + 4096 <= b) {
            lo = std::min(lo, c->blk_hmin[e >> 12]);
            hi = std::max(hi, c->blk_hmax[e >> 12]);
            e += 4096;
        } else {
            lo = std::min(lo, c->ht[e]);
            hi = std::max(hi, c->ht[e]);
            ++e;
        }
    }
    *hmin = lo;
    *hmax = hi;
}

// host-side stage clock of sw_divide_rounds (SW_DEBUG_TIMING=1: averages printed by sw_destroy)
struct StageClock {
    bool on;
    std::chrono::steady_clock::time_point last;
    explicit StageClock(bool on_) : on(on_) { if (on) last = std::chrono::steady_clock::now(); }
    void mark(double* acc) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        *acc += std::chrono::duration<double, std::micro>(now - last).count();
        last = now;
    }
};

template <int NW>
int do_divide(sw_ctx* c, int64_t first, int64_t K) {
    const int np = c->npad, n = c->n;
    c->ev_used = 0;
    StageClock clk(c->debug_timing);
    c->stage_calls += 1;
    Span sp_total = span_begin(c);
    // ---- sub-batches: the can_see sweep of sub-batch i+1 (stream_cs) overlaps the round loop
    // of sub-batch i (main stream); a kernel boundary separates producer and consumer of a row
    std::vector<int64_t> cut{first};
    if (const char* cs_ = getenv("SW_CUTS"); cs_ && K >= 65536) {  // tuning hook: cut points as fractions of K
        for (const char* q = cs_; *q;) {
            char* end = nullptr;
            const double f = strtod(q, &end);
            if (end == q) break;
            const int64_t bnd = ((first + (int64_t)(f * (double)K)) >> 12) << 12;
            if (bnd > cut.back() && bnd < first + K) cut.push_back(bnd);
            q = *end ? end + 1 : end;
        }
    } else if (K >= 65536 && c->pipe > 1) {
        // a short first sub-batch (its sweep is the only one nothing overlaps), then `pipe` parts.  Round 5: the parts are
        // GRADUATED — weights 0.5, 0.9, 1.2, 1, 1, ... — because a loop now consumes events almost as fast as a sweep produces
        // them (the loop of an even second part waited 0.17 ms for its sweep; profiles/r05l_knobs_256x1M.log: 6.09 -> 5.99 ms)
        const int64_t head = K / 16;
        const int P = c->pipe;
        double w[SW_PROV_ROWS], tot = 0.0;
        for (int s_ = 0; s_ < P; ++s_) { w[s_] = P < 4 ? 1.0 : (s_ == 0 ? 0.5 : s_ == 1 ? 0.9 : s_ == 2 ? 1.2 : 1.0); tot += w[s_]; }
        double acc = 0.0;
        for (int s_ = 0; s_ < P; ++s_) {
            const int64_t at = head + (int64_t)((double)(K - head) * (acc / tot));
            acc += w[s_];
            const int64_t bnd = ((first + at) >> 12) << 12;
            if (bnd > cut.back() && bnd < first + K) cut.push_back(bnd);
        }
    }
    cut.push_back(first + K);
    const int S = (int)cut.size() - 1;
    const bool flow = c->cansee_impl >= 6;
    // Which tally (round 4, profiles/r04l_*): the two-level search with a window of 32 wins on 256-member hashgraphs whose
    // members are about equally active — uniform gossip +2.6 %, two cliques +18 %, stale other-parents +16 %, 4 M events
    // +4.7 %, mild skew +5 % — and loses where a third of the members is 50 times less active (-9 %) and at 64 / 128
    // members (-8 / -10 %).  Decided per large call from the members' event counts; SW_TALLY_IMPL / SW_TALLY_K pin it.
    if (c->tally_auto && c->unit_stake) {
        int impl = 1;
        if (np == 256 && n > 200 && K >= 65536) {
            int32_t lo_ = 0x7fffffff, hi_ = 0;
            for (int m = 0; m < n; ++m) { lo_ = std::min(lo_, c->nev[m]); hi_ = std::max(hi_, c->nev[m]); }
            if (lo_ > 0 && (int64_t)hi_ <= 8 * (int64_t)lo_) impl = 2;
        }
        c->tally_impl = impl;
        if (c->K_auto) c->K = impl == 2 ? 32 : c->K_flat;
    }
    // rows already in the table (event-range split: swept by sw_cansee_range or imported): no sweep, the round
    // loop still waits for whatever the sweep stream has in flight (imports, repairs)
    bool preswept = false;
    for (const auto& pr : c->present) {
        if (first >= pr.first && first + K <= pr.second) preswept = true;
        else if (first < pr.second && first + K > pr.first)
            return fail(c, SW_EINVAL, "divide_rounds [%lld, %lld) straddles the rows [%lld, %lld) already present (sw_cansee_range / sw_import_rows)",
                        (long long)first, (long long)(first + K), (long long)pr.first, (long long)pr.second);
    }
    std::vector<int> hmins(S), nlevs(S);
    int max_nlev = 1;
    int64_t max_k = 1;
    if (!flow && !preswept) {
        CHK(ensure_dag_h(c));  // the level-bucketed kernels need the heights (swirld.py:117-120)
        for (int i = 0; i < S; ++i) {
            int hmax;
            height_span(c, cut[i], cut[i + 1], &hmins[i], &hmax);
            nlevs[i] = hmax - hmins[i] + 1;
            max_nlev = std::max(max_nlev, nlevs[i]);
            max_k = std::max(max_k, cut[i + 1] - cut[i]);
            c->ctr.levels += nlevs[i];
        }
        CHK(dgrow(c, c->d_lev_cnt, max_nlev, 0));
        CHK(dgrow(c, c->d_lev_start, max_nlev + 1, 0));
        CHK(dgrow(c, c->d_lev_cursor, max_nlev, 0));
        CHK(dgrow(c, c->d_desc, max_k, 0));
        CHK(dgrow(c, c->d_lev_pin, max_k, 0));
        CHK(dgrow(c, c->d_lev_pos, max_k, 0));
        CHK(dgrow(c, c->d_lev_cback, max_nlev, 0));
        CHK(dgrow(c, c->d_lev_pinbase, max_nlev + 1, 0));
    }
    while ((int)c->cs_events.size() < S) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->cs_events.push_back(e);
    }
    // ---- enqueue every can_see sweep on its own stream (behind whatever the main stream still has in
    // flight: a small append returns without a host synchronisation)
    hipStream_t cs = c->stream_cs;
    HIPCHK(c, hipEventRecord(c->ev_main_mark, c->stream));
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_main_mark, 0));
    hipEvent_t cs_t0 = nullptr, cs_t1 = nullptr;
    if (c->profiling) { cs_t0 = next_event(c); cs_t1 = next_event(c); (void)hipEventRecord(cs_t0, cs); }
    std::vector<Span> cansee_spans;
    if (c->profiling) { c->tm.resolve_ms = 0.f; c->tm.resolve_launches = 0; }
    // chain positions of the cuts: bounds[i][m] = events of member m below cut[i].  One cut pair that
    // ends at the last appended event (a Node's call) is known on the host: chain lengths at the last
    // divide and now.  Otherwise device binary searches over the chain pool.
    std::vector<int32_t>& bounds_h = c->bounds_stage;
    bool bounds_pending = false;
    bounds_h.resize((size_t)(S + 1) * np);
    CHK(dgrow(c, c->d_bounds, (size_t)(S + 1) * np, 0));
    if (S == 1 && first + K == c->N) {
        std::copy(c->divided_cnt.begin(), c->divided_cnt.end(), bounds_h.begin());
        std::fill(bounds_h.begin() + np, bounds_h.end(), 0);
        std::copy(c->nev.begin(), c->nev.end(), bounds_h.begin() + np);
        HIPCHK(c, hipMemcpyAsync(c->d_bounds.p, bounds_h.data(), bounds_h.size() * sizeof(int32_t), hipMemcpyHostToDevice, cs));
    } else {
        CHK(dgrow(c, c->d_cuts, S + 1, 0));
        std::vector<long long> cuts_ll(cut.begin(), cut.end());
        HIPCHK(c, hipMemcpyAsync(c->d_cuts.p, cuts_ll.data(), (S + 1) * sizeof(long long), hipMemcpyHostToDevice, cs));
        hipLaunchKernelGGL(k_chain_bounds, dim3(S + 1), dim3(np), 0, cs, (const int*)c->d_chain_start.p, (const int*)c->d_chain_cnt.p,
                           (const int*)c->d_chain_ev.p, (const long long*)c->d_cuts.p, np, c->d_bounds.p);
        c->ctr.kernel_launches++;
        // (read back behind the kernel; the host needs the table only for the round loops below: it waits for
        // `ev_bounds` there, after every sweep has been enqueued — the GPU starts sweeping ~0.1 ms earlier)
        HIPCHK(c, hipMemcpyAsync(bounds_h.data(), c->d_bounds.p, bounds_h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, cs));
        if (!c->ev_bounds) HIPCHK(c, hipEventCreateWithFlags(&c->ev_bounds, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_bounds, cs));
        bounds_pending = true;
    }
    // ---- chunk plan: a sub-batch long enough is cut into G chunks that are swept concurrently, each from
    // `halo` events before its start (k_cansee_chunks); their chain positions come from one more search kernel
    c->chunk_plan.assign(S, sw_ctx::ChunkPlan{});
    bool may_chunk = flow && !preswept && np <= 256 && c->chunks > 1 && !c->chunks_off && S <= SW_PROV_ROWS;
    if (c->debug_timing && S > SW_PROV_ROWS)   // (only the SW_CUTS tuning hook can ask for that many sub-batches)
        fprintf(stderr, "[sw] %d sub-batches: more than %d, swept unchunked\n", S, SW_PROV_ROWS);
    if (may_chunk && c->vm.active && !c->vm_scratch_ok) {   // windowed table: the scratch rows are mapped by the first call that can use them
        bool any = false;
        for (int i = 0; i < S; ++i) any = any || (cut[i + 1] - cut[i]) / c->chunk_min >= 2;
        if (any) CHK(vm_map_scratch(c));
    }
    if (may_chunk && (c->vm.active ? c->vm_scratch_ok : c->d_L.cap >= table_elems(c, c->cap))) {   // (the halo scratch rows exist: they live behind the table's last row)
        std::vector<long long>& ccuts = c->ccuts_stage;
        ccuts.clear();
        int max_g = 0;
        for (int i = 0; i < S; ++i) {
            const int64_t a = cut[i], len = cut[i + 1] - cut[i];
            const int G = (int)std::min<int64_t>(c->chunks, len / c->chunk_min);
            if (G < 2) continue;
            sw_ctx::ChunkPlan& pl = c->chunk_plan[i];
            pl.G = G;
            pl.row0 = (int)ccuts.size();
            for (int k = 0; k <= G; ++k) pl.a[k] = a + len * k / G;
            for (int k = 0; k < G; ++k) {
                pl.w[k] = k == 0 ? a : std::max(a, pl.a[k] - c->halo);
                ccuts.push_back(pl.w[k]);
                ccuts.push_back(pl.a[k]);
            }
            ccuts.push_back(pl.a[G]);   // rows 2G and 2G + 1: the end, twice — chunk k ends at row 2k + 3 for every k
            ccuts.push_back(pl.a[G]);
            max_g = std::max(max_g, G);
        }
        if (!ccuts.empty()) {
            CHK(dgrow(c, c->d_ccuts, ccuts.size(), 0));
            CHK(dgrow(c, c->d_cbnd, ccuts.size() * (size_t)np, 0));
            // (pageable source: the copy is staged before the call returns)
            HIPCHK(c, hipMemcpyAsync(c->d_ccuts.p, ccuts.data(), ccuts.size() * sizeof(long long), hipMemcpyHostToDevice, cs));
            hipLaunchKernelGGL(k_chain_bounds, dim3((unsigned)ccuts.size()), dim3(np), 0, cs, (const int*)c->d_chain_start.p,
                               (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, (const long long*)c->d_ccuts.p, np, c->d_cbnd.p);
            HIPCHK(c, hipMemsetAsync(c->d_prov, 0, (size_t)SW_PROV_ROWS * (SW_MAX_CHUNKS + 1) * sizeof(unsigned), cs));
            c->ctr.kernel_launches++;
        }
    }
    for (int i = 0; i < S; ++i) {
        const int64_t a = cut[i], k = cut[i + 1] - cut[i];
        Span scs{};
        if (preswept) {
            scs = span_begin(c, cs);
        } else if (flow && c->chunk_plan[i].G >= 2) {
            scs = span_begin(c, cs);
            CHK(launch_cansee_chunks<NW>(c, i));
        } else if (flow) {
            scs = span_begin(c, cs);
            CHK(launch_cansee_flow<NW>(c, i, a));
        } else {
            HIPCHK(c, hipMemsetAsync(c->d_lev_cnt.p, 0, nlevs[i] * sizeof(int32_t), cs));
            HIPCHK(c, hipMemsetAsync(c->d_lev_pin.p, 0, (size_t)k, cs));
            const int eb = (int)((k + 255) / 256);
            const int ringH = cansee_cfg(np, c->ring_H_req).H;
            hipLaunchKernelGGL(k_level_hist, dim3(eb), dim3(256), 0, cs, (const int*)c->d_ht.p, (int)a, (int)k, hmins[i], c->d_lev_cnt.p,
                               (const int*)c->d_cr.p, (const int*)c->d_op.p, (const int*)c->d_seq.p, (const int*)c->d_chain_start.p,
                               (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, ringH, c->d_lev_pin.p);
            hipLaunchKernelGGL(k_level_scan, dim3(1), dim3(1024), 0, cs, (const int*)c->d_lev_cnt.p, nlevs[i], c->d_lev_start.p, c->d_lev_cursor.p);
            HIPCHK(c, hipMemsetAsync(c->d_lev_cback.p, 0, nlevs[i] * sizeof(int32_t), cs));
            hipLaunchKernelGGL(k_level_scatter, dim3(eb), dim3(256), 0, cs, (const int*)c->d_ht.p, (const int*)c->d_cr.p,
                               (const int*)c->d_sp.p, (const int*)c->d_op.p, (const int*)c->d_seq.p, (int)a, (int)k, hmins[i],
                               (const int*)c->d_lev_start.p, c->d_lev_cursor.p, c->d_lev_cback.p, c->d_desc.p, ringH, np,
                               (const unsigned char*)c->d_lev_pin.p, c->d_lev_pos.p);
            // pinned events in front of every level (the front cursors' exclusive scan), then the children's descriptors
            hipLaunchKernelGGL(k_level_scan, dim3(1), dim3(1024), 0, cs, (const int*)c->d_lev_cursor.p, nlevs[i], c->d_lev_pinbase.p, c->d_lev_cback.p);
            hipLaunchKernelGGL(k_level_patch, dim3(eb), dim3(256), 0, cs, (const int*)c->d_ht.p, (const int*)c->d_cr.p, (const int*)c->d_op.p,
                               (const int*)c->d_seq.p, (int)a, (int)k, hmins[i], (const int*)c->d_lev_start.p, (const int*)c->d_lev_pinbase.p,
                               c->d_desc.p, ringH, np, (const int*)c->d_chain_start.p, (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p,
                               (const int*)c->d_lev_pos.p);
            c->ctr.kernel_launches += 2;
            c->ctr.kernel_launches += 3;
            scs = span_begin(c, cs);
            CHK(launch_cansee<NW>(c, nlevs[i], a));
        }
        span_end(c, scs, cs);
        if (c->profiling) cansee_spans.push_back(scs);
        HIPCHK(c, hipEventRecord(c->cs_events[i], cs));
    }
    if (c->profiling) (void)hipEventRecord(cs_t1, cs);
    clk.mark(&c->stage_us[0]);

    if (bounds_pending) HIPCHK(c, hipEventSynchronize(c->ev_bounds));
    // ---- round loops, one per sub-batch, each over the events visible so far
    CHK(ensure_rounds(c, std::max(c->R, 1) + c->BATCH + 4));
    Span sp_rl = span_begin(c);
    hipEvent_t fin_t0 = nullptr;
    float tally_ms = 0.f;
    int tally_launches = 0;
    // members' visible chain lengths before this call and after every sub-batch: rows of the cut table
    std::vector<int32_t> clen_prev(bounds_h.begin(), bounds_h.begin() + np), clen(np, 0);
    bool aux_armed = false;
    std::function<int()> pending_aux = []() -> int { return SW_OK; };
    // start round of sub-batch i's loop = the smallest front round among the members it adds events to; a member's first event
    // (swirld.py:195-198) opens round 0 for it: lo[0][m], chain position 0, uploaded in front of the loop
    auto start_round = [&](int i, bool* row0_dirty) -> int {
        int r_start = 0x7fffffff;
        *row0_dirty = false;
        std::copy(bounds_h.begin() + (size_t)(i + 1) * np, bounds_h.begin() + (size_t)(i + 2) * np, clen.begin());
        for (int m = 0; m < n; ++m) {
            if (clen[m] > clen_prev[m]) {  // member touched by this sub-batch
                if (c->front[m] < 0) {     // its root (swirld.py:195-198): lo[0][m], chain position 0
                    c->lo0_h[m] = c->first_ev[m];
                    c->front[m] = 0;
                    *row0_dirty = true;
                }
                r_start = std::min(r_start, c->front[m]);
            }
        }
        if (r_start == 0x7fffffff) r_start = std::max(c->R - 1, 0);
        return r_start;
    };
    auto upload_row0 = [&]() -> int {
        HIPCHK(c, hipMemcpyAsync(c->d_lo.p, c->lo0_h.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        std::copy(c->front.begin(), c->front.end(), c->front_dev.begin());
        HIPCHK(c, hipMemcpyAsync(c->d_front.p, c->front_dev.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        return SW_OK;
    };
    // what follows a finished loop i (its state is in c->h_rb): the chunk counters of its sweep, the host mirror of the front rounds,
    // and the finalize / witness-row / voter-mask launches of its sub-batch, armed here and enqueued behind the next loop's shot
    auto after_loop = [&](int i, int r_start, int64_t fin_from, hipEvent_t loop_done) -> int {
        if (c->chunk_plan[i].G >= 2) {
            // the sweep of this sub-batch is complete (the loop waited for it) and its counters came back with the
            // loop state: provisional entries per chunk, entries the repair changed.  (A loop never reports SW_OK
            // without at least one read-back behind a shot that waited for cs_events[i]: h_rb is this sub-batch's or a later one.)
            const unsigned* pv = reinterpret_cast<const unsigned*>(c->h_rb + ((unsigned char*)c->d_prov - c->d_rb));
            const sw_ctx::ChunkPlan& pl = c->chunk_plan[i];
            for (int k = 1; k < pl.G; ++k) {
                const unsigned cnt = pv[(size_t)i * SW_MAX_CHUNKS + k];
                const int64_t len = pl.a[k + 1] - pl.a[k];
                c->ctr.chunk_provisional += cnt;
                if (cnt > (unsigned)std::min<int64_t>((len * c->n) / (32 * (c->chunk_cfg == 1 ? 2 : 4)), 0x7fffffff)) {
                    c->ctr.chunk_resweeps++;
                    c->chunks_off = true;   // this hashgraph has members silent for longer than the halo: sweep unchunked from now on
                }
            }
            c->ctr.chunk_repaired += pv[(size_t)SW_PROV_ROWS * SW_MAX_CHUNKS + i];
        }
        // host mirror of the per-member front round (kept by the resolve kernel, read back with the loop state)
        const int R = c->R;
        for (int m = 0; m < n; ++m) c->front[m] = std::max(c->front[m], c->front_dev[m]);
        clen_prev.swap(clen);
        clk.mark(&c->stage_us[3]);
        // The rounds of every event below `limit` are final now (later sub-batches only add lo
        // entries that compare greater than every existing event), so their round numbers,
        // sees-masks, witness rows and voter masks are produced on a third stream, overlapping the
        // round loop of the next sub-batch — and ENQUEUED behind that loop's first shot (they must wait for
        // this loop's kernels, which the aux stream learns from an event recorded behind them).
        CHK(ensure_rounds(c, R + 2));
        const int64_t a0 = fin_from, k0 = cut[i + 1] - fin_from;   // (the early part of the last sub-batch is done)
        const int rs_ = r_start, i_ = i;
        aux_armed = true;
        const int S_ = S;
        pending_aux = [c, np, R, a0, k0, rs_, i_, S_, loop_done, &fin_t0, &aux_armed]() -> int {
            if (!aux_armed) return SW_OK;
            aux_armed = false;
            hipStream_t ax = c->stream_aux;
            HIPCHK(c, hipStreamWaitEvent(ax, loop_done, 0));
            if (i_ == 0 && c->profiling) { fin_t0 = next_event(c); (void)hipEventRecord(fin_t0, ax); }
            // (a finalize that runs beside the next round loop is throttled: fewer workgroups, less pressure on the loop's gathers;
            // the last one has nothing to hide behind and takes the whole GPU)
            const int blocks = (int)std::min<int64_t>((k0 + 3) / 4, i_ == S_ - 1 ? 8192 : c->fin_blocks);
            CHK(launch_finalize<NW>(c, ax, a0, k0, R, blocks));
            const int total = (R - rs_) * np;
            if (total > 0)
                hipLaunchKernelGGL(k_witness_table, dim3((total + 255) / 256), dim3(256), 0, ax,
                                   (const int*)c->d_lo.p, R, rs_, np, c->d_wit.p);
            c->ctr.kernel_launches += 1;
            return launch_voter_masks<NW>(c, rs_, R, ax);
        };
        clk.mark(&c->stage_us[4]);
        return SW_OK;
    };
    for (int i = 0; i < S; ++i) {
        const int64_t limit = cut[i + 1];
        bool row0_dirty = false;
        const int r_start = start_round(i, &row0_dirty);
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->cs_events[i], 0));
        if (row0_dirty) CHK(upload_row0());
        clk.mark(&c->stage_us[1]);
        const bool dbg_t = c->debug_timing && K >= 65536;
        const auto dbg_t0 = std::chrono::steady_clock::now();
        const int64_t dbg_it0 = c->ctr.round_iterations;
        if (dbg_t) (void)hipStreamSynchronize(c->stream);  // separates "waiting for the sweep" from the loop itself
        const auto dbg_t1 = std::chrono::steady_clock::now();
        // the last sub-batch has no later loop to hide its finalize behind: most of it runs beside the end of its own loop
        int64_t fin_from = cut[i];
        const std::function<int(const RState&)> early_fin = [c, np, &fin_from, limit, &pending_aux](const RState& st) -> int {
            CHK(pending_aux());   // (the previous sub-batch's launches go first: same stream, same order as without the early part)
            const int64_t upto = std::min<int64_t>(st.mlo, limit);
            if (st.iter <= 0 || upto < fin_from + 16384) return SW_OK;
            // events below the band of round st.r: round <= st.r - 1, rows 0 .. st.r of the table are committed and final
            CHK(ensure_rounds(c, st.r + 2));
            const int64_t k1 = upto - fin_from;
            CHK(launch_finalize<NW>(c, c->stream_aux, fin_from, k1, st.r + 1, c->fin_blocks));
            fin_from = upto;
            return SW_OK;
        };
        const bool split_fin = i == S - 1 && cut[i + 1] - cut[i] >= 65536;
        CHK(run_round_loop<NW>(c, r_start, limit, cut[i + 1] - cut[i], c->d_bounds.p + (size_t)(i + 1) * np, &tally_ms, &tally_launches, &pending_aux,
                               split_fin ? &early_fin : nullptr, c->fin_band ? cut[i] : 0x7fffffff));
        CHK(pending_aux());   // (a loop that returned before its first shot — never — would have left it undone)
        if (dbg_t) {
            const auto dbg_t2 = std::chrono::steady_clock::now();
            const double w = std::chrono::duration<double, std::milli>(dbg_t1 - dbg_t0).count();
            const double l = std::chrono::duration<double, std::milli>(dbg_t2 - dbg_t1).count();
            const int64_t its = c->ctr.round_iterations - dbg_it0;
            fprintf(stderr, "[sw] sub-batch %d: %lld events, waited %.3f ms for the sweep, loop %.3f ms, %lld iterations (%.1f us each)\n",
                    i, (long long)(cut[i + 1] - cut[i]), w, l, (long long)its, its ? l * 1e3 / (double)its : 0.0);
        }
        clk.mark(&c->stage_us[2]);
        HIPCHK(c, hipEventRecord(c->ev_loop_done, c->stream));
        CHK(after_loop(i, r_start, fin_from, c->ev_loop_done));
    }
    CHK(pending_aux());   // the last sub-batch's
    span_end(c, sp_rl);

    const int R = c->R;
    hipEvent_t fin_t1 = nullptr;
    if (c->profiling) { fin_t1 = next_event(c); (void)hipEventRecord(fin_t1, c->stream_aux); }
    // Everything later calls enqueue goes to the main stream (getters, decide_fame, find_order): it
    // waits for the other two streams ON THE DEVICE; the host returns without a synchronisation
    // (a Node makes one such call per gossip step — every host sync costs it ~15 us).
    HIPCHK(c, hipEventRecord(c->ev_aux_done, c->stream_aux));
    HIPCHK(c, hipEventRecord(c->ev_cs_done, c->stream_cs));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_cs_done, 0));
    span_end(c, sp_total);
    if (c->profiling || c->debug_timing) {
        HIPCHK(c, hipStreamSynchronize(c->stream_aux));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream_cs));
    }
    HIPCHK(c, hipGetLastError());
    std::copy(bounds_h.begin() + (size_t)S * np, bounds_h.begin() + (size_t)(S + 1) * np, c->divided_cnt.begin());
    clk.mark(&c->stage_us[5]);
    c->sw_dirty_from = std::max(R, 1);  // voter masks are up to date
    if (first + K == c->N) std::copy(c->head.begin(), c->head.end(), c->divided_head.begin());
    else for (int64_t e = first; e < first + K; ++e) c->divided_head[c->cr[e]] = (int32_t)e;
    c->divided = first + K;
    c->ctr.events_divided += K;
    c->ctr.rounds = R;
    if (c->profiling) {
        float ms = 0.f;
        if (cs_t0 && cs_t1) (void)hipEventElapsedTime(&ms, cs_t0, cs_t1);
        c->tm.can_see_ms = ms;   // on its own stream: overlaps rounds_ms
        c->tm.rounds_ms = span_ms(sp_rl);
        c->tm.tally_ms = tally_ms;
        c->tm.tally_launches = tally_launches;
        { float fm = 0.f; if (fin_t0 && fin_t1) (void)hipEventElapsedTime(&fm, fin_t0, fin_t1); c->tm.finalize_ms = fm; }  // aux stream span (overlaps)
        c->tm.total_ms = span_ms(sp_total);
        c->tm.cansee_kernel_ms = 0.f;
        c->tm.cansee_launches = (int32_t)cansee_spans.size();
        for (const Span& s_ : cansee_spans) c->tm.cansee_kernel_ms += span_ms(s_);
    }
    return SW_OK;
}

// first round not in `consensus` (swirld.py:226-228)
int first_undecided_round(const sw_ctx* c) {
    int max_c = 0;
    while (max_c < c->R && c->cons_h[max_c]) ++max_c;
    return max_c;
}

// voter masks (if stale) + the elections of the candidate rounds max_c + part, max_c + part + nparts, ...
template <int NW>
int fame_launch(sw_ctx* c, int max_c, int part, int nparts, Span* sp_el) {
    const int np = c->npad, R = c->R;
    if (c->payload_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_payload, 0));  // coin bits of a bulk append
    // voter masks: normally already produced by divide_rounds (per sub-batch, overlapped)
    const uint32_t tot2 = 2u * c->tot;
    if (c->sw_dirty_from < R || R > c->Sw_rows) CHK(launch_voter_masks<NW>(c, std::min(c->sw_dirty_from, R), R, c->stream));
    c->sw_dirty_from = std::max(R, 1);
    const bool tiled = NW >= 2 && c->elect_impl == 1;
    if (tiled) CHK(dgrow(c, c->d_rsc, (size_t)3 * c->Rcap, 0));
    hipLaunchKernelGGL(k_fame_prep, dim3(std::max(1, std::min(64, (3 * R + 255) / 256))), dim3(256), 0, c->stream, c->d_newc.p + sizeof(FameCounters), R,
                       tiled ? c->d_rsc.p : nullptr, tiled ? 3 * R : 0);
    const int nblk = R - max_c - part > 0 ? (R - max_c - part + nparts - 1) / nparts : 0;
    const int call_idx = (int)c->fame_calls.size();
    if (sp_el) *sp_el = span_begin(c);
    if (nblk > 0) {
        bool split_done = false;
#define SW_ELECT_ARGS (const int*)c->d_wit.p, (const u64*)c->d_Sw.p, (const unsigned char*)c->d_coin.p, (const uint32_t*)c->d_stake.p, \
                      tot2, c->coin_period, max_c, R, np, c->d_fam.p, c->d_cons.p, c->d_newc.p + sizeof(FameCounters), c->d_fc, c->d_dec_call.p, c->d_dec_by.p, call_idx, part, nparts
        if constexpr (NW >= 2) {  // NW threads per candidate, a round as 64 NW / CG workgroups of CG candidates (k_elections_tiled)
            if (c->elect_impl == 1) {
#define SW_ELECT_TILED(CG_)                                                                                                             \
    do {                                                                                                                                \
        if (c->unit_stake) hipLaunchKernelGGL((k_elections_tiled<NW, true, CG_>), dim3(nblk * (64 * NW / (CG_))), dim3((CG_) * NW), 0, c->stream, SW_ELECT_ARGS, c->d_rsc.p);  \
        else hipLaunchKernelGGL((k_elections_tiled<NW, false, CG_>), dim3(nblk * (64 * NW / (CG_))), dim3((CG_) * NW), 0, c->stream, SW_ELECT_ARGS, c->d_rsc.p);                \
    } while (0)
                if constexpr (NW == 2) SW_ELECT_TILED(128);
                else if constexpr (NW == 4) {
                    if (c->elect_cg == 64) SW_ELECT_TILED(64);
                    else if (c->elect_cg == 256) SW_ELECT_TILED(256);
                    else SW_ELECT_TILED(128);
                } else if constexpr (NW == 8) SW_ELECT_TILED(128);
                else SW_ELECT_TILED(64);
#undef SW_ELECT_TILED
                split_done = true;
            }
        }
        if (!split_done) {
            if (c->unit_stake) hipLaunchKernelGGL((k_elections<NW, true>), dim3(nblk), dim3(np), 0, c->stream, SW_ELECT_ARGS);
            else hipLaunchKernelGGL((k_elections<NW, false>), dim3(nblk), dim3(np), 0, c->stream, SW_ELECT_ARGS);
        }
#undef SW_ELECT_ARGS
        c->ctr.kernel_launches++;
    }
    if (sp_el) span_end(c, *sp_el);
    HIPCHK(c, hipGetLastError());
    return SW_OK;
}

void fame_counters(sw_ctx* c, const FameCounters& fc) {
    c->ctr.voter_evals += (int64_t)(fc.voter_evals - c->fc_seen.voter_evals);
    c->ctr.majority_evals += (int64_t)(fc.majority_evals - c->fc_seen.majority_evals);
    c->ctr.coin_votes += (int64_t)(fc.coin_votes - c->fc_seen.coin_votes);
    c->ctr.coin_flips += (int64_t)(fc.coin_flips - c->fc_seen.coin_flips);
    c->fc_seen = fc;
}

template <int NW>
int do_fame(sw_ctx* c, int32_t* new_rounds, int cap, int* n_new) {
    const int R = c->R;
    c->ev_used = 0;
    Span sp = span_begin(c), sp_el{};
    const int max_c = first_undecided_round(c);
    CHK(fame_launch<NW>(c, max_c, 0, 1, &sp_el));
    // fame counters + new_c flags: one copy into pinned memory
    HIPCHK(c, hipMemcpyAsync(c->h_fame, c->d_newc.p, sizeof(FameCounters) + (size_t)R, hipMemcpyDeviceToHost, c->stream));
    span_end(c, sp);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    FameCounters fc;
    memcpy(&fc, c->h_fame, sizeof fc);
    const unsigned char* newc = c->h_fame + sizeof(FameCounters);
    const int call_idx = (int)c->fame_calls.size();
    c->fame_calls.push_back({max_c, R, c->divided});
    int cnt = 0;
    for (int r = 0; r < R; ++r)
        if (newc[r]) {
            c->cons_h[r] = 1;
            c->cons_call[r] = call_idx;
            if (cnt < cap && new_rounds) new_rounds[cnt] = r;
            ++cnt;
        }
    if (n_new) *n_new = cnt;
    fame_counters(c, fc);
    if (c->profiling) { c->tm.fame_ms = span_ms(sp); c->tm.elections_ms = span_ms(sp_el); }
    if (cnt > cap) return fail(c, SW_ERANGE, "new_rounds capacity %d < %d", cap, cnt);
    return SW_OK;
}

// Candidate-partitioned decide_fame (multi-GPU row (e)): this part's share of the elections; the
// context's own tables receive this part's decisions only, nothing enters `consensus` yet.
template <int NW>
int do_fame_partial(sw_ctx* c, int part, int nparts, int8_t* famous, uint8_t* decided) {
    const int np = c->npad, n = c->n, R = c->R;
    const int max_c = first_undecided_round(c);
    CHK(fame_launch<NW>(c, max_c, part, nparts, nullptr));
    std::vector<signed char> fam((size_t)R * np);
    HIPCHK(c, hipMemcpyAsync(c->h_fame, c->d_newc.p, sizeof(FameCounters) + (size_t)R, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fam.data(), c->d_fam.p, fam.size(), hipMemcpyDeviceToHost, c->stream));
    // the kernel marks the rounds it completed in the device-side consensus flags; they become
    // official in sw_commit_fame, for every part alike
    HIPCHK(c, hipStreamSynchronize(c->stream));
    FameCounters fc;
    memcpy(&fc, c->h_fame, sizeof fc);
    const unsigned char* newc = c->h_fame + sizeof(FameCounters);
    for (int r = 0; r < R; ++r) {
        decided[r] = newc[r];
        const bool mine = r >= max_c && (r - max_c) % nparts == part;
        for (int m = 0; m < n; ++m) famous[(size_t)r * n + m] = (r < max_c || mine) ? fam[(size_t)r * np + m] : (signed char)-1;
    }
    fame_counters(c, fc);
    return SW_OK;
}

template <class T>
int get_round_rows(sw_ctx* c, const T* src, int r0, int r1, T* out, T absent) {
    if (!c || !out) return SW_EINVAL;
    if (r0 < 0 || r1 < r0) return fail(c, SW_ERANGE, "bad round range");
    const int np = c->npad, n = c->n;
    HIPCHK(c, hipSetDevice(c->device));
    const int rr = std::min(r1, c->R);
    std::vector<T> tmp((size_t)std::max(rr - r0, 0) * np);
    if (!tmp.empty()) {
        HIPCHK(c, hipMemcpyAsync(tmp.data(), src + (size_t)r0 * np, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int r = r0; r < r1; ++r)
        for (int m = 0; m < n; ++m)
            out[(size_t)(r - r0) * n + m] = r < rr ? tmp[(size_t)(r - r0) * np + m] : absent;
    return SW_OK;
}


// Windowed mode: evict every can_see row no later call can read.  Rows still needed: every member's
// latest event (the self-parent row of its next event; recent other-parents), every member's first
// not-yet-ordered event and what follows it (find_order's chain searches start at the ordered
// prefix), and everything from the thresholds lo[r][.] of the oldest round still in play — the first
// round not in `consensus` or the oldest front round of a member, whichever is older (band rows,
// witness rows, voter masks, famous witnesses of rounds yet to be ordered).  A silent member pins
// the horizon at its last event; the reference keeps every row forever (swirld.py:69-72).
int window_evict(sw_ctx* c) {
    if (!c->vm.active || c->divided != c->N || c->N == 0) return SW_OK;
    const int n = c->n, np = c->npad;
    int64_t horizon = c->N;
    int fmin = 0x7fffffff;
    CHK(ensure_pool_h(c));
    // members that have LAPSED (sw_set_window_lapse; the reference has no such notion: it keeps every row): silent for more
    // than `window_lapse` events.  Their latest row, their unordered tail and their front round stop holding the window
    // back; in exchange their further events are refused — waking one would restart the round loop at its old front
    // round, whose band begins at thresholds long evicted.  No other path reads their old rows: a round's band and
    // tally only follow members with a witness in it, find_order only chains with a famous witness in the rounds it orders.
    if (c->window_lapse > 0) {
        if ((int)c->lapsed.size() != n) c->lapsed.assign(n, 0);
        for (int m = 0; m < n; ++m)
            if (c->head[m] >= 0 && (int64_t)c->head[m] < c->N - c->window_lapse) c->lapsed[m] = 1;
    }
    auto is_lapsed = [&](int m) { return c->window_lapse > 0 && c->lapsed[m]; };
    for (int m = 0; m < n; ++m) {
        if (is_lapsed(m)) continue;
        if (c->head[m] >= 0) horizon = std::min<int64_t>(horizon, c->head[m]);
        if (c->ord_pos[m] < c->nev[m]) horizon = std::min<int64_t>(horizon, c->chain_ev_h[(size_t)c->chain_start_h[m] + c->ord_pos[m]]);
        if (c->front[m] >= 0) fmin = std::min(fmin, c->front[m]);
    }
    const int rmin = std::min(first_undecided_round(c), fmin == 0x7fffffff ? 0 : fmin);
    if (rmin < c->R) {
        std::vector<int32_t> row(np);
        HIPCHK(c, hipMemcpyAsync(row.data(), c->d_lo.p + (size_t)rmin * np, np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int m = 0; m < n; ++m) if (row[m] != SW_INF && !is_lapsed(m)) horizon = std::min<int64_t>(horizon, row[m]);
    } else horizon = 0;
    const size_t rowbytes = (size_t)np * sizeof(int32_t);
    CHK(vm_evict_below(c, (size_t)std::max<int64_t>(horizon, 0) * rowbytes));
    c->first_resident = (int64_t)((c->vm.lo * c->vm.chunk + rowbytes - 1) / rowbytes);
    return SW_OK;
}

// pinned staging of a find_order call: [rounds: nr][ordpos: npad][OrderInfo init][read-back block = the layout of d_oblk][hostflag: nr]
int ensure_order_host(sw_ctx* c, size_t bytes) {
    if (bytes <= c->h_ord_cap) return SW_OK;
    if (c->h_ord) { HIPCHK(c, hipDeviceSynchronize()); (void)hipHostFree(c->h_ord); c->h_ord = nullptr; c->h_ord_cap = 0; }
    size_t nc = bytes + bytes / 2 + 4096;
    if (hipHostMalloc((void**)&c->h_ord, nc, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(c, SW_ENOMEM, "hipHostMalloc(%zu bytes) for find_order failed", nc); }
    c->h_ord_cap = nc;
    return SW_OK;
}

// pinned staging of the group bounds of a bulk find_order call
int ensure_order_stage(sw_ctx* c, size_t ints) {
    if (ints <= c->h_ord_stage_cap) return SW_OK;
    if (c->h_ord_stage) { HIPCHK(c, hipDeviceSynchronize()); (void)hipHostFree(c->h_ord_stage); c->h_ord_stage = nullptr; c->h_ord_stage_cap = 0; }
    const size_t nc = ints * 2 + 256;
    if (hipHostMalloc((void**)&c->h_ord_stage, nc * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(c, SW_ENOMEM, "hipHostMalloc for find_order's group table failed"); }
    c->h_ord_stage_cap = nc;
    return SW_OK;
}

constexpr int order_tile(int NW) { return NW <= 8 ? 16 : 8; }   // positions per tile of k_order_median: 16 KB of LDS at 256 members (8 workgroups per CU)

template <int NW>
int do_find_order(sw_ctx* c, std::vector<int32_t> rounds, int32_t* out_events, int64_t cap, int64_t* n_out) {
    const int np = c->npad, n = c->n;
    const bool dbg = getenv("SW_DEBUG_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[find_order] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    std::sort(rounds.begin(), rounds.end());  // sorted(new_c), swirld.py:283
    rounds.erase(std::unique(rounds.begin(), rounds.end()), rounds.end());
    const int nr = (int)rounds.size();
    if (nr == 0) return SW_OK;
    if (c->payload_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_payload, 0));  // timestamps / signatures of a bulk append
    if (rounds.front() < 0 || rounds.back() >= c->R)
        return fail(c, SW_ERANGE, "find_order: round outside [0, %d) (KeyError in the reference)", c->R);
    if ((int64_t)(nr + 1) * np >= (1ll << 31)) return fail(c, SW_ERANGE, "find_order: %d rounds x %d columns exceed 2^31 table entries", nr, np);
    // ---- tables of the call, all on the device: f_w (:284), q (:288-293), ordered prefixes per entry, segment offsets ----
    // device block read back with ONE copy: [OrderInfo][ord_new: npad ints][acc_off: nr + 1 long long]
    const size_t oblk_ll = sizeof(OrderInfo) / 8 + (size_t)np / 2 + (size_t)nr + 1;
    const size_t h_rounds_off = 0, h_ordpos_off = ((size_t)nr * 4 + 63) & ~(size_t)63, h_init_off = h_ordpos_off + (size_t)np * 4,
                 h_blk_off = h_init_off + 64, h_flag_off = h_blk_off + ((oblk_ll * 8 + 63) & ~(size_t)63);
    CHK(ensure_order_host(c, h_flag_off + (size_t)nr * 4));
    int32_t* h_rounds = reinterpret_cast<int32_t*>(c->h_ord + h_rounds_off);
    int32_t* h_ordpos = reinterpret_cast<int32_t*>(c->h_ord + h_ordpos_off);
    OrderInfo* h_init = reinterpret_cast<OrderInfo*>(c->h_ord + h_init_off);
    const OrderInfo* h_info = reinterpret_cast<const OrderInfo*>(c->h_ord + h_blk_off);
    const int32_t* h_ordnew = reinterpret_cast<const int32_t*>(c->h_ord + h_blk_off + sizeof(OrderInfo));
    const long long* acc_off = reinterpret_cast<const long long*>(c->h_ord + h_blk_off + sizeof(OrderInfo) + (size_t)np * 4);
    int32_t* hostflag = reinterpret_cast<int32_t*>(c->h_ord + h_flag_off);
    std::copy(rounds.begin(), rounds.end(), h_rounds);
    std::fill(h_ordpos, h_ordpos + np, 0);
    std::copy(c->ord_pos.begin(), c->ord_pos.end(), h_ordpos);
    *h_init = OrderInfo{0, 0x7fffffff, 0, {0, 0, 0, 0}};
    CHK(dgrow(c, c->d_ord_rounds, nr, 0));
    CHK(dgrow(c, c->d_fwm, (size_t)2 * nr * np, 0));   // [famous witness | its chain position] per (entry, member)
    CHK(dgrow(c, c->d_q, (size_t)nr * np, 0));
    CHK(dgrow(c, c->d_ordat, (size_t)(nr + 1) * np, 0));
    CHK(dgrow(c, c->d_seg, (size_t)3 * nr * np, 0));
    CHK(dgrow(c, c->d_rowsum, nr, 0));
    CHK(dgrow(c, c->d_oblk, oblk_ll, 0));
    CHK(dgrow(c, c->d_ordpos, np, 0));
    OrderInfo* d_info = reinterpret_cast<OrderInfo*>(c->d_oblk.p);
    int32_t* d_ordnew = reinterpret_cast<int32_t*>(c->d_oblk.p) + sizeof(OrderInfo) / 4;
    long long* d_acc_off = c->d_oblk.p + sizeof(OrderInfo) / 8 + np / 2;
    int32_t *d_seg_start = c->d_seg.p, *d_seg_len = c->d_seg.p + (size_t)nr * np, *d_seg_off = c->d_seg.p + (size_t)2 * nr * np;
    HIPCHK(c, hipMemcpyAsync(c->d_ord_rounds.p, h_rounds, (size_t)nr * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ordpos.p, h_ordpos, (size_t)np * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_info, h_init, sizeof(OrderInfo), hipMemcpyHostToDevice, c->stream));
    int32_t* d_fwseq = c->d_fwm.p + (size_t)nr * np;
    hipLaunchKernelGGL(k_order_prep, dim3(nr), dim3(np), 0, c->stream, (const int*)c->d_ord_rounds.p, (const int*)c->d_wit.p,
                       (const signed char*)c->d_fam.p, (const int*)c->d_seq.p, n, np, c->d_fwm.p, d_fwseq, d_info);
    hipLaunchKernelGGL(k_order_bounds<NW>, dim3(nr, NW), dim3(64 * NW), 0, c->stream, (const int*)c->d_fwm.p, (const int*)c->d_L.p,
                       (const int*)c->d_seq.p, (const uint32_t*)c->d_stake.p, c->tot, (const int*)c->d_chain_start.p,
                       (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, n, (int)(c->N - 1), c->d_q.p);
    hipLaunchKernelGGL(k_order_runmax, dim3(1), dim3(np), 0, c->stream, (const int*)c->d_q.p, (const int*)c->d_ordpos.p, nr, np,
                       c->d_ordat.p, d_seg_start, d_seg_len, d_ordnew);
    hipLaunchKernelGGL(k_order_rowscan, dim3(nr), dim3(np), 0, c->stream, (const int*)d_seg_len, np, d_seg_off, c->d_rowsum.p);
    hipLaunchKernelGGL(k_order_offsets, dim3(1), dim3(1024), 0, c->stream, (const int*)c->d_rowsum.p, nr, d_acc_off, d_info);
    c->ctr.kernel_launches += 5;
    HIPCHK(c, hipMemcpyAsync(c->h_ord + h_blk_off, c->d_oblk.p, oblk_ll * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    lap("tables");
    if (h_info->undecided_ri != 0x7fffffff)
        return fail(c, SW_EINVAL, "find_order: round %d has an undecided witness (KeyError on self.famous[w], swirld.py:284)", rounds[h_info->undecided_ri]);
    const int64_t n_acc = h_info->n_acc;
    if (n_acc > 0x7ffffff0ll) return fail(c, SW_ERANGE, "find_order: more than 2^31 events in one call");
    std::vector<int32_t> acc_ev;   // host copy: fetched only for the rounds the host has to sort
    std::vector<double> ts;        // (host copy of the timestamps: only for rounds the host sorts)
    std::vector<unsigned char> white_h;
    std::fill(hostflag, hostflag + nr, 0);
    const size_t tx_at = c->transactions.size();
    if (!c->transactions.resize(tx_at + (size_t)n_acc)) return fail(c, SW_ENOMEM, "pinned host memory for the ordered events");   // the device-sorted order lands in place
    int32_t* sorted = c->transactions.data() + tx_at;
    bool copied_early_ok = false, any_host_sorted = false;   // out_events already holds the order / the host re-sorted rounds of it
    if (n_acc) {
        CHK(dgrow(c, c->d_acc_ev, n_acc, 0));
        CHK(dgrow(c, c->d_acc_ri, n_acc, 0));
        CHK(dgrow(c, c->d_ts, n_acc, 0));
        CHK(dgrow(c, c->d_white, (size_t)nr * 64, 0));
        CHK(dgrow(c, c->d_sorted, n_acc, 0));
        CHK(dgrow(c, c->d_hostflag, nr, 0));
        // rounds too large for the LDS sort: the same network over global scratch, one workgroup per such round (behind everything else)
        std::vector<int32_t> big_ri;
        std::vector<long long> big_off{0};
        for (int i = 0; i < nr; ++i) {
            const int64_t len = acc_off[i + 1] - acc_off[i];
            if (len <= SORT_CAP || getenv("SW_ORDER_BIG_HOST")) continue;   // (test hook: oversize rounds to the host, as before round 4)
            int64_t m = 1;
            while (m < len) m <<= 1;
            big_ri.push_back(i);
            big_off.push_back(big_off.back() + m);
        }
        // device sort of the round entries [i0, i1) by (ts, first 8 whitened key bytes), then their part of the order to the host
        auto sort_and_fetch = [&](hipStream_t st, int i0, int i1, bool fetch) -> int {
            hipLaunchKernelGGL(k_order_sort, dim3(i1 - i0), dim3(1024), 0, st, (const int*)c->d_acc_ev.p,
                               (const long long*)d_acc_off, (const double*)c->d_ts.p, (const unsigned char*)c->d_sig.p,
                               (const unsigned char*)c->d_white.p, i0, c->d_sorted.p, c->d_hostflag.p);
            c->ctr.kernel_launches++;
            const int64_t a0 = acc_off[i0], na = acc_off[i1] - a0;
            if (fetch && na) HIPCHK(c, hipMemcpyAsync(sorted + a0, c->d_sorted.p + a0, na * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            return SW_OK;
        };
        const bool fetch_early = big_ri.empty();   // (oversize rounds are sorted last: then the whole order travels at the end)
        // what only the consumers of the table need (the ordered events round-major, the whitening keys, cleared flags): enqueued
        // on THEIR stream, beside the first walk
        auto consumer_preamble = [&](hipStream_t st) -> int {
            HIPCHK(c, hipMemsetAsync(c->d_hostflag.p, 0, nr * sizeof(int32_t), st));
            hipLaunchKernelGGL(k_order_segments, dim3((unsigned)(((size_t)nr * np + 255) / 256)), dim3(256), 0, st, (const int*)d_seg_start,
                               (const int*)d_seg_len, (const int*)d_seg_off, (const long long*)d_acc_off, (const int*)c->d_chain_start.p,
                               (const int*)c->d_chain_ev.p, np, nr * np, c->d_acc_ev.p, c->d_acc_ri.p);
            hipLaunchKernelGGL(k_order_white, dim3(nr), dim3(1024), 0, st, (const int*)c->d_fwm.p, n, np,
                               (const unsigned char*)c->d_sig.p, c->d_white.p);
            c->ctr.kernel_launches += 2;
            return SW_OK;
        };
        // A call that orders many events: the first-descendant table (k_order_walk) instead of one binary search per
        // (event, famous witness) pair.  Same samples, by construction and by test (SW_ORDER_BULK=<events> moves the
        // threshold, 0 = never).
        const int64_t bulk_min = getenv("SW_ORDER_BULK") ? atoll(getenv("SW_ORDER_BULK")) : 16384;   // (read per call: the tests force either path)
        // Beyond 512 members every call takes the table: the search kernel's 16-word instance keeps ~90 wave masks alive around
        // its divergent chain searches, the compiler spills them, and the timestamps came back wrong AND different from run
        // to run (1024 members x 100 k events, profiles/r06_order_search_16_words.txt) — it is not built.
        const bool bulk = NW > 8 || (bulk_min > 0 && n_acc >= bulk_min);
        hipStream_t tail = c->stream;   // the stream the last kernels of the call are on
        if (bulk) {
            // The table is built for GROUPS of consecutive round entries whose events fit SW_ORDER_SLAB_MB [256 per 256 columns: 4 groups per 1 M events — 128 and 512 are both 0.15-0.3 ms slower, profiles/r06_measured_and_dropped.txt]
            // of table — a slab that stays allocated instead of one table as large as the can_see rows of everything the call
            // orders — and there are TWO slabs: the samples, the sort and the read-back of group g run on a second stream
            // beside the walk of group g + 1 (the walk is bound by its stores, the samples by their selection loops).
            constexpr int P = order_tile(NW);
            const int64_t slab_mb = getenv("SW_ORDER_SLAB_MB") ? atoll(getenv("SW_ORDER_SLAB_MB")) : 256 * std::max(1, NW / 4);
            const int64_t pad = (int64_t)P * np;   // every chain rounds its positions up to whole tiles
            const int64_t slab_pos = std::max<int64_t>(2 * pad, (slab_mb << 20) / ((int64_t)n * 4));
            struct Group { int i0, i1; };
            std::vector<Group> groups;
            int64_t stride = 0;
            for (int i0 = 0; i0 < nr;) {
                int i1 = i0 + 1;
                while (i1 < nr && acc_off[i1 + 1] - acc_off[i0] + pad <= slab_pos) ++i1;
                if (acc_off[i1] > acc_off[i0]) { groups.push_back({i0, i1}); stride = std::max<int64_t>(stride, acc_off[i1] - acc_off[i0] + pad); }
                i0 = i1;
            }
            // the planes of the members sit `stride` entries apart and are read / written at the same offsets at the same
            // time: 4 KB x odd + 256 B between them, so that they spread over the memory channels whatever the interleaving
            // granule (a power of two — 131 072 entries for a full slab — put all 256 planes' lines on one channel)
            stride = (((stride + 1023) >> 10) | 1) * 1024 + 64;
            const bool two = groups.size() > 1 && !getenv("SW_ORDER_ONE_STREAM");
            CHK(dgrow(c, c->d_fd, (size_t)stride * n * (two ? 2 : 1), 0));
            CHK(dgrow(c, c->d_grp, (OrderGroup::ints(np) + 2) * std::max<size_t>(groups.size(), 1), 0));
            int32_t* d_gbounds = c->d_grp.p + OrderGroup::ints(np) * groups.size();
            CHK(ensure_order_stage(c, groups.size() * 2));
            for (size_t gi = 0; gi < groups.size(); ++gi) { c->h_ord_stage[2 * gi] = groups[gi].i0; c->h_ord_stage[2 * gi + 1] = groups[gi].i1; }
            HIPCHK(c, hipMemcpyAsync(d_gbounds, c->h_ord_stage, groups.size() * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_order_group, dim3((unsigned)groups.size()), dim3(np), 0, c->stream, (const int*)c->d_ordat.p, (const int*)c->d_fwm.p,
                               (const int*)c->d_chain_start.p, (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, (const int*)d_gbounds, n, np, P, c->d_grp.p);
            c->ctr.kernel_launches++;
            constexpr int CW = NW >= 4 ? 256 : 64 * NW;
            const int ncg = np / CW;
            int S = getenv("SW_ORDER_S") ? atoi(getenv("SW_ORDER_S")) : (1024 + n * ncg - 1) / (n * ncg);
            S = std::min(std::max(S, 1), 64);
            if (two) {
                if (!c->stream_ord) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_ord, hipStreamNonBlocking));
                for (auto& st : c->stream_srt) if (!st) HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                while (c->ord_events.size() < 3 * groups.size() + 3) {
                    hipEvent_t e;
                    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->ord_events.push_back(e);
                }
            }
            hipStream_t s1 = two ? c->stream_ord : c->stream;
            lap("group");
            CHK(consumer_preamble(s1));   // (the tables it reads are complete: the host has waited for them)
            if (dbg) { (void)hipStreamSynchronize(s1); lap("preamble"); }
            {   // timestamps in chain order from the position in front of every member's first unordered event on
                CHK(dgrow(c, c->d_tch, c->d_chain_ev.cap, 0));
                int longest = 1;
                for (int m = 0; m < n; ++m) longest = std::max(longest, c->nev[m] - c->ord_pos[m] + 1);
                hipLaunchKernelGGL(k_order_tchain, dim3(n, (unsigned)std::min(64, (longest + 255) / 256)), dim3(256), 0, s1, (const int*)c->d_chain_start.p,
                                   (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, (const int*)c->d_ordpos.p, (const double*)c->d_t.p, c->d_tch.p);
                c->ctr.kernel_launches++;
            }
            for (size_t gi = 0; gi < groups.size(); ++gi) {
                const Group& g = groups[gi];
                int* grp = c->d_grp.p + OrderGroup::ints(np) * gi;
                int* fd = c->d_fd.p + (two && (gi & 1) ? (size_t)stride * n : 0);
                const int64_t a0 = acc_off[g.i0], na = acc_off[g.i1] - a0;
                if (two && gi >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ord_events[3 * (gi - 2) + 1], 0));   // the slab's last reader
                hipLaunchKernelGGL(k_order_walk<CW>, dim3((unsigned)((size_t)n * ncg * S)), dim3(CW), 0, c->stream, (const int*)c->d_L.p,
                                   (const int*)c->d_seq.p, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p, (const int*)c->d_ordat.p,
                                   g.i0, g.i1, (const int*)grp, np, S, P, (long long)stride, fd);
                if (dbg) { (void)hipStreamSynchronize(s1); lap("walk"); }
                if (two) {
                    HIPCHK(c, hipEventRecord(c->ord_events[3 * gi], c->stream));
                    HIPCHK(c, hipStreamWaitEvent(s1, c->ord_events[3 * gi], 0));
                }
                hipLaunchKernelGGL((k_order_median<NW, P>), dim3((unsigned)(na / P + n)), dim3(256), 0, s1, (const int*)fd,
                                   (long long)stride, (const int*)grp, (const int*)c->d_ordat.p, g.i0, g.i1, n, (const int*)d_fwseq,
                                   (const long long*)d_acc_off, (const int*)d_seg_off, (const int*)c->d_chain_start.p, (const double*)c->d_tch.p,
                                   c->d_ts.p, d_info);
                if (dbg) { (void)hipStreamSynchronize(s1); lap("median"); }
                hipStream_t s2 = s1;
                if (two) {
                    HIPCHK(c, hipEventRecord(c->ord_events[3 * gi + 1], s1));
                    if (!getenv("SW_ORDER_SORT_INLINE")) {
                        s2 = c->stream_srt[gi & 1];
                        HIPCHK(c, hipStreamWaitEvent(s2, c->ord_events[3 * gi + 1], 0));
                    }
                }
                c->ctr.kernel_launches += 2;
                CHK(sort_and_fetch(s2, g.i0, g.i1, fetch_early));
                if (two && fetch_early) HIPCHK(c, hipEventRecord(c->ord_events[3 * gi + 2], s2));
            }
            // (round entries that order nothing keep their zeroed flag and have nothing to sort)
            if (two) {   // the side streams join the main one
                hipStream_t side[3] = {c->stream_ord, c->stream_srt[0], c->stream_srt[1]};
                for (int k = 0; k < 3; ++k) {
                    hipEvent_t e = c->ord_events[3 * groups.size() + k];
                    HIPCHK(c, hipEventRecord(e, side[k]));
                    HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
                }
            }
            if (two && fetch_early && out_events && !getenv("SW_ORDER_HOST") && !getenv("SW_ORDER_LATE_COPY")) {
                // the caller's copy of the order, group by group as it arrives, while the device works on the later groups
                for (size_t gi = 0; gi < groups.size(); ++gi) {
                    HIPCHK(c, hipEventSynchronize(c->ord_events[3 * gi + 2]));
                    const int64_t a0 = acc_off[groups[gi].i0], a1 = std::min<int64_t>(acc_off[groups[gi].i1], cap);
                    if (a1 > a0) memcpy(out_events + a0, sorted + a0, (size_t)(a1 - a0) * sizeof(int32_t));
                }
                copied_early_ok = true;
            }
            lap("walk+median");
        } else {
            CHK(consumer_preamble(c->stream));
            if constexpr (NW <= 8) {
                hipLaunchKernelGGL(k_order_times<NW>, dim3((unsigned)((n_acc + 3) / 4)), dim3(256), 0, c->stream,
                                   (const int*)c->d_acc_ev.p, (const int*)c->d_acc_ri.p, (int)n_acc, (const int*)c->d_fwm.p,
                                   (const int*)c->d_L.p, (const int*)c->d_cr.p, (const int*)c->d_seq.p,
                                   (const double*)c->d_t.p, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p,
                                   (const int*)c->d_ordpos.p, c->d_ts.p, d_info);
                c->ctr.kernel_launches++;
            }
            CHK(sort_and_fetch(c->stream, 0, nr, fetch_early));
        }
        if (!big_ri.empty()) {
            CHK(dgrow(c, c->d_big_ri, big_ri.size(), 0));
            CHK(dgrow(c, c->d_big_off, big_off.size(), 0));
            CHK(dgrow(c, c->d_sk_ts, (size_t)big_off.back(), 0));
            CHK(dgrow(c, c->d_sk_k8, (size_t)big_off.back(), 0));
            CHK(dgrow(c, c->d_sk_ev, (size_t)big_off.back(), 0));
            // (pageable sources: the copies are staged before the call returns)
            HIPCHK(c, hipMemcpyAsync(c->d_big_ri.p, big_ri.data(), big_ri.size() * sizeof(int32_t), hipMemcpyHostToDevice, tail));
            HIPCHK(c, hipMemcpyAsync(c->d_big_off.p, big_off.data(), big_off.size() * sizeof(long long), hipMemcpyHostToDevice, tail));
            hipLaunchKernelGGL(k_order_sort_big, dim3((unsigned)big_ri.size()), dim3(1024), 0, tail, (const int*)c->d_big_ri.p,
                               (const long long*)c->d_big_off.p, (const int*)c->d_acc_ev.p, (const long long*)d_acc_off,
                               (const double*)c->d_ts.p, (const unsigned char*)c->d_sig.p, (const unsigned char*)c->d_white.p,
                               c->d_sk_ts.p, c->d_sk_k8.p, c->d_sk_ev.p, c->d_sorted.p, c->d_hostflag.p);
            c->ctr.kernel_launches++;
            HIPCHK(c, hipMemcpyAsync(sorted, c->d_sorted.p, n_acc * sizeof(int32_t), hipMemcpyDeviceToHost, tail));
        }
        lap("sort kernels");
        HIPCHK(c, hipMemcpyAsync(c->h_ord + h_blk_off, d_info, sizeof(OrderInfo), hipMemcpyDeviceToHost, tail));
        HIPCHK(c, hipMemcpyAsync(hostflag, c->d_hostflag.p, nr * sizeof(int32_t), hipMemcpyDeviceToHost, tail));
        HIPCHK(c, hipStreamSynchronize(tail));
        HIPCHK(c, hipGetLastError());
        if (h_info->index_err) {
            c->transactions.resize(tx_at);   // nothing of this call is kept
            return fail(c, SW_ERANGE, "find_order: an event is seen by a single famous witness (IndexError at swirld.py:305)");
        }
        bool any_flag = false;
        if (getenv("SW_ORDER_HOST")) std::fill(hostflag, hostflag + nr, 1);  // test hook: host sort
        for (int i = 0; i < nr; ++i) { any_flag = any_flag || hostflag[i]; c->ctr.order_rounds_host_sorted += hostflag[i] ? 1 : 0; }
        any_host_sorted = any_flag;
        if (any_flag) {  // rare: oversize round or a (ts, 8-byte key) tie: the host needs ts, the whitening keys and the signatures
            acc_ev.resize((size_t)n_acc);
            ts.resize((size_t)n_acc);
            white_h.resize((size_t)nr * 64);
            HIPCHK(c, hipMemcpyAsync(ts.data(), c->d_ts.p, n_acc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(acc_ev.data(), c->d_acc_ev.p, n_acc * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(white_h.data(), c->d_white.p, white_h.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            CHK(ensure_sig_h(c));
        }
    }
    lap("read-back");
    // final order inside each round: (consensus timestamp, whitened signature), swirld.py:306
    struct Item { double ts; uint64_t k8; int32_t ev; };
    int64_t produced = 0;
    std::vector<Item> items;
    for (int i = 0; i < nr; ++i) {
        if (!n_acc || !hostflag[i]) {  // sorted on the device: already in `transactions`
            produced += acc_off[i + 1] - acc_off[i];
            continue;
        }
        const unsigned char* white = white_h.data() + (size_t)i * 64;  // swirld.py:285
        items.clear();
        for (int64_t a = acc_off[i]; a < acc_off[i + 1]; ++a) {
            Item it{ts[a], 0, acc_ev[a]};
            const unsigned char* sg = c->sig_h.data() + (size_t)acc_ev[a] * 64;
            for (int b = 0; b < 8; ++b) it.k8 = (it.k8 << 8) | (unsigned char)(white[b] ^ sg[b]);
            items.push_back(it);
        }
        std::sort(items.begin(), items.end(), [&](const Item& x, const Item& y) {
            if (x.ts != y.ts) return x.ts < y.ts;
            if (x.k8 != y.k8) return x.k8 < y.k8;
            const unsigned char* sx = c->sig_h.data() + (size_t)x.ev * 64;
            const unsigned char* sy = c->sig_h.data() + (size_t)y.ev * 64;
            for (int b = 8; b < 64; ++b) {
                const unsigned char kx = white[b] ^ sx[b], ky = white[b] ^ sy[b];
                if (kx != ky) return kx < ky;
            }
            return x.ev < y.ev;
        });
        for (const Item& it : items) sorted[produced++] = it.ev;   // swirld.py:307-309
    }
    if (out_events && n_acc && !(copied_early_ok && !any_host_sorted)) memcpy(out_events, sorted, (size_t)std::min<int64_t>(n_acc, cap) * sizeof(int32_t));
    lap("sort");
    std::copy(h_ordnew, h_ordnew + n, c->ord_pos.begin());
    CHK(window_evict(c));
    if (n_out) *n_out = produced;
    if (produced > cap) return fail(c, SW_ERANGE, "find_order: out_events capacity %lld < %lld", (long long)cap, (long long)produced);
    return SW_OK;
}


}  // namespace

// ====================================================================================
//                                     C-ABI
// ====================================================================================
extern "C" {

int sw_version(void) { return 7; }

const char* sw_last_error(const sw_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sw_create(int n_members, const uint64_t* stake, int coin_period, int device, sw_ctx** out) {
    if (!out) return fail(nullptr, SW_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_members < 1 || n_members > SW_MAX_MEMBERS) return fail(nullptr, SW_EINVAL, "n_members must be in [1, %d]", SW_MAX_MEMBERS);
    if (!stake) return fail(nullptr, SW_EINVAL, "stake is NULL");
    if (coin_period < 1) return fail(nullptr, SW_EINVAL, "coin_period must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SW_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, SW_ENODEV, "device %d out of range (%d devices)", device, ndev);
    uint64_t tot = 0;
    bool unit = true;
    for (int i = 0; i < n_members; ++i) {
        if (stake[i] > (1ull << 30)) return fail(nullptr, SW_EOVERFLOW, "stake[%d] too large for the 32-bit tally", i);
        tot += stake[i];
        unit = unit && stake[i] == 1;
    }
    if (tot >= (1ull << 30)) return fail(nullptr, SW_EOVERFLOW, "total stake %llu >= 2^30", (unsigned long long)tot);
    sw_ctx* c = new sw_ctx();
    c->n = n_members;
    int nw = 1;
    while (nw * 64 < n_members) nw *= 2;
    c->nw = nw;
    c->npad = nw * 64;
    c->coin_period = coin_period;
    c->device = device;
    c->unit_stake = unit;
    c->tot = (uint32_t)tot;
    c->stake_h.assign(c->npad, 0);
    for (int i = 0; i < n_members; ++i) c->stake_h[i] = (uint32_t)stake[i];
    c->head.assign(n_members, -1);
    c->first_ev.assign(n_members, -1);
    c->front.assign(n_members, -1);
    c->front_dev.assign(c->npad, -1);
    c->divided_head.assign(c->npad, -1);
    c->ord_pos.assign(n_members, 0);
    c->lo0_h.assign(c->npad, SW_INF);
    c->NEARCAP = std::max(64 * c->npad, 4096);
    c->MCAP = std::max(1024 * c->npad, 65536);
    if (c->npad > 512) {
        // 1024 members: the tally runs at the L2 roofline (n^2 / 8 bytes of hop masks per evaluation), so fewer
        // evaluations per round pay: a round ends >= 10-11 chain positions behind its start there, a window of 12
        // slots from offset 10 finds it in 1.1 iterations per round at uniform gossip (41.3 -> 48.7 M ev/s at
        // 2 M events; coin-round stress +8 %, two cliques -5 %: profiles/r03y_*)
        c->K = 12;
        c->skip = 10;
    }
    // Tuning / diagnostic switches (environment, read here once per context).  None changes results; every one is
    // VALIDATED: a value that is not an integer inside the documented range fails sw_create with SW_EINVAL naming
    // the variable, instead of being silently clamped or read as 0.
    std::string knob_err;
    auto knob = [&](const char* name, long long lo, long long hi, auto* dst) {
        const char* s = getenv(name);
        if (!s || !*s) return;
        char* end = nullptr;
        const long long v = strtoll(s, &end, 10);
        if (*end != 0 || v < lo || v > hi) {
            if (knob_err.empty()) { char b[160]; snprintf(b, sizeof b, "%s=%s: an integer in [%lld, %lld]", name, s, lo, hi); knob_err = b; }
            return;
        }
        *dst = static_cast<std::remove_pointer_t<decltype(dst)>>(v);
    };
    int graph = c->use_graph ? 1 : 0;
    c->K_auto = !(getenv("SW_TALLY_K") && *getenv("SW_TALLY_K"));
    c->tally_auto = !(getenv("SW_TALLY_IMPL") && *getenv("SW_TALLY_IMPL"));
    knob("SW_TALLY_K", 1, 63, &c->K);             // (rounded to a multiple of 4 and capped at 60 below: a member's row of the candidate table has 64 columns)
    knob("SW_BAND", 64, 1 << 28, &c->NEARCAP);
    knob("SW_BAND_MAX", 64, 1 << 28, &c->MCAP);
    c->MCAP = std::max(c->MCAP, c->NEARCAP);
    knob("SW_BATCH", 1, 4096, &c->BATCH);
    knob("SW_GRAPH", 0, 1, &graph);
    knob("SW_GRAPH_BIG", 0, 512, &c->graph_big);
    c->graph_big &= ~1;
    c->use_graph = graph != 0;
    // 1024-thread workgroups (1024 members): one per CU is resident, 512 would run as two shifts (profiles/r04n_*: 41.4 -> 40.1 ms at 1024 x 2 M)
    if (c->npad >= 1024) c->band_blocks = 256;
    knob("SW_BAND_BLOCKS", 1, 4096, &c->band_blocks);
    knob("SW_TALLY_PF", 0, 1, &c->tally_pf);
    knob("SW_MID_PCT", 0, 99, &c->mid_pct);
    knob("SW_FIN_BAND", 0, 1, &c->fin_band);
    knob("SW_FIN_BLOCKS", 64, 8192, &c->fin_blocks);
    // beyond 512 members the sweep is the level-bucketed kernel (1024-thread workgroups, one per CU) and the loop kernels are
    // throughput-bound: side by side they cost each other more than the overlap hides (round 6, profiles/r06_pipe_1024.txt: 1024
    // members / 2 M events 32.55 -> 31.49 ms, 8 M events 125.4 -> 124.0, coin-round stress 44.5 -> 38.9; at 512 members the opposite,
    // 20.7 -> 26.3): every sweep of a call first, then ONE loop
    if (c->npad > 512) c->pipe = 1;
    knob("SW_PIPE", 1, SW_PROV_ROWS - 1, &c->pipe);   // (head + pipe sub-batches: every one of them has a row of chunk counters)
    if (c->npad > 256) c->cansee_impl = 3;  // wide member counts: the level-bucketed streaming kernel is faster than the dataflow sweep there
    knob("SW_CANSEE_IMPL", 2, 6, &c->cansee_impl);
    if (c->cansee_impl == 4 || c->cansee_impl == 5) knob_err = "SW_CANSEE_IMPL: 6 (dataflow sweep), 2 or 3 (level-bucketed sweep)";
    knob("SW_TALLY_IMPL", 0, 2, &c->tally_impl);   // 0 column-lane, 1 bit-sliced (one wave per slot), 2 bit-sliced two-level search (one workgroup per member)
    knob("SW_FLOW_CFG", 0, 3, &c->flow_cfg);
    knob("SW_CHUNKS", 1, SW_MAX_CHUNKS, &c->chunks);
    knob("SW_CHUNK_CFG", 0, 2, &c->chunk_cfg);
    knob("SW_BAND_FAST", 0, 1, &c->band_fast);
    if (c->npad > 256) c->tally_filter = 1;
    knob("SW_TALLY_FILTER", 0, 1, &c->tally_filter);
    knob("SW_SPLIT_EMULATE", 0, SW_MAX_PARTS, &c->split_emulate);   // (measurement: the parts of a split played by one context; pin SW_TALLY_IMPL=1 with it)
    knob("SW_SHOT_PCT", 10, 400, &c->shot_pct);
    knob("SW_SHOT_EXTRA", 0, 64, &c->shot_extra);
    knob("SW_CHUNK_MIN", 64, 1 << 30, &c->chunk_min);
    c->halo = 32 * (int64_t)c->npad;
    knob("SW_HALO", 0, 1 << 24, &c->halo);
    knob("SW_ELECT_IMPL", 0, 1, &c->elect_impl);
    knob("SW_ELECT_CG", 64, 256, &c->elect_cg);
    if (c->elect_cg != 64 && c->elect_cg != 128 && c->elect_cg != 256) knob_err = "SW_ELECT_CG: 64, 128 or 256";
    knob("SW_GALLOP", 0, 255, &c->gallop_after);
    knob("SW_SKIP", 0, 32, &c->skip);
    knob("SW_RING_H", 0, 64, &c->ring_H_req);      // ring depth of the level-bucketed sweep (0 = automatic)
    if (!knob_err.empty()) {
        delete c;
        return fail(nullptr, SW_EINVAL, "%s", knob_err.c_str());
    }
    c->debug_timing = getenv("SW_DEBUG_TIMING") != nullptr;
    if (c->debug_timing) {
        if (hipMalloc(&c->d_flow_dbg, 8 * sizeof(u64)) != hipSuccess) c->d_flow_dbg = nullptr;
        else (void)hipMemset(c->d_flow_dbg, 0, 8 * sizeof(u64));
    }
    if (getenv("SW_DEBUG_CLOCKS")) {  // diagnostics: phase stamps of the round-loop kernels
        c->dbg_minor = atoi(getenv("SW_DEBUG_CLOCKS")) >= 2 ? 0 : 1;
        if (atoi(getenv("SW_DEBUG_CLOCKS")) == 3) {
            if (hipMalloc(&c->d_dbg_blk, (size_t)SW_DBG_MAX_ITERS * 2 * 2048 * 8) != hipSuccess) c->d_dbg_blk = nullptr;
            else (void)hipMemset(c->d_dbg_blk, 0, (size_t)SW_DBG_MAX_ITERS * 2 * 2048 * 8);
        }
        if (hipMalloc(&c->d_dbg, (size_t)SW_DBG_MAX_ITERS * 32 * 8) != hipSuccess) c->d_dbg = nullptr;
        else (void)hipMemset(c->d_dbg, 0, (size_t)SW_DBG_MAX_ITERS * 32 * 8);
    }
    c->nev.assign(n_members, 0);
    c->BATCH = (c->BATCH + 1) & ~1;  // even: the loop state is double-buffered by iteration parity
    // npad*K waves, 4 per workgroup (npad is a multiple of 64 anyway); a member's row of the candidate
    // table has 64 slots and slot 0 is the window header, so at most 60 candidates after rounding
    c->K = std::min((c->K + 3) & ~3, 60);
    c->K_flat = c->K;
    auto bail = [&](int rc) { g_create_error = c->err; sw_destroy(c); return rc; };
#define CCHK(expr) do { int rc_ = (expr); if (rc_ != SW_OK) return bail(rc_); } while (0)
#define CHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(c, SW_EIO, "%s: %s", #expr, hipGetErrorString(e_)); return bail(SW_EIO); } } while (0)
    CHIP(hipSetDevice(device));
    {
        // the round loop is the critical path: its stream gets the highest priority, the can_see
        // sweeps (which have slack when pipelined) the lowest
        int least = 0, greatest = 0;
        const char* pe = getenv("SW_PRIO");
        const bool use_prio = !pe || atoi(pe) != 0;
        if (use_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            CHIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
            CHIP(hipStreamCreateWithPriority(&c->stream_cs, hipStreamNonBlocking, least));
            CHIP(hipStreamCreateWithPriority(&c->stream_aux, hipStreamNonBlocking, least));
            CHIP(hipStreamCreateWithPriority(&c->stream_io, hipStreamNonBlocking, least));
        } else {
            (void)hipGetLastError();
            CHIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            CHIP(hipStreamCreateWithFlags(&c->stream_cs, hipStreamNonBlocking));
            CHIP(hipStreamCreateWithFlags(&c->stream_aux, hipStreamNonBlocking));
            CHIP(hipStreamCreateWithFlags(&c->stream_io, hipStreamNonBlocking));
        }
        // experiment (SW_CS_CUS=<count>): confine the can_see sweeps to the first <count> compute units of
        // the CU mask, so that the round loop's kernels find the others free of polling waves
        if (const char* e = getenv("SW_CS_CUS")) {
            const int cus = atoi(e);
            if (cus > 0 && cus < 256) {
                uint32_t mask[8] = {0};
                for (int i = 0; i < cus; ++i) mask[i >> 5] |= 1u << (i & 31);
                hipStream_t s2 = nullptr;
                if (hipExtStreamCreateWithCUMask(&s2, 8, mask) == hipSuccess) { (void)hipStreamDestroy(c->stream_cs); c->stream_cs = s2; }
                else (void)hipGetLastError();
            }
        }
        CHIP(hipEventCreateWithFlags(&c->ev_payload, hipEventDisableTiming));
        CHIP(hipEventCreateWithFlags(&c->ev_small, hipEventDisableTiming));
    }
    {
        static_assert(2 * sizeof(RState) + sizeof(int) <= 256, "readback block header");
        const size_t tree_off = 256 + (size_t)c->npad * sizeof(int32_t);
        const size_t prov_off = tree_off + (size_t)c->npad * sizeof(int32_t);
        c->rb_bytes = prov_off + (size_t)SW_PROV_ROWS * (SW_MAX_CHUNKS + 1) * sizeof(unsigned);
        CHIP(hipMalloc((void**)&c->d_rb, c->rb_bytes));
        CHIP(hipMemset(c->d_rb, 0, c->rb_bytes));
        CHIP(hipHostMalloc((void**)&c->h_rb_all, c->rb_bytes * SW_PROV_ROWS, hipHostMallocDefault));
        c->h_rb = c->h_rb_all;
        for (int i = 0; i < SW_PROV_ROWS; ++i) {
            hipEvent_t e1, e2;
            CHIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            c->rb_events.push_back(e1);
            CHIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            c->shot_events.push_back(e2);
        }
        c->d_state = reinterpret_cast<RState*>(c->d_rb);
        c->d_flow_err = reinterpret_cast<int*>(c->d_rb + 2 * sizeof(RState));
        c->d_front.p = reinterpret_cast<int32_t*>(c->d_rb + 256);
        c->d_treecnt = reinterpret_cast<int32_t*>(c->d_rb + tree_off);
        c->d_prov = reinterpret_cast<unsigned*>(c->d_rb + prov_off);
    }
    CHIP(hipMalloc((void**)&c->d_err, sizeof(int)));

    const int np = c->npad;
    CCHK(dgrow(c, c->d_stake, np, 0));
    CHIP(hipMemcpy(c->d_stake.p, c->stake_h.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice));
    CCHK(dgrow(c, c->d_evalround, 2 * np, 0));
    CCHK(dgrow(c, c->d_evalpos, 2 * np, 0));
    CCHK(dgrow(c, c->d_lo_r, 2 * np, 0));
    CCHK(dgrow(c, c->d_cur, 2 * np, 0));
    CCHK(dgrow(c, c->d_unres, 2 * np, 0));
    CCHK(dgrow(c, c->d_lo_next, 2 * np, 0));
    CCHK(dgrow(c, c->d_pos_next, 2 * np, 0));
    CCHK(dgrow(c, c->d_found64, 2 * np, 0));
    CCHK(dgrow(c, c->d_farslot, 2 * np, 0));
    CCHK(dgrow(c, c->d_force, 2 * np, 0));
    CCHK(dgrow(c, c->d_cand, (size_t)2 * np * 64, 0));
    CCHK(dgrow(c, c->d_gallop, 2 * np, 0));
    CCHK(fill_i32(c, c->d_front.p, np, -1));
    c->divided_cnt.assign(np, 0);
    CHIP(hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming));
    CHIP(hipEventCreateWithFlags(&c->ev_cs_done, hipEventDisableTiming));
    CHIP(hipEventCreateWithFlags(&c->ev_main_mark, hipEventDisableTiming));
    CHIP(hipEventCreateWithFlags(&c->ev_loop_done, hipEventDisableTiming));
    CCHK(dgrow(c, c->d_chain_len, np, 0));
    CCHK(dgrow(c, c->d_chain_start, np, 0));
    CCHK(dgrow(c, c->d_chain_cnt, np, 0));
    // band masks; row 0 stays all-zero (the bit-sliced tally points "not a hop" at it), the band
    // proper starts at row 1
    CCHK(dgrow(c, c->d_Mb, ((size_t)c->MCAP + 1) * c->nw, 0));
    if (hipMemset(c->d_Mb.p, 0, (size_t)c->nw * sizeof(u64)) != hipSuccess) { sw_destroy(c); return SW_EIO; }
    CCHK(dgrow(c, c->d_Pc, (size_t)c->MCAP + 1, 0));
    if (hipMemset(c->d_Pc.p, 0, sizeof(int32_t)) != hipSuccess) { sw_destroy(c); return SW_EIO; }
    CCHK(fill_i32(c, c->d_evalround.p, 2 * np, -1));
    CCHK(fill_i32(c, c->d_evalpos.p, 2 * np, 0));
    CCHK(fill_i32(c, c->d_lo_r.p, 2 * np, SW_INF));
    CCHK(fill_i32(c, c->d_cur.p, 2 * np, 0));
    CCHK(fill_i32(c, c->d_lo_next.p, 2 * np, SW_INF));
    CCHK(fill_i32(c, c->d_pos_next.p, 2 * np, 0));
    CHIP(hipMemsetAsync(c->d_found64.p, 0xff, 2 * np * sizeof(u64), c->stream));
    CHIP(hipMemsetAsync(c->d_unres.p, 0, 2 * np * sizeof(int32_t), c->stream));
    CCHK(ensure_rounds(c, 256));
    CHIP(hipStreamSynchronize(c->stream));
#undef CCHK
#undef CHIP
    *out = c;
    return SW_OK;
}

static void split_dissolve(SplitGroup* g);

int sw_destroy(sw_ctx* c) {
    if (c && c->split) split_dissolve(c->split);
    if (!c) return SW_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->debug_timing && c->d_flow_dbg) {
        u64 d[8] = {0};
        (void)hipMemcpy(d, c->d_flow_dbg, sizeof d, hipMemcpyDeviceToHost);
        fprintf(stderr, "[sw] dataflow sweep, column 0: %llu events, %llu rows re-read from memory, %llu starved lane-trips, %llu wave-trips over %llu wave runs "
                "(%.1f trips per wave run), loader passes %llu (+%llu idle)\n", d[0], d[1], d[2], d[3], d[6], d[6] ? (double)d[3] / (double)d[6] : 0.0, d[4], d[5]);
        (void)hipFree(c->d_flow_dbg);
    }
    if (c->debug_timing && c->stage_calls > 0) {
        const double k = 1.0 / (double)c->stage_calls;
        fprintf(stderr, "[sw] sw_divide_rounds host stages, us per call over %lld calls: sweeps enqueued %.1f, loop set-up %.1f, "
                "round loop %.1f, front rows %.1f, aux launches %.1f, final syncs %.1f\n", (long long)c->stage_calls,
                c->stage_us[0] * k, c->stage_us[1] * k, c->stage_us[2] * k, c->stage_us[3] * k, c->stage_us[4] * k, c->stage_us[5] * k);
    }
    if (c->vm.active) { vm_destroy(c); c->d_L.p = nullptr; c->d_L.cap = 0; }
    dfree(c->d_ordpos);
    dfree(c->x_worder); dfree(c->x_wcnt); dfree(c->x_newr); dfree(c->x_queue); dfree(c->x_fw); dfree(c->x_items_ev); dfree(c->x_rounds);
    dfree(c->x_fam_ev); dfree(c->x_votes); dfree(c->x_tbd); dfree(c->x_sm); dfree(c->x_done); dfree(c->x_visited); dfree(c->x_sflag);
    dfree(c->x_white); dfree(c->x_times); dfree(c->x_tsort); dfree(c->x_items_ts); dfree(c->x_hdr);
    dfree(c->d_cr); dfree(c->d_sp); dfree(c->d_op); dfree(c->d_ht); dfree(c->d_seq); dfree(c->d_round); dfree(c->d_L);
    dfree(c->d_chain_ev); dfree(c->d_coin); dfree(c->d_sig); dfree(c->d_t); dfree(c->d_S); dfree(c->d_finlist);
    if (c->d_fin) (void)hipFree(c->d_fin);
    dfree(c->d_cdesc); dfree(c->d_bounds); dfree(c->d_cuts); dfree(c->d_chain_start); dfree(c->d_chain_cnt); dfree(c->d_chain_len); dfree(c->d_stake); dfree(c->d_lev_cnt); dfree(c->d_lev_start); dfree(c->d_lev_pin); dfree(c->d_lev_cback); dfree(c->d_lev_pinbase); dfree(c->d_lev_pos);
    dfree(c->d_lev_cursor); dfree(c->d_desc); dfree(c->d_lo); dfree(c->d_lopos); dfree(c->d_wit);
    dfree(c->d_fam); dfree(c->d_dec_call); dfree(c->d_dec_by); dfree(c->d_cons); dfree(c->d_newc); dfree(c->d_Sw); dfree(c->d_evalround);
    dfree(c->d_evalpos); dfree(c->d_lo_r); dfree(c->d_cur); dfree(c->d_unres); dfree(c->d_lo_next);
    if (c->d_dbg) (void)hipFree(c->d_dbg);
    if (c->d_dbg_blk) (void)hipFree(c->d_dbg_blk);
    dfree(c->d_pos_next); dfree(c->d_found64); dfree(c->d_rsc); dfree(c->d_farslot); dfree(c->d_force); dfree(c->d_cand); dfree(c->d_gallop); dfree(c->d_small); dfree(c->d_Mb); dfree(c->d_Pc);
    if (c->d_rb) (void)hipFree(c->d_rb);
    if (c->h_rb_all) (void)hipHostFree(c->h_rb_all);
    for (hipEvent_t e : c->rb_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->shot_events) (void)hipEventDestroy(e);
    if (c->h_fame) (void)hipHostFree(c->h_fame);
    if (c->d_err) (void)hipFree(c->d_err);
    dfree(c->d_ord_rounds); dfree(c->d_fwm); dfree(c->d_ordat); dfree(c->d_rowsum); dfree(c->d_grp); dfree(c->d_oblk);
    if (c->h_ord) (void)hipHostFree(c->h_ord);
    if (c->h_ord_stage) (void)hipHostFree(c->h_ord_stage);
    if (c->d_split) (void)hipFree(c->d_split);
    c->transactions.release();
    if (c->stream_ord) (void)hipStreamDestroy(c->stream_ord);
    for (auto st : c->stream_srt) if (st) (void)hipStreamDestroy(st);
    for (auto e : c->ord_events) (void)hipEventDestroy(e);
    dfree(c->d_q); dfree(c->d_acc_ev); dfree(c->d_acc_ri); dfree(c->d_ts);
    dfree(c->d_sorted); dfree(c->d_hostflag); dfree(c->d_white);
    dfree(c->d_big_ri); dfree(c->d_sk_ev); dfree(c->d_big_off); dfree(c->d_sk_ts); dfree(c->d_sk_k8);
    dfree(c->d_fd); dfree(c->d_seg); dfree(c->d_tch);
    dfree(c->d_rbnd); dfree(c->d_rcuts);
    if (c->d_rprov) (void)hipFree(c->d_rprov);
    if (c->ev_user) (void)hipEventDestroy(c->ev_user);
    for (int g = 0; g < 4; ++g) {
        if (c->loop_exec[g]) (void)hipGraphExecDestroy(c->loop_exec[g]);
        if (c->loop_graph[g]) (void)hipGraphDestroy(c->loop_graph[g]);
    }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->cs_events) (void)hipEventDestroy(e);
    if (c->stream_io) { (void)hipStreamSynchronize(c->stream_io); (void)hipStreamDestroy(c->stream_io); }
    if (c->ev_payload) (void)hipEventDestroy(c->ev_payload);
    if (c->ev_aux_done) (void)hipEventDestroy(c->ev_aux_done);
    if (c->ev_cs_done) (void)hipEventDestroy(c->ev_cs_done);
    if (c->ev_main_mark) (void)hipEventDestroy(c->ev_main_mark);
    if (c->ev_bounds) (void)hipEventDestroy(c->ev_bounds);
    if (c->ev_loop_done) (void)hipEventDestroy(c->ev_loop_done);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->ev_small) (void)hipEventDestroy(c->ev_small);
    if (c->stream_cs) (void)hipStreamDestroy(c->stream_cs);
    if (c->stream_aux) (void)hipStreamDestroy(c->stream_aux);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SW_OK;
}

int sw_reserve(sw_ctx* c, int64_t n_events) {
    if (!c || n_events < 0) return SW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    return ensure_events(c, n_events);
}

int64_t sw_num_events(const sw_ctx* c) { return c ? c->N : 0; }

// Parallel copy of `bytes` bytes (bulk appends stage t / sig in pinned memory so that the upload
// can run behind the call; one thread cannot saturate the host memory system).
static void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<size_t>(std::max(1u, std::min(hw / 2, 8u)), bytes / (4u << 20) + 1);
    if (nt <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 63) & ~(size_t)63;
    for (int i = 0; i < nt; ++i) {
        const size_t a0 = std::min(bytes, per * i), a1 = std::min(bytes, per * (i + 1));
        if (a1 > a0) th.emplace_back([=] { memcpy((char*)dst + a0, (const char*)src + a0, a1 - a0); });
    }
    for (auto& t : th) t.join();
}

// ---- exact path for forked hashgraphs (exact.hip.h) ----------------------------------------------
// The context's own tables are the state (can_see table, round array, witness / fame / consensus
// tables); what the reference keeps beyond the fast path's per-slot view is added here: the dict order
// of every round's witnesses, fame per EVENT, tbd.  Every call is one launch of one wavefront.
swx::State exact_state(sw_ctx* c) {
    swx::State s{};
    s.n = c->n; s.npad = c->npad; s.coin_period = c->coin_period; s.tot = c->tot; s.stake = c->d_stake.p;
    s.cr = c->d_cr.p; s.sp = c->d_sp.p; s.op = c->d_op.p; s.ht = c->d_ht.p; s.t = c->d_t.p; s.sig = c->d_sig.p;
    s.round = c->d_round.p; s.L = c->d_L.p; s.tbd = c->x_tbd.p; s.fam_ev = c->x_fam_ev.p;
    s.Rcap = c->Rcap; s.wit = c->d_wit.p; s.worder = c->x_worder.p; s.wcnt = c->x_wcnt.p; s.cons = c->d_cons.p;
    s.fam_slot = c->d_fam.p; s.hdr = c->x_hdr.p;
    return s;
}

// per-event and per-round buffers of the exact path follow the context's capacities
int exact_grow(sw_ctx* c) {
    const size_t np = c->npad;
    if (c->x_cap < c->cap) {
        const size_t keep = (size_t)c->x_cap, nc = (size_t)c->cap;
        CHK(dgrow(c, c->x_fam_ev, nc, keep));
        CHK(dgrow(c, c->x_tbd, nc, keep));
        CHK(dgrow(c, c->x_queue, nc + 1, 0));
        CHK(dgrow(c, c->x_visited, nc + 1, 0));
        CHK(dgrow(c, c->x_items_ev, nc + 1, 0));
        CHK(dgrow(c, c->x_items_ts, nc + 1, 0));
        HIPCHK(c, hipMemsetAsync(c->x_fam_ev.p + keep, 0xff, nc - keep, c->stream));
        HIPCHK(c, hipMemsetAsync(c->x_tbd.p + keep, 1, nc - keep, c->stream));
        HIPCHK(c, hipMemsetAsync(c->x_visited.p, 0, c->x_visited.cap, c->stream));  // all zero between calls
        c->x_cap = c->cap;
    }
    if (c->x_Rcap < c->Rcap) {
        const size_t keep = (size_t)c->x_Rcap, nc = (size_t)c->Rcap;
        CHK(dgrow(c, c->x_worder, nc * np, keep * np));
        CHK(dgrow(c, c->x_wcnt, nc, keep));
        CHK(dgrow(c, c->x_done, nc, 0));
        CHK(dgrow(c, c->x_newr, nc, 0));
        HIPCHK(c, hipMemsetAsync(c->x_wcnt.p + keep, 0, (nc - keep) * sizeof(int32_t), c->stream));
        c->x_Rcap = c->Rcap;
    }
    return SW_OK;
}

int exact_read_hdr(sw_ctx* c, long long* hdr) {
    HIPCHK(c, hipMemcpyAsync(hdr, c->x_hdr.p, swx::H_WORDS * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

int exact_set_hdr(sw_ctx* c, int slot, long long v) {
    HIPCHK(c, hipMemcpyAsync(c->x_hdr.p + slot, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

// The first forked event arrives: hand the fast path's state over (nothing of it is recomputed).
int exact_enter(sw_ctx* c) {
    if (c->vm.active) return fail(c, SW_ENOTSUP, "a forked event in windowed mode: the exact path needs every can_see row");
    HIPCHK(c, hipDeviceSynchronize());
    c->payload_pending = false; c->small_pending = false;
    CHK(ensure_dag_h(c));  // host sp / op / height complete, heights on the device
    CHK(ensure_rounds(c, std::max(c->max_height + 3, c->R + 1)));
    const size_t np = c->npad;
    CHK(dgrow(c, c->x_hdr, (size_t)swx::H_WORDS, 0));
    CHK(dgrow(c, c->x_sm, np, 0)); CHK(dgrow(c, c->x_fw, np, 0)); CHK(dgrow(c, c->x_sflag, np, 0));
    CHK(dgrow(c, c->x_times, np, 0)); CHK(dgrow(c, c->x_tsort, np, 0)); CHK(dgrow(c, c->x_white, 64, 0));
    CHK(exact_grow(c));
    long long hdr[swx::H_WORDS] = {0};
    hdr[swx::H_R] = c->R;
    HIPCHK(c, hipMemcpyAsync(c->x_hdr.p, hdr, sizeof hdr, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->x_wcnt.p, 0, (size_t)c->x_Rcap * sizeof(int32_t), c->stream));
    c->x_fc_base = c->fc_seen;
    const long long n_tx = (long long)c->transactions.size();
    if (n_tx) HIPCHK(c, hipMemcpyAsync(c->x_items_ev.p, c->transactions.data(), (size_t)n_tx * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(k_exact_import, dim3(1), dim3(64), 0, c->stream, exact_state(c), (long long)c->N, (const int*)c->x_items_ev.p, n_tx);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->exact = true;
    return SW_OK;
}

// Node.add_event for a context on the exact path (swirld.py:114-120) + the structural half of
// is_valid_event (swirld.py:104-108: 0 or 2 parents, both known, self-parent by the creator,
// other-parent by another member).  Nothing is stored before the whole batch is accepted.
int append_exact(sw_ctx* c, int64_t K, const int32_t* creator, const int32_t* self_parent, const int32_t* other_parent,
                 const double* t, const uint8_t* sig64) {
    const int64_t N0 = c->N;
    const int n = c->n;
    for (int64_t i = 0; i < K; ++i) {
        const int64_t e = N0 + i;
        const int32_t m = creator[i], s = self_parent[i], o = other_parent[i];
        if (m < 0 || m >= n) return fail(c, SW_EINVAL, "event %lld: creator %d out of range", (long long)e, m);
        if ((s < 0) != (o < 0)) return fail(c, SW_EINVAL, "event %lld: must have 0 or 2 parents", (long long)e);
        if (s >= e || o >= e) return fail(c, SW_EINVAL, "event %lld: parent index not earlier (not a topological order)", (long long)e);
        if (s >= 0) {
            if ((s < N0 ? c->cr[s] : creator[s - N0]) != m) return fail(c, SW_EINVAL, "event %lld: self-parent is by another member", (long long)e);
            if ((o < N0 ? c->cr[o] : creator[o - N0]) == m) return fail(c, SW_EINVAL, "event %lld: other-parent is by the same member", (long long)e);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->exact) CHK(exact_enter(c));
    CHK(ensure_events(c, N0 + K));
    CHK(exact_grow(c));
    std::vector<int32_t> hnew(K);
    int hmax = c->max_height;
    for (int64_t i = 0; i < K; ++i) {  // swirld.py:117-120
        const int32_t s = self_parent[i], o = other_parent[i];
        if (s < 0) { hnew[i] = 0; continue; }
        const int32_t hs = s < N0 ? c->ht[s] : hnew[s - N0], ho = o < N0 ? c->ht[o] : hnew[o - N0];
        hnew[i] = std::max(hs, ho) + 1;
        hmax = std::max(hmax, hnew[i]);
    }
    const size_t b4 = (size_t)K * sizeof(int32_t);
    HIPCHK(c, hipMemcpyAsync(c->d_cr.p + N0, creator, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_sp.p + N0, self_parent, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_op.p + N0, other_parent, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ht.p + N0, hnew.data(), b4, hipMemcpyHostToDevice, c->stream));
    if (t) HIPCHK(c, hipMemcpyAsync(c->d_t.p + N0, t, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemsetAsync(c->d_t.p + N0, 0, (size_t)K * 8, c->stream));
    if (sig64) HIPCHK(c, hipMemcpyAsync(c->d_sig.p + (size_t)N0 * 64, sig64, (size_t)K * 64, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemsetAsync(c->d_sig.p + (size_t)N0 * 64, 0, (size_t)K * 64, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_round.p + N0, 0xff, b4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->x_fam_ev.p + N0, 0xff, (size_t)K, c->stream));
    HIPCHK(c, hipMemsetAsync(c->x_tbd.p + N0, 1, (size_t)K, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cr.insert(c->cr.end(), creator, creator + K);
    c->sp.insert(c->sp.end(), self_parent, self_parent + K);
    c->op.insert(c->op.end(), other_parent, other_parent + K);
    c->ht.insert(c->ht.end(), hnew.begin(), hnew.end());
    c->max_height = hmax;
    c->N = N0 + K;
    return SW_OK;
}

int exact_divide(sw_ctx* c, int64_t first, int64_t K) {
    CHK(ensure_rounds(c, c->max_height + 3));  // round <= height: every row a witness can land in exists
    CHK(exact_grow(c));
    CHK(exact_set_hdr(c, swx::H_RC, 0));
    for (int64_t a = first; a < first + K; a += 8192) {  // (bounded launches: one wavefront walks the events in order)
        const int64_t k = std::min<int64_t>(8192, first + K - a);
        hipLaunchKernelGGL(k_exact_divide, dim3(1), dim3(64), 0, c->stream, exact_state(c), (long long)a, (long long)k);
        c->ctr.kernel_launches++;
    }
    HIPCHK(c, hipGetLastError());
    long long hdr[swx::H_WORDS];
    CHK(exact_read_hdr(c, hdr));
    if (hdr[swx::H_RC] != swx::X_OK) { c->poisoned = true; return fail(c, SW_EIO, "exact divide_rounds: a parent of event %lld has no round", hdr[swx::H_ERR_AT]); }
    c->R = (int)hdr[swx::H_R];
    c->divided = first + K;
    c->ctr.events_divided += K;
    c->ctr.rounds = c->R;
    return SW_OK;
}

int exact_fame(sw_ctx* c, int32_t* new_rounds, int cap, int* n_new) {
    const int R = c->R;
    const int max_c = first_undecided_round(c);
    swx::FameScratch x{};
    x.win = std::max(1, R - max_c);
    x.layer = (size_t)c->n * x.win * c->n;
    if (x.layer > ((size_t)4 << 30)) return fail(c, SW_ENOMEM, "exact decide_fame: %d undecided rounds x %d members need %zu GB of vote storage", x.win, c->n, (2 * x.layer) >> 30);
    CHK(exact_grow(c));
    CHK(dgrow(c, c->x_votes, 2 * x.layer, 0));
    x.votes = c->x_votes.p; x.s_m = c->x_sm.p; x.done = c->x_done.p; x.new_rounds = c->x_newr.p;
    CHK(exact_set_hdr(c, swx::H_RC, 0));
    CHK(exact_set_hdr(c, swx::H_NNEW, 0));
    hipLaunchKernelGGL(k_exact_fame, dim3(1), dim3(64), 0, c->stream, exact_state(c), x);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    long long hdr[swx::H_WORDS];
    CHK(exact_read_hdr(c, hdr));
    if (hdr[swx::H_RC] != swx::X_OK)
        return fail(c, SW_EINVAL, "exact decide_fame: a voter strongly sees a member without a witness in the previous round, or a vote is missing (KeyError in the reference, swirld.py:253 / 260)");
    const int cnt = (int)hdr[swx::H_NNEW];
    std::vector<int32_t> nr(std::max(cnt, 1));
    if (cnt) {
        HIPCHK(c, hipMemcpyAsync(nr.data(), c->x_newr.p, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < cnt; ++i) {
        c->cons_h[nr[i]] = 1;
        if (i < cap && new_rounds) new_rounds[i] = nr[i];
    }
    if (n_new) *n_new = cnt;
    FameCounters fc{};
    fc.voter_evals = (u64)hdr[swx::H_VOTER_EVALS] + c->x_fc_base.voter_evals;
    fc.majority_evals = (u64)hdr[swx::H_MAJ_EVALS] + c->x_fc_base.majority_evals;
    fc.coin_votes = (u64)hdr[swx::H_COIN_VOTES] + c->x_fc_base.coin_votes;
    fc.coin_flips = (u64)hdr[swx::H_COIN_FLIPS] + c->x_fc_base.coin_flips;
    fame_counters(c, fc);
    if (cnt > cap) return fail(c, SW_ERANGE, "new_rounds capacity %d < %d", cap, cnt);
    return SW_OK;
}

int exact_order(sw_ctx* c, std::vector<int32_t> rounds, int32_t* out_events, int64_t cap, int64_t* n_out) {
    std::sort(rounds.begin(), rounds.end());  // sorted(new_c), swirld.py:283
    rounds.erase(std::unique(rounds.begin(), rounds.end()), rounds.end());
    const int nr = (int)rounds.size();
    if (nr == 0) return SW_OK;
    if (rounds.front() < 0 || rounds.back() >= c->R)
        return fail(c, SW_ERANGE, "find_order: round outside [0, %d) (KeyError in the reference)", c->R);
    CHK(exact_grow(c));
    CHK(dgrow(c, c->x_rounds, nr, 0));
    HIPCHK(c, hipMemcpyAsync(c->x_rounds.p, rounds.data(), (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CHK(exact_set_hdr(c, swx::H_RC, 0));
    CHK(exact_set_hdr(c, swx::H_NOUT, 0));
    swx::OrderScratch x{};
    x.queue = c->x_queue.p; x.visited = c->x_visited.p; x.fw = c->x_fw.p; x.sflag = c->x_sflag.p; x.times = c->x_times.p;
    x.tsort = c->x_tsort.p; x.white = c->x_white.p; x.items_ev = c->x_items_ev.p; x.items_ts = c->x_items_ts.p;
    hipLaunchKernelGGL(k_exact_order, dim3(1), dim3(64), 0, c->stream, exact_state(c), x, (const int*)c->x_rounds.p, nr);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipGetLastError());
    long long hdr[swx::H_WORDS];
    CHK(exact_read_hdr(c, hdr));
    if (hdr[swx::H_RC] == swx::X_EINDEX)
        return fail(c, SW_ERANGE, "find_order: an event is seen by a single famous witness (IndexError at swirld.py:305)");
    if (hdr[swx::H_RC] != swx::X_OK)
        return fail(c, SW_EINVAL, "find_order: a round has an undecided witness (KeyError on self.famous[w], swirld.py:284)");
    const int64_t produced = hdr[swx::H_NOUT];
    const size_t at = c->transactions.size();
    if (!c->transactions.resize(at + (size_t)produced)) return fail(c, SW_ENOMEM, "pinned host memory for the ordered events");
    if (produced) {
        HIPCHK(c, hipMemcpyAsync(c->transactions.data() + at, c->x_items_ev.p, (size_t)produced * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (n_out) *n_out = produced;
    if (produced > cap) return fail(c, SW_ERANGE, "find_order: out_events capacity %lld < %lld", (long long)cap, (long long)produced);
    if (out_events) memcpy(out_events, c->transactions.data() + at, (size_t)produced * sizeof(int32_t));
    return SW_OK;
}

// state of the exact path back to "nothing divided" (sw_rewind; the events and their forks stay)
int exact_rewind(sw_ctx* c) {
    if (c->N) {
        HIPCHK(c, hipMemsetAsync(c->x_fam_ev.p, 0xff, (size_t)c->N, c->stream));
        HIPCHK(c, hipMemsetAsync(c->x_tbd.p, 1, (size_t)c->N, c->stream));
    }
    if (c->x_Rcap) HIPCHK(c, hipMemsetAsync(c->x_wcnt.p, 0, (size_t)c->x_Rcap * sizeof(int32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->x_hdr.p, 0, swx::H_WORDS * sizeof(long long), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->x_fc_base = c->fc_seen;
    return SW_OK;
}

// A Node's gossip step appends a handful of events: one packed record per event in pinned memory,
// ONE host-to-device copy, ONE kernel (k_ingest_small), no host synchronisation.  The batch has been
// validated; the per-member tables of the caller (head_t, nev_t, first_t) are committed here.
static int append_small(sw_ctx* c, int64_t K, const int32_t* creator, const int32_t* self_parent, const int32_t* other_parent,
                        const double* t, const uint8_t* sig64, const std::vector<int32_t>& seq, std::vector<int32_t>& head_t,
                        std::vector<int32_t>& nev_t, std::vector<int32_t>& first_t) {
    const int64_t N0 = c->N;
    const int n = c->n, np = c->npad;
    CHK(ensure_dag_h(c));  // parents' heights (complete already unless a bulk append came before)
    if ((size_t)K > c->h_small_cap) {
        if (c->small_pending) { HIPCHK(c, hipEventSynchronize(c->ev_small)); c->small_pending = false; }
        if (c->h_small) (void)hipHostFree(c->h_small);
        c->h_small = nullptr; c->h_small_cap = 0;
        const size_t cap = std::max<size_t>(256, (size_t)K * 2);
        if (hipHostMalloc((void**)&c->h_small, cap * sizeof(SmallRec), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, SW_ENOMEM, "hipHostMalloc for the small-append staging buffer failed");
        }
        c->h_small_cap = cap;
    }
    // ---- segments that overflow move to the end of the pool, doubled (device-to-device)
    bool moved = false;
    for (int m = 0; m < n; ++m) {
        if (nev_t[m] <= c->chain_cap[m]) continue;
        int32_t ncap = std::max(16, 2 * c->chain_cap[m]);
        while (ncap < nev_t[m]) ncap *= 2;
        if (c->pool_used + ncap > 0x7fffffff) return fail(c, SW_ERANGE, "chain pool exceeds 2^31 entries");
        const int64_t noff = c->pool_used;
        const int32_t before = c->nev[m];
        CHK(dgrow(c, c->d_chain_ev, (size_t)(noff + ncap) * 2, (size_t)c->pool_used));
        CHK(dgrow(c, c->d_cdesc, c->d_chain_ev.cap, (size_t)c->pool_used));
        if (before) {
            HIPCHK(c, hipMemcpyAsync(c->d_chain_ev.p + noff, c->d_chain_ev.p + c->chain_start_h[m], (size_t)before * sizeof(int32_t),
                                     hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_cdesc.p + noff, c->d_cdesc.p + c->chain_start_h[m], (size_t)before * sizeof(int4),
                                     hipMemcpyDeviceToDevice, c->stream));
        }
        if (c->pool_h_valid) {
            c->chain_ev_h.resize((size_t)(noff + ncap), -1);
            std::copy(c->chain_ev_h.begin() + c->chain_start_h[m], c->chain_ev_h.begin() + c->chain_start_h[m] + before, c->chain_ev_h.begin() + noff);
        }
        c->chain_start_h[m] = (int32_t)noff;
        c->chain_cap[m] = ncap;
        c->pool_used = noff + ncap;
        moved = true;
    }
    if (moved) {
        HIPCHK(c, hipMemcpyAsync(c->d_chain_start.p, c->chain_start_h.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (rare; keeps the host table free to change)
    }
    // ---- packed records
    if (c->small_pending) { HIPCHK(c, hipEventSynchronize(c->ev_small)); c->small_pending = false; }
    const bool dag_complete = (int64_t)c->sp.size() == N0;  // always, after ensure_dag_h
    if (!dag_complete) return fail(c, SW_EIO, "host mirror of the hashgraph is incomplete (internal)");
    c->cr.reserve(N0 + K); c->sp.reserve(N0 + K); c->op.reserve(N0 + K); c->ht.reserve(N0 + K); c->seq_h.reserve(N0 + K);
    c->blk_hmin.resize((size_t)((N0 + K + 4095) >> 12), 0x7fffffff);
    c->blk_hmax.resize((size_t)((N0 + K + 4095) >> 12), -1);
    const bool sig_complete = (int64_t)(c->sig_h.size() / 64) == N0;
    if (sig_complete) c->sig_h.resize((size_t)(N0 + K) * 64);
    if (c->pool_h_valid && c->chain_ev_h.size() < (size_t)c->pool_used) c->chain_ev_h.resize((size_t)c->pool_used, -1);
    for (int64_t i = 0; i < K; ++i) {
        const int64_t e = N0 + i;
        SmallRec& r = c->h_small[i];
        const int32_t m = creator[i], s_ = self_parent[i], o_ = other_parent[i];
        r.cr = m; r.sp = s_; r.op = o_; r.seq = seq[i];
        const int32_t h = s_ < 0 ? 0 : std::max(c->ht[s_], c->ht[o_]) + 1;  // swirld.py:117-120 (parents are earlier: already mirrored)
        r.ht = h;
        r.at = c->chain_start_h[m] + seq[i];
        r.w = o_ < 0 ? 0 : (c->cr[o_] | ((c->seq_h[o_] & 63) << 10));
        r.pad = 0;
        r.t = t ? t[i] : 0.0;
        if (sig64) memcpy(r.sig, sig64 + 64 * i, 64); else memset(r.sig, 0, 64);
        // host mirrors (a later event of this batch may have this one as a parent)
        c->cr.push_back(m); c->sp.push_back(s_); c->op.push_back(o_); c->ht.push_back(h); c->seq_h.push_back(seq[i]);
        c->max_height = std::max(c->max_height, h);
        c->blk_hmin[e >> 12] = std::min(c->blk_hmin[e >> 12], h);
        c->blk_hmax[e >> 12] = std::max(c->blk_hmax[e >> 12], h);
        if (sig_complete) memcpy(c->sig_h.data() + (size_t)e * 64, r.sig, 64);
        if (c->pool_h_valid) c->chain_ev_h[(size_t)r.at] = (int32_t)e;
    }
    // (from here on the host tables are committed; a device failure poisons the context)
    c->head.swap(head_t);
    c->first_ev.swap(first_t);
    c->nev.swap(nev_t);
    c->N = N0 + K;
    CHK(dgrow(c, c->d_small, (size_t)c->h_small_cap * sizeof(SmallRec), 0));
    hipError_t e1 = hipMemcpyAsync(c->d_small.p, c->h_small, (size_t)K * sizeof(SmallRec), hipMemcpyHostToDevice, c->stream);
    if (e1 == hipSuccess) {
        hipLaunchKernelGGL(k_ingest_small, dim3((unsigned)((K + 127) / 128)), dim3(128), 0, c->stream, (const SmallRec*)c->d_small.p, (int)N0, (int)K,
                           c->d_cr.p, c->d_sp.p, c->d_op.p, c->d_seq.p, c->d_ht.p, c->d_t.p, c->d_sig.p, c->d_coin.p, c->d_round.p,
                           c->d_chain_ev.p, c->d_cdesc.p, c->d_chain_cnt.p);
        c->ctr.kernel_launches++;
        e1 = hipGetLastError();
    }
    if (e1 == hipSuccess) e1 = hipEventRecord(c->ev_small, c->stream);
    if (e1 != hipSuccess) { c->poisoned = true; return fail(c, SW_EIO, "small append: %s", hipGetErrorString(e1)); }
    c->small_pending = true;
    return SW_OK;
}

int sw_append_events(sw_ctx* c, int64_t K, const int32_t* creator, const int32_t* self_parent,
                     const int32_t* other_parent, const double* t, const uint8_t* sig64) {
    if (!c) return SW_EINVAL;
    if (c->poisoned) return fail(c, SW_EIO, "context unusable after an earlier device failure");
    if (K < 0 || (K > 0 && (!creator || !self_parent || !other_parent))) return fail(c, SW_EINVAL, "NULL event arrays");
    if (K == 0) return SW_OK;
    if (c->N + K > 0x7ffffff0ll) return fail(c, SW_ERANGE, "more than 2^31 events");
    if (c->exact) return append_exact(c, K, creator, self_parent, other_parent, t, sig64);
    const int64_t N0 = c->N;
    const int n = c->n;
    const bool bulk = K >= 8192 || c->chain_cap.empty() || K * 8 >= N0;
    // ---- 1. host pass: everything of is_valid_event's structural half (swirld.py:104-108) that needs
    // only per-member tables, on COPIES of them — nothing is stored before the whole batch is accepted.
    // Arity; topological order; self-parent = the creator's latest event (which makes it an event
    // of the same creator and refuses forks: the round-synchronous path needs one self-parent chain
    // per member, and a stored fork would make every later sw_divide_rounds fail); chain positions.
    std::vector<int32_t> head_t(c->head), nev_t(c->nev), first_t(c->first_ev), seq(K);
    for (int64_t i = 0; i < K; ++i) {
        const int64_t e = N0 + i;
        const int32_t m = creator[i], s = self_parent[i], o = other_parent[i];
        if (m < 0 || m >= n) return fail(c, SW_EINVAL, "event %lld: creator %d out of range", (long long)e, m);
        if ((s < 0) != (o < 0)) return fail(c, SW_EINVAL, "event %lld: must have 0 or 2 parents", (long long)e);
        if (s >= e || o >= e) return fail(c, SW_EINVAL, "event %lld: parent index not earlier (not a topological order)", (long long)e);
        if (!c->lapsed.empty() && c->lapsed[m])
            return fail(c, SW_ERANGE, "event %lld: member %d was silent for more than %lld events and has lapsed out of the windowed table (sw_set_window_lapse)",
                        (long long)e, m, (long long)c->window_lapse);
        if (s >= 0 && (s < c->first_resident || o < c->first_resident))
            return fail(c, SW_ERANGE, "event %lld: the can_see row of a parent was evicted (windowed mode keeps rows from event %lld on)", (long long)e, (long long)c->first_resident);
        if (head_t[m] != s) {
            if (s >= 0 && (s < N0 ? c->cr[s] : creator[s - N0]) != m)
                return fail(c, SW_EINVAL, "event %lld: self-parent is by another member", (long long)e);
            // a fork (the reference stores it, README.md:84): the context moves to the exact path, which
            // validates the batch again on its own terms
            if (c->forks_mode == 1 && !c->vm.active) return append_exact(c, K, creator, self_parent, other_parent, t, sig64);
            return fail(c, SW_ENOTSUP, "event %lld is a fork (member %d already has %s): forked hashgraphs are outside the "
                        "supported domain; nothing was stored", (long long)e, m, s < 0 ? "a root" : "a later event on that self-parent");
        }
        head_t[m] = (int32_t)e;
        if (first_t[m] < 0) first_t[m] = (int32_t)e;
        seq[i] = nev_t[m]++;
    }
    if (!bulk) {  // small append: the other-parent's creator from the host mirror
        for (int64_t i = 0; i < K; ++i) {
            const int32_t o = other_parent[i];
            if (o >= 0 && (o < N0 ? c->cr[o] : creator[o - N0]) == creator[i])
                return fail(c, SW_EINVAL, "event %lld: other-parent is by the same member", (long long)(N0 + i));
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    // ---- 2. allocations (still nothing committed)
    CHK(ensure_events(c, N0 + K));
    if (c->vm.active) CHK(vm_ensure(c, (size_t)(N0 + K) * c->npad * sizeof(int32_t)));
    if (!bulk) return append_small(c, K, creator, self_parent, other_parent, t, sig64, seq, head_t, nev_t, first_t);
    const size_t b4 = (size_t)K * sizeof(int32_t);
    const bool stage = bulk && (t || sig64) && (size_t)K * 72 <= ((size_t)1 << 30);  // (beyond 1 GiB of payload: plain copies)
    if (stage) {
        const size_t need = (size_t)K * 72;
        if (c->payload_pending) { HIPCHK(c, hipEventSynchronize(c->ev_payload)); c->payload_pending = false; }
        if (need > c->h_pin_cap) {
            if (c->h_pin) (void)hipHostFree(c->h_pin);
            c->h_pin = nullptr; c->h_pin_cap = 0;
            if (hipHostMalloc((void**)&c->h_pin, need, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, SW_ENOMEM, "hipHostMalloc(%zu bytes) for the ingest staging buffer failed", need);
            }
            c->h_pin_cap = need;
        }
    }
    // ---- 3. parent arrays to the (uncommitted) tail of the device arrays; bulk: device half of the validation
    HIPCHK(c, hipMemcpyAsync(c->d_cr.p + N0, creator, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_sp.p + N0, self_parent, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_op.p + N0, other_parent, b4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_seq.p + N0, seq.data(), b4, hipMemcpyHostToDevice, c->stream));
    if (bulk) {
        int verdict = 0x7fffffff;
        HIPCHK(c, hipMemcpyAsync(c->d_err, &verdict, sizeof verdict, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_validate_other_parent, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, c->stream,
                           (const int*)c->d_cr.p, (const int*)c->d_op.p, (int)N0, (int)K, c->d_err);
        c->ctr.kernel_launches++;
        HIPCHK(c, hipMemcpyAsync(&verdict, c->d_err, sizeof verdict, hipMemcpyDeviceToHost, c->stream));
        // the payload is staged while the device validates
        if (stage) {
            if (t) parallel_memcpy(c->h_pin, t, (size_t)K * 8);
            if (sig64) parallel_memcpy(c->h_pin + (size_t)K * 8, sig64, (size_t)K * 64);
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (verdict != 0x7fffffff)
            return fail(c, SW_EINVAL, "event %d: other-parent is by the same member", verdict);
    }
    // ---- 4. commit.  From here on a device failure leaves the context inconsistent: it is poisoned.
#define PCHK(expr) do { int rc_ = (expr); if (rc_ != SW_OK) { c->poisoned = true; return rc_; } } while (0)
#define PHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->poisoned = true; \
        return fail(c, SW_EIO, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)
    c->cr.insert(c->cr.end(), creator, creator + K);
    c->seq_h.insert(c->seq_h.end(), seq.begin(), seq.end());
    // the lazily fetched mirrors stay complete when they were complete (a Node appends a few events per
    // call); after a bulk append they are refilled from the device on demand
    if (!bulk && (int64_t)c->sp.size() == N0) {
        c->sp.insert(c->sp.end(), self_parent, self_parent + K);
        c->op.insert(c->op.end(), other_parent, other_parent + K);
        c->ht.resize(N0 + K);
        c->blk_hmin.resize((size_t)((N0 + K + 4095) >> 12), 0x7fffffff);
        c->blk_hmax.resize((size_t)((N0 + K + 4095) >> 12), -1);
        for (int64_t e = N0; e < N0 + K; ++e) {
            const int32_t s_ = c->sp[e], o_ = c->op[e];
            const int32_t h = s_ < 0 ? 0 : std::max(c->ht[s_], c->ht[o_]) + 1;  // swirld.py:117-120
            c->ht[e] = h;
            c->max_height = std::max(c->max_height, h);
            c->blk_hmin[e >> 12] = std::min(c->blk_hmin[e >> 12], h);
            c->blk_hmax[e >> 12] = std::max(c->blk_hmax[e >> 12], h);
        }
        PHIP(hipMemcpyAsync(c->d_ht.p + N0, c->ht.data() + N0, b4, hipMemcpyHostToDevice, c->stream));
    }
    if (!bulk && (int64_t)(c->sig_h.size() / 64) == N0) {
        c->sig_h.resize((size_t)(N0 + K) * 64);
        if (sig64) memcpy(c->sig_h.data() + (size_t)N0 * 64, sig64, (size_t)K * 64);
        else memset(c->sig_h.data() + (size_t)N0 * 64, 0, (size_t)K * 64);
    }
    c->head.swap(head_t);
    c->first_ev.swap(first_t);
    c->N = N0 + K;
    // payload: timestamps, signatures, coin bits (swirld.py:272) — needed by decide_fame's coin rounds
    // and by find_order only, so a bulk append uploads them behind the call on their own stream
    hipStream_t ps = stage ? c->stream_io : c->stream;
    if (stage) {
        if (t) PHIP(hipMemcpyAsync(c->d_t.p + N0, c->h_pin, (size_t)K * 8, hipMemcpyHostToDevice, ps));
        if (sig64) PHIP(hipMemcpyAsync(c->d_sig.p + (size_t)N0 * 64, c->h_pin + (size_t)K * 8, (size_t)K * 64, hipMemcpyHostToDevice, ps));
    } else {
        if (t) PHIP(hipMemcpyAsync(c->d_t.p + N0, t, (size_t)K * 8, hipMemcpyHostToDevice, ps));
        if (sig64) PHIP(hipMemcpyAsync(c->d_sig.p + (size_t)N0 * 64, sig64, (size_t)K * 64, hipMemcpyHostToDevice, ps));
    }
    if (!t) PHIP(hipMemsetAsync(c->d_t.p + N0, 0, (size_t)K * 8, ps));
    if (!sig64) PHIP(hipMemsetAsync(c->d_sig.p + (size_t)N0 * 64, 0, (size_t)K * 64, ps));
    hipLaunchKernelGGL(k_coin_bits, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ps, (const unsigned char*)c->d_sig.p, (int)N0, (int)K, c->d_coin.p);
    c->ctr.kernel_launches++;
    if (stage) { PHIP(hipEventRecord(c->ev_payload, ps)); c->payload_pending = true; }
    PHIP(hipMemsetAsync(c->d_round.p + N0, 0xff, b4, c->stream));
    // per-member chain pool + chain descriptors: part of the ingest-time layout of the hashgraph store
    PCHK(rebuild_chains(c, nev_t, c->N));
    c->nev.swap(nev_t);
    PHIP(hipStreamSynchronize(c->stream));  // caller buffers may be released on return (bulk payload: staged copy)
#undef PCHK
#undef PHIP
    return SW_OK;
}

int sw_divide_rounds(sw_ctx* c, int64_t first, int64_t K) {
    if (!c) return SW_EINVAL;
    if (K < 0 || first < 0 || first + K > c->N) return fail(c, SW_ERANGE, "events [%lld, %lld) outside the stored hashgraph", (long long)first, (long long)(first + K));
    if (first != c->divided) return fail(c, SW_EINVAL, "divide_rounds must continue at event %lld (got %lld): every event is divided once, in order", (long long)c->divided, (long long)first);
    if (K == 0) return SW_OK;
    if (c->poisoned) return fail(c, SW_EIO, "context unusable after an earlier device failure");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->exact) return exact_divide(c, first, K);
    switch (c->nw) {
        case 1: return do_divide<1>(c, first, K);
        case 2: return do_divide<2>(c, first, K);
        case 4: return do_divide<4>(c, first, K);
        case 8: return do_divide<8>(c, first, K);
        case 16: return do_divide<16>(c, first, K);
    }
    return fail(c, SW_EINVAL, "unsupported member count");
}

int sw_decide_fame(sw_ctx* c, int32_t* new_rounds, int cap, int* n_new) {
    if (!c) return SW_EINVAL;
    if (n_new) *n_new = 0;
    if (c->R <= 0) return fail(c, SW_EINVAL, "decide_fame before any witness exists (max() of an empty dict in the reference, swirld.py:225)");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->exact) return exact_fame(c, new_rounds, cap, n_new);
    switch (c->nw) {
        case 1: return do_fame<1>(c, new_rounds, cap, n_new);
        case 2: return do_fame<2>(c, new_rounds, cap, n_new);
        case 4: return do_fame<4>(c, new_rounds, cap, n_new);
        case 8: return do_fame<8>(c, new_rounds, cap, n_new);
        case 16: return do_fame<16>(c, new_rounds, cap, n_new);
    }
    return fail(c, SW_EINVAL, "unsupported member count");
}

int sw_decide_fame_partial(sw_ctx* c, int part, int nparts, int8_t* famous, uint8_t* decided, int r_cap, int* r_out) {
    if (!c || !famous || !decided) return SW_EINVAL;
    if (c->poisoned) return fail(c, SW_EIO, "context unusable after an earlier device failure");
    if (nparts < 1 || part < 0 || part >= nparts) return fail(c, SW_EINVAL, "part %d of %d", part, nparts);
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_decide_fame_partial is not available on the exact (forked-hashgraph) path");
    if (c->R <= 0) return fail(c, SW_EINVAL, "decide_fame before any witness exists (max() of an empty dict in the reference, swirld.py:225)");
    if (r_out) *r_out = c->R;
    if (r_cap < c->R) return fail(c, SW_ERANGE, "famous / decided hold %d rounds, %d needed", r_cap, c->R);
    HIPCHK(c, hipSetDevice(c->device));
    switch (c->nw) {
        case 1: return do_fame_partial<1>(c, part, nparts, famous, decided);
        case 2: return do_fame_partial<2>(c, part, nparts, famous, decided);
        case 4: return do_fame_partial<4>(c, part, nparts, famous, decided);
        case 8: return do_fame_partial<8>(c, part, nparts, famous, decided);
        case 16: return do_fame_partial<16>(c, part, nparts, famous, decided);
    }
    return fail(c, SW_EINVAL, "unsupported member count");
}

int sw_commit_fame(sw_ctx* c, const int8_t* famous, const uint8_t* decided, int R, int32_t* new_rounds, int cap, int* n_new) {
    if (!c || !famous || !decided) return SW_EINVAL;
    if (n_new) *n_new = 0;
    if (c->poisoned) return fail(c, SW_EIO, "context unusable after an earlier device failure");
    if (R != c->R) return fail(c, SW_EINVAL, "merged table has %d rounds, the context %d", R, c->R);
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_commit_fame is not available on the exact (forked-hashgraph) path");
    HIPCHK(c, hipSetDevice(c->device));
    const int np = c->npad, n = c->n;
    const int max_c = first_undecided_round(c);
    std::vector<signed char> fam((size_t)(R - max_c) * np, (signed char)-1);
    std::vector<unsigned char> cons(R - max_c, 0);
    const int call_idx = (int)c->fame_calls.size();
    c->fame_calls.push_back({max_c, R, c->divided});
    c->votes_partial = true;  // the deciding call / voter of witnesses decided by other parts is unknown here: sw_get_vote refuses
    int cnt = 0;
    for (int r = max_c; r < R; ++r) {
        for (int m = 0; m < n; ++m) fam[(size_t)(r - max_c) * np + m] = famous[(size_t)r * n + m];
        if (decided[r]) {
            c->cons_h[r] = 1;
            c->cons_call[r] = call_idx;
            if (cnt < cap && new_rounds) new_rounds[cnt] = r;
            ++cnt;
        }
        cons[r - max_c] = c->cons_h[r];
    }
    if (R > max_c) {
        HIPCHK(c, hipMemcpyAsync(c->d_fam.p + (size_t)max_c * np, fam.data(), fam.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_cons.p + max_c, cons.data(), cons.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (n_new) *n_new = cnt;
    if (cnt > cap) return fail(c, SW_ERANGE, "new_rounds capacity %d < %d", cap, cnt);
    return SW_OK;
}

// ---- multi-GPU split of the can_see table by event ranges -----------------------
static int range_checks(sw_ctx* c, int64_t first, int64_t K, const char* what) {
    if (!c) return SW_EINVAL;
    if (c->poisoned) return fail(c, SW_EIO, "context unusable after an earlier device failure");
    if (c->exact) return fail(c, SW_ENOTSUP, "%s is not available on the exact (forked-hashgraph) path", what);
    if (c->vm.active) return fail(c, SW_ENOTSUP, "%s is not available with the windowed table", what);
    if (c->npad > 256 || c->cansee_impl != 6) return fail(c, SW_ENOTSUP, "%s needs the chunk-parallel dataflow sweep (at most 256 members)", what);
    if (first < 0 || K <= 0 || first + K > c->N) return fail(c, SW_ERANGE, "%s: events [%lld, %lld) outside the stored hashgraph", what, (long long)first, (long long)(first + K));
    if (first < c->divided) return fail(c, SW_EINVAL, "%s: events below %lld are divided already", what, (long long)c->divided);
    return SW_OK;
}

static void mark_present(sw_ctx* c, int64_t a, int64_t b) {
    c->present.push_back({a, b});
    std::sort(c->present.begin(), c->present.end());
    std::vector<std::pair<int64_t, int64_t>> m;
    for (const auto& pr : c->present) {
        if (!m.empty() && pr.first <= m.back().second) m.back().second = std::max(m.back().second, pr.second);
        else m.push_back(pr);
    }
    c->present.swap(m);
}

int sw_row_stride(const sw_ctx* c) { return c ? c->npad : 0; }
int sw_get_tally_impl(const sw_ctx* c) { return c ? (c->unit_stake ? c->tally_impl : 0) : SW_EINVAL; }

}  // extern "C"
template <int NW>
static int do_cansee_range(sw_ctx* c, int64_t first, int64_t K) {
    const int np = c->npad;
    if ((int)c->ranges.size() >= SW_RANGE_SLOTS) return fail(c, SW_ERANGE, "more than %d ranges between two rewinds", SW_RANGE_SLOTS);
    for (const auto& pr : c->present)
        if (first < pr.second && first + K > pr.first) return fail(c, SW_EINVAL, "sw_cansee_range: rows [%lld, %lld) are present already", (long long)pr.first, (long long)pr.second);
    if (!c->d_rprov) {
        HIPCHK(c, hipMalloc((void**)&c->d_rprov, (size_t)SW_RANGE_SLOTS * (SW_MAX_CHUNKS + 1) * sizeof(unsigned)));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming));
    }
    if (c->d_L.cap < table_elems(c, c->cap)) return fail(c, SW_EIO, "the can_see table has no halo scratch rows (internal)");
    sw_ctx::RangeRec r;
    r.first = first; r.K = K;
    r.slot = (int)c->ranges.size();
    r.row0 = r.slot * (2 * SW_MAX_CHUNKS + 2);
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(c->chunks, K / std::max<int64_t>(c->chunk_min, 1)));
    r.pl.G = G;
    std::vector<long long>& cc = c->rcuts_stage;
    cc.assign((size_t)2 * SW_MAX_CHUNKS + 2, first + K);
    for (int k = 0; k <= G; ++k) r.pl.a[k] = first + K * k / G;
    for (int k = 0; k < G; ++k) {
        r.pl.w[k] = std::max<int64_t>(0, r.pl.a[k] - c->halo);   // chunk 0 has a halo too: nothing below the range is here yet
        cc[2 * k] = r.pl.w[k];
        cc[2 * k + 1] = r.pl.a[k];
    }
    cc[2 * G] = cc[2 * G + 1] = r.pl.a[G];
    hipStream_t cs = c->stream_cs;
    // behind whatever the main stream still has in flight (appends return without a host synchronisation)
    HIPCHK(c, hipEventRecord(c->ev_main_mark, c->stream));
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_main_mark, 0));
    CHK(dgrow(c, c->d_rcuts, (size_t)SW_RANGE_SLOTS * (2 * SW_MAX_CHUNKS + 2), 0));
    CHK(dgrow(c, c->d_rbnd, (size_t)SW_RANGE_SLOTS * (2 * SW_MAX_CHUNKS + 2) * np, 0));
    HIPCHK(c, hipMemcpyAsync(c->d_rcuts.p + r.row0, cc.data(), cc.size() * sizeof(long long), hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipStreamSynchronize(cs));   // (one staging vector per context: reusable on return)
    hipLaunchKernelGGL(k_chain_bounds, dim3((unsigned)cc.size()), dim3(np), 0, cs, (const int*)c->d_chain_start.p,
                       (const int*)c->d_chain_cnt.p, (const int*)c->d_chain_ev.p, (const long long*)(c->d_rcuts.p + r.row0), np,
                       c->d_rbnd.p + (size_t)r.row0 * np);
    unsigned* prov = c->d_rprov + (size_t)r.slot * (SW_MAX_CHUNKS + 1);
    HIPCHK(c, hipMemsetAsync(prov, 0, (SW_MAX_CHUNKS + 1) * sizeof(unsigned), cs));
    c->ctr.kernel_launches++;
    CHK(launch_cansee_plan<NW>(c, r.pl, (const int*)c->d_rbnd.p + (size_t)r.row0 * np, prov, prov + SW_MAX_CHUNKS, 0, true, 0, 0));
    c->ranges.push_back(r);
    mark_present(c, first, first + K);
    return SW_OK;
}

extern "C" {
int sw_cansee_range(sw_ctx* c, int64_t first, int64_t K) {
    CHK(range_checks(c, first, K, "sw_cansee_range"));
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_events(c, c->N));
    switch (c->nw) {
        case 1: return do_cansee_range<1>(c, first, K);
        case 2: return do_cansee_range<2>(c, first, K);
        case 4: return do_cansee_range<4>(c, first, K);
    }
    return fail(c, SW_ENOTSUP, "sw_cansee_range: more than 256 members");
}

int sw_cansee_repair(sw_ctx* c, int64_t first, int64_t K) {
    CHK(range_checks(c, first, K, "sw_cansee_repair"));
    HIPCHK(c, hipSetDevice(c->device));
    for (auto& r : c->ranges) {
        if (r.first != first || r.K != K) continue;
        if (r.repaired) return SW_OK;
        // every row below the range must be here: the repair reads the rows the provisional entries name
        int64_t covered = c->divided;
        for (const auto& pr : c->present) if (pr.first <= covered) covered = std::max(covered, pr.second);
        if (covered < first) return fail(c, SW_EINVAL, "sw_cansee_repair: the rows [%lld, %lld) below the range are not present yet", (long long)covered, (long long)first);
        const int np = c->npad;
        unsigned* prov = c->d_rprov + (size_t)r.slot * (SW_MAX_CHUNKS + 1);
        const int k_from = r.pl.w[0] > 0 ? 0 : 1;   // (a range that starts at event 0 has no halo in front of its first chunk)
        int rc = SW_EINVAL;
        switch (c->nw) {
            case 1: rc = launch_cansee_plan<1>(c, r.pl, (const int*)c->d_rbnd.p + (size_t)r.row0 * np, prov, prov + SW_MAX_CHUNKS, 0, false, k_from, r.pl.G); break;
            case 2: rc = launch_cansee_plan<2>(c, r.pl, (const int*)c->d_rbnd.p + (size_t)r.row0 * np, prov, prov + SW_MAX_CHUNKS, 0, false, k_from, r.pl.G); break;
            case 4: rc = launch_cansee_plan<4>(c, r.pl, (const int*)c->d_rbnd.p + (size_t)r.row0 * np, prov, prov + SW_MAX_CHUNKS, 0, false, k_from, r.pl.G); break;
        }
        CHK(rc);
        r.repaired = true;
        return SW_OK;
    }
    return fail(c, SW_EINVAL, "sw_cansee_repair: [%lld, %lld) is not a range swept by sw_cansee_range", (long long)first, (long long)(first + K));
}

int sw_export_rows(sw_ctx* c, int64_t first, int64_t K, void* dst_device, void* user_stream) {
    if (!c || !dst_device) return SW_EINVAL;
    if (c->exact || c->vm.active) return fail(c, SW_ENOTSUP, "sw_export_rows: fast path with the plain table only");
    if (first < 0 || K <= 0 || first + K > c->N) return fail(c, SW_ERANGE, "sw_export_rows: events [%lld, %lld) outside the stored hashgraph", (long long)first, (long long)(first + K));
    bool ok = first + K <= c->divided;
    for (const auto& pr : c->present) ok = ok || (first >= pr.first && first + K <= pr.second);
    if (!ok) return fail(c, SW_EINVAL, "sw_export_rows: the rows [%lld, %lld) have not been computed here", (long long)first, (long long)(first + K));
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ev_user) HIPCHK(c, hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming));
    hipStream_t cs = c->stream_cs, us = (hipStream_t)user_stream;
    // the destination may still be read by work the caller enqueued earlier (a previous collective)
    HIPCHK(c, hipEventRecord(c->ev_user, us));
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_user, 0));
    HIPCHK(c, hipMemcpyAsync(dst_device, c->d_L.p + (size_t)first * c->npad, (size_t)K * c->npad * sizeof(int32_t), hipMemcpyDeviceToDevice, cs));
    HIPCHK(c, hipEventRecord(c->ev_user, cs));
    HIPCHK(c, hipStreamWaitEvent(us, c->ev_user, 0));
    return SW_OK;
}

int sw_import_rows(sw_ctx* c, int64_t first, int64_t K, const void* src_device, void* user_stream) {
    if (!src_device) return SW_EINVAL;
    CHK(range_checks(c, first, K, "sw_import_rows"));
    for (const auto& pr : c->present)
        if (first < pr.second && first + K > pr.first) return fail(c, SW_EINVAL, "sw_import_rows: rows [%lld, %lld) are present already", (long long)pr.first, (long long)pr.second);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_events(c, c->N));
    if (!c->ev_user) HIPCHK(c, hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming));
    hipStream_t cs = c->stream_cs, us = (hipStream_t)user_stream;
    HIPCHK(c, hipEventRecord(c->ev_user, us));            // what the caller enqueued so far (the collective that fills src)
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_user, 0));
    HIPCHK(c, hipMemcpyAsync(c->d_L.p + (size_t)first * c->npad, src_device, (size_t)K * c->npad * sizeof(int32_t), hipMemcpyDeviceToDevice, cs));
    HIPCHK(c, hipEventRecord(c->ev_user, cs));            // ... and the caller's stream may not reuse src before the copy is done
    HIPCHK(c, hipStreamWaitEvent(us, c->ev_user, 0));
    mark_present(c, first, first + K);
    return SW_OK;
}

int sw_get_range_stats(sw_ctx* c, int64_t* provisional, int64_t* repaired, int64_t* resweeps) {
    if (!c) return SW_EINVAL;
    int64_t pv = 0, rp = 0, rs = 0;
    if (c->d_rprov && !c->ranges.empty()) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream_cs));
        std::vector<unsigned> h((size_t)SW_RANGE_SLOTS * (SW_MAX_CHUNKS + 1));
        HIPCHK(c, hipMemcpy(h.data(), c->d_rprov, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (const auto& r : c->ranges) {
            const unsigned* p = h.data() + (size_t)r.slot * (SW_MAX_CHUNKS + 1);
            const int Cc = c->chunk_cfg == 1 ? 2 : 4;
            for (int k = 0; k < r.pl.G; ++k) {
                pv += p[k];
                const int64_t len = r.pl.a[k + 1] - r.pl.a[k];
                if ((k > 0 || r.pl.w[0] > 0) && p[k] > (unsigned)std::min<int64_t>((len * c->n) / (32 * Cc), 0x7fffffff)) ++rs;
            }
            rp += p[SW_MAX_CHUNKS];
        }
    }
    if (provisional) *provisional = pv;
    if (repaired) *repaired = rp;
    if (resweeps) *resweeps = rs;
    return SW_OK;
}

int sw_set_window(sw_ctx* c, int enable, int chunk_mb) {
    if (!c) return SW_EINVAL;
    if (c->N != 0) return fail(c, SW_EINVAL, "sw_set_window must be called before the first event is appended");
    HIPCHK(c, hipSetDevice(c->device));
    if (!enable) {
        if (c->vm.active) {
            HIPCHK(c, hipDeviceSynchronize());
            vm_destroy(c);
            c->d_L.p = nullptr; c->d_L.cap = 0;
            if (c->cap) CHK(dgrow(c, c->d_L, table_elems(c, c->cap), 0));   // (with the halo scratch rows: ADVICE r3)
        }
        return SW_OK;
    }
    if (c->vm.active) return SW_OK;
    HIPCHK(c, hipDeviceSynchronize());
    dfree(c->d_L);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    HIPCHK(c, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = 4096;
    if (chunk_mb <= 0) chunk_mb = 64;
    if (const char* e_ = getenv("SW_VM_CHUNK_MB")) chunk_mb = std::max(1, atoi(e_));
    VmTable& v = c->vm;
    v.chunk = (((size_t)chunk_mb << 20) + gran - 1) / gran * gran;
    const size_t rowbytes = (size_t)c->npad * sizeof(int32_t);
    size_t va = std::min<size_t>((size_t)0x7ffffff0ull * rowbytes, (size_t)4 << 40);  // up to 4 TB of address space
    va = (va + v.chunk - 1) / v.chunk * v.chunk;
    void* base = nullptr;
    hipError_t e = hipMemAddressReserve(&base, va, 0, nullptr, 0);
    if (e != hipSuccess) { v = VmTable{}; return fail(c, SW_ENOMEM, "hipMemAddressReserve(%zu GB) failed: %s", va >> 30, hipGetErrorString(e)); }
    v.base = (char*)base;
    v.va_bytes = va;
    v.handle.assign(va / v.chunk, hipMemGenericAllocationHandle_t{});
    v.mapped.assign(va / v.chunk, 0);
    v.active = true;
    c->d_L.p = (int32_t*)base;
    c->d_L.cap = va / sizeof(int32_t);
    c->first_resident = 0;
    return SW_OK;
}

int sw_set_window_lapse(sw_ctx* c, int64_t events) {
    if (!c) return SW_EINVAL;
    if (events < 0) return fail(c, SW_EINVAL, "sw_set_window_lapse: a number of events >= 0 (0 = members never lapse)");
    if (!c->vm.active && events > 0) return fail(c, SW_EINVAL, "sw_set_window_lapse needs the windowed table (sw_set_window first)");
    c->window_lapse = events;
    if (events == 0) c->lapsed.clear();
    return SW_OK;
}

int sw_get_window(sw_ctx* c, int64_t* first_resident_event, int64_t* resident_bytes, int64_t* evictions) {
    if (!c) return SW_EINVAL;
    if (first_resident_event) *first_resident_event = c->first_resident;
    if (resident_bytes) {
        if (c->vm.active) { size_t cnt = 0; for (char m_ : c->vm.mapped) cnt += m_ != 0; *resident_bytes = (int64_t)(cnt * c->vm.chunk); }
        else *resident_bytes = (int64_t)(c->d_L.cap * sizeof(int32_t));
    }
    if (evictions) *evictions = c->vm.evictions;
    return SW_OK;
}

int sw_set_forks(sw_ctx* c, int accept) {
    if (!c) return SW_EINVAL;
    if (accept != 0 && accept != 1) return fail(c, SW_EINVAL, "sw_set_forks: 0 (refuse) or 1 (accept)");
    c->forks_mode = accept;
    return SW_OK;
}

int sw_get_exact(sw_ctx* c, int* out) {
    if (!c || !out) return SW_EINVAL;
    *out = c->exact ? 1 : 0;
    return SW_OK;
}

// Members of witnesses[r] in dict insertion order (what iterating self.witnesses[r] yields, swirld.py:234, 240):
// ascending event index of a member's FIRST witness of the round — which is the table entry unless a fork
// sibling replaced it (exact path: kept explicitly).
int sw_get_witness_order(sw_ctx* c, int r, int32_t* members, int* n_out) {
    if (!c || !members || !n_out) return SW_EINVAL;
    *n_out = 0;
    if (r < 0 || r >= c->R) return fail(c, SW_ERANGE, "round %d outside [0, %d)", r, c->R);
    HIPCHK(c, hipSetDevice(c->device));
    const int np = c->npad, n = c->n;
    if (c->exact) {
        int32_t cnt = 0;
        std::vector<int32_t> row(np);
        HIPCHK(c, hipMemcpyAsync(&cnt, c->x_wcnt.p + r, sizeof cnt, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(row.data(), c->x_worder.p + (size_t)r * np, np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < cnt; ++i) members[i] = row[i];
        *n_out = cnt;
        return SW_OK;
    }
    std::vector<int32_t> row(np);
    HIPCHK(c, hipMemcpyAsync(row.data(), c->d_wit.p + (size_t)r * np, np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<std::pair<int32_t, int32_t>> ws;
    for (int m = 0; m < n; ++m) if (row[m] >= 0) ws.push_back({row[m], m});
    std::sort(ws.begin(), ws.end());
    for (size_t i = 0; i < ws.size(); ++i) members[i] = ws[i].second;
    *n_out = (int)ws.size();
    return SW_OK;
}

int sw_rewind(sw_ctx* c) {
    if (!c) return SW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream_cs));   // (calls return without a host synchronisation)
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    const size_t rows = (size_t)c->Rcap * c->npad;
    c->fame_calls.clear();
    c->votes_partial = false;
    c->eval_src = 0;
    std::fill(c->cons_call.begin(), c->cons_call.end(), -1);
    {   // every table back to its initial value, one launch (byte tables as 32-bit words: their allocations are padded to 256 B)
        RewindJob J{};
        int nj = 0;
        auto job = [&](void* p_, size_t words, int v) { if (p_ && words) { J.p[nj] = (int*)p_; J.n[nj] = words; J.v[nj] = v; ++nj; } };
        job(c->d_lo.p, rows, SW_INF);
        job(c->d_lopos.p, rows, 0);
        job(c->d_wit.p, rows, -1);
        job(c->d_dec_call.p, rows, -1);
        job(c->d_dec_by.p, rows, -1);
        job(c->d_evalround.p, 2 * (size_t)c->npad, -1);
        job(c->d_evalpos.p, 2 * (size_t)c->npad, 0);
        job(c->d_front.p, c->npad, -1);
        job(c->d_round.p, (size_t)c->N, -1);
        if (rows % 4 == 0) job(c->d_fam.p, rows / 4, -1);
        else HIPCHK(c, hipMemsetAsync(c->d_fam.p, 0xff, rows, c->stream));
        if (c->Rcap % 4 == 0) job(c->d_cons.p, (size_t)c->Rcap / 4, 0);
        else HIPCHK(c, hipMemsetAsync(c->d_cons.p, 0, c->Rcap, c->stream));
        size_t longest = 1;
        for (int i = 0; i < nj; ++i) longest = std::max<size_t>(longest, J.n[i]);
        if (nj) {
            hipLaunchKernelGGL(k_rewind_fill, dim3((unsigned)std::min<size_t>((longest + 1023) / 1024, 1024), nj), dim3(256), 0, c->stream, J);
            c->ctr.kernel_launches++;
            HIPCHK(c, hipGetLastError());
        }
    }
    std::fill(c->front.begin(), c->front.end(), -1);
    std::fill(c->divided_cnt.begin(), c->divided_cnt.end(), 0);
    std::fill(c->lapsed.begin(), c->lapsed.end(), 0);   // (every row is mapped again below: nobody has lapsed)
    if (c->vm.active && c->vm.lo > 0) {
        // every row is recomputed from event 0: the whole table moves to a fresh reservation (made before
        // the old one is given back, so the addresses differ), physical chunks recycled through the pool
        HIPCHK(c, hipDeviceSynchronize());
        VmTable& v = c->vm;
        void* base = nullptr;
        hipError_t e = hipMemAddressReserve(&base, v.va_bytes, 0, nullptr, 0);
        if (e != hipSuccess) { c->poisoned = true; return fail(c, SW_ENOMEM, "hipMemAddressReserve failed on rewind: %s", hipGetErrorString(e)); }
        for (size_t s_ = 0; s_ < v.mapped.size(); ++s_)
            if (v.mapped[s_]) { HIPCHK(c, hipMemUnmap(v.base + s_ * v.chunk, v.chunk)); v.pool.push_back(v.handle[s_]); v.mapped[s_] = 0; }
        HIPCHK(c, hipMemAddressFree(v.base, v.va_bytes));
        CHK(vm_flush_translations(c));
        v.base = (char*)base;
        v.lo = 0; v.hi = 0;
        c->vm_scratch_ok = false;
        c->d_L.p = (int32_t*)base;
        c->first_resident = 0;
        CHK(vm_ensure(c, (size_t)c->N * c->npad * sizeof(int32_t)));
    }
    c->ranges.clear();
    c->present.clear();
    std::fill(c->divided_head.begin(), c->divided_head.end(), -1);
    std::fill(c->lo0_h.begin(), c->lo0_h.end(), SW_INF);
    std::fill(c->cons_h.begin(), c->cons_h.end(), 0);
    c->divided = 0;
    c->R = 0;
    c->sw_dirty_from = 1;
    c->dbg_iter_base = c->ctr.round_iterations;   // (the phase stamps of the loop kernels count from the rewind)
    c->transactions.clear();
    std::fill(c->ord_pos.begin(), c->ord_pos.end(), 0);
    if (c->exact) CHK(exact_rewind(c));
    return SW_OK;
}

int sw_reset(sw_ctx* c) {
    if (!c) return SW_EINVAL;
    CHK(sw_rewind(c));
    // forget the events as well; device storage (and the launch graphs keyed on it) stays
    c->cr.clear(); c->sp.clear(); c->op.clear(); c->ht.clear(); c->seq_h.clear();
    std::fill(c->head.begin(), c->head.end(), -1);
    std::fill(c->first_ev.begin(), c->first_ev.end(), -1);
    std::fill(c->nev.begin(), c->nev.end(), 0);
    c->blk_hmin.clear(); c->blk_hmax.clear();
    c->max_height = 0;
    c->N = 0;
    c->sig_h.clear();
    c->chain_cap.clear();
    c->chain_ev_h.clear();
    c->pool_used = 0;
    c->pool_h_valid = true;
    if (c->stream_io) HIPCHK(c, hipStreamSynchronize(c->stream_io));  // a payload upload may still be writing t / sig
    c->payload_pending = false;
    c->exact = false;  // (a fresh hashgraph starts on the fast path again)
    c->chunks_off = false;  // ... and with the chunked sweep
    return SW_OK;
}

static void split_dissolve(SplitGroup* g) {
    if (!g) return;
    for (int q = 0; q < g->parts; ++q)
        if (g->ctx[q]) {
            sw_ctx* c = g->ctx[q];
            (void)hipStreamSynchronize(c->stream);
            if (c->split == g) { c->tally_impl = c->split_saved[0]; c->tally_auto = c->split_saved[1] != 0; c->K = c->split_saved[2]; c->K_auto = c->split_saved[3] != 0; }   // (what the link pinned)
            c->split = nullptr;
        }
    for (int w = 0; w < 2; ++w)
        for (int q = 0; q < SW_MAX_PARTS; ++q)
            for (auto e : g->ev[w][q]) if (e) (void)hipEventDestroy(e);
    delete g;
}

int sw_split_link(sw_ctx* const* ctxs, int parts) {
    if (!ctxs || parts < 2 || parts > SW_MAX_PARTS) return fail(nullptr, SW_EINVAL, "sw_split_link: 2 .. %d contexts", SW_MAX_PARTS);
    sw_ctx* c0 = ctxs[0];
    for (int q = 0; q < parts; ++q) {
        sw_ctx* c = ctxs[q];
        if (!c) return fail(c0, SW_EINVAL, "sw_split_link: context %d is NULL", q);
        for (int o = 0; o < q; ++o) if (ctxs[o] == c) return fail(c, SW_EINVAL, "sw_split_link: context %d given twice", q);
        if (c->poisoned) return fail(c, SW_EIO, "context poisoned by an earlier failure");
        if (c->split) return fail(c, SW_EINVAL, "sw_split_link: context %d is linked already", q);
        if (c->exact || c->vm.active || !c->unit_stake)
            return fail(c, SW_ENOTSUP, "sw_split_link: unit stake, the fast path and the plain can_see table only");
        if (c->n != c0->n || c->N != c0->N || c->divided != c0->divided || c->R != c0->R || c->coin_period != c0->coin_period)
            return fail(c, SW_EINVAL, "sw_split_link: context %d does not hold the same hashgraph, divided to the same point, as context 0", q);
        if (c->MCAP != c0->MCAP || c->NEARCAP != c0->NEARCAP || c->skip != c0->skip || c->gallop_after != c0->gallop_after || c->pipe != c0->pipe)
            return fail(c, SW_EINVAL, "sw_split_link: context %d was created under other tuning knobs than context 0", q);
    }
    SplitGroup* g = new SplitGroup();
    g->parts = parts;
    for (int w = 0; w < 2; ++w) for (int q = 0; q < SW_MAX_PARTS; ++q) g->rec[w][q].store(0);
    for (int q = 0; q < parts; ++q) {
        sw_ctx* c = ctxs[q];
        g->ctx[q] = c;
        if (hipSetDevice(c->device) != hipSuccess) { split_dissolve(g); return fail(c0, SW_EIO, "hipSetDevice(%d) failed", c->device); }
        for (int o = 0; o < parts; ++o)
            if (ctxs[o]->device != c->device) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, c->device, ctxs[o]->device);
                if (!can) { split_dissolve(g); return fail(c0, SW_ENOTSUP, "sw_split_link: device %d cannot map the memory of device %d", c->device, ctxs[o]->device); }
                const hipError_t e = hipDeviceEnablePeerAccess(ctxs[o]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); split_dissolve(g); return fail(c0, SW_EIO, "hipDeviceEnablePeerAccess failed: %s", hipGetErrorString(e)); }
                (void)hipGetLastError();
            }
        for (int w = 0; w < 2; ++w)
            for (auto& e : g->ev[w][q])
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { split_dissolve(g); return fail(c0, SW_EIO, "hipEventCreate failed"); }
    }
    for (int q = 0; q < parts; ++q) {
        sw_ctx* c = ctxs[q];
        // the split kernels exist for the one-wave-per-slot tally (the default beyond 256 members)
        c->split_saved[0] = c->tally_impl; c->split_saved[1] = c->tally_auto; c->split_saved[2] = c->K; c->split_saved[3] = c->K_auto;
        c->tally_impl = 1; c->tally_auto = false; c->K = c->K_flat; c->K_auto = false;
        c->split_part = q;
        c->split_iter = 0;
        c->split_failed = false;
        c->split = g;
    }
    return SW_OK;
}

int sw_split_unlink(sw_ctx* c) {
    if (!c) return SW_EINVAL;
    if (!c->split) return SW_OK;
    split_dissolve(c->split);
    return SW_OK;
}

int sw_find_order(sw_ctx* c, const int32_t* rounds, int n_rounds, int32_t* out_events, int64_t cap, int64_t* n_out) {
    if (!c) return SW_EINVAL;
    if (n_out) *n_out = 0;
    if (n_rounds < 0 || (n_rounds > 0 && !rounds)) return fail(c, SW_EINVAL, "find_order: NULL rounds");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int32_t> rs(rounds, rounds + n_rounds);
    if (c->exact) return exact_order(c, rs, out_events, cap, n_out);
    switch (c->nw) {
        case 1: return do_find_order<1>(c, rs, out_events, cap, n_out);
        case 2: return do_find_order<2>(c, rs, out_events, cap, n_out);
        case 4: return do_find_order<4>(c, rs, out_events, cap, n_out);
        case 8: return do_find_order<8>(c, rs, out_events, cap, n_out);
        case 16: return do_find_order<16>(c, rs, out_events, cap, n_out);
    }
    return fail(c, SW_EINVAL, "unsupported member count");
}

// ---- getters ----
static int get_i32(sw_ctx* c, const int32_t* src, int64_t first, int64_t K, int32_t* out, int64_t limit) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > limit) return fail(c, SW_ERANGE, "range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + K), (long long)limit);
    if (!K) return SW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, src + first, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

int sw_get_height(sw_ctx* c, int64_t first, int64_t K, int32_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > c->N) return fail(c, SW_ERANGE, "range outside the stored hashgraph");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_dag_h(c));
    std::copy(c->ht.begin() + first, c->ht.begin() + first + K, out);
    return SW_OK;
}

int sw_get_round(sw_ctx* c, int64_t first, int64_t K, int32_t* out) {
    return get_i32(c, c ? c->d_round.p : nullptr, first, K, out, c ? c->divided : 0);
}

int sw_get_can_see(sw_ctx* c, int64_t first, int64_t K, int32_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > c->divided) return fail(c, SW_ERANGE, "range outside the divided events");
    if (!K) return SW_OK;
    if (first < c->first_resident) return fail(c, SW_ERANGE, "can_see rows below event %lld were evicted (windowed mode)", (long long)c->first_resident);
    HIPCHK(c, hipSetDevice(c->device));
    const int np = c->npad, n = c->n;
    const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / (np * 4));
    std::vector<int32_t> tmp((size_t)std::min(chunk, K) * np);
    for (int64_t a = 0; a < K; a += chunk) {
        const int64_t m = std::min(chunk, K - a);
        HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_L.p + (size_t)(first + a) * np, (size_t)m * np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t i = 0; i < m; ++i) memcpy(out + (size_t)(a + i) * n, tmp.data() + (size_t)i * np, n * sizeof(int32_t));
    }
    return SW_OK;
}

int sw_max_round(sw_ctx* c, int* out) {
    if (!c || !out) return SW_EINVAL;
    *out = c->R - 1;
    return SW_OK;
}

int sw_get_witnesses(sw_ctx* c, int r0, int r1, int32_t* out) {
    return get_round_rows<int32_t>(c, c ? c->d_wit.p : nullptr, r0, r1, out, -1);
}

int sw_get_famous(sw_ctx* c, int r0, int r1, int8_t* out) {
    return get_round_rows<signed char>(c, c ? c->d_fam.p : nullptr, r0, r1, (signed char*)out, (signed char)-1);
}

// Node.famous keyed by EVENT (swirld.py:64): -1 undecided / not a witness, 0 / 1.  On the fast path a
// member has one witness per round for good, so this is the slot table scattered to the events; on the
// exact path a fork sibling may have replaced a decided witness, whose entry the reference keeps.
int sw_get_famous_events(sw_ctx* c, int64_t first, int64_t K, int8_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > c->N) return fail(c, SW_ERANGE, "range outside the stored events");
    if (!K) return SW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->exact) {
        HIPCHK(c, hipMemcpyAsync(out, c->x_fam_ev.p + first, (size_t)K, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return SW_OK;
    }
    memset(out, 0xff, (size_t)K);
    const int np = c->npad, R = c->R;
    if (R <= 0) return SW_OK;
    std::vector<int32_t> wit((size_t)R * np);
    std::vector<signed char> fam((size_t)R * np);
    HIPCHK(c, hipMemcpyAsync(wit.data(), c->d_wit.p, wit.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fam.data(), c->d_fam.p, fam.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < wit.size(); ++i)
        if (wit[i] >= first && wit[i] < first + K) out[wit[i] - first] = fam[i];
    return SW_OK;
}

int sw_get_consensus(sw_ctx* c, int r0, int r1, uint8_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (r0 < 0 || r1 < r0) return fail(c, SW_ERANGE, "bad round range");
    for (int r = r0; r < r1; ++r) out[r - r0] = (r < c->R && r < (int)c->cons_h.size()) ? c->cons_h[r] : 0;
    return SW_OK;
}

int sw_get_sees_mask(sw_ctx* c, int64_t first, int64_t K, uint64_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > c->divided) return fail(c, SW_ERANGE, "range outside the divided events");
    if (!K) return SW_OK;
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_get_sees_mask is not available on the exact (forked-hashgraph) path");
    HIPCHK(c, hipSetDevice(c->device));
    const int nw = c->nw, nwo = (c->n + 63) / 64;
    std::vector<u64> tmp((size_t)K * nw);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_S.p + (size_t)first * nw, tmp.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < K; ++i)
        for (int j = 0; j < nwo; ++j) out[(size_t)i * nwo + j] = tmp[(size_t)i * nw + j];
    return SW_OK;
}

// Node.votes[voter][candidate] (swirld.py:60-61, 256-272).  The elections keep votes as per-round
// member bitmasks, so an entry is recomputed on demand — with the reference's FULL semantics, call
// schedule included:
//  * the VALUE of votes[y][x] is a function of the hashgraph alone (y's strongly-seen set, the
//    votes of those witnesses, y's coin bit): the election of x is replayed level by level up to
//    y's round, every witness voting;
//  * whether the ENTRY exists depends on the decide_fame() calls (Appendix A Q8/Q9): y recorded a
//    vote on x iff in some call both were registered witnesses, y was a voter (round(y) > max_c of
//    that call), round(x) was not yet in `consensus`, and x was still undecided when y's turn came
//    — i.e. the call precedes the one that decided x, or is that call and y comes before the
//    deciding voter in voter order (rounds ascending, registration order inside a round); the
//    deciding voter itself stores nothing.  Per call the context keeps (max_c, rounds, events
//    divided), per round the call that put it into `consensus`, per witness the call and the voter
//    that decided it (written by the election kernels).
int sw_get_vote(sw_ctx* c, int rv, int mv, int rc, int mc, int8_t* out) {
    if (!c || !out) return SW_EINVAL;
    *out = -1;
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_get_vote is not available on the exact (forked-hashgraph) path");
    if (c->votes_partial) return fail(c, SW_ENOTSUP, "sw_get_vote is not available after sw_commit_fame (the deciding voters of other parts' witnesses are not recorded)");
    const int np = c->npad, nw = c->nw, n = c->n;
    if (rc < 0 || rv >= c->R || mv < 0 || mv >= n || mc < 0 || mc >= n) return fail(c, SW_ERANGE, "witness slot outside the table");
    if (rv <= rc) return SW_OK;
    if (rv >= c->Sw_rows || c->sw_dirty_from <= rv) return fail(c, SW_EINVAL, "votes are available after decide_fame has seen these rounds");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_sig_h(c));
    const int D = rv - rc;
    std::vector<int32_t> wit((size_t)(D + 1) * np);
    std::vector<u64> Sw((size_t)D * np * nw);
    int32_t dec[2] = {-1, -1};
    HIPCHK(c, hipMemcpyAsync(wit.data(), c->d_wit.p + (size_t)rc * np, wit.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(Sw.data(), c->d_Sw.p + (size_t)(rc + 1) * np * nw, Sw.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&dec[0], c->d_dec_call.p + (size_t)rc * np + mc, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&dec[1], c->d_dec_by.p + (size_t)rc * np + mc, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int32_t x = wit[mc], y = wit[(size_t)D * np + mv];
    if (x < 0 || y < 0) return SW_OK;  // not witnesses: no entry
    // ---- does the entry exist?  calls c0 .. c1 in which y evaluated x
    const int ncalls = (int)c->fame_calls.size();
    const int64_t born = (int64_t)std::max(x, y) + 1;  // both are divided (hence registered) once `divided` >= born
    int c0 = 0;
    while (c0 < ncalls && c->fame_calls[c0].divided < born) ++c0;
    int c1 = ncalls - 1;
    while (c1 >= c0 && c->fame_calls[c1].max_c + 1 > rv) --c1;          // y must be a voter: round(y) >= max_c + 1
    if (c->cons_call[rc] >= 0) c1 = std::min(c1, c->cons_call[rc]);     // rounds in `consensus` are skipped (swirld.py:233)
    if (dec[0] >= 0) {
        int32_t rz = -1;
        HIPCHK(c, hipMemcpy(&rz, c->d_round.p + dec[1], sizeof rz, hipMemcpyDeviceToHost));
        const bool before_decider = y != dec[1] && (rv < rz || (rv == rz && y < dec[1]));
        c1 = std::min(c1, before_decider ? dec[0] : dec[0] - 1);
    }
    if (c1 < c0) return SW_OK;
    // ---- its value: the election of x replayed up to y's level
    std::vector<char> V(n, 0), Vn(n, 0);
    auto sw_bit = [&](int d, int voter, int member) {  // d = 1..D
        return (int)((Sw[((size_t)(d - 1) * np + voter) * nw + (member >> 6)] >> (member & 63)) & 1ull);
    };
    for (int v = 0; v < n; ++v) V[v] = wit[(size_t)1 * np + v] >= 0 ? (char)sw_bit(1, v, mc) : 0;  // x in s (swirld.py:258)
    if (D == 1) { *out = V[mv]; return SW_OK; }
    const uint64_t tot2 = 2ull * c->tot;
    for (int d = 2; d <= D; ++d) {
        const int32_t* wrow = wit.data() + (size_t)d * np;
        const bool coin_round = (d % c->coin_period) == 0;
        for (int v = 0; v < n; ++v) {
            Vn[v] = 0;
            if (wrow[v] < 0) continue;
            uint64_t yes = 0, all = 0;
            for (int m = 0; m < n; ++m)
                if (sw_bit(d, v, m)) { all += c->stake_h[m]; if (V[m]) yes += c->stake_h[m]; }
            const uint64_t no = all - yes;
            const int vote = !(no > yes);            // majority(): tie -> True (swirld.py:24-27)
            const uint64_t tt = vote ? yes : no;
            Vn[v] = (char)vote;
            if (coin_round && !(3 * tt > tot2)) Vn[v] = (char)(c->sig_h[(size_t)wrow[v] * 64] >> 7);  // swirld.py:272
        }
        if (d == D) { *out = Vn[mv]; return SW_OK; }
        V.swap(Vn);
    }
    return SW_OK;
}

int sw_get_known_heights(sw_ctx* c, int64_t head_event, int32_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (head_event < c->first_resident || head_event >= c->divided) return fail(c, SW_ERANGE, "head %lld is not a divided, resident event", (long long)head_event);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_dag_h(c));  // heights on the device
    const int np = c->npad;
    CHK(dgrow(c, c->d_q, (size_t)3 * np, 0));
    hipLaunchKernelGGL(k_known_heights, dim3(1), dim3(np), 0, c->stream, (const int*)c->d_L.p, (const int*)c->d_ht.p, (int)head_event, np, c->d_q.p);
    c->ctr.kernel_launches++;
    std::vector<int32_t> tmp(np);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_q.p, np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::copy(tmp.begin(), tmp.begin() + c->n, out);
    return SW_OK;
}

int sw_sync_diff(sw_ctx* c, int64_t head_event, const int32_t* known_height, int32_t* pos_first, int32_t* pos_end, int64_t* n_events) {
    if (!c || !known_height || !pos_first || !pos_end) return SW_EINVAL;
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_sync_diff is not available on the exact (forked-hashgraph) path");
    if (head_event < c->first_resident || head_event >= c->divided) return fail(c, SW_ERANGE, "head %lld is not a divided, resident event", (long long)head_event);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_dag_h(c));
    const int np = c->npad, n = c->n;
    CHK(dgrow(c, c->d_q, (size_t)3 * np, 0));
    std::vector<int32_t> kn(np, -1), res((size_t)2 * np);
    std::copy(known_height, known_height + n, kn.begin());
    HIPCHK(c, hipMemcpyAsync(c->d_q.p, kn.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_sync_diff, dim3(1), dim3(np), 0, c->stream, (const int*)c->d_L.p, (const int*)c->d_ht.p, (const int*)c->d_seq.p,
                       (const int*)c->d_cr.p, (const int*)c->d_chain_start.p, (const int*)c->d_chain_ev.p, (const int*)c->d_q.p,
                       (int)head_event, np, c->d_q.p + np, c->d_q.p + 2 * np);
    c->ctr.kernel_launches++;
    HIPCHK(c, hipMemcpyAsync(res.data(), c->d_q.p + np, (size_t)2 * np * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t tot = 0;
    for (int m = 0; m < n; ++m) {
        pos_first[m] = res[m];
        pos_end[m] = res[(size_t)np + m];
        tot += pos_end[m] - pos_first[m];
    }
    if (n_events) *n_events = tot;
    return SW_OK;
}

int sw_get_chain_events(sw_ctx* c, int member, int32_t p0, int32_t p1, int32_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (c->exact) return fail(c, SW_ENOTSUP, "sw_get_chain_events is not available on the exact (forked-hashgraph) path");
    if (member < 0 || member >= c->n || p0 < 0 || p1 < p0 || p1 > c->nev[member]) return fail(c, SW_ERANGE, "chain positions [%d, %d) of member %d", p0, p1, member);
    if (p1 == p0) return SW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_chain_ev.p + c->chain_start_h[member] + p0, (size_t)(p1 - p0) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}

int sw_num_ordered(sw_ctx* c, int64_t* out) {
    if (!c || !out) return SW_EINVAL;
    *out = (int64_t)c->transactions.size();
    return SW_OK;
}

int sw_get_transactions(sw_ctx* c, int64_t first, int64_t K, int32_t* out) {
    if (!c || !out) return SW_EINVAL;
    if (first < 0 || K < 0 || first + K > (int64_t)c->transactions.size()) return fail(c, SW_ERANGE, "range outside the ordered events");
    for (int64_t i = 0; i < K; ++i) out[i] = c->transactions[first + i];
    return SW_OK;
}

// (one counter lives on the device: the events the finalize check sent back to their rows)
static int refresh_device_counters(sw_ctx* c) {
    if (!c->d_fin) return SW_OK;
    unsigned long long tot = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    HIPCHK(c, hipMemcpy(&tot, c->d_fin + 2, sizeof tot, hipMemcpyDeviceToHost));
    c->ctr.finalize_from_rows = (int64_t)tot;
    return SW_OK;
}

int sw_get_counters(sw_ctx* c, sw_counters* out) {
    if (!c || !out) return SW_EINVAL;
    CHK(refresh_device_counters(c));
    *out = c->ctr;
    return SW_OK;
}

int sw_get_counters_sized(sw_ctx* c, void* out, size_t out_bytes) {
    if (!c || !out) return SW_EINVAL;
    CHK(refresh_device_counters(c));
    memcpy(out, &c->ctr, std::min(out_bytes, sizeof(sw_counters)));
    return SW_OK;
}

int sw_set_profiling(sw_ctx* c, int enable) {
    if (!c) return SW_EINVAL;
    c->profiling = enable != 0;
    return SW_OK;
}

int sw_debug_clocks(sw_ctx* c, unsigned long long* out, int64_t cap_words) {
    if (!c || !out) return SW_EINVAL;
    if (!c->d_dbg) return fail(c, SW_EINVAL, "sw_debug_clocks: the context was not created with SW_DEBUG_CLOCKS=1");
    const int64_t nw = std::min<int64_t>(cap_words, (int64_t)SW_DBG_MAX_ITERS * 32);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_dbg, (size_t)nw * 8, hipMemcpyDeviceToHost));
    return SW_OK;
}

int sw_debug_block_clocks(sw_ctx* c, unsigned long long* out, int64_t cap_words) {
    if (!c || !out) return SW_EINVAL;
    if (!c->d_dbg_blk) return fail(c, SW_EINVAL, "sw_debug_block_clocks: the context was not created with SW_DEBUG_CLOCKS=3");
    const int64_t nw = std::min<int64_t>(cap_words, (int64_t)SW_DBG_MAX_ITERS * 2 * 2048);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_dbg_blk, (size_t)nw * 8, hipMemcpyDeviceToHost));
    return SW_OK;
}

int sw_get_timings(sw_ctx* c, sw_timings* out) {
    if (!c || !out) return SW_EINVAL;
    *out = c->tm;
    return SW_OK;
}

int sw_synchronize(sw_ctx* c) {
    if (!c) return SW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SW_OK;
}


// ---- ingest-side crypto batches: stateless, one thread per message ------------
}  // extern "C"

namespace {
__global__ void __launch_bounds__(64)
k_verify_batch(const uint8_t* __restrict__ msgs, const long long* __restrict__ off, const uint8_t* __restrict__ sig,
               const uint8_t* __restrict__ pk, int K, uint8_t* ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    ok[i] = swc::ed25519_verify(sig + (size_t)i * 64, msgs + off[i], (uint64_t)(off[i + 1] - off[i]), pk + (size_t)i * 32) ? 1 : 0;
}
__global__ void __launch_bounds__(64)
k_blake2b_batch(const uint8_t* __restrict__ msgs, const long long* __restrict__ off, int K, uint8_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    swc::blake2b_256(msgs + off[i], (uint64_t)(off[i + 1] - off[i]), out + (size_t)i * 32);
}
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
int crypto_batch(int device, int64_t K, const uint8_t* msgs, const int64_t* off, const uint8_t* sig, const uint8_t* pk, uint8_t* out, bool verify) {
    if (K < 0 || (K > 0 && (!msgs || !off || !out || (verify && (!sig || !pk))))) return fail(nullptr, SW_EINVAL, "NULL batch arrays");
    if (K == 0) return SW_OK;
    if (K > 0x7fffffff) return fail(nullptr, SW_ERANGE, "batch too large");
    for (int64_t i = 0; i < K; ++i) if (off[i + 1] < off[i] || off[i] < 0) return fail(nullptr, SW_EINVAL, "message offsets must be non-decreasing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, SW_ENODEV, "no such HIP device (there is no CPU fallback)");
    HIPCHK(nullptr, hipSetDevice(device));
    const size_t nbytes = (size_t)off[K];
    const size_t out_per = verify ? 1 : 32;
    DevTmp d_m, d_off, d_sig, d_pk, d_out;
    HIPCHK(nullptr, d_m.alloc(nbytes));
    HIPCHK(nullptr, d_off.alloc((size_t)(K + 1) * 8));
    HIPCHK(nullptr, d_out.alloc((size_t)K * out_per));
    HIPCHK(nullptr, hipMemcpy(d_m.p, msgs, nbytes, hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMemcpy(d_off.p, off, (size_t)(K + 1) * 8, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((K + 63) / 64);
    if (verify) {
        HIPCHK(nullptr, d_sig.alloc((size_t)K * 64));
        HIPCHK(nullptr, d_pk.alloc((size_t)K * 32));
        HIPCHK(nullptr, hipMemcpy(d_sig.p, sig, (size_t)K * 64, hipMemcpyHostToDevice));
        HIPCHK(nullptr, hipMemcpy(d_pk.p, pk, (size_t)K * 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_verify_batch, dim3(blocks), dim3(64), 0, nullptr, (const uint8_t*)d_m.p, (const long long*)d_off.p,
                           (const uint8_t*)d_sig.p, (const uint8_t*)d_pk.p, (int)K, (uint8_t*)d_out.p);
    } else {
        hipLaunchKernelGGL(k_blake2b_batch, dim3(blocks), dim3(64), 0, nullptr, (const uint8_t*)d_m.p, (const long long*)d_off.p, (int)K, (uint8_t*)d_out.p);
    }
    HIPCHK(nullptr, hipGetLastError());
    HIPCHK(nullptr, hipMemcpy(out, d_out.p, (size_t)K * out_per, hipMemcpyDeviceToHost));
    return SW_OK;
}
}  // namespace

extern "C" {
int sw_crypto_verify_batch(int device, int64_t K, const uint8_t* msgs, const int64_t* msg_off, const uint8_t* sig64,
                           const uint8_t* pk32, uint8_t* ok) {
    return crypto_batch(device, K, msgs, msg_off, sig64, pk32, ok, true);
}
int sw_crypto_hash_batch(int device, int64_t K, const uint8_t* msgs, const int64_t* msg_off, uint8_t* out32) {
    return crypto_batch(device, K, msgs, msg_off, nullptr, nullptr, out32, false);
}
}  // extern "C"
