/*
 * swirld_hip.h — C-ABI of the MI355X-native virtual-voting hot path of py-swirld.
 *
 * The reference (Lapin0t/py-swirld) has no FFI or plugin interface: its hot path
 * is three methods of one Python class, `Node` (swirld.py:36-328).  This header is
 * therefore the boundary a maintainer would bind with ctypes from inside `Node`
 * (see INTEGRATION.md); each entry point names the reference code it replaces.
 *
 * Conventions
 *  - plain C types only; the caller owns every buffer (typically numpy arrays);
 *  - every function returns SW_OK (0) or a negative errno-style code and never
 *    throws; sw_last_error() gives a human-readable message for the last failure;
 *  - one context per Node view; a context is NOT thread-safe (the reference is
 *    single-threaded by design, README.md:27-28, swirld.py:152); independent
 *    contexts may live on different devices/streams;
 *  - events are addressed by DENSE INDEX = the order in which they were appended,
 *    which must be a topological order (parents before children), exactly the
 *    order `Node.add_event` is called in (swirld.py:114-120, 133-144); members are
 *    addressed by dense index 0..n-1 (the host glue keeps pk -> index);
 *  - "absent" event / parent / witness is -1.
 */
#ifndef SWIRLD_HIP_H
#define SWIRLD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_OK          0
#define SW_EIO        (-5)   /* HIP runtime error (message in sw_last_error)            */
#define SW_ENOMEM     (-12)
#define SW_ENODEV     (-19)  /* no usable GPU / device index out of range               */
#define SW_EINVAL     (-22)
#define SW_ERANGE     (-34)  /* index / round range outside the stored hashgraph        */
#define SW_EOVERFLOW  (-75)  /* total stake too large for the 32-bit tally              */
#define SW_ENOTSUP    (-95)  /* outside the supported domain (forks refused, exact path) */

#define SW_MAX_MEMBERS 1024

typedef struct sw_ctx sw_ctx;

/* ABI version of this header (bumped on any signature change). */
int sw_version(void);

/*
 * Context = the voting state of one Node (swirld.py:38-72): `stake`, `tot_stake`,
 * `min_s` (handled as the exact integer tests 3x > 2T and 2x > T, Appendix A Q1),
 * module constant C = coin_period (swirld.py:17).
 * device = HIP device ordinal.  Fails with SW_ENODEV when no GPU is present: there
 * is no CPU fallback in this library.
 */
int sw_create(int n_members, const uint64_t* stake, int coin_period, int device, sw_ctx** out);
int sw_destroy(sw_ctx* ctx);
const char* sw_last_error(const sw_ctx* ctx);   /* ctx may be NULL: last create() error */

/* Pre-size device storage for n_events events (optional; append grows on demand). */
int sw_reserve(sw_ctx* ctx, int64_t n_events);

/*
 * Mirror of Node.add_event (swirld.py:114-120) for K events in topological order:
 * stores creator / parents, computes `height` (0 for roots, 1+max(parent heights)).
 * self_parent/other_parent are dense indices (both -1 for a root, swirld.py:85-87).
 * t = Event.t (float64 timestamp), sig64 = Event.s (64-byte signature; its first
 * byte's top bit is the coin bit of swirld.py:272, all 64 bytes feed the whitening
 * of swirld.py:281-285).  t and sig64 may be NULL (zeros are stored).
 * Validation mirrors is_valid_event's structural half (swirld.py:104-108): parents
 * must exist, self-parent must be by the same creator, other-parent by another.
 * A fork (an event whose self-parent is not its creator's latest event, or a second
 * root of one member) is stored, as the reference stores it (no fork detection,
 * README.md:84), and moves the context to the EXACT path (sw_set_forks below): from
 * then on every call runs the reference's own statements on the device, one wavefront,
 * results identical to the reference's on forked input — and far slower than the
 * round-synchronous path, which needs one self-parent chain per member.
 */
int sw_append_events(sw_ctx* ctx, int64_t K, const int32_t* creator, const int32_t* self_parent,
                     const int32_t* other_parent, const double* t, const uint8_t* sig64);
int64_t sw_num_events(const sw_ctx* ctx);

/*
 * Node.divide_rounds(events) (swirld.py:187-222) for the K events [first, first+K):
 * fills can_see rows, round numbers and the witness table.  `first` must equal the
 * number of events already divided (the reference processes every new event exactly
 * once, in order: swirld.py:325).
 */
int sw_divide_rounds(sw_ctx* ctx, int64_t first, int64_t K);

/*
 * Node.decide_fame() (swirld.py:224-277).  Writes the newly decided rounds (`new_c`,
 * swirld.py:274-277) in ascending order to new_rounds[0..*n_new) and adds them to
 * the consensus set.  cap = capacity of new_rounds; SW_ERANGE if too small (state is
 * still updated, *n_new holds the required size).
 */
int sw_decide_fame(sw_ctx* ctx, int32_t* new_rounds, int cap, int* n_new);

/*
 * Multi-GPU building block (no reference counterpart: the reference is one thread): the
 * elections of decide_fame are independent per candidate witness (swirld.py:256-272 given
 * `witnesses` and the voters' strongly-seen sets), so `nparts` contexts holding the same divided
 * hashgraph can each run the candidate rounds max_c + part, max_c + part + nparts, ... .
 * sw_decide_fame_partial writes this part's view — famous[R][n_members] (-1 undecided or not
 * owned) and decided[R] (1: every witness of the round is decided, swirld.py:274-275) — and commits
 * nothing.  The element-wise MAX of all parts' tables (one all-reduce; py-swirld_amd/partition.py)
 * given to sw_commit_fame on every part leaves each context exactly as sw_decide_fame() would:
 * same famous table, consensus set and new_c — and nothing else: the deciding call / voter of a
 * witness (what Node.votes' existence rule needs) is recorded only on the part that ran its
 * election, so after a sw_commit_fame sw_get_vote returns SW_ENOTSUP until sw_reset / sw_rewind.
 */
int sw_decide_fame_partial(sw_ctx* ctx, int part, int nparts, int8_t* famous, uint8_t* decided,
                           int r_cap, int* r_out);
int sw_commit_fame(sw_ctx* ctx, const int8_t* famous, const uint8_t* decided, int R,
                   int32_t* new_rounds, int cap, int* n_new);

/*
 * Multi-GPU building block, part 2 (no reference counterpart — its `network` is a dict,
 * swirld.py:40, 337): the can_see table split by EVENT RANGES.  Every part holds the whole hashgraph
 * (sw_append_events: 16 B per event) and computes the can_see rows of its own range only; the rows
 * are exchanged (RCCL broadcast / all-gather of int32 rows, py-swirld_amd/partition.py StrongSplit),
 * after which sw_divide_rounds finds them in place and runs the round loop without sweeping.
 *   sw_cansee_range   sweeps the rows of [first, first + K) from a halo in front of the range
 *                     (the chunk-parallel sweep, k_cansee_chunks: a parent below the halo is a leaf,
 *                     entries the window cannot know are PROVISIONAL and counted).  Asynchronous.
 *   sw_cansee_repair  repairs those entries from the final rows below the range — which must have
 *                     been imported (or computed) before; device-gated: costs nothing when the sweep
 *                     counted none, sweeps the range again from final rows when they are too many.
 *                     Ranges are repaired in ascending order.
 *   sw_export_rows    copies the rows of [first, first + K) to caller-provided DEVICE memory
 *                     (K * row_stride int32, row_stride = members padded to a multiple of 64:
 *                     sw_row_stride), ordered after the sweep / repair; `user_stream` (a hipStream_t,
 *                     may be NULL = the null stream) is made to wait for the copy, so a collective
 *                     enqueued on it afterwards sends complete rows.
 *   sw_import_rows    the opposite direction: waits (on the device) for what `user_stream` has
 *                     enqueued so far, copies the rows into the table and marks them present.
 * Rows present (swept by sw_cansee_range, imported) are not swept again by sw_divide_rounds; a
 * sw_divide_rounds call must lie entirely inside or entirely outside the present ranges.
 * Fast path with at most 256 members and the plain (non-windowed) table only: SW_ENOTSUP otherwise.
 * sw_get_range_stats synchronises and returns the provisional entries counted, the entries the repair
 * changed and the ranges swept a second time, since the last sw_rewind.
 */
int sw_row_stride(const sw_ctx* ctx);
int sw_cansee_range(sw_ctx* ctx, int64_t first, int64_t K);
int sw_cansee_repair(sw_ctx* ctx, int64_t first, int64_t K);
int sw_export_rows(sw_ctx* ctx, int64_t first, int64_t K, void* dst_device, void* user_stream);
int sw_import_rows(sw_ctx* ctx, int64_t first, int64_t K, const void* src_device, void* user_stream);
int sw_get_range_stats(sw_ctx* ctx, int64_t* provisional, int64_t* repaired, int64_t* resweeps);

/*
 * Multi-GPU building block, part 3 (last bullet; no reference counterpart): ONE hashgraph's ROUND LOOP over
 * `parts` linked contexts — one per GPU of this process, or several on one GPU — split INSIDE an iteration (swirld.py:208-216
 * once per candidate event): the band events whose threshold masks an iteration builds, and the members whose candidate
 * windows it tallies, are dealt to the parts; every part stores what it produces (mask rows and popcounts, round numbers and
 * sees-masks of the band events, its members' verdict words) into the tables of EVERY part through peer-mapped device memory
 * (plain stores and atomics over xGMI: no collective inside an iteration), and the parts' streams meet at the iteration's two
 * kernel boundaries through events.  Everything else of sw_divide_rounds stays replicated: each part holds the whole hashgraph
 * and sweeps the whole can_see table.
 *   sw_split_link    links `parts` (2 .. 8) contexts that hold the SAME events (same sw_append_events calls) and are divided up
 *                    to the same point; peer access between their devices is enabled.  Unit stake only (SW_ENOTSUP otherwise);
 *                    not on the exact (forked) path, not with the windowed table.
 *   afterwards       EVERY part calls sw_divide_rounds with the same arguments, each from its own host thread (the calls meet
 *                    iteration by iteration; a part that never arrives makes the others fail with SW_EIO after 30 s);
 *                    sw_decide_fame / sw_find_order / getters per context as usual — each part ends with the complete state.
 *   sw_rewind        of linked contexts: EVERY part rewinds and is synchronised (sw_synchronize) before ANY part divides again — a
 *                    part's first band kernel stores into the others' tables, which their rewind would wipe afterwards.
 *   sw_split_unlink  dissolves the group (also done by sw_destroy of any of its contexts).
 * Results are those of an unlinked context (tests/test_gpu_split_loop.py: parts on one GPU against the oracle).
 */
int sw_split_link(sw_ctx* const* ctxs, int parts);
int sw_split_unlink(sw_ctx* ctx);

/*
 * Node.find_order(new_c) (swirld.py:280-311) for the given rounds (processed in
 * ascending order like sorted(new_c)).  Appends to the internal `transactions` list
 * and writes the newly ordered event indices, in final order, to out_events.
 */
int sw_find_order(sw_ctx* ctx, const int32_t* rounds, int n_rounds, int32_t* out_events,
                  int64_t cap, int64_t* n_out);

/* ---- getters (lazy dict views of the Node state; all copy device -> caller) ---- */
int sw_get_height(sw_ctx* ctx, int64_t first, int64_t K, int32_t* out);          /* Node.height   */
int sw_get_round(sw_ctx* ctx, int64_t first, int64_t K, int32_t* out);           /* Node.round    */
/* Node.can_see rows: out[K][n_members], entry = latest event of that member seen, -1 absent */
int sw_get_can_see(sw_ctx* ctx, int64_t first, int64_t K, int32_t* out);
int sw_max_round(sw_ctx* ctx, int* out);                                          /* max(witnesses) */
/* Node.witnesses[r][member] for r in [r0, r1): out[(r1-r0)][n_members]; dict order inside
 * a round = ascending event index (registration order, swirld.py:197, 222). */
int sw_get_witnesses(sw_ctx* ctx, int r0, int r1, int32_t* out);
/* Node.famous for the same table: -1 undecided, 0 False, 1 True (swirld.py:263). */
int sw_get_famous(sw_ctx* ctx, int r0, int r1, int8_t* out);
/* Node.famous keyed by event (swirld.py:64) for the events [first, first+K): -1 = undecided or not a
 * witness, else 0 / 1.  Differs from the slot view only with forks (a replaced witness keeps its entry). */
int sw_get_famous_events(sw_ctx* ctx, int64_t first, int64_t K, int8_t* out);
/* Node.consensus membership for r in [r0, r1): 1 if r in consensus (swirld.py:276). */
int sw_get_consensus(sw_ctx* ctx, int r0, int r1, uint8_t* out);
/* Diagnostic: per event the member bitmask {c_ : round[can_see[e][c_]] == round[e]}
 * (the inner test of swirld.py:211-214 / 250-252); out[K][ceil(n/64)] little-endian words. */
int sw_get_sees_mask(sw_ctx* ctx, int64_t first, int64_t K, uint64_t* out);
/* Diagnostic: Node.votes[voter][candidate] for voter = witness (rv, mv), candidate =
 * witness (rc, mc): -1 = no entry, 0/1 = vote (swirld.py:258-272).  Recomputed from the voter
 * masks with the semantics of one batch decide_fame() call; valid after sw_decide_fame. */
int sw_get_vote(sw_ctx* ctx, int rv, int mv, int rc, int mc, int8_t* out);
/* Node.transactions / Node.idx: total ordered so far, and a slice of the order. */
int sw_num_ordered(sw_ctx* ctx, int64_t* out);
int sw_get_transactions(sw_ctx* ctx, int64_t first, int64_t K, int32_t* out);

/*
 * Gossip side, from the device-resident state.
 * sw_get_known_heights: what Node.sync puts into its request (swirld.py:125-126):
 *   out[member] = height of the newest event of that member the (divided) event `head_event`
 *   can see, -1 if none.
 * sw_sync_diff: what Node.ask_sync answers (swirld.py:154-161): the events a peer that reported
 *   known_height[member] (-1: member unknown to it) is missing, as chain position ranges
 *   [pos_first[m], pos_end[m]) of every member's self-parent chain — the ancestors-or-self of
 *   `head_event` above the peer's heights, the head always included.  Equals the reference's
 *   height-pruned BFS as a SET whenever the heights come from a real can_see row; for arbitrary
 *   heights it is a superset (a receiver drops what it cannot validate).
 * sw_get_chain_events: the event indices at chain positions [p0, p1) of one member.
 */
int sw_get_known_heights(sw_ctx* ctx, int64_t head_event, int32_t* out);
int sw_sync_diff(sw_ctx* ctx, int64_t head_event, const int32_t* known_height, int32_t* pos_first,
                 int32_t* pos_end, int64_t* n_events);
int sw_get_chain_events(sw_ctx* ctx, int member, int32_t p0, int32_t p1, int32_t* out);

/*
 * Ingest-side crypto in batches — what Node.is_valid_event spends its time in
 * (swirld.py:99-103), stateless, one GPU thread per message; message i = msgs[msg_off[i] .. msg_off[i+1]).
 * sw_crypto_verify_batch: ok[i] = 1 iff libsodium's crypto_sign_verify_detached(sig_i, msg_i, pk_i)
 *   would return 0 (swirld.py:99-100 via pysodium): Ed25519 with libsodium 1.0.18's rejections
 *   (non-canonical S, small-order R or key, non-canonical key).
 * sw_crypto_hash_batch: out32[i] = BLAKE2b-256(msg_i) = crypto_generichash(msg_i), the event id
 *   (swirld.py:95, 103).
 * One signature costs a GPU thread ~1-2 ms of latency (a CPU core: ~60 us): batches of thousands
 * (a bulk sync payload) are where the device wins; Node.sync uses it above a batch-size threshold.
 */
int sw_crypto_verify_batch(int device, int64_t K, const uint8_t* msgs, const int64_t* msg_off,
                           const uint8_t* sig64, const uint8_t* pk32, uint8_t* ok);
int sw_crypto_hash_batch(int device, int64_t K, const uint8_t* msgs, const int64_t* msg_off, uint8_t* out32);

/* Exact work counters of the calls so far: used by bench.py's roofline. */
typedef struct sw_counters {
    int64_t events_divided;      /* events through divide_rounds                          */
    int64_t rounds;              /* max round + 1                                          */
    int64_t tally_evals;         /* strongly-sees tallies evaluated by the bulk round loop */
    int64_t round_iterations;    /* bulk round-loop iterations (>= rounds)                 */
    int64_t voter_evals;         /* V: voter tallies in decide_fame (swirld.py:247-254)    */
    int64_t majority_evals;      /* P2: majority() evaluations, d >= 2 (swirld.py:260)     */
    int64_t levels;              /* DAG height levels swept by the can_see kernel          */
    int64_t kernel_launches;
    int64_t far_hops;            /* hop masks rebuilt from rows because the hop lay outside the band */
    int64_t band_events;         /* band events whose threshold mask was built by the round loop      */
    int64_t coin_votes;          /* votes cast in coin rounds, d % C == 0 (swirld.py:267-272)          */
    int64_t coin_flips;          /* ... of which taken from the voter's signature bit (swirld.py:272)  */
    int64_t chunk_sweeps;        /* chunks of the can_see table swept concurrently (k_cansee_chunks)     */
    int64_t chunk_provisional;   /* entries a chunk could not know (ancestor older than its halo)        */
    int64_t chunk_repaired;      /* ... of which changed by the repair kernel                            */
    int64_t chunk_resweeps;      /* chunks swept a second time from final rows (too many to repair)      */
    /* ABI v5 */
    int64_t finalize_from_rows;  /* events whose round / sees-mask no band pass of their own round wrote: recomputed from their rows */
    int64_t order_rounds_host_sorted; /* find_order: rounds the host sorted (a tie on timestamp and the first 8 key bytes)   */
} sw_counters;
int sw_get_counters(sw_ctx* ctx, sw_counters* out);
/* The same for a caller built against another version of this header: copies min(out_bytes, sizeof(sw_counters))
 * bytes (fields are only ever appended), so a shorter struct is never overrun and a longer one keeps its tail. */
int sw_get_counters_sized(sw_ctx* ctx, void* out, size_t out_bytes);
/* Which step-3 kernel of the round loop the most recent sw_divide_rounds used: 0 = k_tally (column lanes, stake-weighted
 * or SW_TALLY_IMPL=0), 1 = k_tally_bits (one wave per candidate slot), 2 = k_tally_tree (one workgroup per member, two-level
 * search).  Without SW_TALLY_IMPL the library chooses per call (DESIGN.md §4 "Which tally"); measurement tools name the
 * kernel they price by this.  No reference counterpart. */
int sw_get_tally_impl(const sw_ctx* ctx);

/* Per-phase GPU time of the most recent divide_rounds / decide_fame call, measured with
 * hipEvents on the context's own stream (ms).  Enabled by sw_set_profiling(ctx, 1). */
typedef struct sw_timings {
    float can_see_ms;        /* span of the can_see stream (overlaps the round loop)     */
    float rounds_ms;         /* round-synchronous strongly-sees loop (all iterations)    */
    float tally_ms;          /* ... of which: the tally kernel (dominant kernel)         */
    int32_t tally_launches;
    float finalize_ms;       /* span of the aux stream: round numbers, sees-masks, witness   */
                             /* rows, voter masks per sub-batch (overlaps the round loop)    */
    float fame_ms;           /* decide_fame: voter tallies + elections                   */
    float total_ms;
    /* per kernel family (hipEvent pairs around each launch, so dispatch gaps are included) */
    float cansee_kernel_ms;  /* sum over the can_see sweep launches                       */
    int32_t cansee_launches;
    float resolve_ms;        /* sum over the k_resolve_band launches that did work         */
    int32_t resolve_launches;
    float elections_ms;      /* the elections kernel of the most recent decide_fame        */
} sw_timings;
int sw_set_profiling(sw_ctx* ctx, int enable);
int sw_get_timings(sw_ctx* ctx, sw_timings* out);
/* Diagnostics: with SW_DEBUG_CLOCKS=1 in the environment at sw_create, the two round-loop kernels
 * stamp their phases (100 MHz clock) per iteration; copies up to cap_words of the
 * [4096 iterations][32] table of the most recent run.  profiles/loop_phases.py reads it. */
int sw_debug_clocks(sw_ctx* ctx, unsigned long long* out, int64_t cap_words);
/* Diagnostics: with SW_DEBUG_CLOCKS=3 every workgroup of the two round-loop kernels stamps the time it was done: copies up
 * to cap_words of the [4096 iterations][2 kernels][2048 workgroups] table.  profiles/block_ends.py reads it. */
int sw_debug_block_clocks(sw_ctx* ctx, unsigned long long* out, int64_t cap_words);

/* Measurement utility (no reference counterpart): forget all voting state (rounds,
 * witnesses, fame, consensus, order) as if divide_rounds had never been called; the
 * appended events stay resident.  Lets bench.py time repeated passes over one DAG. */
int sw_rewind(sw_ctx* ctx);
/* Measurement utility: sw_rewind + forget the appended events as well; device storage stays
 * allocated.  Lets bench.py time repeated end-to-end passes (ingest included) on one context. */
int sw_reset(sw_ctx* ctx);

/*
 * Windowed can_see table (the reference keeps every row forever, swirld.py:69-72,
 * README.md:66-68).  sw_set_window(ctx, 1, chunk_mb) — before the first append — puts the table
 * under HIP virtual memory management: one reserved address range, physical chunks (chunk_mb MB, 0 =
 * 64) mapped as events arrive.  After every sw_find_order the rows no later call can read are
 * evicted (chunks unmapped and recycled): rows older than every member's latest event, than every
 * member's first unordered event, and than the thresholds of the oldest round still in play.
 * Afterwards: sw_get_can_see / the gossip getters on an evicted row, and an appended event whose
 * parent row was evicted, fail with SW_ERANGE; everything else behaves as without a window.
 * sw_get_window: first resident event, bytes of the table currently mapped, number of evictions.
 */
int sw_set_window(sw_ctx* ctx, int enable, int chunk_mb);
int sw_get_window(sw_ctx* ctx, int64_t* first_resident_event, int64_t* resident_bytes, int64_t* evictions);
/* A silent member pins the window at its last event: its latest row is the self-parent of its next event, and
 * its front round keeps that round's thresholds in play (the reference keeps every row: swirld.py:69-72).
 * sw_set_window_lapse(ctx, events > 0) — windowed mode only, default 0 = never — lets a member LAPSE once it
 * has been silent for more than `events` events: it stops holding the window back, and in exchange its further
 * events are refused with SW_ERANGE (so are, as before, events whose other-parent row was evicted) until
 * sw_rewind / sw_reset.  Results for every accepted event stay those of the reference. */
int sw_set_window_lapse(sw_ctx* ctx, int64_t events);

/*
 * Forked hashgraphs (swirld.py:170-184 height-based maxi, :221-222 witness overwrite, README.md:84).
 * sw_set_forks(ctx, 1) [default]: a forked event is accepted and the context switches, once and for
 * good (until sw_reset), to the exact path — csrc/exact.hip.h, the reference's divide_rounds /
 * decide_fame / find_order statement by statement on the device-resident state, the fast path's
 * state handed over as it is.  sw_set_forks(ctx, 0): forked events are refused with SW_ENOTSUP and
 * nothing of the call is stored (what a Node that drops forked events wants).  Not available on the
 * exact path (SW_ENOTSUP): sw_decide_fame_partial / sw_commit_fame, sw_get_vote, sw_get_sees_mask,
 * sw_sync_diff, sw_get_chain_events, the windowed table.
 * sw_get_exact: 1 once the context runs on the exact path.
 * sw_get_witness_order: the members of witnesses[r] in dict insertion order (swirld.py:234, 240) —
 * on the fast path the ascending event index of the table entries; with forks the position of a
 * member's FIRST witness of the round (a fork sibling replaces the value, not the position).
 */
int sw_set_forks(sw_ctx* ctx, int accept);
int sw_get_exact(sw_ctx* ctx, int* out);
int sw_get_witness_order(sw_ctx* ctx, int r, int32_t* members, int* n_out);

/* Block until all work queued on the context's stream is complete. */
int sw_synchronize(sw_ctx* ctx);

/*
 * Host utility (no GPU): synthetic gossip hashgraph with the DAG shape swirld.test()
 * produces (swirld.py:323, 342-344).  Events 0..n-1 are the roots.
 * mode 0: uniform gossip.  mode 1: two cliques, cross-clique probability p0.
 * mode 2: a fraction p0 of members has relative activity p1.  mode 3: stale
 * other-parents (walk back the peer's self-parent chain with probability p0 per step).
 * t (nullable) = float(index); sig64 (nullable) = N*64 seeded random bytes.
 */
int sw_synth_hashgraph(int n, int64_t N, uint64_t seed, int mode, double p0, double p1,
                       int32_t* creator, int32_t* self_parent, int32_t* other_parent,
                       double* t, uint8_t* sig64);

#ifdef __cplusplus
}
#endif
#endif /* SWIRLD_HIP_H */
