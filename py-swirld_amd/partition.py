"""Multi-GPU partition logic of the hot path — what shards, and the collectives.

The path as a whole does NOT strong-scale (DESIGN.md §8 has the numbers): divide_rounds is a chain of
~300 dependent ~25 us iterations at 256 members / 1 M events, its can_see sweep is bound by the
depth of the DAG, not by the number of columns, and decide_fame is 0.2 ms.  `bench.py --gpus N`
therefore runs independent replicas.  This module holds the one split of north_star that is exact
and cheap to state — the candidate-partitioned decide_fame with an all-reduce of the per-witness
fame table — so that it can be validated (CPU: world_size 2 over gloo against a numpy restatement
of the kernels; GPU: two contexts on one device) and priced:

    every rank holds the same divided hashgraph (replicated divide_rounds);
    rank p runs the elections of the candidate rounds max_c + p, max_c + p + P, ...   (swirld.py:256-272
        is independent per candidate witness given `witnesses` and the voters' strongly-seen sets);
    ONE all-reduce(MAX) over the int8 famous table [R][n] (-1 = undecided / not owned) and the
        per-round `decided` flags merges the parts — north_star's "allreduce of per-witness vote
        bitmasks": R*n bytes, 72 KB at 256 members / 1 M events;
    every rank commits the merged table: identical famous / consensus / new_c on all ranks.

`backend` objects need `decide_fame_partial(part, nparts) -> (famous int8 [R][n], decided uint8 [R])`
and `commit_fame(famous, decided) -> new_c`: `engine.Hashgraph` on a GPU (RCCL), a numpy model in
the CPU tests (gloo).

Round 3 adds the split of the can_see TABLE by event ranges (`chunk_cuts`, `RowExchange`): what one
GPU does with G concurrent chunks inside k_cansee_chunks, G ranks do with one chunk
each — every rank sweeps its range from a halo before it with unknown parents as leaves, which needs
NO communication; only entries whose ancestor is older than the halo are repaired from rows of lower
ranks, fetched by `RowExchange` (none at uniform gossip with the default halo).  The sweep's depth
and the table's memory divide by G; the round loop that follows stays one chain of dependent
iterations (cost_model), so this partitions memory and the sweep, not the pass.
"""
import numpy as np


def candidate_rounds(max_c, R, part, nparts):
    """Candidate rounds owned by `part`: max_c + part, max_c + part + nparts, ... < R."""
    return list(range(max_c + part, R, nparts))


def merge_fame_tables(tables):
    """Element-wise MAX of the parts' (famous, decided) tables: -1 / 0 loses against a decision."""
    fam = np.maximum.reduce([np.asarray(t[0], np.int8) for t in tables])
    dec = np.maximum.reduce([np.asarray(t[1], np.uint8) for t in tables])
    return fam, dec


class PartitionedFame:
    """decide_fame over `world` ranks of a torch.distributed process group (None: single process)."""

    def __init__(self, dist=None, rank=0, world=1, device=None):
        self.dist, self.rank, self.world, self.device = dist, int(rank), int(world), device

    def _allreduce_max(self, arr):
        if self.dist is None or self.world == 1:
            return arr
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)          # RCCL reduces device tensors
        t = t.to(torch.int32)              # (one dtype every backend reduces)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().numpy().astype(arr.dtype)

    def decide_fame(self, backend):
        fam, dec = backend.decide_fame_partial(self.rank, self.world)
        fam = self._allreduce_max(fam)
        dec = self._allreduce_max(dec)
        return backend.commit_fame(fam, dec)


def chunk_cuts(a0, b, parts):
    """Event ranges [cuts[k], cuts[k+1]) of `parts` ranks over the events [a0, b)."""
    return [a0 + (b - a0) * k // parts for k in range(parts + 1)]


class RowExchange:
    """Rows of a table partitioned by event ranges: every rank names the rows it needs from other ranks,
    the owners answer.  Two collectives (requests, answers); object collectives here (gloo in the CPU
    test) — on RCCL the same two steps are all_gathers of padded int32 tensors.  `bytes_moved` counts
    the row payload received by this rank."""

    def __init__(self, dist, rank, world, cuts):
        self.dist, self.rank, self.world, self.cuts = dist, int(rank), int(world), list(cuts)
        self.bytes_moved = 0

    def owner(self, e):
        for k in range(self.world):
            if self.cuts[k] <= e < self.cuts[k + 1]:
                return k
        raise IndexError(e)

    def fetch(self, events, local_rows):
        """{event -> row} for `events` (owned by other ranks); `local_rows(e)` serves this rank's own rows to the
        others.  Collective: every rank calls it, with an empty list when it needs nothing."""
        want = sorted(set(int(e) for e in events))
        all_want = [None] * self.world
        self.dist.all_gather_object(all_want, want)
        mine = {e: np.asarray(local_rows(e)) for w in all_want for e in w if self.owner(e) == self.rank}
        all_rows = [None] * self.world
        self.dist.all_gather_object(all_rows, mine)
        got = {}
        for k, rows in enumerate(all_rows):
            if k == self.rank:
                continue
            for e, r in rows.items():
                if e in want:
                    got[e] = r
                    self.bytes_moved += r.nbytes
        return got


def cost_model(n=256, R=284, iterations=310, iter_us=19.1, sweep_ms=3.1, fame_ms=0.23, step_ms=7.9,
               world=8, coll_us=20.0, link_GBps=50.0, links=7):
    """Back-of-envelope strong-scaling bound for ONE hashgraph over `world` GPUs of one node,
    from this build's measured single-GPU numbers (defaults: 256 members / 1 M events, round 3).
    coll_us = latency of one small RCCL collective over xGMI; link_GBps = effective per-link rate.
    Returns the modelled step time per variant (ms)."""
    out = {"single_gpu_ms": step_ms}
    # (1) candidate-partitioned decide_fame: elections / world + one all-reduce of R*n bytes
    out["fame_partitioned_ms"] = step_ms - fame_ms + fame_ms / world + coll_us * 1e-3 + (R * n) / (link_GBps * 1e9) * 1e3
    # (2) round loop with the candidates of each member evaluated on its owner rank: the tally (about
    #     half of an iteration) divides by world, every iteration pays one all-gather of found[] (n ints)
    tally_us = iter_us * 0.45
    out["round_loop_partitioned_ms"] = step_ms - iterations * iter_us * 1e-3 + iterations * (iter_us - tally_us + tally_us / world + coll_us) * 1e-3
    # (3) column-sharded can_see: the sweep is bound by the DAG depth (one dependent step per level
    #     whatever the number of columns), so its time does not shrink; the rows must then be
    #     all-gathered because every tally reads whole rows: N*n*4 bytes * (world-1)/world per rank
    gather_ms = (1_000_000 * n * 4) * (world - 1) / world / (links * link_GBps * 1e9) * 1e3
    out["can_see_sharded_ms"] = step_ms + gather_ms
    # (4) the table split by EVENT RANGES (chunk_cuts / RowExchange; what k_cansee_chunks does with G chunks on
    #     one GPU): every rank sweeps its range from a halo, no communication at uniform gossip, sweep
    #     depth and memory / world.  The round loop stays ONE chain of dependent iterations: it runs rank
    #     after rank (each over the rounds whose candidates it owns), handing over the loop state and the
    #     rows of the last ~4 rounds of events (the band of the next rank's first tallies) — so the pass
    #     saves at most what the single GPU still waits for its own sweep (~0.2 ms) and pays the hand-overs
    band_rows_bytes = 4 * (1_000_000 // R) * n * 4
    handover_ms = coll_us * 1e-3 + band_rows_bytes / (link_GBps * 1e9) * 1e3
    out["can_see_event_ranges_ms"] = step_ms - 0.2 + (world - 1) * handover_ms
    out["best_speedup"] = step_ms / min(out["fame_partitioned_ms"], out["round_loop_partitioned_ms"], out["can_see_sharded_ms"],
                                        out["can_see_event_ranges_ms"])
    return out
