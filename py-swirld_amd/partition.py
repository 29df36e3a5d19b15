"""Multi-GPU partition logic of the hot path — what shards, and the collectives.

The path as a whole does NOT strong-scale (DESIGN.md §8 has the numbers): divide_rounds is a chain of
~300 dependent ~19 us iterations at 256 members / 1 M events, its can_see sweep is bound by the
depth of the DAG, not by the number of columns, and decide_fame is 0.2 ms.  `bench.py --gpus N`
reports independent replicas as `value` and the one-DAG split below as `value_strong`.  First the
candidate-partitioned decide_fame with an all-reduce of the per-witness fame table (CPU: world_size 2
over gloo against a numpy restatement of the kernels; GPU: several contexts on one device):

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

The split of the can_see TABLE by event ranges (`chunk_cuts`): what one GPU does with G concurrent
chunks inside k_cansee_chunks, G ranks do with one range each — every rank sweeps its range
from a halo before it with unknown parents as leaves, which needs NO communication; only entries whose
ancestor is older than the halo are repaired from rows of lower ranks (none at uniform gossip with the
default halo).  Round 4 makes it a device path: `StrongSplit` drives the C-ABI (sw_cansee_range /
sw_cansee_repair / sw_export_rows / sw_import_rows, include/swirld_hip.h) and moves the rows with
broadcasts of int32 tensors (RCCL on GPUs, gloo in the CPU tests); `bench.py` times it as `value_strong`.
`RowExchange` — rows fetched on demand, as all_gathers of padded int32 tensors — is the variant for a table
that does not fit one GPU.  The sweep's depth divides by G; the round loop that follows stays one chain of
dependent iterations and runs on every rank, so this divides the sweep and the elections, not the pass.
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
    the owners answer.  Two collectives — requests and answers — as all_gathers of PADDED int32 tensors
    (RCCL-capable: one dtype, equal shapes on every rank; the pad widths come from one all-reduce(MAX)).
    The building block of a table too large for one GPU, where rows are fetched on demand instead of being
    broadcast (StrongSplit below broadcasts whole ranges).  `bytes_moved` counts the row payload received."""

    def __init__(self, dist, rank, world, cuts, device=None):
        self.dist, self.rank, self.world, self.cuts = dist, int(rank), int(world), list(cuts)
        self.device = device
        self.bytes_moved = 0

    def owner(self, e):
        for k in range(self.world):
            if self.cuts[k] <= e < self.cuts[k + 1]:
                return k
        raise IndexError(e)

    def _gather(self, t):
        out = [t.new_empty(t.shape) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return out

    def fetch(self, events, local_rows):
        """{event -> row} for `events` (owned by other ranks); `local_rows(e)` serves this rank's own rows to the
        others.  Collective: every rank calls it, with an empty list when it needs nothing."""
        import torch
        want = sorted(set(int(e) for e in events))
        cnt = torch.tensor([len(want)], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(cnt, op=self.dist.ReduceOp.MAX)
        width = int(cnt.item())
        if width == 0:
            return {}
        req = torch.full((width,), -1, dtype=torch.int32, device=self.device)
        if want:
            req[:len(want)] = torch.tensor(want, dtype=torch.int32, device=self.device)
        all_req = [r.cpu().numpy() for r in self._gather(req)]
        # my answers: one row per request that I own, in (requesting rank, position) order, padded
        mine = [(k, i, int(e)) for k, rq in enumerate(all_req) for i, e in enumerate(rq) if e >= 0 and self.owner(int(e)) == self.rank]
        rows = [np.asarray(local_rows(e), np.int32) for _, _, e in mine]
        ncols = torch.tensor([rows[0].shape[0] if rows else 0, len(rows)], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(ncols, op=self.dist.ReduceOp.MAX)
        cols, depth = int(ncols[0].item()), int(ncols[1].item())
        ans = torch.full((depth, cols + 2), -1, dtype=torch.int32, device=self.device)   # [requesting rank, event, row...]
        for j, ((k, _i, e), r) in enumerate(zip(mine, rows)):
            ans[j, 0], ans[j, 1] = k, e
            ans[j, 2:] = torch.from_numpy(r).to(ans.device)
        got = {}
        for k, a in enumerate(self._gather(ans)):
            if k == self.rank:
                continue
            a = a.cpu().numpy()
            for row in a:
                if row[0] == self.rank and int(row[1]) in want:
                    got[int(row[1])] = row[2:].copy()
                    self.bytes_moved += (len(row) - 2) * 4
        return got


class HipRangeBackend:
    """engine.Hashgraph + device buffers (torch tensors) for StrongSplit.  The C-ABI orders its copies against the
    stream the collectives are enqueued on (torch's current stream): no host synchronisation in between."""

    def __init__(self, hashgraph, device):
        import torch
        from . import _lib
        _lib.require_single_hip_runtime("HipRangeBackend")   # (a torch stream handle is passed through the C-ABI)
        self.h, self.device, self.torch = hashgraph, device, torch

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def row_buffer(self, K):
        return self.torch.empty(int(K) * self.h.row_stride, dtype=self.torch.int32, device=self.device)

    def cansee_range(self, a, K):
        self.h.cansee_range(a, K)

    def cansee_repair(self, a, K):
        self.h.cansee_repair(a, K)

    def export_rows(self, a, K, buf):
        self.h.export_rows(a, K, buf.data_ptr(), self._stream())

    def import_rows(self, a, K, buf):
        self.h.import_rows(a, K, buf.data_ptr(), self._stream())

    def divide_rounds(self, a, K):
        self.h.divide_rounds(a, K)

    def decide_fame_partial(self, part, nparts):
        return self.h.decide_fame_partial(part, nparts)

    def commit_fame(self, fam, dec):
        return self.h.commit_fame(fam, dec)


class HostStagedRangeBackend(HipRangeBackend):
    """The same C-ABI calls with the travelling rows staged through HOST tensors: for process groups that cannot move
    device memory (gloo in the functional tests; `bench.py --backend gloo --one-device`).  RCCL moves device tensors
    directly (HipRangeBackend)."""

    def __init__(self, hashgraph, device):
        super().__init__(hashgraph, device)
        self._keep = []

    def row_buffer(self, K):
        return self.torch.empty(int(K) * self.h.row_stride, dtype=self.torch.int32)

    def export_rows(self, a, K, buf):
        tmp = self.torch.empty(buf.numel(), dtype=self.torch.int32, device=self.device)
        self.h.export_rows(a, K, tmp.data_ptr(), self._stream())
        buf.copy_(tmp)                  # (a synchronising copy on the current stream, which waits for the export)

    def import_rows(self, a, K, buf):
        tmp = buf.to(self.device)
        self._keep.append(tmp)          # the import copies asynchronously: the staging tensor outlives the call
        self.h.import_rows(a, K, tmp.data_ptr(), self._stream())

    def divide_rounds(self, a, K):
        super().divide_rounds(a, K)     # (host-blocking: every import enqueued before it has completed on return)
        if a + K >= self.h.num_events:
            self._keep.clear()


class StrongSplit:
    """ONE hashgraph over the `world` ranks of a torch.distributed group (north_star: "events are partitioned
    across the GPUs ... allreduce of per-witness vote bitmasks").  Every rank holds the whole
    hashgraph (16 B per event) and the voting state; what is divided is the can_see SWEEP — rank k computes the rows
    of its event range only, from a halo, with NO communication (sw_cansee_range) — and the ELECTIONS
    (PartitionedFame).  The rows then travel: range j is broadcast from rank j (RCCL over xGMI; int32 rows), in
    ascending order, every broadcast enqueued up front so that range j + 1 is in flight while the round loop
    works on range j.  The round loop itself is one chain of dependent iterations and runs on every rank
    (replicated): it is what bounds the pass.

    divide_rounds(backend, N) leaves every rank exactly as a single sw_divide_rounds(0, N) would;
    decide_fame(backend) as sw_decide_fame would."""

    def __init__(self, dist, rank, world, device=None):
        self.dist, self.rank, self.world, self.device = dist, int(rank), int(world), device
        self.fame = PartitionedFame(dist, rank, world, device)
        self._bufs = {}

    def _buffers(self, backend, cuts):
        key = tuple(cuts)
        if key not in self._bufs:   # (kept across steps: one table's worth of staging)
            self._bufs = {key: [backend.row_buffer(cuts[j + 1] - cuts[j]) for j in range(self.world)]}
        return self._bufs[key]

    def divide_rounds(self, backend, N, first=0):
        cuts = chunk_cuts(first, first + N, self.world)
        if self.world == 1:
            backend.divide_rounds(first, N)
            return cuts
        a, K = cuts[self.rank], cuts[self.rank + 1] - cuts[self.rank]
        backend.cansee_range(a, K)                       # asynchronous; every rank at once
        bufs = self._buffers(backend, cuts)
        works = []
        for j in range(self.world):                      # ascending: a range is final once the ranges below it are
            if j == self.rank:
                backend.cansee_repair(a, K)              # device-gated: nothing at uniform gossip
                backend.export_rows(a, K, bufs[j])
            w = self.dist.broadcast(bufs[j], src=j, async_op=True)
            if j != self.rank:
                w.wait()                                  # (the collective's stream, not the host, on RCCL)
                backend.import_rows(cuts[j], cuts[j + 1] - cuts[j], bufs[j])
            works.append(w)
        for j in range(self.world):                      # the round loop, range after range, rows in place
            backend.divide_rounds(cuts[j], cuts[j + 1] - cuts[j])
        for w in works:
            w.wait()
        return cuts

    def decide_fame(self, backend):
        return self.fame.decide_fame(backend)


def emulate_strong_split(backends, N, first=0):
    """The same protocol with all ranks in ONE process (contexts on one GPU, or CPU models): a broadcast becomes
    export by the owner + import by the others.  What the GPU tests and `bench.py --emulate-parts` run; no
    parallelism, same calls in the same order per rank."""
    world = len(backends)
    cuts = chunk_cuts(first, first + N, world)
    if world == 1:
        backends[0].divide_rounds(first, N)
        return cuts
    for k, b in enumerate(backends):
        b.cansee_range(cuts[k], cuts[k + 1] - cuts[k])
    for j in range(world):
        a, K = cuts[j], cuts[j + 1] - cuts[j]
        backends[j].cansee_repair(a, K)
        buf = backends[j].row_buffer(K)
        backends[j].export_rows(a, K, buf)
        for k, b in enumerate(backends):
            if k != j:
                b.import_rows(a, K, buf)
    for b in backends:
        for j in range(world):
            b.divide_rounds(cuts[j], cuts[j + 1] - cuts[j])
    return cuts
