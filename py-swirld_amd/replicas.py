"""Multi-GPU driver helpers of bench.py: the torchrun environment, the timed region (barrier + device
synchronisation on both sides, max over ranks) and the whole-job aggregate.  `value` of a multi-GPU bench line is N
independent replicas — one hashgraph view per GPU, which is what a deployment has (every member holds its own view);
the one-hashgraph split (partition.StrongSplit: event-range can_see sweeps, RCCL row broadcasts, partitioned
decide_fame) is timed next to it as `value_strong` (DESIGN.md §8 says why it cannot win).  torch.distributed is RCCL
on GPUs, gloo in the CPU tests."""
import os
import time


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def replica_seed(base_seed, rank):
    """Every replica gets its own hashgraph (distinct generator seed)."""
    return int(base_seed) + int(rank)


class Replicas:
    """Barrier + max-over-ranks timing of N independent replicas."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = dist_env()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return float(seconds)
        import torch
        t = torch.tensor([float(seconds)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, fn, steps, sync=None):
        """Runs fn(i) for `steps` steps between two barriers; returns the max over ranks."""
        self.barrier()
        if sync:
            sync()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        if sync:
            sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def aggregate_throughput(self, units_per_rank_step, steps, seconds_max):
        """Whole-job units/s = units all ranks processed / slowest rank's time."""
        total_units = self.sum_over_ranks(units_per_rank_step * steps)
        return total_units / seconds_max

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
