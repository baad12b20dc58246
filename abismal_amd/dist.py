"""Multi-GPU plumbing for the mapping path: reads shard over ranks (one process per
GPU, index replicated), and the only collective is the end-of-run sum of the
mapping statistics -- six counters per struct (src/abismal.cpp:865-895), up to
three structs -- over RCCL (backend "nccl") on GPUs, gloo in CPU tests."""
import os

STAT_FIELDS = ("total_reads", "reads_mapped_unique", "reads_mapped_ambiguous", "reads_skipped", "edit_distance",
               "total_bases")


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank`; shards differ in size by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_stats(stats_tensor, elapsed_tensor=None):
    """All-reduce (sum) an int64 tensor of 6 or 18 counters in place; optionally max-reduce a
    float64 elapsed time.  No-op without an initialised process group (single GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return stats_tensor, elapsed_tensor
    dist.all_reduce(stats_tensor, op=dist.ReduceOp.SUM)
    if elapsed_tensor is not None:
        dist.all_reduce(elapsed_tensor, op=dist.ReduceOp.MAX)
    return stats_tensor, elapsed_tensor


def ranks_and_rates(n_reads_rank, elapsed_rank):
    """(ranks that took part, [reads/s of every rank]) -- an all-reduced 1 and an all-gather of each
    rank's own rate; (1, [own rate]) without a process group."""
    import torch
    import torch.distributed as dist
    own = n_reads_rank / elapsed_rank if elapsed_rank > 0 else 0.0
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1, [round(own, 1)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([own], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return int(one.item()), [round(float(t.item()), 1) for t in every]
