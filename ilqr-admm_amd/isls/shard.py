"""Multi-GPU sharding of the trajectory batch: trajectories are independent, so rank r of W owns a
contiguous slice of the batch and the only exchange is one tiny all-reduce per outer iteration that carries the
convergence summary [sum cost, max prim, max dual, #active, #failed] of every shard.

One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm ("gloo" in the CPU tests).
The five numbers of every rank travel in ONE sum-all-reduce of a [W,5] buffer in which each rank fills only its
own row (a gather expressed as the single all-reduce the design calls for; 40 B per rank, latency-bound).
"""
import torch


def shard_range(batch, rank, world):
    """Contiguous slice [lo, hi) of a global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_convergence(out5, rank, world, group=None, buf=None):
    """out5: this shard's [sum cost, max prim, max dual, #active, #failed] (device tensor for nccl, cpu for gloo).
    Returns (summary[5] for the whole batch, per-rank table [W,5]); asynchronous w.r.t. the host for nccl."""
    if buf is None:
        buf = torch.zeros(world, 5, dtype=out5.dtype, device=out5.device)
    else:
        buf.zero_()
    buf[rank].copy_(out5)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    total = torch.stack([buf[:, 0].sum(), buf[:, 1].max(), buf[:, 2].max(), buf[:, 3].sum(), buf[:, 4].sum()])
    return total, buf


def allreduce_table(table, world, group=None):
    """The collective of an outer iteration on a [W,5] table whose row `rank` this shard has just filled and whose other
    rows are zero (`Engine.reduce(table=..., rank=...)` writes it that way in one launch): ONE in-place sum-all-reduce,
    nothing else on the stream.  With world == 1 there is nothing to exchange."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return table


def summarize(table):
    """[sum cost, max prim, max dual, #active, #failed] of the whole batch from the gathered [W,5] table."""
    return torch.stack([table[:, 0].sum(), table[:, 1].max(), table[:, 2].max(), table[:, 3].sum(), table[:, 4].sum()])
