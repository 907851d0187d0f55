"""Multi-GPU sharding of the trajectory batch: trajectories are independent, so rank r of W owns a
contiguous slice of the batch and the only exchange is one tiny all-reduce per outer iteration that carries the
convergence summary [sum cost, max prim, max dual, #active, #failed] of every shard -- started asynchronously
(`TableExchange`) so that no kernel of the next iteration waits for it.

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


class TableExchange:
    """The per-iteration collective off the critical path: `post()` lets the shard write its row of a [W,5] table
    (`fill(table, rank)`, e.g. `Engine.reduce`) and starts the sum-all-reduce WITHOUT making the compute stream wait for it
    (`async_op=True`: RCCL runs it on its own stream behind the row's kernel), so the next outer iteration's kernels start
    while the 40 bytes per rank are still crossing xGMI.  Nothing on the device consumes the table -- it is the host's
    convergence summary -- so the host reads it late: `latest()` returns the newest table whose exchange has been waited
    for, `finish()` waits for everything posted.  `depth` tables rotate; a table is waited for before it is refilled, which
    never stalls in practice (the exchange of `depth` iterations ago is long done)."""

    def __init__(self, world, rank, dtype=torch.float64, device="cpu", group=None, depth=2):
        self.world, self.rank, self.group, self.depth = int(world), int(rank), group, int(depth)
        self.tables = [torch.zeros(self.world, 5, dtype=dtype, device=device) for _ in range(self.depth)]
        self.work = [None] * self.depth
        self.posted = 0

    def _wait(self, k):
        if self.work[k] is not None:
            self.work[k].wait()              # nccl: the current stream waits (no host block); gloo: the host waits
            self.work[k] = None

    def post(self, fill):
        """fill(table, rank) must leave this shard's numbers in row `rank` and zeros in the other rows."""
        k = self.posted % self.depth
        self._wait(k)
        fill(self.tables[k], self.rank)
        if self.world > 1:
            import torch.distributed as dist
            self.work[k] = dist.all_reduce(self.tables[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.posted += 1
        return self.tables[k]

    def latest(self, lag=1):
        """Table posted `lag` posts ago (lag = 1: the one before the newest), its exchange waited for; None before that."""
        i = self.posted - 1 - lag
        if i < 0 or lag >= self.depth:
            return None
        self._wait(i % self.depth)
        return self.tables[i % self.depth]

    def finish(self):
        """Wait for every exchange posted; returns the newest table (None when nothing was posted)."""
        for k in range(self.depth):
            self._wait(k)
        return self.tables[(self.posted - 1) % self.depth] if self.posted else None


def summarize(table):
    """[sum cost, max prim, max dual, #active, #failed] of the whole batch from the gathered [W,5] table."""
    return torch.stack([table[:, 0].sum(), table[:, 1].max(), table[:, 2].max(), table[:, 3].sum(), table[:, 4].sum()])
