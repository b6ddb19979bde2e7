"""Data parallelism for the FNN hot path: one process per GPU, torch.distributed over RCCL.

The reference has no distributed code; this is new.  The partition follows
from its loss being a batch SUM (python/FNN_wnzh.py:173): the dense gradient of a global batch is
the sum of the gradients of its shards, so ONE all-reduce of ONE flat f32 bucket per step is the
only exchange.  The dense bucket is ~0.5 MB (latency-bound on xGMI), hence a single collective,
never one per tensor.  Embedding tables are replicated and each rank applies the sparse-row
updates of its own shard locally (north_star); replicas' tables therefore drift apart on rows
that several ranks touch -- reported, not hidden.  The decay constant of
the sparse update uses the GLOBAL batch length (python/FNN_wnzh.py:304).  Dropout rows are per
batch, not per example, so every rank must be given the same rows.
"""


def shard_bounds(n, world, rank):
    """Contiguous shard `rank` of `n` examples (file order kept inside a shard); the first
    n % world shards take one extra example."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallelFNN(object):
    """Wraps an engine exposing step_begin / step_end / stream (FNNEngine on a GPU).  `group` is a
    torch.distributed process group (None = default)."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.engine = engine
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def shard(self, ids, y):
        lo, hi = shard_bounds(len(y), self.world, self.rank)
        return ids[lo:hi], y[lo:hi]

    def train_step_local(self, ids_local, y_local, mask1, mask2, global_batch, want_loss=False):
        """ids_local / y_local: this rank's shard.  Returns the global loss sum if asked."""
        eng = self.engine
        bucket = eng.step_begin(ids_local, y_local, mask1, mask2, b_size=global_batch)
        work = self._all_reduce(bucket)
        if hasattr(eng, 'step_scatter'):
            eng.step_scatter()                   # the sparse-row half runs under the collective
        if work is not None:
            work.wait()
        loss = eng.step_end(want_loss=want_loss)
        if want_loss:
            import torch
            t = torch.tensor([loss], dtype=torch.float64, device=bucket.device)
            self.dist.all_reduce(t, group=self.group)
            return float(t.item())
        return None

    def train_step(self, ids, y, mask1, mask2, want_loss=False):
        """ids / y: the GLOBAL batch, identical on every rank; each rank trains its shard."""
        ids_l, y_l = self.shard(ids, y)
        return self.train_step_local(ids_l, y_l, mask1, mask2, len(y), want_loss)

    def _all_reduce(self, bucket):
        stream = getattr(self.engine, 'stream', None)
        if stream is not None and bucket.is_cuda:
            import torch
            with torch.cuda.stream(stream):          # ordered after step_begin's kernels
                return self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group)
        return None
