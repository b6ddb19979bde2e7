"""Data parallelism for the FNN hot path: one process per GPU, torch.distributed over RCCL.

The reference has no distributed code; this is new.  The partition follows
from its loss being a batch SUM (python/FNN_wnzh.py:173): the dense gradient of a global batch is
the sum of the gradients of its shards, so ONE all-reduce of ONE flat f32 bucket per step is the
only exchange.  The dense bucket is ~0.5 MB (latency-bound on xGMI), hence a single collective,
never one per tensor.  Embedding tables are replicated and each rank applies the sparse-row
updates of its own shard locally (north_star); replicas' tables therefore drift apart on rows
that several ranks touch -- reported, not hidden.  `sparse='exchange'` is
the exact alternative: one all-gather of every rank's (ids, slot gradients) per
step, after which every rank applies ALL row updates in global example order, so that replicas
stay identical to the single-process run (the multi-GPU parity mode).  The decay constant of
the sparse update uses the GLOBAL batch length (python/FNN_wnzh.py:304).  Dropout rows are per
batch, not per example, so every rank must be given the same rows.

Two forms of the step (include/fnn_hip.h):
* native (an FNNEngine on a GPU): the library issues the collective itself on its own stream, between
  the second and the third launch of the ordinary train step (fnn_dp_init: RCCL, the default under the
  `nccl` backend; fnn_dp_init_custom with torch.distributed collectives under any other backend, e.g.
  the gloo rehearsal with every rank on one GPU);
* portable (any engine with step_begin / step_end, e.g. the CPU test double of tests/test_dp.py): the
  collective is issued here between the two calls.
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

    def __init__(self, engine, group=None, sparse='local', native=None, payload=None, collective='rccl'):
        """payload: 'slabs' | 'bucket' | None (= $FNN_DP_PAYLOAD, else slabs) -- what the dense collective of the native step
        carries; collective: 'rccl' (the library's RCCL call, or torch.distributed callbacks under another backend) or 'p2p'
        (one-shot all-reduce over peer pointers inside the update launch; the ranks' exchange regions are opened through
        hipIpc handles passed around with all_gather_object)."""
        import torch.distributed as dist
        assert sparse in ('local', 'exchange') and payload in (None, 'slabs', 'bucket') and collective in ('rccl', 'p2p')
        self.sparse = sparse
        self.payload, self.want_collective = payload, collective
        self.engine = engine
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = hasattr(engine, 'dp_init') if native is None else bool(native)
        self.collective = None
        if self.native:
            self._init_native()

    def _init_native(self):
        """RCCL inside the library when the process group is RCCL's (one GPU per rank); torch.distributed collectives
        as the library's callbacks otherwise."""
        import torch
        dist, eng = self.dist, self.engine
        if dist.get_backend(self.group) == 'nccl':
            torch.cuda.set_device(eng.device)               # the object broadcast below moves through the current device
            box = [eng.dp_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            eng.dp_init(self.rank, self.world, box[0], self.sparse)
            self.collective = 'rccl (library, on the engine stream)'
        else:
            def allreduce(view):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

            def allgather(send, recv):
                parts = list(recv.view(self.world, -1).unbind(0))
                dist.all_gather(parts, send, group=self.group)
                for i, p in enumerate(parts):                  # gloo returns copies for non-contiguous outputs
                    if p.data_ptr() != recv.view(self.world, -1)[i].data_ptr():
                        recv.view(self.world, -1)[i].copy_(p)
            eng.dp_init_custom(self.rank, self.world, allreduce, allgather, self.sparse)
            self.collective = 'torch.distributed/%s (library callbacks)' % dist.get_backend(self.group)
        if self.payload is not None:
            eng.dp_set_payload(self.payload)
        if self.want_collective == 'p2p':
            mine = eng.dp_p2p_export()
            handles = [None] * self.world
            dist.all_gather_object(handles, mine, group=self.group)
            eng.dp_p2p_attach(handles)
            dist.barrier(group=self.group)                     # every rank has opened every region before the first step
            self.collective = 'p2p (peer pointers inside the update launch; region: %s)' % eng.dp_config()['region']
        self.config = eng.dp_config()

    def shard(self, ids, y):
        lo, hi = shard_bounds(len(y), self.world, self.rank)
        return ids[lo:hi], y[lo:hi]

    def train_step_local(self, ids_local, y_local, mask1, mask2, global_batch, want_loss=False):
        """ids_local / y_local: this rank's shard.  Returns the global loss sum if asked."""
        eng = self.engine
        if self.native:                          # three launches + one collective, all inside fnn_train_step
            out = eng.train_step(ids_local, y_local, mask1, mask2, b_size=global_batch, want_loss=want_loss)
            if want_loss:
                import torch
                t = torch.tensor([out['loss']], dtype=torch.float64, device=eng.device if self.dist.get_backend(self.group) == 'nccl' else 'cpu')
                self.dist.all_reduce(t, group=self.group)
                return float(t.item())
            return None
        bucket = eng.step_begin(ids_local, y_local, mask1, mask2, b_size=global_batch)
        work = self._all_reduce(bucket)
        if self.sparse == 'exchange':
            self._exchange_sparse(ids_local, len(y_local), global_batch)
        elif hasattr(eng, 'step_scatter'):
            eng.step_scatter()                   # the sparse-row half runs under the collective
        self._wait(work)
        loss = eng.step_end(want_loss=want_loss)
        if want_loss:
            import torch
            t = torch.tensor([loss], dtype=torch.float64, device=bucket.device)
            self.dist.all_reduce(t, group=self.group)
            return float(t.item())
        return None

    def _wait(self, work):
        """ProcessGroupNCCL's wait() makes the CURRENT stream wait for the collective: it has to be the engine's stream, the one
        step_end launches the dense update on -- waiting on another stream would let the update read the bucket while the
        all-reduce is still writing it."""
        if work is None:
            return
        stream = getattr(self.engine, 'stream', None)
        if stream is not None:
            import torch
            with torch.cuda.stream(stream):
                work.wait()
        else:
            work.wait()

    def train_step(self, ids, y, mask1, mask2, want_loss=False):
        """ids / y: the GLOBAL batch, identical on every rank; each rank trains its shard."""
        ids_l, y_l = self.shard(ids, y)
        return self.train_step_local(ids_l, y_l, mask1, mask2, len(y), want_loss)

    def _exchange_sparse(self, ids_local, n_local, global_batch):
        """All-gather (ids, gx') of every shard, padded to the longest shard with empty (-1) ids so
        that the gathered order is the global example order, then apply the whole batch's row
        updates on this rank."""
        import torch
        eng = self.engine
        gx = eng.sparse_grad(n_local)                                   # [n_local, R] float
        ids_t = ids_local if isinstance(ids_local, torch.Tensor) else torch.as_tensor(ids_local)
        ids_t = ids_t.to(device=gx.device, dtype=torch.int32)
        longest = -(-global_batch // self.world)
        ids_pad = torch.full((longest, ids_t.shape[1]), -1, dtype=torch.int32, device=gx.device)
        gx_pad = torch.zeros((longest, gx.shape[1]), dtype=gx.dtype, device=gx.device)
        stream = getattr(eng, 'stream', None)
        ctx = torch.cuda.stream(stream) if (stream is not None and gx.is_cuda) else _Null()
        with ctx:                                                       # ordered after step_begin's kernels
            ids_pad[:n_local].copy_(ids_t)
            gx_pad[:n_local].copy_(gx)
            ids_all = [torch.empty_like(ids_pad) for _ in range(self.world)]
            gx_all = [torch.empty_like(gx_pad) for _ in range(self.world)]
            self.dist.all_gather(ids_all, ids_pad, group=self.group)
            self.dist.all_gather(gx_all, gx_pad, group=self.group)
            ids_g = torch.cat(ids_all).contiguous()
            gx_g = torch.cat(gx_all).contiguous()
        eng.step_scatter_global(ids_g, gx_g)

    def _all_reduce(self, bucket):
        stream = getattr(self.engine, 'stream', None)
        if stream is not None and bucket.is_cuda:
            import torch
            with torch.cuda.stream(stream):          # ordered after step_begin's kernels
                return self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group)
        return None


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DataParallelSparseRBM(object):
    """The mini-batch sparse CD-1 pre-trainer (rbm_sparse_batch; the throughput mode of SURVEY 8d config 5, NOT the reference's
    online schedule) sharded over the ranks of a process group: every rank holds its contiguous shard of every global mini-batch
    and the table replicated; per mini-batch ONE all-reduce of S*H + H floats (the positional steps and the hidden-bias terms)
    keeps wstep / hidbias identical on every rank, while each rank applies the row updates of its own examples (rows several ranks
    touch drift apart, as in the FNN step's LOCAL mode).  The SNN fine-tune step shards through DataParallelFNN with a bag-mode
    engine (the native step carries the bag-bias gradient in its slabs / bucket); the online trainer is sequential: replicas only.
    `allreduce(view)`: sums a float32 tensor view in place over the ranks (default: torch.distributed.all_reduce on `group`)."""

    def __init__(self, group=None, allreduce=None, device=0):
        import torch
        from . import _capi
        self.torch, self.lib, self.C = torch, _capi.load(), __import__('ctypes')
        self.device = torch.device('cuda', device)
        if allreduce is None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)

            def allreduce(view):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        self._allreduce = allreduce
        from .engine import _tensor_from_ptr
        C = self.C

        def _cb(ctx, buf, n, stream):
            try:
                self.torch.cuda.current_stream(self.device).synchronize()      # the sums are complete before a host-staged collective reads them
                self._allreduce(_tensor_from_ptr(self.torch, buf, n, self.device))
                self.torch.cuda.current_stream(self.device).synchronize()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return -1
        self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(_cb)

    def epoch(self, W, dW, visbias, dvis, hidbias, wstep, vid, vval, unif, M_local, M_global, weightcost=2e-4, rate=1e-4, momentum=0.9):
        """One pass over this rank's examples (device tensors, shapes as rbm_sparse_batch takes them); returns this rank's squared error."""
        C = self.C
        N, S = vid.shape
        H = W.shape[1]
        err = C.c_double()
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.rbm_sparse_batch_dp(W.data_ptr(), dW.data_ptr(), visbias.data_ptr(), dvis.data_ptr(), hidbias.data_ptr(), wstep.data_ptr(),
                                          vid.data_ptr(), vval.data_ptr(), unif.data_ptr(), N, int(M_local), int(M_global), H, S, weightcost, rate, rate,
                                          rate, momentum, C.cast(self._cb, C.c_void_p), None, C.byref(err), st)
        if rc != 0:
            raise RuntimeError((self.lib.rbm_last_error() or b'').decode())
        return err.value
