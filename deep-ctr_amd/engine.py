"""FNNEngine: PyTorch-ROCm plumbing (device memory, streams) around the C ABI of libfnn_hip.so.

The engine stands where the compiled Theano callables `train` / `predict` stand in the reference
(python/FNN_wnzh.py:177-183) and where its Python gather / sparse-update loops stand
(:87-96, :299-306).  All arithmetic happens in the HIP library; torch only owns buffers.
"""
import ctypes as C

import numpy as np

from . import _capi


class FNNError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "fnn_hip error %d: %s" % (code, msg))
        self.code = code


_ACTS = {'tanh': _capi.FNN_ACT_TANH, 'sigmoid': _capi.FNN_ACT_SIGMOID, 'linear': _capi.FNN_ACT_LINEAR}


class FNNEngine(object):
    """One handle = one GPU + one HIP stream.

    Parameters follow python/FNN_wnzh.py:18-49: hidden1, hidden2, lr, lambda1, lambda_fm, acti_type.
    precision: 'f32' (parity mode: exact-f32 MFMA), 'bf16' (throughput mode) or 'bf16x3' (operands as bf16 pairs, three bf16
    MFMAs per product: 16 significant bits at 4.4x the f32 MFMA rate -- FNN_PREC_BF16X3 of include/fnn_hip.h).
    """

    def __init__(self, n_fields=16, k=11, hidden1=300, hidden2=100, max_batch=4096, precision='bf16',
                 acti_type='tanh', lr=0.001, lambda1=0.0, lambda_fm=0.1, reg_all=False, device=0,
                 mode='fm', hidden0=0):
        import torch
        if not torch.cuda.is_available():
            raise FNNError(_capi.FNN_ERR_HIP, "no HIP device visible to PyTorch-ROCm; the FNN hot path has "
                                              "no CPU fallback")
        self._torch = torch
        self.lib = _capi.load()
        self.device = torch.device('cuda', device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.F, self.K, self.H1, self.H2 = n_fields, k, hidden1, hidden2
        self.xdim = 1 + n_fields * k
        self.bag = mode == 'bag'               # SNN fine-tune input layer (python/SNN_RBM.py:238-291)
        if self.bag:
            self.K = self.xdim = hidden0
        self.max_batch = max_batch
        self.precision = precision
        cfg = _capi.fnn_cfg(n_fields, k, hidden1, hidden2, max_batch,
                            _capi.PRECISIONS[precision],
                            _ACTS[acti_type], 1 if reg_all else 0, lr, lambda1, lambda_fm, device,
                            C.c_void_p(self.stream.cuda_stream),
                            _capi.FNN_MODE_BAG if self.bag else _capi.FNN_MODE_FM, hidden0)
        h = C.c_void_p()
        rc = self.lib.fnn_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise FNNError(rc, (self.lib.fnn_last_error(None) or b'').decode())
        self.h = h
        self.n_rows = 0
        self._bucket = None
        self.dp_world = 1

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc != 0:
            raise FNNError(rc, (self.lib.fnn_last_error(self.h) or b'').decode())

    def close(self):
        if getattr(self, 'h', None):
            self.lib.fnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, a, dtype):
        """numpy / torch -> contiguous torch tensor of `dtype` on this device."""
        torch = self._torch
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dtype).contiguous()
        else:
            t = torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()
        return t

    def to_device(self, ids, y):
        """(ids int32 [N,F], y int32 [N]) as resident device tensors (for evaluate / train slices)."""
        return self._dev(ids, self._torch.int32), self._dev(y, self._torch.int32)

    def _enter(self):
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))

    def _leave(self):
        self._torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def sync(self):
        self._ck(self.lib.fnn_sync(self.h))

    def set_hparams(self, lr, lambda1, lambda_fm):
        self._ck(self.lib.fnn_set_hparams(self.h, lr, lambda1, lambda_fm))

    # ------------------------------------------------------------------ state
    def set_table(self, rows, field_of_row, w0):
        """rows [D,K] float, field_of_row [D] int  (feat_weights / feat_field / w_0,
        python/FNN_wnzh.py:62-84)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        fo = np.ascontiguousarray(field_of_row, dtype=np.int32)
        assert rows.ndim == 2 and rows.shape[1] == self.K and fo.shape[0] == rows.shape[0]
        self._ck(self.lib.fnn_set_table(self.h, rows.ctypes.data, rows.shape[0], fo.ctypes.data,
                                        float(w0), _capi.FNN_MEM_HOST))
        self.n_rows = rows.shape[0]

    def get_table(self):
        out = np.empty((self.n_rows, self.K), dtype=np.float32)
        self._ck(self.lib.fnn_get_table(self.h, out.ctypes.data, _capi.FNN_MEM_HOST))
        return out

    def get_rows(self, row_ids):
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        out = np.empty((ids.shape[0], self.K), dtype=np.float32)
        self._ck(self.lib.fnn_get_rows(self.h, ids.ctypes.data, ids.shape[0], out.ctypes.data,
                                       _capi.FNN_MEM_HOST))
        return out

    def set_bag_bias(self, bb0):
        b = np.ascontiguousarray(bb0, dtype=np.float32)
        assert b.shape == (self.K,)
        self._ck(self.lib.fnn_set_bag_bias(self.h, b.ctypes.data, _capi.FNN_MEM_HOST))

    def get_bag_bias(self):
        b = np.empty(self.K, dtype=np.float32)
        self._ck(self.lib.fnn_get_bag_bias(self.h, b.ctypes.data, _capi.FNN_MEM_HOST))
        return b

    def set_dense(self, p):
        """p: dict w1 [xdim,H1], b1 [H1], w2 [H1,H2], b2 [H2], w3 [H2], b3 scalar."""
        for layer, (wn, bn) in enumerate((('w1', 'b1'), ('w2', 'b2'), ('w3', 'b3')), start=1):
            W = np.ascontiguousarray(p[wn], dtype=np.float32)
            b = np.ascontiguousarray(np.atleast_1d(p[bn]), dtype=np.float32)
            self._ck(self.lib.fnn_set_dense(self.h, layer, W.ctypes.data, b.ctypes.data, _capi.FNN_MEM_HOST))

    def get_dense(self):
        shapes = {1: ((self.xdim, self.H1), (self.H1,)), 2: ((self.H1, self.H2), (self.H2,)),
                  3: ((self.H2,), (1,))}
        out = {}
        for layer, (ws, bs) in shapes.items():
            W = np.empty(ws, dtype=np.float32)
            b = np.empty(bs, dtype=np.float32)
            self._ck(self.lib.fnn_get_dense(self.h, layer, W.ctypes.data, b.ctypes.data, _capi.FNN_MEM_HOST))
            out['w%d' % layer] = W
            out['b%d' % layer] = b if layer < 3 else float(b[0])
        return out

    # ------------------------------------------------------------------ hot path
    def gather(self, ids):
        """A3: ids [B,F] int32 (slot f = field f, -1 = empty) -> x [B,xdim] float32 tensor."""
        torch = self._torch
        ids_t = self._dev(ids, torch.int32)
        B = ids_t.shape[0]
        x = torch.empty((B, self.xdim), dtype=torch.float32, device=self.device)
        self._enter()
        self._ck(self.lib.fnn_gather(self.h, ids_t.data_ptr(), B, x.data_ptr(), _capi.FNN_MEM_DEVICE))
        self._leave()
        return x

    def _step_args(self, ids, y, mask1, mask2):
        torch = self._torch
        return (self._dev(ids, torch.int32), self._dev(y, torch.float32),
                self._dev(mask1, torch.uint8), self._dev(mask2, torch.uint8))

    def train_step(self, ids, y, mask1, mask2, b_size=0, want_p=False, want_gx=False, want_loss=True):
        """One pass of the hot loop body (python/FNN_wnzh.py:296-306).  Returns a dict with the
        optional outputs: 'loss' (float, sum of xent), 'p' [B], 'gx' [B,xdim]."""
        torch = self._torch
        ids_t, y_t, m1, m2 = self._step_args(ids, y, mask1, mask2)
        B = ids_t.shape[0]
        assert m1.numel() == self.H1 and m2.numel() == self.H2 and y_t.numel() == B
        p = torch.empty(B, dtype=torch.float32, device=self.device) if want_p else None
        gx = torch.empty((B, self.xdim), dtype=torch.float32, device=self.device) if want_gx else None
        loss = C.c_float(0.0)
        self._enter()
        self._ck(self.lib.fnn_train_step(
            self.h, ids_t.data_ptr(), y_t.data_ptr(), B, m1.data_ptr(), m2.data_ptr(), int(b_size),
            p.data_ptr() if want_p else None, gx.data_ptr() if want_gx else None, _capi.FNN_MEM_DEVICE,
            C.byref(loss) if want_loss else None))
        self._leave()
        out = {}
        if want_loss:
            out['loss'] = float(loss.value)
        if want_p:
            out['p'] = p
        if want_gx:
            out['gx'] = gx
        return out

    def train_epoch(self, ids_d, yf_d, batch_size, masks1, masks2, first=0, n_steps=None, shadowed=None):
        """Steps first .. first + n_steps - 1 of the hot loop (python/FNN_wnzh.py:293-306) over RESIDENT arrays: batch j is rows
        [j * batch_size, (j + 1) * batch_size) of ids_d int32 [N, F] / yf_d float32 [N] (a short last batch keeps its length as
        b_size, as the reference's line count does), its dropout rows are row j of masks1 [n, H1] / masks2 [n, H2] (uint8, host or
        device).  The next batch is announced to every step (fnn_prefetch_ids), `shadowed` (int32 [n, 3] = (example, field, row),
        sorted by example: DataFM.load_ids(want_shadowed=True)) feeds fnn_set_shadowed per batch.  The loop makes the raw C calls
        itself: at the reference's batch size of 100 the per-call Python of train_step (~85-120 us) is three times the step."""
        torch = self._torch
        assert ids_d.is_cuda and ids_d.dtype == torch.int32 and ids_d.is_contiguous() and yf_d.is_cuda and yf_d.dtype == torch.float32
        N = ids_d.shape[0]
        n_all = (N + batch_size - 1) // batch_size
        n_steps = n_all - first if n_steps is None else n_steps
        m1, m2 = self._dev(masks1, torch.uint8), self._dev(masks2, torch.uint8)
        assert m1.shape == (m1.shape[0], self.H1) and m2.shape == (m2.shape[0], self.H2) and min(m1.shape[0], m2.shape[0]) >= first + n_steps
        lib, h, F = self.lib, self.h, self.F
        ip, yp, p1, p2 = ids_d.data_ptr(), yf_d.data_ptr(), m1.data_ptr(), m2.data_ptr()
        sh = None if shadowed is None or len(shadowed) == 0 else np.ascontiguousarray(shadowed, dtype=np.int32)
        self._enter()
        for j in range(first, first + n_steps):
            lo = j * batch_size
            B = min(batch_size, N - lo)
            if B <= 0:
                break
            nlo = lo + batch_size
            if j + 1 < first + n_steps and nlo < N and B <= 4096:
                rc = lib.fnn_prefetch_ids(h, ip + nlo * F * 4, min(batch_size, N - nlo))
                if rc != 0:
                    self._ck(rc)
            if sh is not None:
                a, b = np.searchsorted(sh[:, 0], [lo, lo + B])
                if b > a:
                    part = sh[a:b].copy()
                    part[:, 0] -= lo
                    self._ck(lib.fnn_set_shadowed(h, part.ctypes.data, len(part), _capi.FNN_MEM_HOST))
            rc = lib.fnn_train_step(h, ip + lo * F * 4, yp + lo * 4, B, p1 + j * self.H1, p2 + j * self.H2, B, None, None,
                                    _capi.FNN_MEM_DEVICE, None)
            if rc != 0:
                self._ck(rc)
        self._leave()
        self._keep = (ids_d, yf_d, m1, m2)

    def set_shadowed(self, tfr):
        """fnn_set_shadowed: int32 [n, 3] = (example t, field, row) of the features that a later feature of the same field
        shadows in the NEXT train_step / step_begin batch; their rows take the sparse update too, as in the reference's loop
        over every feature of a line (python/FNN_wnzh.py:300-306)."""
        a = np.ascontiguousarray(tfr, dtype=np.int32).reshape(-1, 3)
        self._enter()
        self._ck(self.lib.fnn_set_shadowed(self.h, a.ctypes.data if len(a) else None, len(a), _capi.FNN_MEM_HOST))

    def prefetch_ids(self, ids_t):
        """Scheduling hint: start grouping the ids of an upcoming batch (a device int32 tensor that
        will be passed unchanged to train_step / step_begin)."""
        assert ids_t.is_cuda and ids_t.dtype == self._torch.int32 and ids_t.is_contiguous()
        self._enter()
        self._ck(self.lib.fnn_prefetch_ids(self.h, ids_t.data_ptr(), ids_t.shape[0]))

    def step_begin(self, ids, y, mask1, mask2, b_size=0):
        """Data-parallel half step: everything but the dense SGD.  Returns the flat dense-gradient
        bucket as a torch tensor aliasing the library's buffer (all-reduce it, then step_end())."""
        ids_t, y_t, m1, m2 = self._step_args(ids, y, mask1, mask2)
        B = ids_t.shape[0]
        self._enter()
        self._ck(self.lib.fnn_step_begin(self.h, ids_t.data_ptr(), y_t.data_ptr(), B, m1.data_ptr(),
                                         m2.data_ptr(), int(b_size), None, None, _capi.FNN_MEM_DEVICE))
        self._keep = (ids_t, y_t, m1, m2)
        return self.grad_bucket()

    def grad_bucket(self):
        if self._bucket is None:
            ptr, n = C.c_void_p(), C.c_int64()
            self._ck(self.lib.fnn_dense_grad_bucket(self.h, C.byref(ptr), C.byref(n)))
            self._bucket = _tensor_from_ptr(self._torch, ptr.value, n.value, self.device)
        return self._bucket

    # ------------------------------------------------------------------ native data parallelism
    @staticmethod
    def dp_unique_id():
        """Rank 0: the 128-byte ncclUniqueId every rank passes to dp_init (fnn_dp_unique_id)."""
        lib = _capi.load()
        buf = C.create_string_buffer(128)
        rc = lib.fnn_dp_unique_id(buf)
        if rc != 0:
            raise FNNError(rc, (lib.fnn_last_error(None) or b'').decode())
        return buf.raw

    def dp_init(self, rank, world, unique_id, sparse='local'):
        """fnn_dp_init: from now on train_step() is the data-parallel step (RCCL all-reduce of the weight-gradient slabs
        between its second and third launch, on the engine's stream).  Collective: every rank calls it."""
        assert len(unique_id) == 128
        self._torch.cuda.set_device(self.device)
        self._ck(self.lib.fnn_dp_init(self.h, int(rank), int(world), C.c_char_p(unique_id),
                                      _capi.FNN_DP_SPARSE_EXCHANGE if sparse == 'exchange' else _capi.FNN_DP_SPARSE_LOCAL))
        self.dp_world = int(world)

    def dp_init_custom(self, rank, world, allreduce, allgather=None, sparse='local'):
        """fnn_dp_init_custom with Python collectives: allreduce(view) sums a float32 tensor view of the library's buffer over the
        ranks in place; allgather(send_view, recv_view) fills recv (uint8 views; recv = world blocks of len(send)).  Both are
        called with the engine's stream current and must leave their work ordered on it."""
        torch = self._torch

        def _ar(ctx, buf, n, stream):
            try:
                with torch.cuda.stream(self.stream):
                    allreduce(_tensor_from_ptr(torch, buf, n, self.device))
                return 0
            except Exception:          # an exception cannot cross the C frames: report and fail the step
                import traceback
                traceback.print_exc()
                return -1

        def _ag(ctx, send, recv, nbytes, stream):
            try:
                with torch.cuda.stream(self.stream):
                    allgather(_tensor_from_ptr(torch, send, nbytes, self.device, 'u1'),
                              _tensor_from_ptr(torch, recv, nbytes * world, self.device, 'u1'))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return -1
        self._cb = (_capi.ALLREDUCE_FN(_ar), _capi.ALLGATHER_FN(_ag) if allgather is not None else None)   # kept alive with the engine
        self._ck(self.lib.fnn_dp_init_custom(self.h, int(rank), int(world), C.cast(self._cb[0], C.c_void_p),
                                             C.cast(self._cb[1], C.c_void_p) if self._cb[1] is not None else None, None,
                                             _capi.FNN_DP_SPARSE_EXCHANGE if sparse == 'exchange' else _capi.FNN_DP_SPARSE_LOCAL))
        self.dp_world = int(world)

    def dp_shutdown(self):
        self._ck(self.lib.fnn_dp_shutdown(self.h))
        self.dp_world = 1

    def dp_set_payload(self, payload):
        """'slabs' (2 MB between launch 2 and 3, three launches) or 'bucket' (0.5 MB after launch 3, a fourth launch updates)."""
        self._ck(self.lib.fnn_dp_set_payload(self.h, {'slabs': _capi.FNN_DP_PAYLOAD_SLABS, 'bucket': _capi.FNN_DP_PAYLOAD_BUCKET}[payload]))

    def dp_p2p_export(self, same_process=False):
        """This rank's 64-byte handle of its exchange region (fnn_dp_p2p_export); hand all ranks' handles to dp_p2p_attach."""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.fnn_dp_p2p_export(self.h, buf, 1 if same_process else 0))
        return buf.raw

    def dp_p2p_attach(self, handles, same_process=False):
        """handles: the `world` 64-byte handles in rank order; afterwards the dense collective is the one-shot peer-pointer
        all-reduce inside the update launch (FNN_DP_COLLECTIVE_P2P)."""
        blob = b''.join(handles)
        assert len(blob) == 64 * self.dp_world
        self._ck(self.lib.fnn_dp_p2p_attach(self.h, C.c_char_p(blob), 1 if same_process else 0))
        self._ck(self.lib.fnn_dp_set_collective(self.h, _capi.FNN_DP_COLLECTIVE_P2P))

    def dp_set_collective(self, name):
        self._ck(self.lib.fnn_dp_set_collective(self.h, _capi.FNN_DP_COLLECTIVE_P2P if name == 'p2p' else _capi.FNN_DP_COLLECTIVE_CALLBACK))

    def dp_p2p_max_wait_us(self):
        us = C.c_double(0.0)
        self._ck(self.lib.fnn_dp_p2p_max_wait_us(self.h, C.byref(us)))
        return us.value

    def dp_config(self):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self.lib.fnn_dp_get_config(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {'payload': ('slabs', 'bucket')[a.value], 'collective': ('callback', 'p2p')[b.value],
                'region': ('none', 'uncached', 'fine-grained', 'plain')[c.value]}

    def step_scatter(self):
        """Enqueue the sparse-row half of a begun step (overlaps an async all-reduce of the bucket)."""
        self._ck(self.lib.fnn_step_scatter(self.h))

    def sparse_grad(self, B):
        """gx' [B, row_floats] float32 of the begun step (slot layout: column 16 f + l), a view of the
        library's buffer -- what the exact data-parallel mode all-gathers."""
        ptr, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.fnn_sparse_grad(self.h, C.byref(ptr), C.byref(n)))
        return _tensor_from_ptr(self._torch, ptr.value, B * n.value, self.device).view(B, n.value)

    def step_scatter_global(self, ids_g, gxp_g):
        """Exact data-parallel mode: sparse-row SGD of the all-gathered global batch (ids_g [B_g, F]
        int32, -1 = padding; gxp_g [B_g, row_floats] float32), instead of step_scatter()."""
        torch = self._torch
        assert ids_g.is_cuda and ids_g.dtype == torch.int32 and ids_g.is_contiguous()
        assert gxp_g.is_cuda and gxp_g.dtype == torch.float32 and gxp_g.is_contiguous() and gxp_g.shape[0] == ids_g.shape[0]
        self._ck(self.lib.fnn_step_scatter_global(self.h, ids_g.data_ptr(), gxp_g.data_ptr(), ids_g.shape[0]))
        self._keep_g = (ids_g, gxp_g)

    def step_end(self, want_loss=False):
        loss = C.c_float(0.0)
        self._ck(self.lib.fnn_step_end(self.h, C.byref(loss) if want_loss else None))
        self._leave()
        return float(loss.value) if want_loss else None

    def last_loss(self):
        loss = C.c_float(0.0)
        self._ck(self.lib.fnn_last_loss(self.h, C.byref(loss)))
        return float(loss.value)

    def predict(self, ids):
        """A4': p [B] float32 tensor (python/FNN_wnzh.py:183)."""
        torch = self._torch
        ids_t = self._dev(ids, torch.int32)
        n = ids_t.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        self._enter()
        for lo in range(0, n, self.max_batch):
            hi = min(n, lo + self.max_batch)
            self._ck(self.lib.fnn_predict(self.h, ids_t[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr(),
                                          _capi.FNN_MEM_DEVICE))
        self._leave()
        return out

    def evaluate(self, ids, y, want_p=False):
        """A10 (python/FNN_wnzh.py:193-221): predict every example and compute AUC / RMSE / logloss
        on the device (fnn_eval).  Returns {'auc', 'rmse', 'logloss'[, 'p']}.  One class only ->
        FNNError(FNN_ERR_RANGE), as roc_auc_score raises."""
        torch = self._torch
        ids_t, y_t = self._dev(ids, torch.int32), self._dev(y, torch.int32)
        n = ids_t.shape[0]
        p = torch.empty(n, dtype=torch.float32, device=self.device) if want_p else None
        auc, rmse, ll = C.c_double(), C.c_double(), C.c_double()
        self._enter()
        self._ck(self.lib.fnn_eval(self.h, ids_t.data_ptr(), y_t.data_ptr(), n, _capi.FNN_MEM_DEVICE, C.byref(auc),
                                   C.byref(rmse), C.byref(ll), p.data_ptr() if want_p else None))
        self._leave()
        out = {'auc': auc.value, 'rmse': rmse.value, 'logloss': ll.value}
        if want_p:
            out['p'] = p
        return out

    # ------------------------------------------------------------------ profiling hook
    def prof_enable(self, on=True):
        self._ck(self.lib.fnn_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._ck(self.lib.fnn_prof_reset(self.h))

    def prof_get(self, which):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.lib.fnn_prof_get(self.h, which.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def _tensor_from_ptr(torch, ptr, n, device, kind='f4'):
    """Zero-copy view of `n` elements (float32, or 'u1' bytes) of device memory owned by the library."""

    class _Holder(object):
        pass

    hld = _Holder()
    hld.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': ('<' if kind != 'u1' else '|') + kind, 'data': (int(ptr), False),
                                    'version': 2}
    return torch.as_tensor(hld, device=device)
