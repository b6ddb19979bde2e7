"""Inner-product FNN family on MI355X: the arithmetic of the reference's TensorFlow classes
`FNN_IP_L3` / `FNN_IP_L5` / `FNN_IP_L7` (python/FNN_IP_L7.py:5-133) behind include/ipnn_hip.h.
`IPNNEngine` is the PyTorch-ROCm plumbing; the three class names of the reference are kept as
constructors with its `_rch_argv` layout (X_dim, X_feas, rank, h1..hN, act_func), its `forward`
role (`train_step` / `predict`) and its `dump` keys (`W`, `V`, `b`, `h{i}_w`, `h{i}_b`).
Categorical fields only; optimiser 'sgd', 'adam' or 'ftrl' (python/tf_util.py:15-29)."""
import ctypes as C
import pickle

import numpy as np

from . import _capi
from .engine import FNNError


class IPNNEngine(object):
    def __init__(self, n_fields, k, hidden, act='relu', max_batch=4096, precision='bf16', lr=1e-4, keep_prob=0.5, device=0,
                 pairs=True, optimizer='sgd', adam_eps=1e-8, adam_betas=(0.9, 0.999), reduce='sum'):
        import torch
        if not torch.cuda.is_available():
            raise FNNError(_capi.FNN_ERR_HIP, "no HIP device visible to PyTorch-ROCm; no CPU fallback")
        self._torch, self.lib = torch, _capi.load()
        self.device = torch.device('cuda', device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.F, self.K, self.hidden = n_fields, k, list(hidden)
        self.d = [n_fields * k + (n_fields * (n_fields - 1) // 2 if pairs else 0) + 1] + self.hidden + [1]
        hid = (C.c_int32 * 8)(*(self.hidden + [0] * (8 - len(self.hidden))))
        cfg = _capi.ipnn_cfg(n_fields, k, len(self.hidden), hid, _capi.IPNN_ACTS[act], 1 if pairs else 0, max_batch,
                             1 if precision == 'bf16' else 0, lr, keep_prob, {'sgd': 0, 'adam': 1, 'ftrl': 2}[optimizer], adam_betas[0],
                             adam_betas[1], adam_eps, device, C.c_void_p(self.stream.cuda_stream))
        h = C.c_void_p()
        rc = self.lib.ipnn_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise FNNError(rc, (self.lib.ipnn_last_error(None) or b'').decode())
        self.h = h
        self.reduce = reduce                      # 'sum' or 'mean' (python/FNN_IP_L7.py:83-86: anything but 'sum' is the mean)
        self._ck(self.lib.ipnn_set_loss_mean(self.h, 0 if reduce == 'sum' else 1))

    def _ck(self, rc):
        if rc != 0:
            raise FNNError(rc, (self.lib.ipnn_last_error(self.h) or b'').decode())

    def close(self):
        if getattr(self, 'h', None):
            self.lib.ipnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, table, b, Ws, biases):
        t = np.ascontiguousarray(table, dtype=np.float32)
        self._ck(self.lib.ipnn_set_table(self.h, t.ctypes.data, t.shape[0]))
        self._ck(self.lib.ipnn_set_b(self.h, float(b)))
        for i, (W, bias) in enumerate(zip(Ws, biases), start=1):
            W = np.ascontiguousarray(W, dtype=np.float32).reshape(self.d[i - 1], self.d[i])
            bias = np.ascontiguousarray(np.atleast_1d(bias), dtype=np.float32)
            self._ck(self.lib.ipnn_set_layer(self.h, i, W.ctypes.data, bias.ctypes.data))

    def get_params(self):
        Ws, bs = [], []
        for i in range(1, len(self.d)):
            W = np.empty((self.d[i - 1], self.d[i]), np.float32)
            b = np.empty(self.d[i], np.float32)
            self._ck(self.lib.ipnn_get_layer(self.h, i, W.ctypes.data, b.ctypes.data))
            Ws.append(W); bs.append(b)
        bb = C.c_float()
        self._ck(self.lib.ipnn_get_b(self.h, C.byref(bb)))
        return float(bb.value), Ws, bs

    def get_rows(self, row_ids):
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        out = np.empty((len(ids), self.K), np.float32)
        self._ck(self.lib.ipnn_get_rows(self.h, ids.ctypes.data, len(ids), out.ctypes.data))
        return out

    def _dev(self, a, dtype):
        torch = self._torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def train_step(self, ids, y, masks=None, want_logits=False, want_loss=True):
        """masks: list of len(hidden)+1 uint8 arrays [B, d_t] (keep-masks for z1 and every hidden layer)."""
        torch = self._torch
        ids_t, y_t = self._dev(ids, torch.int32), self._dev(y, torch.float32)
        B = ids_t.shape[0]
        mts, marr = None, None
        if masks is not None:
            mts = [self._dev(m, torch.uint8) for m in masks]
            assert len(mts) == len(self.hidden) + 1 and all(m.shape == (B, self.d[t]) for t, m in enumerate(mts))
            marr = (C.c_void_p * len(mts))(*[m.data_ptr() for m in mts])
        logits = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        loss = C.c_float()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ck(self.lib.ipnn_train_step(self.h, ids_t.data_ptr(), y_t.data_ptr(), B, marr,
                                          logits.data_ptr() if want_logits else None, C.byref(loss) if want_loss else None))
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self._keep = (ids_t, y_t, mts)
        scale = 1.0 if self.reduce == 'sum' else 1.0 / B     # the library returns the sum of the per-example losses
        return {'loss': float(loss.value) * scale if want_loss else None, 'logits': logits}

    def predict(self, ids):
        torch = self._torch
        ids_t = self._dev(ids, torch.int32)
        out = torch.empty(ids_t.shape[0], dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for lo in range(0, ids_t.shape[0], 4096):
            hi = min(ids_t.shape[0], lo + 4096)
            self._ck(self.lib.ipnn_predict(self.h, ids_t[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr()))
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out

    def evaluate(self, ids, y):
        """python/baseline.py:382-437: predictions + AUC / RMSE / logloss on the device (ipnn_eval)."""
        torch = self._torch
        ids_t, y_t = self._dev(ids, torch.int32), self._dev(y, torch.int32)
        auc, rmse, ll = C.c_double(), C.c_double(), C.c_double()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ck(self.lib.ipnn_eval(self.h, ids_t.data_ptr(), y_t.data_ptr(), ids_t.shape[0], C.byref(auc), C.byref(rmse), C.byref(ll)))
        return {'auc': auc.value, 'rmse': rmse.value, 'logloss': ll.value}

    def sync(self):
        self._ck(self.lib.ipnn_sync(self.h))


class _IPFamily(object):
    """Constructor signature of python/FNN_IP_L7.py:5: (cat_sizes, offsets, batch_size, _rch_argv,
    _init_argv, _ptmzr_argv, _reg_argv, mode, eval_size).  _rch_argv = [X_dim, X_feas, rank,
    h1.., act_func]; _init_argv = ['uniform', lo, hi, seeds, path] (python/tf_util.py:41-82: a
    pickle path seeds any subset of the variables); _ptmzr_argv = ['sgd', lr, ...]."""
    N_HIDDEN = 0
    PAIRS = True

    def __init__(self, cat_sizes, offsets, batch_size, _rch_argv, _init_argv, _ptmzr_argv, _reg_argv, mode='train',
                 eval_size=0, precision='bf16'):
        X_dim, X_feas, rank = _rch_argv[:3]
        hidden, act = list(_rch_argv[3:-1]), _rch_argv[-1]
        assert len(hidden) == self.N_HIDDEN
        if _ptmzr_argv[0] not in ('sgd', 'adam', 'ftrl'):            # python/tf_util.py:15-29 (anything else: plain gradient descent there)
            raise NotImplementedError("optimizer %r: sgd, adam and ftrl are built" % (_ptmzr_argv[0],))
        self.keep = _reg_argv[0] if mode == 'train' else 1.0
        self.eng = IPNNEngine(X_feas, rank + 1, hidden, act, max_batch=max(batch_size, eval_size, 1), precision=precision,
                              lr=_ptmzr_argv[1], keep_prob=self.keep, pairs=self.PAIRS, optimizer=_ptmzr_argv[0],
                              adam_eps=_ptmzr_argv[2] if _ptmzr_argv[0] == 'adam' else 1e-8,
                              reduce='sum' if _ptmzr_argv[-1] == 'sum' else 'mean')     # python/FNN_IP_L7.py:83-86
        lo, hi, seeds, path = _init_argv[1], _init_argv[2], _init_argv[3], _init_argv[-1]
        var_map = pickle.load(open(path, 'rb')) if path else {}
        d = self.eng.d
        rs = [np.random.RandomState(s) for s in seeds]
        j = 0

        def rnd(shape):
            nonlocal j
            v = rs[j % len(rs)].uniform(lo, hi, size=shape); j += 1
            return v
        W = var_map['W'] if 'W' in var_map else rnd((X_dim, 1))
        V = var_map['V'] if 'V' in var_map else rnd((X_dim, rank))
        b = float(np.asarray(var_map.get('b', 0.0)).ravel()[0])
        Ws, bs = [], []
        for i in range(1, len(d)):
            Ws.append(var_map['h%d_w' % i] if 'h%d_w' % i in var_map else rnd((d[i - 1], d[i])))
            bs.append(var_map['h%d_b' % i] if 'h%d_b' % i in var_map else np.zeros(d[i]))
        self.eng.set_params(np.concatenate([W, V], axis=1), b, Ws, bs)
        self.rank, self.X_dim = rank, X_dim

    def train_step(self, ids, y, masks=None):
        return self.eng.train_step(ids, y, masks)

    def forward(self, ids, v_wts=None):
        """Predictions for categorical ids [N, X_feas].  The reference's `forward(N, M, v_wts, c_ids, c_wts, ...)`
        (python/FNN_IP_L7.py:102-106) also takes Criteo's 13 numeric fields, each a value times a row (:103): not built --
        the path here is the iPinYou shape, one categorical id per field, every weight 1."""
        if v_wts is not None:
            raise NotImplementedError("numeric value-weighted fields (python/FNN_IP_L7.py:103) are not built: categorical ids only")
        return self.eng.predict(ids)

    def dump(self, model_path):
        """python/FNN_IP_L7.py:135-143: var_map pickle (touched rows only are current on the host:
        the full table is read back row by row)."""
        b, Ws, bs = self.eng.get_params()
        rows = self.eng.get_rows(np.arange(self.X_dim))
        var_map = {'W': rows[:, :1], 'V': rows[:, 1:], 'b': np.array([b], np.float32)}
        for i, (W, bb) in enumerate(zip(Ws, bs), start=1):
            var_map['h%d_w' % i] = W
            var_map['h%d_b' % i] = bb
        pickle.dump(var_map, open(model_path, 'wb'))


class FNN_IP_L3(_IPFamily):
    N_HIDDEN = 3


class FNN_IP_L5(_IPFamily):
    N_HIDDEN = 5


class FNN_IP_L7(_IPFamily):
    N_HIDDEN = 7


class FNN(_IPFamily):
    """The reference's plain TensorFlow `FNN` class (python/FNN.py:5-101): z1 = [e_0 .. e_{F-1} | b], two
    hidden layers, activation and inverted dropout before every matmul; var_map keys W, V, b, h1_w .. h3_b.
    (Its Criteo numeric fields -- a value times a row, :78 -- are not built: categorical fields only.)"""
    N_HIDDEN = 2
    PAIRS = False
