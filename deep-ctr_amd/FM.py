"""Factorisation-machine pre-training on MI355X: the arithmetic of the reference's TensorFlow class
`FM` (python/FM.py) behind include/fm_hip.h, with the class's constructor signature, `dump` keys
(`W`, `V`, `b`) and the roles of its graph outputs (`train_step` = ptmzr + loss + train_preds,
`forward` = test_preds).  `write_fm_model` closes the loop the reference leaves open: it writes the
`fm.model.txt` text format that python/FNN_wnzh.py:62-84 parses, so that FM -> FNN runs end to end.
Random init uses NumPy RandomState(seed) streams (TensorFlow's cannot be reproduced here)."""
import ctypes as C
import pickle

import numpy as np

from . import _capi
from .engine import FNNError


class FM(object):
    def __init__(self, batch_size, _rch_argv, _init_argv, _ptmzr_argv, _reg_argv, mode='train', eval_size=0, device=0):
        import torch
        if not torch.cuda.is_available():
            raise FNNError(_capi.FNN_ERR_HIP, "no HIP device visible to PyTorch-ROCm; no CPU fallback")
        X_dim, X_feas, rank = _rch_argv                              # python/FM.py:7
        if _ptmzr_argv[0] != 'sgd':
            raise NotImplementedError("only plain SGD is built (the reference's Adam/FTRL: python/tf_util.py:15-29)")
        self._torch, self.lib = torch, _capi.load()
        self.device = torch.device('cuda', device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.X_dim, self.X_feas, self.rank = X_dim, X_feas, rank
        self.lr = float(_ptmzr_argv[1])
        self.reduce_mean = 0 if _ptmzr_argv[-1] == 'sum' else 1      # :38-41
        self.lam = float(_reg_argv[0]) if mode == 'train' else 0.0
        self.log = 'input dim: %d, features: %d, rank: %d, ' % (X_dim, X_feas, rank)
        h = C.c_void_p()
        rc = self.lib.fm_create(X_feas, rank + 1, min(4096, max(batch_size, eval_size, 1)), device,
                                C.c_void_p(self.stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise FNNError(rc, (self.lib.fm_last_error(None) or b'').decode())
        self.h = h
        lo, hi, seeds, path = _init_argv[1], _init_argv[2], _init_argv[3], _init_argv[-1]
        var_map = pickle.load(open(path, 'rb')) if path else {}     # python/tf_util.py:41-82
        W = var_map['W'] if 'W' in var_map else np.random.RandomState(seeds[0]).uniform(lo, hi, (X_dim, 1))
        V = var_map['V'] if 'V' in var_map else np.random.RandomState(seeds[1 % len(seeds)]).uniform(lo, hi, (X_dim, rank))
        b = float(np.asarray(var_map.get('b', 0.0)).ravel()[0])
        self.set_params(np.concatenate([np.asarray(W).reshape(X_dim, 1), np.asarray(V)], axis=1), b)

    def _ck(self, rc):
        if rc != 0:
            raise FNNError(rc, (self.lib.fm_last_error(self.h) or b'').decode())

    def close(self):
        if getattr(self, 'h', None):
            self.lib.fm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, rows, b):
        t = np.ascontiguousarray(rows, dtype=np.float32)
        self._ck(self.lib.fm_set_table(self.h, t.ctypes.data, t.shape[0]))
        self._ck(self.lib.fm_set_b(self.h, float(b)))

    def get_params(self):
        rows = np.empty((self.X_dim, self.rank + 1), np.float32)
        self._ck(self.lib.fm_get_table(self.h, rows.ctypes.data))
        b = C.c_float()
        self._ck(self.lib.fm_get_b(self.h, C.byref(b)))
        return rows, float(b.value)

    def _dev(self, a, dtype):
        torch = self._torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def train_step(self, ids, y, want_p=False, want_loss=True):
        """ids [B, X_feas] int32 (-1 = absent), y [B].  Returns {'loss', 'p'} (python/ipinyou.py:171)."""
        torch = self._torch
        ids_t, y_t = self._dev(ids, torch.int32), self._dev(y, torch.float32)
        B = ids_t.shape[0]
        p = torch.empty(B, dtype=torch.float32, device=self.device) if want_p else None
        loss = C.c_float()
        if B > 4096:                # one call = ONE optimiser step (python/ipinyou.py:171); splitting it would change the update
            raise FNNError(_capi.FNN_ERR_ARG, "FM.train_step: batch %d > 4096 (fm_create's largest step)" % B)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ck(self.lib.fm_train_step(self.h, ids_t.data_ptr(), y_t.data_ptr(), B, self.lr, self.lam, self.reduce_mean,
                                        p.data_ptr() if want_p else None, C.byref(loss) if want_loss else None))
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self._keep = (ids_t, y_t)
        return {'loss': float(loss.value) if want_loss else None, 'p': p}

    def forward(self, ids):
        torch = self._torch
        ids_t = self._dev(ids, torch.int32)
        out = torch.empty(ids_t.shape[0], dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for lo in range(0, ids_t.shape[0], 4096):
            hi = min(ids_t.shape[0], lo + 4096)
            self._ck(self.lib.fm_predict(self.h, ids_t[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr()))
        self._ck(self.lib.fm_sync(self.h))
        return out

    def dump(self, model_path):                                      # python/FM.py:66-69
        rows, b = self.get_params()
        pickle.dump({'W': rows[:, :1], 'V': rows[:, 1:], 'b': np.array([b], np.float32)}, open(model_path, 'wb'))
        print('model dumped at %s' % model_path)

    def write_fm_model(self, path, field_of_row, field_names, feat_ids=None):
        """`fm.model.txt` as python/FNN_wnzh.py:68-84 reads it: `w_0 feat_num rank`, then per feature
        `feat w v_1..v_rank <fieldname>:<feat>`.  repr() of the float32 values keeps them exact."""
        rows, b = self.get_params()
        feat_ids = np.arange(len(rows)) if feat_ids is None else np.asarray(feat_ids)
        with open(path, 'w') as f:
            f.write('%r %d %d\n' % (float(b), len(rows), self.rank))
            for i in range(len(rows)):
                f.write('%d %s %s:%d\n' % (feat_ids[i], ' '.join(repr(float(v)) for v in rows[i]),
                                           field_names[int(field_of_row[i])], feat_ids[i]))
