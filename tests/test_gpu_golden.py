"""The HIP path against the committed fixtures of tests/golden/make_golden_more.py, through the C ABI: sparse CD-1 (online
and mini-batch), dense CD-1, one SNN fine-tune step, the inner-product stack's logits.  f32 mode; tolerances are relative
to the size of the parameter change (the kernels compute in f32, the fixtures are float64)."""
import ctypes as C
import os

import numpy as np
import pytest

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import _capi
from deep_ctr_amd.engine import FNNEngine
from deep_ctr_amd.ipnn import IPNNEngine

pytestmark = pytest.mark.gpu


def _dev(a, dtype):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype)).to(torch.device('cuda', 0)).contiguous()


def _close(got, ref, init, tol=2e-3):
    return np.abs(got - ref).max() <= tol * np.abs(ref - init).max() + 1e-7


def test_sparse_cd1_fixture(built, golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, 'rbm_sparse.npz'))
    lib = _capi.load()
    st = torch.cuda.current_stream(torch.device('cuda', 0)).cuda_stream
    N, S = g['vid'].shape
    H = g['W0'].shape[1]
    vid, vval, unif = _dev(g['vid'], np.int32), _dev(g['vval'], np.uint8), _dev(g['unif'], np.float32)
    for mode in ('online', 'mb4'):
        W, vb, hb = _dev(g['W0'], np.float32), _dev(g['vb0'], np.float32), _dev(g['hb0'], np.float32)
        ws = torch.zeros((S, H), dtype=torch.float32, device=W.device)
        err = C.c_double()
        if mode == 'online':
            rc = lib.rbm_sparse_epoch(W.data_ptr(), vb.data_ptr(), hb.data_ptr(), ws.data_ptr(), vid.data_ptr(), vval.data_ptr(),
                                      unif.data_ptr(), N, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st)
        else:
            dW, dvis = torch.zeros_like(W), torch.zeros_like(vb)
            rc = lib.rbm_sparse_batch(W.data_ptr(), dW.data_ptr(), vb.data_ptr(), dvis.data_ptr(), hb.data_ptr(), ws.data_ptr(),
                                      vid.data_ptr(), vval.data_ptr(), unif.data_ptr(), N, 4, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9,
                                      C.byref(err), st)
        assert rc == 0, lib.rbm_last_error()
        assert _close(W.cpu().numpy(), g['W_' + mode], g['W0']), mode
        assert _close(vb.cpu().numpy(), g['vb_' + mode], g['vb0']), mode
        assert _close(hb.cpu().numpy(), g['hb_' + mode], g['hb0']), mode
        np.testing.assert_allclose(ws.cpu().numpy(), g['ws_' + mode], rtol=2e-3, atol=1e-9)
        assert abs(err.value - float(g['err_' + mode])) <= 1e-5 * float(g['err_' + mode])


def test_dense_cd1_fixture(built, golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, 'rbm_dense.npz'))
    lib = _capi.load()
    st = torch.cuda.current_stream(torch.device('cuda', 0)).cuda_stream
    h = C.c_void_p()
    assert lib.rbm_dense_create(12, 8, 20, 0, 0, st, C.byref(h)) == 0, lib.rbm_last_error()
    W0, vb0, hb0 = (np.ascontiguousarray(g[k], np.float32) for k in ('W0', 'vb0', 'hb0'))
    assert lib.rbm_dense_set(h, W0.ctypes.data, vb0.ctypes.data, hb0.ctypes.data) == 0
    X, U = _dev(g['X'], np.float32), _dev(g['unif'], np.float32)
    for b in range(2):
        err = C.c_double()
        assert lib.rbm_dense_cd1(h, X.data_ptr(), 20, U.data_ptr() + b * 20 * 8 * 4, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err)) == 0
        assert abs(err.value - g['err'][b]) <= 1e-4 * g['err'][b]
    W, vb, hb = np.empty((12, 8), np.float32), np.empty(12, np.float32), np.empty(8, np.float32)
    assert lib.rbm_dense_get(h, W.ctypes.data, vb.ctypes.data, hb.ctypes.data) == 0
    lib.rbm_dense_destroy(h)
    assert _close(W, g['W'], g['W0']) and _close(vb, g['vb'], g['vb0']) and _close(hb, g['hb'], g['hb0'])


def test_snn_step_fixture(built, golden_dir):
    g = np.load(os.path.join(golden_dir, 'snn_step.npz'))
    p0 = {k: (g['p0_' + k] if g['p0_' + k].ndim else float(g['p0_' + k])) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')}
    eng = FNNEngine(16, 0, 20, 12, max_batch=256, precision='f32', lr=float(g['lr']), lambda1=float(g['lambda1']), lambda_fm=0.0,
                    reg_all=True, mode='bag', hidden0=200)
    eng.set_table(g['ww0'].astype(np.float32), np.zeros(g['ww0'].shape[0], np.int32), 0.0)
    eng.set_bag_bias(g['bb0'].astype(np.float32))
    eng.set_dense(p0)
    np.testing.assert_allclose(eng.gather(g['ids']).cpu().numpy(), g['x'], rtol=2e-6, atol=1e-7)
    out = eng.train_step(g['ids'], g['y'].astype(np.float32), g['r1'], g['r2'], want_p=True, want_gx=True)
    np.testing.assert_allclose(out['p'].cpu().numpy(), g['p_drop'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out['gx'].cpu().numpy(), g['gx'], rtol=2e-3, atol=2e-5 * np.abs(g['gx']).max() + 1e-9)
    assert abs(out['loss'] - float(g['loss'])) <= 2e-5 * max(1.0, abs(float(g['loss'])))
    rows = eng.get_table()[g['touched']]
    assert _close(rows, g['rows_after'], g['ww0'][g['touched']], 1e-3)
    assert _close(eng.get_bag_bias(), g['bb0_after'], g['bb0'], 1e-3)
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        assert _close(d[k], g['p1_' + k], g['p0_' + k], 1e-3), k
    eng.close()


def test_ip_l7_logits_fixture(built, golden_dir):
    g = np.load(os.path.join(golden_dir, 'ip_l7.npz'))
    hidden = [g['W%d' % i].shape[1] for i in range(7)]
    eng = IPNNEngine(16, 11, hidden, 'relu', max_batch=256, precision='f32', lr=0.01, keep_prob=1.0)
    eng.set_params(g['table'], float(g['b']), [g['W%d' % i] for i in range(8)], [g['bias%d' % i] for i in range(8)])
    pr = eng.predict(g['ids']).cpu().numpy()
    np.testing.assert_allclose(pr, 1.0 / (1.0 + np.exp(-g['logits'])), rtol=2e-4, atol=1e-6)
    eng.close()


def test_gather_equals_the_reference_layer_one_arrays(built, golden_dir):
    """A3 against the REFERENCE's own output (tests/golden/ref_run.npz: data_fm.DataFM.get_fxy_fm on 96 demo lines, run in the
    build container by make_golden_ref.py): DataFM -> ids -> fnn_gather must reproduce its x, a float32 copy of the parsed rows."""
    from deep_ctr_amd.data_fm import DataFM
    ref = np.load(os.path.join(golden_dir, 'ref_run.npz'))
    d = DataFM(os.path.join(golden_dir, 'demo', 'fm.model.txt'))
    rows, fo, w0 = d.table()
    eng = FNNEngine(len(d.name_field), d.k, 300, 100, max_batch=256, precision='f32')
    eng.set_table(rows, fo, w0)
    d.engine = eng
    ids = np.stack([d.feats_to_ids([int(v) for v in r if v >= 0]) for r in ref['fm_line_feats']])
    x = eng.gather(ids).cpu().numpy()
    assert np.array_equal(x, ref['fm_line_x'].astype(np.float32))
    # and through the reference-shaped loader call (linecache + fnn_gather)
    f, xb, yb = d.get_batch_data(os.path.join(golden_dir, 'demo', 'train.fm.txt'), 1, 96)
    assert np.array_equal(xb, ref['fm_line_x'].astype(np.float32)) and np.array_equal(yb, ref['fm_line_y'])
    assert f == [[int(v) for v in r if v >= 0] for r in ref['fm_line_feats']]
    eng.close()
