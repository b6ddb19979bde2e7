"""FNN_PREC_BF16X3 (include/fnn_hip.h): operands as bf16 pairs (hi + lo), three bf16 MFMAs per product.  The mode shares every
layout and kernel with the exact-f32 mode and differs in `mma` alone, so it is held to the SAME float64 oracle comparisons as the
f32 mode at a tolerance a few times wider (16 significant bits per operand against 24: observed errors are printed), on the
three-launch path, the layer-by-layer path, bag mode, multi-step runs, and to north_star's 1e-4 logloss / AUC bar on the demo set."""
import os

import numpy as np
import pytest

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from test_gpu_parity import (_check_step, _run_demo_epochs, make_engine, make_problem, make_snn_engine, make_snn_problem)

pytestmark = pytest.mark.gpu
TOL = 8.0             # x the f32 mode's tolerances of _check_step
TOL_TABLE = 40.0      # the table moves by lr * gx: its error is gx's (rtol 2e-3 x TOL), as an absolute 8e-6 on rows of ~0.05


@pytest.mark.parametrize("B,kw", [(1, {}), (64, {"dup_col": 6}), (257, {"dup_col": 0}), (1000, {"empty": [(0, 15)]}), (4096, {})])
def test_train_step_bf16x3_vs_oracle(built, B, kw):
    rows, fo, ids, y, p, r1, r2 = make_problem(B, seed=B, **kw)
    eng = make_engine(rows, fo, p, prec='bf16x3', lr=0.01, lam1=0.02, lamfm=0.1)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.02, 0.1, tol=TOL, tol_table=TOL_TABLE)
    eng.close()


def test_bf16x3_error_sits_between_f32_and_bf16(built):
    """The same step in the three precisions: the error of the predictions against float64 is ordered f32 < bf16x3 << bf16."""
    rows, fo, ids, y, p, r1, r2 = make_problem(1024, seed=5, dup_col=3)
    x = orc.gather(rows.astype(np.float64), ids, -3.0)
    _, _, _, p_drop, _ = orc.train_call({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}, x, y.astype(np.float64),
                                        r1.astype(np.float64), r2.astype(np.float64), 0.01, 0.0, 'tanh')
    err = {}
    for prec in ('f32', 'bf16x3', 'bf16'):
        eng = make_engine(rows, fo, p, prec=prec, lr=0.01)
        err[prec] = float(np.abs(eng.train_step(ids, y, r1, r2, want_p=True)['p'].cpu().numpy() - p_drop).max())
        eng.close()
    print("max |p - p_f64|: %r" % (err,))
    assert err['f32'] <= err['bf16x3'] * 1.5 + 1e-7 and err['bf16x3'] * 30 < err['bf16']


def test_bf16x3_layer_by_layer_path(built, monkeypatch):
    rows, fo, ids, y, p, r1, r2 = make_problem(300, seed=17, dup_col=6, empty=[(1, 2)])
    outs = []
    for nofuse in ('0', '1'):
        monkeypatch.setenv('FNN_NO_FUSE', nofuse)
        eng = make_engine(rows, fo, p, prec='bf16x3', lr=0.01, lam1=0.02)
        _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.02, 0.1, tol=TOL, tol_table=TOL_TABLE)
        outs.append((eng.get_table(), eng.predict(ids).cpu().numpy()))
        eng.close()
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-3, atol=1e-6)


def test_bf16x3_multi_step_sequence(built):
    """Six consecutive steps (fresh masks per step, the reference's stream): state carried on the device -- table, masters and the
    bf16-pair shadows the third launch refreshes -- tracks the oracle; same run as test_multi_step_sequence_f32."""
    from test_gpu_parity import H1, H2
    rows, fo, ids, y, p, r1, r2 = make_problem(600, seed=21)
    eng = make_engine(rows, fo, p, prec='bf16x3', lr=0.002)
    rows64 = rows.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    for j in range(6):
        sl = slice(j * 100, (j + 1) * 100)
        m1, m2 = ms.next()
        out = eng.train_step(ids[sl], y[sl], m1.astype(np.uint8), m2.astype(np.uint8))
        ref = orc.train_step(p64, rows64, -3.0, ids[sl], y[sl].astype(np.float64), m1, m2, 0.002, 0.0, 0.1)
        assert abs(out['loss'] - ref['loss']) <= 2e-4 * abs(ref['loss'])
    np.testing.assert_allclose(eng.get_table(), rows64, rtol=4e-4, atol=4e-6)
    pr = eng.predict(ids).cpu().numpy()
    np.testing.assert_allclose(pr, orc.predict(p64, orc.gather(rows64, ids, -3.0)), rtol=8e-4, atol=4e-6)
    eng.close()


@pytest.mark.parametrize("B,h0", [(37, 200), (1000, 200), (300, 300)])
def test_snn_step_bf16x3_vs_oracle(built, B, h0):
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(B, seed=B, h0=h0, dup_col=6)
    eng = make_snn_engine(ww0, bb0, p, prec='bf16x3', h0=h0)
    ww64, bb64 = ww0.astype(np.float64), bb0.astype(np.float64)
    out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.snn_train_step(p64, ww64, bb64, ids, y.astype(np.float64), r1.astype(float), r2.astype(float), 0.01, 0.001)
    np.testing.assert_allclose(out['p'].cpu().numpy(), ref['p_drop'], rtol=2e-4 * TOL, atol=1e-6 * TOL)
    gs = np.abs(ref['gx']).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), ref['gx'], rtol=2e-3 * TOL, atol=2e-5 * gs * TOL + 1e-9)
    upd = np.abs(ww64 - ww0).max() + 1e-12
    assert np.abs(eng.get_table() - ww64).max() <= 1e-3 * TOL * upd + 2e-7
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        scale = np.abs(p64[k] - p[k]).max() + 1e-12
        assert np.abs(d[k] - p64[k]).max() <= 1e-3 * TOL * scale + 1e-7, k
    eng.close()


def test_demo_epochs_bf16x3_logloss_auc_within_1e4(built, golden_dir):
    """north_star's bar -- logloss and AUC on the demo set within 1e-4 absolute of the float64 restatement, 3 epochs -- in the
    bf16-pair mode (the plain bf16 mode misses it: tests/test_gpu_parity.py::test_demo_epochs_bf16_tracks_oracle)."""
    g = np.load(os.path.join(golden_dir, 'epoch.npz'))
    h = _run_demo_epochs(golden_dir, 'bf16x3')
    d = {'train_auc': float(np.abs(h[:, 0] - g['train_auc']).max()), 'train_logloss': float(np.abs(h[:, 1] - g['train_logloss']).max()),
         'test_auc': float(np.abs(h[:, 2] - g['test_auc']).max()), 'test_logloss': float(np.abs(h[:, 3] - g['test_logloss']).max())}
    print("bf16x3 demo deltas vs the float64 oracle: %r" % (d,))
    try:
        import json
        out = os.path.join(os.path.dirname(golden_dir), '..', 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        json.dump(d, open(os.path.join(out, 'bf16x3_demo_deltas.json'), 'w'))
    except OSError:
        pass
    assert max(d.values()) <= 1e-4


def test_bf16x3_data_parallel_and_eval(built):
    """The pair precision under the native data-parallel step (RCCL, world size 1: bitwise the plain step) and in the device-side
    evaluation pass (metrics equal sklearn's on the engine's own predictions)."""
    from sklearn.metrics import log_loss, roc_auc_score
    from deep_ctr_amd.engine import FNNEngine
    rows, fo, ids, y, p, r1, r2 = make_problem(2 * 700, seed=51, dup_col=6)
    kw = dict(prec='bf16x3', lr=0.01, lam1=0.02, lamfm=0.1)
    plain, dp = make_engine(rows, fo, p, **kw), make_engine(rows, fo, p, **kw)
    dp.dp_init(0, 1, FNNEngine.dp_unique_id())
    for s in range(2):
        sl = slice(s * 700, (s + 1) * 700)
        a = plain.train_step(ids[sl], y[sl], r1, r2)
        b = dp.train_step(ids[sl], y[sl], r1, r2, b_size=700)
        assert a['loss'] == b['loss']
    da, db = plain.get_dense(), dp.get_dense()
    assert all(np.array_equal(da[k], db[k]) for k in da) and np.array_equal(plain.get_table(), dp.get_table())
    dp.dp_shutdown(); dp.close()
    yy = (np.random.RandomState(6).uniform(size=len(ids)) < 0.3).astype(np.int32)
    m = plain.evaluate(ids, yy, want_p=True)
    pp = m['p'].cpu().numpy().astype(np.float64)
    assert abs(m['auc'] - roc_auc_score(yy, pp)) < 1e-12 and abs(m['logloss'] - log_loss(yy, pp, labels=[0, 1])) < 1e-12
    plain.close()
