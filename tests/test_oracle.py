"""CPU tests of the oracle itself: known-answer anchors, finite differences, closed forms, the C
port against the NumPy restatement, and the committed golden vectors.  (PARITY UNPINNED by the
reference: it holds no fixtures; see oracle/fnn_oracle.py.)"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import synth

F, K, H1, H2 = 16, 11, 300, 100
XDIM = 1 + F * K


def small_problem(B=12, h1=7, h2=5, seed=0, n_rows=60, f=4, k=3):
    rng = np.random.RandomState(seed)
    sizes = [n_rows // f] * f
    rows = rng.standard_normal((sum(sizes), k)) * 0.3
    ids = synth.zipf_ids(B, sizes, 1.1, seed + 1)
    y = (rng.uniform(size=B) < 0.4).astype(np.float64)
    xdim = 1 + f * k
    p = {'w1': rng.standard_normal((xdim, h1)) * 0.4, 'b1': rng.standard_normal(h1) * 0.1,
         'w2': rng.standard_normal((h1, h2)) * 0.4, 'b2': rng.standard_normal(h2) * 0.1,
         'w3': rng.standard_normal(h2) * 0.4, 'b3': 0.2}
    r1 = (rng.uniform(size=h1) < 0.6).astype(np.float64)
    r2 = (rng.uniform(size=h2) < 0.6).astype(np.float64)
    return rows, ids, y, p, r1, r2


def test_rng_known_answers():
    """Anchors shared with the reference's RNG use: legacy seed 1234 stream, the
    FNN w1 bound, and the three per-op seeds of RandomStreams(234)."""
    np.random.seed(1234)
    np.testing.assert_allclose(np.random.uniform(size=4),
                               [0.19151945, 0.62210877, 0.43772774, 0.78535858], atol=1e-8)
    p = orc.init_fnn_weights(XDIM, H1, H2, 'tanh', seed=1234)
    bound = np.sqrt(6. / (XDIM + H1))
    assert abs(bound - 0.11215443081840885) < 1e-15
    np.testing.assert_allclose(p['w1'][0, :3] / 4, [-0.06919492, 0.02739008, -0.01396822], atol=1e-8)
    assert np.all(p['w3'] == 0) and p['b3'] == 0.0 and np.all(p['b1'] == 0) and np.all(p['b2'] == 0)
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    assert ms.seeds == [133003720, 999614367, 18391364]
    r1, r2 = ms.next()
    assert r1.shape == (H1,) and r2.shape == (H2,) and set(np.unique(r1)) <= {0.0, 1.0}


def test_gather_matches_per_example_restatement():
    rows, ids, _, _, _, _ = small_problem()
    ids[2, 1] = -1
    k = rows.shape[1]
    fo = synth.field_of_row([15] * 4)
    fw = {int(r): list(rows[r]) for r in range(rows.shape[0])}
    ff = {int(r): int(fo[r]) for r in range(rows.shape[0])}
    x = orc.gather(rows, ids, -1.5)
    # per-example form takes the global 16-field xdim; compare on a 16-field problem instead
    sizes = synth.field_sizes_tiny(200)
    rows16 = np.random.RandomState(3).standard_normal((200, K))
    ids16 = synth.zipf_ids(9, sizes, 1.1, 4)
    fo16 = synth.field_of_row(sizes)
    fw = {int(r): list(rows16[r]) for r in range(200)}
    ff = {int(r): int(fo16[r]) for r in range(200)}
    x16 = orc.gather(rows16, ids16, -1.5)
    for t in range(9):
        xt = orc.feats_to_layer_one_array([int(r) for r in ids16[t]], -1.5, K, XDIM, fw, ff)
        assert np.array_equal(xt, x16[t])
    assert x[2, 1 + 1 * k:1 + 2 * k].sum() == 0 and x[0, 0] == -1.5


@pytest.mark.parametrize("acti", ['tanh', 'sigmoid', 'linear'])
def test_gradients_finite_difference(acti):
    rows, ids, y, p, r1, r2 = small_problem()
    x = orc.gather(rows, ids, -0.7)
    lam = 0.05
    loss, _, g = orc.loss_and_grads(p, x, y, r1, r2, lam, acti)

    def cost(pp, xx):
        _, _, _, _, pd = orc.forward_train(pp, xx, r1, r2, acti)
        xe = -y * np.log(pd) - (1 - y) * np.log(1 - pd)
        return xe.sum() + lam * ((pp['w3'] ** 2).sum() + pp['b3'] ** 2)

    eps = 1e-6
    rng = np.random.RandomState(5)
    for name in ('w1', 'b1', 'w2', 'b2', 'w3'):
        for _ in range(6):
            idx = tuple(rng.randint(s) for s in p[name].shape)
            pp = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
            pp[name][idx] += eps
            up = cost(pp, x)
            pp[name][idx] -= 2 * eps
            dn = cost(pp, x)
            assert abs((up - dn) / (2 * eps) - g[name][idx]) < 1e-5 * max(1, abs(g[name][idx]))
    pp = dict(p); pp['b3'] = p['b3'] + eps; up = cost(pp, x)
    pp['b3'] = p['b3'] - eps; dn = cost(pp, x)
    assert abs((up - dn) / (2 * eps) - g['b3']) < 1e-5
    for _ in range(8):
        t, i = rng.randint(x.shape[0]), rng.randint(x.shape[1])
        xx = x.copy(); xx[t, i] += eps; up = cost(p, xx)
        xx[t, i] -= 2 * eps; dn = cost(p, xx)
        assert abs((up - dn) / (2 * eps) - g['x'][t, i]) < 1e-5


def test_train_call_returns_pre_update_and_applies_sgd():
    rows, ids, y, p, r1, r2 = small_problem()
    x = orc.gather(rows, ids, -0.7)
    p0 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    gx, pre, loss, _, g = orc.train_call(p, x, y, r1, r2, lr=0.01)
    for k in p0:
        assert np.array_equal(np.asarray(pre[k]), np.asarray(p0[k]))
        np.testing.assert_allclose(np.asarray(p[k]), np.asarray(p0[k]) - 0.01 * np.asarray(g[k]), rtol=0, atol=1e-15)


def test_scatter_sequential_equals_closed_form():
    rows, ids, y, p, r1, r2 = small_problem(B=40)
    ids[:, 0] = ids[0, 0]                       # one row hit by every example
    ids[7, 2] = -1
    gx = np.random.RandomState(6).standard_normal((40, 1 + 4 * 3))
    a = orc.scatter_sgd(rows.copy(), ids, gx, 0.05, 0.3, b_size=40)
    b = orc.scatter_sgd_closed_form(rows.copy(), ids, gx, 0.05, 0.3, b_size=40)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
    untouched = np.setdiff1d(np.arange(rows.shape[0]), np.unique(ids[ids >= 0]))
    assert np.array_equal(a[untouched], rows[untouched])          # decay only on touched rows
    c = orc.scatter_sgd(rows.copy(), ids, gx, 0.05, 0.3, b_size=400)   # global batch under DP
    assert not np.allclose(a, c)


def test_metrics_match_sklearn():
    from sklearn.metrics import log_loss, mean_squared_error, roc_auc_score
    rng = np.random.RandomState(1)
    y = (rng.uniform(size=500) < 0.3).astype(int)
    p = np.round(rng.uniform(size=500), 2)        # ties on purpose
    assert abs(orc.roc_auc(y, p) - roc_auc_score(y, p)) < 1e-12
    assert abs(orc.rmse(y, p) - np.sqrt(mean_squared_error(y, p))) < 1e-12
    p = np.clip(p, 0.01, 0.99)
    assert abs(orc.logloss(y, p) - log_loss(y, p)) < 1e-12


def _c_oracle(built):
    lib = C.CDLL(os.path.join(os.path.dirname(built.ORACLE_LIB), "libfnn_oracle.so"))

    class Cfg(C.Structure):
        _fields_ = [("F", C.c_int), ("K", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("lr", C.c_double),
                    ("lambda1", C.c_double), ("lambda_fm", C.c_double), ("w0", C.c_double)]
    lib.oracle_train_step.restype = C.c_double
    return lib, Cfg


def test_c_port_matches_numpy_oracle(built):
    lib, Cfg = _c_oracle(built)
    sizes = synth.field_sizes_tiny(300)
    rows = np.random.RandomState(2).standard_normal((300, K)) * 0.05
    ids = np.ascontiguousarray(synth.zipf_ids(50, sizes, 1.1, 3))
    ids[4, 7] = -1
    y = (np.random.RandomState(4).uniform(size=50) < 0.3).astype(np.float64)
    h1, h2 = 20, 9
    p = orc.init_fnn_weights(XDIM, h1, h2, 'tanh', 1234)
    p['w3'] = np.random.RandomState(5).uniform(-.1, .1, h2); p['b3'] = 0.03
    r1 = (np.random.RandomState(6).uniform(size=h1) < .5).astype(np.float64)
    r2 = (np.random.RandomState(7).uniform(size=h2) < .5).astype(np.float64)
    cfg = Cfg(F, K, h1, h2, 0.01, 0.02, 0.1, -3.0)
    rows_c = rows.copy()
    pc = {k: (np.ascontiguousarray(v.copy()) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    b3 = C.c_double(p['b3'])
    gx = np.empty((50, XDIM)); pd = np.empty(50)
    dp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    loss_c = lib.oracle_train_step(C.byref(cfg), dp(rows_c), dp(ids), dp(y), 50, dp(r1), dp(r2), 50,
                                   dp(pc['w1']), dp(pc['b1']), dp(pc['w2']), dp(pc['b2']), dp(pc['w3']),
                                   C.byref(b3), dp(gx), dp(pd))
    ref = orc.train_step(p, rows, -3.0, ids, y, r1, r2, 0.01, 0.02, 0.1)
    assert abs(loss_c - ref['loss']) < 1e-10
    np.testing.assert_allclose(gx, ref['gx'], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(pd, ref['p_drop'], rtol=1e-12)
    np.testing.assert_allclose(rows_c, rows, rtol=1e-12, atol=1e-15)
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        np.testing.assert_allclose(pc[k], p[k], rtol=1e-10, atol=1e-13)
    assert abs(b3.value - p['b3']) < 1e-13


def test_golden_vectors_reproduce(golden_dir):
    """The oracle regenerates the committed fixtures bit for bit (NumPy float64, fixed seeds)."""
    init = np.load(os.path.join(golden_dir, 'init.npz'))
    p = orc.init_fnn_weights(XDIM, H1, H2, 'tanh', seed=1234)
    assert np.array_equal(init['w1_corner'], p['w1'][:2, :4])
    assert np.array_equal(init['w2_corner'], p['w2'][:2, :4])
    masks = np.load(os.path.join(golden_dir, 'masks.npz'))
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    assert list(masks['seeds']) == ms.seeds
    for i in range(3):
        r1, r2 = ms.next()
        assert np.array_equal(masks['r1'][i], r1.astype(np.uint8))
        assert np.array_equal(masks['r2'][i], r2.astype(np.uint8))
    g = np.load(os.path.join(golden_dir, 'step.npz'))
    w0, k, xdim, fw, ff = orc.parse_fm_model(os.path.join(golden_dir, 'demo', 'fm.model.txt'))
    assert (w0, k, xdim, len(fw)) == (-3.0, K, XDIM, 1000)
    feat_ids = synth.feat_id_of_row(np.arange(1000))
    rows = np.array([fw[int(f)] for f in feat_ids])
    pp = orc.init_fnn_weights(XDIM, H1, H2, 'tanh', seed=1234)
    pp['w3'] = g['w3']; pp['b3'] = float(g['b3'])
    pp = {k_: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else float(np.float32(v)))
          for k_, v in pp.items()}
    res = orc.train_step(pp, rows, w0, g['ids'], g['y'], g['r1'].astype(float), g['r2'].astype(float),
                         float(g['lr']), float(g['lambda1']), float(g['lambda_fm']))
    assert np.array_equal(res['x'], g['x']) and np.array_equal(res['gx'], g['gx'])
    assert res['loss'] == float(g['loss'])
    assert np.array_equal(rows[g['touched']], g['rows_after'])
    assert np.array_equal(pp['w3'], g['w3_after'])


def test_demo_text_files_parse(golden_dir):
    feats, y = orc.parse_line(open(os.path.join(golden_dir, 'demo', 'train.fm.txt')).readline())
    assert len(feats) == 16 and y in (0, 1)


# ------------------------------------------------------------------ RBM oracle (A7 / A7')
def test_sparse_cd1_matches_loop_form_of_the_reference():
    """An independent restatement with the reference's own scalar double loops
    (rbm_sparse.py:340-366, 447-495) must agree with the vectorised oracle step."""
    from oracle import rbm_oracle as ro
    rng = np.random.RandomState(5)
    nvis, H, S = 50, 6, 4
    st = ro.SparseRBMState(nvis, H, S, rng)
    st.weightstep[:] = rng.standard_normal((S, H)) * 1e-3         # a non-trivial momentum buffer
    W, vb, hb, ws = st.W.copy(), st.visbias.copy(), st.hidbias.copy(), st.weightstep.copy()
    keys, v = [3, 9, 10, 41], [0, 1, 0, 1]
    draw = np.random.RandomState(9)
    u = np.random.RandomState(9).uniform(size=(1, H))
    err = ro.sparse_cd1_example(st, keys, v, draw)
    hid = np.zeros(H)
    for i in range(H):                                           # _update_hidden
        sm = 0.0
        for j, f in enumerate(keys):
            sm += W[f][i] * float(v[j])
        hid[i] = 1.0 / (1.0 + np.exp(-(sm + hb[i])))
    poscorr = np.outer(np.array(v, float), hid)
    posact = hid.copy()
    hs = np.where(u[0] < hid, 1.0, np.floor(hid))
    vis = np.array([1.0 / (1.0 + np.exp(-(np.dot(hs, W[f]) + vb[f]))) for f in keys])
    hid2 = np.array([1.0 / (1.0 + np.exp(-(sum(W[f][i] * vis[j] for j, f in enumerate(keys)) + hb[i]))) for i in range(H)])
    step = poscorr - np.outer(vis, hid2)
    for j, f in enumerate(keys):
        step[j] -= 0.0002 * W[f]
    step *= 1e-4
    ws2 = ws * 0.9 + step
    for j, f in enumerate(keys):
        np.testing.assert_allclose(st.W[f], W[f] + 2 * ws2[j], rtol=1e-13, atol=1e-16)     # applied twice
        assert abs(st.visbias[f] - (vb[f] + (v[j] - vis[j]) * 1e-4)) < 1e-15
    np.testing.assert_allclose(st.hidbias, hb + (posact - hid2) * 1e-4, rtol=1e-13)
    np.testing.assert_allclose(st.weightstep, ws2, rtol=1e-13)
    assert abs(err - ((vis - np.array(v)) ** 2).sum()) < 1e-14
    untouched = [r for r in range(nvis) if r not in keys]
    assert np.array_equal(st.W[untouched], W[untouched])


def test_dense_cd1_and_lower_layers():
    from oracle import rbm_oracle as ro
    rng = np.random.RandomState(2)
    W0, b0 = rng.standard_normal((30, 5)), rng.standard_normal(5)
    W1, b1 = rng.standard_normal((5, 4)), rng.standard_normal(4)
    dicts = [ro.dense_line_dict([3, 8]), ro.dense_line_dict([9, 10])]
    out = ro.lower_layers([W0, b0, W1, b1], dicts)
    pre = np.array([W0[3] + W0[8] + b0, W0[10] + b0])            # 9 was zeroed by 10's fake
    np.testing.assert_allclose(out, 1 / (1 + np.exp(-(pre @ W1 + b1))), rtol=1e-13)   # no sigmoid in between
    st = ro.DenseRBMState(4, 3, rng)
    W = st.W.copy()
    batch = rng.uniform(size=(7, 4))
    e1 = ro.dense_cd1_batch(st, batch, np.random.RandomState(1))
    assert e1 > 0 and st.W.shape == (4, 3) and not np.array_equal(st.W, W)
    np.testing.assert_allclose(st.W - W, st.weightstep, rtol=1e-6, atol=1e-15)          # first step: W += momentum buffer
    with pytest.raises(AssertionError):
        ro.dense_cd_train(st, [W0, b0], [[3, 8]] * 4, rng, minibatch=2)               # empty last batch in the reference


# ------------------------------------------------------------------ N2: denoising autoencoders
def test_dae_oracle_gradients_and_quirks(tmp_path):
    """oracle/dae_oracle.py: finite differences of the tied-weight dA step; Q1 (sparse_da returns the
    un-trained table), Q2 (state before the last update), Q3 (running sum over hidden units)."""
    from oracle import dae_oracle as do
    rng = np.random.RandomState(0)
    W = rng.standard_normal((7, 5)) * 0.3; b = rng.standard_normal(5) * 0.1; bv = rng.standard_normal(7) * 0.1
    x = rng.uniform(0.1, 0.9, 7)
    c, gW, dy, d = do.da_grads(W, b, bv, x)
    eps = 1e-6
    for (i, j) in ((0, 0), (3, 2), (6, 4)):
        Wp, Wm = W.copy(), W.copy(); Wp[i, j] += eps; Wm[i, j] -= eps
        assert abs((do.da_cost(Wp, b, bv, x)[0] - do.da_cost(Wm, b, bv, x)[0]) / (2 * eps) - gW[i, j]) < 1e-7
    for j in (0, 4):
        bp, bm = b.copy(), b.copy(); bp[j] += eps; bm[j] -= eps
        assert abs((do.da_cost(W, bp, bv, x)[0] - do.da_cost(W, bm, bv, x)[0]) / (2 * eps) - dy[j]) < 1e-7
    for i in (0, 6):
        vp, vm = bv.copy(), bv.copy(); vp[i] += eps; vm[i] -= eps
        assert abs((do.da_cost(W, b, vp, x)[0] - do.da_cost(W, b, vm, x)[0]) / (2 * eps) - d[i]) < 1e-7
    lines = [([3 * i + 2 + 50 * t for i in range(4)], [1, 1, 0, 1]) for t in range(5)]
    table, b_pre, st = do.sparse_da(8, 6, lines, sparse_len=300, epochs=2)
    rs = np.random.RandomState(123); rs.randint(2 ** 30)
    rs.uniform(size=(8, 6)); rs.uniform(size=(300, 6))
    bound = 4 * np.sqrt(6. / (300 + 6))
    assert np.array_equal(table, rs.uniform(low=-bound, high=bound, size=(300, 6)))          # Q1
    assert not np.array_equal(b_pre, st['b']) and np.abs(st['b']).max() > 0                   # Q2
    res = [table, b_pre]
    h = do.propagate(res, lines[0][0])
    bag = sum(table[r] for r in lines[0][0])
    np.testing.assert_allclose(h, do.sigmoid(np.cumsum(bag) + b_pre))                         # Q3
    Wl, bl, st2 = do.da(6, 4, lines, res, epochs=1)
    assert not np.array_equal(Wl, st2['W'])                                                   # Q2 again


def test_vectorised_cpu_variant_equals_the_sequential_oracle():
    """oracle.train_step_vec (the multi-core CPU baseline of bench.py) against the line-by-line
    restatement: same gather, same dense step, same rows after the decayed sparse update, with
    duplicates, an empty field and a global batch length."""
    rng = np.random.RandomState(3)
    F, K, H1, H2, B = 16, 4, 9, 5, 60
    sizes = [5] * F
    rows = rng.standard_normal((sum(sizes), K)) * 0.2
    off = np.cumsum([0] + sizes[:-1])
    ids = (off + rng.randint(0, 5, (B, F))).astype(np.int32)
    ids[7, 3] = -1
    y = (rng.uniform(size=B) < 0.4).astype(np.float64)
    p = {'w1': rng.standard_normal((1 + F * K, H1)) * 0.3, 'b1': np.zeros(H1), 'w2': rng.standard_normal((H1, H2)) * 0.3,
         'b2': np.zeros(H2), 'w3': rng.standard_normal(H2) * 0.3, 'b3': 0.1}
    r1 = (rng.uniform(size=H1) < 0.5).astype(np.float64); r2 = (rng.uniform(size=H2) < 0.5).astype(np.float64)
    pa = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ra, rb = rows.copy(), rows.copy()
    a = orc.train_step(pa, ra, -1.5, ids, y, r1, r2, 0.05, 0.01, 0.2, b_size=80)
    b = orc.train_step_vec(p, rb, -1.5, ids, y, r1, r2, 0.05, 0.01, 0.2, b_size=80)
    np.testing.assert_allclose(orc.gather_vec(rows, ids, -1.5), orc.gather(rows, ids, -1.5), rtol=0, atol=0)
    np.testing.assert_allclose(b['gx'], a['gx'], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(rb, ra, rtol=1e-12, atol=1e-14)
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        np.testing.assert_allclose(p[k], pa[k], rtol=1e-13, atol=1e-15)


def test_fm_oracle_gradient_is_the_dense_sgd_of_the_loss():
    """oracle/fm_oracle.py: one sgd_step equals theta - lr * numerical gradient of loss_value (data term
    + lambda * l2_loss over ALL parameters), with a repeated row and an absent field."""
    from oracle import fm_oracle as fo
    rng = np.random.RandomState(2)
    rows = rng.standard_normal((12, 4)) * 0.3
    ids = rng.randint(0, 12, (5, 3)).astype(np.int32); ids[1, 2] = -1; ids[2] = ids[0]
    y = (rng.uniform(size=5) < 0.5).astype(np.float64)
    for reduce_mean in (True, False):
        r = rows.copy()
        b_new, data, p = fo.sgd_step(r, 0.2, ids, y, 0.1, 0.05, reduce_mean)
        eps = 1e-6
        for (i, j) in ((int(ids[0, 0]), 0), (int(ids[0, 1]), 2), (11, 3)):
            rp, rm = rows.copy(), rows.copy(); rp[i, j] += eps; rm[i, j] -= eps
            g = (fo.loss_value(rp, 0.2, ids, y, 0.05, reduce_mean)[0] - fo.loss_value(rm, 0.2, ids, y, 0.05, reduce_mean)[0]) / (2 * eps)
            assert abs(r[i, j] - (rows[i, j] - 0.1 * g)) < 1e-8
        gb = (fo.loss_value(rows, 0.2 + eps, ids, y, 0.05, reduce_mean)[0] - fo.loss_value(rows, 0.2 - eps, ids, y, 0.05, reduce_mean)[0]) / (2 * eps)
        assert abs(b_new - (0.2 - 0.1 * gb)) < 1e-8
        assert abs(data - fo.loss_value(rows, 0.2, ids, y, 0.05, reduce_mean)[1]) < 1e-12


def test_ipnn_oracle_ftrl_rule():
    """oracle.ftrl_step restates TensorFlow's ApplyFtrl (python/tf_util.py:21-24, defaults): one variable by hand, and the
    dense-gradient consequence for untouched rows."""
    from oracle import ipnn_oracle as io
    rng = np.random.RandomState(5)
    F, K = 4, 3
    table = rng.standard_normal((30, K)) * 0.2
    ids = np.stack([rng.randint(0, 10, size=8) + 10 * 0 for _ in range(F)], 1)        # rows 10.. never touched
    y = (rng.uniform(size=8) < 0.5).astype(np.float64)
    d = [F * K + F * (F - 1) // 2 + 1, 5, 1]
    params = {'b': 0.1, 'W': [rng.uniform(-.3, .3, (d[i], d[i + 1])) for i in range(2)], 'bias': [rng.uniform(-.1, .1, d[i + 1]) for i in range(2)]}
    w0 = params['W'][1][2, 0]
    _, _, g = io.loss_and_grads(params, table, ids, y, 'tanh')
    g0 = g['W'][1][2, 0]
    st = io.ftrl_state(params, table)
    lr = 0.05
    io.ftrl_step(params, table, ids, y, 'tanh', lr, st)
    na = 0.1 + g0 * g0
    lin = g0 - (np.sqrt(na) - np.sqrt(0.1)) / lr * w0
    assert abs(params['W'][1][2, 0] - (-lin / (np.sqrt(na) / lr))) < 1e-15
    assert abs(st['W'][1][0][2, 0] - na) < 1e-15 and abs(st['W'][1][1][2, 0] - lin) < 1e-15
    assert not table[10:].any() and table[:10].any()


def test_rbm_oracle_minibatch_of_one_is_the_online_trainer():
    """oracle.sparse_cd1_minibatch with batches of one example reproduces sparse_cd1_example exactly (the mini-batch mode
    generalises the reference's online update; it is not the reference's schedule for M > 1)."""
    from oracle import rbm_oracle as ro
    rng = np.random.RandomState(2)
    feats = [sorted(rng.choice(np.arange(1, 60, 2), size=16, replace=False).tolist()) for _ in range(12)]
    a = ro.SparseRBMState(62, 10, 32, np.random.RandomState(9))
    b = ro.SparseRBMState(62, 10, 32, np.random.RandomState(9))
    ra, rb = np.random.RandomState(4), np.random.RandomState(4)
    for f in feats:
        keys, v = ro.sparse_line_dict(f)
        ea = ro.sparse_cd1_example(a, keys, v, ra)
        eb = ro.sparse_cd1_minibatch(b, [(keys, v)], rb)
        assert abs(ea - eb) < 1e-12
    for x, y in ((a.W, b.W), (a.visbias, b.visbias), (a.hidbias, b.hidbias), (a.weightstep, b.weightstep)):
        np.testing.assert_allclose(x, y, rtol=0, atol=1e-15)


def test_more_golden_vectors_reproduce(golden_dir):
    """tests/golden/make_golden_more.py fixtures (rbm_sparse, rbm_dense, snn_step, ip_l7): the oracles reproduce them."""
    from oracle import ipnn_oracle as ipo
    from oracle import rbm_oracle as ro

    class Replay(object):
        def __init__(self, u):
            self.u, self.i = u, 0

        def uniform(self, size=None):
            n = int(np.prod(size)) // self.u.shape[1]
            out = self.u[self.i:self.i + n].reshape(size)
            self.i += n
            return out
    g = np.load(os.path.join(golden_dir, 'rbm_sparse.npz'))
    st = ro.SparseRBMState(g['W0'].shape[0], g['W0'].shape[1], 32, np.random.RandomState(0))
    st.W, st.visbias, st.hidbias = g['W0'].copy(), g['vb0'].copy(), g['hb0'].copy()
    rp = Replay(g['unif'])
    err = sum(ro.sparse_cd1_example(st, list(k), v, rp) for k, v in zip(g['vid'], g['vval']))
    np.testing.assert_allclose(st.W, g['W_online'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(st.weightstep, g['ws_online'], rtol=0, atol=1e-16)
    assert abs(err - float(g['err_online'])) < 1e-12
    g = np.load(os.path.join(golden_dir, 'rbm_dense.npz'))
    ds = ro.DenseRBMState(12, 8, np.random.RandomState(0))
    ds.W, ds.visbias, ds.hidbias = g['W0'].copy(), g['vb0'].copy(), g['hb0'].copy()
    rp = Replay(g['unif'])
    e = [ro.dense_cd1_batch(ds, g['X'], rp), ro.dense_cd1_batch(ds, g['X'], rp)]
    np.testing.assert_allclose(ds.W, g['W'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(e, g['err'], rtol=1e-13)
    g = np.load(os.path.join(golden_dir, 'snn_step.npz'))
    p = {k: (g['p0_' + k].copy() if g['p0_' + k].ndim else float(g['p0_' + k])) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')}
    ww, bb = g['ww0'].copy(), g['bb0'].copy()
    res = orc.snn_train_step(p, ww, bb, g['ids'], g['y'], g['r1'].astype(float), g['r2'].astype(float), float(g['lr']), float(g['lambda1']))
    np.testing.assert_allclose(res['p_drop'], g['p_drop'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(ww[g['touched']], g['rows_after'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(p['w1'], g['p1_w1'], rtol=0, atol=1e-15)
    g = np.load(os.path.join(golden_dir, 'ip_l7.npz'))
    params = {'b': float(g['b']), 'W': [g['W%d' % i] for i in range(8)], 'bias': [g['bias%d' % i] for i in range(8)]}
    _, z1 = ipo.z1_of(g['table'], params['b'], g['ids'])
    np.testing.assert_allclose(z1, g['z1'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(ipo.predict(params, g['table'], g['ids'], 'relu'), 1 / (1 + np.exp(-g['logits'])), rtol=1e-12)


def test_scatter_over_feature_lists_reduces_to_the_id_matrix_form():
    """With one feature per field the reference's loop over feature lists (scatter_sgd_feats) IS scatter_sgd on the id
    matrix; with a shadowed or repeated feature it visits more rows / visits a row twice (python/FNN_wnzh.py:300-306)."""
    rng = np.random.RandomState(0)
    F, K, B = 4, 3, 50
    sizes = [3, 5, 2, 6]
    offs = np.cumsum([0] + sizes[:-1])
    rows = rng.standard_normal((sum(sizes), K))
    ids = np.stack([offs[f] + rng.randint(0, sizes[f], size=B) for f in range(F)], axis=1)
    gx = rng.standard_normal((B, 1 + F * K))
    field_of = {int(r): int(np.searchsorted(offs, r, side='right') - 1) for r in range(sum(sizes))}
    ident = {r: r for r in field_of}
    a, b = rows.copy(), rows.copy()
    orc.scatter_sgd(a, ids, gx, 0.1, 0.3)
    orc.scatter_sgd_feats(b, [list(map(int, ids[t])) for t in range(B)], ident, field_of, gx, 0.1, 0.3)
    assert np.array_equal(a, b)
    # one line, field 1 holds features 3 then 4 (4 wins the gather), feature 0 twice: rows 3 and 4 both move, row 0 twice
    c = 1 - 2 * 0.3 * 0.1 / 1
    r = rows.copy()
    orc.scatter_sgd_feats(r, [[0, 3, 4, 0]], ident, field_of, gx[:1], 0.1, 0.3, b_size=1)
    g0, g1 = gx[0, 1:1 + K], gx[0, 1 + K:1 + 2 * K]
    np.testing.assert_allclose(r[3], rows[3] * c - 0.1 * g1, rtol=1e-14)
    np.testing.assert_allclose(r[4], rows[4] * c - 0.1 * g1, rtol=1e-14)
    np.testing.assert_allclose(r[0], (rows[0] * c - 0.1 * g0) * c - 0.1 * g0, rtol=1e-14)
