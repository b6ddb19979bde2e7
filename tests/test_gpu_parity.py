"""GPU parity tests: the HIP path, called through the C ABI (libfnn_hip.so), against the float64
oracle on the same seeded inputs, against the committed golden fixtures, and through
size-independent properties at the full BASELINE shape.

Tolerances (north_star: logloss/AUC within 1e-4 absolute of the CPU reference):
  gather                exact (a copy)
  f32 mode, one step    p_drop rtol 1e-4, gx rtol 2e-3 (+atol 1e-6), rows rtol 1e-5 / atol 2e-7
  bf16 mode, one step   p_drop atol 2e-2, gx within 5e-2 of the gx scale (bf16 has 8 mantissa bits)
  epochs on the demo    |d logloss| <= 1e-4, |d AUC| <= 1e-4 in f32 mode
"""
import os

import numpy as np
import pytest

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import _capi, synth
from deep_ctr_amd.engine import FNNEngine, FNNError

pytestmark = pytest.mark.gpu

F, K, H1, H2 = 16, 11, 300, 100
XDIM = 1 + F * K


def f32r(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def make_problem(B, n_rows=1000, seed=0, h1=H1, h2=H2, dup_col=None, empty=()):
    sizes = synth.field_sizes_tiny(n_rows) if n_rows < 100000 else synth.field_sizes_ipinyou(n_rows)
    rows = synth.fm_table(sum(sizes), K, 0.05, seed)
    fo = synth.field_of_row(sizes)
    ids = synth.zipf_ids(B, sizes, 1.1, seed + 1)
    if dup_col is not None:
        ids[:, dup_col] = ids[0, dup_col]
    for (t, f) in empty:
        ids[t, f] = -1
    rng = np.random.RandomState(seed + 2)
    y = (rng.uniform(size=B) < 0.2).astype(np.float32)
    p = orc.init_fnn_weights(XDIM, h1, h2, 'tanh', seed=1234)
    p['w3'] = rng.uniform(-0.1, 0.1, h2)
    p['b1'] = rng.uniform(-0.1, 0.1, h1)
    p['b2'] = rng.uniform(-0.1, 0.1, h2)
    p['b3'] = 0.05
    p = {k: (f32r(v) if isinstance(v, np.ndarray) else float(np.float32(v))) for k, v in p.items()}
    r1 = (rng.uniform(size=h1) < 0.5).astype(np.uint8)
    r2 = (rng.uniform(size=h2) < 0.5).astype(np.uint8)
    return rows, fo, ids, y, p, r1, r2


def make_engine(rows, fo, p, w0=-3.0, prec='f32', max_batch=4096, lr=0.001, lam1=0.0, lamfm=0.1, h1=H1,
                h2=H2, acti='tanh'):
    eng = FNNEngine(F, K, h1, h2, max_batch=max_batch, precision=prec, acti_type=acti, lr=lr, lambda1=lam1,
                    lambda_fm=lamfm)
    eng.set_table(rows, fo, w0)
    eng.set_dense(p)
    return eng


def test_library_reports_version(built):
    lib = _capi.load()
    assert b"gfx950" in lib.fnn_version()


@pytest.mark.parametrize("B", [1, 7, 100, 1000])
def test_gather_exact(built, B):
    rows, fo, ids, y, p, r1, r2 = make_problem(B, empty=[(0, 3)])
    eng = make_engine(rows, fo, p)
    x = eng.gather(ids).cpu().numpy()
    ref = orc.gather(rows.astype(np.float64), ids, -3.0).astype(np.float32)
    assert np.array_equal(x, ref)
    eng.close()


def test_set_get_roundtrip(built):
    rows, fo, ids, y, p, r1, r2 = make_problem(10)
    eng = make_engine(rows, fo, p)
    assert np.array_equal(eng.get_table(), rows)
    assert np.array_equal(eng.get_rows([3, 999, 0]), rows[[3, 999, 0]])
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        assert np.array_equal(d[k], p[k].astype(np.float32)), k
    assert d['b3'] == np.float32(p['b3'])
    eng.close()


def _check_step(eng, rows, ids, y, p, r1, r2, lr, lam1, lamfm, w0=-3.0, b_size=0, tol=1.0, acti='tanh', tol_table=None):
    rows64 = rows.astype(np.float64).copy()
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    out = eng.train_step(ids, y, r1, r2, b_size=b_size, want_p=True, want_gx=True)
    x = orc.gather(rows64, ids, w0)
    gx, pre, loss, p_drop, g = orc.train_call(p64, x, y.astype(np.float64), r1.astype(np.float64),
                                              r2.astype(np.float64), lr, lam1, acti)
    orc.scatter_sgd(rows64, ids, gx, lr, lamfm, b_size if b_size > 0 else None)
    np.testing.assert_allclose(out['p'].cpu().numpy(), p_drop, rtol=1e-4 * tol, atol=1e-6 * tol)
    gscale = np.abs(gx).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), gx, rtol=2e-3 * tol, atol=2e-5 * gscale * tol + 1e-9)
    assert abs(out['loss'] - loss) <= 2e-5 * tol * max(1.0, abs(loss))
    tt = tol if tol_table is None else tol_table
    np.testing.assert_allclose(eng.get_table(), rows64, rtol=1e-5 * tt, atol=2e-7 * tt)
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        gs = lr * np.abs(g[k]).max()
        np.testing.assert_allclose(d[k], p64[k], rtol=1e-5 * tol, atol=1e-3 * gs * tol + 1e-7, err_msg=k)
    assert abs(d['b3'] - p64['b3']) <= 1e-5 * tol
    return rows64, p64


@pytest.mark.parametrize("B,kw", [
    (1, {}), (5, {"empty": [(2, 3), (4, 0)]}), (64, {"dup_col": 6}), (100, {}), (257, {"dup_col": 0}),
    (1000, {"empty": [(0, 15)]}),
])
def test_train_step_f32_vs_oracle(built, B, kw):
    rows, fo, ids, y, p, r1, r2 = make_problem(B, seed=B, **kw)
    eng = make_engine(rows, fo, p, lr=0.01, lam1=0.02, lamfm=0.1)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.02, 0.1)
    eng.close()


@pytest.mark.parametrize("B,kw", [(5, {"empty": [(2, 3), (4, 0)]}), (257, {"dup_col": 0}), (1000, {"empty": [(0, 15)]})])
def test_train_step_f32_vs_oracle_four_waves(built, monkeypatch, B, kw):
    """The strip kernel's four-wave form (FNN_STEP1_WAVES=4; eight waves is the default since round 3) against the oracle."""
    monkeypatch.setenv('FNN_STEP1_WAVES', '4')
    rows, fo, ids, y, p, r1, r2 = make_problem(B, seed=B, **kw)
    eng = make_engine(rows, fo, p, lr=0.01, lam1=0.02, lamfm=0.1)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.02, 0.1)
    eng.close()


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
def test_layer_by_layer_path_matches_strip_kernel(built, prec, monkeypatch):
    """FNN_NO_FUSE=1 selects the layer-by-layer kernels (the path for shapes the fused strip
    kernel is not instantiated for); both paths must agree with the oracle and with each other."""
    rows, fo, ids, y, p, r1, r2 = make_problem(300, seed=17, dup_col=6, empty=[(1, 2)])
    outs = []
    for nofuse in ('0', '1'):
        monkeypatch.setenv('FNN_NO_FUSE', nofuse)
        eng = make_engine(rows, fo, p, prec=prec, lr=0.01, lam1=0.02)
        if prec == 'f32':
            _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.02, 0.1)
        else:
            eng.train_step(ids, y, r1, r2)
        outs.append((eng.get_table(), eng.get_dense(), eng.predict(ids).cpu().numpy()))
        eng.close()
    tol = 1e-6 if prec == 'f32' else 2e-3
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=tol, atol=tol * 1e-1)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=tol * 10, atol=tol)
    for k in ('w1', 'w2', 'w3'):
        np.testing.assert_allclose(outs[0][1][k], outs[1][1][k], rtol=tol * 10, atol=tol)


def test_virtual_two_rank_dp_equals_single_gpu_dense(built):
    """Two engines stand for two ranks: each runs fnn_step_begin on its contiguous half with the
    GLOBAL batch length, the flat buckets are summed (what the RCCL all-reduce does) and
    fnn_step_end applies them.  Dense tensors must equal the single-engine full-batch step, also
    with an L2 term (which must not be all-reduced), and each rank's table holds its shard's rows."""
    import torch
    rows, fo, ids, y, p, r1, r2 = make_problem(512, seed=41, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1)
    full = make_engine(rows, fo, p, **kw)
    full.train_step(ids, y, r1, r2)
    ref_dense, ref_rows = full.get_dense(), full.get_table()
    full.close()
    ranks = [make_engine(rows, fo, p, **kw) for _ in range(2)]
    halves = [slice(0, 256), slice(256, 512)]
    buckets = [e.step_begin(ids[h], y[h], r1, r2, b_size=512) for e, h in zip(ranks, halves)]
    for e in ranks:
        e.sync()
    tot = buckets[0] + buckets[1]
    for e, b in zip(ranks, buckets):
        b.copy_(tot)
    torch.cuda.synchronize()
    for e in ranks:
        e.step_end()
        e.sync()
    for e in ranks:
        d = e.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(d[k] - p[k].astype(np.float32)).max() + 1e-12
            assert np.abs(d[k] - ref_dense[k]).max() <= 2e-4 * scale + 1e-7, k
        assert abs(d['b3'] - ref_dense['b3']) < 1e-6
    # rows touched by one half only must match the full-batch result on the rank that owns that half
    t0, t1 = set(np.unique(ids[halves[0]])), set(np.unique(ids[halves[1]]))
    only0 = np.array(sorted(t0 - t1)); only1 = np.array(sorted(t1 - t0))
    np.testing.assert_allclose(ranks[0].get_rows(only0), ref_rows[only0], rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(ranks[1].get_rows(only1), ref_rows[only1], rtol=1e-5, atol=2e-7)
    assert np.array_equal(ranks[0].get_rows(only1), rows[only1])       # never touched on rank 0
    for e in ranks:
        e.close()


def test_virtual_two_rank_exchange_mode_equals_single_gpu_tables(built):
    """Exact data-parallel mode (fnn_sparse_grad + fnn_step_scatter_global): two engines stand for
    two ranks, (ids, gx') of the two shards are concatenated (what the all-gather does, the shorter
    shard padded with -1 ids), every rank applies the global batch's row updates -> BOTH tables equal
    the single-engine full-batch step on every touched row, two steps in a row."""
    import torch
    rows, fo, ids, y, p, r1, r2 = make_problem(2 * 500, seed=43, dup_col=6)
    kw = dict(lr=0.01, lam1=0.0, lamfm=0.1)
    full = make_engine(rows, fo, p, **kw)
    ranks = [make_engine(rows, fo, p, **kw) for _ in range(2)]
    for step in range(2):
        sl = slice(step * 500, (step + 1) * 500)
        ids_s, y_s = ids[sl], y[sl]
        full.train_step(ids_s, y_s, r1, r2)
        cut = [slice(0, 256), slice(256, 500)]                         # unequal shards: 256 + 244
        buckets = [e.step_begin(ids_s[h], y_s[h], r1, r2, b_size=500) for e, h in zip(ranks, cut)]
        dev = buckets[0].device
        ids_g = torch.full((512, F), -1, dtype=torch.int32, device=dev)
        gx_g = torch.zeros((512, 256), dtype=torch.float32, device=dev)
        for r, (e, h) in enumerate(zip(ranks, cut)):
            e.sync()
            n = h.stop - h.start
            ids_g[256 * r:256 * r + n] = torch.as_tensor(ids_s[h]).to(dev)
            gx_g[256 * r:256 * r + n] = e.sparse_grad(n)
        tot = buckets[0] + buckets[1]
        torch.cuda.synchronize()
        for e, b in zip(ranks, buckets):
            b.copy_(tot)
        torch.cuda.synchronize()
        for e in ranks:
            e.step_scatter_global(ids_g, gx_g)
            e.step_end()
            e.sync()
    ref_rows, ref_dense = full.get_table(), full.get_dense()
    touched = np.unique(ids[:1000])
    change = np.abs(ref_rows[touched] - rows[touched].astype(np.float32)).max()
    for e in ranks:
        got = e.get_table()
        assert np.abs(got[touched] - ref_rows[touched]).max() <= 2e-4 * change + 1e-7
        untouched = np.setdiff1d(np.arange(rows.shape[0]), touched)
        assert np.array_equal(got[untouched], ref_rows[untouched])
        d = e.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(ref_dense[k] - p[k].astype(np.float32)).max() + 1e-12
            assert np.abs(d[k] - ref_dense[k]).max() <= 5e-4 * scale + 1e-7, k
    assert np.array_equal(ranks[0].get_table(), ranks[1].get_table())   # replicas stay identical
    for e in ranks + [full]:
        e.close()


@pytest.mark.parametrize("B,prec", [(64, 'f32'), (700, 'f32'), (4096, 'f32'), (700, 'bf16')])
def test_two_features_of_one_field_both_rows_are_updated(built, B, prec):
    """python/FNN_wnzh.py:300-306 walks EVERY feature of a line: a feature that a later feature of the same field shadows
    in the layer-one array (:91-96) still takes `row * c - lr * gx[slot]`, and a feature listed twice takes it twice.  The
    lines here hold up to three features of a field (fnn_set_shadowed carries the shadowed ones); reference: the oracle's
    loop over feature lists, two steps in a row, the second without shadowed features (the list is consumed)."""
    rng = np.random.RandomState(B)
    sizes = synth.field_sizes_tiny(1000)
    offs = np.cumsum([0] + sizes[:-1])
    rows = synth.fm_table(sum(sizes), K, 0.05, 3)
    fo = synth.field_of_row(sizes)
    _, _, _, _, p, r1, r2 = make_problem(8, seed=B)
    feats = []                                              # feature id == row id here
    for t in range(B):
        ft = []
        for f in range(F):
            n = 1 if rng.uniform() < 0.8 else int(rng.randint(0, 4))     # 0..3 features of this field
            pick = list(offs[f] + rng.randint(0, min(sizes[f], 5), size=n))   # few distinct rows: many repeats across examples
            if n == 3 and rng.uniform() < 0.5:
                pick[2] = pick[0]                                            # the same feature twice in one line
            ft += pick
        rng.shuffle(ft)
        feats.append([int(v) for v in ft])
    fbig = int(np.argmax(sizes))                            # a row that ONLY ever appears shadowed: first on line 0, its field's
    rare = int(offs[fbig] + sizes[fbig] - 1)                # usual feature behind it
    feats[0] = [rare] + feats[0] + [int(offs[fbig])]
    ident = {int(r): int(r) for r in range(rows.shape[0])}
    field_of = {int(r): int(fo[r]) for r in range(rows.shape[0])}
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    lr, lam1, lamfm = 0.01, 0.0, 0.1
    rows64 = rows.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.train_step_feats(p64, rows64, -3.0, feats, ident, field_of, F, y.astype(np.float64), r1.astype(float), r2.astype(float),
                               lr, lam1, lamfm)
    ids = ref['ids'].astype(np.int32)
    shadow = []
    for t, ft in enumerate(feats):
        seen = {}
        for feat in ft:
            if field_of[feat] in seen:
                shadow.append((t, field_of[feat], seen[field_of[feat]]))
            seen[field_of[feat]] = feat
    assert len(shadow) > B // 20
    eng = make_engine(rows, fo, p, prec=prec, lr=lr, lam1=lam1, lamfm=lamfm)
    eng.set_shadowed(np.asarray(shadow, np.int32))
    out = eng.train_step(ids, y, r1, r2, want_gx=True)
    got = eng.get_table()
    change = np.abs(rows64 - rows).max()
    tol = 3e-4 if prec == 'f32' else 8e-2
    assert np.abs(got - rows64).max() <= tol * change + 2e-7
    if prec == 'f32':
        assert abs(out['loss'] - ref['loss']) <= 2e-5 * abs(ref['loss'])
        # rows that ONLY shadowed features touch moved too (the round-1 path left them alone)
        main_rows = set(int(v) for v in ids[ids >= 0])
        only_shadow = np.array(sorted(set(s[2] for s in shadow) - main_rows))
        assert len(only_shadow) > 0 and np.all(np.abs(got[only_shadow] - rows[only_shadow]).max(axis=1) > 0)
    # second step, no shadowed features announced: the plain update again
    ids_b = synth.zipf_ids(B, sizes, 1.1, 9)
    eng.train_step(ids_b, y, r1, r2)
    orc.train_step(p64, rows64, -3.0, ids_b, y.astype(np.float64), r1.astype(float), r2.astype(float), lr, lam1, lamfm)
    change = np.abs(rows64 - rows).max()
    assert np.abs(eng.get_table() - rows64).max() <= tol * change + 2e-7
    eng.close()


def test_shadowed_list_is_validated(built):
    rows, fo, ids, y, p, r1, r2 = make_problem(32, seed=4)
    eng = make_engine(rows, fo, p)
    eng.set_shadowed(np.array([[40, 0, 1]], np.int32))          # example 40 of a batch of 32
    eng.train_step(ids, y, r1, r2, want_loss=False)
    with pytest.raises(FNNError) as e:
        eng.sync()
    assert e.value.code == _capi.FNN_ERR_RANGE
    eng.close()


def test_prefetch_ids_changes_nothing(built):
    """fnn_prefetch_ids is a scheduling hint: with or without it the state after several steps is
    bitwise identical (and so is a run where the hint named a batch that never came)."""
    import torch
    rows, fo, ids, y, p, r1, r2 = make_problem(1200, seed=31, dup_col=3)
    res = []
    for mode in ('none', 'next', 'wrong'):
        eng = make_engine(rows, fo, p, prec='bf16', lr=0.01)
        dev_ids = [torch.as_tensor(ids[j * 300:(j + 1) * 300]).to(eng.device).contiguous() for j in range(4)]
        for j in range(4):
            if mode == 'next' and j + 1 < 4:
                eng.prefetch_ids(dev_ids[j + 1])
            if mode == 'wrong':
                eng.prefetch_ids(dev_ids[(j + 2) % 4])
            eng.train_step(dev_ids[j], y[j * 300:(j + 1) * 300], r1, r2, want_loss=False)
        eng.sync()
        res.append((eng.get_table(), eng.get_dense()))
        eng.close()
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0])
        for k in ('w1', 'w2', 'w3', 'b1', 'b2'):
            assert np.array_equal(res[0][1][k], other[1][k])


def test_train_step_f32_global_batch_decay(built):
    """Data-parallel callers pass the GLOBAL batch length for the decay constant (:304)."""
    rows, fo, ids, y, p, r1, r2 = make_problem(50, seed=3)
    eng = make_engine(rows, fo, p, lr=0.05, lamfm=0.3)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.05, 0.0, 0.3, b_size=400)
    eng.close()


@pytest.mark.parametrize("acti", ['sigmoid', 'linear'])
def test_other_activations(built, acti):
    rows, fo, ids, y, p, r1, r2 = make_problem(40, seed=11)
    eng = make_engine(rows, fo, p, lr=0.01, acti=acti)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.0, 0.1, acti=acti)
    pr = eng.predict(ids).cpu().numpy()
    d = {k: (f32r(v) if isinstance(v, np.ndarray) else v) for k, v in eng.get_dense().items()}
    ref = orc.predict(d, orc.gather(eng.get_table().astype(np.float64), ids, -3.0), acti)
    np.testing.assert_allclose(pr, ref, rtol=1e-4, atol=1e-6)
    eng.close()


def test_small_hidden_sizes(built):
    rows, fo, ids, y, p, r1, r2 = make_problem(33, seed=5, h1=20, h2=9)
    eng = make_engine(rows, fo, p, lr=0.01, h1=20, h2=9)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.0, 0.1)
    eng.close()


def test_multi_step_sequence_f32(built):
    """Several consecutive steps: state carried on the device must track the oracle."""
    rows, fo, ids, y, p, r1, r2 = make_problem(600, seed=21)
    eng = make_engine(rows, fo, p, lr=0.002)
    rows64 = rows.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    for j in range(6):
        sl = slice(j * 100, (j + 1) * 100)
        m1, m2 = ms.next()
        out = eng.train_step(ids[sl], y[sl], m1.astype(np.uint8), m2.astype(np.uint8))
        ref = orc.train_step(p64, rows64, -3.0, ids[sl], y[sl].astype(np.float64), m1, m2, 0.002, 0.0, 0.1)
        assert abs(out['loss'] - ref['loss']) <= 1e-4 * abs(ref['loss'])
    np.testing.assert_allclose(eng.get_table(), rows64, rtol=1e-4, atol=1e-6)
    pr = eng.predict(ids).cpu().numpy()
    np.testing.assert_allclose(pr, orc.predict(p64, orc.gather(rows64, ids, -3.0)), rtol=2e-4, atol=1e-6)
    eng.close()


def test_golden_step(built, golden_dir):
    from deep_ctr_amd.data_fm import DataFM
    g = np.load(os.path.join(golden_dir, 'step.npz'))
    data = DataFM(os.path.join(golden_dir, 'demo', 'fm.model.txt'))
    rows, fo, w0 = data.table()
    p = orc.init_fnn_weights(XDIM, H1, H2, 'tanh', seed=1234)
    p['w3'] = g['w3']; p['b3'] = float(g['b3'])
    eng = make_engine(rows, fo, p, w0=w0, lr=float(g['lr']), lam1=float(g['lambda1']), lamfm=float(g['lambda_fm']))
    assert np.array_equal(eng.gather(g['ids']).cpu().numpy(), g['x'].astype(np.float32))
    out = eng.train_step(g['ids'], g['y'], g['r1'], g['r2'], want_p=True, want_gx=True)
    np.testing.assert_allclose(out['p'].cpu().numpy(), g['p_drop'], rtol=1e-4)
    np.testing.assert_allclose(out['gx'].cpu().numpy(), g['gx'], rtol=2e-3, atol=2e-5 * np.abs(g['gx']).max())
    assert abs(out['loss'] - float(g['loss'])) <= 2e-5 * float(g['loss'])
    np.testing.assert_allclose(eng.get_rows(g['touched']), g['rows_after'], rtol=1e-5, atol=2e-7)
    d = eng.get_dense()
    np.testing.assert_allclose(d['w3'], g['w3_after'], rtol=1e-5, atol=1e-7)
    assert abs(d['w1'].astype(np.float64).sum() - float(g['w1_after_sum'])) < 1e-2
    eng.close()


def _run_demo_epochs(golden_dir, prec, epochs=3):
    from deep_ctr_amd.data_fm import DataFM
    from deep_ctr_amd import dl_utils as ut
    from sklearn.metrics import log_loss, roc_auc_score
    demo = os.path.join(golden_dir, 'demo')
    data = DataFM(os.path.join(demo, 'fm.model.txt'))
    rows, fo, w0 = data.table()
    ut.seed_global(1234)
    p = ut.init_fnn_weights(XDIM, H1, H2, 'tanh')
    eng = make_engine(rows, fo, p, w0=w0, prec=prec, lr=0.001, lam1=0.0, lamfm=0.1)
    tr_ids, tr_y = data.load_ids(os.path.join(demo, 'train.fm.txt'))
    te_ids, te_y = data.load_ids(os.path.join(demo, 'test.fm.txt'))
    srng = ut.RandomStreams(234)
    srng.binomial(size=(1, XDIM), n=1, p=1)
    r1 = srng.binomial(size=(1, H1), n=1, p=0.5)
    r2 = srng.binomial(size=(1, H2), n=1, p=0.5)
    hist = []
    for ep in range(epochs):
        for j in range(len(tr_y) // 100):
            sl = slice(j * 100, (j + 1) * 100)
            eng.train_step(tr_ids[sl], tr_y[sl], r1.draw()[0], r2.draw()[0], want_loss=False)
        ptr = eng.predict(tr_ids).cpu().numpy().astype(np.float64)
        pte = eng.predict(te_ids).cpu().numpy().astype(np.float64)
        hist.append((roc_auc_score(tr_y, ptr), log_loss(tr_y, ptr), roc_auc_score(te_y, pte), log_loss(te_y, pte)))
    eng.close()
    return np.array(hist)


def test_demo_epochs_f32_logloss_auc_within_1e4(built, golden_dir):
    """BASELINE config 1 / north_star: logloss and AUC on the demo set within 1e-4 absolute of the
    float64 restatement (same init, masks, order), 3 epochs, reference defaults."""
    g = np.load(os.path.join(golden_dir, 'epoch.npz'))
    h = _run_demo_epochs(golden_dir, 'f32')
    assert np.abs(h[:, 0] - g['train_auc']).max() <= 1e-4
    assert np.abs(h[:, 1] - g['train_logloss']).max() <= 1e-4
    assert np.abs(h[:, 2] - g['test_auc']).max() <= 1e-4
    assert np.abs(h[:, 3] - g['test_logloss']).max() <= 1e-4


def test_demo_epochs_bf16_tracks_oracle(built, golden_dir):
    """bf16 throughput mode: same run.  Observed on MI355X (profiles/r02b_bf16_demo_deltas.json): logloss 1.3e-4 (train) /
    1.1e-4 (test), AUC 1.7e-4 (train) / 1.7e-3 (test; 500 examples: one swapped pair moves it by 1e-4) from the float64
    oracle; asserted at about twice that -- not the 1e-4 of the f32 mode, and stated as such."""
    g = np.load(os.path.join(golden_dir, 'epoch.npz'))
    h = _run_demo_epochs(golden_dir, 'bf16')
    d = {'train_logloss': float(np.abs(h[:, 1] - g['train_logloss']).max()), 'test_logloss': float(np.abs(h[:, 3] - g['test_logloss']).max()),
         'train_auc': float(np.abs(h[:, 0] - g['train_auc']).max()), 'test_auc': float(np.abs(h[:, 2] - g['test_auc']).max())}
    print("bf16 demo deltas vs the float64 oracle: %r" % (d,))
    try:                                                    # kept beside the profiles (DESIGN.md quotes them)
        import json
        os.makedirs(os.path.join(os.path.dirname(golden_dir), '..', 'gpurun_out'), exist_ok=True)
        json.dump(d, open(os.path.join(os.path.dirname(golden_dir), '..', 'gpurun_out', 'bf16_demo_deltas.json'), 'w'))
    except OSError:
        pass
    assert d['train_logloss'] <= 3e-4 and d['test_logloss'] <= 3e-4
    assert d['train_auc'] <= 4e-4 and d['test_auc'] <= 4e-3


def test_train_step_bf16_vs_oracle(built):
    rows, fo, ids, y, p, r1, r2 = make_problem(512, seed=9, dup_col=12)
    eng = make_engine(rows, fo, p, prec='bf16', lr=0.01)
    out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
    rows64 = rows.astype(np.float64)
    ref = orc.train_step(p, rows64, -3.0, ids, y.astype(np.float64), r1.astype(float), r2.astype(float),
                         0.01, 0.0, 0.1)
    assert np.abs(out['p'].cpu().numpy() - ref['p_drop']).max() < 2e-2
    gs = np.abs(ref['gx']).max()
    assert np.abs(out['gx'].cpu().numpy() - ref['gx']).max() < 5e-2 * gs
    assert abs(out['loss'] - ref['loss']) < 2e-2 * ref['loss']
    upd = np.abs(rows64 - rows.astype(np.float64)).max()          # largest row update of the step
    assert np.abs(eng.get_table() - rows64).max() < 5e-2 * upd + 1e-6
    eng.close()


def test_host_pointer_abi_like_the_integration_stub(built):
    """FNN_MEM_HOST: plain numpy arrays through the C ABI, as the reference-side stub of
    INTEGRATION.md does (no torch in the call path)."""
    import ctypes as C
    lib = _capi.load()
    rows, fo, ids, y, p, r1, r2 = make_problem(100, seed=51, dup_col=2)
    cfg = _capi.fnn_cfg(F, K, H1, H2, 4096, _capi.FNN_PREC_F32, 0, 0, 0.01, 0.0, 0.1, 0, None)
    h = C.c_void_p()
    assert lib.fnn_create(C.byref(cfg), C.byref(h)) == 0
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fo32 = np.ascontiguousarray(fo, np.int32)
    assert lib.fnn_set_table(h, ptr(rows), rows.shape[0], ptr(fo32), -3.0, _capi.FNN_MEM_HOST) == 0
    for layer, (wn, bn) in enumerate((('w1', 'b1'), ('w2', 'b2'), ('w3', 'b3')), 1):
        W = np.ascontiguousarray(p[wn], np.float32); b = np.ascontiguousarray(np.atleast_1d(p[bn]), np.float32)
        assert lib.fnn_set_dense(h, layer, ptr(W), ptr(b), _capi.FNN_MEM_HOST) == 0
    ids32 = np.ascontiguousarray(ids, np.int32)
    x = np.empty((100, XDIM), np.float32)
    assert lib.fnn_gather(h, ptr(ids32), 100, ptr(x), _capi.FNN_MEM_HOST) == 0
    rows64 = rows.astype(np.float64)
    assert np.array_equal(x, orc.gather(rows64, ids, -3.0).astype(np.float32))
    pd = np.empty(100, np.float32); gx = np.empty((100, XDIM), np.float32); loss = C.c_float()
    assert lib.fnn_train_step(h, ptr(ids32), ptr(y), 100, ptr(r1), ptr(r2), 100, ptr(pd), ptr(gx),
                              _capi.FNN_MEM_HOST, C.byref(loss)) == 0
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.train_step(p64, rows64, -3.0, ids, y.astype(np.float64), r1.astype(float), r2.astype(float),
                         0.01, 0.0, 0.1)
    np.testing.assert_allclose(pd, ref['p_drop'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gx, ref['gx'], rtol=2e-3, atol=2e-5 * np.abs(ref['gx']).max())
    assert abs(loss.value - ref['loss']) <= 2e-5 * ref['loss']
    pr = np.empty(100, np.float32)
    assert lib.fnn_predict(h, ptr(ids32), 100, ptr(pr), _capi.FNN_MEM_HOST) == 0
    np.testing.assert_allclose(pr, orc.predict(p64, orc.gather(rows64, ids, -3.0)), rtol=1e-4, atol=1e-6)
    tab = np.empty_like(rows)
    assert lib.fnn_get_table(h, ptr(tab), _capi.FNN_MEM_HOST) == 0
    np.testing.assert_allclose(tab, rows64, rtol=1e-5, atol=2e-7)
    assert lib.fnn_destroy(h) == 0


def test_fnn_script_on_demo_matches_golden_epochs(built, golden_dir, tmp_path, monkeypatch):
    """`python FNN.py` semantics end to end (BASELINE configs[0] on the demo set): parse the text
    files, init from seed 1234, masks from RandomStreams(234), 2 epochs; AUC and logloss within 1e-4
    of the oracle's epochs."""
    import importlib.util
    from deep_ctr_amd import dl_utils
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('DEEPCTR_DATA_DIR', os.path.join(golden_dir, 'demo'))
    monkeypatch.setenv('DEEPCTR_EPOCHS', '2')
    monkeypatch.setattr(dl_utils, 'log_path', str(tmp_path / 'log'))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fnn_script', os.path.join(root, 'deep-ctr_amd', 'FNN.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.run(['FNN.py'])
    g = np.load(os.path.join(golden_dir, 'epoch.npz'))
    assert len(hist) == 2
    for i, hrec in enumerate(hist):
        assert abs(hrec['test_auc'] - g['test_auc'][i]) <= 1e-4
        assert abs(hrec['test_logloss'] - g['test_logloss'][i]) <= 1e-4
    log = (tmp_path / 'log' / 'fm2997.txt').read_text()
    assert 'Test Err:0' in log and 'Minimal test error is' in log
    assert (tmp_path / 'mlp3fm_test_2997.p').exists()


def test_out_of_range_id_is_an_error(built):
    rows, fo, ids, y, p, r1, r2 = make_problem(8)
    eng = make_engine(rows, fo, p)
    bad = ids.copy(); bad[3, 2] = rows.shape[0] + 5
    with pytest.raises(FNNError) as ei:
        eng.train_step(bad, y, r1, r2)
    assert ei.value.code == _capi.FNN_ERR_RANGE
    eng.train_step(ids, y, r1, r2)              # the handle stays usable
    with pytest.raises(FNNError):
        eng.train_step(np.zeros((5000, F), np.int32), np.zeros(5000, np.float32), r1, r2)   # B > max_batch
    eng.close()


def test_calls_before_setup_fail_loudly(built):
    eng = FNNEngine(F, K, H1, H2, max_batch=64, precision='f32')
    with pytest.raises(FNNError) as ei:
        eng.predict(np.zeros((4, F), np.int32))
    assert ei.value.code == _capi.FNN_ERR_STATE
    eng.close()


# ----------------------------------------------------------------- full BASELINE shape (config 2)
@pytest.fixture(scope="module")
def full_problem():
    B = 4096
    return make_problem(B, n_rows=synth.IPINYOU_DIMS, seed=1234)


def test_full_shape_step_f32_vs_oracle(built, full_problem):
    """16 fields, 937,670 one-hot dims, k=10, batch 4096: one f32 step against the oracle."""
    rows, fo, ids, y, p, r1, r2 = full_problem
    eng = make_engine(rows, fo, p, lr=0.001)
    out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
    rows64 = rows.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.train_step(p64, rows64, -3.0, ids, y.astype(np.float64), r1.astype(float), r2.astype(float),
                         0.001, 0.0, 0.1)
    np.testing.assert_allclose(out['p'].cpu().numpy(), ref['p_drop'], rtol=1e-4, atol=1e-6)
    gs = np.abs(ref['gx']).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), ref['gx'], rtol=2e-3, atol=2e-5 * gs)
    touched = np.unique(ids)
    np.testing.assert_allclose(eng.get_rows(touched), rows64[touched], rtol=1e-5, atol=2e-7)
    assert abs(out['loss'] - ref['loss']) <= 2e-5 * ref['loss']
    eng.close()


def test_full_shape_step_bf16_vs_oracle(built, full_problem):
    """The HEADLINE precision at the headline shape (BASELINE configs[1]: 937,670 rows, batch 4096, bf16) against the float64
    oracle, at the tolerances of test_train_step_bf16_vs_oracle: p_drop within 2e-2, gx within 5 % of its scale, the loss within
    2 %, every touched row within 5 % of the largest row update, untouched rows bit for bit (round-2 review: this shape had
    property checks only in bf16)."""
    rows, fo, ids, y, p, r1, r2 = full_problem
    eng = make_engine(rows, fo, p, prec='bf16', lr=0.001)
    out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
    rows64 = rows.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.train_step(p64, rows64, -3.0, ids, y.astype(np.float64), r1.astype(float), r2.astype(float), 0.001, 0.0, 0.1)
    assert np.abs(out['p'].cpu().numpy() - ref['p_drop']).max() < 2e-2
    gs = np.abs(ref['gx']).max()
    assert np.abs(out['gx'].cpu().numpy() - ref['gx']).max() < 5e-2 * gs
    assert abs(out['loss'] - ref['loss']) < 2e-2 * ref['loss']
    touched = np.unique(ids)
    upd = np.abs(rows64[touched] - rows[touched].astype(np.float64)).max()
    assert np.abs(eng.get_rows(touched) - rows64[touched]).max() < 5e-2 * upd + 1e-6
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):                    # the dense SGD step: each tensor's update within 5 % of its size
        scale = np.abs(p64[k] - p[k]).max() + 1e-12
        assert np.abs(d[k] - p64[k]).max() <= 5e-2 * scale + 1e-7, k
    un = np.setdiff1d(np.random.RandomState(3).randint(0, rows.shape[0], 20000), touched)
    assert np.array_equal(eng.get_rows(un), rows[un])
    eng.close()


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
def test_full_shape_properties(built, full_problem, prec):
    """Size-independent properties at the full shape:
    (1) untouched rows are bit-identical after a step (decay only on touched rows, :299-306);
    (2) the dense gradient of a batch equals the sum of the gradients of its two halves (the loss
        is a SUM, python/FNN_wnzh.py:173) -- the identity data parallelism relies on;
    (3) two identical runs are bitwise reproducible (integer-atomic scatter, fixed-order reductions).
    """
    rows, fo, ids, y, p, r1, r2 = full_problem
    B = ids.shape[0]

    def run(sel):
        eng = make_engine(rows, fo, p, prec=prec, lr=0.001)
        bucket = eng.step_begin(ids[sel], y[sel], r1, r2, b_size=B)
        eng.step_end()                      # leaves the bucket untouched
        eng.sync()
        g = bucket.cpu().numpy().astype(np.float64)
        tab = eng.get_table()
        eng.close()
        return g, tab

    g_all, tab_all = run(slice(0, B))
    g_a, _ = run(slice(0, B // 2))
    g_b, _ = run(slice(B // 2, B))
    scale = np.abs(g_all).max()
    tol = 1e-5 if prec == 'f32' else 1e-5      # products are identical per example in both modes
    assert np.abs(g_all - (g_a + g_b)).max() <= tol * scale
    mask = np.ones(rows.shape[0], bool); mask[np.unique(ids)] = False
    assert np.array_equal(tab_all[mask], rows[mask])
    g_again, tab_again = run(slice(0, B))
    assert np.array_equal(g_all, g_again) and np.array_equal(tab_all, tab_again)


# ----------------------------------------------------------------- SNN fine-tune path (A8)
def make_snn_problem(B, n_rows=600, h0=200, seed=0, dup_col=None, empty=()):
    sizes = synth.field_sizes_tiny(n_rows)
    rng = np.random.RandomState(seed)
    ww0 = (rng.standard_normal((sum(sizes), h0)) * 0.1).astype(np.float32)
    bb0 = (rng.standard_normal(h0) * 0.1).astype(np.float32)
    ids = synth.zipf_ids(B, sizes, 1.1, seed + 1)
    if dup_col is not None:
        ids[:, dup_col] = ids[0, dup_col]
    for (t, f) in empty:
        ids[t, f] = -1
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    p = {'w1': f32r(rng.uniform(-0.3, 0.3, (h0, H1))), 'b1': f32r(rng.uniform(-0.1, 0.1, H1)),
         'w2': f32r(rng.uniform(-0.3, 0.3, (H1, H2))), 'b2': f32r(rng.uniform(-0.1, 0.1, H2)),
         'w3': f32r(rng.uniform(-0.2, 0.2, H2)), 'b3': 0.05}
    r1 = (rng.uniform(size=H1) < 0.9).astype(np.uint8)
    r2 = (rng.uniform(size=H2) < 0.9).astype(np.uint8)
    return ww0, bb0, ids, y, p, r1, r2


def make_snn_engine(ww0, bb0, p, prec='f32', lr=0.01, lam1=0.001, h0=200, max_batch=4096):
    eng = FNNEngine(F, 0, H1, H2, max_batch=max_batch, precision=prec, lr=lr, lambda1=lam1, lambda_fm=0.0,
                    reg_all=True, mode='bag', hidden0=h0)
    eng.set_table(ww0, np.zeros(ww0.shape[0], np.int32), 0.0)
    eng.set_bag_bias(bb0)
    eng.set_dense(p)
    return eng


@pytest.mark.parametrize("B,kw", [(1, {}), (37, {"empty": [(3, 2)]}), (300, {"dup_col": 6}), (1000, {}),
                                  # hidden0 = 300, the reference's default for every advertiser but 2997 (python/SNN_RBM.py:25): bag rows
                                  # padded to 320 floats, the CX = 5 instance of the strip kernel; 256 and 316 are that instance's edges
                                  (1, {"h0": 300}), (300, {"h0": 300, "dup_col": 6, "empty": [(5, 0)]}), (1000, {"h0": 300}),
                                  (130, {"h0": 256}), (130, {"h0": 316}), (130, {"h0": 252}), (130, {"h0": 192})])
def test_snn_step_f32_vs_oracle(built, B, kw):
    """A8: bag + sigmoid gather, MLP with L2 on all six tensors, per-example row updates without
    decay (python/SNN_RBM.py:238-291) against the float64 oracle."""
    kw = dict(kw)
    h0 = kw.pop("h0", 200)
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(B, seed=B, h0=h0, **kw)
    eng = make_snn_engine(ww0, bb0, p, h0=h0)
    x = eng.gather(ids).cpu().numpy()
    ww64, bb64 = ww0.astype(np.float64), bb0.astype(np.float64)
    np.testing.assert_allclose(x, orc.snn_bag(ww64, bb64, ids), rtol=2e-6, atol=1e-7)
    out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.snn_train_step(p64, ww64, bb64, ids, y.astype(np.float64), r1.astype(float), r2.astype(float),
                             0.01, 0.001)
    np.testing.assert_allclose(out['p'].cpu().numpy(), ref['p_drop'], rtol=2e-4, atol=1e-6)
    gs = np.abs(ref['gx']).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), ref['gx'], rtol=2e-3, atol=2e-5 * gs + 1e-9)
    assert abs(out['loss'] - ref['loss']) <= 2e-5 * max(1.0, abs(ref['loss']))
    upd = np.abs(ww64 - ww0).max() + 1e-12
    assert np.abs(eng.get_table() - ww64).max() <= 1e-3 * upd + 2e-7
    bupd = np.abs(bb64 - bb0).max() + 1e-12
    assert np.abs(eng.get_bag_bias() - bb64).max() <= 1e-3 * bupd + 2e-7
    d = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        scale = np.abs(p64[k] - p[k]).max() + 1e-12
        assert np.abs(d[k] - p64[k]).max() <= 1e-3 * scale + 1e-7, k
    pr = eng.predict(ids).cpu().numpy()
    np.testing.assert_allclose(pr, orc.snn_predict(p64, ww64, bb64, ids), rtol=3e-4, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("prefetch", [False, True])
def test_snn_same_row_in_several_columns(built, prefetch, tmp_path):
    """python/SNN_RBM.py:248-253 lists a line's ACTIVE features in line order, so a feature's column depends on the line: a
    token with value != 1 shifts everything behind it, and a feature may be listed twice.  The update of a row is grouped per
    column; rows that several columns of a batch hold must still receive every occurrence's delta (round-1 advisor finding:
    lost updates).  Lines go through the native SNN reader; two steps, the second one grouped ahead (fnn_prefetch_ids)."""
    import torch
    from deep_ctr_amd import ingest
    rng = np.random.RandomState(17)
    B, n_rows, h0 = 600, 400, 200
    lines = []
    for t in range(2 * B):
        feats = list(rng.randint(0, n_rows, size=16))
        if t % 3 == 0:
            feats[5] = feats[1]                                   # the same feature twice on the line
        vals = [1] * 16
        for j in rng.choice(16, size=rng.randint(0, 4), replace=False):
            vals[j] = int(rng.choice([0, 2]))                     # inactive: everything behind it moves one column left
        lines.append('%d %s' % (rng.randint(0, 2), ' '.join('%d:%d' % (f, v) for f, v in zip(feats, vals))))
    path = tmp_path / 'train.fm.txt'
    path.write_text('\n'.join(lines) + '\n')
    ids, _, yi = ingest.parse_examples(str(path), ingest.MODE_SNN_ACTIVE, None, 16)
    # the same row does sit in different columns, across examples and inside one
    cols = {}
    for t in range(B):
        for f in range(16):
            if ids[t, f] >= 0:
                cols.setdefault(int(ids[t, f]), set()).add(f)
    assert sum(len(c) > 1 for c in cols.values()) > 100
    ww0, bb0, _, _, p, r1, r2 = make_snn_problem(8, n_rows=n_rows, seed=3, h0=h0)
    ww0 = ww0[:n_rows].copy()
    y = yi.astype(np.float32)
    eng = make_snn_engine(ww0, bb0, p, h0=h0)
    ww64, bb64 = ww0.astype(np.float64), bb0.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    dev_ids = [torch.as_tensor(np.ascontiguousarray(ids[s * B:(s + 1) * B])).cuda() for s in range(2)]
    for s in range(2):
        sl = slice(s * B, (s + 1) * B)
        if prefetch and s == 0:
            eng.prefetch_ids(dev_ids[1])
        out = eng.train_step(dev_ids[s], y[sl], r1, r2)
        ref = orc.snn_train_step(p64, ww64, bb64, ids[sl], y[sl].astype(np.float64), r1.astype(float), r2.astype(float), 0.01, 0.001)
        assert abs(out['loss'] - ref['loss']) <= 5e-5 * max(1.0, abs(ref['loss']))
    upd = np.abs(ww64 - ww0).max()
    assert np.abs(eng.get_table() - ww64).max() <= 1e-3 * upd + 3e-7
    assert np.abs(eng.get_bag_bias() - bb64).max() <= 1e-3 * np.abs(bb64 - bb0).max() + 3e-7
    eng.close()


def test_virtual_two_rank_dp_bag_mode(built):
    """the SNN fine-tune shards like the FNN step.  Two bag-mode engines stand for two ranks (fnn_step_begin on
    each half, buckets summed as the all-reduce would, fnn_step_end): dense tensors and the bag bias equal the
    single-engine full-batch step; rows only one half touches equal the full-batch result on that rank."""
    import torch
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(512, seed=9, dup_col=4)
    full = make_snn_engine(ww0, bb0, p)
    full.train_step(ids, y, r1, r2)
    ref_dense, ref_rows, ref_bb = full.get_dense(), full.get_table(), full.get_bag_bias()
    full.close()
    ranks = [make_snn_engine(ww0, bb0, p) for _ in range(2)]
    halves = [slice(0, 256), slice(256, 512)]
    buckets = [e.step_begin(ids[h], y[h], r1, r2, b_size=512) for e, h in zip(ranks, halves)]
    for e in ranks:
        e.sync()
    tot = buckets[0] + buckets[1]
    for b in buckets:
        b.copy_(tot)
    torch.cuda.synchronize()
    for e in ranks:
        e.step_end()
        e.sync()
    for e in ranks:
        d = e.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(d[k] - p[k].astype(np.float32)).max() + 1e-12
            assert np.abs(d[k] - ref_dense[k]).max() <= 5e-4 * scale + 1e-7, k
        bscale = np.abs(ref_bb - bb0).max() + 1e-12
        assert np.abs(e.get_bag_bias() - ref_bb).max() <= 5e-4 * bscale + 1e-7
    t0, t1 = set(np.unique(ids[halves[0]])), set(np.unique(ids[halves[1]]))
    only0 = np.array(sorted(t0 - t1)); only1 = np.array(sorted(t1 - t0))
    np.testing.assert_allclose(ranks[0].get_table()[only0], ref_rows[only0], rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(ranks[1].get_table()[only1], ref_rows[only1], rtol=1e-5, atol=2e-7)
    for e in ranks:
        e.close()


@pytest.mark.parametrize("h0", [200, 300])
def test_snn_step_bf16_tracks_oracle(built, h0):
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(512, seed=5, dup_col=3, h0=h0)
    eng = make_snn_engine(ww0, bb0, p, prec='bf16', h0=h0)
    out = eng.train_step(ids, y, r1, r2, want_p=True)
    ww64, bb64 = ww0.astype(np.float64), bb0.astype(np.float64)
    ref = orc.snn_train_step(p, ww64, bb64, ids, y.astype(np.float64), r1.astype(float), r2.astype(float), 0.01, 0.001)
    assert np.abs(out['p'].cpu().numpy() - ref['p_drop']).max() < 3e-2
    assert abs(out['loss'] - ref['loss']) < 3e-2 * ref['loss']
    upd = np.abs(ww64 - ww0).max()
    assert np.abs(eng.get_table() - ww64).max() < 8e-2 * upd + 1e-6
    eng.close()


def test_snn_full_shape_reproducible(built):
    """BASELINE config 5 shape for the fine-tune step: 937,670 x 200 table (750 MB), batch 4096;
    two runs are bitwise identical and untouched rows keep their bits."""
    h0, B = 200, 4096
    sizes = synth.field_sizes_ipinyou()
    rng = np.random.RandomState(0)
    ww0 = np.random.default_rng(0).standard_normal((sum(sizes), h0), dtype=np.float32) * np.float32(0.05)
    bb0 = np.zeros(h0, np.float32)
    ids = synth.zipf_ids(B, sizes, 1.1, 1)
    y = (rng.uniform(size=B) < 0.02).astype(np.float32)
    _, _, _, _, p, r1, r2 = make_snn_problem(4, seed=1)
    res = []
    for _ in range(2):
        eng = make_snn_engine(ww0, bb0, p, prec='bf16', lr=0.001, lam1=0.0)
        eng.train_step(ids, y, r1, r2)
        touched = np.unique(ids)
        res.append((eng.get_rows(touched), eng.get_bag_bias(), eng.get_rows(np.array([5, 77777, 500000]))))
        eng.close()
    nd = int((res[0][0] != res[1][0]).sum())
    assert nd == 0, "touched rows differ between two runs: %d values in %d rows, max |d| %.3e" % (
        nd, int((res[0][0] != res[1][0]).any(axis=1).sum()), float(np.abs(res[0][0] - res[1][0]).max()))
    assert np.array_equal(res[0][1], res[1][1]), "bag bias differs between two runs: max |d| %.3e" % float(np.abs(res[0][1] - res[1][1]).max())
    untouched = np.setdiff1d(np.array([5, 77777, 500000]), np.unique(ids))
    for i, r in enumerate([5, 77777, 500000]):
        if r in untouched:
            assert np.array_equal(res[0][2][i], ww0[r])
    assert not np.array_equal(res[0][0], ww0[np.unique(ids)])


# ------------------------------------------------------------------ A10: evaluation pass on the device
def test_eval_metrics_equal_sklearn(built):
    """fnn_eval: predictions of 9,001 examples (three max_batch chunks, ragged tail, many exact ties
    from repeated examples) -> AUC / RMSE / logloss on the device against sklearn (the metric
    oracle of python/FNN_wnzh.py:219-220, python/baseline.py:427-429) on the same float32 predictions."""
    from sklearn.metrics import log_loss, mean_squared_error, roc_auc_score
    rows, fo, ids, y, p, r1, r2 = make_problem(3000, seed=77)
    p['w3'] = np.random.RandomState(5).uniform(-0.5, 0.5, H2)
    ids = np.concatenate([ids, ids, ids, ids[:1]])                       # every example three times: tie groups
    rng = np.random.RandomState(6)
    yy = (rng.uniform(size=len(ids)) < 0.3).astype(np.int32)
    eng = make_engine(rows, fo, p, max_batch=4096)
    m = eng.evaluate(ids, yy, want_p=True)
    pp = m['p'].cpu().numpy()
    np.testing.assert_array_equal(pp, eng.predict(ids).cpu().numpy())
    p64 = pp.astype(np.float64)
    assert abs(m['auc'] - roc_auc_score(yy, p64)) < 1e-12
    assert abs(m['rmse'] - np.sqrt(mean_squared_error(yy, p64))) < 1e-12
    assert abs(m['logloss'] - log_loss(yy, p64, labels=[0, 1])) < 1e-12
    m2 = eng.evaluate(ids, yy)
    assert (m2['auc'], m2['rmse'], m2['logloss']) == (m['auc'], m['rmse'], m['logloss'])    # bitwise reproducible
    from deep_ctr_amd.engine import FNNError
    with pytest.raises(FNNError):                                           # roc_auc_score: ValueError
        eng.evaluate(ids[:100], np.zeros(100, np.int32))
    eng.close()


def test_train_step_with_64bit_sort_keys(built):
    """A table of 1,100,000 rows makes n_rows * 4096 exceed 2^32, so the grouping runs on 64-bit
    (row << 32 | t) keys: two steps (the second with an announced batch, i.e. the run sort + rank merge
    roles inside the step launches) against the oracle, with duplicates and rows near the top of the range."""
    B = 700
    rows, fo, ids, y, p, r1, r2 = make_problem(2 * B, n_rows=1100000, seed=23)
    top = rows.shape[0] - 1
    ids[5, 15] = top; ids[9, 15] = top; ids[B + 3, 15] = top                # the very last row, repeated
    eng = make_engine(rows, fo, p, lr=0.01, lam1=0.0, lamfm=0.1)
    import torch
    ids_d = torch.as_tensor(ids).to(eng.device).contiguous()
    rows64 = rows.astype(np.float64).copy()
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    for step in range(2):
        sl = slice(step * B, (step + 1) * B)
        if step == 0:
            eng.prefetch_ids(ids_d[B:2 * B])
        eng.train_step(ids_d[sl], y[sl], r1, r2, want_loss=False)
        x = orc.gather(rows64, ids[sl], -3.0)
        gx, _, _, _, _ = orc.train_call(p64, x, y[sl].astype(np.float64), r1.astype(np.float64), r2.astype(np.float64), 0.01, 0.0)
        orc.scatter_sgd(rows64, ids[sl], gx, 0.01, 0.1)
    touched = np.unique(ids[ids >= 0])
    change = np.abs(rows64[touched] - rows[touched]).max()
    assert np.abs(eng.get_rows(touched) - rows64[touched]).max() <= 2e-3 * change + 2e-7      # two f32 steps
    assert eng.lib.fnn_sync(eng.h) == 0
    eng.close()


def test_small_max_batch_handle(built):
    """A handle created for a small max_batch (as __graft_entry__.smoke does) still groups 4096 slots
    per field in the three-launch path: its grouping buffers must be sized for that, not for max_batch
    (an out-of-bounds write here once faulted the GPU)."""
    rows, fo, ids, y, p, r1, r2 = make_problem(100, seed=5, dup_col=2)
    eng = make_engine(rows, fo, p, max_batch=128, lr=0.01, lam1=0.0, lamfm=0.1)
    _check_step(eng, rows, ids, y, p, r1, r2, 0.01, 0.0, 0.1)
    eng.close()


@pytest.mark.parametrize("B,max_batch", [(5000, 8192), (9000, 16384)])
def test_train_step_above_4096_examples(built, B, max_batch):
    """B > 4096 leaves the three-launch path for the layer-by-layer kernels and the per-field LDS sort
    of 8,192 / 16,384 keys (k_sort<8> / <16>): one step against the oracle (vectorised A6, itself
    checked against the sequential loop in tests/test_oracle.py)."""
    rows, fo, ids, y, p, r1, r2 = make_problem(B, n_rows=3000, seed=B, dup_col=6)
    eng = make_engine(rows, fo, p, max_batch=max_batch, lr=0.001, lam1=0.0, lamfm=0.1)
    out = eng.train_step(ids, y, r1, r2, want_p=True)
    rows64 = rows.astype(np.float64).copy()
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    x = orc.gather(rows64, ids, -3.0)
    gx, _, loss, p_drop, g = orc.train_call(p64, x, y.astype(np.float64), r1.astype(np.float64), r2.astype(np.float64), 0.001, 0.0)
    orc.scatter_sgd_vec(rows64, ids, gx, 0.001, 0.1)
    np.testing.assert_allclose(out['p'].cpu().numpy(), p_drop, rtol=1e-4, atol=1e-6)
    assert abs(out['loss'] - loss) <= 5e-5 * abs(loss)
    touched = np.unique(ids)
    change = np.abs(rows64[touched] - rows[touched]).max()
    assert np.abs(eng.get_rows(touched) - rows64[touched]).max() <= 2e-3 * change + 2e-7
    d = eng.get_dense()
    for k in ('w1', 'w2', 'w3'):
        gs = 0.001 * np.abs(g[k]).max()
        np.testing.assert_allclose(d[k], p64[k], rtol=1e-5, atol=2e-3 * gs + 1e-7, err_msg=k)
    eng.close()


def test_gather_beyond_max_batch(built):
    """fnn_gather takes any number of examples (the reference's evaluation pass gathers 100,000 lines at a time,
    python/FNN_wnzh.py:193-209): one launch for device pointers, max_batch-sized chunks through the staging buffers for host pointers."""
    rows, fo, ids, y, p, r1, r2 = make_problem(1000, seed=4, empty=[(0, 3), (999, 15)])
    eng = make_engine(rows, fo, p, max_batch=256)
    ref = orc.gather(rows.astype(np.float64), ids, -3.0).astype(np.float32)
    assert np.array_equal(eng.gather(ids).cpu().numpy(), ref)                      # device pointers: 1000 > max_batch = 256
    import ctypes as C
    x = np.empty((1000, XDIM), np.float32)
    ids32 = np.ascontiguousarray(ids, np.int32)
    rc = eng.lib.fnn_gather(eng.h, ids32.ctypes.data_as(C.c_void_p), 1000, x.ctypes.data_as(C.c_void_p), _capi.FNN_MEM_HOST)
    assert rc == 0 and np.array_equal(x, ref)                                      # host pointers: four chunks
    eng.close()


def test_train_epoch_equals_the_step_loop(built):
    """FNNEngine.train_epoch (resident arrays, raw C calls, masks drawn ahead, shadowed features per batch) leaves bit for bit the
    state of the per-step loop it replaces in FNN.py -- full batches, a short last batch, and a start in the middle of the epoch."""
    rows, fo, ids, y, p, r1, r2 = make_problem(1030, seed=44, dup_col=5)
    rng = np.random.RandomState(3)
    M1 = (rng.uniform(size=(11, H1)) < 0.5).astype(np.uint8)
    M2 = (rng.uniform(size=(11, H2)) < 0.5).astype(np.uint8)
    sh = np.array([[5, fo[7], 7], [5, fo[411], 411], [250, fo[2], 2], [1029, fo[900], 900]], np.int32)   # (example, field OF THE ROW, row), sorted by example
    bad = make_engine(rows, fo, p)
    with pytest.raises(FNNError, match='does not belong'):          # a row under another field would be updated by two groups at once
        bad.set_shadowed(np.array([[5, (fo[7] + 1) % 16, 7]], np.int32))
    bad.close()
    a, b = make_engine(rows, fo, p, lr=0.01, lam1=0.02), make_engine(rows, fo, p, lr=0.01, lam1=0.02)
    for j in range(11):                                            # 10 batches of 100 and one of 30
        lo, hi = j * 100, min(1030, (j + 1) * 100)
        part = sh[(sh[:, 0] >= lo) & (sh[:, 0] < hi)].copy()
        if len(part):
            part[:, 0] -= lo
            a.set_shadowed(part)
        a.train_step(ids[lo:hi], y[lo:hi], M1[j], M2[j], b_size=hi - lo, want_loss=False)
    ids_d, y_d = b.to_device(ids, y.astype(np.int32))
    yf = y_d.float()
    b.train_epoch(ids_d, yf, 100, M1, M2, 0, 4, sh)                 # in two pieces, as FNN.py reads the dense state before the last batch
    b.train_epoch(ids_d, yf, 100, M1, M2, 4, None, sh)
    da, db = a.get_dense(), b.get_dense()
    bad = {k: float(np.abs(np.asarray(da[k], np.float64) - np.asarray(db[k], np.float64)).max()) for k in da if not np.array_equal(da[k], db[k])}
    ta, tb = a.get_table(), b.get_table()
    assert not bad and np.array_equal(ta, tb), "step loop vs train_epoch differ: dense %r, table rows %r" % (
        bad, np.unique(np.argwhere(ta != tb)[:, 0])[:10].tolist())
    a.close(); b.close()


@pytest.mark.parametrize("waves", [None, '4'])
@pytest.mark.parametrize("mode,prec", [('fm', 'bf16'), ('fm', 'f32'), ('fm', 'bf16x3'), ('bag', 'bf16')])
def test_write_through_stores_change_no_bit(built, monkeypatch, mode, prec, waves):
    """The strip kernel's training outputs leave by write-through stores (gx' regrouped into whole lines), FNN_WT_STORES=0 keeps
    plain stores: the same values either way, so 30 back-to-back steps (no host synchronisation between them) must leave the
    table, the dense tensors and the bag bias bit-equal -- a store the next launch did not see in time would show here."""
    steps, B = 30, 700
    res = []
    if waves is not None:
        monkeypatch.setenv('FNN_STEP1_WAVES', waves)          # the four-wave form of the strip kernel (eight is the default)
    for wt in ('0', None):
        if wt is None:
            monkeypatch.delenv('FNN_WT_STORES', raising=False)
        else:
            monkeypatch.setenv('FNN_WT_STORES', wt)
        if mode == 'fm':
            rows, fo, ids, y, p, r1, r2 = make_problem(steps * B, seed=5, dup_col=6)
            eng = make_engine(rows, fo, p, prec=prec, lr=0.01, lam1=0.001, lamfm=0.1)
        else:
            ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(steps * B, seed=6, dup_col=4)
            eng = make_snn_engine(ww0, bb0, p, prec=prec)
        for s in range(steps):
            eng.train_step(ids[s * B:(s + 1) * B], y[s * B:(s + 1) * B], r1, r2, want_loss=False)
        d = eng.get_dense()
        res.append([eng.get_table()] + [np.asarray(d[k]) for k in sorted(d)] + ([eng.get_bag_bias()] if mode == 'bag' else []))
        eng.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
