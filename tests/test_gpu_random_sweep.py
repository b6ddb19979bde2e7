"""Seeded random sweep of the FNN train step against the float64 oracle: batch sizes around every boundary the kernels have (strips
of 16, padding to 256, sort runs of 256, the 4096 cap), table sizes from eighty rows (every row hit hundreds of times: long
multi-chunk segments) to thousands, empty fields, a whole column on one row, activations, hyper-parameters, explicit b_size, and the
two precisions that carry parity claims (f32, bf16 pairs).  Each case is one step checked tensor by tensor (_check_step)."""
import numpy as np
import pytest

import deep_ctr_amd  # noqa: F401
from test_gpu_parity import _check_step, make_engine, make_problem

pytestmark = pytest.mark.gpu

BATCHES = [1, 2, 15, 16, 17, 31, 255, 256, 257, 511, 513, 1000, 2047, 2049, 4095, 4096]


@pytest.mark.parametrize("seed", list(range(32)))
def test_random_step_vs_oracle(built, seed):
    rng = np.random.RandomState(1000 + seed)
    B = BATCHES[seed % len(BATCHES)] if seed < 2 * len(BATCHES) - 8 else int(rng.randint(1, 4097))
    n_rows = int(rng.choice([80, 120, 1000, 5000]))
    kw = {}
    if rng.uniform() < 0.3:
        kw['dup_col'] = int(rng.randint(0, 16))
    if rng.uniform() < 0.5:
        kw['empty'] = [(int(rng.randint(0, B)), int(rng.randint(0, 16))) for _ in range(int(rng.randint(1, 6)))]
    acti = str(rng.choice(['tanh', 'sigmoid', 'linear']))
    lr, lam1, lamfm = float(rng.choice([0.001, 0.01])), float(rng.choice([0.0, 0.02])), float(rng.choice([0.0, 0.1, 0.5]))
    b_size = int(rng.choice([0, 0, B + 7]))
    prec = 'bf16x3' if seed % 2 else 'f32'
    rows, fo, ids, y, p, r1, r2 = make_problem(B, n_rows=n_rows, seed=seed, **kw)
    eng = make_engine(rows, fo, p, prec=prec, lr=lr, lam1=lam1, lamfm=lamfm, acti=acti)
    tol = 8.0 if prec == 'bf16x3' else 1.0
    try:
        _check_step(eng, rows, ids, y, p, r1, r2, lr, lam1, lamfm, b_size=b_size, tol=tol, acti=acti, tol_table=40.0 if prec == 'bf16x3' else 4.0)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_snn_step_vs_oracle(built, seed):
    """The same for the SNN fine-tune step (bag rows of hidden0 floats, python/SNN_RBM.py:238-291): batch sizes at the boundaries,
    hidden0 over the range the strip kernel takes (192..316), small tables (rows hit many times), empty slots."""
    from oracle import fnn_oracle as orc
    from test_gpu_parity import make_snn_engine, make_snn_problem
    rng = np.random.RandomState(2000 + seed)
    B = int(rng.choice([1, 15, 17, 255, 257, 1000, 2049, 4095, 4096]))
    h0 = int(rng.choice([192, 200, 200, 252, 256, 300, 300, 316]))
    kw = {}
    if rng.uniform() < 0.3:
        kw['dup_col'] = int(rng.randint(0, 16))
    if rng.uniform() < 0.5:
        kw['empty'] = [(int(rng.randint(0, B)), int(rng.randint(0, 16))) for _ in range(int(rng.randint(1, 5)))]
    prec = 'bf16x3' if seed % 2 else 'f32'
    tol = 8.0 if prec == 'bf16x3' else 1.0
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(B, n_rows=int(rng.choice([100, 600, 3000])), seed=seed, h0=h0, **kw)
    eng = make_snn_engine(ww0, bb0, p, prec=prec, h0=h0)
    try:
        ww64, bb64 = ww0.astype(np.float64), bb0.astype(np.float64)
        out = eng.train_step(ids, y, r1, r2, want_p=True, want_gx=True)
        p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
        ref = orc.snn_train_step(p64, ww64, bb64, ids, y.astype(np.float64), r1.astype(float), r2.astype(float), 0.01, 0.001)
        np.testing.assert_allclose(out['p'].cpu().numpy(), ref['p_drop'], rtol=2e-4 * tol, atol=1e-6 * tol)
        gs = np.abs(ref['gx']).max()
        np.testing.assert_allclose(out['gx'].cpu().numpy(), ref['gx'], rtol=2e-3 * tol, atol=2e-5 * gs * tol + 1e-9)
        assert abs(out['loss'] - ref['loss']) <= 2e-5 * tol * max(1.0, abs(ref['loss']))
        upd = np.abs(ww64 - ww0).max() + 1e-12
        assert np.abs(eng.get_table() - ww64).max() <= 1e-3 * tol * upd + 2e-7
        bupd = np.abs(bb64 - bb0).max() + 1e-12
        assert np.abs(eng.get_bag_bias() - bb64).max() <= 1e-3 * tol * bupd + 2e-7
        d = eng.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(p64[k] - p[k]).max() + 1e-12
            assert np.abs(d[k] - p64[k]).max() <= 1e-3 * tol * scale + 1e-7, k
    finally:
        eng.close()
