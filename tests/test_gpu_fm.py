"""GPU parity of factorisation-machine pre-training (row N3, python/FM.py) against
oracle/fm_oracle.py through the C ABI of include/fm_hip.h, and the FM -> fm.model.txt -> FNN loop.
f32 vs float64: tolerances relative to the size of the parameter change."""
import os

import numpy as np
import pytest

from oracle import fm_oracle as fo

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import synth
from deep_ctr_amd.FM import FM

pytestmark = pytest.mark.gpu
F, RANK = 16, 10


def f32r(a):
    return np.asarray(a, np.float32).astype(np.float64)


def problem(B, n_rows=500, seed=0):
    rng = np.random.RandomState(seed)
    sizes = synth.field_sizes_tiny(n_rows)
    rows = f32r(rng.standard_normal((sum(sizes), RANK + 1)) * 0.2)
    ids = synth.zipf_ids(B, sizes, 1.1, seed + 1)
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    return sizes, rows, ids, y


@pytest.mark.parametrize("B,reduce,lam", [(1, 'mean', 1e-2), (64, 'sum', 0.0), (700, 'mean', 0.05), (4096, 'sum', 1e-3)])
def test_fm_step_vs_oracle(built, B, reduce, lam):
    sizes, rows, ids, y = problem(B, seed=B)
    if B > 8:
        ids[3, 5] = -1                                            # an absent field
    m = FM(B, [rows.shape[0], F, RANK], ['uniform', -0.001, 0.001, [1, 2], None], ['sgd', 0.05] + ([reduce] if reduce == 'sum' else []),
           [lam], 'train', 0)
    m.set_params(rows, 0.1)
    np.testing.assert_allclose(m.forward(ids).cpu().numpy(), fo.predict(rows, 0.1, ids), rtol=2e-5, atol=1e-6)
    r = rows.copy()
    b = 0.1
    for _ in range(3):                                            # three steps: the lazy decay scale is live
        out = m.train_step(ids, y, want_p=True)
        b, data, p = fo.sgd_step(r, b, ids, y, 0.05, lam, reduce == 'mean')
        np.testing.assert_allclose(out['p'].cpu().numpy(), p, rtol=5e-5, atol=1e-6)
        assert abs(out['loss'] - data) <= 2e-5 * max(1.0, abs(data))
    got, gb = m.get_params()
    change = np.abs(r - rows).max() + 1e-12
    assert np.abs(got - r).max() <= 2e-3 * change + 2e-7
    assert abs(gb - b) <= 2e-3 * abs(b - 0.1) + 2e-7
    m.close()


def test_fm_long_run_folds_the_decay_scale(built):
    """lr * lambda = 0.5 halves the scale every step: after 30 steps it has been folded back into
    the rows at least once; untouched rows must equal rows * 0.5^30-ish exactly as the oracle's dense decay."""
    sizes, rows, ids, y = problem(32, seed=9)
    m = FM(32, [rows.shape[0], F, RANK], ['uniform', -0.001, 0.001, [1, 2], None], ['sgd', 0.5], [1.0], 'train', 0)
    m.set_params(rows, 0.0)
    r, b = rows.copy(), 0.0
    for _ in range(30):
        m.train_step(ids, y, want_loss=False)
        b, _, _ = fo.sgd_step(r, b, ids, y, 0.5, 1.0, True)
    got, gb = m.get_params()
    np.testing.assert_allclose(got, r, rtol=2e-3, atol=1e-9)
    m.close()


def test_fm_pretrain_feeds_the_fnn_script_formats(built, tmp_path):
    """FM pre-training -> write_fm_model -> DataFM parses it back bit-exactly (the text format of
    python/FNN_wnzh.py:68-84), and dump() keeps the reference's var_map keys."""
    import pickle
    from deep_ctr_amd.data_fm import DataFM
    sizes, rows, ids, y = problem(256, seed=4)
    fo_row = synth.field_of_row(sizes)
    m = FM(256, [rows.shape[0], F, RANK], ['uniform', -0.001, 0.001, [0x3210, 0x7654], None], ['sgd', 1e-3], [1e-2], 'train', 0)
    for _ in range(5):
        m.train_step(ids, y, want_loss=False)
    names = sorted(DataFM.name_field, key=DataFM.name_field.get)
    path = str(tmp_path / 'fm.model.txt')
    m.write_fm_model(path, fo_row, names)
    d = DataFM(path)
    got, b = m.get_params()
    assert d.k == RANK + 1 and d.w_0 == np.float64(np.float32(b))
    assert np.array_equal(d.rows.astype(np.float32), got) and np.array_equal(d.field_of_row, fo_row)
    m.dump(str(tmp_path / 'fm.pickle'))
    vm = pickle.load(open(tmp_path / 'fm.pickle', 'rb'))
    assert set(vm) == {'W', 'V', 'b'} and vm['V'].shape == (rows.shape[0], RANK)
    m.close()
