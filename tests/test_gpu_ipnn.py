"""GPU parity of the inner-product FNN family (A9) against oracle/ipnn_oracle.py through the C
ABI of include/ipnn_hip.h.  f32 mode: logits rtol 1e-4, parameter changes within 1e-3 of their size."""
import numpy as np
import pytest

from oracle import ipnn_oracle as io

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import synth
from deep_ctr_amd.ipnn import FNN_IP_L3, IPNNEngine

pytestmark = pytest.mark.gpu
F, K = 16, 11


def f32r(a):
    return np.asarray(a, np.float32).astype(np.float64)


def problem(B, hidden, seed=0, n_rows=600, scale=0.3):
    rng = np.random.RandomState(seed)
    sizes = synth.field_sizes_tiny(n_rows)
    table = f32r(rng.standard_normal((sum(sizes), K)) * 0.2)
    ids = synth.zipf_ids(B, sizes, 1.1, seed + 1)
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    d = [F * K + F * (F - 1) // 2 + 1] + list(hidden) + [1]
    params = {'b': float(np.float32(0.1)), 'W': [f32r(rng.uniform(-scale, scale, (d[i], d[i + 1]))) for i in range(len(d) - 1)],
              'bias': [f32r(rng.uniform(-0.1, 0.1, d[i + 1])) for i in range(len(d) - 1)]}
    masks = [(rng.uniform(size=(B, d[t])) < 0.7).astype(np.uint8) for t in range(len(hidden) + 1)]
    return table, ids, y, params, masks, d


@pytest.mark.parametrize("act,B,hidden,drop", [('relu', 50, [40, 24, 12], True), ('tanh', 130, [40, 24, 12], True),
                                               ('sigmoid', 33, [30, 20], False), ('relu', 300, [70, 60, 50, 40, 30, 20, 10], True),
                                               # wide layers: several 64-column blocks per wave in the strip kernels, many
                                               # 128 x 128 tiles and K splits in the grouped weight-gradient launch
                                               ('relu', 600, [1000, 300, 130], True), ('tanh', 257, [700, 520], False)])
def test_ipnn_step_f32_vs_oracle(built, act, B, hidden, drop):
    table, ids, y, params, masks, d = problem(B, hidden, seed=B, scale=0.3 if max(hidden) < 200 else 0.05)
    keep = 0.7 if drop else 1.0
    eng = IPNNEngine(F, K, hidden, act, max_batch=max(512, B), precision='f32', lr=0.01, keep_prob=keep)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    pr = eng.predict(ids).cpu().numpy()
    np.testing.assert_allclose(pr, io.predict(params, table, ids, act), rtol=2e-4, atol=1e-6)
    out = eng.train_step(ids, y, masks if drop else None, want_logits=True)
    p0 = {'b': params['b'], 'W': [w.copy() for w in params['W']], 'bias': [b.copy() for b in params['bias']]}
    t0 = table.copy()
    m64 = [m.astype(np.float64) for m in masks] if drop else None
    loss, logits, g = io.sgd_step(params, table, ids, y, act, 0.01, m64, keep)
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=2e-4, atol=2e-5)
    assert abs(out['loss'] - loss) <= 5e-5 * max(1.0, abs(loss))
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - p0['W'][t]).max() + 1e-12
        assert np.abs(Ws[t] - params['W'][t]).max() <= 2e-3 * cw + 2e-7, ('W', t)
        cb = np.abs(params['bias'][t] - p0['bias'][t]).max() + 1e-12
        assert np.abs(bs[t] - params['bias'][t]).max() <= 2e-3 * cb + 2e-7, ('b', t)
    assert abs(b - params['b']) <= 2e-3 * abs(params['b'] - p0['b']) + 2e-7
    touched = np.unique(ids)
    ct = np.abs(table - t0).max() + 1e-12
    assert np.abs(eng.get_rows(touched) - table[touched]).max() <= 2e-3 * ct + 2e-7
    eng.close()


def test_ipnn_bf16_and_family_class(built, tmp_path):
    table, ids, y, params, masks, d = problem(256, [400, 400, 200], seed=3, scale=0.05)
    m = FNN_IP_L3(None, None, 256, [table.shape[0], F, K - 1, 400, 400, 200, 'relu'], ['uniform', -0.01, 0.01, [1, 2, 3, 4, 5, 6], None],
                  ['sgd', 0.01, 'sum'], [0.5], 'train', 0, precision='bf16')
    m.eng.set_params(table, params['b'], params['W'], params['bias'])
    pr = m.forward(ids).cpu().numpy()
    assert np.abs(pr - io.predict(params, table, ids, 'relu')).max() < 3e-2
    r = m.train_step(ids, y, [mk for mk in masks])
    assert np.isfinite(r['loss'])
    m.dump(str(tmp_path / 'm.pickle'))
    import pickle
    vm = pickle.load(open(tmp_path / 'm.pickle', 'rb'))
    assert set(vm) == {'W', 'V', 'b', 'h1_w', 'h1_b', 'h2_w', 'h2_b', 'h3_w', 'h3_b', 'h4_w', 'h4_b'}
    assert vm['V'].shape == (table.shape[0], K - 1) and vm['h1_w'].shape == (297, 400)


def test_plain_fnn_class_without_pair_products(built, tmp_path):
    """The reference's plain TensorFlow `FNN` class (python/FNN.py): z1 = [e | b], two hidden layers,
    activation and inverted dropout before every matmul -- the same kernels with the pair products
    switched off (ipnn_cfg.pairs = 0), against the oracle with USE_PAIRS = False."""
    import pickle
    from deep_ctr_amd.ipnn import FNN
    rng = np.random.RandomState(8)
    sizes = synth.field_sizes_tiny(600)
    table = f32r(rng.standard_normal((sum(sizes), K)) * 0.2)
    B, hidden = 200, [48, 20]
    ids = synth.zipf_ids(B, sizes, 1.1, 9)
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    d = [F * K + 1] + hidden + [1]
    params = {'b': float(np.float32(0.1)), 'W': [f32r(rng.uniform(-0.3, 0.3, (d[i], d[i + 1]))) for i in range(3)],
              'bias': [f32r(rng.uniform(-0.1, 0.1, d[i + 1])) for i in range(3)]}
    masks = [(rng.uniform(size=(B, d[t])) < 0.8).astype(np.uint8) for t in range(3)]
    m = FNN(None, None, B, [table.shape[0], F, K - 1] + hidden + ['tanh'], ['uniform', -0.01, 0.01, [1, 2, 3, 4, 5], None],
            ['sgd', 0.01, 'sum'], [0.8], 'train', 0, precision='f32')
    assert m.eng.d == d
    m.eng.set_params(table, params['b'], params['W'], params['bias'])
    io.USE_PAIRS = False
    try:
        np.testing.assert_allclose(m.forward(ids).cpu().numpy(), io.predict(params, table, ids, 'tanh'), rtol=2e-4, atol=1e-6)
        out = m.eng.train_step(ids, y, masks, want_logits=True)
        t0, W0 = table.copy(), [w.copy() for w in params['W']]
        loss, logits, g = io.sgd_step(params, table, ids, y, 'tanh', 0.01, [mk.astype(np.float64) for mk in masks], 0.8)
    finally:
        io.USE_PAIRS = True
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=2e-4, atol=2e-5)
    assert abs(out['loss'] - loss) <= 5e-5 * max(1.0, abs(loss))
    b, Ws, bs = m.eng.get_params()
    for t in range(3):
        cw = np.abs(params['W'][t] - W0[t]).max() + 1e-12
        assert np.abs(Ws[t] - params['W'][t]).max() <= 2e-3 * cw + 2e-7
    touched = np.unique(ids)
    assert np.abs(m.eng.get_rows(touched) - table[touched]).max() <= 2e-3 * np.abs(table - t0).max() + 2e-7
    m.dump(str(tmp_path / 'fnn.pickle'))
    vm = pickle.load(open(tmp_path / 'fnn.pickle', 'rb'))
    assert set(vm) == {'W', 'V', 'b', 'h1_w', 'h1_b', 'h2_w', 'h2_b', 'h3_w', 'h3_b'} and vm['h1_w'].shape == (F * K + 1, 48)


@pytest.mark.parametrize("opt", ['sgd', 'adam'])
def test_mean_reduction_vs_oracle(built, opt):
    """`_ptmzr_argv[-1]` other than 'sum' is tf.reduce_mean (python/FNN_IP_L7.py:83-86): every gradient of the step is
    1 / B of the summed one -- under Adam that is NOT a learning-rate change, so it is checked there too."""
    B, hidden = 96, [40, 24, 12]
    table, ids, y, params, masks, d = problem(B, hidden, seed=77)
    lr = 0.5 if opt == 'sgd' else 1e-3
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=256, precision='f32', lr=lr, keep_prob=0.7, optimizer=opt, reduce='mean')
    eng.set_params(table, params['b'], params['W'], params['bias'])
    st = io.adam_state(params, table)
    t0, W0 = table.copy(), [w.copy() for w in params['W']]
    m64 = [m.astype(np.float64) for m in masks]
    for step in range(2):
        out = eng.train_step(ids, y, masks, want_logits=True)
        if opt == 'sgd':
            loss, logits, _ = io.sgd_step(params, table, ids, y, 'relu', lr, m64, 0.7, reduce='mean')
        else:
            loss, logits, _ = io.adam_step(params, table, ids, y, 'relu', lr, st, m64, 0.7, reduce='mean')
        np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=5e-4, atol=5e-5)
        assert abs(out['loss'] - loss) <= 5e-5 * max(1.0, abs(loss))          # the mean, as the reference's `loss`
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - W0[t]).max()
        assert np.abs(Ws[t] - params['W'][t]).max() <= 5e-3 * cw + 1e-7, t
    touched = np.unique(ids)
    ct = np.abs(table - t0).max()
    assert np.abs(eng.get_rows(touched) - table[touched]).max() <= 5e-3 * ct + 1e-7
    eng.close()


def test_family_class_honours_reduce_and_refuses_numeric_fields(built):
    sizes = synth.field_sizes_tiny(600)
    net = FNN_IP_L3(sizes, np.cumsum([0] + sizes[:-1]), 32, [sum(sizes), F, K - 1, 24, 16, 8, 'relu'], ['uniform', -0.01, 0.01, [1, 2, 3], None],
                    ['sgd', 0.1, 'mean'], [0.5], precision='f32')
    assert net.eng.reduce == 'mean'
    ids = synth.zipf_ids(8, sizes, 1.1, 3)
    net.forward(ids)
    with pytest.raises(NotImplementedError):
        net.forward(ids, v_wts=np.ones((8, 13), np.float32))
    net.eng.close()


def test_adam_steps_vs_oracle(built):
    """Adam as TensorFlow applies it to this family (python/baseline.py:146): three steps, dense moment
    decay of the whole table included (rows no batch touched move too), against oracle.adam_step."""
    table, ids, y, params, masks, d = problem(160, [40, 24, 12], seed=21)
    eng = IPNNEngine(F, K, [40, 24, 12], 'relu', max_batch=256, precision='f32', lr=1e-3, keep_prob=0.7, optimizer='adam',
                     adam_eps=1e-8)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    st = io.adam_state(params, table)
    t0, W0 = table.copy(), [w.copy() for w in params['W']]
    m64 = [m.astype(np.float64) for m in masks]
    for step in range(3):
        out = eng.train_step(ids, y, masks, want_logits=True)
        loss, logits, _ = io.adam_step(params, table, ids, y, 'relu', 1e-3, st, m64, 0.7)
        np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=5e-4, atol=5e-5)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - W0[t]).max()
        assert np.abs(Ws[t] - params['W'][t]).max() <= 5e-3 * cw + 1e-7, t
    rows = eng.get_rows(np.arange(table.shape[0]))
    ct = np.abs(table - t0).max()
    assert np.abs(rows - table).max() <= 5e-3 * ct + 1e-7
    untouched = np.setdiff1d(np.arange(table.shape[0]), np.unique(ids))
    assert len(untouched) > 0 and np.array_equal(rows[untouched], t0[untouched].astype(np.float32))     # zero gradient, zero moments: no move
    eng.close()


def test_ftrl_steps_vs_oracle(built):
    """FTRL as python/tf_util.py:21-24 builds it (TensorFlow defaults: accumulator 0.1, power -0.5, no l1/l2): three
    steps against oracle.ftrl_step; every variable is re-derived from its linear term, so rows of the table that no
    example touched are exactly 0 after the first step (dense table gradient), as in the reference's graph."""
    table, ids, y, params, masks, d = problem(160, [40, 24, 12], seed=22)
    eng = IPNNEngine(F, K, [40, 24, 12], 'relu', max_batch=256, precision='f32', lr=1e-2, keep_prob=0.7, optimizer='ftrl')
    eng.set_params(table, params['b'], params['W'], params['bias'])
    st = io.ftrl_state(params, table)
    m64 = [m.astype(np.float64) for m in masks]
    for step in range(3):
        out = eng.train_step(ids, y, masks, want_logits=True)
        loss, logits, _ = io.ftrl_step(params, table, ids, y, 'relu', 1e-2, st, m64, 0.7)
        np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=2e-3, atol=2e-5)
        assert abs(out['loss'] - loss) <= 1e-4 * abs(loss)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        assert np.abs(Ws[t] - params['W'][t]).max() <= 5e-3 * np.abs(params['W'][t]).max() + 1e-7, t
        assert np.abs(bs[t] - params['bias'][t]).max() <= 5e-3 * np.abs(params['bias'][t]).max() + 1e-7, t
    assert abs(b - params['b']) <= 5e-3 * abs(params['b']) + 1e-7
    rows = eng.get_rows(np.arange(table.shape[0]))
    assert np.abs(rows - table).max() <= 5e-3 * np.abs(table).max() + 1e-7
    untouched = np.setdiff1d(np.arange(table.shape[0]), np.unique(ids))
    assert len(untouched) > 0 and not rows[untouched].any()
    eng.close()


def test_ipnn_bf16_wide_stack_tracks_oracle(built):
    """bf16 compute on BASELINE-sized layers (1000/800/600/400, batch 1024): logits, loss and the direction of every
    weight update follow the float64 oracle within bf16 rounding (tolerances: logits 5e-2 absolute; update cosine 0.99)."""
    hidden = [1000, 800, 600, 400]
    table, ids, y, params, masks, d = problem(1024, hidden, seed=11, scale=0.04)
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=1024, precision='bf16', lr=0.01, keep_prob=0.7)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks, want_logits=True)
    p0 = [w.copy() for w in params['W']]
    loss, logits, g = io.sgd_step(params, table, ids, y, 'relu', 0.01, [m.astype(np.float64) for m in masks], 0.7)
    assert np.abs(out['logits'].cpu().numpy() - logits).max() < 5e-2
    assert abs(out['loss'] - loss) <= 2e-2 * abs(loss)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        du, dv = (Ws[t] - p0[t]).ravel(), (params['W'][t] - p0[t]).ravel()
        cos = float(du @ dv / (np.linalg.norm(du) * np.linalg.norm(dv) + 1e-30))
        assert cos > 0.99, (t, cos)
    eng.close()


@pytest.mark.parametrize("seed", range(8))
def test_ipnn_random_stacks_f32_vs_oracle(built, seed):
    """Seeded random stacks (1-8 hidden layers, widths 1..1100, batch 1..700, any activation, with and without dropout):
    one f32 SGD step through the strip kernels and the grouped weight-gradient launch against the float64 oracle."""
    rng = np.random.RandomState(1000 + seed)
    L = int(rng.randint(1, 9))
    hidden = [int(rng.choice([1, 7, 50, 63, 64, 65, 130, 300, 520, 1023, 1100])) for _ in range(L)]
    if sum(hidden) > 2600:                                   # keep the float64 oracle quick
        hidden = [min(h, 300) for h in hidden]
    B = int(rng.randint(1, 701))
    act = ['relu', 'tanh', 'sigmoid'][int(rng.randint(3))]
    drop = bool(rng.randint(2))
    table, ids, y, params, masks, d = problem(B, hidden, seed=seed, scale=0.05 if max(hidden) >= 200 else 0.3)
    keep = 0.6 if drop else 1.0
    eng = IPNNEngine(F, K, hidden, act, max_batch=max(256, B), precision='f32', lr=0.01, keep_prob=keep)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks if drop else None, want_logits=True)
    p0 = [w.copy() for w in params['W']]
    t0 = table.copy()
    loss, logits, g = io.sgd_step(params, table, ids, y, act, 0.01, [m.astype(np.float64) for m in masks] if drop else None, keep)
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=5e-4, atol=5e-5)
    assert abs(out['loss'] - loss) <= 1e-4 * max(1.0, abs(loss))
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - p0[t]).max() + 1e-12
        assert np.abs(Ws[t] - params['W'][t]).max() <= 3e-3 * cw + 3e-7, (t, hidden, B, act, drop)
    touched = np.unique(ids)
    ct = np.abs(table - t0).max() + 1e-12
    assert np.abs(eng.get_rows(touched) - table[touched]).max() <= 3e-3 * ct + 3e-7
    eng.close()


def test_ipnn_many_steps_track_oracle(built):
    """Twelve consecutive SGD steps with fresh dropout masks and different batches (side-stream work of one step overlaps
    the next step's start): parameters and touched rows still follow the float64 oracle."""
    hidden = [130, 70, 40]
    table, ids0, y0, params, masks0, d = problem(200, hidden, seed=31, scale=0.1)
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=256, precision='f32', lr=0.02, keep_prob=0.7)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    p0 = [w.copy() for w in params['W']]
    t0 = table.copy()
    rng = np.random.RandomState(77)
    sizes = synth.field_sizes_tiny(600)
    touched = set()
    for step in range(12):
        B = int(rng.randint(60, 201))
        ids = synth.zipf_ids(B, sizes, 1.1, 100 + step)
        y = (rng.uniform(size=B) < 0.3).astype(np.float64)
        masks = [(rng.uniform(size=(B, d[t])) < 0.7).astype(np.uint8) for t in range(len(hidden) + 1)]
        out = eng.train_step(ids, y, masks, want_logits=(step == 11))
        loss, logits, _ = io.sgd_step(params, table, ids, y, 'relu', 0.02, [m.astype(np.float64) for m in masks], 0.7)
        touched |= set(np.unique(ids).tolist())
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=2e-3, atol=2e-4)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - p0[t]).max()
        assert np.abs(Ws[t] - params['W'][t]).max() <= 5e-3 * cw + 1e-6, t
    tr = np.array(sorted(touched))
    ct = np.abs(table - t0).max()
    assert np.abs(eng.get_rows(tr) - table[tr]).max() <= 5e-3 * ct + 1e-6
    eng.close()


def test_ipnn_l7_benchmark_shape_bf16(built):
    """The benchmark configuration itself (BASELINE configs[2]: hidden 1000/800/600/400/200/100/50, batch 4096, bf16,
    keep_prob 0.5): one step against the float64 oracle -- logits within 5e-2, loss within 2 %, every weight update
    pointing the oracle's way (cosine > 0.98; bf16 operands, f32 accumulation)."""
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    B = 4096
    table, ids, y, params, masks, d = problem(B, hidden, seed=5, n_rows=3000, scale=0.06)
    masks = [(np.random.RandomState(40 + t).uniform(size=(B, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='bf16', lr=1e-3, keep_prob=0.5)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks, want_logits=True)
    p0 = [w.copy() for w in params['W']]
    loss, logits, g = io.sgd_step(params, table, ids, y, 'relu', 1e-3, [m.astype(np.float64) for m in masks], 0.5)
    assert np.abs(out['logits'].cpu().numpy() - logits).max() < 5e-2
    assert abs(out['loss'] - loss) <= 2e-2 * abs(loss)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        du, dv = (Ws[t] - p0[t]).ravel(), (params['W'][t] - p0[t]).ravel()
        cos = float(du @ dv / (np.linalg.norm(du) * np.linalg.norm(dv) + 1e-30))
        assert cos > 0.98, (t, cos)
    eng.close()


@pytest.mark.parametrize("env", [{'IPNN_STRIP': '0', 'IPNN_GROUP_WGRAD': '0', 'IPNN_SIDE_STREAM': '0'},
                                 {'IPNN_STRIP': '0', 'IPNN_GEMM_LDS': '1'}, {'IPNN_MASK_SIDE': '1', 'IPNN_GROUP_XCD': '0', 'IPNN_STRIP_ROT': '0'}])
def test_ipnn_alternative_paths_match_oracle(built, monkeypatch, env):
    """The paths the A/B knobs select (one GEMM launch per product -- k_gemm_ft, or the LDS-staged k_gemm_lds at batch
    4096 --, one launch per weight-gradient product, everything in line on one stream, the side-stream mask
    transposition, unrotated / launch-order tiles) compute the same step: f32 against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    hidden = [1000, 130]
    B = 4096 if 'IPNN_GEMM_LDS' in env else 300
    table, ids, y, params, masks, d = problem(B, hidden, seed=12, scale=0.05)
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='f32', lr=0.01, keep_prob=0.7)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks, want_logits=True)
    p0 = [w.copy() for w in params['W']]
    loss, logits, g = io.sgd_step(params, table, ids, y, 'relu', 0.01, [m.astype(np.float64) for m in masks], 0.7)
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=5e-4, atol=5e-5)
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - p0[t]).max() + 1e-12
        assert np.abs(Ws[t] - params['W'][t]).max() <= 3e-3 * cw + 3e-7, (t, env)
    eng.close()


def test_strip_pairs_are_bit_identical_to_single_strips(built, monkeypatch):
    """StripDuo (two workgroups per 32-example strip, swapping halves of every wide activation tile through write-through
    stores and a flag) against one workgroup per strip, bf16, the FNN_IP_L7 stack at batch 4096: five train steps with
    dropout.  The pair splits OUTPUT COLUMNS only -- every value is the same sum in the same order -- so logits, every dense
    tensor, b and every touched table row must agree BIT FOR BIT (round-2 advisor: the hand-rolled swap had only loose bf16
    tolerances behind it).  A stale or torn block in the swap shows here at once."""
    import torch
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    B, steps = 4096, 5
    table, ids, y, params, masks, d = problem(B * steps, hidden, seed=77, n_rows=3000, scale=0.05)
    masks = [(np.random.RandomState(5 + t).uniform(size=(B * steps, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    res = []
    for duo in ('1', '0'):
        monkeypatch.setenv('IPNN_STRIP_DUO', duo)
        eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='bf16', lr=0.01, keep_prob=0.5)
        eng.set_params(table, params['b'], params['W'], params['bias'])
        logits = []
        for s in range(steps):
            sl = slice(s * B, (s + 1) * B)
            out = eng.train_step(ids[sl], y[sl], [m[sl] for m in masks], want_logits=True)
            logits.append(out['logits'].cpu().numpy().copy())
        b, Ws, bs = eng.get_params()
        res.append((np.concatenate(logits), b, Ws, bs, eng.get_rows(np.unique(ids))))
        eng.close()
    (la, ba, Wa, bsa, ra), (lb, bb, Wb, bsb, rb) = res
    assert np.isfinite(la).all() and np.abs(la).max() > 0
    assert np.array_equal(la, lb) and ba == bb and np.array_equal(ra, rb)
    for t in range(len(Wa)):
        assert np.array_equal(Wa[t], Wb[t]) and np.array_equal(bsa[t], bsb[t]), t


@pytest.mark.parametrize("env", [{'IPNN_TAIL_NF': '4', 'IPNN_TAIL_NW': '8'}, {'IPNN_TAIL_NF': '2'}, {'IPNN_TAIL_NF': '1', 'IPNN_TAIL_NW': '8'},
                                 {'IPNN_TAIL_FUSE': '0'}, {'IPNN_TAIL_SPLIT': '0'}, {'IPNN_UPDATE_SIDE': '1'}, {'IPNN_IPF_NT': '512'},
                                 {'IPNN_STRIP_WARM': '0'}, {'IPNN_WT': '0'}, {'IPNN_WIDE': '4:8'}, {'IPNN_WIDE': '2:12'}])
def test_launch_forms_of_the_step_are_bit_identical(built, monkeypatch, env):
    """The forms the FNN_IP_L7 step can be launched in -- the narrow tail's items of 4 / 2 / 1 fragments on 8 / 16 waves, both
    passes' tails in one launch or two, no tail split at all, whole blocks or half blocks as the wide launches' items (8 or 12 waves), the dense update on the side stream with its join deferred to
    the next call, 512 / 1024 threads per workgroup in the gather, plain instead of write-through stores of the launches' outputs (a
    write-through store the next launch did not see in time would show here) -- only regroup the same sums: three train steps with dropout
    must leave logits, every dense tensor, b and the touched table rows BIT-equal to the default form's.  (The deferred update
    is also read back between steps here: get_params joins it.)"""
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    B, steps = 4096, 3
    table, ids, y, params, masks, d = problem(B * steps, hidden, seed=78, n_rows=3000, scale=0.05)
    masks = [(np.random.RandomState(9 + t).uniform(size=(B * steps, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    res = []
    for form in (None, env):
        for k, v in (form or {}).items():
            monkeypatch.setenv(k, v)
        eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='bf16', lr=0.01, keep_prob=0.5)
        eng.set_params(table, params['b'], params['W'], params['bias'])
        logits, mids = [], []
        for s in range(steps):
            sl = slice(s * B, (s + 1) * B)
            out = eng.train_step(ids[sl], y[sl], [m[sl] for m in masks], want_logits=True)
            logits.append(out['logits'].cpu().numpy().copy())
            if s == 0:
                mids.append(eng.get_params()[1][0].copy())                # the first layer's weights after ONE step
        b, Ws, bs = eng.get_params()
        res.append((np.concatenate(logits), b, Ws, bs, eng.get_rows(np.unique(ids)), mids[0]))
        eng.close()
    (la, ba, Wa, bsa, ra, ma), (lb, bb, Wb, bsb, rb, mb) = res
    assert np.isfinite(la).all() and np.abs(la).max() > 0
    assert np.array_equal(la, lb) and ba == bb and np.array_equal(ra, rb) and np.array_equal(ma, mb)
    for t in range(len(Wa)):
        assert np.array_equal(Wa[t], Wb[t]) and np.array_equal(bsa[t], bsb[t]), t


@pytest.mark.parametrize("hidden,B", [([640, 500, 300, 70], 1000), ([400, 400, 200], 300), ([1000, 800, 600, 400, 200], 2048),
                                      ([900, 130, 700, 60], 777)])
def test_tail_split_forms_on_other_stacks(built, monkeypatch, hidden, B):
    """Other stacks and batch lengths than FNN_IP_L7's (a different cut between wide and narrow products, one narrow product
    only, a narrow layer BETWEEN wide ones -- then no suffix is cut off but the trailing one --, ragged last strips): the default
    form (tail of 16-example strips, one fragment per wave on 16 waves, both passes in one launch) against round 2's single
    launch per pass (IPNN_TAIL_SPLIT=0), bit for bit over two steps with dropout."""
    steps = 2
    table, ids, y, params, masks, d = problem(B * steps, hidden, seed=B, n_rows=2000, scale=0.05)
    masks = [(np.random.RandomState(3 + t).uniform(size=(B * steps, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    res = []
    for split in ('1', '0'):
        monkeypatch.setenv('IPNN_TAIL_SPLIT', split)
        eng = IPNNEngine(F, K, hidden, 'tanh', max_batch=B, precision='bf16', lr=0.01, keep_prob=0.5)
        eng.set_params(table, params['b'], params['W'], params['bias'])
        logits = []
        for s in range(steps):
            sl = slice(s * B, (s + 1) * B)
            out = eng.train_step(ids[sl], y[sl], [m[sl] for m in masks], want_logits=True)
            logits.append(out['logits'].cpu().numpy().copy())
        pr = eng.predict(ids[:B]).cpu().numpy().copy()
        b, Ws, bs = eng.get_params()
        res.append((np.concatenate(logits), b, Ws, bs, eng.get_rows(np.unique(ids)), pr))
        eng.close()
    (la, ba, Wa, bsa, ra, pa), (lb, bb, Wb, bsb, rb, pb) = res
    assert np.isfinite(la).all() and np.abs(la).max() > 0
    assert np.array_equal(la, lb) and ba == bb and np.array_equal(ra, rb) and np.array_equal(pa, pb)
    for t in range(len(Wa)):
        assert np.array_equal(Wa[t], Wb[t]) and np.array_equal(bsa[t], bsb[t]), t
