"""GPU parity of the SNN pre-training kernels (A7 sparse online CD-1, A7' dense CD-1) against
oracle/rbm_oracle.py, through the C ABI of include/rbm_hip.h.  f32 arithmetic vs the float64
oracle, sequential updates: tolerances are relative to the size of the parameter CHANGE."""
import numpy as np
import pytest

from oracle import rbm_oracle as ro

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import dl_utils, synth
from deep_ctr_amd import sampling_based_gaussian_binary_rbm_sparse as gbrbm

pytestmark = pytest.mark.gpu


def make_lines(tmp_path, n=240, n_rows=300, seed=3):
    """16 features per line with ids 2*row+1, so id-1 never collides with a feature (32 visibles)."""
    sizes = synth.field_sizes_tiny(n_rows)
    ids = synth.zipf_ids(n, sizes, 1.1, seed)
    feats = 2 * ids + 1
    path = tmp_path / 'train.fm.txt'
    with open(path, 'w') as f:
        for t in range(n):
            f.write('0 ' + ' '.join('%d:1' % v for v in feats[t]) + '\n')
    return str(path), [list(map(int, feats[t])) for t in range(n)], 2 * sum(sizes) + 2


def rel_change_err(got, ref, init):
    return np.abs(got - ref).max() / (np.abs(ref - init).max() + 1e-30)


def test_get_rbm_weights_matches_oracle(built, tmp_path):
    path, lines_feats, x_dim = make_lines(tmp_path)
    arr = [x_dim, 40, 24, 12]
    dl_utils.seed_global(1234)
    res = gbrbm.get_rbm_weights(path, arr, ncases=len(lines_feats), batch_size=100000)
    rng = np.random.RandomState(1234)
    ref = ro.get_rbm_weights(lines_feats, arr, rng, batch_size=100000)
    # initial values (for the size of the change): same stream
    rng0 = np.random.RandomState(1234)
    p0 = rng0.uniform(-.1, .1, x_dim * 40 + x_dim + 40)
    W0_init = p0[:x_dim * 40].reshape(x_dim, 40)
    assert res[0].shape == (x_dim, 40) and res[2].shape == (40, 24) and res[4].shape == (24, 12)
    assert rel_change_err(res[0], ref[0], W0_init) < 2e-3          # sparse layer: 3 epochs online, f32
    np.testing.assert_allclose(res[1], ref[1], rtol=0, atol=2e-3 * np.abs(ref[1] - p0[-40:]).max() + 1e-7)
    for k in (2, 3, 4, 5):                                         # dense layers
        np.testing.assert_allclose(res[k], ref[k], rtol=2e-4, atol=2e-6)


def test_dense_cd1_minibatches_and_bf16(built, tmp_path):
    """Two mini-batches per epoch (the short last batch ends the epoch, :286-287) in f32; bf16 runs."""
    path, lines_feats, x_dim = make_lines(tmp_path, n=250)
    arr = [x_dim, 32, 20]
    dl_utils.seed_global(7)
    res = gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=200)
    ref = ro.get_rbm_weights(lines_feats, arr, np.random.RandomState(7), batch_size=200)
    np.testing.assert_allclose(res[2], ref[2], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(res[3], ref[3], rtol=2e-4, atol=2e-6)
    dl_utils.seed_global(7)
    resb = gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=200, precision='bf16')
    assert np.abs(resb[2] - ref[2]).max() < 5e-4
    with pytest.raises(ValueError):
        gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=125)       # empty last batch in the reference


def test_sparse_needs_32_visibles(built, tmp_path):
    p = tmp_path / 'bad.txt'
    p.write_text('0 5:1 6:1 9:1\n')
    with pytest.raises(ValueError):
        gbrbm.sparse_inputs(gbrbm.parse_lines(str(p)))


def test_snn_rbm_script_on_demo_tracks_oracle(built, golden_dir, tmp_path, monkeypatch):
    """`python SNN_RBM.py` end to end on the demo set (BASELINE configs[4] semantics): layer-wise
    CD-1 pre-training (A7, A7') then 2 fine-tune epochs (A8), against the same flow on the float64
    oracles with the reference's RNG consumption order."""
    import importlib.util
    import os
    from oracle import fnn_oracle as orc
    from sklearn.metrics import log_loss, roc_auc_score
    demo = os.path.join(golden_dir, 'demo')
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('DEEPCTR_DATA_DIR', demo)
    monkeypatch.setenv('DEEPCTR_EPOCHS', '2')
    monkeypatch.setenv('DEEPCTR_XDIM', 'auto')
    monkeypatch.setattr(dl_utils, 'log_path', str(tmp_path / 'log'))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('snn_script', os.path.join(root, 'deep-ctr_amd', 'SNN_RBM.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.run(['SNN_RBM.py'])
    assert (tmp_path / 'rbm_2997_.p').exists() and len(hist) == 2

    # the same flow on the oracles
    tr_ids, tr_y = mod.load_active_ids(os.path.join(demo, 'train.fm.txt'))
    te_ids, te_y = mod.load_active_ids(os.path.join(demo, 'test.fm.txt'))
    x_dim = int(max(tr_ids.max(), te_ids.max())) + 1
    H0, H1, H2 = 200, 300, 100
    rng = np.random.RandomState(1234)
    for (a, b) in ((x_dim, H0), (H0, H1), (H1, H2)):          # init_weight x3 consume the stream (SNN_RBM.py:78-80)
        rng.uniform(low=-1, high=1, size=(a, b))
    lines = [[int(v) for v in row if v >= 0] for row in tr_ids]
    ww0, bb0, ww1, bb1, ww2, bb2 = ro.get_rbm_weights(lines, [x_dim, H0, H1, H2], rng, batch_size=100000)
    p = {'w1': ww1.copy(), 'b1': bb1.copy(), 'w2': ww2.copy(), 'b2': bb2.copy(), 'w3': np.zeros(H2), 'b3': 0.0}
    ms = orc.TheanoMaskStream(H1, H2, 0.98, has_r0=False)
    ref = []
    for ep in range(2):
        r1, r2 = ms.next()
        orc.snn_train_step(p, ww0, bb0, tr_ids[:1000], tr_y[:1000].astype(np.float64), r1, r2, 0.001, 0.0)
        pte = orc.snn_predict(p, ww0, bb0, te_ids)
        ref.append((roc_auc_score(te_y, pte), log_loss(te_y, pte, labels=[0, 1])))
    for hrec, (auc, ll) in zip(hist, ref):
        print("SNN demo: auc %.6f vs %.6f, logloss %.6f vs %.6f" % (hrec['test_auc'], auc, hrec['test_logloss'], ll))
        assert abs(hrec['test_auc'] - auc) <= 2e-3
        assert abs(hrec['test_logloss'] - ll) <= 2e-4
