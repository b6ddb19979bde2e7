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
