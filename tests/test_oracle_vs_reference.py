"""The oracle and the host code against outputs of the REFERENCE ITSELF (tests/golden/ref_run.npz, written by
tests/golden/make_golden_ref.py in the build container: the reference's plain-Python/NumPy modules -- data_fm.py,
dl_utils.py, ipinyou.py, sampling_based_gaussian_binary_rbm_sparse.py -- executed through lib2to3; see that script's
header for exactly how).  This is what pins rows A1, A2, A3, A7, A7', A11 and A12 of SURVEY section 8: for those the
oracle is no longer "parity unpinned".  The Theano / TensorFlow rows cannot be run and stay unpinned.  CPU only."""
import os

import numpy as np
import pytest

from oracle import fnn_oracle as orc
from oracle import ingest_oracle as io
from oracle import rbm_oracle as ro

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import dl_utils, ingest
from deep_ctr_amd import ipinyou as ipy
from deep_ctr_amd.data_fm import DataFM

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO = os.path.join(HERE, 'golden', 'demo')


@pytest.fixture(scope='module')
def ref():
    return np.load(os.path.join(HERE, 'golden', 'ref_run.npz'))


def unpad(row):
    return [int(v) for v in row if v >= 0]


# ------------------------------------------------------------------ A1: the FM-model parser
def test_fm_model_parsers_equal_reference_DataFM(ref, built):
    path = os.path.join(DEMO, 'fm.model.txt')
    feats = [int(f) for f in ref['fm_feats']]
    # the two Python restatements
    w0, k, xdim, fw, ff = orc.parse_fm_model(path)
    assert (w0, k, xdim) == (float(ref['fm_w0']), int(ref['fm_k']), int(ref['fm_xdim']))
    assert sorted(fw) == feats
    assert np.array_equal(np.array([fw[f] for f in feats]), ref['fm_weights'])
    assert np.array_equal(np.array([ff[f] for f in feats]), ref['fm_fields'])
    w0b, kb, fwb, ffb = io.parse_fm_model(path)
    assert (w0b, kb) == (w0, k) and fwb == fw and ffb == ff
    assert list(fw)[:50] == [int(v) for v in ref['fm_first_key_order']]            # dict order = file order
    # the host class and the native parser behind it (ctr_fm_model_load)
    d = DataFM(path)
    assert (d.w_0, d.k, d.xdim) == (w0, k, xdim) and d.feat_field == ff
    assert all(list(d.feat_weights[f]) == fw[f] for f in feats)
    rows, feat_ids, fo = d.model.arrays()
    assert [int(v) for v in feat_ids[:50]] == [int(v) for v in ref['fm_first_key_order']]
    order = np.argsort(feat_ids)
    assert np.array_equal(rows[order], ref['fm_weights']) and np.array_equal(fo[order], ref['fm_fields'])


# ------------------------------------------------------------------ A2, A3: line parser and layer-one array
def test_line_parser_and_layer_one_array_equal_reference(ref, built):
    path = os.path.join(DEMO, 'fm.model.txt')
    w0, k, xdim, fw, ff = orc.parse_fm_model(path)
    d = DataFM(path)
    rows64 = np.array([fw[f] for f in d.feat_weights], np.float64)                  # row order of the host table
    ids = []
    for i, line in enumerate(ref['fm_lines']):
        feats, y = orc.parse_line(str(line).strip())
        assert feats == unpad(ref['fm_line_feats'][i]) and y == int(ref['fm_line_y'][i])
        assert np.array_equal(orc.feats_to_layer_one_array(feats, w0, k, xdim, fw, ff), ref['fm_line_x'][i])
        f2, x2, y2 = d.get_fxy_fm(str(line).strip())
        assert f2 == feats and y2 == y and np.array_equal(x2, ref['fm_line_x'][i])
        ids.append(d.feats_to_ids(feats))
    # the id-matrix form every kernel consumes: gather(rows, ids) IS the reference's x
    assert np.array_equal(orc.gather(rows64, np.array(ids), w0), ref['fm_line_x'])


def test_edge_lines_equal_reference(ref, tmp_path, built):
    """Two features of one field (the later one wins), a feature listed twice, tabs and runs of blanks, few fields."""
    mp = tmp_path / 'edge.model.txt'
    mp.write_text(str(ref['edge_model_text']))
    w0, k, xdim, fw, ff = orc.parse_fm_model(str(mp))
    d = DataFM(str(mp))
    assert 1 + ff[13] * k + 2 == int(ref['edge_index_13_2']) == d.feat_layer_one_index(13, 2)
    lines = [str(v) for v in ref['edge_lines']]
    for i, line in enumerate(lines):
        feats, y = orc.parse_line(line)
        assert feats == unpad(ref['edge_feats'][i]) and y == int(ref['edge_y'][i])
        assert np.array_equal(orc.feats_to_layer_one_array(feats, w0, k, xdim, fw, ff), ref['edge_x'][i])
        assert np.array_equal(d.get_xy_fm(line)[0], ref['edge_x'][i])
    # the native reader on the same lines: ids -> gather == the reference's arrays; what "the later one wins" drops is reported
    ep = tmp_path / 'edge.fm.txt'
    ep.write_text('\n'.join(lines) + '\n')
    ids, y, sh = d.load_ids(str(ep), want_shadowed=True)
    rows64 = np.array([fw[f] for f in d.feat_weights], np.float64)
    assert np.array_equal(orc.gather(rows64, ids, w0), ref['edge_x']) and np.array_equal(y, ref['edge_y'])
    assert np.array_equal(sh, d.shadowed_of([unpad(r) for r in ref['edge_feats']])) and len(sh) == 3


# ------------------------------------------------------------------ A11: init_weight, file_len
def test_init_weight_equals_reference_stream(ref):
    dl_utils.seed_global(1234)                                   # python/dl_utils.py:9-10
    for i, (a, b, act) in enumerate(((177, 300, 'sigmoid'), (300, 100, 'tanh'), (5, 4, 'linear'))):
        w, bias = dl_utils.init_weight(a, b, act)
        assert np.array_equal(np.asarray(w, np.float64), ref['init_w%d' % i]) and np.array_equal(bias, ref['init_b%d' % i])
    assert dl_utils.file_len(os.path.join(DEMO, 'train.fm.txt')) == int(ref['file_len_train'])


# ------------------------------------------------------------------ A12: the yzx loaders
def test_ipinyou_loaders_equal_reference(ref, built):
    yzx = os.path.join(DEMO, 'train.yzx.txt')
    np.random.seed(7)
    max_dim, max_fea = ipy.stat(yzx)
    assert [max_dim, max_fea] == [int(v) for v in ref['yzx_stat']] == list(io.yzx_stat(yzx))[:2]
    np.random.seed(11)
    with open(yzx) as fin:
        a = ipy.load_ipinyou_data(fin, 300, max_dim + 1, max_fea + 2)
        b = ipy.load_ipinyou_data(fin, 100000, max_dim + 1, max_fea + 2)
        assert ipy.load_ipinyou_data(fin, 10, max_dim + 1, max_fea + 2) == (None, None, None)
    for got, tag in ((a, 'yzx_load1'), (b, 'yzx_load2')):
        assert np.array_equal(got[0], ref[tag + '_ind']) and np.array_equal(got[1], ref[tag + '_val']) and np.array_equal(got[2], ref[tag + '_y'])
    np.random.seed(13)
    rag = [[3, 5], [7], [1, 2, 9], []]
    Zi, Zv, zy = ipy.feed_zero([list(r) for r in rag], [[1] * len(r) for r in rag], [0, 1, 0, 1], 99, 4)
    assert np.array_equal(Zi, ref['feed_zero_ind']) and np.array_equal(Zv, ref['feed_zero_val']) and np.array_equal(zy, ref['feed_zero_y'])
    # the native whole-file reader returns file order: the same rows as the reference's (shuffled) buffers, as a multiset
    Xi, Xv, yy = ingest.parse_yzx(yzx, max_dim + 1, max_fea + 2)
    ref_rows = np.concatenate([ref['yzx_load1_ind'], ref['yzx_load2_ind']])
    key = lambda m: sorted(map(tuple, np.asarray(m).tolist()))       # noqa: E731
    assert key(Xi) == key(ref_rows) and sorted(yy.tolist()) == sorted(np.concatenate([ref['yzx_load1_y'], ref['yzx_load2_y']]).tolist())


# ------------------------------------------------------------------ A7, A7': the RBM trainers
@pytest.mark.parametrize("tag", ['a', 'b'])
def test_rbm_oracle_equals_reference_get_rbm_weights(ref, tag):
    """Three epochs of the online sparse CD-1 trainer and of the dense CD-1 trainer(s) on top, from the seed the module sets at
    import (1234): every array the reference's get_rbm_weights returns, to round-off (the restatement multiplies in another
    association order here and there; 1e-12 of the values' size)."""
    feats = [[int(v) for v in row] for row in ref['rbm_%s_feats' % tag]]
    arr = [int(v) for v in ref['rbm_%s_arr' % tag]]
    res = ro.get_rbm_weights(feats, arr, np.random.RandomState(1234), batch_size=int(ref['rbm_%s_batch' % tag]))
    assert len(res) == 2 * (len(arr) - 1)
    for i, r in enumerate(res):
        want = ref['rbm_%s_res%d' % (tag, i)]
        assert r.shape == want.shape
        np.testing.assert_allclose(r, want, rtol=0, atol=1e-12 * max(1.0, np.abs(want).max()), err_msg='result %d' % i)
