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


# =================================================================================================================
# Round 3: the plain-Python halves of the Theano / TensorFlow scripts (statements of FNN_wnzh.py, SNN_RBM.py and
# baseline.py executed node by node; make_golden_ref.py part 2).  This pins A6, the update of A8 and the host side of N4.
def fnn_problem(ref, tag):
    """The inputs of an `upd*` fixture as the oracle wants them, and the dense parameters its gx was made with."""
    feats = [int(f) for f in ref['fnn_script_feats']]
    row_of = {f: i for i, f in enumerate(feats)}
    field_of = {f: int(v) for f, v in zip(feats, ref['fnn_script_fields'])}
    lists = [unpad(r) for r in ref[tag + '_feats']]
    lr, lam = (float(v) for v in ref[tag + '_lr_lambda'])
    return feats, row_of, field_of, lists, lr, lam


def fnn_dense(ref, tag, xdim=177):
    seed = int(ref[tag + '_seed'])
    p = orc.init_fnn_weights(xdim, 300, 100)
    p['w3'] = np.random.RandomState(seed).uniform(-0.1, 0.1, 100)
    r1 = (np.random.RandomState(seed + 1).uniform(size=300) < 0.5).astype(np.float64)
    r2 = (np.random.RandomState(seed + 2).uniform(size=100) < 0.5).astype(np.float64)
    return p, r1, r2


def test_script_statements_parse_the_model_like_DataFM(ref):
    """FNN_wnzh.py:62-88 (the script's own copy of the model parser and index helper) == data_fm.DataFM == the oracle."""
    assert np.array_equal(ref['fnn_script_feats'], ref['fm_feats']) and np.array_equal(ref['fnn_script_weights'], ref['fm_weights'])
    assert np.array_equal(ref['fnn_script_fields'], ref['fm_fields'])
    assert [float(v) for v in ref['fnn_script_w0_k_xdim']] == [float(ref['fm_w0']), float(ref['fm_k']), float(ref['fm_xdim'])]
    w0, k, xdim, fw, ff = orc.parse_fm_model(os.path.join(DEMO, 'fm.model.txt'))
    f0 = int(ref['fnn_script_feats'][0])
    assert 1 + ff[f0] * k + 3 == int(ref['fnn_script_index_7_3'])


@pytest.mark.parametrize("tag", ['upd1', 'upd2', 'upd3'])
def test_sparse_row_update_equals_reference_loop(ref, tag, built):
    """A6: the reference's `for t in range(b_size): for feat in ft: for l in range(k)` statement (FNN_wnzh.py:299-306), run on
    these lines and this gx, against every restatement of it: the loop over feature lists (bit for bit: same operations in
    the same order), the id-matrix form and its closed form `row c^m - lr sum_j g_j c^(m-j)` (what the HIP kernels compute),
    and the vectorised CPU baseline of bench.py.  upd2: duplicate-heavy, lr 0.05, lambda_fm 0.3, a blank line in
    the file; upd3: two features of one field on a line, a feature listed twice."""
    feats, row_of, field_of, lists, lr, lam = fnn_problem(ref, tag)
    w0, k, xdim, fw, ff = orc.parse_fm_model(os.path.join(DEMO, 'fm.model.txt'))
    before, after, gx = ref['fnn_script_weights'], ref[tag + '_after'], ref[tag + '_gx']
    # the script's get_batch_data loop + get_fxy on the same file: lines -> feature lists, x, y
    lines = [str(v) for v in ref[tag + '_lines'] if str(v).strip()]
    assert [orc.parse_line(ln)[0] for ln in lines] == lists and [orc.parse_line(ln)[1] for ln in lines] == [int(v) for v in ref[tag + '_y']]
    x = np.array([orc.feats_to_layer_one_array(f, w0, k, xdim, fw, ff) for f in lists])
    assert np.array_equal(x, ref[tag + '_x'])
    rows = before.copy()
    orc.scatter_sgd_feats(rows, lists, row_of, field_of, gx, lr, lam)
    assert np.array_equal(rows, after)
    n_changed = int((np.abs(after - before).max(axis=1) > 0).sum())
    assert n_changed == len({f for ft in lists for f in ft})                    # exactly the listed features' rows moved
    one_per_field = all(len({field_of[f] for f in ft}) == len(ft) for ft in lists)
    assert one_per_field == (tag != 'upd3')
    if one_per_field:
        ids = np.full((len(lists), 16), -1, np.int64)
        for t, ft in enumerate(lists):
            for f in ft:
                ids[t, field_of[f]] = row_of[f]
        assert np.array_equal(orc.scatter_sgd(before.copy(), ids, gx, lr, lam), after)
        tol = 1e-12 * np.abs(after).max()
        assert np.abs(orc.scatter_sgd_closed_form(before, ids, gx, lr, lam) - after).max() <= tol
        assert np.abs(orc.scatter_sgd_vec(before.copy(), ids, gx, lr, lam) - after).max() <= tol
    if tag != 'upd2':
        # the fixture's gx is the oracle's train_call on the reference's x (the dense parameters are re-made from the seed):
        # what the GPU tests rebuild to run the same step on the device
        p, r1, r2 = fnn_dense(ref, tag)
        gx2 = orc.train_call(p, x, ref[tag + '_y'].astype(np.float64), r1, r2, lr, 0.0)[0]
        assert np.array_equal(gx2, gx)


@pytest.mark.parametrize("tag", ['snn1', 'snn2'])
def test_snn_bag_and_update_equal_reference_loops(ref, tag, tmp_path, built):
    """A8: the line loop of get_fi_h1_y (SNN_RBM.py:242-256: a token counts when its value is 1; x = sigmoid(sum of the
    active rows + bb0)) and the update loop of mytrain (:285-291), run by the reference's statements, against the oracle and
    the native reader.  snn2: values 0 / 2, features listed twice, random gx, lr 0.05."""
    s0, s1, x_dim, h0 = (int(v) for v in ref[tag + '_seeds_shape'])
    ww0 = np.random.RandomState(s0).uniform(-0.1, 0.1, (x_dim, h0))
    bb0 = np.random.RandomState(s1).uniform(-0.1, 0.1, h0)
    active = [unpad(r) for r in ref[tag + '_active']]
    lines = [str(v) for v in ref[tag + '_lines']]
    path = tmp_path / 'snn.fm.txt'
    path.write_text('\n'.join(lines) + '\n')
    ids_o, y_o = io.snn_active(str(path))                                       # the Python restatement of the reader
    assert [unpad(r) for r in ids_o] == active and np.array_equal(y_o, ref[tag + '_y'])
    ids_n, _, y_n = ingest.parse_examples(str(path), ingest.MODE_SNN_ACTIVE, None, 16)
    assert [unpad(r) for r in ids_n] == active and np.array_equal(y_n, ref[tag + '_y'])
    ids = np.array([a + [-1] * (16 - len(a)) for a in active], np.int64)
    x = orc.snn_bag(ww0, bb0, ids)
    np.testing.assert_allclose(x, ref[tag + '_x'], rtol=0, atol=1e-15)
    w, b = ww0.copy(), bb0.copy()
    orc.snn_update(w, b, ids, ref[tag + '_x'], ref[tag + '_gx'], float(ref[tag + '_lr']))
    touched = ref[tag + '_touched']
    assert np.array_equal(touched, np.unique(ids[ids >= 0]))
    assert np.abs(w[touched] - ref[tag + '_ww0_after_touched']).max() <= 1e-15 and np.abs(b - ref[tag + '_bb0_after']).max() <= 1e-15
    rest = np.setdiff1d(np.arange(x_dim), touched)
    assert np.array_equal(w[rest], ww0[rest])


def test_early_stop_and_recalibration_equal_reference(ref, capsys):
    """Row N4, host side: deep-ctr_amd/baseline.py (its own formulation: two window means) makes the decision of the
    reference's early_stop (baseline.py:262-281) for every prefix of six metric series under six window settings, both
    metrics (4,608 calls); the re-calibration equals both statements of the driver (:369, :422) to the bit."""
    from deep_ctr_amd import baseline as bl
    assert [float(v) for v in ref['baseline_defaults']] == [0.025, 1.0, 10.0, 10.0]
    assert bl.nds_rate == float(ref['baseline_defaults'][0])
    for i in (0, 1):
        assert np.array_equal(bl.re_calibrate(ref['nds_in']), ref['nds_out%d' % i])
    saved = (bl.least_step, bl.skip_window, bl.smooth_window, bl.stop_window)
    try:
        stops = 0
        for ci, cfg in enumerate(ref['es_cfgs']):
            bl.least_step, bl.skip_window, bl.smooth_window, bl.stop_window = (int(v) for v in cfg)
            for si, s in enumerate(ref['es_series']):
                for mi, metric in enumerate(('auc', 'rmse')):
                    got = [bl.early_stop(n, [float(v) for v in s[:n]], metric) for n in range(1, len(s) + 1)]
                    assert got == [bool(v) for v in ref['es_result'][ci, si, mi]], (ci, si, metric)
                    stops += sum(got)
        assert stops > 1000 and not ref['es_result'][:, 4].any()             # a flat metric never stops
    finally:
        bl.least_step, bl.skip_window, bl.smooth_window, bl.stop_window = saved
    assert 'early stop at step' in capsys.readouterr().out


def test_dae_sampling_and_propagation_equal_reference_loops(ref):
    """Row N2, host side: the token loop of sparse_da (sampling_based_denosing_autoencoder.py:303-311 -- per feature one draw
    int(rng.uniform(a, id)) of a negative visible, skipped when already present) and the lower-layer propagation loop of da
    (:164-188 -- the RUNNING sum over hidden units at layer 0, a sigmoid after every layer), run by the reference's own
    statements, against the oracle and the host code that feeds the kernels."""
    from oracle import dae_oracle as do
    from deep_ctr_amd import sampling_based_denosing_autoencoder as da
    lines = [str(v) for v in ref['dae_lines']]
    parsed = []
    for ln in lines:
        s = ln.strip().replace(':', ' ').split(' ')
        parsed.append(([int(s[f]) for f in range(1, len(s), 2)], [int(s[f + 1]) for f in range(1, len(s), 2)]))
    rs = np.random.RandomState(int(ref['dae_seed']))
    for n, (ids, vals) in enumerate(parsed):
        x, idx = do.sample_negatives(rs, ids, vals)
        assert idx == unpad(ref['dae_samp_idx'][n]) and x == [int(v) for v in ref['dae_samp_x'][n][:len(x)]]
    assert rs.random_sample() == float(ref['dae_next_draw'])                     # the stream stands where the reference's does
    rs = np.random.RandomState(int(ref['dae_seed']))
    idx, x = da.sampled_visibles(rs, parsed, 32)
    assert np.array_equal(idx, ref['dae_samp_idx']) and np.array_equal(x, ref['dae_samp_x']) and rs.random_sample() == float(ref['dae_next_draw'])
    results = [ref['dae_prop_res%d' % i] for i in range(4)]
    for n, row in enumerate(ref['dae_prop_ids']):
        np.testing.assert_allclose(do.propagate(results, unpad(row)), ref['dae_prop_out'][n], rtol=0, atol=1e-14)
