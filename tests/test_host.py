"""CPU tests of the host side: the loader interface mirrors (data_fm, ipinyou, dl_utils), the
synthetic-data writers, and the C-ABI library (loads, exports every declared symbol, fails loudly
without a GPU).  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import _capi, dl_utils, ipinyou, synth
from deep_ctr_amd.data_fm import DataFM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, K, H1, H2 = 16, 11, 300, 100
XDIM = 1 + F * K


# ------------------------------------------------------------------ C ABI
def test_header_symbols_are_exported_and_bound(built):
    hdr = open(os.path.join(ROOT, 'include', 'fnn_hip.h')).read()
    declared = set(re.findall(r'\b(fnn_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.fnn_version()
    hdr = open(os.path.join(ROOT, 'include', 'rbm_hip.h')).read()
    declared = set(re.findall(r'\b(rbm_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_capi.RBM_SIGNATURES), declared ^ set(_capi.RBM_SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    hdr = open(os.path.join(ROOT, 'include', 'ipnn_hip.h')).read()
    hdr_f = open(os.path.join(ROOT, 'include', 'fm_hip.h')).read()
    declared_f = set(re.findall(r'\b(fm_[a-z0-9_]+)\s*\(', hdr_f))
    assert declared_f == set(_capi.FM_SIGNATURES), declared_f ^ set(_capi.FM_SIGNATURES)
    for name in declared_f:
        assert hasattr(lib, name), name
    hdr_d = open(os.path.join(ROOT, 'include', 'dae_hip.h')).read()
    declared_d = set(re.findall(r'\b(dae_[a-z0-9_]+)\s*\(', hdr_d))
    assert declared_d == set(_capi.DAE_SIGNATURES), declared_d ^ set(_capi.DAE_SIGNATURES)
    for name in declared_d:
        assert hasattr(lib, name), name
    hdr_c = open(os.path.join(ROOT, 'include', 'ctr_ingest.h')).read()
    declared_c = set(re.findall(r'\b(ctr_[a-z0-9_]+)\s*\(', hdr_c))
    assert declared_c == set(_capi.CTR_SIGNATURES), declared_c ^ set(_capi.CTR_SIGNATURES)
    for name in declared_c:
        assert hasattr(lib, name), name
    declared = set(re.findall(r'\b(ipnn_[a-z0-9_]+)\s*\(', hdr)) - {'ipnn_cfg'}
    assert declared == set(_capi.IPNN_SIGNATURES), declared ^ set(_capi.IPNN_SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_library_exports_only_the_declared_c_symbols(built):
    """-fvisibility=hidden + the headers' visibility pragmas: the dynamic symbol table of libfnn_hip.so holds the C functions
    the six headers declare and nothing else of ours -- no C++-mangled name (round 2 leaked fnn::group_global and
    fnn::device_metrics), no kernel stub."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = set()
    for tab in (_capi.SIGNATURES, _capi.RBM_SIGNATURES, _capi.IPNN_SIGNATURES, _capi.FM_SIGNATURES, _capi.DAE_SIGNATURES, _capi.CTR_SIGNATURES):
        declared |= set(tab)
    assert declared <= names, declared - names
    extra = {n for n in names - declared if not n.startswith(('__hip', '_init', '_fini', '__bss', '_edata', '_end'))}
    assert not {n for n in extra if n.startswith('_Z')}, sorted(extra)[:10]
    assert not extra, sorted(extra)[:20]


def _struct_fields(header, name):
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), header, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    out = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]
        out += [n.strip().lstrip('*').split('[')[0] for n in names.split(',')]
    return out


def test_cfg_structs_agree_between_header_ctypes_library_and_integration_doc(built):
    """The struct is declared in include/*.h, in _capi.py and in INTEGRATION.md's stub: a stale copy makes *_create
    read past the caller's struct (round-1 finding)."""
    import ctypes as C
    lib = _capi.load()
    hdr = open(os.path.join(ROOT, 'include', 'fnn_hip.h')).read()
    assert _struct_fields(hdr, 'fnn_cfg') == [n for n, _ in _capi.fnn_cfg._fields_]
    assert lib.fnn_cfg_size() == C.sizeof(_capi.fnn_cfg)
    hdr_i = open(os.path.join(ROOT, 'include', 'ipnn_hip.h')).read()
    assert _struct_fields(hdr_i, 'ipnn_cfg') == [n for n, _ in _capi.ipnn_cfg._fields_]
    assert lib.ipnn_cfg_size() == C.sizeof(_capi.ipnn_cfg)
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = doc[doc.index('class fnn_cfg(C.Structure):'):doc.index('lib.fnn_cfg_size.restype')]
    assert re.findall(r'\("([a-z0-9_]+)", C\.', stub) == [n for n, _ in _capi.fnn_cfg._fields_]
    # the stub's positional constructor call passes exactly one value per field
    call = doc[doc.index('cfg = fnn_cfg('):doc.index('h = C.c_void_p()')]
    call = re.sub(r'#.*', '', call)
    args = call[call.index('(') + 1:call.rindex(')')]
    assert len([a for a in args.split(',') if a.strip()]) == len(_capi.fnn_cfg._fields_)


def test_create_validates_arguments_and_fails_loudly_without_gpu(built):
    import torch
    lib = _capi.load()
    h = C.c_void_p()
    bad = _capi.fnn_cfg(16, 16, 300, 100, 256, 0, 0, 0, 0.001, 0.0, 0.1, 0, None)      # k = 16 unsupported
    assert lib.fnn_create(C.byref(bad), C.byref(h)) == _capi.FNN_ERR_ARG
    assert b'k = rank+1' in lib.fnn_last_error(None)
    bad = _capi.fnn_cfg(16, 11, 300, 100, 1 << 20, 0, 0, 0, 0.001, 0.0, 0.1, 0, None)
    assert lib.fnn_create(C.byref(bad), C.byref(h)) == _capi.FNN_ERR_ARG
    assert lib.fnn_create(None, C.byref(h)) == _capi.FNN_ERR_ARG
    if not torch.cuda.is_available():
        ok = _capi.fnn_cfg(16, 11, 300, 100, 256, 0, 0, 0, 0.001, 0.0, 0.1, 0, None)
        assert lib.fnn_create(C.byref(ok), C.byref(h)) == _capi.FNN_ERR_HIP
        assert b'no CPU fallback' in lib.fnn_last_error(None)
        from deep_ctr_amd.engine import FNNEngine, FNNError
        with pytest.raises(FNNError):
            FNNEngine()


def test_null_handle_calls_return_errors(built):
    lib = _capi.load()
    assert lib.fnn_sync(None) == _capi.FNN_ERR_ARG
    assert lib.fnn_destroy(None) == _capi.FNN_ERR_ARG
    assert lib.fnn_predict(None, None, 1, None, 0) == _capi.FNN_ERR_ARG


# ------------------------------------------------------------------ data_fm.DataFM
@pytest.fixture(scope="module")
def demo(golden_dir):
    return os.path.join(golden_dir, 'demo')


def test_datafm_parses_model_like_the_reference(demo):
    d = DataFM(os.path.join(demo, 'fm.model.txt'))
    w0, k, xdim, fw, ff = orc.parse_fm_model(os.path.join(demo, 'fm.model.txt'))
    assert (d.w_0, d.k, d.xdim) == (w0, k, xdim) == (-3.0, K, XDIM)
    assert d.feat_weights == fw and d.feat_field == ff
    assert d.feat_layer_one_index(7, 0) == 1 and d.feat_layer_one_index(7, 3) == 4
    rows, fo, w_0 = d.table()
    assert rows.shape == (1000, K) and rows.dtype == np.float32 and fo.shape == (1000,)
    feat = 3 * 500 + 7
    assert np.array_equal(rows[d.feat_row[feat]], np.asarray(fw[feat], dtype=np.float32))
    assert fo[d.feat_row[feat]] == ff[feat]


def test_datafm_layer_one_array_and_lines(demo):
    d = DataFM(os.path.join(demo, 'fm.model.txt'))
    line = open(os.path.join(demo, 'train.fm.txt')).readline()
    x, y = d.get_xy_fm(line)
    feats, x2, y2 = d.get_fxy_fm(line)
    ref_feats, ref_y = orc.parse_line(line)
    assert feats == ref_feats and y == y2 == ref_y
    ref_x = orc.feats_to_layer_one_array(ref_feats, d.w_0, d.k, d.xdim, d.feat_weights, d.feat_field)
    assert np.array_equal(x, ref_x) and np.array_equal(x2, ref_x)
    ids = d.feats_to_ids(feats)
    assert ids.shape == (16,) and ids.dtype == np.int32 and (ids >= 0).all()
    assert np.array_equal(orc.gather(d.rows, ids[None, :], d.w_0)[0], ref_x)
    # two features of one field: the later one wins (python/data_fm.py:52-53)
    f0 = [f for f in d.feat_field if d.feat_field[f] == 0][:2]
    assert d.feats_to_ids(f0)[0] == d.feat_row[f0[1]]
    with pytest.raises(KeyError):
        d.feats_to_ids([999999])


def test_datafm_batches(demo):
    d = DataFM(os.path.join(demo, 'fm.model.txt'))
    train = os.path.join(demo, 'train.fm.txt')
    farray, ids, y = d.get_batch_ids(train, 1, 5)            # lines 1..5, 1-based (linecache)
    assert len(farray) == 5 and ids.shape == (5, 16) and y.shape == (5,)
    all_ids, all_y = d.load_ids(train)
    assert all_ids.shape == (1200, 16) and np.array_equal(all_ids[:5], ids) and np.array_equal(all_y[:5], y)
    farray, ids, y = d.get_batch_ids(train, 1199, 5)         # runs past EOF: blank lines skipped
    assert len(farray) == 2
    with pytest.raises(RuntimeError):                        # x is a device product: no CPU path
        d.get_batch_data(train, 1, 5)


def test_datafm_unknown_field_name_is_keyerror(tmp_path):
    p = tmp_path / 'fm.model.txt'
    p.write_text('0.1 1 2\n5 0.1 0.2 0.3 bogusfield:1\n')
    with pytest.raises(KeyError):
        DataFM(str(p))


# ------------------------------------------------------------------ ipinyou
def test_ipinyou_loaders(demo):
    path = os.path.join(demo, 'train.yzx.txt')
    max_dim, max_fea = ipinyou.stat(path)
    assert max_fea == 16 and 0 < max_dim < 1000
    with open(path) as fin:
        np.random.seed(5)
        X_ind, X_val, y = ipinyou.load_ipinyou_data(fin, 100, max_dim + 1, max_fea + 2)
        assert X_ind.shape == (100, 18) and X_val.shape == (100, 18) and y.shape == (100,)
        assert (X_ind[:, 16:] == max_dim + 1).all() and (X_val[:, 16:] == 0).all() and (X_val[:, :16] == 1).all()
        n = 100
        while True:
            a, b, c = ipinyou.load_ipinyou_data(fin, 500, max_dim + 1, max_fea)
            if a is None:
                assert b is None and c is None
                break
            n += len(c)
        assert n == 1200
    # collect shuffles with the global RNG (python/ipinyou.py:19)
    with open(path) as fin:
        np.random.seed(1); a = ipinyou.collect(fin, 50)
    with open(path) as fin:
        np.random.seed(1); b = ipinyou.collect(fin, 50)
    with open(path) as fin:
        first50 = [next(fin) for _ in range(50)]
    assert a == b and sorted(a) == sorted(first50) and a != first50
    fo = synth.field_of_row(synth.field_sizes_tiny(1000))
    ids = ipinyou.to_field_ids(X_ind[:, :16], X_val[:, :16], fo)
    assert ids.shape == (100, 16) and (ids >= 0).all()
    assert (fo[ids] == np.arange(16)[None, :]).all()


def test_ipinyou_feed_zero():
    X_ind = [[1, 2], [3], [4, 5, 6]]
    X_val = [[1, 1], [1], [1, 1, 1]]
    np.random.seed(3)
    a, b, c = ipinyou.feed_zero(X_ind, X_val, [0, 1, 0], 99, 4)
    assert a.shape == (3, 4) and sorted(c.tolist()) == [0, 0, 1]
    row = a[list(c).index(1)]
    assert row.tolist() == [3, 99, 99, 99]


# ------------------------------------------------------------------ dl_utils
def test_init_weights_follow_the_reference_rng_stream():
    dl_utils.seed_global(1234)
    p = dl_utils.init_fnn_weights(XDIM, H1, H2, 'tanh')
    ref = orc.init_fnn_weights(XDIM, H1, H2, 'tanh', seed=1234)
    for k in ('w1', 'w2', 'w3', 'b1', 'b2'):
        assert np.array_equal(p[k], ref[k])
    # dl_utils.init_weight: x4 for SIGMOID, x1 for tanh (python/dl_utils.py:48-51)
    dl_utils.seed_global(1234)
    ws, bs = dl_utils.init_weight(20, 30, 'sigmoid')
    dl_utils.seed_global(1234)
    wt, _ = dl_utils.init_weight(20, 30, 'tanh')
    assert np.allclose(ws, 4 * wt) and bs.shape == (30,) and not bs.any()
    assert np.abs(wt).max() <= np.sqrt(6. / 50)


def test_random_streams_match_oracle_masks():
    srng = dl_utils.RandomStreams(234)
    srng.binomial(size=(1, XDIM), n=1, p=1)           # r0: dead, but takes the first seed
    r1 = srng.binomial(size=(1, H1), n=1, p=0.5)
    r2 = srng.binomial(size=(1, H2), n=1, p=0.5)
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    for _ in range(3):
        a, b = ms.next()
        assert np.array_equal(r1.draw()[0], a) and np.array_equal(r2.draw()[0], b)


def test_file_len_and_logging(tmp_path, monkeypatch):
    p = tmp_path / 'f.txt'
    p.write_text('a\nb\nc\n')
    assert dl_utils.file_len(str(p)) == 3
    monkeypatch.setattr(dl_utils, 'log_path', str(tmp_path / 'log'))
    dl_utils.logfile('hello', 'fm2997')
    assert (tmp_path / 'log' / 'fm2997.txt').read_text() == 'hello\n'
    dl_utils.log_p('x', 'y')


# ------------------------------------------------------------------ synth
def test_synth_shapes_and_formats(tmp_path):
    sizes = synth.field_sizes_ipinyou()
    assert len(sizes) == 16 and sum(sizes) == 937670
    ids = synth.zipf_ids(512, sizes, 1.1, 3)
    fo = synth.field_of_row(sizes)
    assert ids.shape == (512, 16) and (fo[ids] == np.arange(16)[None, :]).all()
    assert len(np.unique(ids[:, 0])) <= 7                     # small fields repeat inside a batch
    d = synth.make_demo(str(tmp_path / 'd'), n_train=30, n_test=10, n_feat=200, rank=3, seed=1)
    m = DataFM(str(tmp_path / 'd' / 'fm.model.txt'))
    assert m.k == 4 and len(m.feat_row) == 200
    tr_ids, tr_y = m.load_ids(str(tmp_path / 'd' / 'train.fm.txt'))
    assert np.array_equal(tr_ids, d['ids'][:30]) and np.array_equal(tr_y, d['y'][:30])


# ------------------------------------------------------------------ RBM host logic
def test_rbm_line_dicts_follow_the_reference_orders(tmp_path):
    from deep_ctr_amd import sampling_based_gaussian_binary_rbm_sparse as gbrbm
    from oracle import rbm_oracle as ro
    # sparse trainer: x[id-1]=0 THEN x[id]=1, in line order (rbm_sparse.py:425-437);
    # adjacent ids 10, 11: 11's fake (10) overwrites 10's 1
    keys, v = ro.sparse_line_dict([10, 11, 20])
    assert keys == [9, 10, 11, 19, 20] and v == [0, 0, 1, 0, 1]
    # dense get_batch_x: x[id]=val THEN x[id-1]=0 (rbm_sparse.py:142-156): same outcome here
    x = ro.dense_line_dict([10, 11, 20])
    assert x == {10: 0, 9: 0, 11: 1, 20: 1, 19: 0}
    # opposite order in the line: 11 first, then 10 -> sparse keeps 10 (set after), dense too
    keys, v = ro.sparse_line_dict([11, 10])
    assert dict(zip(keys, v)) == {10: 1, 11: 1, 9: 0}
    p = tmp_path / 't.txt'
    p.write_text('1 10:1 11:1 20:1\n\n0 11:1 10:1 30:0\n')
    lines = gbrbm.parse_lines(str(p))
    assert lines == [([10, 11, 20], [1, 1, 1]), ([11, 10, 30], [1, 1, 0])]
    act = gbrbm.dense_active_ids(lines, 4)
    assert sorted(a for a in act[0] if a >= 0) == [11, 20] and sorted(a for a in act[1] if a >= 0) == [10, 11]


def test_baseline_early_stop_and_recalibration():
    """deep-ctr_amd/baseline.py against python/baseline.py:262-281 and :368-369 worked by hand."""
    from deep_ctr_amd import baseline as bl
    p = np.array([0.5, 0.1, 0.9])
    np.testing.assert_allclose(bl.re_calibrate(p, 0.025), p / (p + (1 - p) / 0.025))
    assert abs(bl.re_calibrate([0.5], 0.025)[0] - 0.025 / 1.025) < 1e-15
    bl.least_step, bl.skip_window, bl.smooth_window, bl.stop_window = 2, 1, 2, 2
    assert not bl.early_stop(2, [0.6, 0.7, 0.8])                 # not after least_step yet
    assert not bl.early_stop(3, [0.6, 0.7, 0.8, 0.9])            # smoothed auc still rising
    assert bl.early_stop(5, [0.6, 0.7, 0.8, 0.7, 0.6])           # smoothed auc falls: (0.65) - (0.75) < 0
    assert bl.early_stop(5, [0.3, 0.2, 0.2, 0.3, 0.4], metric='rmse')
    assert not bl.early_stop(5, [0.9], metric='auc')             # too few smoothed points
    bl.least_step, bl.skip_window, bl.smooth_window, bl.stop_window = 0, 1, 1, 2


def test_bench_refuses_to_run_fewer_ranks_than_asked(built):
    """`python bench.py --gpus N` outside torchrun starts the ranks itself and must fail loudly when it cannot (round 1 ran
    ONE rank and printed n_gpus: 1); under torchrun a WORLD_SIZE that disagrees with --gpus is an error too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FNN_BENCH_REHEARSE')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'GPU(s) visible' in (r.stderr + r.stdout)
    env['WORLD_SIZE'] = '4'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=4' in (r.stderr + r.stdout)


def test_mask_rows_drawn_ahead_equal_per_step_draws():
    """FNN.py draws an epoch's dropout rows in one call (_BinomialOp.draw_rows): the same rows as one draw per `train` call
    (python/FNN_wnzh.py:154,166 through theano's RandomStreams), for the script's dropout rates."""
    from deep_ctr_amd import dl_utils as ut
    for p in (0.5, 0.98, 1.0):
        a, b = ut.RandomStreams(234), ut.RandomStreams(234)
        for s in (a, b):
            s.binomial(size=(1, 177), n=1, p=1)
        oa, ob = a.binomial(size=(1, 300), n=1, p=p), b.binomial(size=(1, 300), n=1, p=p)
        seq = np.concatenate([oa.draw() for _ in range(40)] )
        assert np.array_equal(seq, ob.draw_rows(40))
        assert np.array_equal(oa.draw(), ob.draw_rows(1))            # and the streams stay in step afterwards
