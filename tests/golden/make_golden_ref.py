#!/usr/bin/env python
"""Golden vectors from the REFERENCE ITSELF, for the parts of it that are plain Python + NumPy.

Runs only in the build container (it reads /root/reference, which never travels); what it writes --
tests/golden/ref_run.npz, inputs and the reference's outputs, data only -- is committed and is what
tests/test_oracle_vs_reference.py checks the oracle and the host code against, anywhere.

How the reference is run.  Its files are Python 2 (`print x`), so they do not import under the only interpreter here
(ordinary SyntaxError / ModuleNotFoundError, no permission denial).  Four of them are pure Python + NumPy
once that is out of the way:
    data_fm.py                                      DataFM: FM-model parser, line parser, layer-one array   (A1, A2, A3)
    dl_utils.py                                     init_weight, file_len                                   (A11)
    ipinyou.py                                      collect, stat, load_ipinyou_data, feed_zero             (A12)
    sampling_based_gaussian_binary_rbm_sparse.py    sparse online CD-1, dense CD-1, get_rbm_weights         (A7, A7')
Each file is read as text, passed IN MEMORY through lib2to3 (the stock Python 2 -> 3 fixers: print statements, xrange, ...),
every top-level `import` that fails in this container is skipped (theano, tensorflow and the reference's own Theano-/
TF-dependent modules stay ABSENT: nothing stands in for them -- code that would need them is simply not called), and the
result is executed as a module.  Two spellings the reference uses were removed from today's libraries and are given back their
old meaning for the run: `time.clock` (Python < 3.8; the trainers call it for their progress lines) = time.perf_counter, and
`np.NaN` (NumPy < 2.0; the buffer initialiser of the RBM classes) = np.nan.  No line of the algorithms is touched, nothing of
the reference is written anywhere.
The Theano scripts (FNN_wnzh.py, SNN_RBM.py, SNN_DAE.py, the dA) and the TensorFlow classes cannot be run this way: the
oracle stays "parity unpinned" for them.
"""
import ast
import io
import os
import sys
import tempfile
import time
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/python'
DEMO = os.path.join(HERE, 'demo')


def load_reference(name):
    import lib2to3.refactor as R
    path = os.path.join(REF, name + '.py')
    tool = R.RefactoringTool(R.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(open(path).read() + '\n', path)))
    body = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):     # an import that fails here is skipped, nothing replaces it
            node = ast.Try(body=[node], handlers=[ast.ExceptHandler(type=ast.Name('Exception', ast.Load()), name=None, body=[ast.Pass()])],
                           orelse=[], finalbody=[])
        body.append(node)
    tree.body = body
    ast.fix_missing_locations(tree)
    mod = types.ModuleType('reference_' + name)
    mod.__file__ = path
    with redirect_stdout(io.StringIO()):
        exec(compile(tree, path, 'exec'), mod.__dict__)
    return mod


def pad(lists, fill=-1):
    w = max(len(x) for x in lists)
    return np.array([list(x) + [fill] * (w - len(x)) for x in lists], dtype=np.int64)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference is not here: this script runs in the build container only")
    if not hasattr(time, 'clock'):
        time.clock = time.perf_counter
    if 'NaN' not in np.__dict__:
        np.NaN = np.nan
    out = {}
    work = tempfile.mkdtemp(prefix='refrun_')
    os.makedirs(os.path.join(work, 'cwd'))
    os.chdir(os.path.join(work, 'cwd'))                           # dl_utils.py creates ../log relative to the working directory

    # ---------------------------------------------------------------- A1, A2, A3: data_fm.DataFM
    fm = load_reference('data_fm')
    model_path = os.path.join(DEMO, 'fm.model.txt')
    d = fm.DataFM(model_path)
    feats = sorted(d.feat_weights)
    out['fm_w0'], out['fm_k'], out['fm_xdim'] = np.float64(d.w_0), np.int64(d.k), np.int64(d.xdim)
    out['fm_feats'] = np.array(feats, np.int64)
    out['fm_weights'] = np.array([d.feat_weights[f] for f in feats], np.float64)
    out['fm_fields'] = np.array([d.feat_field[f] for f in feats], np.int64)
    out['fm_first_key_order'] = np.array(list(d.feat_weights)[:50], np.int64)       # dict order = file order (ingestion row order)
    lines = [ln for ln in open(os.path.join(DEMO, 'train.fm.txt')).read().split('\n') if ln.strip()][:96]
    fx = [d.get_fxy_fm(ln.strip()) for ln in lines]
    out['fm_lines'] = np.array(lines)
    out['fm_line_feats'] = pad([f for f, _, _ in fx])
    out['fm_line_x'] = np.array([x for _, x, _ in fx], np.float64)
    out['fm_line_y'] = np.array([y for _, _, y in fx], np.int64)
    x2, y2 = d.get_xy_fm(lines[3].strip())
    assert np.array_equal(x2, fx[3][1]) and y2 == fx[3][2]
    # edge cases on a hand-made model: two features of one field on a line (the later one wins), a repeated feature, tabs / runs of
    # blanks, a line that names fewer fields
    em = ("-2.5 5 2\n10 1 2 3 region:x\n11 0.5 1.0 2.0 hour:y\n12 1 2 3 IP:9\n13 .5 5. -0 IP:10\n14 7 8 9 slotprice:1\n")
    el = ["1 10:1 11:1 12:1", "0 13:1 12:1", "1\t14:0   11:5", "0 10:1", "1 12:1 13:1 12:1 14:1"]
    ep = os.path.join(work, 'edge.model.txt')
    open(ep, 'w').write(em)
    de = fm.DataFM(ep)
    efx = [de.get_fxy_fm(ln) for ln in el]
    out['edge_model_text'], out['edge_lines'] = np.array(em), np.array(el)
    out['edge_feats'] = pad([f for f, _, _ in efx])
    out['edge_x'] = np.array([x for _, x, _ in efx], np.float64)
    out['edge_y'] = np.array([y for _, _, y in efx], np.int64)
    out['edge_index_13_2'] = np.int64(de.feat_layer_one_index(13, 2))

    # ---------------------------------------------------------------- A11: dl_utils.init_weight / file_len
    ut = load_reference('dl_utils')                                   # seeds the global NumPy stream with 1234 at import (:9-10)
    for i, (a, b, act) in enumerate(((177, 300, 'sigmoid'), (300, 100, 'tanh'), (5, 4, 'linear'))):
        w, bias = ut.init_weight(a, b, act)
        out['init_w%d' % i], out['init_b%d' % i] = np.array(w, np.float64), np.array(bias, np.float64)
    out['file_len_train'] = np.int64(ut.file_len(os.path.join(DEMO, 'train.fm.txt')))

    # ---------------------------------------------------------------- A12: ipinyou loaders
    ip = load_reference('ipinyou')
    yzx = os.path.join(DEMO, 'train.yzx.txt')
    np.random.seed(7)                                                 # collect() shuffles every buffer with the global stream (:19)
    max_dim, max_fea = ip.stat(yzx)
    out['yzx_stat'] = np.array([max_dim, max_fea], np.int64)
    np.random.seed(11)
    with open(yzx) as fin:
        Xi, Xv, yy = ip.load_ipinyou_data(fin, 300, max_dim + 1, max_fea + 2)      # pad id / width larger than needed: padding visible
        Xi2, Xv2, yy2 = ip.load_ipinyou_data(fin, 100000, max_dim + 1, max_fea + 2)
        end = ip.load_ipinyou_data(fin, 10, max_dim + 1, max_fea + 2)
    assert end == (None, None, None)
    out['yzx_load1_ind'], out['yzx_load1_val'], out['yzx_load1_y'] = Xi, Xv, yy
    out['yzx_load2_ind'], out['yzx_load2_val'], out['yzx_load2_y'] = Xi2, Xv2, yy2
    np.random.seed(13)
    rag_i = [[3, 5], [7], [1, 2, 9], []]
    with redirect_stdout(io.StringIO()):
        Zi, Zv, zy = ip.feed_zero([list(r) for r in rag_i], [[1] * len(r) for r in rag_i], [0, 1, 0, 1], 99, 4)
    out['feed_zero_ind'], out['feed_zero_val'], out['feed_zero_y'] = Zi, Zv, zy

    # ---------------------------------------------------------------- A7, A7': the NumPy RBM trainers
    def lines_file(path, n, n_rows, seed):
        """16 features per line with odd ids 2 r + 1 (id - 1 is then never a feature: 32 sampled visibles, :388), Zipf-ish rows."""
        rs = np.random.RandomState(seed)
        per = n_rows // 16
        feats = []
        with open(path, 'w') as f:
            for _ in range(n):
                r = [fld * per + min(per - 1, int(rs.zipf(1.3)) - 1) for fld in range(16)]
                ft = [2 * v + 1 for v in r]
                feats.append(ft)
                f.write('%d %s\n' % (rs.randint(0, 2), ' '.join('%d:1' % v for v in ft)))
        return feats
    for tag, n, arr_h, bs, seed in (('a', 240, [40, 24, 12], 100000, 3), ('b', 250, [32, 20], 100, 4)):
        path = os.path.join(work, 'rbm_%s.txt' % tag)
        feats = lines_file(path, n, 320, seed)
        x_dim = 2 * 320 + 2
        rb = load_reference('sampling_based_gaussian_binary_rbm_sparse')       # rng.seed(1234) at import (:7-8)
        with redirect_stdout(io.StringIO()):
            res = rb.get_rbm_weights(path, [x_dim] + arr_h, n, None, batch_size=bs)
        out['rbm_%s_feats' % tag] = np.array(feats, np.int64)
        out['rbm_%s_arr' % tag] = np.array([x_dim] + arr_h, np.int64)
        out['rbm_%s_batch' % tag] = np.int64(bs)
        for i, r in enumerate(res):
            out['rbm_%s_res%d' % (tag, i)] = np.array(r, np.float64)
    np.savez_compressed(os.path.join(HERE, 'ref_run.npz'), **out)
    print('wrote', os.path.join(HERE, 'ref_run.npz'), 'with', len(out), 'arrays')


if __name__ == '__main__':
    main()
