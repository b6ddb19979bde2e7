#!/usr/bin/env python
"""Golden vectors from the REFERENCE ITSELF, for the parts of it that are plain Python + NumPy.

Runs only in the build container (it reads /root/reference, which never travels); what it writes --
tests/golden/ref_run.npz (inputs and the reference's outputs, data only) and ref_run.manifest.txt (sha256 of every reference
file that was run and of every array: the binary can be audited from the diff) -- is committed and is what
tests/test_oracle_vs_reference.py and tests/test_gpu_reference_pins.py check the oracle, the host code and the HIP path
against, anywhere.

How the reference is run.  Its files are Python 2 (`print x`), so they do not import under the only interpreter here
(ordinary SyntaxError / ModuleNotFoundError, no permission denial).  Each file is read as text and passed IN MEMORY
through lib2to3 (the stock Python 2 -> 3 fixers: print statements, xrange, ...); nothing of the reference is written anywhere.

Part 1 -- whole modules.  Four files are pure Python + NumPy once the syntax is out of the way:
    data_fm.py                                      DataFM: FM-model parser, line parser, layer-one array   (A1, A2, A3)
    dl_utils.py                                     init_weight, file_len                                   (A11)
    ipinyou.py                                      collect, stat, load_ipinyou_data, feed_zero             (A12)
    sampling_based_gaussian_binary_rbm_sparse.py    sparse online CD-1, dense CD-1, get_rbm_weights         (A7, A7')
Every top-level `import` that fails in this container is skipped (theano, tensorflow and the reference's own Theano-/
TF-dependent modules stay ABSENT: nothing stands in for them -- code that would need them is simply not called), and the
result is executed as a module.  Two spellings the reference uses were removed from today's libraries and are given back their
old meaning for the run: `time.clock` (Python < 3.8) = time.perf_counter, and `np.NaN` (NumPy < 2.0) = np.nan.

Part 2 (round 3) -- the plain-Python HALVES of the Theano / TensorFlow scripts.  FNN_wnzh.py, SNN_RBM.py and baseline.py
cannot be executed as modules (they build Theano / TensorFlow graphs at import), but the statements around the compiled
callables are plain Python + NumPy.  Their syntax-tree NODES are taken out of the parsed file and executed on given values,
unchanged, in a namespace that holds only NumPy, linecache and the inputs:
    FNN_wnzh.py   :51-53 name_field, :62-84 the FM-model parser, :87-96 feat_layer_one_index / feats_to_layer_one_array,
                  :224-232 the line loop of get_batch_data, :240-253 get_xy / get_fxy,
                  :299-306 THE SPARSE-ROW UPDATE LOOP `for t in range(b_size)`                               (A1-A3, A6)
    SNN_RBM.py    :239-256 the line loop of get_fi_h1_y (active features, bag sum, sigmoid),
                  :285-291 THE UPDATE LOOP of mytrain (bb0 / ww0)                                           (A8)
    baseline.py   :262-281 early_stop, :369 / :422 the NDS re-calibration statements                        (row N4, host side)
    sampling_based_denosing_autoencoder.py   :303-311 the negative-sampling token loop of sparse_da (int(rng.uniform(a, id)) draws),
                  :165-188 the lower-layer propagation loop of da (the running sum over hidden units, a sigmoid after
                  every layer)                                                                               (row N2, host side)
Nothing stands in for Theano: a statement that names `theano` is not executed (get_batch_data's and get_fi_h1_y's final
`numpy.array(..., dtype=theano.config.floatX)` conversions).  The gradients `gx` those loops consume come from the compiled
callable in the reference; here they are inputs of the fixture -- random arrays, or (for the fixtures the GPU step is held to)
the float64 oracle's `train_call` on the reference's own `x` (oracle/fnn_oracle.py: the MLP itself stays unpinned).

Isolation (advisor, round 2).  The reference is untrusted code, so it is executed in a CHILD process that has been handed the
source text and the input data over a pipe, runs as `nobody` (no read access to /root, write access to one temporary
directory) with an empty environment, and returns one .npz; the parent loads it with allow_pickle=False and writes the
committed files.
"""
import ast
import hashlib
import io
import json
import linecache
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import time
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__)) if '__file__' in globals() else None     # the child runs from text
REF = '/root/reference/python'
MODULES = ['data_fm', 'dl_utils', 'ipinyou', 'sampling_based_gaussian_binary_rbm_sparse']
SCRIPTS = ['FNN_wnzh', 'SNN_RBM', 'baseline', 'sampling_based_denosing_autoencoder']
SOURCES = {}                     # name -> text (the child's copy arrives over the pipe)


# ------------------------------------------------------------------------------------------------ running reference text
def reference_tree(name):
    """The file's syntax tree after lib2to3; nothing is executed."""
    import lib2to3.refactor as R
    path = os.path.join(REF, name + '.py')
    tool = R.RefactoringTool(R.get_fixers_from_package('lib2to3.fixes'))
    return ast.parse(str(tool.refactor_string(SOURCES[name] + '\n', path))), path


def load_reference(name):
    """Part 1: a whole module, failing imports skipped."""
    tree, path = reference_tree(name)
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


def run_nodes(nodes, path, ns):
    """Part 2: execute the given statements of a reference file, unchanged, in `ns`."""
    mod = ast.Module(body=list(nodes), type_ignores=[])
    with redirect_stdout(io.StringIO()):
        exec(compile(mod, path, 'exec'), ns)
    return ns


def names_in(node):
    return {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}


def assigned(node):
    return {t.id for t in node.targets if isinstance(t, ast.Name)} if isinstance(node, ast.Assign) else set()


def function(tree, name):
    return next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == name)


def loop_over(tree, target, iter_src):
    """The one `for <target> in <iter_src>` statement of the file."""
    hits = [n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == target
            and ast.unparse(n.iter) == iter_src]
    assert len(hits) == 1, (target, iter_src, len(hits))
    return hits[0]


def through_first_loop(fn):
    """A function's statements up to and including its first `for` (what follows converts with theano.config.floatX)."""
    out = []
    for st in fn.body:
        assert 'theano' not in names_in(st)
        out.append(st)
        if isinstance(st, ast.For):
            return out
    raise AssertionError('no loop in ' + fn.name)


def pad(lists, fill=-1):
    w = max([len(x) for x in lists] + [1])
    return np.array([list(x) + [fill] * (w - len(x)) for x in lists], dtype=np.int64)


# ------------------------------------------------------------------------------------------------ part 1 (whole modules)
def part1(out, work, demo):
    fm = load_reference('data_fm')
    model_path = os.path.join(demo, 'fm.model.txt')
    d = fm.DataFM(model_path)
    feats = sorted(d.feat_weights)
    out['fm_w0'], out['fm_k'], out['fm_xdim'] = np.float64(d.w_0), np.int64(d.k), np.int64(d.xdim)
    out['fm_feats'] = np.array(feats, np.int64)
    out['fm_weights'] = np.array([d.feat_weights[f] for f in feats], np.float64)
    out['fm_fields'] = np.array([d.feat_field[f] for f in feats], np.int64)
    out['fm_first_key_order'] = np.array(list(d.feat_weights)[:50], np.int64)       # dict order = file order (ingestion row order)
    lines = [ln for ln in open(os.path.join(demo, 'train.fm.txt')).read().split('\n') if ln.strip()][:96]
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
    out['file_len_train'] = np.int64(ut.file_len(os.path.join(demo, 'train.fm.txt')))

    # ---------------------------------------------------------------- A12: ipinyou loaders
    ip = load_reference('ipinyou')
    yzx = os.path.join(demo, 'train.yzx.txt')
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


# ------------------------------------------------------------------------------------------------ part 2 (statements of the scripts)
def part2(out, work, demo, given):
    model_path = os.path.join(demo, 'fm.model.txt')

    # ================================================================ FNN_wnzh.py
    tree, path = reference_tree('FNN_wnzh')
    first_def, last_def = function(tree, 'log_p'), function(tree, 'feats_to_layer_one_array')
    head = []                                                          # :51-96 without the log_p lines: the model parser and the index helpers
    keep = {'name_field', 'feat_field', 'feat_weights', 'w_0', 'feat_num', 'k', 'xdim', 'fi', 'first'}
    for node in tree.body:
        if node.lineno > last_def.lineno:
            break
        if isinstance(node, ast.Assign) and assigned(node) & keep and (assigned(node) == {'name_field'} or node.lineno > first_def.lineno):
            head.append(node)
        elif isinstance(node, ast.For) and ast.unparse(node.iter) == 'fi':
            head.append(node)
        elif isinstance(node, ast.FunctionDef) and node.name in ('feat_layer_one_index', 'feats_to_layer_one_array'):
            head.append(node)
    assert [type(n).__name__ for n in head] == ['Assign'] * 9 + ['For', 'FunctionDef', 'FunctionDef'], [ast.unparse(n)[:40] for n in head]
    head += [function(tree, 'get_xy'), function(tree, 'get_fxy')]
    update = loop_over(tree, 't', 'range(b_size)')                     # :299-306
    batch_loop = through_first_loop(function(tree, 'get_batch_data'))  # :225-235

    def fresh_script():
        ns = {'numpy': np, 'linecache': linecache, 'fm_model_file': model_path}
        run_nodes(head, path, ns)
        ns['fi'].close()
        return ns
    ns = fresh_script()
    feats_sorted = sorted(ns['feat_weights'])
    out['fnn_script_feats'] = np.array(feats_sorted, np.int64)
    out['fnn_script_w0_k_xdim'] = np.array([ns['w_0'], ns['k'], ns['xdim']], np.float64)
    out['fnn_script_weights'] = np.array([ns['feat_weights'][f] for f in feats_sorted], np.float64)
    out['fnn_script_fields'] = np.array([ns['feat_field'][f] for f in feats_sorted], np.int64)
    out['fnn_script_index_7_3'] = np.int64(ns['feat_layer_one_index'](feats_sorted[0], 3))

    def table(ns):
        return np.array([ns['feat_weights'][f] for f in feats_sorted], np.float64)

    for tag in ('upd1', 'upd2', 'upd3'):
        g = given[tag]
        lines_path = os.path.join(work, tag + '.txt')
        open(lines_path, 'w').write('\n'.join(g['lines']) + '\n')
        linecache.checkcache()
        ns = fresh_script()
        # get_batch_data(file, index, size): its line loop with get_fxy; a blank line is skipped (:230)
        ns.update(file=lines_path, index=1, size=len(g['lines']) + 3)
        run_nodes(batch_loop, path, ns)
        f, x, y = ns['farray'], np.array(ns['xarray'], np.float64), np.array(ns['yarray'], np.int64)
        assert len(f) == len([ln for ln in g['lines'] if ln.strip()])
        x1, y1 = ns['get_xy'](g['lines'][0].strip())
        assert np.array_equal(x1, x[0]) and y1 == y[0]
        if g.get('x') is not None:                                     # the oracle's x, from which gx was derived: the reference's x, bit for bit
            assert np.array_equal(x, g['x']), tag
        before = table(ns)
        ns.update(f=f, gx=g['gx'], b_size=len(f), lr=g['lr'], lambda_fm=g['lambda_fm'])
        run_nodes([update], path, ns)
        out[tag + '_lines'] = np.array(g['lines'])
        out[tag + '_feats'], out[tag + '_x'], out[tag + '_y'] = pad(f), x, y
        out[tag + '_gx'] = np.array(g['gx'], np.float64)
        out[tag + '_lr_lambda'] = np.array([g['lr'], g['lambda_fm']], np.float64)
        out[tag + '_seed'] = np.int64(g['seed'])               # w3, r1, r2 = RandomState(seed), (seed + 1), (seed + 2): see oracle_inputs
        out[tag + '_after'] = table(ns)
        assert np.array_equal(before, out['fnn_script_weights']) and not np.array_equal(before, out[tag + '_after'])

    # ================================================================ SNN_RBM.py
    tree, path = reference_tree('SNN_RBM')
    bag_loop = through_first_loop(function(tree, 'get_fi_h1_y'))       # :237-256 (two `global` statements, three lists, the loop)
    update = loop_over(tree, 't', 'range(b_size)')                     # :285-291
    for tag in ('snn1', 'snn2'):
        g = given[tag]
        lines_path = os.path.join(work, tag + '.txt')
        open(lines_path, 'w').write('\n'.join(g['lines']) + '\n')
        linecache.checkcache()
        ww0, bb0 = np.array(g['ww0'], np.float64), np.array(g['bb0'], np.float64)
        ns = {'numpy': np, 'linecache': linecache, 'ww0': ww0, 'bb0': bb0, 'file': lines_path, 'index': 1, 'size': len(g['lines']) + 2}
        run_nodes(bag_loop, path, ns)
        fi, x, y = ns['farray'], np.array(ns['xarray'], np.float64), np.array(ns['yarray'], np.int64)
        if g.get('x') is not None:
            np.testing.assert_allclose(x, g['x'], rtol=0, atol=1e-15, err_msg=tag)
        gx = np.array(g['gx'], np.float64)
        ns.update(fi=fi, x=x, gx=gx, b_size=len(fi), lr=g['lr'])
        run_nodes([update], path, ns)
        touched = np.array(sorted({ft for row in fi for ft in row}), np.int64)
        untouched = np.setdiff1d(np.arange(ww0.shape[0]), touched)
        assert np.array_equal(ns['ww0'][untouched], np.array(g['ww0'], np.float64)[untouched])
        out[tag + '_lines'] = np.array(g['lines'])
        out[tag + '_active'], out[tag + '_x'], out[tag + '_y'], out[tag + '_gx'] = pad(fi), x, y, gx
        out[tag + '_lr'] = np.float64(g['lr'])
        out[tag + '_seeds_shape'] = np.array(g['seeds_shape'], np.int64)          # ww0 / bb0 are RandomState(seed).uniform(-.1, .1, shape)
        out[tag + '_touched'] = touched
        out[tag + '_ww0_after_touched'] = np.array(ns['ww0'], np.float64)[touched]
        out[tag + '_bb0_after'] = np.array(ns['bb0'], np.float64)

    # ================================================================ sampling_based_denosing_autoencoder.py (row N2)
    tree, path = reference_tree('sampling_based_denosing_autoencoder')
    tok_loops = [n for n in ast.walk(function(tree, 'sparse_da')) if isinstance(n, ast.For) and ast.unparse(n.iter) == 'range(1, len(s), 2)']
    assert len(tok_loops) == 1 and 'new_sample' in names_in(tok_loops[0])
    lines = given['dae_lines']
    seed = int(given['dae_seed'])
    rs = np.random.RandomState(seed)
    xs, idxs = [], []
    for ln in lines:                                                   # :302-303: s = line.strip().replace(':', ' ').split(' '), x / indexes / a reset per line
        ns = {'s': ln.strip().replace(':', ' ').split(' '), 'k': 2, 'rng': rs, 'x': [], 'indexes': [], 'a': 0}
        run_nodes(tok_loops, path, ns)
        xs.append(ns['x']); idxs.append(ns['indexes'])
    out['dae_lines'], out['dae_seed'] = np.array(lines), np.int64(seed)
    out['dae_samp_x'], out['dae_samp_idx'] = pad(xs), pad(idxs)
    out['dae_next_draw'] = np.float64(rs.random_sample())              # where the stream stands afterwards
    prop = []                                                          # `i=0` and the `for r in results:` behind it (:164-188; a second copy serves the last batch, :198-222)
    for parent in ast.walk(function(tree, 'da')):
        body = getattr(parent, 'body', None)
        if isinstance(body, list):
            for k, st in enumerate(body):
                if isinstance(st, ast.For) and ast.unparse(st.iter) == 'results':
                    assert k > 0 and ast.unparse(body[k - 1]) == 'i = 0'
                    prop.append((st.lineno, [body[k - 1], st]))
    prop = [nodes for _, nodes in sorted(prop, key=lambda x: x[0])]
    assert len(prop) == 2
    res_list = [np.array(a, np.float64) for a in given['dae_results']]
    ns = {'numpy': np, 'np': np, 'results': res_list, 'indexes': [list(r) for r in given['dae_prop_ids']],
          'batcharr': np.array(given['dae_prop_vals'], dtype=np.float32)}
    run_nodes(prop[0], path, ns)
    out['dae_prop_ids'] = pad(given['dae_prop_ids'])
    for i, a in enumerate(res_list):
        out['dae_prop_res%d' % i] = a
    out['dae_prop_out'] = np.array(ns['batcharr'], np.float64)

    # ================================================================ baseline.py
    tree, path = reference_tree('baseline')
    es = function(tree, 'early_stop')                                  # :262-281
    recal = [n for n in ast.walk(tree) if isinstance(n, ast.AugAssign) and isinstance(n.op, ast.Div) and 'nds_rate' in names_in(n)]
    assert sorted(ast.unparse(n) for n in recal) == ['eval_preds /= eval_preds + (1 - eval_preds) / nds_rate', 'p /= p + (1 - p) / nds_rate']
    nds = next(n for n in tree.body if assigned(n) == {'nds_rate'})    # :21
    windows = {n.lineno: n for n in tree.body if assigned(n) & {'skip_window', 'smooth_window', 'stop_window'}}
    ns = run_nodes([nds] + [windows[k] for k in sorted(windows)], path, {'np': np})
    out['baseline_defaults'] = np.array([ns['nds_rate'], ns['skip_window'], ns['smooth_window'], ns['stop_window']], np.float64)
    pr = np.array(given['nds_p'], np.float64)
    for i, node in enumerate(sorted(recal, key=lambda n: n.lineno)):
        ns2 = {'nds_rate': ns['nds_rate'], 'eval_preds': pr.copy(), 'p': pr.copy()}
        run_nodes([node], path, ns2)
        out['nds_out%d' % i] = ns2['eval_preds'] if i == 0 else ns2['p']
    out['nds_in'] = pr
    cfgs = np.array(given['es_cfgs'], np.int64)                        # (least_step, skip_window, smooth_window, stop_window)
    series = np.array(given['es_series'], np.float64)                  # [n_series, T]
    res = np.zeros((len(cfgs), len(series), 2, series.shape[1]), np.int8)
    for ci, (least, skip, smooth, stop) in enumerate(cfgs):
        ns = {'np': np, 'least_step': int(least), 'skip_window': int(skip), 'smooth_window': int(smooth), 'stop_window': int(stop)}
        run_nodes([es], path, ns)
        with redirect_stdout(io.StringIO()):
            for si, s in enumerate(series):
                for mi, metric in enumerate(('auc', 'rmse')):
                    for n in range(1, series.shape[1] + 1):
                        # the driver passes its python list of recorded metrics and the running step (:372-375)
                        res[ci, si, mi, n - 1] = bool(ns['early_stop'](n, [float(v) for v in s[:n]], metric))
    out['es_cfgs'], out['es_series'], out['es_result'] = cfgs, series, res


# ------------------------------------------------------------------------------------------------ child / parent
def child(work):
    payload = pickle.load(sys.stdin.buffer)                            # from the parent (this script): trusted
    SOURCES.update(payload['sources'])
    if not hasattr(time, 'clock'):
        time.clock = time.perf_counter
    if 'NaN' not in np.__dict__:
        np.NaN = np.nan
    demo = os.path.join(work, 'demo')
    os.makedirs(os.path.join(work, 'cwd'))
    os.chdir(os.path.join(work, 'cwd'))                                # dl_utils.py creates ../log relative to the working directory
    out = {}
    part1(out, work, demo)
    part2(out, work, demo, payload['given'])
    np.savez_compressed(os.path.join(work, 'out.npz'), **out)


def oracle_inputs(demo):
    """What part 2 is given: lines, and gradients to feed the reference's update loops (see the header)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import fnn_oracle as orc
    model = os.path.join(demo, 'fm.model.txt')
    w0, k, xdim, fw, ff = orc.parse_fm_model(model)
    demo_lines = [ln.strip() for ln in open(os.path.join(demo, 'train.fm.txt')) if ln.strip()]
    by_field = {}
    for f in fw:
        by_field.setdefault(ff[f], []).append(f)
    given = {}

    def mlp_gx(lines, lr, seed):
        feats_y = [orc.parse_line(ln) for ln in lines if ln.strip()]
        x = np.array([orc.feats_to_layer_one_array(f, w0, k, xdim, fw, ff) for f, _ in feats_y])
        y = np.array([yy for _, yy in feats_y], np.float64)
        p = orc.init_fnn_weights(xdim, 300, 100)
        p['w3'] = np.random.RandomState(seed).uniform(-0.1, 0.1, 100)  # the reference starts at w3 = 0, where gx = 0
        r1 = (np.random.RandomState(seed + 1).uniform(size=300) < 0.5).astype(np.float64)
        r2 = (np.random.RandomState(seed + 2).uniform(size=100) < 0.5).astype(np.float64)
        gx = orc.train_call(p, x, y, r1, r2, lr, 0.0)[0]
        return x, gx
    # upd1: 64 lines of the demo set (Zipf ids: rows hit by many examples), the reference's lr / lambda_fm, gx of a real step
    lines = demo_lines[:64]
    x, gx = mlp_gx(lines, 0.001, 10)
    given['upd1'] = dict(lines=lines, gx=gx, x=x, lr=0.001, lambda_fm=0.1, seed=10)
    # upd2: a duplicate-heavy batch (three candidate features per field), a blank line in the file, large lr / lambda_fm so that the
    # decay c^m is visible, RANDOM gx (the scatter alone: fnn_step_scatter_global on the GPU side)
    rs = np.random.RandomState(31)
    lines = []
    for t in range(48):
        fl = sorted(rs.choice(16, size=rs.randint(10, 17), replace=False))
        lines.append('%d %s' % (rs.randint(0, 2), ' '.join('%d:1' % by_field[f][rs.randint(0, min(3, len(by_field[f])))] for f in fl)))
    lines.insert(20, '')
    given['upd2'] = dict(lines=lines, gx=rs.normal(0, 0.5, (48, xdim)), x=None, lr=0.05, lambda_fm=0.3, seed=31)
    # upd3: two features of one field on a line (both rows are updated, the later one is gathered), a feature listed twice (updated
    # twice), few fields; gx of a real step on the reference's x
    rs = np.random.RandomState(32)
    lines = []
    for t in range(40):
        toks = []
        for f in sorted(rs.choice(16, size=rs.randint(6, 17), replace=False)):
            c = by_field[f]
            a = c[rs.randint(0, min(4, len(c)))]
            toks.append(a)
            u = rs.uniform()
            if u < 0.15:
                toks.append(c[rs.randint(0, min(4, len(c)))])                    # a second feature of the field (maybe the same one)
            elif u < 0.22:
                toks.append(a)                                          # the same feature twice
        lines.append('%d %s' % (rs.randint(0, 2), ' '.join('%d:1' % v for v in toks)))
    x, gx = mlp_gx(lines, 0.01, 40)
    given['upd3'] = dict(lines=lines, gx=gx, x=x, lr=0.01, lambda_fm=0.2, seed=40)

    # snn1: 16 demo lines, hidden0 = 200 (advertiser 2997), gx of a real step on the bag output; snn2: a small hidden0, values 0 / 2
    # (not active, :251), repeated features, random gx
    x_dim = max(fw) + 1

    def bag_case(lines, h0, seeds, lr, real):
        ww0 = np.random.RandomState(seeds[0]).uniform(-0.1, 0.1, (x_dim, h0))
        bb0 = np.random.RandomState(seeds[1]).uniform(-0.1, 0.1, h0)
        act = []
        for ln in lines:
            s = ln.strip().replace(':', ' ').split(' ')
            act.append([int(s[j]) for j in range(1, len(s), 2) if int(s[j + 1]) == 1])
        z = np.array([ww0[a].sum(0) + bb0 if a else bb0 for a in act])
        x = 1.0 / (1.0 + np.exp(-z))
        if real:
            y = np.array([int(ln.split(' ')[0]) for ln in lines], np.float64)
            p = orc.init_fnn_weights(h0, 300, 100)
            p['w3'] = np.random.RandomState(seeds[0] + 5).uniform(-0.1, 0.1, 100)
            r1 = (np.random.RandomState(seeds[0] + 6).uniform(size=300) < 0.5).astype(np.float64)
            r2 = (np.random.RandomState(seeds[0] + 7).uniform(size=100) < 0.5).astype(np.float64)
            gx = orc.train_call(p, x, y, r1, r2, lr, 0.0, reg_all=True)[0]
        else:
            gx = np.random.RandomState(seeds[0] + 5).normal(0, 0.5, x.shape)
        return dict(lines=lines, ww0=ww0, bb0=bb0, gx=gx, x=x if real else None, lr=lr, seeds_shape=[seeds[0], seeds[1], x_dim, h0])
    given['snn1'] = bag_case(demo_lines[64:80], 200, (21, 22), 0.001, True)
    rs = np.random.RandomState(33)
    lines = []
    for t in range(40):
        toks = []
        for f in sorted(rs.choice(16, size=rs.randint(4, 17), replace=False)):
            c = by_field[f]
            toks.append('%d:%d' % (c[rs.randint(0, min(3, len(c)))], [1, 1, 1, 1, 0, 2][rs.randint(0, 6)]))
            if rs.uniform() < 0.1:
                toks.append(toks[-1])
        lines.append('%d %s' % (rs.randint(0, 2), ' '.join(toks)))
    given['snn2'] = bag_case(lines, 12, (23, 24), 0.05, False)

    # early stop: AUC-like series (rise then plateau / fall, noise), the reference's windows and smaller ones
    rs = np.random.RandomState(34)
    T = 64
    t = np.arange(T)
    series = [0.6 + 0.2 * (1 - np.exp(-t / 8.0)) + rs.normal(0, 0.004, T),
              0.8 - 0.002 * t + rs.normal(0, 0.002, T),
              0.7 + 0.1 * np.sin(t / 5.0) + rs.normal(0, 0.01, T),
              0.6 + 0.2 * (1 - np.exp(-t / 6.0)) - 0.004 * np.maximum(0, t - 30),
              np.full(T, 0.75),
              rs.uniform(0.5, 0.9, T)]
    # row N2: the dA module's token loop (negative sampling) on demo lines, and its lower-layer propagation on two lower layers
    given['dae_lines'] = demo_lines[80:104]
    given['dae_seed'] = 123
    rs = np.random.RandomState(35)
    n_vis = max(fw) + 1
    given['dae_results'] = [rs.uniform(-0.3, 0.3, (n_vis, 12)), rs.uniform(-0.1, 0.1, 12), rs.uniform(-0.5, 0.5, (12, 7)), rs.uniform(-0.1, 0.1, 7)]
    given['dae_prop_ids'] = [orc.parse_line(ln)[0] for ln in demo_lines[104:112]]
    given['dae_prop_vals'] = [[1] * 16 for _ in range(8)]
    given['es_series'] = np.array(series)
    given['es_cfgs'] = [(0, 1, 10, 10), (0, 1, 1, 2), (5, 2, 3, 4), (0, 3, 2, 3), (40, 1, 5, 2), (0, 1, 4, 9)]
    given['nds_p'] = np.concatenate([rs.uniform(0, 1, 29), [0.0, 1.0, 0.5]])
    return given


def sha256(b):
    return hashlib.sha256(b).hexdigest()


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference is not here: this script runs in the build container only")
    for name in MODULES + SCRIPTS:
        SOURCES[name] = open(os.path.join(REF, name + '.py')).read()
    work = tempfile.mkdtemp(prefix='refrun_', dir='/tmp')
    try:
        shutil.copytree(os.path.join(HERE, 'demo'), os.path.join(work, 'demo'))
        given = oracle_inputs(os.path.join(work, 'demo'))
        uid = 65534 if os.getuid() == 0 else None
        if uid is not None:
            for root, dirs, files in os.walk(work):
                os.chown(root, uid, uid)
                for f in files:
                    os.chown(os.path.join(root, f), uid, uid)

        def drop():
            if uid is not None:
                os.setgroups([])
                os.setgid(uid)
                os.setuid(uid)
        # the child gets this file's own text (it cannot read /root/repo any more), the sources and the inputs over the pipe
        me = open(os.path.abspath(__file__)).read()
        r = subprocess.run([sys.executable, '-I', '-c', me, '--child', work], input=pickle.dumps({'sources': SOURCES, 'given': given}),
                           cwd=work, env={'PATH': '/usr/bin:/bin', 'HOME': work, 'TMPDIR': work}, preexec_fn=drop)
        if r.returncode:
            raise SystemExit('the child failed (%d)' % r.returncode)
        res = np.load(os.path.join(work, 'out.npz'), allow_pickle=False)
        out = {k: res[k] for k in res.files}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'ref_run.npz'), **out)
    with open(os.path.join(HERE, 'ref_run.manifest.txt'), 'w') as f:
        f.write('# written by tests/golden/make_golden_ref.py; the reference files that were run (sha256 of the text), then every array\n')
        for name in MODULES + SCRIPTS:
            f.write('reference python/%s.py %s\n' % (name, sha256(SOURCES[name].encode())))
        f.write('child_uid %s\n' % ('nobody (65534)' if uid is not None else 'unchanged'))
        for k in sorted(out):
            a = np.ascontiguousarray(out[k])
            f.write('array %s %s %s %s\n' % (k, a.dtype.str, json.dumps(list(a.shape)), sha256(a.tobytes())))
    print('wrote', os.path.join(HERE, 'ref_run.npz'), 'with', len(out), 'arrays')


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        main()
