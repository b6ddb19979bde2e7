"""ORACLE (test infrastructure only -- the product never imports this): the reference's text
parsers restated in Python 3, expression by expression, as the checker of the native ingestion
(include/ctr_ingest.h).  parse_fm_model, the FNN line rules and the yzx readers are PINNED against runs of the reference's own
data_fm.DataFM / ipinyou.py (tests/golden/make_golden_ref.py -> tests/test_oracle_vs_reference.py); snn_active and pairs restate
parsers that sit inside Theano scripts / are driven by them and stay parity unpinned (the reference ships no parser tests).

  parse_fm_model   python/FNN_wnzh.py:62-84  ==  python/data_fm.py:15-44
  fnn_examples     python/FNN_wnzh.py:224-253 (get_batch_data / get_fxy over the whole file)
  snn_active       python/SNN_RBM.py:238-262 (get_fi_h1_y)
  pairs            python/sampling_based_gaussian_binary_rbm_sparse.py:142-156 (get_batch_x)
  yzx_stat / yzx_load  python/ipinyou.py:23-65
"""
import numpy as np

NAME_FIELD = {'weekday': 0, 'hour': 1, 'useragent': 2, 'IP': 3, 'region': 4, 'city': 5, 'adexchange': 6,
              'domain': 7, 'slotid': 8, 'slotwidth': 9, 'slotheight': 10, 'slotvisibility': 11,
              'slotformat': 12, 'creative': 13, 'advertiser': 14, 'slotprice': 15}      # FNN_wnzh.py:51-53


def _lines(path):
    # the reference reads through linecache / text-mode files: universal newlines
    with open(path, 'r', newline=None) as f:
        return f.read().split('\n')


def parse_fm_model(path):
    """-> w_0, k, feat_weights {feat: [k floats]}, feat_field {feat: field} (insertion order = first
    appearance; a repeated feature overwrites, as the dict assignment at :83-84 does)."""
    feat_weights, feat_field = {}, {}
    lines = _lines(path)
    s = lines[0].strip().split()
    w_0 = float(s[0]); rank = int(s[2]); k = rank + 1
    for line in lines[1:]:
        s = line.strip().split()
        if not s:
            continue
        feat = int(s[0])
        weights = [float(s[1 + i]) for i in range(k)]
        tag = s[1 + k]
        field = NAME_FIELD[tag[0:tag.index(':')]]
        feat_weights[feat] = weights
        feat_field[feat] = field
    return w_0, k, feat_weights, feat_field


def fnn_examples(path, feat_field, feat_row, n_fields=16):
    """ids [N, n_fields] (slot = field, later feature of a field wins, -1 empty), y [N]."""
    ids, ys = [], []
    for line in _lines(path):
        if line.strip() == '':
            continue
        s = line.strip().replace(':', ' ').split()
        y = int(s[0])
        feats = [int(s[j]) for j in range(1, len(s), 2)]
        row = [-1] * n_fields
        for f in feats:
            row[feat_field[f]] = feat_row[f]
        ids.append(row); ys.append(y)
    return np.asarray(ids, np.int32).reshape(len(ys), n_fields), np.asarray(ys, np.int32)


def snn_active(path, width=16):
    ids, ys = [], []
    for line in _lines(path):
        if line.strip() == '':
            continue
        s = line.strip().replace(':', ' ').split(' ')
        fi = []
        for f in range(1, len(s), 2):
            if int(s[f + 1]) == 1:
                fi.append(int(s[f]))
        ids.append(fi + [-1] * (width - len(fi))); ys.append(int(s[0]))
    return np.asarray(ids, np.int32).reshape(len(ys), width), np.asarray(ys, np.int32)


def pairs(path, width=16):
    """get_batch_x: per pair the VALUE is evaluated first, then the id (`x[int(s[f])] = int(s[f+1])`); s[0] is never read there --
    the label column returned here is int(s[0]) where that parses, else 0."""
    ids, vals, ys = [], [], []
    for line in _lines(path):
        if line.strip() == '':
            continue
        s = line.strip().replace(':', ' ').split(' ')
        a, v = [], []
        for f in range(1, len(s), 2):
            val = int(s[f + 1])
            a.append(int(s[f])); v.append(val)
        try:
            y = int(s[0])
        except ValueError:
            y = 0
        ids.append(a + [-1] * (width - len(a))); vals.append(v + [0] * (width - len(v))); ys.append(y)
    return (np.asarray(ids, np.int32).reshape(len(ys), width), np.asarray(vals, np.int32).reshape(len(ys), width),
            np.asarray(ys, np.int32))


def _yzx(line):
    fields = line.strip().split()
    return int(fields[0]), [int(tok.split(':')[0]) for tok in fields[2:]]


def yzx_stat(path):
    max_dim = max_fea = 0
    lines = _lines(path)
    if lines and lines[-1] == '':
        lines = lines[:-1]
    for line in lines:
        _, x = _yzx(line)
        max_fea = max(max_fea, len(x)); max_dim = max(max_dim, max(x))
    return max_dim, max_fea


def yzx_load(path, max_dim, max_fea):
    lines = _lines(path)
    if lines and lines[-1] == '':
        lines = lines[:-1]
    X_ind, X_val, ys = [], [], []
    for line in lines:
        y, x = _yzx(line)
        pad = max_fea - len(x)
        ys.append(y); X_ind.append(x + [max_dim] * pad); X_val.append([1] * len(x) + [0] * pad)
    return np.array(X_ind), np.array(X_val), np.array(ys)
