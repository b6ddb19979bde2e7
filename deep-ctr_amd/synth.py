"""Synthetic iPinYou-shaped data in the reference's text formats.

The reference bundles no data (its .gitignore excludes data/), so the demo set and
the throughput inputs are generated here, seeded, in the exact formats its parsers read:
  fm.model.txt   python/FNN_wnzh.py:68-84    `w_0 feat_num rank` then `feat w v.. <field>:<x>`
  *.fm.txt       python/FNN_wnzh.py:240-245  `y id:val id:val ...`
  yzx            python/ipinyou.py:50-53     `y z idx:val ...`
"""
import numpy as np

FIELD_NAMES = ['weekday', 'hour', 'useragent', 'IP', 'region', 'city', 'adexchange', 'domain',
               'slotid', 'slotwidth', 'slotheight', 'slotvisibility', 'slotformat', 'creative',
               'advertiser', 'slotprice']                       # python/FNN_wnzh.py:51-53 order

# iPinYou-like skew; IP absorbs the remainder so the total is 937,670
_IPINYOU_SIZES = [7, 24, 40, None, 35, 370, 5, 100000, 130000, 21, 14, 11, 4, 7000, 9, 4]
IPINYOU_DIMS = 937670


def field_sizes_ipinyou(total=IPINYOU_DIMS):
    rest = sum(s for s in _IPINYOU_SIZES if s is not None)
    return [s if s is not None else total - rest for s in _IPINYOU_SIZES]


def field_sizes_tiny(total=1000, n_fields=16):
    base = [7, 24, 20, 300, 35, 120, 5, 150, 200, 21, 14, 11, 4, 70, 9, 4][:n_fields]
    if total < sum(base):
        scale = float(total) / sum(base)
        base = [max(2, int(b * scale)) for b in base]
    base[3] += total - sum(base)
    assert min(base) >= 1
    return base


def zipf_ids(n, field_sizes, s=1.1, seed=1234):
    """ids int32 [n, F]: global row index per field (fields own contiguous row ranges), rank r of a
    field drawn with p(r) ~ r^-s, so duplicates inside a batch are common (exercises A6)."""
    rng = np.random.RandomState(seed)
    ids = np.empty((n, len(field_sizes)), dtype=np.int32)
    off = 0
    for f, size in enumerate(field_sizes):
        w = np.arange(1, size + 1, dtype=np.float64) ** (-s)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        r = np.searchsorted(cdf, rng.uniform(size=n), side='left')
        # decorrelate popularity from row order inside the field
        perm = rng.permutation(size)
        ids[:, f] = off + perm[np.minimum(r, size - 1)]
        off += size
    return ids


def field_of_row(field_sizes):
    return np.repeat(np.arange(len(field_sizes), dtype=np.int32), field_sizes)


def fm_table(n_rows, k, scale=0.05, seed=1234):
    rng = np.random.RandomState(seed)
    return (rng.standard_normal((n_rows, k)) * scale).astype(np.float32)


def fm_score(rows, ids, w0):
    """FM prediction used only to draw plausible labels: w0 + sum w_i + sum_{i<j} <v_i, v_j>."""
    e = rows[ids].astype(np.float64)                   # [n, F, K]
    lin = e[:, :, 0].sum(axis=1)
    v = e[:, :, 1:]
    s = v.sum(axis=1)
    pair = 0.5 * ((s * s).sum(axis=1) - (v * v).sum(axis=(1, 2)))
    return w0 + lin + pair


def labels_from_fm(rows, ids, w0, seed=1234, boost=4.0):
    rng = np.random.RandomState(seed)
    z = w0 + boost * (fm_score(rows, ids, w0) - w0)
    p = 1 / (1 + np.exp(-z))
    return (rng.uniform(size=len(p)) < p).astype(np.int32)


# --------------------------------------------------------------------------- writers
def feat_id_of_row(row, stride=3, offset=7):
    """Demo feature ids are deliberately not the row index (the reference keys dicts by feat id)."""
    return stride * np.asarray(row) + offset


def write_fm_model(path, w0, rows, fo_row, feat_ids):
    n, k = rows.shape
    with open(path, 'w') as f:
        f.write('%r %d %d\n' % (float(w0), n, k - 1))
        for i in range(n):
            vals = ' '.join(repr(float(v)) for v in rows[i])
            f.write('%d %s %s:%d\n' % (feat_ids[i], vals, FIELD_NAMES[fo_row[i]], i))


def write_fm_data(path, ids, y, feat_ids):
    with open(path, 'w') as f:
        for t in range(len(y)):
            toks = ' '.join('%d:1' % feat_ids[r] for r in ids[t] if r >= 0)
            f.write('%d %s\n' % (y[t], toks))


def write_yzx(path, ids, y):
    with open(path, 'w') as f:
        for t in range(len(y)):
            toks = ' '.join('%d:1' % r for r in ids[t] if r >= 0)
            f.write('%d %d %s\n' % (y[t], 0, toks))


def make_demo(dirname, n_train=2000, n_test=500, n_feat=1000, rank=10, seed=20260410, w0=-3.0):
    """the 'demo tiny' set (16 fields, ~1k features, one id per field)."""
    import os
    os.makedirs(dirname, exist_ok=True)
    sizes = field_sizes_tiny(n_feat)
    rows = fm_table(n_feat, rank + 1, 0.05, seed)
    fo = field_of_row(sizes)
    ids = zipf_ids(n_train + n_test, sizes, 1.1, seed + 1)
    y = labels_from_fm(rows, ids, w0, seed + 2)
    feat_ids = feat_id_of_row(np.arange(n_feat))
    write_fm_model(os.path.join(dirname, 'fm.model.txt'), w0, rows, fo, feat_ids)
    write_fm_data(os.path.join(dirname, 'train.fm.txt'), ids[:n_train], y[:n_train], feat_ids)
    write_fm_data(os.path.join(dirname, 'test.fm.txt'), ids[n_train:], y[n_train:], feat_ids)
    write_yzx(os.path.join(dirname, 'train.yzx.txt'), ids[:n_train], y[:n_train])
    return {'rows': rows, 'field_of_row': fo, 'ids': ids, 'y': y, 'w0': w0, 'sizes': sizes,
            'feat_ids': feat_ids}
