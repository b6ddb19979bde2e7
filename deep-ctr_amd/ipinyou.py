"""Loader interface #2: the iPinYou "yzx" readers of the reference's python/ipinyou.py
(`collect`, `stat`, `load_ipinyou_data`, `feed_zero`), same names, arguments and return values.
Line format: `y z idx:val idx:val ...` (token 1, `z`, is skipped).  Host logic only; the TF FM/LR
driver under `__main__` in the reference is out of scope.
"""
import numpy as np


def collect(fin, size=100000):
    """python/ipinyou.py:11-20: up to `size` lines from the open file, shuffled in place with the
    global NumPy RNG."""
    buf = []
    for _ in range(size):
        line = next(fin, '')
        if line == '':
            break
        buf.append(line)
    np.random.shuffle(buf)
    return buf


def _indices(line):
    fields = line.strip().split()
    return int(fields[0]), [int(tok.split(':')[0]) for tok in fields[2:]]


def stat(path):
    """python/ipinyou.py:23-39: (max_dim, max_fea) over the whole file."""
    max_fea = 0
    max_dim = 0
    with open(path) as fin:
        while True:
            buf = collect(fin)
            if len(buf) < 1:
                break
            for line in buf:
                _, x_ind = _indices(line)
                max_fea = max(max_fea, len(x_ind))
                max_dim = max(max_dim, max(x_ind))
    return max_dim, max_fea


def load_ipinyou_data(fin, size, max_dim, max_fea):
    """python/ipinyou.py:42-65: next `size` lines -> X_ind [n,max_fea] int (pad id = max_dim),
    X_val [n,max_fea] (1 present, 0 pad), y [n]; (None, None, None) at EOF."""
    buf = collect(fin, size)
    if len(buf) < 1:
        return None, None, None
    X_ind, X_val, y = [], [], []
    for line in buf:
        yy, x_ind = _indices(line)
        pad = max_fea - len(x_ind)
        y.append(yy)
        X_ind.append(x_ind + [max_dim] * pad)
        X_val.append([1] * len(x_ind) + [0] * pad)
    return np.array(X_ind), np.array(X_val), np.array(y)


def feed_zero(X_ind, X_val, y, max_dim, max_fea):
    """python/ipinyou.py:68-89: pad ragged in-memory lists, then shuffle examples."""
    for i in range(len(y)):
        pad = max_fea - len(X_ind[i])
        X_ind[i].extend([max_dim] * pad)
        X_val[i].extend([0] * pad)
    X_ind = np.array(X_ind)
    X_val = np.array(X_val)
    y = np.array(y)
    inds = np.arange(len(y))
    np.random.shuffle(inds)
    return X_ind[inds], X_val[inds], y[inds]


def stat_file(path):
    """`stat` as one native pass (ctr_yzx_stat): (max_dim, max_fea).  Unlike `stat` it does not
    shuffle buffers, so it leaves the global NumPy RNG untouched."""
    from . import ingest
    md, mf, _ = ingest.yzx_stat(path)
    return md, mf


def load_ipinyou_file(path, max_dim, max_fea):
    """`load_ipinyou_data` for a whole file in one native pass (ctr_parse_yzx), FILE order: the
    reference shuffles each 10,000-line buffer with the global RNG (python/ipinyou.py:19) -- apply
    a permutation afterwards where that order matters."""
    from . import ingest
    return ingest.parse_yzx(path, max_dim, max_fea)


def to_field_ids(X_ind, X_val, field_of_row):
    """Bridge to the HIP path: padded index lists -> ids int32 [n,16] with slot = field and -1 for
    empty fields (pads have X_val == 0)."""
    n = X_ind.shape[0]
    n_fields = int(field_of_row.max()) + 1
    ids = np.full((n, n_fields), -1, dtype=np.int32)
    for j in range(X_ind.shape[1]):
        present = X_val[:, j] != 0
        rows = X_ind[present, j]
        ids[np.nonzero(present)[0], field_of_row[rows]] = rows
    return ids
