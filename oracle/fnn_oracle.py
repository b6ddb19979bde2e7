"""CPU oracle for the FNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 NumPy restatement of the Theano FNN script of Atomu2014/deep-ctr
(`python/FNN_wnzh.py`, the upstream `FNN.py`).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`deep-ctr_amd/`) never does.

PINNED for A1-A3 (parse_fm_model, parse_line, feats_to_layer_one_array, gather): the
reference's data_fm.DataFM is plain Python + NumPy and was run in the build container
(tests/golden/make_golden_ref.py); tests/test_oracle_vs_reference.py holds these functions
to its outputs, bit for bit.
PARITY UNPINNED for the rest (A4-A6, A8, A10, A11's FNN form): the reference ships no tests,
golden vectors or fixtures and its Theano scripts cannot be executed in the build
container (Theano absent).  Those parts are pinned only by
  * the known-answer anchors of the legacy NumPy RNG it shares with the
    reference (`tests/test_oracle.py::test_rng_known_answers`),
  * finite-difference checks of every gradient it returns, and
  * the closed form of the sequential sparse-row update.
Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
from __future__ import annotations

import math

import numpy as np

N_FIELDS = 16
# python/FNN_wnzh.py:51-53 (same dict in python/data_fm.py:17-18)
NAME_FIELD = {'weekday': 0, 'hour': 1, 'useragent': 2, 'IP': 3, 'region': 4, 'city': 5,
              'adexchange': 6, 'domain': 7, 'slotid': 8, 'slotwidth': 9, 'slotheight': 10,
              'slotvisibility': 11, 'slotformat': 12, 'creative': 13, 'advertiser': 14,
              'slotprice': 15}


# --------------------------------------------------------------------------- A1
def parse_fm_model(path):
    """python/FNN_wnzh.py:62-84 == python/data_fm.py:15-44.

    Line 1: `w_0 feat_num rank`; others: `feat w v_1..v_rank <fieldname>:<rest>`.
    Returns (w_0, k, xdim, feat_weights{feat: [k floats]}, feat_field{feat: 0..15}).
    """
    feat_field, feat_weights = {}, {}
    w_0, k, xdim = 0.0, 0, 0
    first = True
    with open(path, 'r') as fi:
        for line in fi:
            s = line.strip().split()
            if first:
                first = False
                w_0 = float(s[0])
                k = int(s[2]) + 1                      # w and v   (:76)
                xdim = 1 + len(NAME_FIELD) * k         # (:77)
            else:
                feat = int(s[0])
                feat_weights[feat] = [float(s[1 + i]) for i in range(k)]
                name = s[1 + k][0:s[1 + k].index(':')]
                feat_field[feat] = NAME_FIELD[name]    # KeyError on unknown name
    return w_0, k, xdim, feat_weights, feat_field


# --------------------------------------------------------------------------- A2
def parse_line(line):
    """python/FNN_wnzh.py:240-253: `y id:val id:val ...`; values ignored."""
    s = line.replace(':', ' ').split()
    y = int(s[0])
    feats = [int(s[j]) for j in range(1, len(s), 2)]
    return feats, y


# --------------------------------------------------------------------------- A3
def feats_to_layer_one_array(feats, w_0, k, xdim, feat_weights, feat_field):
    """python/FNN_wnzh.py:87-96.  Last feature of a field wins; absent field = 0."""
    x = np.zeros(xdim)
    x[0] = w_0
    for feat in feats:
        lo = 1 + feat_field[feat] * k
        x[lo:lo + k] = feat_weights[feat]
    return x


def gather(rows, ids, w_0):
    """Array form of A3.  rows [D,K] f64, ids [B,F] int (slot f = field f, -1 = empty).

    x[t,0]=w_0; x[t,1+f*K:1+(f+1)*K] = rows[ids[t,f]]  (python/FNN_wnzh.py:91-96).
    """
    B, F = ids.shape
    K = rows.shape[1]
    x = np.zeros((B, 1 + F * K))
    x[:, 0] = w_0
    for f in range(F):
        m = ids[:, f] >= 0
        x[m, 1 + f * K:1 + (f + 1) * K] = rows[ids[m, f]]
    return x


# --------------------------------------------------------------------------- A11
def init_fnn_weights(xdim, hidden1, hidden2, acti_type='tanh', seed=1234):
    """python/FNN_wnzh.py:16-17,106-130,140.  Glorot-uniform, x4 when tanh;
    biases 0; w3 = 0; b3 = 0.  Legacy global-RNG stream, draw order w1 then w2."""
    rng = np.random.RandomState(seed)
    w = rng.uniform(low=-np.sqrt(6. / (xdim + hidden1)), high=np.sqrt(6. / (xdim + hidden1)),
                    size=(xdim, hidden1))
    if acti_type == 'sigmoid':
        ww1 = np.asarray(w)
    elif acti_type == 'tanh':
        ww1 = np.asarray(w * 4)
    else:
        ww1 = np.asarray(rng.uniform(-1, 1, size=(xdim, hidden1)))
    v = rng.uniform(low=-np.sqrt(6. / (hidden1 + hidden2)), high=np.sqrt(6. / (hidden1 + hidden2)),
                    size=(hidden1, hidden2))
    if acti_type == 'sigmoid':
        ww2 = np.asarray(v)
    elif acti_type == 'tanh':
        ww2 = np.asarray(v * 4)
    else:
        ww2 = np.asarray(rng.uniform(-1, 1, size=(hidden1, hidden2)))
    return {'w1': ww1, 'b1': np.zeros(hidden1), 'w2': ww2, 'b2': np.zeros(hidden2),
            'w3': np.zeros(hidden2), 'b3': 0.0}


class TheanoMaskStream(object):
    """Dropout-mask source of python/FNN_wnzh.py:15,144,154,166.

    `RandomStreams(seed=234)` hands each `binomial` op its own
    `RandomState(seedgen.randint(2**30))` in creation order (r0, r1, r2); each
    `train` call draws `binomial(n=1, p, size=(1,H))` once per live op.  r0 is
    dead (graph cut at the supplied input) but still
    consumes the first seed.  Recalled from Theano's shared_randomstreams
    source, not executable here -- masks are inputs at the C-ABI boundary.
    """

    def __init__(self, hidden1, hidden2, dropout, seed=234, has_r0=True):
        seedgen = np.random.RandomState(seed)
        self.seeds = [int(seedgen.randint(2 ** 30)) for _ in range(3 if has_r0 else 2)]
        s1, s2 = self.seeds[-2], self.seeds[-1]
        self._r1 = np.random.RandomState(s1)
        self._r2 = np.random.RandomState(s2)
        self.h1, self.h2, self.p = hidden1, hidden2, dropout

    def next(self):
        r1 = self._r1.binomial(n=1, p=self.p, size=(1, self.h1))[0].astype(np.float64)
        r2 = self._r2.binomial(n=1, p=self.p, size=(1, self.h2))[0].astype(np.float64)
        return r1, r2


# --------------------------------------------------------------------------- A4 / A4'
def _act(z, acti_type):
    if acti_type == 'sigmoid':
        return 1 / (1 + np.exp(-z))
    if acti_type == 'linear':
        return z
    return np.tanh(z)


def forward_train(p, x, r1, r2, acti_type='tanh'):
    """python/FNN_wnzh.py:147-169.  Masks are (H,) rows broadcast over the batch,
    no 1/p rescale; second layer is tanh unconditionally (:165)."""
    h1 = _act(x @ p['w1'] + p['b1'], acti_type)
    d1 = h1 * r1
    t2 = np.tanh(d1 @ p['w2'] + p['b2'])
    d2 = t2 * r2
    p_drop = 1 / (1 + np.exp(-(d2 @ p['w3']) - p['b3']))
    return h1, d1, t2, d2, p_drop


def predict(p, x, acti_type='tanh'):
    """python/FNN_wnzh.py:147-163,170-171,183.  No masks, no (1-p) scaling."""
    h1 = _act(x @ p['w1'] + p['b1'], acti_type)
    h2 = _act(h1 @ p['w2'] + p['b2'], acti_type)
    return 1 / (1 + np.exp(-(h2 @ p['w3']) - p['b3']))


# --------------------------------------------------------------------------- A5
def _dact(h, acti_type):
    if acti_type == 'sigmoid':
        return h * (1 - h)
    if acti_type == 'linear':
        return np.ones_like(h)
    return 1 - h * h


def loss_and_grads(p, x, y, r1, r2, lambda1=0.0, acti_type='tanh', reg_all=False):
    """python/FNN_wnzh.py:172-174.  cost = sum(xent) + lambda1*(sum w3^2 + b3^2)
    (`reg_all`: all six tensors, python/SNN_RBM.py:141-143).  Returns
    (xent_sum, p_drop, grads dict incl. 'x')."""
    h1, d1, t2, d2, p_drop = forward_train(p, x, r1, r2, acti_type)
    y = np.asarray(y, dtype=np.float64)
    xent = -y * np.log(p_drop) - (1 - y) * np.log(1 - p_drop)
    d3 = p_drop - y                                    # dcost/dz3
    gw3 = d2.T @ d3 + 2 * lambda1 * p['w3']
    gb3 = d3.sum() + 2 * lambda1 * p['b3']
    dl2 = np.outer(d3, p['w3']) * r2 * (1 - t2 * t2)
    gw2 = d1.T @ dl2
    gb2 = dl2.sum(axis=0)
    dl1 = (dl2 @ p['w2'].T) * r1 * _dact(h1, acti_type)
    gw1 = x.T @ dl1
    gb1 = dl1.sum(axis=0)
    gx = dl1 @ p['w1'].T
    if reg_all:
        gw2 = gw2 + 2 * lambda1 * p['w2']
        gb2 = gb2 + 2 * lambda1 * p['b2']
        gw1 = gw1 + 2 * lambda1 * p['w1']
        gb1 = gb1 + 2 * lambda1 * p['b1']
    g = {'w1': gw1, 'b1': gb1, 'w2': gw2, 'b2': gb2, 'w3': gw3, 'b3': gb3, 'x': gx}
    return float(xent.sum()), p_drop, g


def train_call(p, x, y, r1, r2, lr, lambda1=0.0, acti_type='tanh', reg_all=False):
    """The compiled `train(x, y)` of python/FNN_wnzh.py:177-182: returns gx and the
    PRE-update dense tensors, then applies theta <- theta - lr*g in place."""
    loss, p_drop, g = loss_and_grads(p, x, y, r1, r2, lambda1, acti_type, reg_all)
    pre = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else float(v))
           for k, v in p.items()}
    for name in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3'):
        p[name] = p[name] - lr * g[name]
    return g['x'], pre, loss, p_drop, g


# --------------------------------------------------------------------------- A6
def scatter_sgd(rows, ids, gx, lr, lambda_fm, b_size=None):
    """python/FNN_wnzh.py:299-306, in place on rows [D,K].

    Sequential over examples t (file order), then that example's features, then
    l<K:  row[l] <- row[l]*(1 - 2*lambda_fm*lr/b_size) - lr*gx[t][1+field*K+l].
    b_size = number of lines of this batch (pass the GLOBAL batch under DP).
    """
    B, F = ids.shape
    K = rows.shape[1]
    if b_size is None:
        b_size = B
    c = (1 - 2. * lambda_fm * lr / b_size)
    for t in range(B):
        gxt = gx[t]
        for f in range(F):
            r = ids[t, f]
            if r < 0:
                continue
            rows[r] = rows[r] * c - lr * gxt[1 + f * K:1 + (f + 1) * K] * 1
    return rows


def scatter_sgd_feats(rows, feats, row_of, field_of, gx, lr, lambda_fm, b_size=None):
    """python/FNN_wnzh.py:299-306 on the reference's own data structure: `feats[t]` is the list of feature ids of line t
    IN LINE ORDER (what get_batch_data returns as `f`), every one of which is visited -- also a feature that a later
    feature of the same field has shadowed in the layer-one array (:91-96 keeps the last), and a feature listed twice:
        for t: for feat in f[t]: for l: w[feat][l] = w[feat][l] * (1 - 2 lambda_fm lr / b_size) - lr * gx[t][1 + field(feat) k + l]
    row_of / field_of: feature id -> row of `rows` / field index.  In place on rows [D, K]."""
    K = rows.shape[1]
    if b_size is None:
        b_size = len(feats)
    c = (1 - 2. * lambda_fm * lr / b_size)
    for t in range(len(feats)):
        gxt = gx[t]
        for feat in feats[t]:
            r, fld = row_of[feat], field_of[feat]
            for l in range(K):
                rows[r][l] = rows[r][l] * c - lr * gxt[1 + fld * K + l] * 1
    return rows


def train_step_feats(p, rows, w_0, feats, row_of, field_of, n_fields, y, r1, r2, lr, lambda1, lambda_fm,
                     acti_type='tanh', b_size=None):
    """The hot loop body (python/FNN_wnzh.py:296-306) on feature lists: gather with "last feature of a field wins"
    (:91-96), train, then the update loop over EVERY listed feature.  Mutates p and rows."""
    ids = np.full((len(feats), n_fields), -1, dtype=np.int64)
    for t, ft in enumerate(feats):
        for feat in ft:
            ids[t, field_of[feat]] = row_of[feat]
    x = gather(rows, ids, w_0)
    gx, pre, loss, p_drop, g = train_call(p, x, y, r1, r2, lr, lambda1, acti_type)
    scatter_sgd_feats(rows, feats, row_of, field_of, gx, lr, lambda_fm, b_size)
    return {'x': x, 'gx': gx, 'loss': loss, 'p_drop': p_drop, 'grads': g, 'pre': pre, 'ids': ids}


def scatter_sgd_closed_form(rows, ids, gx, lr, lambda_fm, b_size=None):
    """Closed form of A6 used to cross-check it: a row hit by m slot-grads g_1..g_m
    (in example order) ends at row*c^m - lr*sum_j g_j*c^(m-j)."""
    B, F = ids.shape
    K = rows.shape[1]
    if b_size is None:
        b_size = B
    c = (1 - 2. * lambda_fm * lr / b_size)
    out = rows.copy()
    touched = {}
    for t in range(B):
        for f in range(F):
            r = int(ids[t, f])
            if r >= 0:
                touched.setdefault(r, []).append(gx[t, 1 + f * K:1 + (f + 1) * K])
    for r, gs in touched.items():
        m = len(gs)
        acc = rows[r] * c ** m
        for j, g in enumerate(gs, start=1):
            acc = acc - lr * g * c ** (m - j)
        out[r] = acc
    return out


def train_step(p, rows, w_0, ids, y, r1, r2, lr, lambda1, lambda_fm,
               acti_type='tanh', b_size=None):
    """One pass of the hot loop body, python/FNN_wnzh.py:296-306:
    gather (A3) -> train (A4,A5) -> sparse-row SGD (A6).  Mutates p and rows."""
    x = gather(rows, ids, w_0)
    gx, pre, loss, p_drop, g = train_call(p, x, y, r1, r2, lr, lambda1, acti_type)
    scatter_sgd(rows, ids, gx, lr, lambda_fm, b_size)
    return {'x': x, 'gx': gx, 'loss': loss, 'p_drop': p_drop, 'grads': g, 'pre': pre}


# --------------------------------------------------------------------------- A10
def roc_auc(y, p):
    """sklearn.metrics.roc_auc_score semantics (python/FNN_wnzh.py:219): rank
    statistic with average ranks for ties."""
    y = np.asarray(y)
    p = np.asarray(p, dtype=np.float64)
    order = np.argsort(p, kind='mergesort')
    ps = p[order]
    ranks = np.empty(len(p), dtype=np.float64)
    i = 0
    n = len(p)
    while i < n:
        j = i
        while j + 1 < n and ps[j + 1] == ps[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    npos = float((y == 1).sum())
    nneg = float(len(y) - npos)
    if npos == 0 or nneg == 0:
        raise ValueError('Only one class present in y_true.')
    return (ranks[y == 1].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def rmse(y, p):
    """python/FNN_wnzh.py:220: sqrt(mean_squared_error(y, yp))."""
    y = np.asarray(y, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    return math.sqrt(float(np.mean((y - p) ** 2)))


def logloss(y, p, eps=1e-15):
    """sklearn.metrics.log_loss semantics as used at python/baseline.py:427-429."""
    y = np.asarray(y, dtype=np.float64)
    p = np.clip(np.asarray(p, dtype=np.float64), eps, 1 - eps)
    return float(-np.mean(y * np.log(p) + (1 - y) * np.log(1 - p)))


# --------------------------------------------------------------------------- epoch loop
def run_epochs(p, rows, w_0, train_ids, train_y, test_ids, test_y, batch_size, lr, lambda1,
               lambda_fm, dropout, epochs, hidden1, hidden2, acti_type='tanh'):
    """python/FNN_wnzh.py:290-343 without the file I/O: n_batch = floor(N/batch)
    (trailing partial batch never trained, :40,:293), per-epoch train/test
    AUC+RMSE(+logloss), early stop `times_reduce` (:329-343)."""
    masks = TheanoMaskStream(hidden1, hidden2, dropout)
    n_batch = len(train_y) // batch_size
    hist = []
    min_err, times_reduce = 0.0, 0
    for ep in range(epochs):
        for j in range(n_batch):
            sl = slice(j * batch_size, (j + 1) * batch_size)
            r1, r2 = masks.next()
            train_step(p, rows, w_0, train_ids[sl], train_y[sl], r1, r2, lr, lambda1, lambda_fm,
                       acti_type)
        ptr = predict(p, gather(rows, train_ids, w_0), acti_type)
        pte = predict(p, gather(rows, test_ids, w_0), acti_type)
        rec = {'epoch': ep,
               'train_auc': roc_auc(train_y, ptr), 'train_rmse': rmse(train_y, ptr),
               'train_logloss': logloss(train_y, ptr),
               'test_auc': roc_auc(test_y, pte), 'test_rmse': rmse(test_y, pte),
               'test_logloss': logloss(test_y, pte)}
        hist.append(rec)
        if rec['test_auc'] > min_err:
            min_err = rec['test_auc']
            if times_reduce < 3:
                times_reduce += 1
        else:
            times_reduce -= 1
        if times_reduce < 0:
            break
    return hist


# --------------------------------------------------------------------------- A8 (SNN fine-tune)
def snn_bag(ww0, bb0, ids):
    """python/SNN_RBM.py:238-262 get_fi_h1_y: x = sigmoid(sum_{f active} ww0[f] + bb0);
    ids [B,F] int, -1 = no feature (a feature whose value is not 1 does not count, :251)."""
    B, F = ids.shape
    s = np.tile(np.asarray(bb0, dtype=np.float64), (B, 1))
    for f in range(F):
        m = ids[:, f] >= 0
        s[m] += ww0[ids[m, f]]
    return 1.0 / (1.0 + np.exp(-s))


def snn_train_step(p, ww0, bb0, ids, y, r1, r2, lr, lambda1, acti_type='tanh'):
    """One pass of python/SNN_RBM.py:281-291: x from the bag (pre-update ww0/bb0), train(x, y)
    with lambda1 on all six dense tensors (:141-143), then per example, in order,
    delta = lr*gx[t]*x[t]*(1-x[t]); bb0 -= delta; ww0[f] -= delta for each active f.
    Mutates p, ww0, bb0 (bb0 must be an ndarray)."""
    x = snn_bag(ww0, bb0, ids)
    gx, pre, loss, p_drop, g = train_call(p, x, y, r1, r2, lr, lambda1, acti_type, reg_all=True)
    snn_update(ww0, bb0, ids, x, gx, lr)
    return {'x': x, 'gx': gx, 'loss': loss, 'p_drop': p_drop, 'grads': g, 'pre': pre}


def snn_update(ww0, bb0, ids, x, gx, lr):
    """The update loop of mytrain, python/SNN_RBM.py:285-291, per example in order:
    bb0 = bb0 - lr*gx[t]*x[t]*(1-x[t]); ww0[f] = ww0[f] - lr*gx[t]*x[t]*(1-x[t]) for every active feature of the line (a
    feature listed twice is visited twice).  ids [B,F], -1 = no feature.  In place (bb0 must be an ndarray).
    Pinned by a run of the reference's own loop (tests/golden/ref_run.npz snn1 / snn2)."""
    B, F = ids.shape
    for t in range(B):
        bb0 -= lr * gx[t] * x[t] * (1 - x[t])
        for f in range(F):
            r = ids[t, f]
            if r >= 0:
                ww0[r] = ww0[r] - lr * gx[t] * x[t] * (1 - x[t])
    return ww0, bb0


def snn_predict(p, ww0, bb0, ids, acti_type='tanh'):
    """python/SNN_RBM.py:162-198 auc_rmse's forward: bag -> predict."""
    return predict(p, snn_bag(ww0, bb0, ids), acti_type)


# --------------------------------------------------------------------------- vectorised CPU variant
def gather_vec(rows, ids, w_0):
    """A3 with one fancy-index gather per batch (CPU baseline variant ii)."""
    B, F = ids.shape
    K = rows.shape[1]
    x = np.empty((B, 1 + F * K))
    x[:, 0] = w_0
    g = rows[np.where(ids >= 0, ids, 0)]                       # [B, F, K]
    g[ids < 0] = 0.0
    x[:, 1:] = g.reshape(B, F * K)
    return x


def scatter_sgd_vec(rows, ids, gx, lr, lambda_fm, b_size=None):
    """A6 in closed form, vectorised: per field a stable argsort groups the (row, t) pairs, a row
    hit by m examples ends at row*c^m - lr*sum_j g_j*c^(m-j) (the reference's sequential loop,
    python/FNN_wnzh.py:299-306; checked against scatter_sgd).  In place on rows."""
    B, F = ids.shape
    K = rows.shape[1]
    c = 1 - 2. * lambda_fm * lr / (b_size if b_size is not None else B)
    for f in range(F):
        col = ids[:, f]
        live = np.nonzero(col >= 0)[0]
        if len(live) == 0:
            continue
        order = live[np.argsort(col[live], kind='stable')]
        r = col[order]
        head = np.r_[True, r[1:] != r[:-1]]
        start = np.nonzero(head)[0]
        seg = np.cumsum(head) - 1
        m = np.diff(np.r_[start, len(r)])
        pos = np.arange(len(r)) - start[seg]                    # j - 1 inside the segment
        w = c ** (m[seg] - 1 - pos)
        g = gx[order, 1 + f * K:1 + (f + 1) * K] * w[:, None]
        acc = np.add.reduceat(g, start, axis=0)
        ur = r[start]
        rows[ur] = rows[ur] * (c ** m)[:, None] - lr * acc
    return rows


def train_step_vec(p, rows, w_0, ids, y, r1, r2, lr, lambda1, lambda_fm, acti_type='tanh', b_size=None):
    """The hot loop body with batch-level NumPy/BLAS calls on all cores: what a fair CPU
    implementation of the reference's step costs (the reference itself walks Python loops)."""
    x = gather_vec(rows, ids, w_0)
    gx, pre, loss, p_drop, g = train_call(p, x, y, r1, r2, lr, lambda1, acti_type)
    scatter_sgd_vec(rows, ids, gx, lr, lambda_fm, b_size)
    return {'gx': gx, 'loss': loss, 'p_drop': p_drop}
