"""CPU oracle for the SNN pre-training path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 NumPy restatement of `python/sampling_based_gaussian_binary_rbm_sparse.py` of
Atomu2014/deep-ctr: the online sparse CD-1 trainer (A7, `sparse_RBM` :294-402 +
`sparse_CDTrainer.train` :413-508), the dense mini-batch CD-1 trainer (A7', `RBM` :10-124 +
`CDTrainer.train` :168-291) and the layer-wise driver `get_rbm_weights` (:510-543).

PINNED (A7, A7') against runs of the reference module itself: it is plain NumPy, so tests/golden/make_golden_ref.py executes
it in the build container (through lib2to3; see that script) and tests/test_oracle_vs_reference.py holds get_rbm_weights --
three epochs of both trainers, two configurations -- to the reference's own arrays within 1e-12.  (A8, the fine-tune step,
lives in a Theano script and stays PARITY UNPINNED: see oracle/fnn_oracle.py.)
"""
import numpy as np


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def binary_threshold(probs, rng):
    """:371-375 / :84-90: one uniform draw of probs' shape; 1 where u < p, else floor(p)."""
    samples = rng.uniform(size=probs.shape)
    out = probs.copy()
    out[samples < probs] = 1.
    return np.floor(out)


def sparse_line_dict(feats):
    """:425-437: for each feature IN LINE ORDER  x[id-1] = 0  then  x[id] = 1  (a later feature's
    id-1 can overwrite an earlier feature's 1).  Returns (sorted ids, values)."""
    x = {}
    for f in feats:
        x[int(f) - 1] = 0
        x[int(f)] = 1
    keys = sorted(x)
    return keys, [x[k] for k in keys]


def dense_line_dict(feats, vals=None):
    """get_batch_x :142-156: for each feature  x[id] = val  then  x[id-1] = 0  (the opposite
    order of the sparse trainer)."""
    x = {}
    for i, f in enumerate(feats):
        x[int(f)] = 1 if vals is None else int(vals[i])
        x[int(f) - 1] = 0
    return x


class SparseRBMState(object):
    """params ~ U(-0.1, 0.1) over one flat [W | visbias | hidbias] buffer (:299-302, :531)."""

    def __init__(self, nvis, nhid, nsparsevis, rng):
        params = rng.uniform(-1. / 10, 1. / 10, nvis * nhid + nvis + nhid)
        self.W = params[:nvis * nhid].reshape(nvis, nhid).copy()
        self.visbias = params[nvis * nhid:nvis * nhid + nvis].copy()
        self.hidbias = params[nvis * nhid + nvis:].copy()
        self.weightstep = np.zeros((nsparsevis, nhid))        # positional momentum buffer (:411)
        self.nsparsevis = nsparsevis


def sparse_cd1_example(st, keys, v, rng, weightcost=0.0002, rates=(1e-4, 1e-4, 1e-4), momentum=0.9):
    """One line of sparse_CDTrainer.train (:423-505).  keys: sorted visible ids (exactly
    nsparsevis of them, :388), v their 0/1 values.  Returns the squared error of the example."""
    vis_rate, hid_rate, w_rate = rates
    keys = list(keys)
    assert len(keys) == st.nsparsevis, "the reference's buffers need exactly nsparsevis visibles (:388)"
    v = np.asarray(v, dtype=np.float64).reshape(1, -1)
    Ws = st.W[keys]                                           # [S, H]
    hid = _sigmoid(v @ Ws + st.hidbias)                       # hid_activate(mf=True)  :340-353
    poscorr = v.T @ hid                                       # :439
    posact = hid.sum(axis=0)                                  # :440
    hid_s = binary_threshold(hid, rng)                        # :441 (one draw of size (1, H))
    vis = _sigmoid(hid_s @ Ws.T + st.visbias[keys])           # mean-field visibles      :356-366
    hid2 = _sigmoid(vis @ Ws + st.hidbias)                    # mean-field hiddens       :392-397
    step = poscorr - vis.T @ hid2                             # :447-448
    step -= weightcost * Ws                                   # :449-452
    step *= w_rate
    st.weightstep *= momentum
    st.weightstep += step
    st.W[keys] += st.weightstep                               # applied TWICE (:461-462)
    st.W[keys] += st.weightstep
    st.visbias[keys] += (v[0] - vis[0]) * vis_rate            # :472-484
    st.hidbias += (posact - hid2.sum(axis=0)) * hid_rate      # :492-495
    return float(((vis - v) ** 2).sum())


def sparse_cd1_minibatch(st, batch, rng, weightcost=0.0002, rates=(1e-4, 1e-4, 1e-4), momentum=0.9):
    """Mini-batch variant of sparse_cd1_example (include/rbm_hip.h rbm_sparse_batch; NOT the reference's schedule,
    which is online -- for a batch of one the two coincide).  batch: list of (keys, v).  Every example reads the
    parameters as they were at the start of the batch; W[f_ej] += 2 (momentum weightstep[j] + step_e[j]); the
    positional buffer becomes momentum weightstep + mean_e step_e.  Uniforms are drawn example by example, in order."""
    vis_rate, hid_rate, w_rate = rates
    W0, vb0, hb0, ws0 = st.W.copy(), st.visbias.copy(), st.hidbias.copy(), st.weightstep.copy()
    step_sum = np.zeros_like(ws0)
    hacc = np.zeros_like(hb0)
    err = 0.0
    for keys, v in batch:
        keys = list(keys)
        v = np.asarray(v, dtype=np.float64).reshape(1, -1)
        Ws = W0[keys]
        hid = _sigmoid(v @ Ws + hb0)
        hid_s = binary_threshold(hid, rng)
        vis = _sigmoid(hid_s @ Ws.T + vb0[keys])
        hid2 = _sigmoid(vis @ Ws + hb0)
        step = (v.T @ hid - vis.T @ hid2 - weightcost * Ws) * w_rate
        step_sum += step
        np.add.at(st.W, keys, 2.0 * (momentum * ws0 + step))
        np.add.at(st.visbias, keys, (v[0] - vis[0]) * vis_rate)
        hacc += (hid - hid2).sum(axis=0)
        err += float(((vis - v) ** 2).sum())
    st.weightstep = momentum * ws0 + step_sum / len(batch)
    st.hidbias = hb0 + hid_rate * hacc
    return err


def sparse_cd_train(st, lines_feats, rng, epochs=3, ncases=None, **kw):
    """sparse_CDTrainer.train over `lines_feats` (a list of feature-id lists, file order)."""
    ncases = ncases or len(lines_feats)
    mses = []
    for _ in range(epochs):
        mse = 0.0
        for feats in lines_feats:
            keys, v = sparse_line_dict(feats)
            mse += sparse_cd1_example(st, keys, v, rng, **kw) / ncases
        mses.append(mse)
    return mses


class DenseRBMState(object):
    def __init__(self, nvis, nhid, rng):
        params = rng.uniform(-1. / 10, 1. / 10, nvis * nhid + nvis + nhid)      # :538
        self.W = params[:nvis * nhid].reshape(nvis, nhid).copy()
        self.visbias = params[nvis * nhid:nvis * nhid + nvis].copy()
        self.hidbias = params[nvis * nhid + nvis:].copy()
        self.weightstep = np.zeros((nvis, nhid))


def lower_layers(results, dicts):
    """CDTrainer.train :198-218: layer 0 = sum of the rows of the ACTIVE (value 1) ids + bias;
    further layers dot + bias with NO nonlinearity in between; ONE sigmoid at the end."""
    W0, b0 = results[0], results[1]
    batch = np.zeros((len(dicts), W0.shape[1]))
    for j, x in enumerate(dicts):
        for f in x:
            if x[f] == 1:
                batch[j] += W0[f]
    batch = batch + b0
    for i in range(2, len(results), 2):
        batch = batch @ results[i] + results[i + 1]
    return _sigmoid(batch)


def dense_cd1_batch(st, batch, rng, weightcost=0.0002, rates=(1e-4, 1e-4, 1e-4), momentum=0.9):
    """One mini-batch of CDTrainer.train (:219-281).  Returns sum((vis - batch)^2)."""
    vis_rate, hid_rate, w_rate = rates
    n = batch.shape[0]
    hid = _sigmoid(batch @ st.W + st.hidbias)
    poscorr = batch.T @ hid
    posact = hid.sum(axis=0)
    hid_s = binary_threshold(hid, rng)                        # one draw of shape (n, nhid)
    vis = _sigmoid(hid_s @ st.W.T + st.visbias)
    hid2 = _sigmoid(vis @ st.W + st.hidbias)
    step = (poscorr - vis.T @ hid2) / n
    step -= weightcost * st.W
    step *= w_rate
    st.weightstep *= momentum
    st.weightstep += step
    st.W += st.weightstep
    st.visbias += (batch.sum(axis=0) - vis.sum(axis=0)) * (vis_rate / n)
    st.hidbias += (posact - hid2.sum(axis=0)) * (hid_rate / n)
    return float(((vis - batch) ** 2).sum())


def dense_cd_train(st, results, lines_feats, rng, epochs=3, minibatch=100000, ncases=None, **kw):
    """CDTrainer.train: batches of `minibatch` lines; the loop ends on the first short batch
    (:286-287) -- a line count that is an exact multiple of `minibatch` makes the reference run one
    more, EMPTY batch whose division by zero poisons the weights with NaN; not reproduced."""
    n_lines = len(lines_feats)
    assert n_lines % minibatch != 0, "reference divides by an empty batch here (NaN weights)"
    ncases = ncases or n_lines
    mses = []
    for _ in range(epochs):
        mse, off = 0.0, 0
        while True:
            dicts = [dense_line_dict(f) for f in lines_feats[off:off + minibatch]]
            batch = lower_layers(results, dicts)
            mse += dense_cd1_batch(st, batch, rng, **kw) / ncases
            off += batch.shape[0]
            if batch.shape[0] < minibatch:
                break
        mses.append(mse)
    return mses


def get_rbm_weights(lines_feats, arr, rng, batch_size=100000, epochs=3, n_sparse_vis=32):
    """:510-543: arr = [x_dim, H0, H1, H2]; layer 0 sparse online CD-1, upper layers dense CD-1.
    Returns [W0, hb0, W1, hb1, W2, hb2]."""
    results = []
    for idx in range(1, len(arr)):
        row, col = int(arr[idx - 1]), int(arr[idx])
        if idx == 1:
            st = SparseRBMState(row, col, n_sparse_vis, rng)
            sparse_cd_train(st, lines_feats, rng, epochs=epochs)
        else:
            st = DenseRBMState(row, col, rng)
            dense_cd_train(st, results, lines_feats, rng, epochs=epochs, minibatch=batch_size)
        results.append(st.W)
        results.append(st.hidbias)
    return results
