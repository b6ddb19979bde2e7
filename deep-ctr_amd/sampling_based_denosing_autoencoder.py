"""Host mirror of the reference's python/sampling_based_denosing_autoencoder.py: `get_da_weights`
(:347-371) with the Theano autoencoders replaced by the HIP kernels behind include/dae_hip.h --
the sparse first layer (`sparse_da`, :234-345), the dense upper layers (`da`, :116-232) and the
lower-layer propagation inside `da` (:163-187).  Same name, arguments and return value; the file
name keeps the reference's spelling.

What the reference's code does is kept, including what it probably did not intend (each is a
tested behaviour of oracle/dae_oracle.py, Q1-Q5 there): the sparse layer's W comes back un-trained
(only its hidden bias learns), both trainers return the state before the last example's update,
layer-0 propagation is a running sum over hidden units, a sigmoid follows every lower layer.
Random numbers are drawn on the host from RandomState(123) in the reference's order and handed to
the kernels."""
import ctypes as C

import numpy as np

from . import _capi
from .sampling_based_gaussian_binary_rbm_sparse import parse_lines

rng = np.random                       # :12-13
rng.seed(1234)


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError("dae_hip error %d: %s" % (rc, (lib.dae_last_error() or b'').decode()))


def _bound(a, b):
    return 4 * np.sqrt(6. / (a + b))


def sampled_visibles(rs, lines, row, k=2):
    """:300-311 for every line of one epoch: per feature k-1 draws int(uniform(a, id)), a = previous
    id + 1, kept when not already sampled; then the feature itself.  One uniform per feature in file
    order, so the stream is drawn in one call and the arithmetic `a + (id - a) * u` vectorised."""
    assert k == 2
    nf = np.array([len(l[0]) for l in lines])
    ids = np.concatenate([np.asarray(l[0], np.int64) for l in lines])
    vals = np.concatenate([np.asarray(l[1], np.float32) for l in lines])
    first = np.zeros(len(ids), bool)
    first[np.concatenate([[0], np.cumsum(nf)[:-1]])] = True
    a = np.where(first, 0, np.roll(ids, 1) + 1).astype(np.float64)
    u = rs.random_sample(len(ids))
    neg = (a + (ids.astype(np.float64) - a) * u).astype(np.int64)       # int(): truncation
    idx = np.zeros((len(lines), row), np.int32)
    x = np.zeros((len(lines), row), np.float32)
    off = 0
    for n, m in enumerate(nf):
        li, lv, ln = ids[off:off + m], vals[off:off + m], neg[off:off + m]
        off += m
        if m * 2 == row and np.all(np.diff(li) > 0) and li[0] >= 0:     # ascending ids: no negative can repeat
            idx[n, 0::2], idx[n, 1::2] = ln, li
            x[n, 1::2] = lv
            continue
        xs, ix = [], []
        for f, v, s in zip(li.tolist(), lv.tolist(), ln.tolist()):
            if s not in ix:
                xs.append(0.)
                ix.append(s)
            xs.append(v)
            ix.append(f)
        if len(ix) != row:
            raise ValueError("line %d gives %d sampled visibles, the sparse autoencoder needs exactly %d "
                             "(Theano shape error in the reference)" % (n + 1, len(ix), row))
        idx[n], x[n] = ix, xs
    return idx, x


def get_da_weights(file, arr, ncases, num_feats=16, batch_size=100000, epochs=3, learning_rate=0.1, device=0, precision='f64'):
    """:347-371.  arr = [x_dim, H0, H1, H2]; returns [W0, b0, W1, b1, W2, b2] (float64 arrays, as
    python/SNN_DAE.py:83-86 pickles them).  precision 'f64' (default) is the reference's own
    (theano.config.floatX): its online lr = 0.1 dynamics are sensitive enough that only an f64 run
    tracks them over a whole pre-training; 'f32' keeps W of the dense layers in registers (faster,
    step-level parity 1e-7, trajectories drift apart after a few thousand steps)."""
    import torch
    lib = _capi.load()
    dev = torch.device('cuda', device)
    st = torch.cuda.current_stream(dev).cuda_stream
    lines = parse_lines(file)
    N, k = len(lines), 2
    n_fields = max(len(l[0]) for l in lines)

    f64 = precision == 'f64'
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    sfx = '_f64' if f64 else ''
    sparse_epoch, dense_epoch = getattr(lib, 'dae_sparse_epoch' + sfx), getattr(lib, 'dae_dense_epoch' + sfx)
    bag_cumsum = getattr(lib, 'dae_bag_cumsum_sigmoid' + sfx)

    def t32(a):                                               # host array -> device tensor of the working precision
        return torch.as_tensor(np.ascontiguousarray(a, dtype=npdt)).to(dev).contiguous()

    results, X = [], None
    for index in range(2, len(arr) + 1):
        row, col = int(arr[index - 2]), int(arr[index - 1])
        rs = np.random.RandomState(123)                                    # :119 / :238
        rs.randint(2 ** 30)                                                # the theano_rng seed
        cost = C.c_double()
        if index == 2:
            sparse_len, row = row, num_feats * k
            rs.uniform(low=-_bound(col, row), high=_bound(col, row), size=(row, col))                  # dA.initial_W (unused)
            rs.uniform(low=-_bound(sparse_len, row), high=_bound(sparse_len, row), size=(sparse_len, col))   # dA.init_sparse_W (unused)
            table = rs.uniform(low=-_bound(sparse_len, col), high=_bound(sparse_len, col), size=(sparse_len, col))   # :264-271
            rs.uniform(low=-_bound(row, col), high=_bound(row, col), size=(row, col))                  # the scratch W (:273-280)
            tab_d = t32(table)
            bh = torch.zeros(col, dtype=tdt, device=dev)
            bv = torch.zeros(row, dtype=tdt, device=dev)
            bprev = torch.zeros(col, dtype=tdt, device=dev)
            for ep in range(epochs):
                idx, x = sampled_visibles(rs, lines, row, k)
                idx_d, x_d = torch.as_tensor(idx).to(dev), t32(x)
                _check(lib, sparse_epoch(tab_d.data_ptr(), sparse_len, bh.data_ptr(), bv.data_ptr(), bprev.data_ptr(),
                                                 idx_d.data_ptr(), x_d.data_ptr(), N, col, row, learning_rate, C.byref(cost), st))
                print('Training epoch %d, cost ' % ep, cost.value / N)
            results += [table if f64 else table.astype(np.float32).astype(np.float64), bprev.cpu().numpy().astype(np.float64)]
            act = np.full((N, n_fields), -1, np.int32)                     # ALL ids of the line (:173-176)
            for n, (ids, _) in enumerate(lines):
                act[n, :len(ids)] = ids
            X = torch.empty((N, col), dtype=tdt, device=dev)
            b0_d = t32(results[1])
            act_d = torch.as_tensor(act).to(dev)
            _check(lib, bag_cumsum(tab_d.data_ptr(), b0_d.data_ptr(), col, sparse_len, act_d.data_ptr(), N,
                                                   n_fields, X.data_ptr(), st))
        else:
            W = t32(rs.uniform(low=-_bound(col, row), high=_bound(col, row), size=(row, col)))
            bh = torch.zeros(col, dtype=tdt, device=dev)
            bv = torch.zeros(row, dtype=tdt, device=dev)
            for ep in range(epochs):
                _check(lib, dense_epoch(W.data_ptr(), bh.data_ptr(), bv.data_ptr(), X.data_ptr(), N, row, col,
                                                learning_rate, 1 if ep == epochs - 1 else 0, C.byref(cost), st))
                print('Training epoch %d, cost ' % ep, cost.value / N)
            results += [W.cpu().numpy().astype(np.float64), bh.cpu().numpy().astype(np.float64)]
            if index < len(arr):                                           # input of the next layer: sigmoid(X W + b)
                Xn = torch.empty((N, col), dtype=tdt, device=dev)
                if f64:
                    _check(lib, lib.dae_affine_sigmoid_f64(X.data_ptr(), W.data_ptr(), bh.data_ptr(), N, row, col, Xn.data_ptr(), st))
                else:
                    _check_rbm(lib, lib.rbm_affine(X.data_ptr(), W.data_ptr(), bh.data_ptr(), N, row, col, Xn.data_ptr(), st))
                    _check_rbm(lib, lib.rbm_sigmoid(Xn.data_ptr(), Xn.numel(), st))
                X = Xn
    torch.cuda.synchronize(dev)
    return results


def _check_rbm(lib, rc):
    if rc != 0:
        raise RuntimeError("rbm_hip error %d: %s" % (rc, (lib.rbm_last_error() or b'').decode()))
