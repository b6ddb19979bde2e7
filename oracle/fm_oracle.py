"""ORACLE (test infrastructure only): the factorisation machine of the reference's python/FM.py in
NumPy float64.  parity unpinned (the reference ships no tests; TensorFlow 0.x cannot run here).

  factorization   python/FM.py:55-64    yhat = b + sum w x + 1/2 (|sum v x|^2 - sum |v|^2 x^2)
  loss            :36-41                 sigmoid xent, 'sum' or mean, + lambda * (l2(W) + l2(V) + l2(b)),
                                         tf.nn.l2_loss(t) = sum(t^2) / 2 -> a DENSE gradient lambda * theta
  SGD             python/tf_util.py:26-29
One feature per field, value 1 (what python/ipinyou.py:42-65 feeds); ids [B, F], -1 = absent.
"""
import numpy as np


def logits(rows, b, ids):
    """rows [D, K] = concat(W, V); returns yhat [B]."""
    live = (ids >= 0)[..., None]
    g = np.where(live, rows[np.where(ids >= 0, ids, 0)], 0.0)          # [B, F, K]
    w, v = g[..., 0], g[..., 1:]
    S = v.sum(axis=1)
    return b + w.sum(axis=1) + 0.5 * ((S * S).sum(axis=1) - (v * v).sum(axis=(1, 2)))


def predict(rows, b, ids):
    return 1.0 / (1.0 + np.exp(-logits(rows, b, ids)))


def loss_value(rows, b, ids, y, lam, reduce_mean):
    z = logits(rows, b, ids)
    xent = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    data = xent.mean() if reduce_mean else xent.sum()
    return data + lam * 0.5 * ((rows * rows).sum() + b * b), data


def sgd_step(rows, b, ids, y, lr, lam, reduce_mean=True):
    """One step in place on rows; returns (new b, data loss, p before the update)."""
    B, F = ids.shape
    z = logits(rows, b, ids)
    p = 1.0 / (1.0 + np.exp(-z))
    xent = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    delta = (p - y) / (B if reduce_mean else 1.0)
    live = ids >= 0
    safe = np.where(live, ids, 0)
    g = np.where(live[..., None], rows[safe], 0.0)
    S = g[..., 1:].sum(axis=1)                                            # [B, rank]
    grad = np.zeros_like(rows)
    gw = np.broadcast_to(delta[:, None], ids.shape)
    gv = delta[:, None, None] * (S[:, None, :] - g[..., 1:])
    np.add.at(grad[:, 0], safe[live], gw[live])
    np.add.at(grad[:, 1:], safe[live], gv[live])
    rows -= lr * (grad + lam * rows)                                      # the dense L2 gradient
    b_new = b - lr * (delta.sum() + lam * b)
    return b_new, (xent.mean() if reduce_mean else xent.sum()), p
