"""The oracle's hand-derived gradients against AUTOMATIC differentiation of the same forward expressions (PyTorch autograd on
the CPU, float64).  The reference obtains its gradients from `T.grad` (python/FNN_wnzh.py:174, python/SNN_RBM.py:144) and from
TensorFlow's optimisers (python/FNN_IP_L7.py:89); neither can run here, so the closed forms the oracle writes out are held to
an independent differentiator of the forward pass as the reference states it -- beside the finite-difference checks of
tests/test_oracle.py, which are accurate to ~1e-6; these agree to ~1e-12.  CPU only; torch is test plumbing here."""
import numpy as np
import pytest
import torch

from oracle import fnn_oracle as orc
from oracle import ipnn_oracle as io

torch.set_default_dtype(torch.float64)


def T(a, grad=False):
    t = torch.tensor(np.asarray(a, dtype=np.float64))
    return t.requires_grad_(grad)


@pytest.mark.parametrize("acti,reg_all", [('tanh', False), ('sigmoid', False), ('linear', False), ('tanh', True)])
def test_fnn_gradients_equal_autograd(acti, reg_all):
    """python/FNN_wnzh.py:144-174: h1 = act(x w1 + b1) r1; d2 = tanh(d1 w2 + b2) r2 (tanh whatever acti_type, :165);
    p = sigmoid(d2 w3 + b3); cost = SUM xent + lambda1 (sum w3^2 + b3^2) (all six tensors in the SNN scripts)."""
    rng = np.random.RandomState(3)
    B, X, H1, H2, lam = 37, 23, 11, 7, 0.3
    p = {'w1': rng.randn(X, H1) * .3, 'b1': rng.randn(H1) * .1, 'w2': rng.randn(H1, H2) * .3, 'b2': rng.randn(H2) * .1,
         'w3': rng.randn(H2) * .3, 'b3': 0.2}
    x, y = rng.randn(B, X), (rng.uniform(size=B) < 0.4).astype(np.float64)
    r1, r2 = (rng.uniform(size=H1) < 0.6).astype(np.float64), (rng.uniform(size=H2) < 0.6).astype(np.float64)
    loss, p_drop, g = orc.loss_and_grads(p, x, y, r1, r2, lam, acti, reg_all)
    tp = {k: T(v, True) for k, v in p.items()}
    tx = T(x, True)
    act = {'tanh': torch.tanh, 'sigmoid': lambda z: 1 / (1 + torch.exp(-z)), 'linear': lambda z: z}[acti]
    d1 = act(tx @ tp['w1'] + tp['b1']) * T(r1)
    d2 = torch.tanh(d1 @ tp['w2'] + tp['b2']) * T(r2)
    pp = 1 / (1 + torch.exp(-(d2 @ tp['w3'] + tp['b3'])))
    xent = (-T(y) * torch.log(pp) - (1 - T(y)) * torch.log(1 - pp)).sum()
    names = ('w1', 'b1', 'w2', 'b2', 'w3', 'b3') if reg_all else ('w3', 'b3')
    cost = xent + lam * sum((tp[k] ** 2).sum() for k in names)
    cost.backward()
    assert abs(loss - float(xent.detach())) <= 1e-12 * abs(loss)
    np.testing.assert_allclose(p_drop, pp.detach().numpy(), rtol=1e-13)
    for k in p:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(g['x'], tx.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_snn_bag_update_is_the_gradient_through_the_sigmoid():
    """python/SNN_RBM.py:248-256,285-291: x = sigmoid(sum of the active rows + bb0); the row / bias update is lr * gx * x (1 - x):
    the chain rule through that sigmoid, which the script applies by hand."""
    rng = np.random.RandomState(5)
    B, F, D, H0, H1, H2 = 9, 4, 30, 8, 6, 5
    ww0, bb0 = rng.randn(D, H0) * .2, rng.randn(H0) * .1
    ids = rng.randint(0, D, size=(B, F))
    ids[2, 1] = -1
    p = {'w1': rng.randn(H0, H1) * .3, 'b1': rng.randn(H1) * .1, 'w2': rng.randn(H1, H2) * .3, 'b2': rng.randn(H2) * .1,
         'w3': rng.randn(H2) * .3, 'b3': 0.1}
    y = (rng.uniform(size=B) < 0.5).astype(np.float64)
    r1, r2 = np.ones(H1), np.ones(H2)
    lr = 0.05
    w_ref, b_ref, p_ref = ww0.copy(), bb0.copy(), {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    orc.snn_train_step(p_ref, w_ref, b_ref, ids, y, r1, r2, lr, 0.0)
    tw, tb = T(ww0, True), T(bb0, True)
    rows = torch.stack([tw[torch.tensor(ids[:, f].clip(0))] * T((ids[:, f] >= 0).astype(np.float64))[:, None] for f in range(F)]).sum(0)
    x = 1 / (1 + torch.exp(-(rows + tb)))
    tp = {k: T(v) for k, v in p.items()}
    d1 = torch.tanh(x @ tp['w1'] + tp['b1'])
    d2 = torch.tanh(d1 @ tp['w2'] + tp['b2'])
    pp = 1 / (1 + torch.exp(-(d2 @ tp['w3'] + tp['b3'])))
    (-T(y) * torch.log(pp) - (1 - T(y)) * torch.log(1 - pp)).sum().backward()
    np.testing.assert_allclose(w_ref, ww0 - lr * tw.grad.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(b_ref, bb0 - lr * tb.grad.numpy(), rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("act,reduce", [('relu', 'sum'), ('tanh', 'sum'), ('sigmoid', 'mean')])
def test_ipnn_gradients_equal_autograd(act, reduce):
    """python/FNN_IP_L7.py:102-133: z1 = [e | pair products | b]; l_{t+1} = dropout(act(l_t)) W + bias with activation and
    inverted dropout BEFORE every matmul, also on z1; loss = sum (or mean) of sigmoid cross-entropy with logits."""
    rng = np.random.RandomState(7)
    F, K, B, hidden, keep = 4, 3, 21, [9, 6], 0.7
    n_rows = 40
    table = rng.randn(n_rows, K) * .3
    ids = rng.randint(0, n_rows, size=(B, F))
    y = (rng.uniform(size=B) < 0.5).astype(np.float64)
    d = [F * K + F * (F - 1) // 2 + 1] + hidden + [1]
    params = {'b': 0.3, 'W': [rng.randn(d[i], d[i + 1]) * .3 for i in range(len(d) - 1)], 'bias': [rng.randn(d[i + 1]) * .1 for i in range(len(d) - 1)]}
    masks = [(rng.uniform(size=(B, d[t])) < keep).astype(np.float64) for t in range(len(hidden) + 1)]
    loss, logits, g = io.loss_and_grads(params, table, ids, y, act, masks, keep, reduce)
    tt, tb = T(table, True), T(np.array(params['b']), True)
    tW, tbias = [T(w, True) for w in params['W']], [T(b, True) for b in params['bias']]
    e = tt[torch.tensor(ids)]                                                    # [B, F, K]
    pairs = [(e[:, i] * e[:, j]).sum(1) for i in range(F) for j in range(i + 1, F)]
    z = torch.cat([e.reshape(B, F * K), torch.stack(pairs, 1), tb.expand(B, 1)], 1)
    fa = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[act]
    l = z
    for t in range(len(tW)):
        l = (fa(l) * T(masks[t]) / keep) @ tW[t] + tbias[t]
    lg = l[:, 0]
    xent = torch.clamp(lg, min=0) - lg * T(y) + torch.log1p(torch.exp(-lg.abs()))
    tl = xent.sum() if reduce == 'sum' else xent.mean()
    tl.backward()
    assert abs(loss - float(tl.detach())) <= 1e-12 * abs(loss)
    np.testing.assert_allclose(logits, lg.detach().numpy(), rtol=1e-12, atol=1e-13)
    for t in range(len(tW)):
        np.testing.assert_allclose(g['W'][t], tW[t].grad.numpy(), rtol=1e-9, atol=1e-12, err_msg='W%d' % t)
        np.testing.assert_allclose(g['bias'][t], tbias[t].grad.numpy(), rtol=1e-9, atol=1e-12)
    assert abs(g['b'] - float(tb.grad)) <= 1e-10
    gt = np.zeros_like(table)
    np.add.at(gt, ids, g['e'])
    np.testing.assert_allclose(gt, tt.grad.numpy(), rtol=1e-9, atol=1e-12)
