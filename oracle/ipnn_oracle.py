"""CPU oracle for the inner-product FNN family -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 NumPy restatement of the arithmetic of `python/FNN_IP_L3.py` / `FNN_IP_L5.py` /
`FNN_IP_L7.py` (TensorFlow-0.x classes of Atomu2014/deep-ctr), SURVEY.md row A9: per-field
embeddings e_i = [w | v] of width K = rank + 1 (:66-69 of FNN_IP_L7.py), pair-wise inner
products p[(i, j)] for i < j in row-major upper-triangular order (:108-111), z1 = [e | p | b]
with the bias LAST (:113-114), then for every layer  l_{t+1} = dropout(act(l_t)) W_t + b_t
-- activation and inverted dropout BEFORE each matmul, on z1 too (:115-132) -- logits = last
layer, loss = sum of sigmoid cross-entropy with logits (:82-88).  Categorical fields only (the 13
numeric Criteo fields of :103 do not exist in the iPinYou shape).  Optimiser: plain SGD, the
north_star's; the reference's Adam (baseline.py:146) is not restated.

PARITY UNPINNED (no fixtures in the reference; TensorFlow absent): pinned by finite differences
(tests/test_oracle.py).
"""
import numpy as np


def act(x, name):
    if name == 'tanh':
        return np.tanh(x)
    if name == 'relu':
        return np.maximum(x, 0.0)
    return 1.0 / (1.0 + np.exp(-x))


def dact(x, name):
    if name == 'tanh':
        return 1.0 - np.tanh(x) ** 2
    if name == 'relu':
        return (x > 0).astype(np.float64)
    s = 1.0 / (1.0 + np.exp(-x))
    return s * (1 - s)


USE_PAIRS = True      # False: z1 = [e | b], the plain TensorFlow `FNN` class (python/FNN.py:76-94)


def pairs(F):
    return [(i, j) for i in range(F - 1) for j in range(i + 1, F)] if USE_PAIRS else []


def z1_of(table, b, ids):
    """ids [B, F] (one id per field).  Returns (e [B,F,K], z1 [B, F*K + F(F-1)/2 + 1])."""
    e = table[ids]
    B, F, K = e.shape
    p = (np.stack([(e[:, i] * e[:, j]).sum(axis=1) for (i, j) in pairs(F)], axis=1) if pairs(F)
         else np.zeros((B, 0)))
    return e, np.concatenate([e.reshape(B, F * K), p, np.full((B, 1), float(b))], axis=1)


def forward(params, table, ids, act_name, masks=None, keep=1.0):
    """params: {'b': scalar, 'W': [W_1..W_{L+1}], 'bias': [b_1..b_{L+1}]}.  masks: list of L+1 0/1
    arrays (for z1 and every hidden layer) or None (drop_out=False).  Returns logits and caches."""
    e, z1 = z1_of(table, params['b'], ids)
    ls, As = [z1], []
    l = z1
    for t, (W, bias) in enumerate(zip(params['W'], params['bias'])):
        a = act(l, act_name)
        if masks is not None:
            a = a * masks[t] / keep                      # tf.nn.dropout: keep, scale by 1/keep_prob
        As.append(a)
        l = a @ W + bias
        ls.append(l)
    return l[:, 0], {'e': e, 'ls': ls, 'As': As}


def loss_and_grads(params, table, ids, y, act_name, masks=None, keep=1.0, reduce='sum'):
    """reduce: 'sum' = tf.reduce_sum(log_loss), anything else = tf.reduce_mean (python/FNN_IP_L7.py:83-86)."""
    logits, c = forward(params, table, ids, act_name, masks, keep)
    y = np.asarray(y, dtype=np.float64)
    scale = 1.0 if reduce == 'sum' else 1.0 / len(y)
    loss = float((np.maximum(logits, 0) - logits * y + np.log1p(np.exp(-np.abs(logits)))).sum()) * scale
    d = (1.0 / (1.0 + np.exp(-logits)) - y)[:, None] * scale       # d loss / d l_{L+1}
    gW, gb = [None] * len(params['W']), [None] * len(params['W'])
    for t in reversed(range(len(params['W']))):
        gW[t] = c['As'][t].T @ d
        gb[t] = d.sum(axis=0)
        da = d @ params['W'][t].T
        m = 1.0 if masks is None else masks[t] / keep
        d = da * m * dact(c['ls'][t], act_name)              # d loss / d l_t   (t = 0: z1)
    e = c['e']
    B, F, K = e.shape
    ge = d[:, :F * K].reshape(B, F, K).copy()
    for n, (i, j) in enumerate(pairs(F)):
        dp = d[:, F * K + n][:, None]
        ge[:, i] += dp * e[:, j]
        ge[:, j] += dp * e[:, i]
    return loss, logits, {'W': gW, 'bias': gb, 'b': float(d[:, -1].sum()), 'e': ge}


def sgd_step(params, table, ids, y, act_name, lr, masks=None, keep=1.0, reduce='sum'):
    """One plain-SGD step on every variable; embedding rows of a batch add up their gradients
    (the gradient through tf.concat / tf.slice is a sum).  Mutates params and table."""
    loss, logits, g = loss_and_grads(params, table, ids, y, act_name, masks, keep, reduce)
    for t in range(len(params['W'])):
        params['W'][t] = params['W'][t] - lr * g['W'][t]
        params['bias'][t] = params['bias'][t] - lr * g['bias'][t]
    params['b'] = params['b'] - lr * g['b']
    gt = np.zeros_like(table)
    np.add.at(gt, ids, g['e'])
    table -= lr * gt
    return loss, logits, g


def adam_state(params, table):
    z = lambda a: np.zeros_like(np.asarray(a, dtype=np.float64))        # noqa: E731
    return {'t': 0, 'W': [(z(w), z(w)) for w in params['W']], 'bias': [(z(b), z(b)) for b in params['bias']],
            'b': [0.0, 0.0], 'table': (z(table), z(table))}


def adam_step(params, table, ids, y, act_name, lr, st, masks=None, keep=1.0, beta1=0.9, beta2=0.999, eps=1e-8, reduce='sum'):
    """One step of TensorFlow's AdamOptimizer (python/tf_util.py:17-20; python/baseline.py:146) on EVERY
    variable: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); m <- beta1 m + (1 - beta1) g; v <- beta2 v +
    (1 - beta2) g^2; theta <- theta - lr_t m / (sqrt(v) + eps).  The table's gradient is dense (zero rows
    for untouched features), so all of its moments decay and all rows move.  Mutates params, table, st."""
    loss, logits, g = loss_and_grads(params, table, ids, y, act_name, masks, keep, reduce)
    st['t'] += 1
    lr_t = lr * np.sqrt(1 - beta2 ** st['t']) / (1 - beta1 ** st['t'])

    def upd(theta, grad, mv):
        m, v = mv
        m[...] = beta1 * m + (1 - beta1) * grad
        v[...] = beta2 * v + (1 - beta2) * grad * grad
        return theta - lr_t * m / (np.sqrt(v) + eps)
    for t in range(len(params['W'])):
        params['W'][t] = upd(params['W'][t], g['W'][t], st['W'][t])
        params['bias'][t] = upd(params['bias'][t], g['bias'][t], st['bias'][t])
    mb = [np.array(st['b'][0]), np.array(st['b'][1])]
    params['b'] = float(upd(np.array(params['b']), np.array(g['b']), mb))
    st['b'] = [float(mb[0]), float(mb[1])]
    gt = np.zeros_like(table)
    np.add.at(gt, ids, g['e'])
    table[...] = upd(table, gt, st['table'])
    return loss, logits, g


def ftrl_state(params, table, init_accum=0.1):
    """(accum, linear) per variable: TensorFlow's FtrlOptimizer slots (initial_accumulator_value 0.1, linear 0)."""
    a = lambda x: np.full_like(np.asarray(x, dtype=np.float64), init_accum)   # noqa: E731
    z = lambda x: np.zeros_like(np.asarray(x, dtype=np.float64))              # noqa: E731
    return {'W': [(a(w), z(w)) for w in params['W']], 'bias': [(a(b), z(b)) for b in params['bias']],
            'b': [init_accum, 0.0], 'table': (a(table), z(table))}


def ftrl_step(params, table, ids, y, act_name, lr, st, masks=None, keep=1.0):
    """One step of tf.train.FtrlOptimizer(learning_rate) (python/tf_util.py:21-24) with TensorFlow's defaults
    learning_rate_power = -0.5, l1 = l2 = 0 on EVERY variable (the ApplyFtrl kernel):
        new_accum = accum + g^2;  linear += g - (sqrt(new_accum) - sqrt(accum)) / lr * var;
        var = -linear / (sqrt(new_accum) / lr)  if |linear| > l1 = 0  else 0;  accum = new_accum.
    The table's gradient is dense (zero rows for untouched features), so a row that no example has touched is
    re-derived from its zero linear term: it becomes 0 at the first step.  Mutates params, table, st."""
    loss, logits, g = loss_and_grads(params, table, ids, y, act_name, masks, keep)

    def upd(theta, grad, al):
        accum, linear = al
        na = accum + grad * grad
        linear[...] = linear + grad - (np.sqrt(na) - np.sqrt(accum)) / lr * theta
        accum[...] = na
        return np.where(linear != 0.0, -linear / (np.sqrt(na) / lr), 0.0)
    for t in range(len(params['W'])):
        params['W'][t] = upd(params['W'][t], g['W'][t], st['W'][t])
        params['bias'][t] = upd(params['bias'][t], g['bias'][t], st['bias'][t])
    ab = [np.array(st['b'][0]), np.array(st['b'][1])]
    params['b'] = float(upd(np.array(params['b']), np.array(g['b']), ab))
    st['b'] = [float(ab[0]), float(ab[1])]
    gt = np.zeros_like(table)
    np.add.at(gt, ids, g['e'])
    table[...] = upd(table, gt, st['table'])
    return loss, logits, g


def predict(params, table, ids, act_name):
    logits, _ = forward(params, table, ids, act_name)
    return 1.0 / (1.0 + np.exp(-logits))
