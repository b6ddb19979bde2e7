"""ORACLE (test infrastructure only): the denoising-autoencoder pre-trainer of the reference,
python/sampling_based_denosing_autoencoder.py, restated in NumPy float64 (Theano's default floatX).
parity unpinned: the reference ships no tests or data; pinned by finite differences of the dA step
and by the legacy-NumPy RandomState(123) known answers the restatement shares with it.

  dA.get_cost_updates   :96-113   -> da_step
  sparse_da             :234-345  (batch_size = 1, k = 2: one sampled negative per feature)
  da                    :116-232  (batch_size = 1)
  get_da_weights        :347-371

Quirks kept (each one is what the code does, not what it seems to intend):
 Q1 sparse_da's returned W is the UN-TRAINED random table: `givens=[(da.W, ww)]` (:253-258) makes the
    output `w` the input `ww`, and the update of the shared W is never read back (:327-332) -- only the
    hidden bias learns.  The visible bias is positional (one per sampled slot, :61-63), like the RBM's
    momentum buffer.
 Q2 both trainers return the outputs of the LAST train call, which Theano evaluates BEFORE that call's
    updates: (w, b) = state before the last example's step.
 Q3 da()'s layer-0 propagation never resets `sum` between hidden units (:173-178): unit k receives the
    running sum over units 0..k of the bag sums, over ALL ids of the line (values ignored).
 Q4 unlike the RBM trainer, da() applies a sigmoid after EVERY lower layer (:187).
 Q5 the dA constructor draws an unused W (and an unused sparse_W) from the same RandomState first (:37-54).
"""
import numpy as np


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def parse(path):
    """`y id:val ...` split on single spaces (:151-156, :297-311): [(ids, vals)] per non-blank line."""
    out = []
    with open(path) as ins:
        for line in ins:
            if line.strip() != "":
                s = line.strip().replace(':', ' ').split(' ')
                out.append(([int(s[f]) for f in range(1, len(s), 2)], [int(s[f + 1]) for f in range(1, len(s), 2)]))
    return out


def da_cost(W, b, bvis, x):
    """:101-106 with corruption_level = 0, one example: cost = L."""
    y = sigmoid(x @ W + b)
    z = sigmoid(y @ W.T + bvis)
    return -np.sum(x * np.log(z) + (1 - x) * np.log(1 - z)), y, z


def da_grads(W, b, bvis, x):
    cost, y, z = da_cost(W, b, bvis, x)
    d = z - x                                   # dL / d(pre-sigmoid z)
    dy = (d @ W) * y * (1 - y)                  # dL / d(pre-sigmoid y)
    gW = np.outer(x, dy) + np.outer(d, y)       # tied weights: encoder + decoder paths
    return cost, gW, dy, d


def _u(rng, bound, shape):
    return rng.uniform(low=-bound, high=bound, size=shape)


def sample_negatives(rng, ids, vals, k=2):
    """:304-311: per feature, k-1 draws int(uniform(a, id)) with a = previous id + 1."""
    x, indexes, a = [], [], 0
    for f, v in zip(ids, vals):
        for _ in range(k - 1):
            new_sample = int(rng.uniform(a, f))
            if new_sample not in indexes:
                x.append(0)
                indexes.append(new_sample)
        x.append(v)
        a = f + 1
        indexes.append(f)
    return x, indexes


def sparse_da(row, col, lines, sparse_len, epochs=3, lr=0.1, k=2):
    rng = np.random.RandomState(123)
    rng.randint(2 ** 30)                                            # theano_rng seed (:239)
    _u(rng, 4 * np.sqrt(6. / (col + row)), (row, col))              # Q5: dA.initial_W
    _u(rng, 4 * np.sqrt(6. / (sparse_len + row)), (sparse_len, col))   # Q5: dA.init_sparse_W
    table = _u(rng, 4 * np.sqrt(6. / (sparse_len + col)), (sparse_len, col))        # :264-271
    _u(rng, 4 * np.sqrt(6. / (row + col)), (row, col))              # the scratch `initial_W` (:273-280)
    b, bvis = np.zeros(col), np.zeros(row)
    b_pre, costs = b.copy(), []
    for _ in range(epochs):
        c = []
        for ids, vals in lines:
            x, indexes = sample_negatives(rng, ids, vals, k)
            if len(indexes) != row:
                raise ValueError("a line gives %d sampled visibles, the graph needs exactly %d" % (len(indexes), row))
            W = table[indexes]                                      # givens: da.W := the gathered rows
            cost, _, dy, d = da_grads(W, b, bvis, np.asarray(x, np.float64))
            b_pre = b.copy()                                        # Q2
            b = b - lr * dy
            bvis = bvis - lr * d
            c.append(cost)                                          # Q1: the table is written back unchanged
        costs.append(float(np.mean(c)))
    return table, b_pre, {'b': b, 'bvis': bvis, 'costs': costs}


def propagate(results, ids):
    """:163-187 for one line: Q3 running sum at layer 0, Q4 sigmoid after every layer."""
    W0, b0 = results[0], results[1]
    bag = np.zeros(W0.shape[1])
    for r in ids:
        bag = bag + W0[r]
    h = sigmoid(np.cumsum(bag) + b0)
    for i in range(2, len(results), 2):
        h = sigmoid(h @ results[i] + results[i + 1])
    return h


def da(row, col, lines, results, epochs=3, lr=0.1):
    rng = np.random.RandomState(123)
    rng.randint(2 ** 30)
    W = _u(rng, 4 * np.sqrt(6. / (col + row)), (row, col))
    b, bvis = np.zeros(col), np.zeros(row)
    W_pre, b_pre, costs = W.copy(), b.copy(), []
    for _ in range(epochs):
        c = []
        for ids, _vals in lines:
            x = propagate(results, ids)
            cost, gW, dy, d = da_grads(W, b, bvis, x)
            W_pre, b_pre = W.copy(), b.copy()                       # Q2
            W = W - lr * gW
            b = b - lr * dy
            bvis = bvis - lr * d
            c.append(cost)
        costs.append(float(np.mean(c)))
    return W_pre, b_pre, {'W': W, 'b': b, 'bvis': bvis, 'costs': costs}


def get_da_weights(lines, arr, num_feats=16):
    """:347-371: arr = [x_dim, H0, H1, H2] -> [W0, b0, W1, b1, W2, b2]."""
    k = 2
    results = []
    for index in range(2, len(arr) + 1):
        row, col = int(arr[index - 2]), int(arr[index - 1])
        if index == 2:
            w, b, _ = sparse_da(num_feats * k, col, lines, sparse_len=row, k=k)
        else:
            w, b, _ = da(row, col, lines, results)
        results += [w, b]
    return results
