import sys, os
sys.path.insert(0, '.')
import numpy as np
import deep_ctr_amd
from deep_ctr_amd import sampling_based_denosing_autoencoder as da, SNN_RBM
from oracle import dae_oracle as do
demo = 'tests/golden/demo'
tr = os.path.join(demo, 'train.fm.txt')
ids, y = SNN_RBM.load_active_ids(tr)
x_dim = int(ids.max()) + 1
lines = do.parse(tr)
arr = [x_dim, 200, 300, 100]
res = da.get_da_weights(tr, arr, ncases=len(lines), num_feats=16)
w0, b0, st0 = do.sparse_da(32, 200, lines, sparse_len=x_dim)
print('oracle sparse costs', st0['costs'])
print('b0 err', np.abs(res[1] - b0).max(), 'scale', np.abs(b0).max())
W1, b1, st1 = do.da(200, 300, lines, [w0, b0])
print('oracle l1 costs', st1['costs'])
rs = np.random.RandomState(123); rs.randint(2 ** 30)
bd = 4 * np.sqrt(6. / 500)
Wi = rs.uniform(-bd, bd, (200, 300))
print('W1 err', np.abs(res[2] - W1).max(), 'moved', np.abs(W1 - Wi).max(), 'b1 err', np.abs(res[3] - b1).max(), np.abs(b1).max())
X = np.array([do.propagate([w0, b0], l[0]) for l in lines])
print('X sat frac', np.mean((X < 1e-6) | (X > 1 - 1e-6)), X.min(), X.max())
W2, b2, st2 = do.da(300, 100, lines, [w0, b0, W1, b1])
print('oracle l2 costs', st2['costs'])
print('W2 err', np.abs(res[4] - W2).max(), 'b2 err', np.abs(res[5] - b2).max(), np.abs(b2).max())
