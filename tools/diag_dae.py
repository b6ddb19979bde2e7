import sys, ctypes as C, tempfile, pathlib
sys.path.insert(0, '.')
import numpy as np, torch
import deep_ctr_amd
from deep_ctr_amd import _capi, synth
from oracle import dae_oracle as do
lib = _capi.load()
dev = torch.device('cuda', 0)
st = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.RandomState(1)
for (row, col) in ((40, 24), (100, 100), (200, 300), (300, 100)):
    for N in (1, 2, 5, 50):
        X = rng.uniform(0.05, 0.95, (N, row)).astype(np.float32)
        b = 4 * np.sqrt(6. / (row + col))
        W0 = rng.uniform(-b, b, (row, col)).astype(np.float32)
        W, bh, bv = W0.astype(np.float64), np.zeros(col), np.zeros(row)
        cs = 0
        for n in range(N):
            c, gW, dy, d = do.da_grads(W, bh, bv, X[n].astype(np.float64)); cs += c
            W, bh, bv = W - 0.1 * gW, bh - 0.1 * dy, bv - 0.1 * d
        Wd = torch.as_tensor(W0).to(dev).contiguous(); bhd = torch.zeros(col, device=dev); bvd = torch.zeros(row, device=dev)
        Xd = torch.as_tensor(X).to(dev).contiguous(); cost = C.c_double()
        rc = lib.dae_dense_epoch(Wd.data_ptr(), bhd.data_ptr(), bvd.data_ptr(), Xd.data_ptr(), N, row, col, 0.1, 0, C.byref(cost), st)
        print(row, col, N, 'rc', rc, 'dW', np.abs(Wd.cpu().numpy() - W).max(), 'dbh', np.abs(bhd.cpu().numpy() - bh).max(),
              'dbv', np.abs(bvd.cpu().numpy() - bv).max(), 'cost', cost.value, cs, 'moved', np.abs(W - W0).max())
# bag cumsum
H, n_rows, F, n = 40, 500, 16, 30
W0 = rng.standard_normal((n_rows, H)).astype(np.float32) * 0.3
b0 = rng.standard_normal(H).astype(np.float32) * 0.1
ids = rng.randint(0, n_rows, (n, F)).astype(np.int32); ids[3, 5:] = -1
out = torch.empty((n, H), device=dev)
tW, tb, ti = torch.as_tensor(W0).to(dev), torch.as_tensor(b0).to(dev), torch.as_tensor(ids).to(dev)
rc = lib.dae_bag_cumsum_sigmoid(tW.data_ptr(), tb.data_ptr(), H, n_rows, ti.data_ptr(), n, F, out.data_ptr(), st)
ref = np.array([do.sigmoid(np.cumsum(sum(W0[r].astype(np.float64) for r in row if r >= 0)) + b0) for row in ids])
print('bag rc', rc, np.abs(out.cpu().numpy() - ref).max())
