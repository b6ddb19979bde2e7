"""Where does the online sparse CD-1 pass spend its time?  Same kernel, shapes varied: table size (memory
latency: 937,670 rows = HBM, 1,000 rows = L2-resident) and hidden width (compute / LDS work)."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import deep_ctr_amd
from deep_ctr_amd import _capi
lib = _capi.load()
dev = torch.device('cuda', 0); st = torch.cuda.current_stream(dev).cuda_stream
N, S = 4096, 32
for D in (937670, 1000):
    for H in (200, 64, 8):
        rng = np.random.default_rng(1)
        base = np.sort(rng.integers(0, (D - 2) // 2, (N, 16)) * 2 + 1, axis=1)
        vid = np.empty((N, S), np.int32); vid[:, 0::2] = base - 1; vid[:, 1::2] = base; vid.sort(axis=1)
        vval = ((vid % 2) == 1).astype(np.uint8)
        W = torch.as_tensor(rng.uniform(-.1, .1, (D, H)).astype(np.float32)).to(dev)
        vb = torch.zeros(D, device=dev); hb = torch.zeros(H, device=dev); ws = torch.zeros((S, H), device=dev)
        vd, vv = torch.as_tensor(vid).to(dev), torch.as_tensor(vval).to(dev); un = torch.rand((N, H), device=dev)
        err = C.c_double()
        def run():
            rc = lib.rbm_sparse_epoch(W.data_ptr(), vb.data_ptr(), hb.data_ptr(), ws.data_ptr(), vd.data_ptr(), vv.data_ptr(), un.data_ptr(),
                                      N, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st)
            assert rc == 0
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); run(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        print('D=%7d H=%3d  %.2f us per example' % (D, H, dt / N * 1e6))
