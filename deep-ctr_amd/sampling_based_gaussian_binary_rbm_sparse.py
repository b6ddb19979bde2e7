"""SNN pre-training on MI355X: same entry point as the reference's
python/sampling_based_gaussian_binary_rbm_sparse.py -- `get_rbm_weights(file, arr, ncases, ...)`
(:510-543) -- with the NumPy CD-1 trainers replaced by the HIP kernels behind include/rbm_hip.h:
the online sparse CD-1 pass (`sparse_CDTrainer.train`, :413-508) and the dense mini-batch CD-1
step (`CDTrainer.train`, :168-291).  Host logic here: line parsing with the reference's dict
semantics, parameter init and the uniform draws from the global legacy NumPy stream (so that the
random numbers are the reference's), orchestration.  No CPU compute path.
"""
import ctypes as C

import numpy as np

from . import _capi

rng = np.random            # python/...rbm_sparse.py:7-8 (seeded by the caller: dl_utils.seed_global)


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError("rbm_hip error %d: %s" % (rc, (lib.rbm_last_error() or b'').decode()))


def parse_lines(path, width=64):
    """`y id:val id:val ...` split on single spaces (:423-424); returns (ids, vals) per line.  The
    token work is the native pass (ctr_parse_examples, CTR_MODE_PAIRS)."""
    from . import ingest
    ids, vals, _ = ingest.parse_examples(path, ingest.MODE_PAIRS, None, width)
    out = []
    for a, v in zip(ids.tolist(), vals.tolist()):
        n = a.index(-1) if -1 in a else len(a)
        out.append((a[:n], v[:n]))
    return out


def sparse_inputs(lines, n_sparse_vis=32):
    """:425-437: per line, in line order, x[id-1] = 0 then x[id] = 1; sorted ids and their values.
    The reference's buffers need exactly n_sparse_vis visibles (:388)."""
    vid = np.zeros((len(lines), n_sparse_vis), dtype=np.int32)
    vval = np.zeros((len(lines), n_sparse_vis), dtype=np.uint8)
    for n, (ids, _) in enumerate(lines):
        x = {}
        for f in ids:
            x[f - 1] = 0
            x[f] = 1
        keys = sorted(x)
        if len(keys) != n_sparse_vis:
            raise ValueError("line %d has %d sampled visibles, the sparse RBM needs exactly %d (reference :388)"
                             % (n + 1, len(keys), n_sparse_vis))
        vid[n] = keys
        vval[n] = [x[k] for k in keys]
    return vid, vval


def dense_active_ids(lines, n_fields=16):
    """get_batch_x :142-156: x[id] = val then x[id-1] = 0; the ids whose value is 1 are summed
    (:205-211).  Returns int32 [N, n_fields], -1 padded."""
    out = np.full((len(lines), n_fields), -1, dtype=np.int32)
    for n, (ids, vals) in enumerate(lines):
        x = {}
        for f, v in zip(ids, vals):
            x[f] = v
            x[f - 1] = 0
        act = [f for f in x if x[f] == 1]
        out[n, :len(act)] = act
    return out


def get_rbm_weights(file, arr, ncases, fm_model_file=None, batch_size=1, epochs=3, precision='f32', device=0, sparse_minibatch=1):
    """:510-543.  arr = [x_dim, H0, H1, H2]; returns [W0, hb0, W1, hb1, W2, hb2] (NumPy float64
    arrays, as the caller pickles them, python/SNN_RBM.py:82-88).  weightcost 2e-4, rates 1e-4,
    momentum 0.9 (:405-411, :159-166).  sparse_minibatch > 1: the sparse layer in mini-batches (rbm_sparse_batch;
    NOT the reference's online schedule -- a throughput mode, see include/rbm_hip.h)."""
    import torch
    lib = _capi.load()
    dev = torch.device('cuda', device)
    st = torch.cuda.current_stream(dev).cuda_stream
    lines = parse_lines(file)
    N = len(lines)
    results = []
    wc, rate, mom = 0.0002, 1e-4, 0.9
    n_fields = max(len(l[0]) for l in lines)

    def t32(a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev).contiguous()

    X = None
    for idx in range(1, len(arr)):
        row, col = int(arr[idx - 1]), int(arr[idx])
        params = rng.uniform(-1. / 10, 1. / 10, row * col + row + col)          # :531 / :538
        W = params[:row * col].reshape(row, col)
        vb = params[row * col:row * col + row]
        hb = params[row * col + row:]
        if idx == 1:
            vid, vval = sparse_inputs(lines)
            Wd, vbd, hbd = t32(W), t32(vb), t32(hb)
            ws = torch.zeros((32, col), dtype=torch.float32, device=dev)
            vid_d = torch.as_tensor(vid).to(dev)
            vval_d = torch.as_tensor(vval).to(dev)
            for _ in range(epochs):
                unif = t32(rng.uniform(size=(N, col)))                           # one (1,H) draw per line (:441)
                err = C.c_double()
                if sparse_minibatch > 1:
                    if _ == 0:
                        dW, dvis = torch.zeros_like(Wd), torch.zeros_like(vbd)
                    _check(lib, lib.rbm_sparse_batch(Wd.data_ptr(), dW.data_ptr(), vbd.data_ptr(), dvis.data_ptr(), hbd.data_ptr(),
                                                     ws.data_ptr(), vid_d.data_ptr(), vval_d.data_ptr(), unif.data_ptr(), N,
                                                     int(sparse_minibatch), col, 32, wc, rate, rate, rate, mom, C.byref(err), st))
                else:
                    _check(lib, lib.rbm_sparse_epoch(Wd.data_ptr(), vbd.data_ptr(), hbd.data_ptr(), ws.data_ptr(),
                                                     vid_d.data_ptr(), vval_d.data_ptr(), unif.data_ptr(), N, col, 32,
                                                     wc, rate, rate, rate, mom, C.byref(err), st))
                print("Done epoch: MSE=%f" % (err.value / ncases))
            results.append(Wd.cpu().numpy().astype(np.float64))
            results.append(hbd.cpu().numpy().astype(np.float64))
            # input of the next layer: sum of the active rows + bias, no nonlinearity yet (:199-212)
            act = torch.as_tensor(dense_active_ids(lines, n_fields)).to(dev)
            X = torch.empty((N, col), dtype=torch.float32, device=dev)
            _check(lib, lib.rbm_bag_sum(Wd.data_ptr(), hbd.data_ptr(), col, row, act.data_ptr(), N, n_fields,
                                        X.data_ptr(), st))
        else:
            if batch_size < N and N % batch_size == 0:
                raise ValueError("the reference runs an EMPTY last mini-batch here and divides by zero "
                                 "(rbm_sparse.py:286-287); choose a batch_size that does not divide the line count")
            Xin = X.clone()
            _check(lib, lib.rbm_sigmoid(Xin.data_ptr(), Xin.numel(), st))        # ONE sigmoid at the end (:218)
            h = C.c_void_p()
            mb = min(batch_size, N)
            _check(lib, lib.rbm_dense_create(row, col, mb, 1 if precision == 'bf16' else 0, device, st, C.byref(h)))
            W32, vb32, hb32 = (np.ascontiguousarray(a, dtype=np.float32) for a in (W, vb, hb))
            _check(lib, lib.rbm_dense_set(h, W32.ctypes.data, vb32.ctypes.data, hb32.ctypes.data))
            for _ in range(epochs):
                off, mse = 0, 0.0
                while True:
                    n = min(mb, N - off)
                    unif = t32(rng.uniform(size=(n, col)))
                    err = C.c_double()
                    _check(lib, lib.rbm_dense_cd1(h, Xin[off:off + n].data_ptr(), n, unif.data_ptr(), wc, rate, rate,
                                                  rate, mom, C.byref(err)))
                    mse += err.value / ncases
                    off += n
                    if n < mb or off >= N:
                        break
                print("Done epoch: MSE=%f" % mse)
            _check(lib, lib.rbm_dense_get(h, W32.ctypes.data, vb32.ctypes.data, hb32.ctypes.data))
            _check(lib, lib.rbm_dense_destroy(h))
            results.append(W32.astype(np.float64))
            results.append(hb32.astype(np.float64))
            # next layer's input: previous pre-sigmoid output . W + b (no nonlinearity between, :213-217)
            Wd, hbd = t32(W32), t32(hb32)
            Xn = torch.empty((N, col), dtype=torch.float32, device=dev)
            _check(lib, lib.rbm_affine(X.data_ptr(), Wd.data_ptr(), hbd.data_ptr(), N, row, col, Xn.data_ptr(), st))
            X = Xn
    torch.cuda.synchronize(dev)
    return results
