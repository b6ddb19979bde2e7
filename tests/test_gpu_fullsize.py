"""Full-size parity: BASELINE's table shapes (937,670 rows) for the configs whose round-1 tests used small tables --
FNN_IP_L7 (configs[2]), the SNN fine-tune step and the online sparse CD-1 pass (configs[4]).  The float64 oracles
run on the TOUCHED rows only (ids remapped through a sorted list of the rows the batch names: a monotone map, so
every "sorted ids" / "example order" rule of the reference is preserved); rows nobody touched must come back bit
for bit.  (FNN L3 at full size: tests/test_gpu_parity.py::test_full_shape_step_f32_vs_oracle.)
"""
import ctypes as C

import numpy as np
import pytest

from oracle import fnn_oracle as orc
from oracle import ipnn_oracle as io
from oracle import rbm_oracle as ro

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import _capi, synth
from deep_ctr_amd.engine import FNNEngine
from deep_ctr_amd.ipnn import IPNNEngine

pytestmark = pytest.mark.gpu
F, K, H1, H2 = 16, 11, 300, 100
SIZES = synth.field_sizes_ipinyou()
D = sum(SIZES)


def f32r(a):
    return np.asarray(a, np.float32).astype(np.float64)


def compact(ids):
    """(touched rows sorted, ids remapped into them; -1 stays -1)."""
    touched = np.unique(ids[ids >= 0])
    idc = np.where(ids >= 0, np.searchsorted(touched, np.maximum(ids, 0)), -1)
    return touched, idc


def untouched_sample(touched, n=20000, seed=0):
    cand = np.random.RandomState(seed).randint(0, D, size=n)
    return np.setdiff1d(cand, touched)


def test_tables_are_the_benchmark_shape():
    assert D == 937670 and len(SIZES) == F


def test_ipnn_l7_step_bf16_on_the_full_table(built):
    """BASELINE configs[2] as bench.py runs it: 937,670 x 11 table, hidden 1000/800/600/400/200/100/50 relu, keep_prob 0.5,
    batch 4096 Zipf ids, bf16.  One SGD step against the float64 oracle on the touched rows: logits within 1e-3, loss
    within 2e-4, every dense update and the touched rows' update pointing the oracle's way (cosine > 0.99), untouched rows
    unchanged (round 3: tightened from 5e-2 / 2 % / 0.98 to about four times what was observed)."""
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    B = 4096
    rng = np.random.RandomState(11)
    table = synth.fm_table(D, K, 0.2, 1234)
    ids = synth.zipf_ids(B, SIZES, 1.1, 77)
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    d = [F * K + F * (F - 1) // 2 + 1] + hidden + [1]
    params = {'b': float(np.float32(0.1)), 'W': [f32r(rng.uniform(-0.06, 0.06, (d[i], d[i + 1]))) for i in range(len(d) - 1)],
              'bias': [f32r(rng.uniform(-0.1, 0.1, d[i + 1])) for i in range(len(d) - 1)]}
    masks = [(np.random.RandomState(40 + t).uniform(size=(B, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='bf16', lr=1e-3, keep_prob=0.5)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks, want_logits=True)
    touched, idc = compact(ids)
    tc = table[touched].astype(np.float64)
    t0 = tc.copy()
    p0 = [w.copy() for w in params['W']]
    loss, logits, g = io.sgd_step(params, tc, idc, y, 'relu', 1e-3, [m.astype(np.float64) for m in masks], 0.5)
    obs = {'logits_max_abs_err': float(np.abs(out['logits'].cpu().numpy() - logits).max()), 'loss_rel_err': float(abs(out['loss'] - loss) / abs(loss))}
    b, Ws, bs = eng.get_params()
    cosines = []
    for t in range(len(Ws)):
        du, dv = (Ws[t] - p0[t]).ravel(), (params['W'][t] - p0[t]).ravel()
        cosines.append(float(du @ dv / (np.linalg.norm(du) * np.linalg.norm(dv) + 1e-30)))
    got = eng.get_rows(touched).astype(np.float64)
    du, dv = (got - t0).ravel(), (tc - t0).ravel()
    obs['dense_update_cosines'] = cosines
    obs['row_update_cosine'] = float(du @ dv / (np.linalg.norm(du) * np.linalg.norm(dv) + 1e-30))
    obs['row_err_over_largest_update'] = float(np.abs(got - tc).max() / np.abs(tc - t0).max())
    try:                                                    # kept beside the profiles (DESIGN.md quotes them)
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        json.dump(obs, open(os.path.join(root, 'gpurun_out', 'ipnn_fullsize_bf16_observed.json'), 'w'))
    except OSError:
        pass
    # observed on MI355X (profiles/r03_ipnn_fullsize_bf16_observed.json); asserted at about twice that.  The tight anchor of this
    # shape is the f32 test below.
    # logits 2.4e-4, loss 2.1e-5, cosines 0.9942 .. 0.999998 (lowest at the widest layer), rows 0.9938 / 0.079
    assert obs['logits_max_abs_err'] < 1e-3 and obs['loss_rel_err'] <= 2e-4
    assert min(cosines) > 0.99, cosines
    assert obs['row_update_cosine'] > 0.99 and obs['row_err_over_largest_update'] <= 0.12
    un = untouched_sample(touched)
    assert np.array_equal(eng.get_rows(un), table[un])
    eng.close()


def test_ipnn_l7_step_f32_on_the_full_table(built):
    """The tight anchor of BASELINE configs[2] at its full shape: the same problem in the f32 mode (exact-f32 MFMA), held to the
    tolerances of tests/test_gpu_ipnn.py::test_ipnn_step_f32_vs_oracle -- logits rtol 2e-4, the loss to 5e-5, every dense
    tensor's and every touched row's UPDATE within 2e-3 of its size, untouched rows bit for bit.  (Round-2 review: the bf16 test
    above is loose by necessity -- seven bf16 layers -- and nothing tight stood behind this shape.)"""
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    B = 4096
    rng = np.random.RandomState(11)
    table = synth.fm_table(D, K, 0.2, 1234)
    ids = synth.zipf_ids(B, SIZES, 1.1, 77)
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    d = [F * K + F * (F - 1) // 2 + 1] + hidden + [1]
    params = {'b': float(np.float32(0.1)), 'W': [f32r(rng.uniform(-0.06, 0.06, (d[i], d[i + 1]))) for i in range(len(d) - 1)],
              'bias': [f32r(rng.uniform(-0.1, 0.1, d[i + 1])) for i in range(len(d) - 1)]}
    masks = [(np.random.RandomState(40 + t).uniform(size=(B, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    eng = IPNNEngine(F, K, hidden, 'relu', max_batch=B, precision='f32', lr=1e-3, keep_prob=0.5)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    out = eng.train_step(ids, y, masks, want_logits=True)
    touched, idc = compact(ids)
    tc = table[touched].astype(np.float64)
    t0 = tc.copy()
    p0 = {'b': params['b'], 'W': [w.copy() for w in params['W']], 'bias': [b.copy() for b in params['bias']]}
    loss, logits, g = io.sgd_step(params, tc, idc, y, 'relu', 1e-3, [m.astype(np.float64) for m in masks], 0.5)
    np.testing.assert_allclose(out['logits'].cpu().numpy(), logits, rtol=2e-4, atol=2e-5)
    assert abs(out['loss'] - loss) <= 5e-5 * max(1.0, abs(loss))
    b, Ws, bs = eng.get_params()
    for t in range(len(Ws)):
        cw = np.abs(params['W'][t] - p0['W'][t]).max() + 1e-12
        assert np.abs(Ws[t] - params['W'][t]).max() <= 2e-3 * cw + 2e-7, ('W', t)
        cb = np.abs(params['bias'][t] - p0['bias'][t]).max() + 1e-12
        assert np.abs(bs[t] - params['bias'][t]).max() <= 2e-3 * cb + 2e-7, ('b', t)
    assert abs(b - params['b']) <= 2e-3 * abs(params['b'] - p0['b']) + 2e-7
    ct = np.abs(tc - t0).max() + 1e-12
    assert np.abs(eng.get_rows(touched) - tc).max() <= 2e-3 * ct + 2e-7
    un = untouched_sample(touched)
    assert np.array_equal(eng.get_rows(un), table[un])
    eng.close()


def test_snn_finetune_step_f32_on_the_full_table(built):
    """BASELINE configs[4], fine-tune half: 937,670 x 200 bag table (750 MB), hidden 300/100, batch 4096 Zipf ids, f32, one
    step against oracle.snn_train_step (python/SNN_RBM.py:238-291) on the touched rows."""
    h0, B = 200, 4096
    rng = np.random.RandomState(5)
    ww0 = np.random.default_rng(0).standard_normal((D, h0), dtype=np.float32) * np.float32(0.05)
    bb0 = (rng.standard_normal(h0) * 0.1).astype(np.float32)
    ids = synth.zipf_ids(B, SIZES, 1.1, 3)
    ids[7, 2] = -1
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    p = {'w1': f32r(rng.uniform(-0.3, 0.3, (h0, H1))), 'b1': f32r(rng.uniform(-0.1, 0.1, H1)),
         'w2': f32r(rng.uniform(-0.3, 0.3, (H1, H2))), 'b2': f32r(rng.uniform(-0.1, 0.1, H2)),
         'w3': f32r(rng.uniform(-0.2, 0.2, H2)), 'b3': 0.05}
    r1 = (rng.uniform(size=H1) < 0.9).astype(np.uint8)
    r2 = (rng.uniform(size=H2) < 0.9).astype(np.uint8)
    eng = FNNEngine(F, 0, H1, H2, max_batch=B, precision='f32', lr=0.01, lambda1=0.001, lambda_fm=0.0, reg_all=True, mode='bag',
                    hidden0=h0)
    eng.set_table(ww0, np.zeros(D, np.int32), 0.0)
    eng.set_bag_bias(bb0)
    eng.set_dense(p)
    out = eng.train_step(ids, y, r1, r2, want_p=True)
    touched, idc = compact(ids)
    wc = ww0[touched].astype(np.float64)
    w_init = wc.copy()
    bb64 = bb0.astype(np.float64)
    p64 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = orc.snn_train_step(p64, wc, bb64, idc, y.astype(np.float64), r1.astype(float), r2.astype(float), 0.01, 0.001)
    np.testing.assert_allclose(out['p'].cpu().numpy(), ref['p_drop'], rtol=3e-4, atol=1e-6)
    assert abs(out['loss'] - ref['loss']) <= 3e-5 * max(1.0, abs(ref['loss']))
    upd = np.abs(wc - w_init).max()
    assert np.abs(eng.get_rows(touched) - wc).max() <= 1e-3 * upd + 3e-7
    assert np.abs(eng.get_bag_bias() - bb64).max() <= 1e-3 * np.abs(bb64 - bb0).max() + 3e-7
    dn = eng.get_dense()
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        scale = np.abs(p64[k] - p[k]).max() + 1e-12
        assert np.abs(dn[k] - p64[k]).max() <= 1e-3 * scale + 1e-7, k
    un = untouched_sample(touched)
    assert np.array_equal(eng.get_rows(un), ww0[un])
    eng.close()


def test_sparse_rbm_online_pass_on_the_full_table(built):
    """BASELINE configs[4], pre-training half: the reference's exact online CD-1 pass (python/sampling_based_gaussian_
    binary_rbm_sparse.py:413-508) over 320 examples on the 937,670 x 200 table, against oracle.sparse_cd1_example on the
    touched rows (the same uniform draws replayed)."""
    import torch
    lib = _capi.load()
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    N, H, S = 320, 200, 32
    rng = np.random.RandomState(9)
    feats = np.sort(2 * (synth.zipf_ids(N, SIZES, 1.1, 5).astype(np.int64) // 2) + 1, axis=1)       # odd ids: id - 1 is never a feature
    vid = np.empty((N, S), np.int32)
    vval = np.empty((N, S), np.uint8)
    lines = []
    for n in range(N):
        uniq = np.unique(feats[n])
        while len(uniq) < 16:                                      # the reference needs exactly 32 visibles (:388)
            extra = 2 * rng.randint(0, D // 2 - 1) + 1
            uniq = np.unique(np.append(uniq, extra))
        keys, v = ro.sparse_line_dict([int(f) for f in uniq])
        assert len(keys) == S
        vid[n], vval[n] = keys, v
        lines.append((keys, v))
    W0 = np.random.default_rng(1).uniform(-0.1, 0.1, (D, H)).astype(np.float32)
    vb0 = np.random.default_rng(2).uniform(-0.1, 0.1, D).astype(np.float32)
    hb0 = np.random.default_rng(3).uniform(-0.1, 0.1, H).astype(np.float32)
    unif = rng.uniform(size=(N, H))
    Wd, vbd, hbd = torch.as_tensor(W0).to(dev), torch.as_tensor(vb0).to(dev), torch.as_tensor(hb0).to(dev)
    ws = torch.zeros((S, H), dtype=torch.float32, device=dev)
    vid_d, vval_d = torch.as_tensor(vid).to(dev).contiguous(), torch.as_tensor(vval).to(dev).contiguous()
    ud = torch.as_tensor(unif.astype(np.float32)).to(dev)
    err = C.c_double()
    rc = lib.rbm_sparse_epoch(Wd.data_ptr(), vbd.data_ptr(), hbd.data_ptr(), ws.data_ptr(), vid_d.data_ptr(), vval_d.data_ptr(),
                              ud.data_ptr(), N, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st)
    assert rc == 0, lib.rbm_last_error()
    torch.cuda.synchronize()
    touched = np.unique(vid)
    ost = object.__new__(ro.SparseRBMState)                        # the oracle's state on the touched rows only
    ost.W, ost.visbias, ost.hidbias = W0[touched].astype(np.float64), vb0[touched].astype(np.float64), hb0.astype(np.float64)
    ost.weightstep, ost.nsparsevis = np.zeros((S, H)), S
    Wi = ost.W.copy()

    class Replay(object):
        def __init__(self):
            self.i = 0

        def uniform(self, size=None):
            self.i += 1
            return unif[self.i - 1].astype(np.float32).astype(np.float64).reshape(size)
    rp, e_ref = Replay(), 0.0
    for keys, v in lines:
        e_ref += ro.sparse_cd1_example(ost, list(np.searchsorted(touched, keys)), v, rp)
    Wg = Wd.cpu().numpy()
    assert np.abs(Wg[touched] - ost.W).max() / (np.abs(ost.W - Wi).max() + 1e-30) < 2e-3
    np.testing.assert_allclose(ws.cpu().numpy(), ost.weightstep, rtol=2e-3, atol=1e-9)
    vg = vbd.cpu().numpy()
    assert np.abs(vg[touched] - ost.visbias).max() <= 2e-3 * np.abs(ost.visbias - vb0[touched]).max() + 1e-7
    assert np.abs(hbd.cpu().numpy() - ost.hidbias).max() <= 2e-3 * np.abs(ost.hidbias - hb0).max() + 1e-7
    assert abs(err.value - e_ref) <= 1e-4 * e_ref
    mask = np.ones(D, bool); mask[touched] = False
    assert np.array_equal(Wg[mask], W0[mask]) and np.array_equal(vg[mask], vb0[mask])
