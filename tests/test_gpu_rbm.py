"""GPU parity of the SNN pre-training kernels (A7 sparse online CD-1, A7' dense CD-1) against
oracle/rbm_oracle.py, through the C ABI of include/rbm_hip.h.  f32 arithmetic vs the float64
oracle, sequential updates: tolerances are relative to the size of the parameter CHANGE."""
import numpy as np
import pytest

from oracle import rbm_oracle as ro

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import dl_utils, synth
from deep_ctr_amd import sampling_based_gaussian_binary_rbm_sparse as gbrbm

pytestmark = pytest.mark.gpu


def make_lines(tmp_path, n=240, n_rows=300, seed=3):
    """16 features per line with ids 2*row+1, so id-1 never collides with a feature (32 visibles)."""
    sizes = synth.field_sizes_tiny(n_rows)
    ids = synth.zipf_ids(n, sizes, 1.1, seed)
    feats = 2 * ids + 1
    path = tmp_path / 'train.fm.txt'
    with open(path, 'w') as f:
        for t in range(n):
            f.write('0 ' + ' '.join('%d:1' % v for v in feats[t]) + '\n')
    return str(path), [list(map(int, feats[t])) for t in range(n)], 2 * sum(sizes) + 2


def rel_change_err(got, ref, init):
    return np.abs(got - ref).max() / (np.abs(ref - init).max() + 1e-30)


def test_get_rbm_weights_matches_oracle(built, tmp_path):
    path, lines_feats, x_dim = make_lines(tmp_path)
    arr = [x_dim, 40, 24, 12]
    dl_utils.seed_global(1234)
    res = gbrbm.get_rbm_weights(path, arr, ncases=len(lines_feats), batch_size=100000)
    rng = np.random.RandomState(1234)
    ref = ro.get_rbm_weights(lines_feats, arr, rng, batch_size=100000)
    # initial values (for the size of the change): same stream
    rng0 = np.random.RandomState(1234)
    p0 = rng0.uniform(-.1, .1, x_dim * 40 + x_dim + 40)
    W0_init = p0[:x_dim * 40].reshape(x_dim, 40)
    assert res[0].shape == (x_dim, 40) and res[2].shape == (40, 24) and res[4].shape == (24, 12)
    assert rel_change_err(res[0], ref[0], W0_init) < 2e-3          # sparse layer: 3 epochs online, f32
    np.testing.assert_allclose(res[1], ref[1], rtol=0, atol=2e-3 * np.abs(ref[1] - p0[-40:]).max() + 1e-7)
    for k in (2, 3, 4, 5):                                         # dense layers
        np.testing.assert_allclose(res[k], ref[k], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("tag", ['a', 'b'])
def test_get_rbm_weights_against_the_reference_run(built, tmp_path, golden_dir, tag):
    """The HIP pre-training (sparse online CD-1 + dense CD-1 layers, three epochs each) against what the REFERENCE's own
    NumPy module returned for the same lines and seed (tests/golden/ref_run.npz, written by make_golden_ref.py in the build
    container): f32 kernels vs the reference's float64, tolerances relative to the size of the parameter change."""
    import os
    ref = np.load(os.path.join(golden_dir, 'ref_run.npz'))
    feats = ref['rbm_%s_feats' % tag]
    arr = [int(v) for v in ref['rbm_%s_arr' % tag]]
    path = tmp_path / 'train.fm.txt'
    with open(path, 'w') as f:
        for row in feats:
            f.write('0 ' + ' '.join('%d:1' % v for v in row) + '\n')
    dl_utils.seed_global(1234)
    res = gbrbm.get_rbm_weights(str(path), arr, ncases=len(feats), batch_size=int(ref['rbm_%s_batch' % tag]))
    x_dim, H0 = arr[0], arr[1]
    p0 = np.random.RandomState(1234).uniform(-.1, .1, x_dim * H0 + x_dim + H0)
    W0_init = p0[:x_dim * H0].reshape(x_dim, H0)
    assert rel_change_err(res[0], ref['rbm_%s_res0' % tag], W0_init) < 2e-3
    np.testing.assert_allclose(res[1], ref['rbm_%s_res1' % tag], rtol=0, atol=2e-3 * np.abs(ref['rbm_%s_res1' % tag] - p0[-H0:]).max() + 1e-7)
    for k in range(2, len(res)):
        np.testing.assert_allclose(res[k], ref['rbm_%s_res%d' % (tag, k)], rtol=3e-4, atol=3e-6)


def test_dense_cd1_minibatches_and_bf16(built, tmp_path):
    """Two mini-batches per epoch (the short last batch ends the epoch, :286-287) in f32; bf16 runs."""
    path, lines_feats, x_dim = make_lines(tmp_path, n=250)
    arr = [x_dim, 32, 20]
    dl_utils.seed_global(7)
    res = gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=200)
    ref = ro.get_rbm_weights(lines_feats, arr, np.random.RandomState(7), batch_size=200)
    np.testing.assert_allclose(res[2], ref[2], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(res[3], ref[3], rtol=2e-4, atol=2e-6)
    dl_utils.seed_global(7)
    resb = gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=200, precision='bf16')
    assert np.abs(resb[2] - ref[2]).max() < 5e-4
    with pytest.raises(ValueError):
        gbrbm.get_rbm_weights(path, arr, ncases=250, batch_size=125)       # empty last batch in the reference


def test_sparse_minibatch_mode(built, tmp_path):
    """rbm_sparse_batch (the throughput mode; not the reference's schedule): with mini-batches of 1 it is the online
    trainer; with mini-batches of 64 / 200 it follows oracle.sparse_cd1_minibatch.  Round 3: the row update groups the
    mini-batch's (row, entry) pairs with the library's own stable radix sort and sums every row's run in example order -- no
    float atomics, so two runs agree bit for bit (asserted; 200 examples in ONE mini-batch: rows hit ~20 times, runs that
    cross the 32-entry chunks and go through the second level)."""
    import ctypes as C
    import torch
    from deep_ctr_amd import _capi
    path, lines_feats, x_dim = make_lines(tmp_path, n=200)
    lib, H, S = _capi.load(), 40, 32
    vid, vval = gbrbm.sparse_inputs(gbrbm.parse_lines(path))
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    again = {}
    for M in (1, 64, 200, 200):
        rng = np.random.RandomState(5)
        ost = ro.SparseRBMState(x_dim, H, S, rng)
        W0, vb0 = ost.W.copy(), ost.visbias.copy()
        Wd = torch.as_tensor(ost.W.astype(np.float32)).to(dev); vb = torch.as_tensor(ost.visbias.astype(np.float32)).to(dev)
        hb = torch.as_tensor(ost.hidbias.astype(np.float32)).to(dev)
        ws = torch.zeros((S, H), dtype=torch.float32, device=dev)
        dW, dvis = torch.zeros_like(Wd), torch.zeros_like(vb)
        unif = rng.uniform(size=(len(lines_feats), H))
        ud = torch.as_tensor(unif.astype(np.float32)).to(dev)
        err = C.c_double()
        vid_d, vval_d = torch.as_tensor(vid).to(dev).contiguous(), torch.as_tensor(vval).to(dev).contiguous()   # held until the call returns
        assert vid_d.shape == (len(lines_feats), S) and vid_d.dtype == torch.int32 and vval_d.dtype == torch.uint8
        rc = lib.rbm_sparse_batch(Wd.data_ptr(), dW.data_ptr(), vb.data_ptr(), dvis.data_ptr(), hb.data_ptr(), ws.data_ptr(),
                                  vid_d.data_ptr(), vval_d.data_ptr(), ud.data_ptr(),
                                  len(lines_feats), M, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st)
        assert rc == 0, lib.rbm_last_error()

        class Replay(object):                       # the oracle draws uniform(size=(1, H)) per example: replay the same numbers
            def __init__(self):
                self.i = 0

            def uniform(self, size=None):
                self.i += 1
                return unif[self.i - 1].reshape(size)
        rp, e_ref = Replay(), 0.0
        dicts = [ro.sparse_line_dict(f) for f in lines_feats]
        if M == 1:
            for keys, v in dicts:
                e_ref += ro.sparse_cd1_example(ost, keys, v, rp)
        else:
            for n0 in range(0, len(dicts), M):
                e_ref += ro.sparse_cd1_minibatch(ost, dicts[n0:n0 + M], rp)
        assert rel_change_err(Wd.cpu().numpy().astype(np.float64), ost.W, W0) < 2e-3, M
        np.testing.assert_allclose(ws.cpu().numpy(), ost.weightstep, rtol=2e-3, atol=1e-9)
        assert abs(err.value - e_ref) <= 1e-4 * e_ref
        assert not dW.any().item() and not dvis.any().item()          # the scratch accumulators come back zero
        state = (Wd.cpu().numpy().copy(), vb.cpu().numpy().copy(), hb.cpu().numpy().copy(), ws.cpu().numpy().copy())
        np.testing.assert_allclose(state[1], ost.visbias, rtol=0, atol=2e-3 * np.abs(ost.visbias - vb0).max() + 1e-9)
        if M in again:
            for x, z in zip(again[M], state):
                assert np.array_equal(x, z)                         # bit-reproducible
        again[M] = state


def test_sparse_needs_32_visibles(built, tmp_path):
    p = tmp_path / 'bad.txt'
    p.write_text('0 5:1 6:1 9:1\n')
    with pytest.raises(ValueError):
        gbrbm.sparse_inputs(gbrbm.parse_lines(str(p)))


def test_snn_rbm_script_on_demo_tracks_oracle(built, golden_dir, tmp_path, monkeypatch):
    """`python SNN_RBM.py` end to end on the demo set (BASELINE configs[4] semantics): layer-wise
    CD-1 pre-training (A7, A7') then 2 fine-tune epochs (A8), against the same flow on the float64
    oracles with the reference's RNG consumption order."""
    import importlib.util
    import os
    from oracle import fnn_oracle as orc
    from sklearn.metrics import log_loss, roc_auc_score
    demo = os.path.join(golden_dir, 'demo')
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('DEEPCTR_DATA_DIR', demo)
    monkeypatch.setenv('DEEPCTR_EPOCHS', '2')
    monkeypatch.setenv('DEEPCTR_XDIM', 'auto')
    monkeypatch.setattr(dl_utils, 'log_path', str(tmp_path / 'log'))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('snn_script', os.path.join(root, 'deep-ctr_amd', 'SNN_RBM.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.run(['SNN_RBM.py'])
    assert (tmp_path / 'rbm_2997_.p').exists() and len(hist) == 2

    # the same flow on the oracles
    tr_ids, tr_y = mod.load_active_ids(os.path.join(demo, 'train.fm.txt'))
    te_ids, te_y = mod.load_active_ids(os.path.join(demo, 'test.fm.txt'))
    x_dim = int(max(tr_ids.max(), te_ids.max())) + 1
    H0, H1, H2 = 200, 300, 100
    rng = np.random.RandomState(1234)
    for (a, b) in ((x_dim, H0), (H0, H1), (H1, H2)):          # init_weight x3 consume the stream (SNN_RBM.py:78-80)
        rng.uniform(low=-1, high=1, size=(a, b))
    lines = [[int(v) for v in row if v >= 0] for row in tr_ids]
    ww0, bb0, ww1, bb1, ww2, bb2 = ro.get_rbm_weights(lines, [x_dim, H0, H1, H2], rng, batch_size=100000)
    p = {'w1': ww1.copy(), 'b1': bb1.copy(), 'w2': ww2.copy(), 'b2': bb2.copy(), 'w3': np.zeros(H2), 'b3': 0.0}
    ms = orc.TheanoMaskStream(H1, H2, 0.98, has_r0=False)
    ref = []
    for ep in range(2):
        r1, r2 = ms.next()
        orc.snn_train_step(p, ww0, bb0, tr_ids[:1000], tr_y[:1000].astype(np.float64), r1, r2, 0.001, 0.0)
        pte = orc.snn_predict(p, ww0, bb0, te_ids)
        ref.append((roc_auc_score(te_y, pte), log_loss(te_y, pte, labels=[0, 1])))
    for hrec, (auc, ll) in zip(hist, ref):
        print("SNN demo: auc %.6f vs %.6f, logloss %.6f vs %.6f" % (hrec['test_auc'], auc, hrec['test_logloss'], ll))
        assert abs(hrec['test_auc'] - auc) <= 2e-3
        assert abs(hrec['test_logloss'] - ll) <= 2e-4


def test_sparse_minibatch_data_parallel_two_virtual_ranks(built, tmp_path):
    """rbm_sparse_batch_dp / dp.DataParallelSparseRBM: every rank holds its shard of every global mini-batch; ONE all-reduce of
    S*H + H floats per mini-batch keeps wstep and hidbias identical on the ranks.  Two virtual ranks (two host threads, an
    all-reduce that meets at a barrier) against ONE process running the global mini-batches: after one global mini-batch the
    positional steps and the hidden bias agree with the single run and are bit-identical across the ranks; a row only one
    rank's examples touch equals the single run's row; after three mini-batches the ranks still hold identical wstep / hidbias."""
    import ctypes as C
    import threading
    import torch
    from deep_ctr_amd import _capi
    from deep_ctr_amd.dp import DataParallelSparseRBM
    path, lines_feats, x_dim = make_lines(tmp_path, n=200)
    lib, H, S, Mg = _capi.load(), 40, 32, 64
    vid, vval = gbrbm.sparse_inputs(gbrbm.parse_lines(path))
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(5)
    ost = ro.SparseRBMState(x_dim, H, S, rng)
    unif = rng.uniform(size=(len(lines_feats), H)).astype(np.float32)

    def fresh():
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)      # noqa: E731
        W, vb, hb = t(ost.W), t(ost.visbias), t(ost.hidbias)
        return W, torch.zeros_like(W), vb, torch.zeros_like(vb), hb, torch.zeros((S, H), dtype=torch.float32, device=dev)

    for nmb in (1, 3):
        N = nmb * Mg
        # one process, global mini-batches
        W, dW, vb, dvis, hb, ws = fresh()
        vd, vv, ud = (torch.as_tensor(a[:N]).to(dev).contiguous() for a in (vid, vval, unif))
        err = C.c_double()
        st = torch.cuda.current_stream(dev).cuda_stream
        assert lib.rbm_sparse_batch(W.data_ptr(), dW.data_ptr(), vb.data_ptr(), dvis.data_ptr(), hb.data_ptr(), ws.data_ptr(), vd.data_ptr(),
                                    vv.data_ptr(), ud.data_ptr(), N, Mg, H, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st) == 0
        torch.cuda.synchronize()
        # two ranks, 32 examples of every mini-batch each
        bar, views = threading.Barrier(2), [None, None]

        def allreduce_for(r):
            def fn(view):
                views[r] = view
                bar.wait()
                tot = views[0] + views[1]
                torch.cuda.synchronize()
                bar.wait()
                view.copy_(tot)
                torch.cuda.synchronize()
                bar.wait()
            return fn
        sel = [np.concatenate([np.arange(b * Mg + r * 32, b * Mg + r * 32 + 32) for b in range(nmb)]) for r in (0, 1)]
        states, errs, fails = [fresh(), fresh()], [0.0, 0.0], []

        def go(r):
            try:
                d = DataParallelSparseRBM(allreduce=allreduce_for(r))
                Wr, dWr, vbr, dvr, hbr, wsr = states[r]
                a = [torch.as_tensor(x[sel[r]]).to(dev).contiguous() for x in (vid, vval, unif)]
                errs[r] = d.epoch(Wr, dWr, vbr, dvr, hbr, wsr, a[0], a[1], a[2], 32, Mg)
            except BaseException as e:       # noqa: B902
                fails.append(e); bar.abort()
        ths = [threading.Thread(target=go, args=(r,)) for r in (0, 1)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not fails, fails
        torch.cuda.synchronize()
        (W0, _, vb0, _, hb0, ws0), (W1, _, vb1, _, hb1, ws1) = states
        assert torch.equal(ws0, ws1) and torch.equal(hb0, hb1)             # every rank applied the same sums
        if nmb == 1:
            np.testing.assert_allclose(ws0.cpu().numpy(), ws.cpu().numpy(), rtol=1e-4, atol=1e-10)
            np.testing.assert_allclose(hb0.cpu().numpy(), hb.cpu().numpy(), rtol=1e-6, atol=1e-9)
            assert abs(errs[0] + errs[1] - err.value) <= 1e-6 * err.value
            rows = [set(np.unique(vid[s])) for s in sel]
            only0 = np.array(sorted(rows[0] - rows[1]), np.int64)
            assert len(only0) > 0
            np.testing.assert_allclose(W0.cpu().numpy()[only0], W.cpu().numpy()[only0], rtol=1e-6, atol=1e-9)
            only1 = np.array(sorted(rows[1] - rows[0]), np.int64)
            assert torch.equal(W0[only1], torch.as_tensor(ost.W.astype(np.float32)).to(dev)[only1])    # rank 0 never moved them


@pytest.mark.parametrize("H,S,M,N", [(8, 16, 37, 100), (256, 32, 50, 120), (200, 32, 300, 300), (40, 5, 7, 30)])
def test_sparse_minibatch_edge_shapes(built, H, S, M, N):
    """The sorted row update at the edges: fewer than 32 visibles per example, the widest hidden layer, a short LAST mini-batch
    (N not a multiple of M: invalid entries in the sorted segment), all examples in one mini-batch, ids that repeat inside a
    mini-batch across different positions (the grouping is by row over the whole mini-batch, not per position) -- against the
    float64 mini-batch oracle; two runs bit-identical."""
    import ctypes as C
    import torch
    from deep_ctr_amd import _capi
    lib = _capi.load()
    rng = np.random.RandomState(H + S + M)
    n_rows = 400
    # S distinct sorted ids per example out of few rows: heavy repetition across examples, at any position
    vid = np.stack([np.sort(rng.choice(60 if S <= 16 else n_rows, size=S, replace=False)) for _ in range(N)]).astype(np.int32)
    vval = (rng.uniform(size=(N, S)) < 0.5).astype(np.uint8)
    W0 = rng.uniform(-0.1, 0.1, (n_rows, H)); vb0 = rng.uniform(-0.1, 0.1, n_rows); hb0 = rng.uniform(-0.1, 0.1, H)
    ws0 = rng.uniform(-1e-3, 1e-3, (S, H))
    unif = rng.uniform(size=(N, H))
    Wi = W0.astype(np.float32).astype(np.float64)

    class Replay(object):
        def __init__(self):
            self.i = 0

        def uniform(self, size=None):
            self.i += 1
            return unif.astype(np.float32).astype(np.float64)[self.i - 1].reshape(size)
    dev = torch.device('cuda', 0)
    stm = torch.cuda.current_stream(dev).cuda_stream
    runs = []
    for _ in range(2):
        t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dt).contiguous()      # noqa: E731
        Wd, vb, hb, ws = t(W0), t(vb0), t(hb0), t(ws0)
        dW, dvis = torch.zeros_like(Wd), torch.zeros_like(vb)
        vd, vv, ud = t(vid, torch.int32), t(vval, torch.uint8), t(unif)
        err = C.c_double()
        rc = lib.rbm_sparse_batch(Wd.data_ptr(), dW.data_ptr(), vb.data_ptr(), dvis.data_ptr(), hb.data_ptr(), ws.data_ptr(), vd.data_ptr(),
                                  vv.data_ptr(), ud.data_ptr(), N, M, H, S, 2e-4, 1e-2, 1e-2, 1e-2, 0.9, C.byref(err), stm)
        assert rc == 0, lib.rbm_last_error()
        runs.append((Wd.cpu().numpy(), vb.cpu().numpy(), hb.cpu().numpy(), ws.cpu().numpy(), err.value))
    for a, b in zip(runs[0][:4], runs[1][:4]):
        assert np.array_equal(a, b)
    W, vbg, hbg, wsg, e = runs[0]
    # rates 1e-2 (100x the reference's) so that the updates stand clear of f32 round-off
    ref_rates = dict(weightcost=0.0002, rates=(1e-2, 1e-2, 1e-2))
    st2 = ro.SparseRBMState(n_rows, H, S, np.random.RandomState(0))
    st2.W, st2.visbias, st2.hidbias, st2.weightstep = Wi.copy(), vb0.astype(np.float32).astype(np.float64), hb0.astype(np.float32).astype(np.float64), \
        ws0.astype(np.float32).astype(np.float64)
    rp2, e2 = Replay(), 0.0
    for n0 in range(0, N, M):
        e2 += ro.sparse_cd1_minibatch(st2, [(list(vid[n]), vval[n].astype(np.float64)) for n in range(n0, min(N, n0 + M))], rp2, **ref_rates)
    assert rel_change_err(W.astype(np.float64), st2.W, Wi) < 2e-3
    np.testing.assert_allclose(wsg, st2.weightstep, rtol=2e-3, atol=1e-8)
    assert np.abs(hbg - st2.hidbias).max() <= 2e-3 * np.abs(st2.hidbias - hb0).max() + 1e-7
    assert np.abs(vbg - st2.visbias).max() <= 2e-3 * np.abs(st2.visbias - vb0).max() + 1e-7
    assert abs(e - e2) <= 1e-4 * e2
