"""The HIP path against what the REFERENCE'S OWN STATEMENTS produced (tests/golden/ref_run.npz part 2, written by
tests/golden/make_golden_ref.py in the build container): the sparse-row update loop python/FNN_wnzh.py:299-306 and the SNN
update loop python/SNN_RBM.py:285-291, executed node by node on fixed lines and gradients.  The reference never travels; the
arrays do.  The dense MLP between gather and update has no reference run behind it (Theano): where a test needs a real gx the
fixture's gx is the float64 oracle's, and the device's f32 gx differs from it by f32 round-off only.

  upd2   fnn_step_scatter_global fed the FIXTURE'S gx: the HIP grouping + two-level closed-form update alone vs the loop
  upd1   a whole f32 fnn_train_step on 64 demo lines: table afterwards vs the loop's
  upd3   the same with two features of one field / repeated features (fnn_set_shadowed)
  snn1   a whole f32 bag-mode step (hidden0 = 200): ww0 / bb0 afterwards vs the loop's
"""
import os

import numpy as np
import pytest

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd.engine import FNNEngine

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
F, K = 16, 11


@pytest.fixture(scope='module')
def ref():
    return np.load(os.path.join(HERE, 'golden', 'ref_run.npz'))


def unpad(row):
    return [int(v) for v in row if v >= 0]


def f32r(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def problem(ref, tag):
    feats = [int(f) for f in ref['fnn_script_feats']]                 # table row i = i-th feature in id order
    row_of = {f: i for i, f in enumerate(feats)}
    field_of = {f: int(v) for f, v in zip(feats, ref['fnn_script_fields'])}
    lists = [unpad(r) for r in ref[tag + '_feats']]
    ids = np.full((len(lists), F), -1, np.int32)
    shadow = []
    for t, ft in enumerate(lists):
        seen = {}
        for f in ft:
            if field_of[f] in seen:
                shadow.append((t, field_of[f], row_of[seen[field_of[f]]]))
            seen[field_of[f]] = f
            ids[t, field_of[f]] = row_of[f]
    lr, lam = (float(v) for v in ref[tag + '_lr_lambda'])
    return ids, np.asarray(shadow, np.int32).reshape(-1, 3), lr, lam


def dense(ref, tag, xdim):
    seed = int(ref[tag + '_seed']) if tag.startswith('upd') else int(ref[tag + '_seeds_shape'][0]) + 5
    p = orc.init_fnn_weights(xdim, 300, 100)
    p['w3'] = np.random.RandomState(seed).uniform(-0.1, 0.1, 100)
    r1 = (np.random.RandomState(seed + 1).uniform(size=300) < 0.5).astype(np.uint8)
    r2 = (np.random.RandomState(seed + 2).uniform(size=100) < 0.5).astype(np.uint8)
    return p, r1, r2


def fm_engine(ref, lr, lam, p=None):
    eng = FNNEngine(F, K, 300, 100, max_batch=256, precision='f32', lr=lr, lambda1=0.0, lambda_fm=lam)
    eng.set_table(ref['fnn_script_weights'].astype(np.float32), ref['fnn_script_fields'].astype(np.int32), float(ref['fnn_script_w0_k_xdim'][0]))
    eng.set_dense(p if p is not None else orc.init_fnn_weights(1 + F * K, 300, 100))
    return eng


def check_table(got, before, after, rel):
    """got (f32, from the device) vs the reference loop's float64 table: the UPDATE of every row within `rel` of the largest
    update (+ one f32 rounding of the row itself), untouched rows untouched."""
    before32 = before.astype(np.float32)
    moved = np.abs(after - before).max(axis=1) > 0
    assert np.array_equal(got[~moved], before32[~moved])
    upd = np.abs(after - before).max()
    assert np.abs(got - after).max() <= rel * upd + 6e-8 * np.abs(after).max(), (np.abs(got - after).max(), upd)
    assert np.abs(got[moved] - before32[moved]).max() > 0.5 * upd


def test_scatter_kernels_vs_reference_loop_given_its_gx(ref, built):
    """upd2: 48 duplicate-heavy lines (three candidate features per field, hot rows hit ~16 times: decay powers c^16 with
    lr 0.05, lambda_fm 0.3 -> c = 0.999375), the reference loop's own random gx handed to the device in slot layout."""
    import torch
    ids, shadow, lr, lam = problem(ref, 'upd2')
    assert len(shadow) == 0
    gx = ref['upd2_gx']
    gxp = np.zeros((len(ids), 256), np.float32)
    for f in range(F):
        gxp[:, 16 * f:16 * f + K] = gx[:, 1 + f * K:1 + (f + 1) * K]
    eng = fm_engine(ref, lr, lam)
    B = len(ids)
    eng.step_begin(ids, np.zeros(B, np.float32), np.ones(300, np.uint8), np.ones(100, np.uint8), b_size=B)   # opens a step; w3 = 0: its own gx is 0
    eng.step_scatter_global(torch.as_tensor(ids).cuda().contiguous(), torch.as_tensor(gxp).cuda().contiguous())
    eng.step_end()
    eng.sync()
    # the device rounds gx and the rows to f32 on the way in: give the reference loop's restatement (bit-equal to the loop on the
    # CPU, tests/test_oracle_vs_reference.py) the same rounded inputs for a tight bound, and hold the result to the loop's own output too
    exact = orc.scatter_sgd(f32r(ref['fnn_script_weights']), ids, f32r(gx), lr, lam)
    got = eng.get_table()
    assert np.abs(got - exact).max() <= 3e-7 * np.abs(exact).max()
    check_table(got, ref['fnn_script_weights'], ref['upd2_after'], 2e-6)
    eng.close()


@pytest.mark.parametrize("tag", ['upd1', 'upd3'])
def test_train_step_table_vs_reference_loop(ref, tag, built):
    """A whole device step (gather -> MLP -> dense SGD -> sparse-row SGD) on the fixture's lines; its gx against the fixture's
    (oracle f64), its table against the table the reference's loop left."""
    ids, shadow, lr, lam = problem(ref, tag)
    assert (len(shadow) > 5) == (tag == 'upd3')
    p, r1, r2 = dense(ref, tag, 1 + F * K)
    eng = fm_engine(ref, lr, lam, p)
    x = eng.gather(ids).cpu().numpy()
    assert np.array_equal(x, ref[tag + '_x'].astype(np.float32))               # A3 against the script's own get_fxy
    if len(shadow):
        eng.set_shadowed(shadow)
    out = eng.train_step(ids, ref[tag + '_y'].astype(np.float32), r1, r2, want_gx=True)
    gs = np.abs(ref[tag + '_gx']).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), ref[tag + '_gx'], rtol=2e-3, atol=2e-5 * gs)
    check_table(eng.get_table(), ref['fnn_script_weights'], ref[tag + '_after'], 3e-4)
    eng.close()


def test_snn_step_vs_reference_loops(ref, built):
    """snn1: x of the bag kernel against the reference's get_fi_h1_y loop; ww0 / bb0 after one f32 step against the
    reference's update loop."""
    s0, s1, x_dim, h0 = (int(v) for v in ref['snn1_seeds_shape'])
    ww0 = np.random.RandomState(s0).uniform(-0.1, 0.1, (x_dim, h0))
    bb0 = np.random.RandomState(s1).uniform(-0.1, 0.1, h0)
    active = [unpad(r) for r in ref['snn1_active']]
    ids = np.array([a + [-1] * (F - len(a)) for a in active], np.int32)
    lr = float(ref['snn1_lr'])
    p, r1, r2 = dense(ref, 'snn1', h0)
    eng = FNNEngine(F, 0, 300, 100, max_batch=256, precision='f32', lr=lr, lambda1=0.0, lambda_fm=0.0, reg_all=True, mode='bag', hidden0=h0)
    eng.set_table(ww0.astype(np.float32), np.zeros(x_dim, np.int32), 0.0)
    eng.set_bag_bias(bb0.astype(np.float32))
    eng.set_dense(p)
    np.testing.assert_allclose(eng.gather(ids).cpu().numpy(), ref['snn1_x'], rtol=2e-6, atol=1e-7)
    out = eng.train_step(ids, ref['snn1_y'].astype(np.float32), r1, r2, want_gx=True)
    gs = np.abs(ref['snn1_gx']).max()
    np.testing.assert_allclose(out['gx'].cpu().numpy(), ref['snn1_gx'], rtol=2e-3, atol=2e-5 * gs)
    touched = ref['snn1_touched']
    got = eng.get_table()
    upd = np.abs(ref['snn1_ww0_after_touched'] - ww0[touched]).max()
    assert np.abs(got[touched] - ref['snn1_ww0_after_touched']).max() <= 1e-3 * upd + 6e-8 * 0.1
    rest = np.setdiff1d(np.arange(x_dim), touched)
    assert np.array_equal(got[rest], ww0.astype(np.float32)[rest])
    bupd = np.abs(ref['snn1_bb0_after'] - bb0).max()
    assert np.abs(eng.get_bag_bias() - ref['snn1_bb0_after']).max() <= 1e-3 * bupd + 6e-8 * 0.1
    eng.close()


def test_dae_lower_layer_propagation_vs_reference_loop(ref, built):
    """Row N2: dae_bag_cumsum_sigmoid_f64 (layer 0: the reference's running sum over hidden units, then the sigmoid) and
    dae_affine_sigmoid_f64 (every further layer) against the output of the reference's own propagation loop
    (sampling_based_denosing_autoencoder.py:164-188, tests/golden/ref_run.npz dae_prop_*)."""
    import ctypes as C
    import torch
    from deep_ctr_amd import _capi
    lib = _capi.load()
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    W0, b0, W1, b1 = (torch.as_tensor(ref['dae_prop_res%d' % i]).to(dev).contiguous() for i in range(4))
    ids = torch.as_tensor(ref['dae_prop_ids'].astype(np.int32)).to(dev).contiguous()
    n, F = ids.shape
    H0, H1 = W0.shape[1], W1.shape[1]
    X = torch.empty((n, H0), dtype=torch.float64, device=dev)
    assert lib.dae_bag_cumsum_sigmoid_f64(W0.data_ptr(), b0.data_ptr(), H0, W0.shape[0], ids.data_ptr(), n, F, X.data_ptr(), st) == 0, lib.dae_last_error()
    Y = torch.empty((n, H1), dtype=torch.float64, device=dev)
    assert lib.dae_affine_sigmoid_f64(X.data_ptr(), W1.data_ptr(), b1.data_ptr(), n, H0, H1, Y.data_ptr(), st) == 0, lib.dae_last_error()
    torch.cuda.synchronize()
    np.testing.assert_allclose(Y.cpu().numpy(), ref['dae_prop_out'], rtol=0, atol=1e-13)
