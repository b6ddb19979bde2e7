"""More committed fixtures (SURVEY.md 8c list): rbm_sparse.npz, rbm_dense.npz, snn_step.npz, ip_l7.npz, from the
float64 oracles.  PARITY UNPINNED by the reference (it holds no vectors and cannot run here); these pin the build's
own restatements, and the HIP path is checked against the same numbers.  Every random draw the kernels take as an
input (uniforms, masks) is stored, so a test replays it.  Run from the repo root:
    python tests/golden/make_golden_more.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deep_ctr_amd  # noqa: E402,F401
from deep_ctr_amd import synth  # noqa: E402
from oracle import fnn_oracle as orc  # noqa: E402
from oracle import ipnn_oracle as ipo  # noqa: E402
from oracle import rbm_oracle as ro  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class Replay(object):
    """rng stand-in that hands out pre-drawn uniforms row by row."""
    def __init__(self, u):
        self.u, self.i = u, 0

    def uniform(self, size=None):
        n = int(np.prod(size)) // self.u.shape[1]
        out = self.u[self.i:self.i + n].reshape(size)
        self.i += n
        return out


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def main():
    rng = np.random.RandomState(20260411)
    # ---- rbm_sparse.npz: 10 lines of 16 odd ids over 120 visibles, H = 24: online pass, then a mini-batch pass (M = 4)
    nvis, H, S, N = 120, 24, 32, 10
    feats = [sorted((2 * rng.choice(59, size=16, replace=False) + 1).tolist()) for _ in range(N)]
    dicts = [ro.sparse_line_dict(f) for f in feats]
    vid = np.array([k for k, _ in dicts], np.int32)
    vval = np.array([v for _, v in dicts], np.uint8)
    st = ro.SparseRBMState(nvis, H, S, rng)
    st.W, st.visbias, st.hidbias = f32(st.W), f32(st.visbias), f32(st.hidbias)
    W0, vb0, hb0 = st.W.copy(), st.visbias.copy(), st.hidbias.copy()
    unif = f32(rng.uniform(size=(N, H)))
    rp = Replay(unif)
    err = sum(ro.sparse_cd1_example(st, k, v, rp) for k, v in dicts)
    on = dict(W=st.W.copy(), vb=st.visbias.copy(), hb=st.hidbias.copy(), ws=st.weightstep.copy(), err=err)
    st.W, st.visbias, st.hidbias, st.weightstep = W0.copy(), vb0.copy(), hb0.copy(), np.zeros((S, H))
    rp = Replay(unif)
    errb = sum(ro.sparse_cd1_minibatch(st, dicts[n0:n0 + 4], rp) for n0 in range(0, N, 4))
    np.savez(os.path.join(OUT, 'rbm_sparse.npz'), vid=vid, vval=vval, unif=unif, W0=W0, vb0=vb0, hb0=hb0,
             W_online=on['W'], vb_online=on['vb'], hb_online=on['hb'], ws_online=on['ws'], err_online=on['err'],
             W_mb4=st.W, vb_mb4=st.visbias, hb_mb4=st.hidbias, ws_mb4=st.weightstep, err_mb4=errb)

    # ---- rbm_dense.npz: two CD-1 mini-batches, 20 x 12 -> 8
    ds = ro.DenseRBMState(12, 8, rng)
    ds.W, ds.visbias, ds.hidbias = f32(ds.W), f32(ds.visbias), f32(ds.hidbias)
    X = f32(rng.uniform(size=(20, 12)))
    U = f32(rng.uniform(size=(40, 8)))
    d0 = dict(W=ds.W.copy(), vb=ds.visbias.copy(), hb=ds.hidbias.copy())
    rp = Replay(U)
    e1 = ro.dense_cd1_batch(ds, X, rp)
    e2 = ro.dense_cd1_batch(ds, X, rp)
    np.savez(os.path.join(OUT, 'rbm_dense.npz'), X=X, unif=U, W0=d0['W'], vb0=d0['vb'], hb0=d0['hb'], W=ds.W, vb=ds.visbias,
             hb=ds.hidbias, ws=ds.weightstep, err=np.array([e1, e2]))

    # ---- snn_step.npz: one fine-tune step (bag of 16 rows of H0 = 200 -> 20 -> 12; the HIP strip kernel is built for bag
    # widths 193..255), duplicate-heavy ids, an empty field
    sizes = synth.field_sizes_tiny(120)
    B, H0, H1, H2 = 48, 200, 20, 12
    ids = synth.zipf_ids(B, sizes, 1.1, 3).astype(np.int32)
    ids[3, 5] = -1
    ids[:, 2] = ids[0, 2]
    y = (rng.uniform(size=B) < 0.3).astype(np.float64)
    ww0 = f32(rng.standard_normal((sum(sizes), H0)) * 0.1)
    bb0 = f32(rng.standard_normal(H0) * 0.05)
    p = {'w1': f32(rng.uniform(-.3, .3, (H0, H1))), 'b1': f32(rng.uniform(-.1, .1, H1)), 'w2': f32(rng.uniform(-.3, .3, (H1, H2))),
         'b2': f32(rng.uniform(-.1, .1, H2)), 'w3': f32(rng.uniform(-.3, .3, H2)), 'b3': float(np.float32(0.02))}
    r1 = (rng.uniform(size=H1) < 0.9).astype(np.float64)
    r2 = (rng.uniform(size=H2) < 0.9).astype(np.float64)
    p0 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ww, bb = ww0.copy(), bb0.copy()
    res = orc.snn_train_step(p, ww, bb, ids, y, r1, r2, 0.01, 0.001)          # tanh layers, as python/SNN_RBM.py:105-140
    touched = np.unique(ids[ids >= 0])
    np.savez(os.path.join(OUT, 'snn_step.npz'), ids=ids, y=y, r1=r1.astype(np.uint8), r2=r2.astype(np.uint8), ww0=ww0, bb0=bb0,
             lr=0.01, lambda1=0.001, x=res['x'], gx=res['gx'], p_drop=res['p_drop'], loss=res['loss'], touched=touched,
             rows_after=ww[touched], bb0_after=bb, **{'p0_' + k: np.asarray(v) for k, v in p0.items()},
             **{'p1_' + k: np.asarray(v) for k, v in p.items()})

    # ---- ip_l7.npz: z1 and logits of a 7-layer inner-product stack for fixed ids (python/FNN_IP_L7.py:102-133)
    F, K = 16, 11
    hidden = [40, 36, 32, 28, 24, 20, 16]
    sizes = synth.field_sizes_tiny(200)
    table = f32(rng.standard_normal((sum(sizes), K)) * 0.2)
    idsp = synth.zipf_ids(32, sizes, 1.1, 9)
    d = [F * K + F * (F - 1) // 2 + 1] + hidden + [1]
    params = {'b': float(np.float32(0.1)), 'W': [f32(rng.uniform(-.25, .25, (d[i], d[i + 1]))) for i in range(len(d) - 1)],
              'bias': [f32(rng.uniform(-.1, .1, d[i + 1])) for i in range(len(d) - 1)]}
    _, z1 = ipo.z1_of(table, params['b'], idsp)
    logits, _ = ipo.forward(params, table, idsp, 'relu')
    np.savez(os.path.join(OUT, 'ip_l7.npz'), ids=idsp, table=table, b=params['b'], z1=z1, logits=logits,
             **{'W%d' % i: w for i, w in enumerate(params['W'])}, **{'bias%d' % i: w for i, w in enumerate(params['bias'])})
    print("wrote rbm_sparse.npz rbm_dense.npz snn_step.npz ip_l7.npz")


if __name__ == '__main__':
    main()
