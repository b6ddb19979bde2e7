"""Generates the committed fixtures under tests/golden/ from the float64 oracle.

PARITY UNPINNED by the reference (it holds no golden vectors and cannot run here);
these vectors pin the build's own restatement so that the oracle, the C port and the HIP path are
all checked against the same numbers.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deep_ctr_amd  # noqa: E402,F401
from deep_ctr_amd import synth  # noqa: E402
from oracle import fnn_oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # demo tiny set in the three text formats (smaller counts keep the repo light)
    demo = synth.make_demo(os.path.join(OUT, 'demo'), n_train=1200, n_test=400, n_feat=1000, rank=10,
                           seed=20260410, w0=-3.0)
    F, K, H1, H2 = 16, 11, 300, 100
    xdim = 1 + F * K
    rows = demo['rows'].astype(np.float64)
    w0 = demo['w0']

    # init.npz: w1/w2 of python/FNN_wnzh.py:106-130 with seed 1234 (corner values only + checksums)
    p = orc.init_fnn_weights(xdim, H1, H2, 'tanh', seed=1234)
    np.savez(os.path.join(OUT, 'init.npz'), w1_corner=p['w1'][:2, :4], w2_corner=p['w2'][:2, :4],
             w1_sum=p['w1'].sum(), w2_sum=p['w2'].sum(), w1_abs=np.abs(p['w1']).sum())

    # masks.npz: first 3 rows of r1/r2 of RandomStreams(234), p = 0.5
    ms = orc.TheanoMaskStream(H1, H2, 0.5)
    m = [ms.next() for _ in range(3)]
    np.savez(os.path.join(OUT, 'masks.npz'), seeds=np.array(ms.seeds),
             r1=np.array([a for a, _ in m]).astype(np.uint8), r2=np.array([b for _, b in m]).astype(np.uint8))

    # step.npz: one train step on a duplicate-heavy batch with an empty field
    B = 64
    ids = demo['ids'][:B].copy()
    ids[5, 3] = -1                      # an absent field
    ids[:, 6] = ids[0, 6]               # every example hits the same row
    y = demo['y'][:B].astype(np.float64)
    p['w3'] = np.random.RandomState(77).uniform(-0.1, 0.1, H2)
    p['b3'] = 0.05
    # f32-representable inputs so the HIP path starts from identical values
    p = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else float(np.float32(v)))
         for k, v in p.items()}
    r1, r2 = m[0]
    lr, lam1, lamfm = 0.001, 0.0, 0.1
    rows_s = rows.copy()
    p_s = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    res = orc.train_step(p_s, rows_s, w0, ids, y, r1, r2, lr, lam1, lamfm)
    touched = np.unique(ids[ids >= 0])
    np.savez(os.path.join(OUT, 'step.npz'), ids=ids, y=y, r1=r1.astype(np.uint8), r2=r2.astype(np.uint8),
             w3=p['w3'], b3=p['b3'], lr=lr, lambda1=lam1, lambda_fm=lamfm,
             x=res['x'], gx=res['gx'], p_drop=res['p_drop'], loss=res['loss'],
             gw3=res['grads']['w3'], gb3=res['grads']['b3'], gb2=res['grads']['b2'], gb1=res['grads']['b1'],
             gw2_corner=res['grads']['w2'][:4, :4], gw1_corner=res['grads']['w1'][:4, :4],
             w1_after_sum=p_s['w1'].sum(), w2_after_sum=p_s['w2'].sum(), w3_after=p_s['w3'], b3_after=p_s['b3'],
             touched=touched, rows_after=rows_s[touched])

    # epoch.npz: reference defaults (batch 100, lr .001, dropout .5, lambda_fm .1), 3 epochs
    p_e = orc.init_fnn_weights(xdim, H1, H2, 'tanh', seed=1234)
    p_e = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in p_e.items()}
    rows_e = rows.copy()
    n_train = 1200
    hist = orc.run_epochs(p_e, rows_e, w0, demo['ids'][:n_train], demo['y'][:n_train],
                          demo['ids'][n_train:], demo['y'][n_train:], 100, 0.001, 0.0, 0.1, 0.5, 3, H1, H2)
    keys = ['train_auc', 'train_rmse', 'train_logloss', 'test_auc', 'test_rmse', 'test_logloss']
    np.savez(os.path.join(OUT, 'epoch.npz'), **{k: np.array([h[k] for h in hist]) for k in keys})
    for h in hist:
        print(h)


if __name__ == '__main__':
    main()
