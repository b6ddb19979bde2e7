"""Bitwise repeatability of many back-to-back steps (no host sync between them): FNN (bf16, f32), SNN, FNN_IP_L7.
A store that is not visible to the next launch in time shows up as a run that differs from the others."""
import hashlib, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import deep_ctr_amd  # noqa
from deep_ctr_amd.ipnn import IPNNEngine
import test_gpu_parity as tp
import test_gpu_ipnn as ti


def digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def fnn_run(prec, steps, B):
    rows, fo, ids, y, p, r1, r2 = tp.make_problem(steps * B, seed=5, dup_col=6)
    eng = tp.make_engine(rows, fo, p, prec=prec, lr=0.01, lam1=0.001, lamfm=0.1)
    for s in range(steps):
        eng.train_step(ids[s * B:(s + 1) * B], y[s * B:(s + 1) * B], r1, r2, want_loss=False)
    d = eng.get_dense()
    out = digest([eng.get_table()] + [np.asarray(d[k]) for k in sorted(d)])
    eng.close()
    return out


def snn_run(steps, B):
    ww0, bb0, ids, y, p, r1, r2 = tp.make_snn_problem(steps * B, seed=6, dup_col=4)
    eng = tp.make_snn_engine(ww0, bb0, p, prec='bf16')
    for s in range(steps):
        eng.train_step(ids[s * B:(s + 1) * B], y[s * B:(s + 1) * B], r1, r2, want_loss=False)
    d = eng.get_dense()
    out = digest([eng.get_table(), eng.get_bag_bias()] + [np.asarray(d[k]) for k in sorted(d)])
    eng.close()
    return out


def ip_run(steps, B):
    hidden = [1000, 800, 600, 400, 200, 100, 50]
    table, ids, y, params, masks, d = ti.problem(B * 4, hidden, seed=7, n_rows=3000, scale=0.05)
    masks = [(np.random.RandomState(2 + t).uniform(size=(B * 4, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    eng = IPNNEngine(ti.F, ti.K, hidden, 'relu', max_batch=B, precision='bf16', lr=0.01, keep_prob=0.5)
    eng.set_params(table, params['b'], params['W'], params['bias'])
    for s in range(steps):
        sl = slice((s % 4) * B, (s % 4 + 1) * B)
        eng.train_step(ids[sl], y[sl], [m[sl] for m in masks])
    b, Ws, bs = eng.get_params()
    out = digest([eng.get_rows(np.unique(ids))] + Ws + bs)
    eng.close()
    return out


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ok = True
    for name, fn in (('fnn bf16', lambda: fnn_run('bf16', 60, 700)), ('fnn f32', lambda: fnn_run('f32', 60, 700)), ('snn bf16', lambda: snn_run(40, 900)),
                     ('ipnn bf16', lambda: ip_run(40, 4096))):
        ds = [fn() for _ in range(reps)]
        same = len(set(ds)) == 1
        ok &= same
        print(name, 'repeatable' if same else 'DIFFERS', ds if not same else ds[0])
    sys.exit(0 if ok else 1)
