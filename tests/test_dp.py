"""Data-parallel orchestration (deep-ctr_amd/dp.py) over gloo, world_size 2, on CPU.

The HIP engine cannot run here, so a test double with the engine's step_begin / step_end contract
is built on the float64 oracle (tests may use the oracle; the product never does).  What is
checked is the host logic the N>1 path adds: contiguous sharding, ONE all-reduce of the flat
dense bucket, identical dropout rows on every rank, the GLOBAL batch length in the sparse decay,
and local sparse updates."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fnn_oracle as orc

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import synth
from deep_ctr_amd.dp import DataParallelFNN, shard_bounds

F, K, H1, H2 = 16, 3, 12, 7
XDIM = 1 + F * K
NAMES = ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')


def test_shard_bounds_cover_the_batch():
    for n, w in ((4096, 8), (10, 3), (7, 8), (100, 1)):
        cuts = [shard_bounds(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


class OracleEngine(object):
    """step_begin / step_end / grad bucket of FNNEngine, on the CPU oracle."""
    stream = None

    def __init__(self, p, rows, w0, lr, lam1, lamfm):
        self.p, self.rows, self.w0, self.lr, self.lam1, self.lamfm = p, rows, w0, lr, lam1, lamfm
        self.sizes = [int(np.asarray(p[k]).size) for k in NAMES]
        self.bucket = torch.zeros(sum(self.sizes), dtype=torch.float64)
        self.pend = None

    def step_begin(self, ids, y, m1, m2, b_size=0):
        x = orc.gather(self.rows, ids, self.w0)
        self.loss, _, g = orc.loss_and_grads(self.p, x, y, m1, m2, self.lam1)
        # the L2 term is not a per-example sum: keep it out of the all-reduced bucket
        g['w3'] = g['w3'] - 2 * self.lam1 * self.p['w3']
        g['b3'] = g['b3'] - 2 * self.lam1 * self.p['b3']
        self.bucket.copy_(torch.from_numpy(np.concatenate([np.asarray(g[k], dtype=np.float64).ravel() for k in NAMES])))
        self.pend = (np.asarray(ids), g['x'], b_size if b_size > 0 else None)
        return self.bucket

    def step_scatter(self):
        if self.pend is not None:
            ids, gx, b_size = self.pend
            orc.scatter_sgd(self.rows, ids, gx, self.lr, self.lamfm, b_size)
            self.pend = None

    # exact mode: the engine exposes gx (reference layout here; the HIP engine uses its slot layout)
    def sparse_grad(self, B):
        return torch.from_numpy(np.ascontiguousarray(self.pend[1][:B]))

    def step_scatter_global(self, ids_g, gx_g):
        ids_g, gx_g = ids_g.numpy(), gx_g.numpy()
        live = ids_g[:, 0] >= 0                               # padding rows carry -1 ids
        orc.scatter_sgd(self.rows, ids_g[live], gx_g[live], self.lr, self.lamfm, self.pend[2])
        self.pend = None

    def step_end(self, want_loss=False):
        self.step_scatter()
        flat, off = self.bucket.numpy(), 0
        for k, n in zip(NAMES, self.sizes):
            g = flat[off:off + n].reshape(np.shape(self.p[k]))
            if k in ('w3', 'b3'):
                g = g + 2 * self.lam1 * self.p[k]
            self.p[k] = self.p[k] - self.lr * (g if k != 'b3' else float(g))
            off += n
        return self.loss if want_loss else None


def _problem():
    sizes = [6] * F
    rng = np.random.RandomState(0)
    rows = rng.standard_normal((sum(sizes), K)) * 0.3
    ids = synth.zipf_ids(40, sizes, 1.1, 1)
    y = (rng.uniform(size=40) < 0.4).astype(np.float64)
    p = {'w1': rng.standard_normal((XDIM, H1)) * 0.3, 'b1': np.zeros(H1), 'w2': rng.standard_normal((H1, H2)) * 0.3,
         'b2': np.zeros(H2), 'w3': rng.standard_normal(H2) * 0.3, 'b3': 0.1}
    masks = [((rng.uniform(size=H1) < 0.5).astype(np.float64), (rng.uniform(size=H2) < 0.5).astype(np.float64))
             for _ in range(2)]
    return rows, ids, y, p, masks


def _worker(rank, world, port, out_dir, sparse='local', per_step=20):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rows, ids, y, p, masks = _problem()
    eng = OracleEngine(p, rows, -1.0, 0.05, 0.01, 0.2)
    dp = DataParallelFNN(eng, sparse=sparse)
    losses = []
    for step in range(2):
        sl = slice(step * per_step, (step + 1) * per_step)
        losses.append(dp.train_step(ids[sl], y[sl], masks[step][0], masks[step][1], want_loss=True))
        if step == 0:
            snap = {'s0_' + k: np.array(eng.p[k]) for k in NAMES}
            snap['s0_rows'] = eng.rows.copy()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), rows=eng.rows, losses=np.array(losses),
             **{k: np.asarray(eng.p[k]) for k in NAMES}, **snap)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_dp_matches_single_process_dense_and_shards_sparse(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')

    # single process, full batch: after the first step (identical tables everywhere) the dense
    # tensors must agree with it (the loss is a SUM, python/FNN_wnzh.py:173) and so must the loss
    rows, ids, y, p, masks = _problem()
    x = orc.gather(rows, ids[:20], -1.0)
    gx, _, loss, _, _ = orc.train_call(p, x, y[:20], masks[0][0], masks[0][1], 0.05, 0.01)
    assert abs(loss - r0['losses'][0]) < 1e-9 and abs(loss - r1['losses'][0]) < 1e-9
    for k in NAMES:
        np.testing.assert_allclose(r0['s0_' + k], np.asarray(p[k]), rtol=1e-10, atol=1e-12)
    # each rank applied only its shard's sparse updates, with the GLOBAL batch length (20) in the decay
    for rank, res in ((0, r0), (1, r1)):
        lo, hi = shard_bounds(20, 2, rank)
        mine = rows.copy()
        orc.scatter_sgd(mine, ids[:20][lo:hi], gx[lo:hi], 0.05, 0.2, b_size=20)
        np.testing.assert_allclose(res['s0_rows'], mine, rtol=1e-12, atol=1e-14)
    # replicas: dense tensors identical across ranks after every step, tables differ (documented)
    for k in NAMES:
        np.testing.assert_allclose(r0[k], r1[k], rtol=0, atol=0)
    assert not np.array_equal(r0['rows'], r1['rows'])

    # replay both ranks deterministically to pin the final tables
    for rank, res in ((0, r0), (1, r1)):
        rows_r, ids_r, y_r, p_r, masks_r = _problem()
        engs = [OracleEngine(_problem()[3], _problem()[0], -1.0, 0.05, 0.01, 0.2) for _ in range(2)]
        for step in range(2):
            sl = slice(step * 20, (step + 1) * 20)
            buckets = []
            for r in range(2):
                lo, hi = shard_bounds(20, 2, r)
                buckets.append(engs[r].step_begin(ids_r[sl][lo:hi], y_r[sl][lo:hi], masks_r[step][0],
                                                  masks_r[step][1], b_size=20).clone())
            tot = buckets[0] + buckets[1]
            for r in range(2):
                engs[r].bucket.copy_(tot)
                engs[r].step_end()
        np.testing.assert_allclose(res['rows'], engs[rank].rows, rtol=1e-12, atol=1e-14)
        for k in NAMES:
            np.testing.assert_allclose(res[k], np.asarray(engs[rank].p[k]), rtol=1e-12, atol=1e-14)


@pytest.mark.timeout(180)
@pytest.mark.parametrize("per_step", [20, 19])          # 19: unequal shards (10 + 9), padded in the exchange
def test_two_rank_exchange_mode_equals_the_single_process_run(tmp_path, per_step):
    """sparse='exchange' (parity mode): all-gather of (ids, gx), every rank applies the
    whole batch's row updates in global example order -> tables AND dense tensors of both replicas
    equal the single-process run after every step."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'exchange', per_step), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    rows, ids, y, p, masks = _problem()
    losses = []
    for step in range(2):
        sl = slice(step * per_step, (step + 1) * per_step)
        x = orc.gather(rows, ids[sl], -1.0)
        gx, _, loss, _, _ = orc.train_call(p, x, y[sl], masks[step][0], masks[step][1], 0.05, 0.01)
        orc.scatter_sgd(rows, ids[sl], gx, 0.05, 0.2)
        losses.append(loss)
    for res in (r0, r1):
        np.testing.assert_allclose(res['losses'], losses, rtol=1e-10)
        np.testing.assert_allclose(res['rows'], rows, rtol=1e-11, atol=1e-13)
        for k in NAMES:
            np.testing.assert_allclose(res[k], np.asarray(p[k]), rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(r0['rows'], r1['rows'])
