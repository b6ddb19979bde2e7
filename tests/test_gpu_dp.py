"""GPU tests of the NATIVE data-parallel step (include/fnn_hip.h: fnn_dp_init / fnn_dp_init_custom): the ordinary
fnn_train_step with ONE collective between its second and third launch.

One GPU is all a test box has, so several ranks are stood for in two ways:
  * RCCL itself with world size 1 (the real communicator, the real ncclAllReduce on the engine's stream);
  * two VIRTUAL ranks: two engines on the one GPU, each driven by its own host thread, joined by
    fnn_dp_init_custom callbacks that meet at a barrier and sum / concatenate the two ranks' buffers -- what the
    all-reduce / all-gather do.  The kernels, launch order and buffer handling are the ones an 8-GPU job runs.
Reference for every comparison: ONE engine stepping the whole global batch (f32 mode), itself checked against the
float64 oracle in test_gpu_parity.py.
"""
import threading

import numpy as np
import pytest

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import synth
from deep_ctr_amd.engine import FNNEngine

from test_gpu_parity import F, K, make_engine, make_problem

pytestmark = pytest.mark.gpu


class VirtualRanks(object):
    """Collectives between `world` engines living in one process, one host thread per engine."""

    def __init__(self, world):
        import torch
        self.torch, self.world = torch, world
        self.bar = threading.Barrier(world)
        self.views = [None] * world
        self.sends = [None] * world
        self.calls = {'allreduce': 0, 'allgather': 0}

    def allreduce_for(self, rank):
        def fn(view):
            torch = self.torch
            torch.cuda.current_stream().synchronize()           # this rank's producers are done
            self.views[rank] = view
            self.bar.wait()
            tot = self.views[0].clone()
            for v in self.views[1:]:
                tot += v
            torch.cuda.current_stream().synchronize()
            self.bar.wait()                                      # everyone has read every buffer
            view.copy_(tot)
            torch.cuda.current_stream().synchronize()
            self.bar.wait()
            if rank == 0:
                self.calls['allreduce'] += 1
        return fn

    def allgather_for(self, rank):
        def fn(send, recv):
            torch = self.torch
            torch.cuda.current_stream().synchronize()
            self.sends[rank] = send
            self.bar.wait()
            recv.copy_(torch.cat(self.sends))
            torch.cuda.current_stream().synchronize()
            self.bar.wait()
            if rank == 0:
                self.calls['allgather'] += 1
        return fn

    def run(self, fns):
        errs = []

        def wrap(fn):
            def go():
                try:
                    fn()
                except BaseException as e:       # noqa: B902 -- reported to the main thread
                    errs.append(e)
                    self.bar.abort()
            return go
        ths = [threading.Thread(target=wrap(fn)) for fn in fns]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ths), "a virtual rank hung"
        if errs:
            raise errs[0]


def _dense_close(d, ref, p, tol=3e-4):
    for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
        scale = np.abs(ref[k] - np.asarray(p[k], np.float32)).max() + 1e-12
        assert np.abs(d[k] - ref[k]).max() <= tol * scale + 1e-7, k
    assert abs(d['b3'] - ref['b3']) <= tol * abs(ref['b3'] - float(p['b3'])) + 1e-6


def test_rccl_world_one_is_the_single_gpu_step(built):
    """fnn_dp_init with the real RCCL communicator (1 rank): the all-reduce of the slabs is the identity, so three steps
    leave the bitwise state of a plain engine -- and the collective did run on the engine's stream (profiling slot)."""
    rows, fo, ids, y, p, r1, r2 = make_problem(3 * 700, seed=51, dup_col=6)
    kw = dict(lr=0.01, lam1=0.02, lamfm=0.1)
    plain, dp = make_engine(rows, fo, p, **kw), make_engine(rows, fo, p, **kw)
    dp.dp_init(0, 1, FNNEngine.dp_unique_id())
    dp.prof_enable(True)
    for s in range(3):
        sl = slice(s * 700, (s + 1) * 700)
        a = plain.train_step(ids[sl], y[sl], r1, r2)
        b = dp.train_step(ids[sl], y[sl], r1, r2, b_size=700)
        assert a['loss'] == b['loss']
    assert dp.prof_get('allreduce')[1] == 3
    da, db = plain.get_dense(), dp.get_dense()
    for k in da:
        assert np.array_equal(da[k], db[k]), k
    assert np.array_equal(plain.get_table(), dp.get_table())
    dp.dp_shutdown()                                             # back to single-process steps
    a = plain.train_step(ids[:700], y[:700], r1, r2)
    b = dp.train_step(ids[:700], y[:700], r1, r2)
    assert a['loss'] == b['loss'] and np.array_equal(plain.get_table(), dp.get_table())
    plain.close(); dp.close()


def _run_local(rows, fo, p, ids, y, r1, r2, kw, G, steps, cut, prefetch):
    """`steps` native LOCAL-mode steps on two virtual ranks; returns per-rank (dense, table) and the per-step losses."""
    import torch
    ranks = [make_engine(rows, fo, p, **kw) for _ in range(2)]
    vr = VirtualRanks(2)
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, vr.allreduce_for(r), vr.allgather_for(r), sparse='local')
    dev_ids = [[torch.as_tensor(np.ascontiguousarray(ids[s * G:(s + 1) * G][c])).cuda() for s in range(steps)] for c in cut]
    losses = [[], []]

    def rank_fn(r):
        def go():
            for s in range(steps):
                sl = slice(s * G, (s + 1) * G)
                if prefetch and s + 1 < steps:
                    ranks[r].prefetch_ids(dev_ids[r][s + 1])
                out = ranks[r].train_step(dev_ids[r][s], y[sl][cut[r]], r1, r2, b_size=G)
                losses[r].append(out['loss'])
        return go
    vr.run([rank_fn(0), rank_fn(1)])
    assert vr.calls['allreduce'] == steps and vr.calls['allgather'] == 0
    state = [(e.get_dense(), e.get_table()) for e in ranks]
    for e in ranks:
        e.close()
    return state, losses


def test_two_virtual_ranks_local_sparse(built):
    """LOCAL mode, one step: dense tensors of both ranks equal the single-engine step of the global batch (also with an L2
    term, which must not be summed over ranks); a rank's table holds its own shard's row updates with the GLOBAL batch
    length in the decay.  (After the first step the replicas' tables differ by construction -- DESIGN.md section 6 -- so a
    second step has no single-engine reference: see the next test.)"""
    G = 1000
    rows, fo, ids, y, p, r1, r2 = make_problem(G, seed=61, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1)
    full = make_engine(rows, fo, p, **kw)
    ref_loss = full.train_step(ids, y, r1, r2)['loss']
    ref_dense, ref_rows = full.get_dense(), full.get_table()
    full.close()
    cut = [slice(0, 512), slice(512, G)]                            # unequal shards: 512 + 488
    state, losses = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, 1, cut, False)
    assert abs(losses[0][0] + losses[1][0] - ref_loss) <= 2e-5 * abs(ref_loss)
    for dense, _ in state:
        _dense_close(dense, ref_dense, p)
    # rows only ONE rank touched equal the full-batch result on that rank, and are untouched on the other
    t = [set(np.unique(ids[c])) for c in cut]
    only0, only1 = np.array(sorted(t[0] - t[1])), np.array(sorted(t[1] - t[0]))
    change = np.abs(ref_rows - rows.astype(np.float32)).max()
    assert np.abs(state[0][1][only0] - ref_rows[only0]).max() <= 3e-4 * change + 1e-7
    assert np.abs(state[1][1][only1] - ref_rows[only1]).max() <= 3e-4 * change + 1e-7
    assert np.array_equal(state[0][1][only1], rows[only1].astype(np.float32))


def test_two_virtual_ranks_local_sparse_prefetch_changes_nothing(built):
    """Three LOCAL-mode steps with the next batch's grouping riding on the step's launches (fnn_prefetch_ids) leave bit for
    bit the state of three steps without it, on both ranks."""
    G, steps = 1000, 3
    rows, fo, ids, y, p, r1, r2 = make_problem(steps * G, seed=63, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1)
    cut = [slice(0, 512), slice(512, G)]
    a, la = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, steps, cut, False)
    b, lb = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, steps, cut, True)
    assert la == lb
    for (da, ta), (db, tb) in zip(a, b):
        assert np.array_equal(ta, tb)
        for k in da:
            assert np.array_equal(da[k], db[k]), k


def test_two_virtual_ranks_layer_by_layer_path(built):
    """Batches above 4096 examples take the layer-by-layer kernels; under native data parallelism their flat dense-gradient
    bucket is what the library all-reduces (before its update kernel).  Two virtual ranks of 4,500 + 4,300 examples against
    one engine stepping the 8,800; EXCHANGE has no layer-by-layer form and says so."""
    G = 8800
    rows, fo, ids, y, p, r1, r2 = make_problem(G, seed=81, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1, max_batch=16384)
    full = make_engine(rows, fo, p, **kw)
    ref_loss = full.train_step(ids, y, r1, r2)['loss']
    ref_dense, ref_rows = full.get_dense(), full.get_table()
    full.close()
    cut = [slice(0, 4500), slice(4500, G)]
    state, losses = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, 1, cut, False)
    assert abs(losses[0][0] + losses[1][0] - ref_loss) <= 3e-5 * abs(ref_loss)
    for dense, _ in state:
        _dense_close(dense, ref_dense, p, tol=5e-4)
    t = [set(np.unique(ids[c])) for c in cut]
    only0 = np.array(sorted(t[0] - t[1]))
    change = np.abs(ref_rows - rows.astype(np.float32)).max()
    assert np.abs(state[0][1][only0] - ref_rows[only0]).max() <= 3e-4 * change + 1e-7
    eng = make_engine(rows, fo, p, **kw)
    with pytest.raises(Exception) as e:
        eng.dp_init_custom(0, 2, lambda v: None, lambda a, b: None, sparse='exchange')
    assert 'three-launch' in str(e.value)
    eng.close()


def test_two_virtual_ranks_bag_mode_native(built):
    """The SNN fine-tune step (FNN_MODE_BAG) shards through the same native step: its slabs also carry the bag-bias gradient.
    Dense tensors and bb0 of both ranks equal the single-engine step of the global batch; rows one rank alone touched equal
    the full-batch result there."""
    from test_gpu_parity import make_snn_engine, make_snn_problem
    G = 900
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(G, seed=19, dup_col=4)
    full = make_snn_engine(ww0, bb0, p)
    ref_loss = full.train_step(ids, y, r1, r2)['loss']
    ref_dense, ref_rows, ref_bb = full.get_dense(), full.get_table(), full.get_bag_bias()
    full.close()
    ranks = [make_snn_engine(ww0, bb0, p) for _ in range(2)]
    vr = VirtualRanks(2)
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, vr.allreduce_for(r), None, sparse='local')
    cut = [slice(0, 512), slice(512, G)]
    losses = [None, None]

    def rank_fn(r):
        def go():
            losses[r] = ranks[r].train_step(ids[cut[r]], y[cut[r]], r1, r2, b_size=G)['loss']
        return go
    vr.run([rank_fn(0), rank_fn(1)])
    assert abs(losses[0] + losses[1] - ref_loss) <= 3e-5 * abs(ref_loss)
    for e in ranks:
        d = e.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(ref_dense[k] - np.asarray(p[k], np.float32)).max() + 1e-12
            assert np.abs(d[k] - ref_dense[k]).max() <= 5e-4 * scale + 1e-7, k
        assert np.abs(e.get_bag_bias() - ref_bb).max() <= 5e-4 * np.abs(ref_bb - bb0).max() + 1e-7
    t = [set(np.unique(ids[c])) for c in cut]
    only0 = np.array(sorted(t[0] - t[1]))
    np.testing.assert_allclose(ranks[0].get_table()[only0], ref_rows[only0], rtol=1e-5, atol=2e-7)
    for e in ranks:
        e.close()


def test_two_virtual_ranks_exchange_keeps_replicas_identical(built):
    """EXCHANGE mode: the step all-gathers (ids, gx') of the shards and every rank applies the global batch's row updates in
    global example order -> both tables equal the single-engine run on every row, bit for bit with each other."""
    G, steps = 900, 2
    rows, fo, ids, y, p, r1, r2 = make_problem(steps * G, seed=71, dup_col=6)
    kw = dict(lr=0.01, lam1=0.0, lamfm=0.1, max_batch=512)
    full = make_engine(rows, fo, p, lr=0.01, lam1=0.0, lamfm=0.1)
    ranks = [make_engine(rows, fo, p, **kw) for _ in range(2)]
    vr = VirtualRanks(2)
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, vr.allreduce_for(r), vr.allgather_for(r), sparse='exchange')
    cut = [slice(0, 500), slice(500, G)]                            # 500 + 400 examples, blocks of 512 rows in the exchange

    def rank_fn(r):
        def go():
            for s in range(steps):
                sl = slice(s * G, (s + 1) * G)
                ranks[r].train_step(ids[sl][cut[r]], y[sl][cut[r]], r1, r2, b_size=G)
        return go
    for s in range(steps):
        sl = slice(s * G, (s + 1) * G)
        full.train_step(ids[sl], y[sl], r1, r2)
    vr.run([rank_fn(0), rank_fn(1)])
    assert vr.calls['allreduce'] == steps and vr.calls['allgather'] == 2 * steps
    ref_rows, ref_dense = full.get_table(), full.get_dense()
    touched = np.unique(ids)
    change = np.abs(ref_rows[touched] - rows[touched].astype(np.float32)).max()
    for e in ranks:
        got = e.get_table()
        assert np.abs(got[touched] - ref_rows[touched]).max() <= 3e-4 * change + 1e-7
        untouched = np.setdiff1d(np.arange(rows.shape[0]), touched)
        assert np.array_equal(got[untouched], ref_rows[untouched])
        _dense_close(e.get_dense(), ref_dense, p, tol=5e-4)
    assert np.array_equal(ranks[0].get_table(), ranks[1].get_table())
    for e in ranks + [full]:
        e.close()


def test_scatter_global_at_eight_times_4096(built):
    """BASELINE configs[3] in the exact mode: a global batch of 8 x 4096 = 32,768 examples.  fnn_step_scatter_global (rocPRIM
    grouping beyond 16,384 keys per field) must leave the table the float64 oracle's sequential update leaves, given the
    same slot gradients: hot rows of the small fields are hit thousands of times (decay powers up to c^32768)."""
    import torch
    from oracle import fnn_oracle as orc
    Bg = 32768
    sizes = synth.field_sizes_tiny(3000)
    rows = synth.fm_table(sum(sizes), K, 0.05, 5)
    fo = synth.field_of_row(sizes)
    ids = synth.zipf_ids(Bg, sizes, 1.1, 6)
    ids[100:140, 3] = -1                                             # some empty slots
    rng = np.random.RandomState(7)
    gxp = np.zeros((Bg, 256), np.float32)
    gx = (rng.standard_normal((Bg, 1 + F * K)) * 0.01).astype(np.float32)
    for f in range(F):
        gxp[:, 16 * f:16 * f + K] = gx[:, 1 + f * K:1 + (f + 1) * K]
    _, _, _, _, p, r1, r2 = make_problem(8, seed=1)
    lr, lamfm = 0.01, 0.1
    eng = make_engine(rows, fo, p, lr=lr, lamfm=lamfm, max_batch=256)
    eng.step_begin(ids[:256], np.zeros(256, np.float32), r1, r2, b_size=Bg)      # opens a step; its own rows are not applied
    eng.step_scatter_global(torch.as_tensor(ids).cuda().contiguous(), torch.as_tensor(gxp).cuda().contiguous())
    eng.step_end()
    eng.sync()
    ref = rows.astype(np.float64)
    orc.scatter_sgd(ref, ids, gx.astype(np.float64), lr, lamfm, Bg)
    got = eng.get_table()
    change = np.abs(ref - rows).max()
    assert np.abs(got - ref).max() <= 2e-6 * max(change, np.abs(ref).max())
    untouched = np.setdiff1d(np.arange(rows.shape[0]), np.unique(ids[ids >= 0]))
    assert np.array_equal(got[untouched], rows[untouched].astype(np.float32))
    eng.close()
