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


def _p2p_same_process(ranks):
    """The exchange regions of engines living in this process: plain pointers instead of hipIpc handles."""
    handles = [e.dp_p2p_export(same_process=True) for e in ranks]
    for e in ranks:
        e.dp_p2p_attach(handles, same_process=True)


def _run_local(rows, fo, p, ids, y, r1, r2, kw, G, steps, cut, prefetch, payload=None, p2p=False, shadow=None):
    """`steps` native LOCAL-mode steps on two virtual ranks; returns per-rank (dense, table) and the per-step losses.
    payload: None (the default, slabs) / 'slabs' / 'bucket'; p2p: the dense collective is the peer-pointer all-reduce inside
    the update launch (no callback runs); shadow: per rank, the (t, field, row) list of step 0 or None."""
    import torch
    ranks = [make_engine(rows, fo, p, **kw) for _ in range(2)]
    vr = VirtualRanks(2)
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, vr.allreduce_for(r), vr.allgather_for(r), sparse='local')
        if payload is not None:
            e.dp_set_payload(payload)
    if p2p:
        _p2p_same_process(ranks)
        assert ranks[0].dp_config() == {'payload': 'bucket', 'collective': 'p2p', 'region': ranks[0].dp_config()['region']}
    dev_ids = [[torch.as_tensor(np.ascontiguousarray(ids[s * G:(s + 1) * G][c])).cuda() for s in range(steps)] for c in cut]
    losses = [[], []]

    def rank_fn(r):
        def go():
            for s in range(steps):
                sl = slice(s * G, (s + 1) * G)
                if prefetch and s + 1 < steps:
                    ranks[r].prefetch_ids(dev_ids[r][s + 1])
                if shadow is not None and s == 0 and shadow[r] is not None:
                    ranks[r].set_shadowed(shadow[r])
                out = ranks[r].train_step(dev_ids[r][s], y[sl][cut[r]], r1, r2, b_size=G)
                losses[r].append(out['loss'])
        return go
    vr.run([rank_fn(0), rank_fn(1)])
    assert vr.calls['allreduce'] == (0 if p2p else steps) and vr.calls['allgather'] == 0
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


def test_two_virtual_ranks_bag_mode_exchange(built):
    """FNN_MODE_BAG under FNN_DP_SPARSE_EXCHANGE: the step all-gathers (ids, delta rows) and every rank applies the shards one
    after the other in rank order -> the two replicas' tables are bit-identical to each other after two steps, and equal the
    single-engine steps of the global batches on EVERY row (rows both ranks touch included) up to the rounding of two partial
    sums per row instead of one."""
    from test_gpu_parity import make_snn_engine, make_snn_problem
    G, steps = 900, 2
    ww0, bb0, ids, y, p, r1, r2 = make_snn_problem(steps * G, seed=23, dup_col=4)
    full = make_snn_engine(ww0, bb0, p)
    for s in range(steps):
        full.train_step(ids[s * G:(s + 1) * G], y[s * G:(s + 1) * G], r1, r2)
    ref_dense, ref_rows, ref_bb = full.get_dense(), full.get_table(), full.get_bag_bias()
    full.close()
    ranks = [make_snn_engine(ww0, bb0, p, max_batch=512) for _ in range(2)]
    vr = VirtualRanks(2)
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, vr.allreduce_for(r), vr.allgather_for(r), sparse='exchange')
    cut = [slice(0, 500), slice(500, G)]

    def rank_fn(r):
        def go():
            for s in range(steps):
                sl = slice(s * G, (s + 1) * G)
                ranks[r].train_step(ids[sl][cut[r]], y[sl][cut[r]], r1, r2, b_size=G)
        return go
    vr.run([rank_fn(0), rank_fn(1)])
    assert vr.calls['allreduce'] == steps and vr.calls['allgather'] == 2 * steps
    touched = np.unique(ids[ids >= 0])
    change = np.abs(ref_rows[touched] - ww0[touched].astype(np.float32)).max()
    for e in ranks:
        got = e.get_table()
        assert np.abs(got[touched] - ref_rows[touched]).max() <= 3e-4 * change + 1e-7
        untouched = np.setdiff1d(np.arange(ww0.shape[0]), touched)
        assert np.array_equal(got[untouched], ref_rows[untouched])
        d = e.get_dense()
        for k in ('w1', 'b1', 'w2', 'b2', 'w3'):
            scale = np.abs(ref_dense[k] - np.asarray(p[k], np.float32)).max() + 1e-12
            assert np.abs(d[k] - ref_dense[k]).max() <= 5e-4 * scale + 1e-7, k
        assert np.abs(e.get_bag_bias() - ref_bb).max() <= 5e-4 * np.abs(ref_bb - bb0).max() + 1e-7
    assert np.array_equal(ranks[0].get_table(), ranks[1].get_table())
    assert np.array_equal(ranks[0].get_bag_bias(), ranks[1].get_bag_bias())
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


@pytest.mark.parametrize("Bg", [32768, 20000])
def test_scatter_global_at_eight_times_4096(built, Bg):
    """BASELINE configs[3] in the exact mode: a global batch of 8 x 4096 = 32,768 examples.  fnn_step_scatter_global (rocPRIM
    grouping beyond 16,384 keys per field) must leave the table the float64 oracle's sequential update leaves, given the
    same slot gradients: hot rows of the small fields are hit thousands of times (decay powers up to c^32768)."""
    import torch
    from oracle import fnn_oracle as orc
    # (20,000: a global batch that does not fill its power-of-two segment -- 12,768 invalid entries per field sort behind the rows)
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


# ----------------------------------------------------------------- round 3: what the collective carries, and who performs it
def test_world_one_bucket_payload_and_p2p_are_the_single_gpu_step(built):
    """World size 1, where every collective is the identity: the bucket payload (launch 3 without its update, RCCL all-reduce
    of the 0.5 MB bucket, k_update) and the p2p collective (k_p2p_update summing the one region) leave, after three steps, the
    bitwise state of a plain engine -- the same sums in the same order, the same update arithmetic."""
    rows, fo, ids, y, p, r1, r2 = make_problem(3 * 700, seed=52, dup_col=6)
    kw = dict(lr=0.01, lam1=0.02, lamfm=0.1)
    plain, bk, pp = (make_engine(rows, fo, p, **kw) for _ in range(3))
    bk.dp_init(0, 1, FNNEngine.dp_unique_id())
    bk.dp_set_payload('bucket')
    pp.dp_init(0, 1, FNNEngine.dp_unique_id())
    _p2p_same_process([pp])
    assert bk.dp_config()['payload'] == 'bucket' and pp.dp_config()['collective'] == 'p2p'
    for e in (bk, pp):
        e.prof_enable(True)
    for s in range(3):
        sl = slice(s * 700, (s + 1) * 700)
        a = plain.train_step(ids[sl], y[sl], r1, r2)
        for e in (bk, pp):
            assert e.train_step(ids[sl], y[sl], r1, r2, b_size=700)['loss'] == a['loss']
    assert bk.prof_get('allreduce')[1] == 3 and pp.prof_get('allreduce')[1] == 0 and pp.prof_get('p2p_update')[1] == 3
    da = plain.get_dense()
    for e in (bk, pp):
        db = e.get_dense()
        for k in da:
            assert np.array_equal(da[k], db[k]), k
        assert np.array_equal(plain.get_table(), e.get_table())
    for e in (plain, bk, pp):
        e.close()


def test_two_virtual_ranks_payload_variants(built):
    """Two virtual ranks, three LOCAL steps, slabs or bucket through the callback.  Each form tracks ONE engine stepping the
    global batch (first step: the replicas' tables still agree); inside a form both ranks hold bit-identical dense tensors
    (every rank forms the same sum).  (The peer-pointer form needs ranks in processes of their own: the rehearsal tests below.)"""
    G, steps = 1000, 3
    rows, fo, ids, y, p, r1, r2 = make_problem(steps * G, seed=65, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1)
    full = make_engine(rows, fo, p, **kw)
    ref_loss = full.train_step(ids[:G], y[:G], r1, r2)['loss']
    ref_dense = full.get_dense()
    full.close()
    cut = [slice(0, 512), slice(512, G)]
    first = {}
    for name, kws in (('slabs', dict(payload='slabs')), ('bucket', dict(payload='bucket'))):
        st1, l1 = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, 1, cut, False, **kws)
        assert abs(l1[0][0] + l1[1][0] - ref_loss) <= 2e-5 * abs(ref_loss), name
        for dense, _ in st1:
            _dense_close(dense, ref_dense, p)
        first[name] = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, steps, cut, True, **kws)
        (d0, _), (d1, _) = first[name][0]
        for k in d0:
            assert np.array_equal(d0[k], d1[k]), (name, k)


@pytest.mark.parametrize("form", ['slabs', 'bucket'])
@pytest.mark.parametrize("why", ['shadowed', 'ragged'])
def test_a_rank_on_the_layer_by_layer_path_pairs_with_fast_peers(built, form, why):
    """Round-2 advisor: the collective must not depend on a rank's own shard.  Rank 0 is pushed to the layer-by-layer kernels
    -- by shadowed features in ITS shard only, or by a shard above 4096 examples beside one below -- while rank 1 takes the
    three-launch path.  Both issue the same collective (no hang, no mismatched count) and both end with the dense tensors of
    one engine stepping the global batch."""
    import torch
    if why == 'shadowed':
        G, cut, mb = 1000, [slice(0, 512), slice(512, 1000)], 4096
    else:
        G, cut, mb = 7500, [slice(0, 4500), slice(4500, 7500)], 16384
    rows, fo, ids, y, p, r1, r2 = make_problem(G, seed=91, dup_col=6)
    kw = dict(lr=0.01, lam1=0.05, lamfm=0.1, max_batch=mb)
    shadow = None
    if why == 'shadowed':
        # rank 0's examples 3, 4 and 9 carry a second feature of field 2 (another row of that field), shadowed in the gather
        rows_f2 = np.nonzero(fo == 2)[0]
        shadow = [np.array([(t, 2, int(rows_f2[(t + 1) % len(rows_f2)])) for t in (3, 4, 9)], np.int32), None]
    full = make_engine(rows, fo, p, **kw)
    if shadow is not None:
        full.set_shadowed(shadow[0])
    ref_loss = full.train_step(ids, y, r1, r2)['loss']
    ref_dense, ref_rows = full.get_dense(), full.get_table()
    full.close()
    state, losses = _run_local(rows, fo, p, ids, y, r1, r2, kw, G, 1, cut, False, shadow=shadow, payload=form)
    assert abs(losses[0][0] + losses[1][0] - ref_loss) <= 3e-5 * abs(ref_loss)
    for dense, _ in state:
        _dense_close(dense, ref_dense, p, tol=5e-4)
    for k in state[0][0]:
        assert np.array_equal(state[0][0][k], state[1][0][k]), k
    if shadow is not None:                                   # the shadowed rows moved on rank 0 (and only there)
        sh_rows = shadow[0][:, 2]
        assert np.all(np.abs(state[0][1][sh_rows] - rows[sh_rows]).max(axis=1) > 0)
        change = np.abs(ref_rows - rows.astype(np.float32)).max()
        only0 = np.array(sorted(set(int(v) for v in sh_rows) - set(int(v) for v in np.unique(ids[cut[1]]))), np.int64)
        if len(only0):                                       # rows no example of rank 1 touched: the full-batch result
            assert np.abs(state[0][1][only0] - ref_rows[only0]).max() <= 3e-4 * change + 1e-7


def test_p2p_missing_peer_fails_loudly_and_leaves_the_weights(built, monkeypatch):
    """A peer that never reaches the step: the update launch gives up when its clock-bounded wait runs out ($FNN_P2P_TIMEOUT_MS,
    1.5 s here, 30 s by default), raises the error word, and the dense tensors keep their values; the next host read reports
    FNN_ERR_HIP instead of hanging."""
    from deep_ctr_amd.engine import FNNError
    monkeypatch.setenv('FNN_P2P_TIMEOUT_MS', '1500')
    rows, fo, ids, y, p, r1, r2 = make_problem(300, seed=93)
    ranks = [make_engine(rows, fo, p) for _ in range(2)]
    for r, e in enumerate(ranks):
        e.dp_init_custom(r, 2, lambda v: None, None, sparse='local')
    _p2p_same_process(ranks)
    before = ranks[0].get_dense()
    with pytest.raises(FNNError) as e:
        ranks[0].train_step(ids, y, r1, r2, b_size=600)      # rank 1 never steps
    assert 'peer' in str(e.value)
    after = ranks[0].get_dense()
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    for e in ranks:
        e.close()


# ----------------------------------------------------------------- two PROCESSES on the one GPU (tests/dp_rehearsal_worker.py)
def _rehearse(tmp_path, form, case):
    import subprocess
    import socket
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'tests', 'dp_rehearsal_worker.py'), '--form', form, '--case', case,
                        '--out', str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return [np.load(str(tmp_path / ('%s_%s_rank%d.npz' % (form, case, k)))) for k in (0, 1)]


def test_two_processes_p2p_equals_bucket_through_the_callback(built, tmp_path):
    """Three LOCAL steps on two rank PROCESSES sharing the GPU: the peer-pointer all-reduce (exchange regions opened through
    hipIpc handles, flags raised across the process boundary) against the bucket summed by the callback collective.  Both add
    the two ranks' buckets, so every dense tensor, both tables and the per-step losses agree BIT FOR BIT between the two forms;
    inside a form the two ranks hold identical dense tensors."""
    a, b = _rehearse(tmp_path, 'bucket', 'plain'), _rehearse(tmp_path, 'p2p', 'plain')
    for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3'):
        assert np.array_equal(b[0][k], b[1][k]), k
        for r in (0, 1):
            assert np.array_equal(a[r][k], b[r][k]), (k, r)
    for r in (0, 1):
        assert np.array_equal(a[r]['table'], b[r]['table']) and np.array_equal(a[r]['losses'], b[r]['losses'])
    assert not np.array_equal(b[0]['table'], b[1]['table'])        # LOCAL mode: each replica holds its own shard's row updates


@pytest.mark.parametrize("case", ['shadowed', 'ragged'])
def test_two_processes_p2p_with_one_rank_on_the_layer_by_layer_path(built, tmp_path, case):
    """The rank-invariance of the collective, for the peer-pointer form: rank 0 takes the layer-by-layer kernels (shadowed
    features in its shard / a shard above 4096 examples), rank 1 the three launches; both meet in the same update launch and
    end with the dense tensors the callback form produces."""
    a, b = _rehearse(tmp_path, 'bucket', case), _rehearse(tmp_path, 'p2p', case)
    for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3'):
        assert np.array_equal(b[0][k], b[1][k]) and np.array_equal(a[0][k], b[0][k]), k
    assert np.array_equal(a[0]['table'], b[0]['table'])
