#!/usr/bin/env python
"""bench.py -- examples/sec of the FNN hot path on N MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): FNN L3, 16 fields, 937,670 one-hot
dims, k=10 (row width K=11), hidden 300/100 tanh, batch 4096 per GPU, bf16 MFMA with f32
accumulation and f32 master weights, synthetic Zipf(1.1) ids, inputs resident in HBM.
A "step" is one pass of the reference's hot loop body (python/FNN_wnzh.py:296-306): gather ->
train(x, y) -> dense SGD -> sparse-row SGD, for one batch, through the C ABI of libfnn_hip.so.
N > 1: one process per GPU, batch sharded data-parallel (weak scaling: 4096 examples per GPU), the
library's native step: the same three launches with ONE RCCL all-reduce (the split-K weight-gradient
slabs, on the library's own stream) between the second and the third.  `python bench.py --gpus N`
without a torchrun environment starts the N ranks itself (fresh child processes, before this process
touches the GPU) and fails if it cannot.

The default N = 1 run also times, briefly, the f32 parity mode of the same step and the other BASELINE
configs (SNN fine-tune and pre-training, FNN_IP_L7, the standalone gathers) and nests them under
`precision_f32` / `extra_workloads` of the ONE JSON line rank 0 prints.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F, K, H1, H2 = 16, 11, 300, 100
XDIM = 1 + F * K
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# bf16x3: three bf16 MFMAs per product.  The strip kernel (k_step1) issues them as v_mfma_f32_16x16x32_bf16 over pairs of k-steps
# (a third of the bf16 rate); the weight-gradient role still issues v_mfma_f32_16x16x16_bf16, whose nominal rate is half of that
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 6}
MFMA_PEAK_BY_KERNEL = {('bf16x3', 'step1'): 2500.0 / 3, ('bf16x3', 'mlp'): 2500.0 / 3}
# north_star's parity bar (|d logloss|, |d AUC| <= 1e-4 against the float64 oracle on the demo set): which precision modes meet it
# (tests/test_gpu_parity.py::test_demo_epochs_f32_logloss_auc_within_1e4, tests/test_gpu_bf16x3.py; bf16: profiles/r02b_bf16_demo_deltas.json)
MEETS_PARITY_BAR = {'f32': True, 'bf16x3': True, 'bf16': False}

# ALGORITHMIC bytes / flops per example of each kernel
ALGO = {
    'gather':   ('hbm', 64 + 704 + 708),                 # ids + 16 rows of 44 B + x (177 f32)
    'scatter':  ('hbm', 704 + 704 + 704 + 64),           # gx + row read + row write + ids (scatter+finalize)
    'fwd1':     ('mfma', 2 * XDIM * H1),
    'fwd2':     ('mfma', 2 * H1 * H2),
    'bwd1':     ('mfma', 2 * H1 * H2),
    'gx':       ('mfma', 2 * XDIM * H1),
    'wgrad':    ('mfma', 2 * (XDIM * H1 + H1 * H2)),
    'head':     ('hbm', H2 * 2 * 3 + 8),                 # d2 read, delta2 written in two layouts, y, p
    # fused strip kernel: forward (x.w1, d1.w2, d2.w3) + backward-data (delta2.w2^T, delta1.w1^T)
    'mlp':      ('mfma', 2 * (XDIM * H1 + H1 * H2 + H2) + 2 * (H1 * H2 + XDIM * H1)),
    # the three launches of the fast path (fnn_step_kernels.hip.h); each is priced by its main role
    'step1':    ('mfma', 2 * (XDIM * H1 + H1 * H2 + H2) + 2 * (H1 * H2 + XDIM * H1)),   # MLP strips (+ next sort)
    'step2':    ('hbm', 704 + 704 + 704 + 64),                                           # scatter L1 (+ wgrad)
    'step3':    ('hbm', 0),
}
STEP_MIN_BYTES = 2180                                    # fused train-step minimum, B/example


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as FRESH child processes (this process has not touched the
    GPU and does not from here on), pass their output through, exit with their code."""
    import socket
    import subprocess
    import torch
    rehearse = bool(os.environ.get('FNN_BENCH_REHEARSE'))
    ndev = torch.cuda.device_count()                      # counting devices does not initialise the GPU
    if not rehearse and ndev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (set FNN_BENCH_REHEARSE=1 for a control-flow rehearsal of "
                         "the ranks on one GPU over gloo)" % (args.gpus, ndev))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4096, help='examples per GPU per step')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32', 'bf16x3'],
                    help='bf16x3 (operands as bf16 pairs, three MFMAs per product): the fnn / snn / e2e workloads')
    ap.add_argument('--optimizer', default='sgd', choices=['sgd', 'adam', 'ftrl'],
                    help='--workload ipnn only: sgd (BASELINE configs[2]), or the reference family\'s adam / ftrl (dense table pass per step)')
    ap.add_argument('--workload', default='fnn', choices=['fnn', 'snn', 'ipnn', 'gather', 'rbm', 'e2e', 'pretrain'],
                    help='fnn: BASELINE configs[1] (default; at N = 1 the other workloads ride along as extra_workloads).  '
                         'snn: the SNN fine-tune step of configs[4] (H0=200 bag rows).  '
                         'ipnn: FNN_IP_L7 train step of configs[2] (7 hidden layers, MFMA stack).  '
                         'gather: the standalone embedding gathers (A3: FM rows; A8: 200-wide bag rows) against the HBM roofline.  '
                         'rbm: SNN pre-training of configs[4] -- the exact online sparse CD-1 pass and a dense CD-1 layer.  '
                         'e2e: one epoch of `python FNN.py` end to end on synthetic TEXT files of the config shape: native parse of '
                         'fm.model.txt and train.fm.txt, the training steps, the evaluation pass (row N1 of SURVEY 8f)')
    ap.add_argument('--e2e-lines', type=int, default=1 << 20, help='--workload e2e: lines of the synthetic train.fm.txt')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='headline workload only (no precision_f32 / extra_workloads legs)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="weak: --batch examples per GPU (the contract's default); strong: --batch is the GLOBAL batch, split over the ranks "
                         "(SURVEY 8d config 4, '4096 split 8-way')")
    ap.add_argument('--dp-sparse', default='local', choices=['local', 'exchange'],
                    help='N > 1: local = every rank applies its own shard\'s row updates (north_star); exchange = the exact mode')
    ap.add_argument('--dp-collective', default='rccl', choices=['rccl', 'p2p'],
                    help='N > 1: who performs the dense all-reduce of the step -- RCCL on the library stream, or the one-shot '
                         'peer-pointer all-reduce inside the update launch (hipIpc-mapped exchange regions)')
    ap.add_argument('--dp-payload', default=None, choices=['slabs', 'bucket'],
                    help='N > 1, --dp-collective rccl: what the all-reduce carries -- the split-K slabs (2 MB, three launches) or the flat '
                         'bucket (0.5 MB, four launches: the default at N > 1, or $FNN_DP_PAYLOAD); p2p always carries the bucket')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        if args.workload not in ('fnn', 'snn'):   # (e2e, pretrain included: one process)
            raise SystemExit("--gpus %d: only the fnn / snn steps shard (replicas of %s are not launched)" % (args.gpus, args.workload))
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or without torchrun: bench.py starts the "
                         "ranks itself)" % (args.gpus, world, args.gpus))
    if args.precision == 'bf16x3' and args.workload in ('ipnn', 'rbm', 'gather', 'pretrain'):
        raise SystemExit("--precision bf16x3 is the FNN / SNN engine's (workloads fnn, snn, e2e)")
    if args.workload == 'ipnn':
        out = bench_ipnn(args)
    elif args.workload == 'gather':
        out = bench_gather(args)
    elif args.workload == 'rbm':
        out = bench_rbm(args)
    elif args.workload == 'e2e':
        out = bench_e2e(args)
    elif args.workload == 'pretrain':
        out = bench_pretrain(args)
    else:
        out = bench_fnn(args, args.precision, args.workload == 'snn')
        if out is not None and world == 1 and args.workload == 'fnn' and not args.no_extras:
            import copy
            short = copy.copy(args)
            # the steps of these legs cost microseconds (their set-up dominates): same step counts as the headline, shorter CPU samples
            short.cpu_seconds = min(args.cpu_seconds, 4.0)

            def leg(fn, *a):
                try:
                    r = fn(*a)
                    return {k: r.get(k) for k in ('value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline', 'cpu_baseline', 'kernel_ms',
                                                  'train_logloss_last_step', 'meets_parity_bar_1e-4', 'fm_gather_uniform', 'sparse_minibatch_4096', 'dense_cd1_200x300',
                                                  'bag_gather_zipf', 'fm_gather', 'fm_gather_100k', 'bag_gather_100k', 'dae_online', 'phases_s', 'ingest', 'train_examples_per_s',
                                                  'eval_examples_per_s') if k in r}
                except Exception as e:                    # an extra leg must not cost the headline line
                    return {'error': '%s: %s' % (type(e).__name__, e)}
            if args.precision == 'bf16':
                # the mode that meets the 1e-4 parity bar (tests/test_gpu_parity.py::test_demo_epochs_f32_logloss_auc_within_1e4), timed
                # on the same workload beside the bf16 headline
                short.no_cpu_baseline = True
                out['precision_f32'] = leg(bench_fnn, short, 'f32', False)
                # ... and the bf16-pair mode, which meets it too (tests/test_gpu_bf16x3.py) at 4.4x the f32 MFMA rate
                out['precision_bf16x3'] = leg(bench_fnn, short, 'bf16x3', False)
                short.no_cpu_baseline = args.no_cpu_baseline
                # which number is the parity-grade one: the headline precision (bf16) misses north_star's 1e-4 bar on the demo set
                # (logloss 1.3e-4, AUC 1.7e-3); the two legs beside it meet it
                out['parity_grade'] = {'bar': '|d logloss| and |d AUC| <= 1e-4 against the float64 oracle, demo set, 3 epochs (north_star)',
                                       'headline_dtype': 'bf16', 'headline_meets_bar': False,
                                       'legs_meeting_bar': {k: (out[k].get('value') if isinstance(out.get(k), dict) else None)
                                                            for k in ('precision_bf16x3', 'precision_f32')}}
            ex = {}
            ex['snn_finetune'] = leg(bench_fnn, short, args.precision, True)
            ip = copy.copy(short); ip.steps, ip.warmup = min(args.steps, 50), min(args.warmup, 5)
            ex['fnn_ip_l7'] = leg(bench_ipnn, ip)
            ex['gather'] = leg(bench_gather, short)
            ex['snn_pretrain_rbm'] = leg(bench_rbm, short)
            ex['fm_and_dae_pretrain'] = leg(bench_pretrain, short)
            e2 = copy.copy(short); e2.e2e_lines = min(args.e2e_lines, 1 << 18)
            ex['fnn_script_epoch_from_text'] = leg(bench_e2e, e2)
            out['extra_workloads'] = ex
            g = ex['gather']
            if 'error' not in g and out.get('roofline') is not None:
                # north_star: the gather's achieved fraction of the HBM roofline.  standalone kernels measured in this run;
                # the gather phase INSIDE the fused strip kernel from the committed per-phase s_memtime stamps (profiles/)
                keys = ('achieved', 'frac', 'traffic', 'hbm_frac_from_counters', 'unit', 'avg_launch_ms', 'algorithmic_per_example', 'ids', 'examples_per_launch')
                pick = lambda d: {k: d.get(k) for k in keys}           # noqa: E731
                out['roofline']['gather'] = {
                    'standalone_A3_fm_rows': pick(g['fm_gather']), 'standalone_A3_fm_rows_uniform_ids': pick(g['fm_gather_uniform']),
                    'standalone_A8_bag_rows_uniform_ids': pick(g['roofline']), 'standalone_A8_bag_rows_zipf_ids': pick(g['bag_gather_zipf']),
                    'standalone_A3_fm_rows_100k_per_launch': pick(g['fm_gather_100k']),
                    'standalone_A8_bag_rows_uniform_ids_100k_per_launch': pick(g['bag_gather_100k']),
                    'note': 'frac = ALGORITHMIC bytes / launch time / peak; hbm_frac_from_counters = counter bytes (committed PMC passes) / launch '
                            'time / peak. The 60 MB FM table is Infinity-Cache resident and Zipf ids repeat: where the two differ, the counters '
                            'say what HBM delivered',
                    'in_step_A3': in_step_gather(args.batch), 'peak': HBM_PEAK_GBS}
    if out is not None:
        print(json.dumps(out))
    teardown_dist()


_DIST = None


def teardown_dist():
    global _DIST
    if _DIST is not None:
        _DIST.barrier()
        _DIST.destroy_process_group()
        _DIST = None


def in_step_gather(B):
    """Gather phase of the fused strip kernel k_step1 (P0: ids, then rows, into the LDS tile), from the newest committed
    per-phase s_memtime stamps (profiles/*_step1_phases.json; tools/exp/mlp_stamps.hip, a diagnostic build -- the product kernel
    executes no stamp).  Algorithmic bytes: 64 (ids) + 704 (rows) per example; x is never materialised."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_step1_phases.json')))
    if not files:
        return None
    d = json.load(open(files[-1]))
    us = d.get('gather_phase_us_median')
    if not us:
        return None
    ach = (64 + 704) * B / (us * 1e-6) / 1e9
    return {'phase_us_median': us, 'achieved': ach, 'frac': ach / HBM_PEAK_GBS, 'unit': 'GB/s', 'algorithmic_per_example': 64 + 704,
            'source': os.path.basename(files[-1]), 'ids': d.get('ids')}


def bench_fnn(args, precision, snn):
    """The FNN L3 train step (BASELINE configs[1]) or, snn=True, the SNN fine-tune step (configs[4]); every rank calls it,
    rank 0 gets the result dict (the others None)."""
    global _DIST
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import _capi, synth
    from deep_ctr_amd import dl_utils as ut
    from deep_ctr_amd.engine import FNNEngine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = _DIST
    rehearse = bool(os.environ.get('FNN_BENCH_REHEARSE'))
    force_dp = bool(os.environ.get('FNN_BENCH_FORCE_DP'))      # exercise the DP code path with any world size
    if (world > 1 or force_dp) and dist is None:
        import torch.distributed as dist
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if 'MASTER_ADDR' not in os.environ:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if rehearse:              # control-flow rehearsal on ONE GPU: all ranks share cuda:0, gloo collectives
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        _DIST = dist
    if rehearse:
        local_rank = 0
    dev = torch.device('cuda', local_rank)
    B = args.batch
    if args.scaling == 'strong':
        if args.batch % (16 * world):
            raise SystemExit("--scaling strong: --batch %d does not split into %d shards of whole 16-example strips" % (args.batch, world))
        B = args.batch // world
    NB = 32                                               # distinct resident batches, cycled

    # ---- synthetic workload, seeded (table identical on every rank; ids differ per rank)
    sizes = synth.field_sizes_ipinyou()
    rows = synth.fm_table(sum(sizes), K, 0.05, 1234)
    fo = synth.field_of_row(sizes)
    ids_np = synth.zipf_ids(NB * B, sizes, 1.1, 1234 + 1000 * rank)
    y_np = (np.random.RandomState(99 + rank).uniform(size=NB * B) < 0.02).astype(np.float32)
    ut.seed_global(1234)
    p0 = ut.init_fnn_weights(XDIM, H1, H2, 'tanh')
    srng = ut.RandomStreams(234)
    srng.binomial(size=(1, XDIM), n=1, p=1)
    o1 = srng.binomial(size=(NB, H1), n=1, p=0.5)
    o2 = srng.binomial(size=(NB, H2), n=1, p=0.5)
    m1_np = o1.draw().astype(np.uint8)
    m2_np = o2.draw().astype(np.uint8)

    H0 = 200
    if snn:       # python/SNN_RBM.py:52-58: H0=200, H1=300, H2=100, lr=.001, dropout=.98, lambda1=0
        eng = FNNEngine(F, 0, H1, H2, max_batch=B, precision=precision, lr=0.001, lambda1=0.0,
                        lambda_fm=0.0, reg_all=True, device=local_rank, mode='bag', hidden0=H0)
        ww0 = np.random.default_rng(1234).standard_normal((sum(sizes), H0), dtype=np.float32) * np.float32(0.05)
        eng.set_table(ww0, fo, 0.0)
        eng.set_bag_bias(np.zeros(H0, np.float32))
        ut.seed_global(1234)
        w1s, _ = ut.init_weight(H0, H1, 'sigmoid'); w2s, _ = ut.init_weight(H1, H2, 'sigmoid')
        eng.set_dense({'w1': w1s, 'b1': np.zeros(H1), 'w2': w2s, 'b2': np.zeros(H2), 'w3': np.zeros(H2), 'b3': 0.0})
        del ww0
    else:
        eng = FNNEngine(F, K, H1, H2, max_batch=B, precision=precision, lr=0.001, lambda1=0.0,
                        lambda_fm=0.1, device=local_rank)
        eng.set_table(rows, fo, -3.0)
        eng.set_dense(p0)
    ids = torch.as_tensor(ids_np).to(dev).contiguous()
    y = torch.as_tensor(y_np).to(dev).contiguous()
    m1 = torch.as_tensor(m1_np).to(dev).contiguous()
    m2 = torch.as_tensor(m2_np).to(dev).contiguous()
    torch.cuda.synchronize(dev)
    lib, h = eng.lib, eng.h
    gB = B * world

    # ---- data parallelism: the library's native step (RCCL on its own stream; torch.distributed callbacks in the gloo rehearsal).
    # Should the native set-up fail on this node, the portable split step with torch.distributed's all-reduce still measures.
    collective, dp_error, split = None, None, False
    if dist is not None:
        from deep_ctr_amd.dp import DataParallelFNN
        try:
            # payload: with real peers the flat bucket is the default here -- a quarter of the slabs' bytes through the collective for
            # one more launch (2.5 us at world 1, profiles/r03_dp_forms_world1.json); no multi-GPU run has compared the two yet
            payload = args.dp_payload or os.environ.get('FNN_DP_PAYLOAD') or ('bucket' if world > 1 else None)
            dpw = DataParallelFNN(eng, sparse=args.dp_sparse, payload=payload, collective=args.dp_collective)
            collective = dpw.collective
        except Exception as e:
            dp_error = '%s: %s' % (type(e).__name__, e)
            ok = torch.tensor([0], dtype=torch.int32, device=dev if not rehearse else 'cpu')
        else:
            ok = torch.tensor([1], dtype=torch.int32, device=dev if not rehearse else 'cpu')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same path
        if int(ok.item()) == 0:
            if collective is not None:
                eng.dp_shutdown()
            split, collective = True, 'torch.distributed all_reduce of the flat bucket (portable split step)'
    bucket = eng.grad_bucket()

    def step(i):
        if not os.environ.get('FNN_BENCH_NOPREFETCH') and args.dp_sparse == 'local':
            nb = (i + 1) % NB                             # hand the NEXT batch's ids to the library early
            lib.fnn_prefetch_ids(h, ids.data_ptr() + nb * B * F * 4, B)
        b = i % NB
        a = (h, ids.data_ptr() + b * B * F * 4, y.data_ptr() + b * B * 4, B, m1.data_ptr() + b * H1,
             m2.data_ptr() + b * H2, gB)
        if not split:
            rc = lib.fnn_train_step(*a, None, None, _capi.FNN_MEM_DEVICE, None)       # N > 1: the collective is inside
        else:
            rc = lib.fnn_step_begin(*a, None, None, _capi.FNN_MEM_DEVICE)
            if rc == 0:
                dist.all_reduce(bucket)                   # ordered on the current stream = the engine's
                rc = lib.fnn_step_scatter(h)
                if rc == 0:
                    rc = lib.fnn_step_end(h, None)
        if rc != 0:
            raise RuntimeError(lib.fnn_last_error(h).decode())

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(eng.stream):
        for i in range(args.warmup):
            step(i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        t_enq = time.perf_counter() - t0                  # host time to enqueue the steps
        sync_all()
        dt = time.perf_counter() - t0
    eng.sync()
    last_loss = eng.last_loss() / B
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if not rehearse else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = gB * args.steps / dt

    # ---- per-kernel device time (HIP events on the library's own streams), same step loop
    roofline = None
    kern_ms = {}
    # every rank runs the profiled steps (under data parallelism a step holds a collective: rank 0 alone would wait
    # for ever); only rank 0 brackets its launches with events and reports
    eng.prof_enable(rank == 0)
    if rank == 0:
        eng.prof_reset()
    with torch.cuda.stream(eng.stream):
        for i in range(min(args.steps, 100)):
            step(i)
    sync_all()
    if rank == 0:
        for name in ('empty', 'step1', 'step2', 'allreduce', 'p2p_update', 'update', 'allgather', 'step3', 'step2_dense', 'step2_sparse', 'step3_dense', 'step3_sparse',
                     'sort_global', 'scatter_global', 'sort_now', 'mlp', 'gather', 'fwd1', 'fwd2', 'head', 'bwd1', 'gx', 'wgrad', 'reduce', 'update', 'sort',
                     'scatter', 'finalize'):
            kern_ms[name] = eng.prof_get(name)[0]
        eng.prof_enable(False)
        merged = dict(kern_ms)
        merged['scatter'] = kern_ms['scatter'] + kern_ms['finalize']
        cand = {k: v for k, v in merged.items() if k in ALGO and ALGO[k][1] > 0}
        dom = max(cand, key=cand.get)
        bound, per_ex = ALGO[dom]
        if snn:       # bag forward 64 + 16*800 B; update 2*12,800 + 800 B per example
            bound, per_ex = {'step1': ('hbm', 64 + 16 * 800), 'step2': ('hbm', 2 * 16 * 800 + 800)}.get(dom, (bound, per_ex))
        t_s = cand[dom] * 1e-3
        if bound == 'hbm':
            ach = per_ex * B / t_s / 1e9
            peak, unit = HBM_PEAK_GBS, 'GB/s'
        else:
            ach = per_ex * B / t_s / 1e12
            peak, unit = MFMA_PEAK_BY_KERNEL.get((precision, dom), MFMA_PEAK_TFLOPS[precision]), 'TFLOP/s'
        traffic = pmc_traffic(dom, snn) if precision == 'bf16' else None       # the committed PMC passes are of the bf16 mode
        roofline = {'kernel': dom, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit,
                    'frac': ach / peak, 'traffic': traffic,
                    # what the counters say reached HBM per launch, over the same launch time: beside the algorithmic fraction, never
                    # instead of it (hot rows served by L2 / Infinity Cache make the algorithmic figure exceed what HBM delivered)
                    'hbm_frac_from_counters': (traffic / t_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    'avg_launch_ms': cand[dom],
                    # an event-bracketed slot = kernel + the event mechanism (a back-to-back pair alone: 'event_pair_ms');
                    # the kernel-only average of the committed rocprofv3 summary of this command is quoted beside it
                    'event_pair_ms': kern_ms.get('empty'),
                    # a CONSTANT read from the committed rocprofv3 summary of this command (profiles/), not measured in this run
                    'committed_rocprof_avg_ms': None if snn else rocprof_avg_ms(dom, precision),
                    'algorithmic_per_example': per_ex,
                    'step': {'achieved': (39264 if snn else STEP_MIN_BYTES) * B / (ms_per_step * 1e-3) / 1e9,
                             'frac': (39264 if snn else STEP_MIN_BYTES) * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'unit': 'GB/s'}}

    # ---- N > 1: replicas of the exact mode must hold identical tables (a checksum of each rank's table, min == max over ranks)
    dp_check = None
    if dist is not None and world > 1 and args.dp_sparse == 'exchange' and not split:
        try:
            # the same rows on every rank: every 229th row of the table (the small fields' hot rows among them) -- all ranks'
            # shards have updated them, in global example order if the exchange works
            probe = np.arange(0, rows.shape[0], 229, dtype=np.int64)[:4096]
            got = torch.as_tensor(eng.get_rows(probe)).double().sum().reshape(1).to(dev if not rehearse else 'cpu')
            lo, hi = got.clone(), got.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dp_check = {'rows_probed': int(len(probe)), 'checksum_min': float(lo.item()), 'checksum_max': float(hi.item()),
                        'replicas_identical': bool(lo.item() == hi.item())}
        except Exception as e:
            dp_check = {'error': '%s: %s' % (type(e).__name__, e)}

    # ---- CPU baseline: the C port of the oracle on this host, 1 core, bounded sample
    cpu = cpu_vec = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not snn:
        cpu = cpu_baseline(rows, ids_np, y_np, m1_np, m2_np, p0, B, args.cpu_seconds)
        cpu_vec = cpu_baseline_vectorised(rows, ids_np, y_np, m1_np, m2_np, p0, B, args.cpu_seconds)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and snn:
        # the float64 restatement of python/SNN_RBM.py:238-291 (oracle.fnn_oracle.snn_train_step: NumPy bag + MLP, the
        # reference's per-example Python loop for the row updates) on the same table / ids, bounded sample
        from oracle import fnn_oracle as orc
        ww64 = (np.random.default_rng(1234).standard_normal((sum(sizes), H0), dtype=np.float32) * np.float32(0.05)).astype(np.float64)
        bb64 = np.zeros(H0)
        ut.seed_global(1234)
        w1s, _ = ut.init_weight(H0, H1, 'sigmoid'); w2s, _ = ut.init_weight(H1, H2, 'sigmoid')
        pc = {'w1': np.array(w1s, np.float64), 'b1': np.zeros(H1), 'w2': np.array(w2s, np.float64), 'b2': np.zeros(H2), 'w3': np.zeros(H2), 'b3': 0.0}
        n, t0c = 0, time.perf_counter()
        while True:
            b = n % NB
            sl = slice(b * B, (b + 1) * B)
            orc.snn_train_step(pc, ww64, bb64, ids_np[sl], y_np[sl].astype(np.float64), m1_np[b].astype(np.float64), m2_np[b].astype(np.float64),
                               0.001, 0.0)
            n += 1
            el = time.perf_counter() - t0c
            if el >= args.cpu_seconds or n >= 200:
                break
        cpu = {'value': n * B / el, 'unit': 'examples/sec', 'cores': 1, 'kind': 'port',
               'sample': '%d steps of batch %d on the same table/ids (oracle.fnn_oracle.snn_train_step: NumPy float64, per-example row '
                         'updates in the interpreter as in the reference; host has %d cores)' % (n, B, os.cpu_count())}
        del ww64

    out = None
    if rank == 0:
        out = {
            'metric': 'examples/sec', 'value': value, 'unit': 'examples/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': precision, 'data': 'synthetic',
            'config': {'workload': ('SNN fine-tune step: 16 fields, 937670 x 200 bag table, hidden 300/100 tanh, batch '
                                    '%d per GPU, Zipf(1.1) ids' % B if snn else
                                    'FNN L3 train step: 16 fields, 937670 one-hot dims, k=10, hidden 300/100 '
                                    'tanh, batch %d per GPU, Zipf(1.1) ids' % B),
                       'per_gpu_batch': B, 'global_batch': gB,
                       'parallelism': 'dp%d' % world if world > 1 else 'single'},
            'train_logloss_last_step': last_loss,
            'meets_parity_bar_1e-4': MEETS_PARITY_BAR[precision],
            'host_enqueue_ms_per_step': t_enq / args.steps * 1e3,
            'roofline': roofline, 'cpu_baseline': cpu, 'cpu_baseline_vectorised': cpu_vec, 'kernel_ms': kern_ms,
        }
        if dist is not None:
            cfg = eng.dp_config() if not split else {'payload': 'bucket', 'collective': 'torch.distributed', 'region': 'none'}
            p2p = cfg['collective'] == 'p2p'
            ev = kern_ms.get('empty') or 0.0
            out['data_parallel'] = {'collective': collective, 'sparse_rows': args.dp_sparse,
                                    'payload': cfg['payload'], 'exchange_region': cfg['region'],
                                    'p2p_max_flag_wait_us': eng.dp_p2p_max_wait_us() if p2p else None,     # longest wait for a peer's flag in any step (rank skew)
                                    # event-bracketed slot of the collective on rank 0, the event pair's own cost taken off; for p2p the
                                    # slot is the update launch that performs the all-reduce (wait for the peers + sum + update)
                                    'collective_us': max(0.0, ((kern_ms.get('p2p_update') if p2p else kern_ms.get('allreduce')) or 0.0) - ev) * 1e3,
                                    'collective_slot': 'p2p_update (all-reduce + dense update in one launch)' if p2p else 'allreduce',
                                    'launches_per_step': ('five + all-reduce of the flat bucket' if split else
                                                          'three + the p2p update launch (sums every rank\'s bucket over peer pointers)' if p2p else
                                                          'three + all-reduce of the flat bucket + update launch' if cfg['payload'] == 'bucket' else
                                                          'three + one all-reduce of the split-K weight-gradient slabs'),
                                    'native_setup_error': dp_error, 'exact_mode_check': dp_check,
                                    'rehearsal_all_ranks_on_one_gpu': rehearse}
    eng.close()
    del ids, y, m1, m2
    torch.cuda.empty_cache()
    return out


def bench_e2e(args):
    """One epoch of the FNN script end to end, from TEXT: what the reference does per epoch around its Theano call
    (python/FNN_wnzh.py:224-253 linecache + get_fxy per line, :193-221 the evaluation pass re-parsing the whole file) against
    the native path: fm.model.txt (937,670 features) and train.fm.txt parsed once by ctr_ingest.h, ids resident in HBM,
    4096-example train steps with the next batch's grouping riding on each, fnn_eval on the device.  Files are synthetic
    (written to a temporary directory, config shape, Zipf ids); the value is examples/s over parse + copy + train + eval."""
    import shutil
    import tempfile
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import ingest, synth
    from deep_ctr_amd import dl_utils as ut
    from deep_ctr_amd.data_fm import DataFM
    from deep_ctr_amd.engine import FNNEngine
    N, B = args.e2e_lines, args.batch
    sizes = synth.field_sizes_ipinyou()
    D = sum(sizes)
    rows = synth.fm_table(D, K, 0.05, 1234)
    fo = synth.field_of_row(sizes)
    ids_np = synth.zipf_ids(N, sizes, 1.1, 4321)
    y_np = (np.random.RandomState(5).uniform(size=N) < 0.02).astype(np.int32)
    tmp = tempfile.mkdtemp(prefix='fnn_e2e_', dir=os.environ.get('TMPDIR', '/tmp'))
    try:
        t0 = time.perf_counter()
        mpath, tpath = os.path.join(tmp, 'fm.model.txt'), os.path.join(tmp, 'train.fm.txt')
        names = np.array(synth.FIELD_NAMES)[fo]
        with open(mpath, 'w') as f:                               # feature id = row index here
            f.write('%r %d %d\n' % (-3.0, D, K - 1))
            for lo in range(0, D, 65536):
                hi = min(D, lo + 65536)
                body = np.char.mod('%.7g', rows[lo:hi])
                f.write('\n'.join('%d %s %s:%d' % (i, ' '.join(body[i - lo]), names[i], i) for i in range(lo, hi)) + '\n')
        np.savetxt(tpath, np.column_stack([y_np, ids_np]), fmt='%d ' + ' '.join(['%d:1'] * F))
        t_gen = time.perf_counter() - t0
        sz_m, sz_t = os.path.getsize(mpath), os.path.getsize(tpath)
        threads = ingest.n_threads()
        # ---- A1: the model
        t0 = time.perf_counter()
        data = DataFM(mpath)
        t_model = time.perf_counter() - t0
        # ---- A2: the examples
        t0 = time.perf_counter()
        ids, yy, sh = data.load_ids(tpath, want_shadowed=True)
        t_parse = time.perf_counter() - t0
        assert ids.shape == (N, F) and len(sh) == 0
        # ---- the binary id cache: written once (not counted), what a later run reads instead of parsing
        cdir = os.path.join(tmp, 'cache')
        data.load_ids(tpath, want_shadowed=True, cache_dir=cdir)
        t0 = time.perf_counter()
        ids_c, yy_c, _ = data.load_ids(tpath, want_shadowed=True, cache_dir=cdir)
        t_cache = time.perf_counter() - t0
        assert np.array_equal(ids_c, ids) and np.array_equal(yy_c, yy)
        del ids_c, yy_c
        # ---- engine + copies
        t0 = time.perf_counter()
        eng = FNNEngine(F, data.k, H1, H2, max_batch=B, precision=args.precision, lr=0.001, lambda1=0.0, lambda_fm=0.1)
        r32, fo2, w0 = data.table()
        eng.set_table(r32, fo2, w0)
        ut.seed_global(1234)
        eng.set_dense(ut.init_fnn_weights(XDIM, H1, H2, 'tanh'))
        ids_d, y_d = eng.to_device(ids, yy)
        yf_d = y_d.float()
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        m1 = torch.ones(H1, dtype=torch.uint8, device=ids_d.device)
        m2 = torch.ones(H2, dtype=torch.uint8, device=ids_d.device)
        nb = N // B
        lib, h = eng.lib, eng.h
        from deep_ctr_amd import _capi

        def epoch():
            with torch.cuda.stream(eng.stream):
                for j in range(nb):
                    if j + 1 < nb:
                        lib.fnn_prefetch_ids(h, ids_d.data_ptr() + (j + 1) * B * F * 4, B)
                    rc = lib.fnn_train_step(h, ids_d.data_ptr() + j * B * F * 4, yf_d.data_ptr() + j * B * 4, B, m1.data_ptr(), m2.data_ptr(), B,
                                            None, None, _capi.FNN_MEM_DEVICE, None)
                    if rc != 0:
                        raise RuntimeError(lib.fnn_last_error(h).decode())
            eng.sync()
        epoch()                                                   # warm-up epoch (clocks, caches)
        t0 = time.perf_counter()
        epoch()
        t_train = time.perf_counter() - t0
        y_np2 = y_np.copy(); y_np2[:16] = 1                       # both classes present whatever the draw
        y_d2 = torch.as_tensor(y_np2).to(ids_d.device)
        eng.evaluate(ids_d[:B], y_d2[:B])
        t0 = time.perf_counter()
        ev = eng.evaluate(ids_d, y_d2)
        t_eval = time.perf_counter() - t0
        eng.close()
        # ---- the reference's way, on a bounded sample: per-line Python parsing (oracle/ingest_oracle.fnn_examples = get_fxy per line)
        cpu = None
        if not args.no_cpu_baseline:
            from oracle import ingest_oracle as ino
            n_s = min(N, 50000)
            spath = os.path.join(tmp, 'sample.fm.txt')
            with open(tpath) as fi, open(spath, 'w') as fo_:
                for _ in range(n_s):
                    fo_.write(fi.readline())
            ff = dict(zip(range(D), fo.tolist()))
            fr = {i: i for i in range(D)}
            t0 = time.perf_counter()
            ino.fnn_examples(spath, ff, fr, F)
            t_py = time.perf_counter() - t0
            cpu = {'value': n_s / t_py, 'unit': 'lines/sec', 'cores': 1, 'kind': 'port',
                   'sample': 'parse of %d lines by oracle.ingest_oracle.fnn_examples (the reference parses every line again for every epoch and every '
                             'evaluation pass, python/FNN_wnzh.py:224-253,193-221); host has %d cores' % (n_s, os.cpu_count())}
        t_all = t_parse + t_setup + t_train + t_eval
        out = {
            'metric': 'examples/sec', 'value': N / (t_parse + t_train + t_eval), 'unit': 'examples/sec', 'n_gpus': 1, 'steps': nb, 'warmup': nb,
            'ms_per_step': t_train / nb * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'one epoch of the FNN script from text: parse train.fm.txt (%d lines, %.0f MB) + %d train steps of %d + '
                                   'evaluation of all lines; model file 937670 features (%.0f MB) parsed once' % (N, sz_t / 1e6, nb, B, sz_m / 1e6)},
            'phases_s': {'write synthetic files (not counted)': t_gen, 'parse fm.model.txt': t_model, 'parse train.fm.txt': t_parse,
                         'read the binary id cache instead (later runs)': t_cache,
                         'engine set-up + copies to HBM (once per run, not counted)': t_setup, 'train epoch': t_train, 'evaluation pass': t_eval},
            'ingest': {'threads': threads, 'model_MB_per_s': sz_m / 1e6 / t_model, 'model_rows_per_s': D / t_model,
                       'examples_MB_per_s': sz_t / 1e6 / t_parse, 'examples_lines_per_s': N / t_parse, 'cache_lines_per_s': N / t_cache},
            'value_from_id_cache': N / (t_cache + t_train + t_eval),
            'train_examples_per_s': nb * B / t_train, 'eval_examples_per_s': N / t_eval, 'eval': {k: ev[k] for k in ('auc', 'rmse', 'logloss')},
            'roofline': None, 'cpu_baseline': cpu,
            'note': 'value = lines / (parse + train epoch + evaluation): what one epoch of the script costs end to end once the files are parsed '
                    'natively and the ids stay in HBM; the reference re-parses per epoch at the cpu_baseline rate',
        }
        del t_all
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_rbm(args):
    """SNN pre-training (BASELINE configs[4]; python/sampling_based_gaussian_binary_rbm_sparse.py) at the
    config shape: the sparse first layer 937,670 x 200 (750 MB), 32 sampled visibles per example, in the
    reference's EXACT online mode (one example at a time, sequential by definition: one workgroup), and
    the dense 200 -> 300 CD-1 layer on mini-batches of 4096 (MFMA kernels).  A 'step' = 4096 examples."""
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import _capi, synth
    lib = _capi.load()
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    N, H0, H1, S = 4096, 200, 300, 32
    sizes = synth.field_sizes_ipinyou()
    D = sum(sizes)
    rng = np.random.default_rng(3)
    ids = np.sort(2 * synth.zipf_ids(N, sizes, 1.1, 5).astype(np.int64) % (D - 2) + 1, axis=1)     # 16 odd ids: id - 1 never collides
    vid = np.empty((N, S), np.int32); vid[:, 0::2] = ids - 1; vid[:, 1::2] = ids
    vid.sort(axis=1)
    vval = np.isin(vid, ids).astype(np.uint8) if False else ((vid % 2) == 1).astype(np.uint8)
    W = torch.as_tensor(rng.uniform(-0.1, 0.1, (D, H0)).astype(np.float32)).to(dev)
    vb = torch.zeros(D, dtype=torch.float32, device=dev); hb = torch.zeros(H0, dtype=torch.float32, device=dev)
    ws = torch.zeros((S, H0), dtype=torch.float32, device=dev)
    vid_d, vval_d = torch.as_tensor(vid).to(dev), torch.as_tensor(vval).to(dev)
    unif = torch.rand((N, H0), device=dev)
    err = C.c_double()

    def sparse_pass():
        rc = lib.rbm_sparse_epoch(W.data_ptr(), vb.data_ptr(), hb.data_ptr(), ws.data_ptr(), vid_d.data_ptr(), vval_d.data_ptr(),
                                  unif.data_ptr(), N, H0, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9, C.byref(err), st)
        if rc != 0:
            raise RuntimeError(lib.rbm_last_error().decode())
    for _ in range(max(1, args.warmup // 10)):
        sparse_pass()
    torch.cuda.synchronize(dev)
    n_sp = max(2, args.steps // 20)
    t0 = time.perf_counter()
    for _ in range(n_sp):
        sparse_pass()
    torch.cuda.synchronize(dev)
    dt_sp = (time.perf_counter() - t0) / n_sp
    # mini-batch mode of the sparse layer (rbm_sparse_batch; not the reference's schedule): 16 mini-batches of 4096
    NBm = 16
    ids_b = np.sort(2 * synth.zipf_ids(NBm * N, sizes, 1.1, 6).astype(np.int64) % (D - 2) + 1, axis=1)
    vid_b = np.empty((NBm * N, S), np.int32); vid_b[:, 0::2] = ids_b - 1; vid_b[:, 1::2] = ids_b
    vid_b.sort(axis=1)
    vval_b = ((vid_b % 2) == 1).astype(np.uint8)
    vidb_d, vvalb_d = torch.as_tensor(vid_b).to(dev), torch.as_tensor(vval_b).to(dev)
    unif_b = torch.rand((NBm * N, H0), device=dev)
    dW = torch.zeros_like(W); dvis = torch.zeros_like(vb)

    def batch_pass():
        rc = lib.rbm_sparse_batch(W.data_ptr(), dW.data_ptr(), vb.data_ptr(), dvis.data_ptr(), hb.data_ptr(), ws.data_ptr(),
                                  vidb_d.data_ptr(), vvalb_d.data_ptr(), unif_b.data_ptr(), NBm * N, N, H0, S, 2e-4, 1e-4, 1e-4, 1e-4, 0.9,
                                  C.byref(err), st)
        if rc != 0:
            raise RuntimeError(lib.rbm_last_error().decode())
    batch_pass()
    torch.cuda.synchronize(dev)
    n_b = max(2, args.steps // 20)
    t0 = time.perf_counter()
    for _ in range(n_b):
        batch_pass()
    torch.cuda.synchronize(dev)
    dt_b = (time.perf_counter() - t0) / n_b / NBm               # per mini-batch of 4096
    per_ex_b = S * H0 * 4 * 5 + S * 8 + H0 * 4                 # rows read, dW RMW twice (add, grab), W RMW
    del dW, dvis
    # dense CD-1 layer 200 -> 300
    h = C.c_void_p()
    if lib.rbm_dense_create(H0, H1, N, 1 if args.precision == 'bf16' else 0, 0, st, C.byref(h)) != 0:
        raise RuntimeError(lib.rbm_last_error().decode())
    Wd = rng.uniform(-0.1, 0.1, (H0, H1)).astype(np.float32); v0 = np.zeros(H0, np.float32); h0 = np.zeros(H1, np.float32)
    lib.rbm_dense_set(h, Wd.ctypes.data, v0.ctypes.data, h0.ctypes.data)
    X = torch.rand((N, H0), device=dev); U = torch.rand((N, H1), device=dev)
    for _ in range(args.warmup):
        lib.rbm_dense_cd1(h, X.data_ptr(), N, U.data_ptr(), 2e-4, 1e-4, 1e-4, 1e-4, 0.9, None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lib.rbm_dense_cd1(h, X.data_ptr(), N, U.data_ptr(), 2e-4, 1e-4, 1e-4, 1e-4, 0.9, None)
    torch.cuda.synchronize(dev)
    dt_de = (time.perf_counter() - t0) / args.steps
    lib.rbm_dense_destroy(h)
    per_ex = S * H0 * 4 * 2 + S * 8 + H0 * 4                   # 32 rows read + written, ids/values, uniforms
    del W, vb, unif, unif_b, vidb_d, vvalb_d, X, U
    torch.cuda.empty_cache()
    return ({
        'metric': 'examples/sec', 'value': N / dt_sp, 'unit': 'examples/sec', 'n_gpus': 1, 'steps': n_sp, 'warmup': args.warmup,
        'ms_per_step': dt_sp * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'SNN pre-training, sparse RBM layer 937670 x 200, 32 sampled visibles per example, EXACT online CD-1 '
                               '(batch 1, sequential: one workgroup); a step = %d examples' % N},
        'roofline': {'kernel': 'k_rbm_sparse', 'bound': 'hbm', 'achieved': per_ex * N / dt_sp / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': per_ex * N / dt_sp / 1e9 / HBM_PEAK_GBS, 'traffic': None, 'algorithmic_per_example': per_ex,
                     'note': 'latency-bound by construction: example n reads the rows example n-1 wrote'},
        'sparse_minibatch_4096': {'examples_per_sec': N / dt_b, 'ms_per_minibatch': dt_b * 1e3, 'hbm_GBs': per_ex_b * N / dt_b / 1e9,
                                  'frac_of_hbm_peak': per_ex_b * N / dt_b / 1e9 / HBM_PEAK_GBS, 'algorithmic_per_example': per_ex_b,
                                  'note': 'rbm_sparse_batch: every example of a mini-batch reads start-of-batch parameters (NOT the reference schedule)'},
        'dense_cd1_200x300': {'examples_per_sec': N / dt_de, 'ms_per_minibatch_of_4096': dt_de * 1e3, 'dtype': args.precision},
        'cpu_baseline': None})


def bench_gather(args):
    """The embedding gathers alone, 16,384 examples per launch (fnn_gather, reference-shaped output):
    A3 -- 16 FM rows of 44 B -> x [177] f32 (python/FNN_wnzh.py:87-96): 64 + 704 + 708 = 1,476 algorithmic
    bytes per example, from a 60 MB table (Infinity-Cache resident);
    A8 -- sigmoid(sum of 16 bag rows of 800 B + bias) -> x [200] f32 (python/SNN_RBM.py:238-262):
    64 + 12,800 + 800 = 13,664 B per example from a 750 MB table: the HBM-bound gather."""
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import _capi, synth
    from deep_ctr_amd.engine import FNNEngine
    B, NB, H0 = 16384, 8, 200
    dev = torch.device('cuda', 0)
    sizes = synth.field_sizes_ipinyou()
    fo = synth.field_of_row(sizes)
    ids_by = {'zipf': synth.zipf_ids(NB * B, sizes, 1.1, 1234),            # the step benchmark's ids: hot rows hit L2 / Infinity Cache
              # every slot draws from the WHOLE table (the bag gather ignores fields): 262k distinct 800-byte rows per launch
              'uniform': np.random.default_rng(7).integers(0, sum(sizes), (NB * B, F), dtype=np.int64).astype(np.int32)}
    res = {}
    # '*_100k': one launch of 100,000 examples, the chunk the reference's evaluation pass gathers at a time (python/FNN_wnzh.py:193-209);
    # at 16 x 100,000 draws from 937,670 rows a bag row is read about twice per launch
    only = os.environ.get('FNN_GATHER_ONLY')                # one variant per process: the PMC passes of tools/pmc_gather.sh (same kernel names)
    for name, per_ex, dist in (('fm', 64 + 704 + 708, 'zipf'), ('fm_uniform', 64 + 704 + 708, 'uniform'), ('bag', 64 + 16 * 800 + 800, 'uniform'),
                               ('bag_zipf', 64 + 16 * 800 + 800, 'zipf'), ('fm_100k', 64 + 704 + 708, 'zipf'), ('bag_100k', 64 + 16 * 800 + 800, 'uniform')):
        if only and name != only:
            continue
        ids = torch.as_tensor(ids_by[dist]).to(dev).contiguous()
        if name.startswith('fm'):
            eng = FNNEngine(F, K, H1, H2, max_batch=B, precision='bf16', device=0)
            eng.set_table(synth.fm_table(sum(sizes), K, 0.05, 1234), fo, -3.0)
            xdim = XDIM
        else:
            eng = FNNEngine(F, 0, H1, H2, max_batch=B, precision='bf16', device=0, mode='bag', hidden0=H0)
            eng.set_table(np.random.default_rng(1).standard_normal((sum(sizes), H0), dtype=np.float32) * np.float32(0.05), fo, 0.0)
            eng.set_bag_bias(np.zeros(H0, np.float32))
            xdim = H0
        Bk = 100000 if name.endswith('_100k') else B
        x = torch.empty((Bk, xdim), dtype=torch.float32, device=dev)
        lib, h = eng.lib, eng.h

        def run(n):
            for i in range(n):
                off = (i * Bk) % (NB * B - Bk + 1)
                rc = lib.fnn_gather(h, ids.data_ptr() + off * F * 4, Bk, x.data_ptr(), _capi.FNN_MEM_DEVICE)
                if rc != 0:
                    raise RuntimeError(lib.fnn_last_error(h).decode())
        with torch.cuda.stream(eng.stream):
            run(args.warmup)
            eng.sync()
            eng.prof_enable(True); eng.prof_reset()
            t0 = time.perf_counter()
            run(args.steps)
            eng.sync()
            dt = time.perf_counter() - t0
        ms = eng.prof_get('gather_ref')[0]
        eng.prof_enable(False)
        ach = per_ex * Bk / (ms * 1e-3) / 1e9
        traffic = pmc_traffic_gather(name)
        res[name] = {'ids': dist, 'examples_per_launch': Bk, 'avg_launch_ms': ms, 'wall_ms_per_launch': dt / args.steps * 1e3,
                     'algorithmic_per_example': per_ex, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                     # HBM bytes per launch from the committed counter passes of this variant (FETCH doubled + WRITE), and the fraction
                     # of the HBM peak THEY amount to over this run's launch time: what the memory system delivered, beside the algorithmic
                     # figure above (which cache-served rows push past it)
                     'traffic': traffic, 'hbm_frac_from_counters': (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     'examples_per_sec': Bk / (ms * 1e-3)}
        eng.close()
        del ids, x
        torch.cuda.empty_cache()
    if only:
        return {'metric': 'examples/sec', 'value': res[only]['examples_per_sec'], 'unit': 'examples/sec', 'n_gpus': 1, 'variant': only, only: res[only]}
    return ({
        'metric': 'examples/sec', 'value': res['fm']['examples_per_sec'], 'unit': 'examples/sec', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': res['fm']['avg_launch_ms'], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'standalone embedding gathers: FM rows (16 x 44 B -> x[177]) and SNN bag (16 x 800 B -> x[200]), '
                               '937670 rows, Zipf(1.1) ids'},
        'roofline': dict(res['bag'], kernel='k_bag_ref (A8 gather, uniform ids: every row read misses the caches)', bound='hbm'),
        'bag_gather_zipf': res['bag_zipf'], 'fm_gather': res['fm'], 'fm_gather_uniform': res['fm_uniform'], 'fm_gather_100k': res['fm_100k'], 'bag_gather_100k': res['bag_100k'],
        'cpu_baseline': None})


def bench_pretrain(args):
    """The two pre-trainers of SURVEY 8(f) that had parity tests but no measurement: FM pre-training (row N3: python/FM.py:55-64, a step
    of batch 4096 at the config shape -- 937,670 features, 16 fields, rank 10, plain SGD with the dense L2 term) and the SNN-DAE online
    trainers (row N2: sampling_based_denosing_autoencoder.py -- batch 1, sequential by definition, one workgroup each)."""
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import _capi, synth
    lib = _capi.load()
    dev = torch.device('cuda', 0)
    stream = torch.cuda.Stream(device=dev)
    st = C.c_void_p(stream.cuda_stream)
    B, NB = args.batch, 16
    sizes = synth.field_sizes_ipinyou()
    D = sum(sizes)
    # ---------------------------------------------------------------- FM pre-training
    h = C.c_void_p()
    if lib.fm_create(F, K, B, 0, st, C.byref(h)) != 0:
        raise RuntimeError((lib.fm_last_error(None) or b'').decode())
    rows = synth.fm_table(D, K, 0.01, 77)
    if lib.fm_set_table(h, rows.ctypes.data, D) != 0 or lib.fm_set_b(h, 0.0) != 0:
        raise RuntimeError(lib.fm_last_error(h).decode())
    ids = torch.as_tensor(synth.zipf_ids(NB * B, sizes, 1.1, 99)).to(dev).contiguous()
    y = torch.as_tensor((np.random.RandomState(3).uniform(size=NB * B) < 0.02).astype(np.float32)).to(dev)

    def fm_steps(n):
        for i in range(n):
            j = i % NB
            rc = lib.fm_train_step(h, ids.data_ptr() + j * B * F * 4, y.data_ptr() + j * B * 4, B, 1e-4, 1e-6, 1, None, None)
            if rc != 0:
                raise RuntimeError(lib.fm_last_error(h).decode())
    with torch.cuda.stream(stream):
        fm_steps(args.warmup)
        lib.fm_sync(h)
        t0 = time.perf_counter()
        fm_steps(args.steps)
        lib.fm_sync(h)
        dt_fm = (time.perf_counter() - t0) / args.steps
    lib.fm_destroy(h)
    fm_bytes = 64 + 704 + 704 + 4                                     # ids, rows read, rows written, label
    # ---------------------------------------------------------------- SNN-DAE online trainers
    N, H0, H1, S = 20000, 200, 300, 32
    rng = np.random.default_rng(5)
    out_dae = {}
    for prec, npdt, tdt, sfx, lrv in (('f32', np.float32, torch.float32, '', C.c_float(0.1)), ('f64', np.float64, torch.float64, '_f64', C.c_double(0.1))):
        table = torch.as_tensor(rng.uniform(-0.05, 0.05, (D, H0)).astype(npdt)).to(dev)
        idx_np = np.sort(synth.zipf_ids(N, sizes, 1.1, 8).astype(np.int64), axis=1)
        idx2 = np.empty((N, S), np.int32); idx2[:, 0::2] = (idx_np + 1) % D; idx2[:, 1::2] = idx_np      # negative, positive, ...
        xv = np.zeros((N, S), npdt); xv[:, 1::2] = 1
        idx_d, x_d = torch.as_tensor(idx2).to(dev), torch.as_tensor(xv).to(dev)
        bh = torch.zeros(H0, dtype=tdt, device=dev); bv = torch.zeros(S, dtype=tdt, device=dev); bp = torch.zeros(H0, dtype=tdt, device=dev)
        cost = C.c_double()
        sp, de = getattr(lib, 'dae_sparse_epoch' + sfx), getattr(lib, 'dae_dense_epoch' + sfx)
        cs = torch.cuda.current_stream(dev).cuda_stream

        def run_sparse():
            if sp(table.data_ptr(), D, bh.data_ptr(), bv.data_ptr(), bp.data_ptr(), idx_d.data_ptr(), x_d.data_ptr(), N, H0, S, lrv, C.byref(cost), cs) != 0:
                raise RuntimeError(lib.dae_last_error().decode())
        run_sparse(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); run_sparse(); torch.cuda.synchronize(dev)
        dt_s = time.perf_counter() - t0
        X = torch.as_tensor(rng.uniform(0, 1, (N, H0)).astype(npdt)).to(dev)
        W = torch.as_tensor(rng.uniform(-0.1, 0.1, (H0, H1)).astype(npdt)).to(dev)
        bh2 = torch.zeros(H1, dtype=tdt, device=dev); bv2 = torch.zeros(H0, dtype=tdt, device=dev)

        def run_dense():
            if de(W.data_ptr(), bh2.data_ptr(), bv2.data_ptr(), X.data_ptr(), N, H0, H1, lrv, 0, C.byref(cost), cs) != 0:
                raise RuntimeError(lib.dae_last_error().decode())
        run_dense(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); run_dense(); torch.cuda.synchronize(dev)
        dt_d = time.perf_counter() - t0
        out_dae[prec] = {'sparse_layer_examples_per_sec': N / dt_s, 'sparse_layer_us_per_example': dt_s / N * 1e6,
                         'dense_200x300_examples_per_sec': N / dt_d, 'dense_200x300_us_per_example': dt_d / N * 1e6}
        del table, X, W
        torch.cuda.empty_cache()
    return {
        'metric': 'examples/sec', 'value': B / dt_fm, 'unit': 'examples/sec', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt_fm * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'FM pre-training step (python/FM.py): 937670 features, 16 fields, rank 10, batch %d, SGD + dense L2, mean loss, '
                               'Zipf(1.1) ids; beside it the SNN-DAE online trainers (batch 1; %d examples per epoch)' % (B, N)},
        'roofline': {'kernel': 'fm_train_step (four launches: forward/loss, grouping, two-level sparse-row update)', 'bound': 'hbm',
                     'achieved': fm_bytes * B / dt_fm / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': fm_bytes * B / dt_fm / 1e9 / HBM_PEAK_GBS,
                     'traffic': None, 'algorithmic_per_example': fm_bytes, 'note': 'wall time per step (latency-bound like the FNN step)'},
        'dae_online': dict(out_dae, note='sequential by definition (example n + 1 starts from the parameters example n left): one persistent '
                                         'workgroup; f64 is what get_da_weights runs by default (the reference\'s floatX)'),
        'cpu_baseline': None}


IP_HIDDEN = [1000, 800, 600, 400, 200, 100, 50]        # python/baseline.py:139 (FNN_IP_L7)


def bench_ipnn(args):
    """BASELINE configs[2]: FNN_IP_L7 train step (python/FNN_IP_L7.py:102-133 forward + loss + SGD),
    16 fields, 937,670 rows of K=11, z1 = 297, hidden 1000/800/600/400/200/100/50 relu, keep_prob 0.5
    (per-element keep-masks are inputs, resident in HBM), batch 4096, bf16 MFMA / f32 accumulate.
    Single GPU (replicas under --gpus N are not launched: the config names 1 x MI355X)."""
    import torch
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd import synth
    from deep_ctr_amd.ipnn import IPNNEngine
    B, NB, NM = args.batch, 16, 4
    dev = torch.device('cuda', 0)
    sizes = synth.field_sizes_ipinyou()
    rows = synth.fm_table(sum(sizes), K, 0.05, 1234)
    ids_np = synth.zipf_ids(NB * B, sizes, 1.1, 1234)
    y_np = (np.random.RandomState(99).uniform(size=NB * B) < 0.02).astype(np.float32)
    d = [F * K + F * (F - 1) // 2 + 1] + IP_HIDDEN + [1]
    eng = IPNNEngine(F, K, IP_HIDDEN, 'relu', max_batch=B, precision=args.precision, lr=1e-4, keep_prob=0.5, optimizer=args.optimizer)
    rs = np.random.RandomState(1234)
    # uniform(-.01, .01) as python/baseline.py:140 would leave relu activations ~0 after 7 layers;
    # Glorot-scale weights keep every layer's arithmetic live (timing does not depend on the values)
    Ws = [rs.uniform(-1, 1, (d[i], d[i + 1])).astype(np.float32) * np.float32(np.sqrt(6.0 / (d[i] + d[i + 1]))) for i in range(len(d) - 1)]
    eng.set_params(rows, 0.0, Ws, [np.zeros(d[i + 1], np.float32) for i in range(len(d) - 1)])
    ids = torch.as_tensor(ids_np).to(dev).contiguous()
    y = torch.as_tensor(y_np).to(dev).contiguous()
    g = torch.Generator(device=dev); g.manual_seed(234)
    masks = [[(torch.rand((B, d[t]), device=dev, generator=g) < 0.5).to(torch.uint8).contiguous() for t in range(len(IP_HIDDEN) + 1)]
             for _ in range(NM)]
    marr = [(C.c_void_p * len(m))(*[x.data_ptr() for x in m]) for m in masks]
    torch.cuda.synchronize(dev)
    lib, h = eng.lib, eng.h

    def step(i):
        b = i % NB
        rc = lib.ipnn_train_step(h, ids.data_ptr() + b * B * F * 4, y.data_ptr() + b * B * 4, B, marr[i % NM], None, None)
        if rc != 0:
            raise RuntimeError(lib.ipnn_last_error(h).decode())

    for i in range(args.warmup):
        step(i)
    eng.sync(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    t_enq = time.perf_counter() - t0
    eng.sync(); torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    loss = C.c_float()
    lib.ipnn_train_step(h, ids.data_ptr(), y.data_ptr(), B, marr[0], None, C.byref(loss))
    # per-segment device time (HIP events on the library's stream)
    seg = {}
    if hasattr(lib, 'ipnn_prof_enable'):
        lib.ipnn_prof_enable(h, 1)
        for i in range(min(args.steps, 50)):
            step(i)
        eng.sync()
        for name in ('sort', 'mask_t', 'ip_fwd', 'fwd', 'bwd', 'wgrad', 'ip_bwd', 'scatter', 'adam_table', 'update'):
            ms = C.c_double()
            lib.ipnn_prof_get(h, name.encode(), C.byref(ms))
            seg[name] = ms.value
        lib.ipnn_prof_enable(h, 0)
    prod = sum(d[i] * d[i + 1] for i in range(len(d) - 1))
    fl = {'fwd': 2 * prod, 'bwd': 2 * (prod - d[0] * d[1]) + 2 * d[0] * d[1], 'wgrad': 2 * prod}
    flops_ex = 6 * prod + 3 * 2 * (F * (F - 1) // 2) * K          # + inner products fwd and their two-sided bwd
    ms_per_step = dt / args.steps * 1e3
    peak = MFMA_PEAK_TFLOPS[args.precision]
    roof = None
    if seg.get('fwd', 0) > 0 and seg.get('bwd', 0) > 0:
        # the narrow tail's forward half rides in the backward tail's launch (k_ip_strip_tail, inside the 'bwd' slot): the two
        # passes are priced together -- splitting the slots by pass would credit the forward with time it did not spend
        fl = {'stack': fl['fwd'] + fl['bwd'], 'wgrad': fl['wgrad']}
        seg = dict(seg, stack=seg['fwd'] + seg['bwd'])
    if seg and any(seg.get(k, 0) > 0 for k in fl):
        dom = max(fl, key=lambda k: seg.get(k, 0.0))
        ach = fl[dom] * B / (seg[dom] * 1e-3) / 1e12
        kname = {'fwd': 'k_ip_strip_fwd (deep stack forward, one launch)', 'bwd': 'k_ip_strip_bwd (deep stack backward-data, one launch)',
                 'stack': 'k_ip_strip_fwd + k_ip_strip_tail + k_ip_strip_bwd (deep stack forward and backward-data: wide products as pairs of '
                          '32-example strips, the narrow tail of both passes in one launch of 16-example strips)',
                 'wgrad': 'k_gemm_group (all weight gradients, one launch)'}[dom]
        roof = {'kernel': kname, 'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': ach / peak, 'traffic': pmc_traffic_ipnn(dom), 'avg_launch_ms': seg[dom], 'algorithmic_per_example': fl[dom]}
    ach_step = flops_ex * B / (ms_per_step * 1e-3) / 1e12
    if roof is None:
        roof = {'kernel': 'whole step', 'bound': 'mfma', 'achieved': ach_step, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': ach_step / peak, 'traffic': None, 'algorithmic_per_example': flops_ex}
    roof['step'] = {'achieved': ach_step, 'frac': ach_step / peak, 'unit': 'TFLOP/s', 'flops_per_example': flops_ex}
    cpu = None
    if not args.no_cpu_baseline and args.optimizer == 'sgd':
        # the NumPy float64 restatement (oracle/ipnn_oracle.py, BLAS threads of the host) on the same batches, ~10 s
        from oracle import ipnn_oracle as ipo
        params = {'b': 0.0, 'W': [w.astype(np.float64) for w in Ws], 'bias': [np.zeros(d[i + 1]) for i in range(len(d) - 1)]}
        table64 = rows.astype(np.float64)
        mk = [m.cpu().numpy().astype(np.float64) for m in masks[0]]
        n_done, t0c = 0, time.perf_counter()
        while n_done < 2 or time.perf_counter() - t0c < 10.0:
            b = n_done % NB
            ipo.sgd_step(params, table64, ids_np[b * B:(b + 1) * B], y_np[b * B:(b + 1) * B], 'relu', 1e-4, mk, 0.5)
            n_done += 1
            if n_done >= 50:
                break
        dtc = time.perf_counter() - t0c
        threads = 1
        try:
            from threadpoolctl import threadpool_info
            threads = max([t.get('num_threads', 1) for t in threadpool_info()] or [1])
        except Exception:
            pass
        cpu = {'value': n_done * B / dtc, 'unit': 'examples/sec', 'cores': threads, 'kind': 'port',
               'sample': '%d steps of batch %d on the same ids / masks (oracle.ipnn_oracle.sgd_step: NumPy float64, BLAS on %d threads; host has %s cores)'
                         % (n_done, B, threads, os.cpu_count())}
    eng.close()
    del masks, marr, ids, y
    torch.cuda.empty_cache()
    return ({
        'metric': 'examples/sec', 'value': B * args.steps / dt, 'unit': 'examples/sec', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': 'FNN_IP_L7 train step: 16 fields, 937670 rows, k=10, z1=297, hidden 1000/800/600/400/200/100/50 relu, '
                               'keep_prob 0.5 (mask inputs), batch %d, %s' % (B, args.optimizer.upper()), 'per_gpu_batch': B, 'global_batch': B,
                   'parallelism': 'single'},
        'train_logloss_last_step': loss.value / B, 'host_enqueue_ms_per_step': t_enq / args.steps * 1e3,
        'roofline': roof, 'cpu_baseline': cpu, 'kernel_ms': seg})


def pmc_traffic(kernel, snn=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, tools/pmc_traffic.sh; the newest
    profiles/*_pmc_traffic.json, KB).  gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports
    half the bytes of wide coalesced reads, so it is doubled (an upper bound for the mixed access
    widths of these kernels)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_snn.json' if snn else '*_pmc_traffic.json')))
    if not files:
        return None
    tag = {'step1': 'k_step1', 'step2': 'k_step2', 'step3': 'k_step3'}.get(kernel)
    for name, v in json.load(open(files[-1])).items():
        if tag and tag in name and 'FETCH_SIZE_KB_per_launch' in v and 'WRITE_SIZE_KB_per_launch' in v:
            return (2.0 * v['FETCH_SIZE_KB_per_launch'] + v['WRITE_SIZE_KB_per_launch']) * 1024.0
    return None


def pmc_traffic_gather(variant):
    """HBM bytes per launch of the standalone gather kernels, per variant (ids distribution / examples per launch), from the newest
    committed counter passes (profiles/*_pmc_traffic_gather.json, written by tools/pmc_gather.sh: one variant per process, FETCH_SIZE
    and WRITE_SIZE in separate passes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_gather.json')))
    if not files:
        return None
    v = json.load(open(files[-1])).get(variant)
    if not v or 'FETCH_SIZE_KB_per_launch' not in v or 'WRITE_SIZE_KB_per_launch' not in v:
        return None
    # FETCH_SIZE on gfx950 halves WIDE coalesced reads only; other widths want a calibration on a known byte count
    # (MI355X_MICROARCH.md, HBM).  The uniform-id variants are that calibration (profiles/r03_pmc_traffic_gather.json): A3's 64-byte
    # rows (4 lanes x 16 B) -- 262,144 random rows = 16.8 MB + 1 MB of ids expected, 16.4 MB counted: factor 1; A8's 800-byte rows
    # (50 lanes x 16 B, 7.25 128-byte lines each) -- 243 MB expected, 120 MB counted: factor 2.
    factor = 1.0 if variant.startswith('fm') else 2.0
    return (factor * v['FETCH_SIZE_KB_per_launch'] + v['WRITE_SIZE_KB_per_launch']) * 1024.0


def pmc_traffic_ipnn(seg):
    """HBM-side bytes per launch of the inner-product family's dominant kernel, from the newest committed PMC passes
    (profiles/*_pmc_traffic_ipnn.json; FETCH_SIZE doubled as in pmc_traffic)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_ipnn.json')))
    tags = {'fwd': ('k_ip_strip_fwd',), 'bwd': ('k_ip_strip_bwd',), 'wgrad': ('k_gemm_group',),
            'stack': ('k_ip_strip_fwd', 'k_ip_strip_tail', 'k_ip_strip_bwd')}.get(seg)
    if not files or not tags:
        return None
    tot, hit = 0.0, False
    for name, v in json.load(open(files[-1])).items():       # 'stack': the sum over its launches (wide and tail instances)
        if any(t in name for t in tags) and 'FETCH_SIZE_KB_per_launch' in v and 'WRITE_SIZE_KB_per_launch' in v:
            tot += (2.0 * v['FETCH_SIZE_KB_per_launch'] + v['WRITE_SIZE_KB_per_launch']) * 1024.0
            hit = True
            if seg != 'stack':
                break
    return tot if hit else None


def rocprof_avg_ms(kernel, precision='bf16'):
    """Average duration of `kernel` in the newest committed rocprofv3 --kernel-trace --stats summary of this command IN THIS
    PRECISION (profiles/r*_fnn_<precision>_kernel_stats.csv; for bf16 also the older r*_kernel_stats.csv of the headline
    command), for comparison with the event-bracketed time measured live."""
    import csv
    import glob
    allf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.csv')))
    files = [f for f in allf if os.path.basename(f).endswith('_fnn_%s_kernel_stats.csv' % precision)]
    if not files and precision == 'bf16':
        files = [f for f in allf if 'ipnn' not in f and '_fnn_' not in os.path.basename(f)]
    tag = {'step1': 'k_step1', 'step2': 'k_step2', 'step3': 'k_step3'}.get(kernel)
    if not files or not tag:
        return None
    for r in csv.DictReader(open(files[-1])):
        if tag in r.get('Name', ''):
            return float(r['AverageNs']) * 1e-6
    return None


def cpu_baseline(rows, ids_np, y_np, m1_np, m2_np, p0, B, seconds):
    """Times oracle/fnn_oracle.c (float64, scalar, 1 thread) on the same table / ids."""
    import __graft_entry__ as g
    if not os.path.exists(g.ORACLE_LIB):
        g.build()
    lib = C.CDLL(g.ORACLE_LIB)

    class Cfg(C.Structure):
        _fields_ = [("F", C.c_int), ("K", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("lr", C.c_double),
                    ("lambda1", C.c_double), ("lambda_fm", C.c_double), ("w0", C.c_double)]
    lib.oracle_train_step.restype = C.c_double
    cfg = Cfg(F, K, H1, H2, 0.001, 0.0, 0.1, -3.0)
    rows64 = np.ascontiguousarray(rows, dtype=np.float64)
    w = {k: np.ascontiguousarray(np.array(v, dtype=np.float64)) for k, v in p0.items() if k != 'b3'}
    b3 = C.c_double(0.0)
    dp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    n, t0 = 0, time.perf_counter()
    while True:
        b = n % (len(y_np) // B)
        sl = slice(b * B, (b + 1) * B)
        ids = np.ascontiguousarray(ids_np[sl]); yy = np.ascontiguousarray(y_np[sl], dtype=np.float64)
        r1 = np.ascontiguousarray(m1_np[b], dtype=np.float64); r2 = np.ascontiguousarray(m2_np[b], dtype=np.float64)
        lib.oracle_train_step(C.byref(cfg), dp(rows64), dp(ids), dp(yy), B, dp(r1), dp(r2), B, dp(w['w1']),
                              dp(w['b1']), dp(w['w2']), dp(w['b2']), dp(w['w3']), C.byref(b3), None, None)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {'value': n * B / el, 'unit': 'examples/sec', 'cores': 1, 'kind': 'port',
            'sample': '%d steps of batch %d on the same table/ids (oracle/fnn_oracle.c, float64, scalar; host has '
                      '%d cores)' % (n, B, os.cpu_count())}


def cpu_baseline_vectorised(rows, ids_np, y_np, m1_np, m2_np, p0, B, seconds):
    """SURVEY 8d variant (ii): the oracle's batch-level NumPy step (fancy-index gather, BLAS GEMMs,
    argsort-grouped closed-form row update) in float64 on all the host's cores -- what a fair CPU
    implementation costs; the scalar C port above stands for the reference's per-element loops."""
    from oracle import fnn_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([t.get('num_threads', 1) for t in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rows64 = np.array(rows, dtype=np.float64)
    p = {k: (np.array(v, dtype=np.float64) if not np.isscalar(v) else float(v)) for k, v in p0.items()}
    n, t0 = 0, time.perf_counter()
    while True:
        b = n % (len(y_np) // B)
        sl = slice(b * B, (b + 1) * B)
        orc.train_step_vec(p, rows64, -3.0, ids_np[sl], y_np[sl].astype(np.float64), m1_np[b].astype(np.float64),
                           m2_np[b].astype(np.float64), 0.001, 0.0, 0.1)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 2000:
            break
    return {'value': n * B / el, 'unit': 'examples/sec', 'cores': threads, 'kind': 'port',
            'sample': '%d steps of batch %d on the same table/ids (oracle.fnn_oracle.train_step_vec: NumPy float64, BLAS on %d '
                      'threads; host has %d cores)' % (n, B, threads, os.cpu_count())}


if __name__ == '__main__':
    main()
