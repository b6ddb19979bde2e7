"""One rank of a two-process data-parallel rehearsal on ONE GPU (started by tests/test_gpu_dp.py through
`python -m torch.distributed.run --nproc-per-node 2`, gloo for the side channel and the callback collectives).  Separate
processes are what the peer-pointer all-reduce needs to be tested honestly: the exchange regions cross the process boundary as
hipIpc handles, and each rank's kernels sit in hardware queues of their own (two engines of one process can share a queue, and
a kernel that waits for a flag its peer raises from BEHIND it in the same queue never sees it).
Writes <out>/<form>_<case>_rank<r>.npz: dense tensors, table, per-step losses."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--form', required=True, choices=['slabs', 'bucket', 'p2p'])
    ap.add_argument('--case', required=True, choices=['plain', 'shadowed', 'ragged'])
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import deep_ctr_amd  # noqa: F401
    from deep_ctr_amd.dp import DataParallelFNN
    from test_gpu_parity import make_engine, make_problem
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    steps = 3 if a.case == 'plain' else 1
    if a.case == 'ragged':
        G, cut, mb = 7500, [slice(0, 4500), slice(4500, 7500)], 16384
    else:
        G, cut, mb = 1000, [slice(0, 512), slice(512, 1000)], 4096
    rows, fo, ids, y, p, r1, r2 = make_problem(steps * G, seed=65 if a.case == 'plain' else 91, dup_col=6)
    eng = make_engine(rows, fo, p, lr=0.01, lam1=0.05, lamfm=0.1, max_batch=mb)
    dp = DataParallelFNN(eng, sparse='local', payload=None if a.form == 'p2p' else a.form, collective='p2p' if a.form == 'p2p' else 'rccl')
    assert dp.config['collective'] == ('p2p' if a.form == 'p2p' else 'callback') and dp.config['payload'] == ('slabs' if a.form == 'slabs' else 'bucket')
    losses = []
    for s in range(steps):
        sl = slice(s * G, (s + 1) * G)
        if a.case == 'shadowed' and rank == 0:        # rank 0 alone is pushed to the layer-by-layer kernels
            rows_f2 = np.nonzero(fo == 2)[0]
            eng.set_shadowed(np.array([(t, 2, int(rows_f2[(t + 1) % len(rows_f2)])) for t in (3, 4, 9)], np.int32))
        losses.append(eng.train_step(ids[sl][cut[rank]], y[sl][cut[rank]], r1, r2, b_size=G)['loss'])
    eng.sync()
    d = eng.get_dense()
    np.savez(os.path.join(a.out, '%s_%s_rank%d.npz' % (a.form, a.case, rank)), table=eng.get_table(), losses=np.array(losses),
             b3=np.float64(d['b3']), **{k: v for k, v in d.items() if k != 'b3'})
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
