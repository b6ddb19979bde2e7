"""Run-to-run and form-to-form bit-identity of the FNN_IP step under an environment knob (diagnostic)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import deep_ctr_amd  # noqa
from deep_ctr_amd.ipnn import IPNNEngine
from test_gpu_ipnn import problem, F, K
knob = sys.argv[1] if len(sys.argv) > 1 else 'IPNN_XT16'
for hidden, B in (([640, 500, 300, 70], 1000), ([1000, 800, 600, 400, 200, 100, 50], 4096)):
    steps = 3
    table, ids, y, params, masks, d = problem(B * steps, hidden, seed=B, n_rows=2000, scale=0.05)
    masks = [(np.random.RandomState(3 + t).uniform(size=(B * steps, d[t])) < 0.5).astype(np.uint8) for t in range(len(hidden) + 1)]
    def run(env):
        for k in ('IPNN_TAIL_SPLIT', 'IPNN_TAIL_FUSE', knob):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = IPNNEngine(F, K, hidden, 'tanh', max_batch=B, precision='bf16', lr=0.01, keep_prob=0.5)
        eng.set_params(table, params['b'], params['W'], params['bias'])
        out = []
        for s in range(steps):
            sl = slice(s * B, (s + 1) * B)
            o = eng.train_step(ids[sl], y[sl], [m[sl] for m in masks], want_logits=True)
            out.append(o['logits'].cpu().numpy().copy())
        b, Ws, bs = eng.get_params()
        eng.close()
        return np.concatenate(out), Ws
    ref, Wr = run({})
    bad = 0
    for rep in range(6):
        for env in ({knob: '1'}, {knob: '1', 'IPNN_TAIL_SPLIT': '0'}, {knob: '1', 'IPNN_TAIL_FUSE': '0'}):
            l, W = run(env)
            diffs = [(t, int((W[t] != Wr[t]).sum())) for t in range(len(W)) if (W[t] != Wr[t]).any()]
            if diffs or not np.array_equal(l, ref):
                bad += 1
                print(hidden[:2], B, rep, env, 'logits equal', np.array_equal(l, ref), 'W diffs', diffs)
    print(hidden[:2], B, 'runs that differ from the plain form:', bad, 'of 18')
