"""`python FNN.py [advertiser] [flag]` -- the FNN training script on MI355X.

Script behaviour (CLI, hyper-parameters, log lines, epoch loop, early stop, prediction pickles)
follows the reference's Theano script python/FNN_wnzh.py (the upstream `FNN.py`).
The Theano graph and the two Python loops around it are replaced by FNNEngine (libfnn_hip.so):
gather, MLP forward/backward, dense SGD and the sparse-row SGD all run as HIP kernels.

Differences a user can see: the data directory can be overridden with DEEPCTR_DATA_DIR (default
`../data`, as python/FNN_wnzh.py:28-30), DEEPCTR_EPOCHS caps the epoch count, DEEPCTR_PRECISION
selects f32 (default; parity mode) or bf16, and logloss is logged beside AUC and RMSE.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deep_ctr_amd  # noqa: E402,F401
from deep_ctr_amd import dl_utils as ut  # noqa: E402
from deep_ctr_amd.data_fm import DataFM  # noqa: E402
from deep_ctr_amd.engine import FNNEngine  # noqa: E402
from deep_ctr_amd.ipnn import FNN  # noqa: E402,F401  (the TensorFlow-style class of the same name, python/FNN.py:4)


def run(argv, out=None):
    """The module-level code of python/FNN_wnzh.py:15-348 as a function; returns the history."""
    srng = ut.RandomStreams(seed=234)                          # :15
    ut.seed_global(1234)                                       # :16-17 (and dl_utils.py:9-10)
    batch_size = 100                                           # :18-24
    lr = 0.002
    lambda1 = 0.1
    hidden1 = 300
    hidden2 = 100
    acti_type = 'tanh'
    epoch = int(os.environ.get('DEEPCTR_EPOCHS', 100))
    advertiser = '2997'
    if len(argv) > 1:
        advertiser = argv[1]
    data_dir = os.environ.get('DEEPCTR_DATA_DIR', '../data')
    train_file = os.path.join(data_dir, 'train.fm.txt')        # :28-30
    test_file = os.path.join(data_dir, 'test.fm.txt')
    fm_model_file = os.path.join(data_dir, 'fm.model.txt')
    if len(argv) > 2 and advertiser == 'all':                  # :32-35
        train_file = train_file + '.5.txt'
    elif len(argv) > 2:
        train_file = train_file + '.10.txt'
    print(train_file)

    train_size = ut.file_len(train_file)                       # :38-40
    n_batch = train_size // batch_size
    x_drop = 1
    if advertiser == '2997':                                   # :43-49
        lr = 0.001
        x_drop = dropout = 0.5
        hidden1 = 300
        hidden2 = 100
        lambda1 = 0.0
        lambda_fm = 0.1
    # any other advertiser: `dropout` / `lambda_fm` are undefined in the reference and the next
    # statement raises NameError there; same here.

    def log_p(msg, m=""):
        ut.logfile(msg, "fm" + str(advertiser))

    log_p('ad:' + str(advertiser))
    log_p('batch_size:' + str(batch_size))

    data = DataFM(fm_model_file)                               # :62-84
    k, xdim = data.k, data.xdim
    log_p('drop_mlp3fm.py|ad:' + advertiser + '|drop:' + str(dropout) + '|b_size:' + str(batch_size) +
          ' | X:' + str(xdim) + ' | Hidden 1:' + str(hidden1) + ' | Hidden 2:' + str(hidden2) +
          ' | L_r:' + str(lr) + ' | activation1:' + str(acti_type) + ' | lambda:' + str(lambda1))

    weights = ut.init_fnn_weights(xdim, hidden1, hidden2, acti_type)   # :106-130,140

    # dropout rows, in the graph's creation order (:144,154,166); r0 is dead but takes a seed
    r0 = srng.binomial(size=(1, xdim), n=1, p=x_drop)          # noqa: F841
    r1 = srng.binomial(size=(1, hidden1), n=1, p=dropout)
    r2 = srng.binomial(size=(1, hidden2), n=1, p=dropout)

    precision = os.environ.get('DEEPCTR_PRECISION', 'f32')
    eng = FNNEngine(n_fields=len(data.name_field), k=k, hidden1=hidden1, hidden2=hidden2,
                    max_batch=max(batch_size, 4096), precision=precision, acti_type=acti_type,
                    lr=lr, lambda1=lambda1, lambda_fm=lambda_fm)
    rows, field_of_row, w_0 = data.table()
    eng.set_table(rows, field_of_row, w_0)
    eng.set_dense(weights)
    data.engine = eng

    train_ids, train_y, train_sh = data.load_ids(train_file, want_shadowed=True)   # parsed once, not per epoch
    test_ids, test_y = data.load_ids(test_file)

    train_ids_d, train_y_d = eng.to_device(train_ids, train_y)   # resident in HBM for every epoch's passes
    test_ids_d, test_y_d = eng.to_device(test_ids, test_y)
    train_yf_d = train_y_d.float()

    def get_err_bat(ids, y):                                   # :193-221, metrics on the device (fnn_eval)
        m = eng.evaluate(ids, y)
        return m['auc'], m['rmse'], m['logloss']

    def fmt_time(t):
        return str(int(t / 60)) + 'm ' + str(int(t % 60)) + 's'

    print("Training model:")
    best = eng.get_dense()                                     # :278-284
    min_err = 0
    min_err_epoch = 0
    times_reduce = 0
    hist = []
    for i in range(epoch):                                     # :290
        start_time = time.time()
        pre_step = best
        # :293-306.  The epoch's dropout rows are drawn ahead (the same stream: _BinomialOp.draw_rows) and the steps run
        # through FNNEngine.train_epoch -- resident ids, raw C calls, the next batch announced to every step; `train` returns
        # PRE-update tensors (:298), so the dense state is read before the last batch
        n_run = n_batch if (n_batch - 1) * batch_size + 1 <= train_size else (train_size + batch_size - 1) // batch_size
        m1, m2 = r1.draw_rows(n_run), r2.draw_rows(n_run)
        if n_run > 1:
            eng.train_epoch(train_ids_d, train_yf_d, batch_size, m1, m2, 0, n_run - 1, train_sh)
        if n_run > 0:
            if n_run == n_batch:
                pre_step = eng.get_dense()
            eng.train_epoch(train_ids_d, train_yf_d, batch_size, m1, m2, n_run - 1, 1, train_sh)
        eng.sync()
        print('training: ' + fmt_time(time.time() - start_time))

        start_time = time.time()
        auc, rmse, ll = get_err_bat(train_ids_d, train_y_d)
        log_p('\t\tTraining Err: \t' + str(i) + '\t' + str(auc) + '\t' + str(rmse))
        print('training error: ' + fmt_time(time.time() - start_time))

        start_time = time.time()
        auc, rmse, ll = get_err_bat(test_ids_d, test_y_d)
        log_p('Test Err:' + str(i) + '\t' + str(auc) + '\t' + str(rmse))
        log_p('Test logloss:' + str(i) + '\t' + str(ll))
        print('test error: ' + fmt_time(time.time() - start_time))
        hist.append({'epoch': i, 'test_auc': auc, 'test_rmse': rmse, 'test_logloss': ll})

        if auc > min_err:                                      # :329-343
            best = pre_step
            min_err = auc
            min_err_epoch = i
            if times_reduce < 3:
                times_reduce += 1
        else:
            times_reduce -= 1
        if times_reduce < 0:
            break
    log_p('Minimal test error is ' + str(min_err) + ' , at EPOCH ' + str(min_err_epoch))

    eng.set_dense(best)                                        # get_pred, :257-276,345-348
    ut.save_weights("mlp3fm_train_" + advertiser + ".p", eng.predict(train_ids).cpu().numpy())
    ut.save_weights("mlp3fm_test_" + advertiser + ".p", eng.predict(test_ids).cpu().numpy())
    return hist


if __name__ == '__main__':
    run(sys.argv)
