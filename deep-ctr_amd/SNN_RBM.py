"""`python SNN_RBM.py [advertiser] [flag]` -- the SNN-RBM script on MI355X.

Script behaviour (CLI, hyper-parameters, pre-train cache, log lines, epoch loop, early stop)
follows the reference's python/SNN_RBM.py.  Layer-wise CD-1 pre-training runs on the HIP kernels
behind include/rbm_hip.h (`get_rbm_weights`), the fine-tune loop on FNNEngine in bag mode: the
embedding-bag + sigmoid input layer (`get_fi_h1_y`, :238-262), the Theano graph (:105-153) and the
per-example row updates (:285-291) are HIP kernels.

Environment: DEEPCTR_DATA_DIR (default ../data), DEEPCTR_EPOCHS, DEEPCTR_PRECISION (f32 | bf16),
DEEPCTR_XDIM (default 133465 as hard-coded at :49; `auto` = largest feature id + 1),
DEEPCTR_LOG_DIR.
"""
import os
import pickle
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deep_ctr_amd  # noqa: E402,F401
from deep_ctr_amd import dl_utils as ut  # noqa: E402
from deep_ctr_amd import sampling_based_gaussian_binary_rbm_sparse as gbrbm  # noqa: E402
from deep_ctr_amd.engine import FNNEngine  # noqa: E402


def load_active_ids(path, n_fields=16):
    """The parse of get_fi_h1_y / auc_rmse (:238-262, :176-186): split on single spaces, a feature
    counts only when its value is 1.  Returns (ids int32 [N, n_fields] with -1 padding, y [N]).
    One native pass over the file (ctr_parse_examples, CTR_MODE_SNN_ACTIVE)."""
    from deep_ctr_amd import ingest
    ids, _, y = ingest.parse_examples(path, ingest.MODE_SNN_ACTIVE, None, n_fields)
    return ids, y


def run(argv, kind='rbm'):
    """kind = 'rbm': python/SNN_RBM.py.  kind = 'dae': python/SNN_DAE.py -- the same script with the
    denoising-autoencoder pre-trainer, its own 2997 hyper-parameters (:42-50), no train-file suffix
    flag and the cache file dropda_<adv>_.p (:82-86)."""
    srng = ut.RandomStreams(seed=234)                          # :18
    ut.seed_global(1234)                                       # :19-21 (and the two imports before it)
    batch_size = 1000                                          # :22-29
    lr = 0.0006
    lambda1 = 0.0001
    hidden0 = 300
    hidden1 = 300
    hidden2 = 100
    acti_type = 'tanh'
    epoch = int(os.environ.get('DEEPCTR_EPOCHS', 100))
    advertiser = '2997'
    if len(argv) > 1:
        advertiser = argv[1]
    data_dir = os.environ.get('DEEPCTR_DATA_DIR', '../data')
    train_file = os.path.join(data_dir, 'train.fm.txt')        # :32-34
    test_file = os.path.join(data_dir, 'test.fm.txt')
    fm_model_file = os.path.join(data_dir, 'fm.model.txt')
    if kind == 'dae':
        pass                                                   # python/SNN_DAE.py:30-36: no suffix variants
    elif len(argv) > 2 and advertiser == 'all':                # :39-44
        train_file = train_file + '.10.txt'
    elif len(argv) > 2 and argv[2] == "mod2":
        train_file = train_file + '.2.txt'
    elif len(argv) > 2:
        train_file = train_file + '.5.txt'
    print(train_file)
    train_size = ut.file_len(train_file)                       # :46-48
    n_batch = train_size // batch_size
    x_dim = 133465                                             # :49
    dropout = 1
    if advertiser == '2997':                                   # :52-58
        hidden0 = 200
        hidden1 = 300
        hidden2 = 100
        lr = 0.001
        dropout = 0.98
        lambda1 = 0
        if kind == 'dae':                                      # python/SNN_DAE.py:42-50
            lr = 0.0005
            dropout = 0.99
    train_ids, train_y = load_active_ids(train_file)
    test_ids, test_y = load_active_ids(test_file)
    xd = os.environ.get('DEEPCTR_XDIM')
    if xd == 'auto':
        x_dim = int(max(train_ids.max(), test_ids.max())) + 1
    elif xd:
        x_dim = int(xd)

    def log_p(msg, m=""):
        ut.log_p(msg, "drop_mlp4da" + str(advertiser))

    log_p('drop_mlp4da.py|ad:' + advertiser + '|drop:' + str(dropout) + '|b_size:' + str(batch_size) + ' | X:' +
          str(x_dim) + ' | Hidden 0:' + str(hidden0) + ' | Hidden 1:' + str(hidden1) + ' | Hidden 2:' + str(hidden2) +
          ' | L_r:' + str(lr) + ' | activation1:' + str(acti_type) + ' | lambda:' + str(lambda1))

    arr = [x_dim, hidden0, hidden1, hidden2]
    ww0, bb0 = ut.init_weight(x_dim, hidden0, 'sigmoid')       # :78-80 (consume the RNG even when a cache exists)
    ww1, bb1 = ut.init_weight(hidden0, hidden1, 'sigmoid')
    ww2, bb2 = ut.init_weight(hidden1, hidden2, 'sigmoid')

    precision = os.environ.get('DEEPCTR_PRECISION', 'f32')
    wfile = ("dropda_" if kind == 'dae' else "rbm_") + str(advertiser) + "_.p"     # :82-88 / SNN_DAE.py:82
    if os.path.isfile(wfile):
        (ww0, bb0, ww1, bb1, ww2, bb2) = pickle.load(open(wfile, "rb"))
    elif kind == 'dae':
        from deep_ctr_amd import sampling_based_denosing_autoencoder as da
        with open(train_file) as fi:                           # num_feats, python/SNN_DAE.py:75-81
            numf = len(fi.readline().strip().split(':')) - 1
        ww0, bb0, ww1, bb1, ww2, bb2 = da.get_da_weights(train_file, arr, num_feats=numf, ncases=train_size,
                                                         batch_size=100000)
        pickle.dump((ww0, bb0, ww1, bb1, ww2, bb2), open(wfile, "wb"))
    else:
        ww0, bb0, ww1, bb1, ww2, bb2 = gbrbm.get_rbm_weights(train_file, arr, ncases=train_size, batch_size=100000,
                                                             fm_model_file=fm_model_file, precision=precision)
        pickle.dump((ww0, bb0, ww1, bb1, ww2, bb2), open(wfile, "wb"))

    ww3 = ut.rng.uniform(-0.05, 0.05, hidden2)                 # :91 (drawn, then overwritten)
    ww3 = numpy.zeros(hidden2)
    bb3 = 0.

    r1 = srng.binomial(size=(1, hidden1), n=1, p=dropout)      # :117,130 (no input mask in this script)
    r2 = srng.binomial(size=(1, hidden2), n=1, p=dropout)

    eng = FNNEngine(n_fields=train_ids.shape[1], k=0, hidden1=hidden1, hidden2=hidden2, max_batch=4096,
                    precision=precision, acti_type=acti_type, lr=lr, lambda1=lambda1, lambda_fm=0.0, reg_all=True,
                    mode='bag', hidden0=hidden0)
    eng.set_table(ww0, numpy.zeros(ww0.shape[0], numpy.int32), 0.0)
    eng.set_bag_bias(bb0)
    eng.set_dense({'w1': ww1, 'b1': bb1, 'w2': ww2, 'b2': bb2, 'w3': ww3, 'b3': bb3})

    def auc_rmse(ids, y):                                      # :162-198
        m = eng.evaluate(ids, y)                               # predictions and metrics stay on the device (fnn_eval)
        return m['auc'], m['rmse'], m['logloss']

    def fmt_time(t):
        return str(int(t / 60)) + 'm ' + str(int(t % 60)) + 's'

    train_ids_d, train_y_d = eng.to_device(train_ids, train_y)   # resident in HBM: no per-batch copies, no per-epoch parsing
    test_ids_d, test_y_d = eng.to_device(test_ids, test_y)
    train_yf_d = train_y_d.float()
    print("Training model:")                                   # mytrain, :269-323
    min_err = 0
    min_err_epoch = 0
    times_reduce = 0
    hist = []
    for i in range(epoch):
        start_time = time.time()
        # the epoch's dropout rows drawn ahead (same stream), the steps through FNNEngine.train_epoch: resident ids, raw C calls,
        # the next batch announced to every step (grouped beside this step's work)
        if n_batch > 0:
            eng.train_epoch(train_ids_d, train_yf_d, batch_size, r1.draw_rows(n_batch), r2.draw_rows(n_batch), 0, n_batch)
        eng.sync()
        print('training: ' + fmt_time(time.time() - start_time))
        start_time = time.time()
        auc, rmse, ll = auc_rmse(train_ids_d, train_y_d)
        log_p('\t\tTraining Err: \t' + str(i) + '\t' + str(auc) + '\t' + str(rmse))
        print('training error: ' + fmt_time(time.time() - start_time))
        start_time = time.time()
        auc, rmse, ll = auc_rmse(test_ids_d, test_y_d)
        log_p('Test Err:' + str(i) + '\t' + str(auc) + '\t' + str(rmse))
        log_p('Test logloss:' + str(i) + '\t' + str(ll))
        print('test error: ' + fmt_time(time.time() - start_time))
        hist.append({'epoch': i, 'test_auc': auc, 'test_rmse': rmse, 'test_logloss': ll})
        if auc > min_err:                                      # :314-322
            min_err = auc
            min_err_epoch = i
            if times_reduce < 3:
                times_reduce += 1
        else:
            times_reduce -= 1
        if times_reduce < -2:
            break
    log_p('Minimal test error is ' + str(min_err) + ' , at EPOCH ' + str(min_err_epoch))
    return hist


if __name__ == '__main__':
    run(sys.argv)
