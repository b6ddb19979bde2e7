"""Host-side utilities of the Theano family, same names as the reference's python/dl_utils.py
(logging, init_weight, file_len, savers) plus the two host-side random sources the hot path
consumes: the weight initialiser and the dropout-row stream.  Pure host logic, no compute path.
"""
import os
import pickle
from time import gmtime, strftime

import numpy

# python/dl_utils.py:9-10 seeds the global legacy RNG at import; here the stream is an object so
# importing this module has no side effect -- call seed_global() for script parity.
rng = numpy.random


def seed_global(seed=1234):
    rng.seed(seed)


log_path = os.environ.get('DEEPCTR_LOG_DIR', '../log/')     # python/dl_utils.py:13


def _log_file():
    if not os.path.exists(log_path):                        # python/dl_utils.py:14-15
        os.makedirs(log_path)
    return os.path.join(log_path, 'log-' + strftime("%Y-%m-%d", gmtime()))


def save_weights(file, tuple_weights):                      # python/dl_utils.py:19-20
    pickle.dump(tuple_weights, open(file, "wb"))


def save_prediction(file, prediction):                      # python/dl_utils.py:23-24
    pickle.dump(prediction, open(file, "wb"))


def log(msg, file=""):                                      # python/dl_utils.py:28-30
    with open(_log_file() + file + '.txt', "a+") as myfile:
        myfile.write(msg + "\n")


def logfile(msg, file):                                     # python/dl_utils.py:33-36
    print(msg)
    if not os.path.exists(log_path):
        os.makedirs(log_path)
    with open(os.path.join(log_path, file + '.txt'), "a+") as myfile:
        myfile.write(msg + "\n")


def log_p(msg, file=""):                                    # python/dl_utils.py:39-41
    log(msg, file)
    print(msg)


def _glorot(fan_in, fan_out, gain_for, acti_type):
    """One Glorot-uniform draw from the global legacy RNG, bound sqrt(6/(fan_in+fan_out)), scaled
    x4 when acti_type == gain_for; any other acti_type than sigmoid/tanh draws U(-1,1) instead
    (after the first draw, as the reference does)."""
    bound = numpy.sqrt(6. / (fan_in + fan_out))
    w = rng.uniform(low=-bound, high=bound, size=(fan_in, fan_out))
    if acti_type not in ('sigmoid', 'tanh'):
        return numpy.asarray(rng.uniform(-1, 1, size=(fan_in, fan_out)))
    return numpy.asarray(w * 4) if acti_type == gain_for else numpy.asarray(w)


def init_weight(hidden1, hidden2, acti_type):
    """python/dl_utils.py:44-56: x4 when SIGMOID (the opposite convention of
    python/FNN_wnzh.py:109-114); returns (W [hidden1,hidden2], zeros(hidden2))."""
    return _glorot(hidden1, hidden2, 'sigmoid', acti_type), numpy.zeros(hidden2)


def init_fnn_weights(xdim, hidden1, hidden2, acti_type='tanh'):
    """python/FNN_wnzh.py:106-130,140: w1 then w2 from the global RNG (x4 when TANH), zero biases,
    w3 = 0, b3 = 0."""
    ww1 = _glorot(xdim, hidden1, 'tanh', acti_type)
    ww2 = _glorot(hidden1, hidden2, 'tanh', acti_type)
    return {'w1': ww1, 'b1': numpy.zeros(hidden1), 'w2': ww2, 'b2': numpy.zeros(hidden2),
            'w3': numpy.zeros(hidden2), 'b3': 0.0}


class RandomStreams(object):
    """Host stand-in for theano's shared_randomstreams.RandomStreams as the scripts use it
    (python/FNN_wnzh.py:15,144,154,166): every `binomial` op gets its own
    RandomState(seedgen.randint(2**30)) in creation order and draws one (1,H) row per `train`
    call.  The rows are handed to the HIP step as uint8 masks."""

    def __init__(self, seed=234):
        self._seedgen = numpy.random.RandomState(seed)
        self.ops = []

    def binomial(self, size, n=1, p=0.5):
        st = numpy.random.RandomState(int(self._seedgen.randint(2 ** 30)))
        op = _BinomialOp(st, size, n, p)
        self.ops.append(op)
        return op


class _BinomialOp(object):
    def __init__(self, state, size, n, p):
        self.state, self.size, self.n, self.p = state, size, n, p

    def draw(self):
        return self.state.binomial(n=self.n, p=self.p, size=self.size)

    def draw_rows(self, n):
        """The next n draws of a (1, H) op as one [n, H] array: the legacy RandomState binomial consumes its stream element by
        element, so one call of size (n, H) IS n calls of size (1, H) (tests/test_host.py holds it to that)."""
        assert len(self.size) == 2 and self.size[0] == 1
        return self.state.binomial(n=self.n, p=self.p, size=(n, self.size[1]))


def file_len(fname):                                        # python/dl_utils.py:111-115
    with open(fname) as f:
        for i, l in enumerate(f):
            pass
    return i + 1


def feats_len(fname):                                       # python/dl_utils.py:119-122
    with open(fname) as f:
        l = len(f.readline().split(','))
    return (l - 1)
