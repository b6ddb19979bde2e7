"""`python SNN_DAE.py [advertiser]` -- the SNN-DAE script on MI355X.

Follows the reference's python/SNN_DAE.py: the SNN fine-tune loop of SNN_RBM.py (embedding-bag +
sigmoid input layer, three-layer MLP with dropout rows, per-example row updates -- all HIP kernels
of FNNEngine's bag mode) on top of layer-wise denoising-autoencoder pre-training
(`sampling_based_denosing_autoencoder.get_da_weights`, HIP kernels behind include/dae_hip.h),
cached in `dropda_<adv>_.p`.  Environment variables as SNN_RBM.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deep_ctr_amd  # noqa: E402,F401
from deep_ctr_amd import SNN_RBM  # noqa: E402


def run(argv):
    return SNN_RBM.run(argv, kind='dae')


if __name__ == '__main__':
    run(sys.argv)
