"""deep-ctr on MI355X: the FNN / SNN CTR hot path of Atomu2014/deep-ctr as hand-written HIP
kernels for gfx950 behind a C ABI (include/fnn_hip.h, libfnn_hip.so), plus the host-side mirror
of the reference's loader / script interface (data_fm.py, ipinyou.py, dl_utils.py, FNN.py).

There is no CPU compute path in this package: without libfnn_hip.so and a HIP device the
engine raises."""
from . import _capi            # noqa: F401  (ctypes binding; loading is lazy)
from .engine import FNNEngine, FNNError  # noqa: F401

__all__ = ["FNNEngine", "FNNError"]
