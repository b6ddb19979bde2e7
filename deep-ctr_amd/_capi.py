"""ctypes binding of libfnn_hip.so (C ABI: include/fnn_hip.h).

This is the same stub a maintainer of the reference would add next to `FNN_wnzh.py` to replace
`theano.function` (python/FNN_wnzh.py:177-183); see INTEGRATION.md.  The library is looked up
in-tree (deep-ctr_amd/libfnn_hip.so, built by `__graft_entry__.build()`); a missing library is
an ImportError-class failure, never a silent fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FNN_HIP_LIB: another build of the same library (A/B measurements of a kernel change on one box); never a different backend
LIB_PATH = os.environ.get("FNN_HIP_LIB") or os.path.join(_HERE, "libfnn_hip.so")

FNN_OK = 0
FNN_ERR_ARG, FNN_ERR_HIP, FNN_ERR_STATE, FNN_ERR_RANGE, FNN_ERR_NOMEM = -1, -2, -3, -4, -5
FNN_PREC_F32, FNN_PREC_BF16, FNN_PREC_BF16X3 = 0, 1, 2
PRECISIONS = {'f32': FNN_PREC_F32, 'bf16': FNN_PREC_BF16, 'bf16x3': FNN_PREC_BF16X3}
FNN_ACT_TANH, FNN_ACT_SIGMOID, FNN_ACT_LINEAR = 0, 1, 2
FNN_MEM_HOST, FNN_MEM_DEVICE = 0, 1
FNN_MODE_FM, FNN_MODE_BAG = 0, 1
FNN_DP_SPARSE_LOCAL, FNN_DP_SPARSE_EXCHANGE = 0, 1
FNN_DP_PAYLOAD_SLABS, FNN_DP_PAYLOAD_BUCKET = 0, 1
FNN_DP_COLLECTIVE_CALLBACK, FNN_DP_COLLECTIVE_P2P = 0, 1
# collective callbacks of fnn_dp_init_custom: (ctx, buf, n_floats, stream) / (ctx, send, recv, bytes_per_rank, stream) -> 0 = OK
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class fnn_cfg(C.Structure):
    _fields_ = [("n_fields", C.c_int32), ("k", C.c_int32), ("hidden1", C.c_int32),
                ("hidden2", C.c_int32), ("max_batch", C.c_int32), ("precision", C.c_int32),
                ("act", C.c_int32), ("reg_all", C.c_int32), ("lr", C.c_float),
                ("lambda1", C.c_float), ("lambda_fm", C.c_float), ("device", C.c_int32),
                ("stream", C.c_void_p), ("mode", C.c_int32), ("h0", C.c_int32)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes): every symbol include/fnn_hip.h declares
SIGNATURES = {
    "fnn_version": (C.c_char_p, []),
    "fnn_cfg_size": (C.c_uint64, []),
    "fnn_last_error": (C.c_char_p, [_vp]),
    "fnn_create": (_i, [C.POINTER(fnn_cfg), C.POINTER(_vp)]),
    "fnn_destroy": (_i, [_vp]),
    "fnn_set_hparams": (_i, [_vp, _f, _f, _f]),
    "fnn_stream": (_vp, [_vp]),
    "fnn_sync": (_i, [_vp]),
    "fnn_set_table": (_i, [_vp, _vp, _i64, _vp, _f, _i]),
    "fnn_get_table": (_i, [_vp, _vp, _i]),
    "fnn_get_rows": (_i, [_vp, _vp, _i64, _vp, _i]),
    "fnn_set_dense": (_i, [_vp, _i, _vp, _vp, _i]),
    "fnn_get_dense": (_i, [_vp, _i, _vp, _vp, _i]),
    "fnn_set_bag_bias": (_i, [_vp, _vp, _i]),
    "fnn_get_bag_bias": (_i, [_vp, _vp, _i]),
    "fnn_gather": (_i, [_vp, _vp, _i, _vp, _i]),
    "fnn_train_step": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, C.POINTER(_f)]),
    "fnn_set_shadowed": (_i, [_vp, _vp, _i, _i]),
    "fnn_prefetch_ids": (_i, [_vp, _vp, _i]),
    "fnn_dp_unique_id": (_i, [_vp]),
    "fnn_dp_init": (_i, [_vp, _i, _i, _vp, _i]),
    "fnn_dp_init_custom": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i]),
    "fnn_dp_shutdown": (_i, [_vp]),
    "fnn_dp_set_payload": (_i, [_vp, _i]),
    "fnn_dp_p2p_export": (_i, [_vp, _vp, _i]),
    "fnn_dp_p2p_attach": (_i, [_vp, _vp, _i]),
    "fnn_dp_set_collective": (_i, [_vp, _i]),
    "fnn_dp_get_config": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "fnn_dp_p2p_max_wait_us": (_i, [_vp, C.POINTER(C.c_double)]),
    "fnn_step_begin": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i]),
    "fnn_dense_grad_bucket": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "fnn_step_scatter": (_i, [_vp]),
    "fnn_sparse_grad": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "fnn_step_scatter_global": (_i, [_vp, _vp, _vp, _i]),
    "fnn_step_end": (_i, [_vp, C.POINTER(_f)]),
    "fnn_last_loss": (_i, [_vp, C.POINTER(_f)]),
    "fnn_predict": (_i, [_vp, _vp, _i, _vp, _i]),
    "fnn_eval": (_i, [_vp, _vp, _vp, _i64, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "fnn_prof_enable": (_i, [_vp, _i]),
    "fnn_prof_reset": (_i, [_vp]),
    "fnn_prof_get": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
}

_d = C.c_double
# every symbol include/rbm_hip.h declares
RBM_SIGNATURES = {
    "rbm_last_error": (C.c_char_p, []),
    "rbm_sparse_epoch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _f, _f, _f, _f, C.POINTER(_d), _vp]),
    "rbm_sparse_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _f, _f, _f, _f, _f, C.POINTER(_d), _vp]),
    "rbm_sparse_batch_dp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, C.POINTER(_d), _vp]),
    "rbm_dense_create": (_i, [_i, _i, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "rbm_dense_destroy": (_i, [_vp]),
    "rbm_dense_set": (_i, [_vp, _vp, _vp, _vp]),
    "rbm_dense_get": (_i, [_vp, _vp, _vp, _vp]),
    "rbm_dense_cd1": (_i, [_vp, _vp, _i, _vp, _f, _f, _f, _f, _f, C.POINTER(_d)]),
    "rbm_bag_sum": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _vp, _vp]),
    "rbm_affine": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "rbm_sigmoid": (_i, [_vp, _i64, _vp]),
}

class ipnn_cfg(C.Structure):
    _fields_ = [("n_fields", C.c_int32), ("k", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32 * 8),
                ("act", C.c_int32), ("pairs", C.c_int32), ("max_batch", C.c_int32), ("precision", C.c_int32), ("lr", C.c_float),
                ("keep_prob", C.c_float), ("optimizer", C.c_int32), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
                ("adam_eps", C.c_float), ("device", C.c_int32), ("stream", C.c_void_p)]


IPNN_ACTS = {'tanh': 0, 'sigmoid': 1, 'relu': 3}
# every symbol include/ipnn_hip.h declares
IPNN_SIGNATURES = {
    "ipnn_last_error": (C.c_char_p, [_vp]),
    "ipnn_cfg_size": (C.c_uint64, []),
    "ipnn_create": (_i, [C.POINTER(ipnn_cfg), C.POINTER(_vp)]),
    "ipnn_destroy": (_i, [_vp]),
    "ipnn_sync": (_i, [_vp]),
    "ipnn_set_table": (_i, [_vp, _vp, _i64]),
    "ipnn_get_rows": (_i, [_vp, _vp, _i64, _vp]),
    "ipnn_set_b": (_i, [_vp, _f]),
    "ipnn_get_b": (_i, [_vp, C.POINTER(_f)]),
    "ipnn_set_layer": (_i, [_vp, _i, _vp, _vp]),
    "ipnn_get_layer": (_i, [_vp, _i, _vp, _vp]),
    "ipnn_train_step": (_i, [_vp, _vp, _vp, _i, _vp, _vp, C.POINTER(_f)]),
    "ipnn_set_loss_mean": (_i, [_vp, _i]),
    "ipnn_predict": (_i, [_vp, _vp, _i, _vp]),
    "ipnn_eval": (_i, [_vp, _vp, _vp, _i64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ipnn_prof_enable": (_i, [_vp, _i]),
    "ipnn_prof_get": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
}

# every symbol include/dae_hip.h declares
DAE_SIGNATURES = {
    "dae_last_error": (C.c_char_p, []),
    "dae_sparse_epoch": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, C.POINTER(C.c_double), _vp]),
    "dae_dense_epoch": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, C.POINTER(C.c_double), _vp]),
    "dae_bag_cumsum_sigmoid": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _vp, _vp]),
    "dae_sparse_epoch_f64": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, C.c_double, C.POINTER(C.c_double), _vp]),
    "dae_dense_epoch_f64": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, C.c_double, _i, C.POINTER(C.c_double), _vp]),
    "dae_bag_cumsum_sigmoid_f64": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _vp, _vp]),
    "dae_affine_sigmoid_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
}

# every symbol include/fm_hip.h declares
FM_SIGNATURES = {
    "fm_last_error": (C.c_char_p, [_vp]),
    "fm_create": (_i, [_i, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "fm_destroy": (_i, [_vp]),
    "fm_sync": (_i, [_vp]),
    "fm_set_table": (_i, [_vp, _vp, _i64]),
    "fm_get_table": (_i, [_vp, _vp]),
    "fm_get_rows": (_i, [_vp, _vp, _i64, _vp]),
    "fm_set_b": (_i, [_vp, _f]),
    "fm_get_b": (_i, [_vp, C.POINTER(_f)]),
    "fm_train_step": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, C.POINTER(_f)]),
    "fm_predict": (_i, [_vp, _vp, _i, _vp]),
}

_i64p = C.POINTER(_i64)
# every symbol include/ctr_ingest.h declares (host code: native text ingestion)
CTR_SIGNATURES = {
    "ctr_last_error": (C.c_char_p, []),
    "ctr_fm_model_load": (_i, [C.c_char_p, C.POINTER(C.c_char_p), _i, _i, C.POINTER(_vp)]),
    "ctr_fm_model_free": (None, [_vp]),
    "ctr_fm_model_n_rows": (_i64, [_vp]),
    "ctr_fm_model_k": (_i, [_vp]),
    "ctr_fm_model_w0": (C.c_double, [_vp]),
    "ctr_fm_model_copy": (_i, [_vp, _vp, _vp, _vp]),
    "ctr_fm_model_from_arrays": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(_vp)]),
    "ctr_count_lines": (_i, [C.c_char_p, _i, _i64p, _i64p]),
    "ctr_parse_examples": (_i, [C.c_char_p, _i, _vp, _i, _i, _i64, _vp, _vp, _vp, _i64p]),
    "ctr_parse_examples_ex": (_i, [C.c_char_p, _i, _vp, _i, _i, _i64, _vp, _vp, _vp, _i64p, _i64, _vp, _i64p]),
    "ctr_yzx_stat": (_i, [C.c_char_p, _i, _i64p, _i64p, _i64p]),
    "ctr_parse_yzx": (_i, [C.c_char_p, _i, _i64, _i64, _i, _vp, _vp, _vp, _i64p]),
}

_lib = None


def load():
    """Load libfnn_hip.so and declare every prototype.  Raises if the library is missing:
    the product path has no other backend."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7).  Import torch
    # FIRST so that the dynamic loader resolves libfnn_hip.so's libamdhip64.so.7 to that already
    # loaded runtime; the other order maps two HIP/HSA runtimes into one process and the second
    # one finds no device.  (A plain C host links against /opt/rocm as usual.)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfnn_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(RBM_SIGNATURES.items()) + list(IPNN_SIGNATURES.items()) + \
            list(CTR_SIGNATURES.items()) + list(DAE_SIGNATURES.items()) + list(FM_SIGNATURES.items()):
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # the structs are declared twice (include/*.h and above): a mismatch would make *_create read past the caller's struct
    for name, st in (("fnn_cfg_size", fnn_cfg), ("ipnn_cfg_size", ipnn_cfg)):
        if getattr(lib, name)() != C.sizeof(st):
            raise ImportError("%s() = %d but ctypes declares %d bytes: _capi.py is out of date with include/*.h"
                              % (name, getattr(lib, name)(), C.sizeof(st)))
    _lib = lib
    return lib
