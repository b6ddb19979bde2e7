"""Native text ingestion (include/ctr_ingest.h, host code of libfnn_hip.so): the reference's
per-line Python parsers -- python/FNN_wnzh.py:62-84 and :224-253, python/SNN_RBM.py:238-262,
python/sampling_based_gaussian_binary_rbm_sparse.py:142-156, python/ipinyou.py:23-65 -- as one
multi-threaded mmap pass per file.  Errors keep the reference's exception types: ValueError for
a malformed token, KeyError for an unknown feature or field name, IOError for a missing file."""
import ctypes as C
import os

import numpy as np

from . import _capi

MODE_FNN, MODE_SNN_ACTIVE, MODE_PAIRS = 0, 1, 2
_EXC = {-1: ValueError, -2: IOError, -3: ValueError, -4: KeyError, -5: IndexError, -6: IndexError}


def n_threads():
    return max(1, min(16, os.cpu_count() or 1))


def _ck(lib, rc):
    if rc != 0:
        raise _EXC.get(rc, RuntimeError)((lib.ctr_last_error() or b'').decode())


class FMModel(object):
    """Parsed `fm.model.txt` (A1) or an id map built from arrays; owns the native handle."""

    def __init__(self, handle):
        self.lib = _capi.load()
        self.h = handle

    @classmethod
    def load(cls, path, field_names, threads=None):
        lib = _capi.load()
        names = (C.c_char_p * len(field_names))(*[n.encode() for n in field_names])
        h = C.c_void_p()
        _ck(lib, lib.ctr_fm_model_load(os.fsencode(path), names, len(field_names), threads or n_threads(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, feat_ids, field_of_row, k, n_fields):
        lib = _capi.load()
        fi = np.ascontiguousarray(feat_ids, dtype=np.int64)
        fo = np.ascontiguousarray(field_of_row, dtype=np.int32)
        h = C.c_void_p()
        _ck(lib, lib.ctr_fm_model_from_arrays(fi.ctypes.data, fo.ctypes.data, len(fi), k, n_fields, C.byref(h)))
        return cls(h)

    @property
    def n_rows(self):
        return int(self.lib.ctr_fm_model_n_rows(self.h))

    @property
    def k(self):
        return int(self.lib.ctr_fm_model_k(self.h))

    @property
    def w0(self):
        return float(self.lib.ctr_fm_model_w0(self.h))

    def arrays(self, want_rows=True):
        """(rows float64 [n,k] or None, feat_ids int64 [n], field_of_row int32 [n]), file order."""
        n, k = self.n_rows, self.k
        rows = np.empty((n, k), np.float64) if want_rows else None
        feat = np.empty(n, np.int64)
        fo = np.empty(n, np.int32)
        _ck(self.lib, self.lib.ctr_fm_model_copy(self.h, rows.ctypes.data if want_rows else None, feat.ctypes.data, fo.ctypes.data))
        return rows, feat, fo

    def close(self):
        if getattr(self, 'h', None):
            self.lib.ctr_fm_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_lines(path, threads=None):
    """(lines, non-blank lines) -- `file_len` of python/dl_utils.py:111-115 counts the former."""
    lib = _capi.load()
    nl, ne = C.c_int64(), C.c_int64()
    _ck(lib, lib.ctr_count_lines(os.fsencode(path), threads or n_threads(), C.byref(nl), C.byref(ne)))
    return nl.value, ne.value


def parse_examples(path, mode, model=None, width=16, threads=None, want_shadowed=False):
    """Whole file -> (ids int32 [N, width], vals int32 [N, width] or None, y int32 [N]); want_shadowed (MODE_FNN): also the
    features a later feature of their field replaced in `ids`, int32 [n, 3] = (example, field, row) in file order
    (ctr_parse_examples_ex; the reference's update loop still visits them, python/FNN_wnzh.py:300-306)."""
    lib = _capi.load()
    threads = threads or n_threads()
    _, n = count_lines(path, threads)
    ids = np.empty((n, width), np.int32)
    vals = np.empty((n, width), np.int32) if mode == MODE_PAIRS else None
    y = np.empty(n, np.int32)
    got, nsh = C.c_int64(), C.c_int64()
    cap = 1024 if want_shadowed else 0
    while True:
        sh = np.empty((cap, 3), np.int32)
        rc = lib.ctr_parse_examples_ex(os.fsencode(path), mode, model.h if model is not None else None, width, threads, n,
                                       ids.ctypes.data, vals.ctypes.data if vals is not None else None, y.ctypes.data, C.byref(got),
                                       cap, sh.ctypes.data if want_shadowed else None, C.byref(nsh))
        if want_shadowed and rc == _capi_err_cap() and nsh.value > cap:        # rare: lines with several features per field
            cap = int(nsh.value)
            continue
        _ck(lib, rc)
        break
    assert got.value == n
    if want_shadowed:
        return ids, vals, y, sh[:nsh.value].copy()
    return ids, vals, y


# ---------------------------------------------------------------------------------------------- binary id cache
# "int32 [N,16] + uint8 y[N] binary cache".  The parsed form of an example file -- row indices of
# the FM model, labels, shadowed features -- written once beside a key that names what it was parsed from: the text file's
# size and mtime, the parse mode and width, and a digest of the model's (feature id, field) table in row order (the ids ARE row
# indices of that table).  A cache whose key differs is ignored and rewritten.  Opt-in: `cache_dir=` or the environment
# variable FNN_IDS_CACHE (a directory); the reference writes no such files, so nothing is written by default.
_CACHE_MAGIC = b'CTRIDS02'


def model_digest(model):
    import hashlib
    if model is None:
        return '-'
    _, feat, fo = model.arrays(want_rows=False)
    h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(feat, np.int64).tobytes())
    h.update(np.ascontiguousarray(fo, np.int32).tobytes())
    return h.hexdigest()


def _cache_key(path, mode, model, width, digest=None):
    st = os.stat(path)
    return 'size=%d mtime_ns=%d mode=%d width=%d model=%s' % (st.st_size, st.st_mtime_ns, mode, width, digest or model_digest(model))


def cache_file(path, cache_dir, mode=MODE_FNN):
    import hashlib
    tag = hashlib.blake2b(os.path.abspath(path).encode(), digest_size=6).hexdigest()
    return os.path.join(cache_dir, '%s.%s.m%d.ids' % (os.path.basename(path), tag, mode))


def write_ids_cache(cpath, key, ids, vals, y, shadowed):
    """header: magic, key length + key, n, width, n_shadowed, flags; then ids int32 [n,width], (vals int32 [n,width]), y uint8 [n]
    when every label fits a byte else int32 [n], shadowed int32 [ns,3].  Written to a temporary name and renamed."""
    n, width = ids.shape
    y = np.ascontiguousarray(y, np.int32)
    y8 = bool(n == 0 or (y.min() >= 0 and y.max() <= 255))
    sh = np.zeros((0, 3), np.int32) if shadowed is None else np.ascontiguousarray(shadowed, np.int32)
    kb = key.encode()
    tmp = '%s.tmp%d' % (cpath, os.getpid())
    with open(tmp, 'wb') as f:
        f.write(_CACHE_MAGIC)
        np.array([len(kb), n, width, len(sh), (1 if y8 else 0) | (2 if vals is not None else 0) | (4 if shadowed is not None else 0)], np.int64).tofile(f)
        f.write(kb)
        f.write(b'\0' * (-f.tell() % 64))
        np.ascontiguousarray(ids, np.int32).tofile(f)
        if vals is not None:
            np.ascontiguousarray(vals, np.int32).tofile(f)
        (y.astype(np.uint8) if y8 else y).tofile(f)
        f.write(b'\0' * (-f.tell() % 4))
        sh.tofile(f)
    os.replace(tmp, cpath)


def read_ids_cache(cpath, key, want_shadowed=False):
    """(ids, vals, y[, shadowed]) or None when the file is missing, was written for another key, lacks the shadow list that is
    asked for, or is cut short."""
    try:
        with open(cpath, 'rb') as f:
            if f.read(8) != _CACHE_MAGIC:
                return None
            hdr = np.fromfile(f, np.int64, 5)
            if len(hdr) != 5:
                return None
            klen, n, width, ns, flags = (int(v) for v in hdr)
            if klen != len(key.encode()) or f.read(klen).decode(errors='replace') != key or (want_shadowed and not flags & 4):
                return None
            f.seek(-f.tell() % 64, 1)
            ids = np.fromfile(f, np.int32, n * width)
            vals = np.fromfile(f, np.int32, n * width) if flags & 2 else None
            y = np.fromfile(f, np.uint8 if flags & 1 else np.int32, n)
            f.seek(-f.tell() % 4, 1)
            sh = np.fromfile(f, np.int32, ns * 3)
            if len(ids) != n * width or len(y) != n or len(sh) != ns * 3 or (vals is not None and len(vals) != n * width):
                return None
    except (IOError, OSError):
        return None
    out = (ids.reshape(n, width), None if vals is None else vals.reshape(n, width), y.astype(np.int32))
    return out + (sh.reshape(ns, 3),) if want_shadowed else out


def parse_examples_cached(path, mode, model=None, width=16, threads=None, want_shadowed=False, cache_dir=None, digest=None):
    """parse_examples behind the binary cache; cache_dir None -> $FNN_IDS_CACHE -> no cache at all."""
    cache_dir = cache_dir or os.environ.get('FNN_IDS_CACHE')
    if not cache_dir:
        return parse_examples(path, mode, model, width, threads, want_shadowed)
    os.makedirs(cache_dir, exist_ok=True)
    key, cpath = _cache_key(path, mode, model, width, digest), cache_file(path, cache_dir, mode)
    got = read_ids_cache(cpath, key, want_shadowed)
    if got is not None:
        return got
    res = parse_examples(path, mode, model, width, threads, want_shadowed=(mode == MODE_FNN))    # the cache always carries the shadow list
    write_ids_cache(cpath, key, res[0], res[1], res[2], res[3] if mode == MODE_FNN else None)
    return (res if want_shadowed else res[:3]) if mode == MODE_FNN else res


def _capi_err_cap():
    return -5          # CTR_ERR_CAP (include/ctr_ingest.h)


def yzx_stat(path, threads=None):
    """python/ipinyou.py:23-39 `stat`: (max_dim, max_fea), plus the line count."""
    lib = _capi.load()
    n, md, mf = C.c_int64(), C.c_int64(), C.c_int64()
    _ck(lib, lib.ctr_yzx_stat(os.fsencode(path), threads or n_threads(), C.byref(n), C.byref(md), C.byref(mf)))
    return md.value, mf.value, n.value


def parse_yzx(path, max_dim, max_fea, threads=None):
    """python/ipinyou.py:42-65 for the whole file, in file order: X_ind, X_val [n, max_fea] int64, y [n]."""
    lib = _capi.load()
    threads = threads or n_threads()
    n, _ = count_lines(path, threads)
    X_ind = np.empty((n, max_fea), np.int64)
    X_val = np.empty((n, max_fea), np.int64)
    y = np.empty(n, np.int64)
    got = C.c_int64()
    _ck(lib, lib.ctr_parse_yzx(os.fsencode(path), threads, n, max_dim, max_fea, X_ind.ctypes.data, X_val.ctypes.data,
                               y.ctypes.data, C.byref(got)))
    return X_ind[:got.value], X_val[:got.value], y[:got.value]
