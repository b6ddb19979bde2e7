"""Loader interface #1: same class / method names as the reference's python/data_fm.py
(`DataFM`, `feat_layer_one_index`, `feats_to_layer_one_array`, `get_batch_data`, `get_xy_fm`,
`get_fxy_fm`), re-built around a dense row table so that the layer-one array `x` becomes a
device-side product of the HIP gather instead of 16 dict lookups per example.

Host logic only (text parsing, id mapping).  `get_batch_data` needs an attached FNNEngine for
`x` (fnn_gather); the id fast path (`get_batch_ids`, `load_ids`) is what the training loop uses.
"""
import linecache
import os

import numpy


class DataFM(object):
    # python/data_fm.py:17-18
    name_field = {'weekday': 0, 'hour': 1, 'useragent': 2, 'IP': 3, 'region': 4, 'city': 5,
                  'adexchange': 6, 'domain': 7, 'slotid': 8, 'slotwidth': 9, 'slotheight': 10,
                  'slotvisibility': 11, 'slotformat': 12, 'creative': 13, 'advertiser': 14,
                  'slotprice': 15}

    def __init__(self, fm_model_file, engine=None):
        """Parses `fm.model.txt` (python/data_fm.py:15-44): line 1 `w_0 feat_num rank`, then
        `feat w v_1..v_rank <fieldname>:<rest>`.  Unknown field name -> KeyError, as there.  The
        parse is the native multi-threaded one (include/ctr_ingest.h); the reference's dict
        attributes `feat_field` / `feat_weights` are built from its arrays on first use."""
        from . import ingest
        self.fm_model_file = fm_model_file
        self.engine = engine
        names = sorted(self.name_field, key=self.name_field.get)
        self.model = ingest.FMModel.load(fm_model_file, names)
        self.w_0 = self.model.w0
        self.k = self.model.k                                  # w and v
        self.xdim = 1 + len(self.name_field) * self.k
        self.rows, self.feat_ids, self.field_of_row = self.model.arrays()
        self._dicts = None
        self._digest = None                                    # of the feature table, for the binary id cache

    def _build_dicts(self):
        if self._dicts is None:
            feats = self.feat_ids.tolist()
            self._dicts = (dict(zip(feats, self.field_of_row.tolist())), dict(zip(feats, self.rows.tolist())),
                           dict(zip(feats, range(len(feats)))))
        return self._dicts

    @property
    def feat_field(self):                                     # python/data_fm.py:20
        return self._build_dicts()[0]

    @property
    def feat_weights(self):                                   # python/data_fm.py:21 (parse-time weights)
        return self._build_dicts()[1]

    @property
    def feat_row(self):
        """feat id -> row index of the dense table (file order; a repeated id keeps its first row)."""
        return self._build_dicts()[2]

    # ------------------------------------------------------------------ reference API
    def feat_layer_one_index(self, feat, l):                  # python/data_fm.py:46-47
        return 1 + self.feat_field[feat] * self.k + l

    def feats_to_layer_one_array(self, feats):
        """python/data_fm.py:49-54, single example, host utility (not the hot path: batches go
        through fnn_gather).  Reads the parse-time weights, like the reference's class does."""
        x = numpy.zeros(self.xdim)
        x[0] = self.w_0
        for feat in feats:
            lo = self.feat_layer_one_index(feat, 0)
            x[lo:lo + self.k] = self.feat_weights[feat]
        return x

    def get_xy_fm(self, line):                                # python/data_fm.py:72-77
        feats, y = self._parse(line)
        return self.feats_to_layer_one_array(feats), y

    def get_fxy_fm(self, line):                               # python/data_fm.py:79-84
        feats, y = self._parse(line)
        return feats, self.feats_to_layer_one_array(feats), y

    def get_batch_data(self, file, index, size):              # 1,5 -> lines 1,2,3,4,5
        """python/data_fm.py:57-70: (farray list of feature-id lists, xarray [b,xdim] float32,
        yarray [b] int32); blank lines skipped.  x is produced on the GPU by fnn_gather from the
        engine's CURRENT table (the reference's script form re-reads its updated feat_weights too,
        python/FNN_wnzh.py:95)."""
        farray, ids, yarray = self.get_batch_ids(file, index, size)
        if self.engine is None:
            raise RuntimeError("DataFM.get_batch_data: attach an FNNEngine (engine=...): the layer-one "
                               "array is a device-side product of fnn_gather; there is no CPU path")
        if len(farray) == 0:
            return farray, numpy.zeros((0, self.xdim), dtype=numpy.float32), yarray
        xarray = self.engine.gather(ids).cpu().numpy()
        return farray, xarray, yarray

    # ------------------------------------------------------------------ id fast path
    @staticmethod
    def _parse(line):
        s = line.replace(':', ' ').split()
        return [int(s[j]) for j in range(1, len(s), 2)], int(s[0])

    def feats_to_ids(self, feats):
        """feature ids of one example -> int32 [16] row indices, slot = field, -1 = empty field;
        a later feature of the same field overwrites an earlier one (python/data_fm.py:52-53).
        Unknown feature -> KeyError (python/FNN_wnzh.py:95)."""
        out = numpy.full(len(self.name_field), -1, dtype=numpy.int32)
        for feat in feats:
            out[self.feat_field[feat]] = self.feat_row[feat]
        return out

    def get_batch_ids(self, file, index, size):
        farray, ids, ys = [], [], []
        for i in range(index, index + size):
            line = linecache.getline(file, i)
            if line.strip() != '':
                feats, y = self._parse(line.strip())
                farray.append(feats)
                ids.append(self.feats_to_ids(feats))
                ys.append(y)
        ids = numpy.asarray(ids, dtype=numpy.int32).reshape(len(ys), len(self.name_field))
        return farray, ids, numpy.asarray(ys, dtype=numpy.int32)

    def load_ids(self, file, want_shadowed=False, cache_dir=None):
        """Whole file -> (ids int32 [N,16], y int32 [N]); blank lines skipped.  One native pass
        (ctr_parse_examples, CTR_MODE_FNN) instead of get_fxy per line per epoch.  want_shadowed: also int32 [n, 3] =
        (example, field, row) of the features a later feature of the same field shadows (see shadowed_of).
        cache_dir (or $FNN_IDS_CACHE): keep the parsed arrays as a binary file there, keyed by the text file's size / mtime and
        this model's feature table (ingest.parse_examples_cached); a later run reads that instead of parsing."""
        from . import ingest
        if self._digest is None and (cache_dir or os.environ.get('FNN_IDS_CACHE')):
            self._digest = ingest.model_digest(self.model)
        res = ingest.parse_examples_cached(file, ingest.MODE_FNN, self.model, len(self.name_field), want_shadowed=want_shadowed,
                                           cache_dir=cache_dir, digest=self._digest)
        return (res[0], res[2], res[3]) if want_shadowed else (res[0], res[2])

    def shadowed_of(self, feats_per_example):
        """(example, field, row) int32 [n, 3] of every feature of the given lines (lists of feature ids, as get_batch_ids
        returns them) that a later feature of the same field overwrites in feats_to_ids: the reference's update loop visits
        them all the same (python/FNN_wnzh.py:300-306) -- FNNEngine.set_shadowed carries them into the next train step."""
        out = []
        for t, feats in enumerate(feats_per_example):
            seen = {}
            for feat in feats:
                fld = self.feat_field[feat]
                if fld in seen:
                    out.append((t, fld, seen[fld]))
                seen[fld] = self.feat_row[feat]
        return numpy.asarray(out, dtype=numpy.int32).reshape(len(out), 3)

    def table(self):
        """(rows float32 [D,K], field_of_row int32 [D], w_0) for FNNEngine.set_table."""
        return self.rows.astype(numpy.float32), self.field_of_row, self.w_0

    def write_fm_model(self, path, rows=None):
        """Checkpoint the FM rows in the `fm.model.txt` text format (python/FNN_wnzh.py:68-84 reads it;
        the reference never saves the rows its sparse update changes).  `rows` [D, K]: default = the
        attached engine's current table (fnn_get_table), else the parse-time rows.  repr() keeps the
        float32 values exact, so DataFM(path) reproduces them bit for bit."""
        if rows is None:
            rows = self.engine.get_table() if self.engine is not None else self.rows
        names = sorted(self.name_field, key=self.name_field.get)
        with open(path, 'w') as f:
            f.write('%r %d %d\n' % (float(self.w_0), len(rows), self.k - 1))
            for i in range(len(rows)):
                f.write('%d %s %s:%d\n' % (self.feat_ids[i], ' '.join(repr(float(v)) for v in rows[i]),
                                           names[int(self.field_of_row[i])], self.feat_ids[i]))
