"""Native text ingestion (include/ctr_ingest.h) against oracle/ingest_oracle.py -- the reference's
Python parsers restated -- on the committed demo files and on hand-made edge cases: blank and
whitespace-only lines, \\r\\n and lone \\r terminators, no final newline, repeated fields, repeated
model ids, unknown features / field names, malformed tokens, and multi-threaded cuts (every thread
count must give the same arrays and the same first error line).  Integer work: bit-exact."""
import os

import numpy as np
import pytest

from oracle import ingest_oracle as io

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import ingest
from deep_ctr_amd.data_fm import DataFM

NAMES = sorted(io.NAME_FIELD, key=io.NAME_FIELD.get)


def _maps(path):
    w0, k, fw, ff = io.parse_fm_model(path)
    feat_row = {f: i for i, f in enumerate(fw)}
    return w0, k, fw, ff, feat_row


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_demo_model_and_files_equal_the_python_parsers(built, golden_dir, threads):
    demo = os.path.join(golden_dir, 'demo')
    mpath = os.path.join(demo, 'fm.model.txt')
    w0, k, fw, ff, feat_row = _maps(mpath)
    m = ingest.FMModel.load(mpath, NAMES, threads=threads)
    rows, feat, fo = m.arrays()
    assert m.w0 == w0 and m.k == k and m.n_rows == len(fw)
    assert feat.tolist() == list(fw)
    assert np.array_equal(rows, np.array([fw[f] for f in fw]))          # float(): bit-exact
    assert fo.tolist() == [ff[f] for f in fw]
    for name in ('train.fm.txt', 'test.fm.txt'):
        path = os.path.join(demo, name)
        ids, _, y = ingest.parse_examples(path, ingest.MODE_FNN, m, 16, threads=threads)
        rid, ry = io.fnn_examples(path, ff, feat_row)
        assert np.array_equal(ids, rid) and np.array_equal(y, ry)
        assert ingest.count_lines(path, threads) == (len(ry), len(ry))
        a, _, ya = ingest.parse_examples(path, ingest.MODE_SNN_ACTIVE, None, 16, threads=threads)
        ra, rya = io.snn_active(path)
        assert np.array_equal(a, ra) and np.array_equal(ya, rya)
        pi, pv, py_ = ingest.parse_examples(path, ingest.MODE_PAIRS, None, 16, threads=threads)
        ri, rv, ryp = io.pairs(path)
        assert np.array_equal(pi, ri) and np.array_equal(pv, rv) and np.array_equal(py_, ryp)
    ypath = os.path.join(demo, 'train.yzx.txt')
    md, mf, n = ingest.yzx_stat(ypath, threads)
    assert (md, mf) == io.yzx_stat(ypath)
    X_ind, X_val, y = ingest.parse_yzx(ypath, md, mf, threads)
    rX, rV, rY = io.yzx_load(ypath, md, mf)
    assert len(y) == n and np.array_equal(X_ind, rX) and np.array_equal(X_val, rV) and np.array_equal(y, rY)


MODEL = ("-2.5 6 2\n"
         "10 0.1 -0.2 3e-1 weekday:1\n"
         "\n"
         "11  +0.5\t1.0 2.0   hour:x:y\n"
         "12 1 2 3 IP:9\r\n"
         "   \n"
         "13 .5 5. -0 IP:10\r"
         "10 9 8 7 region:again\n"                 # repeated id: later line wins, first row kept
         "14 inf nan 1e400 slotprice:1")           # no final newline


def test_edge_cases_of_the_model_and_example_parsers(built, tmp_path):
    mp = tmp_path / 'fm.model.txt'
    mp.write_bytes(MODEL.encode())
    w0, k, fw, ff, feat_row = _maps(str(mp))
    for threads in (1, 4):
        m = ingest.FMModel.load(str(mp), NAMES, threads=threads)
        rows, feat, fo = m.arrays()
        assert feat.tolist() == [10, 11, 12, 13, 14] == list(fw)
        assert fo.tolist() == [4, 1, 3, 3, 15] == [ff[f] for f in fw]
        ref = np.array([fw[f] for f in fw])
        assert np.array_equal(rows, ref, equal_nan=True) and m.w0 == -2.5 and m.k == 3
    ex = ("1 10:1 11:1 12:1\n"
          "\n"
          "0 13:1 12:1\r\n"                         # two features of field IP: the later one wins
          " \t \n"
          "1\t14:0   11:5\r"                        # tabs and runs of spaces (FNN reader only)
          "0 10:1")                                 # no final newline
    ep = tmp_path / 'train.fm.txt'
    ep.write_bytes(ex.encode())
    ids, _, y = ingest.parse_examples(str(ep), ingest.MODE_FNN, m, 16, threads=1)
    rid, ry = io.fnn_examples(str(ep), ff, feat_row)
    assert np.array_equal(ids, rid) and np.array_equal(y, ry) and len(y) == 4
    assert ids[1, 3] == feat_row[12] and ingest.count_lines(str(ep)) == (6, 4)
    # what "the later one wins" drops is reported for the update loop, which visits every feature (python/FNN_wnzh.py:300-306)
    for threads in (1, 3):
        ids2, _, y2, sh = ingest.parse_examples(str(ep), ingest.MODE_FNN, m, 16, threads=threads, want_shadowed=True)
        assert np.array_equal(ids2, rid) and np.array_equal(y2, ry)
        assert sh.tolist() == [[1, 3, feat_row[13]]]
    # the SNN readers split on single spaces: line 5 is a ValueError there, as in the reference
    with pytest.raises(ValueError, match=r'train\.fm\.txt:5'):
        ingest.parse_examples(str(ep), ingest.MODE_SNN_ACTIVE, None, 16)
    with pytest.raises(ValueError):
        io.snn_active(str(ep))
    sp = tmp_path / 'snn.txt'
    sp.write_bytes(b"1 5:1 6:0 7:1\n0 8:2\n\n1 9:1 3:1 4:1\n")
    a, _, ya = ingest.parse_examples(str(sp), ingest.MODE_SNN_ACTIVE, None, 4)
    ra, rya = io.snn_active(str(sp), 4)
    assert np.array_equal(a, ra) and np.array_equal(ya, rya) and a[0].tolist() == [5, 7, -1, -1]
    pi, pv, _ = ingest.parse_examples(str(sp), ingest.MODE_PAIRS, None, 4)
    ri, rv, _ = io.pairs(str(sp), 4)
    assert np.array_equal(pi, ri) and np.array_equal(pv, rv)
    with pytest.raises(IndexError):                         # more features than the row is wide
        ingest.parse_examples(str(sp), ingest.MODE_PAIRS, None, 2)


def test_shadowed_features_many_lines_and_the_python_side(built, tmp_path):
    """Lines with two or three features of one field, and one feature twice: the native parser's shadow list equals what
    DataFM.shadowed_of derives from the reference-style feature lists, in file order, for any thread count and any
    initial capacity (the retry after CTR_ERR_CAP)."""
    from deep_ctr_amd.data_fm import DataFM
    mp = tmp_path / 'fm.model.txt'
    feats = list(range(100, 160))
    fld = [i % 16 for i in range(60)]
    with open(mp, 'w') as f:
        f.write('-1.5 60 2\n')
        for ft, fl in zip(feats, fld):
            f.write('%d 0.1 0.2 0.3 %s:%d\n' % (ft, NAMES[fl], ft))
    rng = np.random.RandomState(3)
    lines, ref_lists = [], []
    for t in range(3000):
        ft = list(rng.choice(feats, size=rng.randint(1, 20)))
        lines.append('%d %s' % (t % 2, ' '.join('%d:1' % v for v in ft)))
        ref_lists.append(ft)
    ep = tmp_path / 'train.fm.txt'
    ep.write_text('\n'.join(lines) + '\n')
    data = DataFM(str(mp))
    want = data.shadowed_of(ref_lists)
    assert len(want) > 1024                                   # beyond the first capacity guess
    for threads in (1, 4):
        ids, y, sh = data.load_ids(str(ep), want_shadowed=True) if threads == 4 else \
            (lambda r: (r[0], r[2], r[3]))(ingest.parse_examples(str(ep), ingest.MODE_FNN, data.model, 16, threads=1, want_shadowed=True))
        assert np.array_equal(sh, want)
        assert np.array_equal(ids, np.stack([data.feats_to_ids(ft) for ft in ref_lists]))


def test_errors_keep_the_reference_exception_and_name_the_first_bad_line(built, tmp_path):
    mp = tmp_path / 'fm.model.txt'
    mp.write_bytes(MODEL.encode())
    m = ingest.FMModel.load(str(mp), NAMES)
    body = "".join("1 10:1 12:1\n" for _ in range(3000))
    bad = tmp_path / 'bad.txt'
    bad.write_bytes((body + "0 99:1\n" + body + "x 10:1\n").encode())      # KeyError at 3001, ValueError at 6002
    for threads in (1, 2, 7):
        with pytest.raises(KeyError, match=r'bad\.txt:3001'):
            ingest.parse_examples(str(bad), ingest.MODE_FNN, m, 16, threads=threads)
    bad.write_bytes((body + "1 10:1 1.5:1\n").encode())
    with pytest.raises(ValueError, match=r'bad\.txt:3001'):
        ingest.parse_examples(str(bad), ingest.MODE_FNN, m, 16)
    # every id of the line is read before any is looked up (the list comprehension of python/data_fm.py:72): junk after an
    # unknown id is the ValueError, not the KeyError
    bad.write_bytes(b"1 10:1\n1 99:1 zz:1\n")
    with pytest.raises(ValueError, match=r'bad\.txt:2'):
        ingest.parse_examples(str(bad), ingest.MODE_FNN, m, 16)
    with pytest.raises(ValueError):
        io.fnn_examples(str(bad), *_maps(str(mp))[3:])
    with pytest.raises(IOError):
        ingest.parse_examples(str(tmp_path / 'missing.txt'), ingest.MODE_FNN, m, 16)
    bm = tmp_path / 'badmodel.txt'
    bm.write_bytes(b"0 1 1\n5 0.1 0.2 colour:3\n")
    with pytest.raises(KeyError, match=r'badmodel\.txt:2.*colour'):
        ingest.FMModel.load(str(bm), NAMES)
    with pytest.raises(KeyError):
        io.parse_fm_model(str(bm))
    bm.write_bytes(b"0 1 1\n5 0.1 weekday:3\n")                         # one weight short
    with pytest.raises(ValueError):
        ingest.FMModel.load(str(bm), NAMES)
    empty = tmp_path / 'empty.txt'
    empty.write_bytes(b"")
    ids, _, y = ingest.parse_examples(str(empty), ingest.MODE_FNN, m, 16)
    assert ids.shape == (0, 16) and y.shape == (0,)


def test_large_file_thread_counts_agree(built, tmp_path):
    rng = np.random.RandomState(5)
    n_feat = 5000
    fields = rng.randint(0, 16, n_feat)
    with open(tmp_path / 'fm.model.txt', 'w') as f:
        f.write("0.5 %d 3\n" % n_feat)
        for i in range(n_feat):
            f.write("%d %s %s:%d\n" % (1000 + 7 * i, " ".join(repr(float(v)) for v in rng.standard_normal(4)), NAMES[fields[i]], i))
    with open(tmp_path / 'train.fm.txt', 'w') as f:
        for t in range(20000):
            feats = rng.randint(0, n_feat, rng.randint(1, 17))
            f.write("%d %s\n" % (rng.randint(0, 2), " ".join("%d:1" % (1000 + 7 * j) for j in feats)))
            if t % 997 == 0:
                f.write("\n")
    d = DataFM(str(tmp_path / 'fm.model.txt'))
    w0, k, fw, ff, feat_row = _maps(str(tmp_path / 'fm.model.txt'))
    assert d.feat_field == ff and d.feat_row == feat_row and d.feat_weights == fw and d.w_0 == w0
    rid, ry = io.fnn_examples(str(tmp_path / 'train.fm.txt'), ff, feat_row)
    for threads in (1, 2, 5, 16):
        ids, _, y = ingest.parse_examples(str(tmp_path / 'train.fm.txt'), ingest.MODE_FNN, d.model, 16, threads=threads)
        assert np.array_equal(ids, rid) and np.array_equal(y, ry)
    ids, y = d.load_ids(str(tmp_path / 'train.fm.txt'))
    assert np.array_equal(ids, rid)
    # the per-batch reference API reads the same lines (linecache, 1-based, blank lines skipped)
    f_arr, bid, by = d.get_batch_ids(str(tmp_path / 'train.fm.txt'), 1, 50)
    assert np.array_equal(bid, rid[:len(by)]) and np.array_equal(by, ry[:len(by)])


def test_fm_model_checkpoint_round_trip(built, golden_dir, tmp_path):
    """DataFM.write_fm_model -> DataFM: same ids, fields, w_0 and (float32) rows, bit for bit."""
    d = DataFM(os.path.join(golden_dir, 'demo', 'fm.model.txt'))
    rows = (d.rows * 1.5 + 0.25).astype(np.float32)              # stand-in for rows the sparse update changed
    path = str(tmp_path / 'ckpt.fm.model.txt')
    d.write_fm_model(path, rows)
    e = DataFM(path)
    assert e.w_0 == d.w_0 and e.k == d.k and np.array_equal(e.feat_ids, d.feat_ids) and np.array_equal(e.field_of_row, d.field_of_row)
    assert np.array_equal(e.rows.astype(np.float32), rows)


def test_sparse_and_dense_id_sets_and_odd_integer_spellings(built, tmp_path):
    """The parser keeps a direct table for dense feature ids and an open-addressing one for sparse ids (ids up to 2^62 here):
    both against the Python restatement, with the integer spellings int() accepts beside plain digits ('+7', '007', a
    19-digit id, a negative label)."""
    rng = np.random.RandomState(11)
    n = 5000
    for tag, feats in (('dense', rng.permutation(3 * n)[:n].astype(np.int64)),
                       ('sparse', np.unique(rng.randint(1, 2 ** 62, size=n + 50, dtype=np.int64))[:n])):
        feats = feats.copy(); rng.shuffle(feats)
        feats[0] = 1234567890123456789 if tag == 'sparse' else feats[0]
        fo = rng.randint(0, 16, size=n).astype(np.int32)
        m = ingest.FMModel.from_arrays(feats, fo, 3, 16)
        ff = {int(f): int(q) for f, q in zip(feats, fo)}
        feat_row = {int(f): i for i, f in enumerate(feats)}
        lines = []
        for t in range(4000):
            pick = feats[rng.randint(0, n, size=rng.randint(1, 17))]
            lines.append('%d %s' % (t % 2, ' '.join('%d:1' % v for v in pick)))
        lines.append('-1 +%d:1 00%d:+1' % (feats[1], feats[2]))
        lines.append('+0 %d:1' % feats[0])
        ep = tmp_path / ('%s.fm.txt' % tag)
        ep.write_text('\n'.join(lines) + '\n')
        rid, ry = io.fnn_examples(str(ep), ff, feat_row)
        for threads in (1, 5):
            ids, _, y = ingest.parse_examples(str(ep), ingest.MODE_FNN, m, 16, threads=threads)
            assert np.array_equal(ids, rid) and np.array_equal(y, ry), tag
        assert ry[-2] == -1 and rid[-1, fo[0]] == 0
        ep.write_text('1 %d:1 %d:1\n' % (feats[3], int(feats.max()) + 1))
        with pytest.raises(KeyError):
            ingest.parse_examples(str(ep), ingest.MODE_FNN, m, 16)


def test_binary_id_cache_round_trip_and_invalidation(built, golden_dir, tmp_path, monkeypatch):
    """the parsed ids / labels / shadow list as a binary cache.  A second load reads the cache (the text file
    can even be unreadable garbage of the same size and mtime); touching the text, or loading with another model, re-parses."""
    import shutil
    demo = os.path.join(golden_dir, 'demo')
    src = tmp_path / 'train.fm.txt'
    shutil.copy(os.path.join(demo, 'train.fm.txt'), src)
    cdir = tmp_path / 'cache'
    d = DataFM(os.path.join(demo, 'fm.model.txt'))
    ids0, y0, sh0 = d.load_ids(str(src), want_shadowed=True)
    ids1, y1, sh1 = d.load_ids(str(src), want_shadowed=True, cache_dir=str(cdir))       # parses and writes
    files = os.listdir(cdir)
    assert len(files) == 1 and files[0].startswith('train.fm.txt.')
    assert np.array_equal(ids0, ids1) and np.array_equal(y0, y1) and np.array_equal(sh0, sh1)
    # same size + mtime, different bytes: only a cache hit can return the old arrays
    st = os.stat(src)
    src.write_bytes(b'x' * st.st_size)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))
    ids2, y2 = d.load_ids(str(src), cache_dir=str(cdir))
    assert np.array_equal(ids0, ids2) and np.array_equal(y0, y2) and ids2.dtype == np.int32 and y2.dtype == np.int32
    monkeypatch.setenv('FNN_IDS_CACHE', str(cdir))                                       # the environment form
    ids3, y3, sh3 = d.load_ids(str(src), want_shadowed=True)
    assert np.array_equal(ids0, ids3) and np.array_equal(sh0, sh3)
    monkeypatch.delenv('FNN_IDS_CACHE')
    # a changed text file (new mtime) is parsed again: the garbage now raises
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns + 10 ** 9))
    with pytest.raises(ValueError):
        d.load_ids(str(src), cache_dir=str(cdir))
    # another model (rows in another order): the key differs, the cache is not used
    shutil.copy(os.path.join(demo, 'train.fm.txt'), src)
    ids4, y4 = d.load_ids(str(src), cache_dir=str(cdir))
    rows, feat, fo = d.model.arrays()
    m2 = ingest.FMModel.from_arrays(feat[::-1].copy(), fo[::-1].copy(), d.k, 16)
    r = ingest.parse_examples_cached(str(src), ingest.MODE_FNN, m2, 16, cache_dir=str(cdir))
    assert np.array_equal(r[0][r[0] >= 0], len(feat) - 1 - ids4[ids4 >= 0]) and np.array_equal(ids4, ids0)
    # a truncated cache file is ignored
    cp = os.path.join(cdir, os.listdir(cdir)[0])
    open(cp, 'r+b').truncate(os.path.getsize(cp) // 2)
    ids5, y5 = d.load_ids(str(src), cache_dir=str(cdir))
    assert np.array_equal(ids5, ids0)
    # labels outside a byte are kept as int32; the pair mode carries its values
    sp = tmp_path / 'snn.txt'
    sp.write_bytes(b"300 5:1 6:0 7:1\n-2 8:2\n")
    for _ in range(2):
        pi, pv, py_ = ingest.parse_examples_cached(str(sp), ingest.MODE_PAIRS, None, 4, cache_dir=str(cdir))
        assert py_.tolist() == [300, -2] and pv[0].tolist() == [1, 0, 1, 0] and pi[1].tolist() == [8, -1, -1, -1]
