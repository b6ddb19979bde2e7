"""Property test of the native line parser (ctr_parse_examples, CTR_MODE_FNN / CTR_MODE_SNN_ACTIVE / CTR_MODE_PAIRS) against the
reference's Python expressions restated in oracle/ingest_oracle.py, on RANDOM files: labels and feature tokens drawn from valid,
unknown and malformed spellings, separators from blanks / tabs / runs / ':' in odd places, every line terminator, blank lines
anywhere, any thread count.  Either both sides return the same arrays, or both raise -- the native side with the reference's
exception type.  (Spellings the Python 3 oracle and the Python 2 reference disagree on -- '1_0', non-ASCII digits -- and ids of
19+ digits, which the native parser reports as malformed, are not generated.)  The draws are derandomised: every run checks the
same files (a one-off run of 3,000 random files per test found nothing beyond what is fixed here).  CPU only."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ingest_oracle as io

import deep_ctr_amd  # noqa: F401
from deep_ctr_amd import ingest

FEATS = [10, 11, 12, 13, 14, 1234567, 3]
FIELD = {10: 4, 11: 1, 12: 3, 13: 3, 14: 15, 1234567: 0, 3: 7}
ROW = {f: i for i, f in enumerate(FEATS)}

label = st.sampled_from(['0', '1', '1', '0', '-1', '+1', '7', 'x', '1.0', ''])
feat_tok = st.one_of(st.sampled_from([str(f) for f in FEATS] * 3 + ['99', '+10', '011', 'ab', '1.5', '', '-3', '12a']))
val_tok = st.sampled_from(['1', '1', '1', '0', '5', '+1', 'q', ''])
sep = st.sampled_from([' ', ' ', ' ', '\t', '  ', ' \t '])
pair = st.builds(lambda f, c, v: f + c + v, feat_tok, st.sampled_from([':', ':', ':', ' ', '::']), val_tok)


@st.composite
def line(draw):
    if draw(st.integers(0, 9)) == 0:
        return draw(st.sampled_from(['', ' ', '\t', '   ']))
    toks = [draw(label)] + draw(st.lists(pair, min_size=0, max_size=6))
    out = toks[0]
    for t in toks[1:]:
        out += draw(sep) + t
    return draw(st.sampled_from(['', '', ' ', '\t'])) + out + draw(st.sampled_from(['', '', ' ', ' \t']))


files = st.builds(lambda ls, terms, last: ''.join(l + t for l, t in zip(ls, terms)) + last,
                  st.lists(line(), min_size=0, max_size=12),
                  st.lists(st.sampled_from(['\n', '\n', '\n', '\r\n', '\r']), min_size=12, max_size=12),
                  st.sampled_from(['', '1 10:1', '0 12:1 13:1']))


def outcome(fn):
    try:
        return ('ok', fn())
    except (ValueError, KeyError, IndexError) as e:
        return ('err', type(e))


@pytest.fixture(scope='module')
def model(built):
    return ingest.FMModel.from_arrays(np.array(FEATS, np.int64), np.array([FIELD[f] for f in FEATS], np.int32), 3, 16)


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=files, threads=st.sampled_from([1, 2, 5]))
def test_random_files_fnn_mode(model, tmp_path, text, threads):
    p = tmp_path / 'f.txt'
    p.write_bytes(text.encode())
    want = outcome(lambda: io.fnn_examples(str(p), FIELD, ROW))
    got = outcome(lambda: ingest.parse_examples(str(p), ingest.MODE_FNN, model, 16, threads=threads))
    assert got[0] == want[0], (text, got, want)
    if want[0] == 'ok':
        assert np.array_equal(got[1][0], want[1][0]) and np.array_equal(got[1][2], want[1][1]), text
    else:
        assert got[1] is want[1], (text, got, want)


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=files, threads=st.sampled_from([1, 3]))
def test_random_files_snn_and_pair_modes(built, tmp_path, text, threads):
    p = tmp_path / 'f.txt'
    p.write_bytes(text.encode())
    for mode, ref in ((ingest.MODE_SNN_ACTIVE, lambda: io.snn_active(str(p), 8)), (ingest.MODE_PAIRS, lambda: io.pairs(str(p), 8))):
        want = outcome(ref)
        got = outcome(lambda: ingest.parse_examples(str(p), mode, None, 8, threads=threads))
        assert got[0] == want[0], (mode, text, got, want)
        if want[0] == 'ok':
            assert np.array_equal(got[1][0], want[1][0]) and np.array_equal(got[1][2], want[1][-1]), (mode, text)
            if mode == ingest.MODE_PAIRS:
                assert np.array_equal(got[1][1], want[1][1]), text
        else:
            assert got[1] is want[1], (mode, text, got, want)


# ---------------------------------------------------------------------------------------------- the FM-model file (A1)
NAMES = sorted(io.NAME_FIELD, key=io.NAME_FIELD.get)
wtok = st.sampled_from(['0.1', '-0.25', '3e-1', '.5', '5.', '+0.5', '1e400', '-1e-400', 'inf', '-inf', 'nan', 'Infinity', '7', 'abc', '1.2.3', '', '0x10'])
tagtok = st.sampled_from(['weekday:1', 'hour:x:y', 'IP:9', 'slotprice:', 'colour:3', 'region', ':5', 'city:0'])


@st.composite
def model_line(draw, k):
    if draw(st.integers(0, 11)) == 0:
        return draw(st.sampled_from(['', '  ', '\t']))
    toks = [draw(st.sampled_from(['10', '11', '12', '10', '+13', '014', 'z', '1.0', '-2']))]
    toks += draw(st.lists(wtok, min_size=max(0, k - 1), max_size=k + 1))
    if draw(st.integers(0, 7)) != 0:
        toks.append(draw(tagtok))
    if draw(st.integers(0, 5)) == 0:
        toks.append('extra')
    return draw(st.sampled_from(['', ' '])) + draw(st.sampled_from([' ', '\t', '  '])).join(toks)


@st.composite
def model_file(draw):
    rank = draw(st.integers(0, 3))
    head = draw(st.sampled_from(['-2.5 6 %d' % rank, '0 0 %d junk' % rank, '1e-3\t9\t%d' % rank]))
    lines = draw(st.lists(model_line(rank + 1), min_size=0, max_size=8))
    terms = draw(st.lists(st.sampled_from(['\n', '\n', '\r\n', '\r']), min_size=9, max_size=9))
    return head + terms[0] + ''.join(l + t for l, t in zip(lines, terms[1:])) + draw(st.sampled_from(['', '12 ' + ' '.join(['1'] * (rank + 1)) + ' IP:1']))


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=model_file(), threads=st.sampled_from([1, 3]))
def test_random_fm_model_files(built, tmp_path, text, threads):
    p = tmp_path / 'm.txt'
    p.write_bytes(text.encode())
    want = outcome(lambda: io.parse_fm_model(str(p)))

    def native():
        m = ingest.FMModel.load(str(p), NAMES, threads=threads)
        return (m.w0, m.k) + m.arrays()
    got = outcome(native)
    assert got[0] == want[0], (text, got, want)
    if want[0] == 'err':
        assert got[1] is want[1], (text, got, want)
        return
    w0, k, fw, ff = want[1]
    gw0, gk, rows, feat, fo = got[1]
    assert (gw0, gk) == (w0, k) and feat.tolist() == list(fw) and fo.tolist() == [ff[f] for f in fw], text
    assert np.array_equal(rows, np.array([fw[f] for f in fw], np.float64).reshape(len(fw), k), equal_nan=True), text


# ---------------------------------------------------------------------------------------------- the yzx readers (A12)
ytok = st.sampled_from(['5:1', '12:1', '7:0', '3', '40:1:2', '+8:1', 'a:1', ':1', '', '009:1'])


@st.composite
def yzx_file(draw):
    lines = []
    for _ in range(draw(st.integers(0, 8))):
        toks = [draw(st.sampled_from(['0', '1', '1', '-1', 'y'])), draw(st.sampled_from(['0', '300', 'z']))] + draw(st.lists(ytok, min_size=0, max_size=5))
        lines.append(draw(st.sampled_from(['', ' '])) + draw(st.sampled_from([' ', '\t', '  '])).join(toks[:draw(st.integers(1, len(toks)))]))
    terms = draw(st.lists(st.sampled_from(['\n', '\n', '\r\n']), min_size=8, max_size=8))
    return ''.join(l + t for l, t in zip(lines, terms))


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=yzx_file(), threads=st.sampled_from([1, 3]))
def test_random_yzx_files(built, tmp_path, text, threads):
    """python/ipinyou.py:23-65: `fields = line.strip().split()`, y = int(fields[0]), ids = int(tok.split(':')[0]) of fields[2:]; a
    line with no feature makes stat's max() raise ValueError; a blank line is an IndexError (fields[0])."""
    p = tmp_path / 'y.txt'
    p.write_bytes(text.encode())
    want = outcome(lambda: io.yzx_stat(str(p)))
    got = outcome(lambda: ingest.yzx_stat(str(p), threads)[:2])
    assert got[0] == want[0], (text, got, want)
    if want[0] == 'err':
        assert got[1] is want[1], (text, got, want)
        return
    assert tuple(got[1]) == tuple(want[1]), text
    md, mf = want[1]
    rX, rV, rY = io.yzx_load(str(p), md + 1, mf + 1)
    X, V, Y = ingest.parse_yzx(str(p), md + 1, mf + 1, threads)
    assert np.array_equal(X, rX.reshape(X.shape)) and np.array_equal(V, rV.reshape(V.shape)) and np.array_equal(Y, rY), text
