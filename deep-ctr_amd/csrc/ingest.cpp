// ingest.cpp -- native text ingestion behind include/ctr_ingest.h (host code, no HIP calls).
//
// What it replaces in Atomu2014/deep-ctr: the interpreter loops around the training step --
// python/FNN_wnzh.py:62-84 (FM model), :224-253 (linecache.getline + get_fxy per line, redone for
// every batch, epoch and evaluation pass), python/SNN_RBM.py:238-262, python/ipinyou.py:23-65.
// Design: mmap the file, cut it into one byte range per thread at line boundaries, pass 1 counts
// lines / examples per range (prefix sums give every range its output offset and first line
// number), pass 2 parses straight into the caller's arrays.  Token rules follow the Python
// expressions of the reference exactly (see the mode comments in the header).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ctr_ingest.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

// str.strip() / str.split() whitespace of the reference's Python 2 byte strings
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

struct Mapped {
    const char* p = nullptr; size_t n = 0; int fd = -1; bool owns = true;
    ~Mapped() { if (owns && p && n) munmap(const_cast<char*>(p), n); if (fd >= 0) close(fd); }
    int open_file(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(CTR_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
        struct stat st;
        if (fstat(fd, &st) != 0) return fail(CTR_ERR_IO, std::string("cannot stat ") + path);
        n = (size_t)st.st_size;
        if (n == 0) { p = nullptr; return CTR_OK; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { n = 0; return fail(CTR_ERR_IO, std::string("cannot mmap ") + path + ": " + strerror(errno)); }
        madvise(m, n, MADV_SEQUENTIAL);
        p = static_cast<const char*>(m);
        return CTR_OK;
    }
};

// end of the line that starts at `s`: [s, e) is the text, `next` the start of the following line
inline void line_end(const char* p, size_t n, size_t s, size_t& e, size_t& next) {
    if (s >= n) { e = next = n; return; }
    const char* nl = static_cast<const char*>(memchr(p + s, '\n', n - s));
    const size_t i_nl = nl ? (size_t)(nl - p) : n;
    const char* cr = static_cast<const char*>(memchr(p + s, '\r', i_nl - s));      // a lone '\r' ends a line too
    const size_t i = cr ? (size_t)(cr - p) : i_nl;
    e = i;
    if (i >= n) { next = n; return; }
    next = (p[i] == '\r' && i + 1 < n && p[i + 1] == '\n') ? i + 2 : i + 1;
}
// first line start >= lo (lo > 0): a line starts right after a terminator
inline size_t first_line_start(const char* p, size_t n, size_t lo) {
    if (lo == 0) return 0;
    size_t i = lo - 1;
    while (i < n && p[i] != '\n' && p[i] != '\r') ++i;
    if (i >= n) return n;
    return (p[i] == '\r' && i + 1 < n && p[i + 1] == '\n') ? i + 2 : i + 1;
}
inline bool blank(const char* p, size_t s, size_t e) {
    for (size_t i = s; i < e; ++i) if (!is_ws(p[i])) return false;
    return true;
}

struct Range { size_t lo, hi; int64_t lines = 0, examples = 0, line0 = 0, ex0 = 0; };

std::vector<Range> cut(const Mapped& f, int n_threads) {
    int nt = std::max(1, n_threads);
    if (f.n < (size_t)nt * 4096) nt = 1;
    std::vector<Range> r(nt);
    for (int t = 0; t < nt; ++t) {
        const size_t lo = f.n * (size_t)t / nt, hi = f.n * (size_t)(t + 1) / nt;
        r[t].lo = first_line_start(f.p, f.n, lo);
        r[t].hi = (t + 1 == nt) ? f.n : first_line_start(f.p, f.n, hi);
    }
    return r;
}

template <typename Fn> void run_threads(int n, Fn fn) {
    if (n == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
}

void count_ranges(const Mapped& f, std::vector<Range>& rs) {
    run_threads((int)rs.size(), [&](int t) {
        Range& r = rs[t];
        size_t s = r.lo, e, nx;
        int64_t lines = 0, ex = 0;
        while (s < r.hi) { line_end(f.p, f.n, s, e, nx); ++lines; if (!blank(f.p, s, e)) ++ex; s = nx; }
        r.lines = lines; r.examples = ex;
    });
    int64_t l = 0, x = 0;
    for (auto& r : rs) { r.line0 = l; r.ex0 = x; l += r.lines; x += r.examples; }
}

// Python int(): optional surrounding whitespace, optional sign, decimal digits
inline bool py_int(const char* b, const char* e, int64_t& out) {
    while (b < e && is_ws(*b)) ++b;
    while (e > b && is_ws(e[-1])) --e;
    if (b >= e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') { neg = *b == '-'; ++b; }
    if (b >= e) return false;
    uint64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        if (v > (UINT64_MAX - 9) / 10) return false;
        v = v * 10 + (uint64_t)(*b - '0');
    }
    if (v > (uint64_t)INT64_MAX) return false;
    out = neg ? -(int64_t)v : (int64_t)v;
    return true;
}
inline bool py_float(const char* b, const char* e, double& out) {
    if (b < e && *b == '+') ++b;                           // from_chars takes no leading '+'
    auto r = std::from_chars(b, e, out);
    if (r.ec == std::errc::result_out_of_range && r.ptr == e) {        // float('1e400') = inf, float('1e-400') = 0.0
        const std::string z(b, e);
        out = strtod(z.c_str(), nullptr);
        return true;
    }
    return r.ec == std::errc() && r.ptr == e;
}

struct Tok { const char* b; const char* e; };
struct SepTable { bool t[256]; SepTable() { for (int c = 0; c < 256; ++c) t[c] = is_ws((char)c) || c == ':'; } };

// `.replace(':', ' ').split()` (runs = true) or `.strip().replace(':', ' ').split(' ')` (runs = false)
inline void tokenize(const char* p, size_t s, size_t e, bool runs, std::vector<Tok>& out) {
    out.clear();
    if (runs) {
        static const SepTable sep;
        size_t i = s;
        while (i < e) {
            while (i < e && sep.t[(unsigned char)p[i]]) ++i;
            if (i >= e) break;
            const size_t b = i;
            while (i < e && !sep.t[(unsigned char)p[i]]) ++i;
            out.push_back(Tok{p + b, p + i});
        }
    } else {
        while (s < e && is_ws(p[s])) ++s;
        while (e > s && is_ws(p[e - 1])) --e;
        size_t b = s;
        for (size_t i = s; i <= e; ++i) {
            if (i == e || p[i] == ' ' || p[i] == ':') { out.push_back(Tok{p + b, p + i}); b = i + 1; }
        }
    }
}

struct Err {                                                // first error in line order wins
    std::atomic<int64_t> line{INT64_MAX}; int code = CTR_OK; std::string msg; std::atomic_flag lock = ATOMIC_FLAG_INIT;
    void set(int64_t ln, int c, const std::string& m) {
        while (lock.test_and_set(std::memory_order_acquire)) {}
        if (ln < line.load()) { line = ln; code = c; msg = m; }
        lock.clear(std::memory_order_release);
    }
};

// open-addressing map feat id -> row
struct IdMap {
    std::vector<int64_t> key; std::vector<int32_t> val; uint64_t mask = 0;
    static uint64_t h(int64_t k) { uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull; return x ^ (x >> 29); }
    void init(size_t n) { size_t c = 16; while (c < 2 * n + 2) c <<= 1; key.assign(c, INT64_MIN); val.assign(c, -1); mask = c - 1; }
    int32_t* slot(int64_t k, bool& found) {
        uint64_t i = h(k) & mask;
        while (key[i] != INT64_MIN && key[i] != k) i = (i + 1) & mask;
        found = key[i] == k;
        if (!found) key[i] = k;
        return &val[i];
    }
    int32_t get(int64_t k) const {
        uint64_t i = h(k) & mask;
        while (key[i] != INT64_MIN && key[i] != k) i = (i + 1) & mask;
        return key[i] == k ? val[i] : -1;
    }
};

// What the line parser asks per feature: id -> (row, field), one cache line per question and prefetchable, so that the 16
// questions of a line overlap their misses (the table of a 937,670-row model is several MB: at one dependent miss after another
// the parser spent most of its time waiting for them).  Dense id sets (max id < 4 n + 1024, the usual index files) get a direct
// table, others an open-addressing table of 16-byte entries.  Built once the model is complete; read-only afterwards.
struct Finder {
    struct E { int64_t key; int32_t row; int32_t fld; };
    std::vector<uint64_t> direct;                            // (row + 1) | field << 32; 0 = absent
    std::vector<E> tab; uint64_t mask = 0;
    void build(const std::vector<int64_t>& feat, const std::vector<int32_t>& field) {
        direct.clear(); tab.clear();
        const size_t n = feat.size();
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int64_t v : feat) { lo = std::min(lo, v); hi = std::max(hi, v); }
        if (n && lo >= 0 && (uint64_t)hi < 4 * (uint64_t)n + 1024) {
            direct.assign((size_t)hi + 1, 0);
            for (size_t i = 0; i < n; ++i) direct[(size_t)feat[i]] = (uint64_t)(i + 1) | (uint64_t)(uint32_t)field[i] << 32;
            return;
        }
        size_t c = 16; while (c < 2 * n + 2) c <<= 1;
        tab.assign(c, E{INT64_MIN, -1, -1}); mask = c - 1;
        for (size_t i = 0; i < n; ++i) {
            uint64_t j = IdMap::h(feat[i]) & mask;
            while (tab[j].row >= 0) j = (j + 1) & mask;
            tab[j] = E{feat[i], (int32_t)i, field[i]};
        }
    }
    inline void prefetch(int64_t k) const {
        if (!direct.empty()) { if ((uint64_t)k < direct.size()) __builtin_prefetch(&direct[(size_t)k]); }
        else __builtin_prefetch(&tab[IdMap::h(k) & mask]);
    }
    inline bool find(int64_t k, int32_t& row, int32_t& fld) const {
        if (!direct.empty()) {
            if ((uint64_t)k >= direct.size()) return false;
            const uint64_t v = direct[(size_t)k];
            if (!v) return false;
            row = (int32_t)(uint32_t)v - 1; fld = (int32_t)(v >> 32);
            return true;
        }
        uint64_t j = IdMap::h(k) & mask;
        while (tab[j].row >= 0 && tab[j].key != k) j = (j + 1) & mask;
        if (tab[j].row < 0) return false;
        row = tab[j].row; fld = tab[j].fld;
        return true;
    }
};

}  // namespace

struct ctr_fm_model {
    int k = 0, n_fields = 0; double w0 = 0.0; bool from_file = false;     // from_file: `rows` is meaningful (possibly empty)
    std::vector<int64_t> feat; std::vector<int32_t> field; std::vector<double> rows;
    IdMap map;                                               // while the model is being put together (later lines overwrite)
    Finder finder;                                           // what the line parser reads
};

extern "C" {

const char* ctr_last_error(void) { return g_err.c_str(); }

int ctr_fm_model_load(const char* path, const char* const* field_names, int n_fields, int n_threads, ctr_fm_model** out)
{
    if (!path || !field_names || n_fields < 1 || !out) return fail(CTR_ERR_ARG, "null argument");
    *out = nullptr;
    Mapped f;
    int rc = f.open_file(path);
    if (rc != CTR_OK) return rc;
    // header: `w_0 feat_num rank` (python/FNN_wnzh.py:68-78)
    size_t e0 = 0, body = 0;
    line_end(f.p, f.n, 0, e0, body);
    std::vector<Tok> tk;
    tokenize(f.p, 0, e0, true, tk);
    // (the header is split on whitespace only; a ':' cannot occur in three numbers)
    double w0; int64_t rank;
    if (tk.size() < 3 || !py_float(tk[0].b, tk[0].e, w0) || !py_int(tk[2].b, tk[2].e, rank) || rank < 0 || rank > 1 << 20)
        return fail(CTR_ERR_PARSE, std::string(path) + ":1: header must be `w_0 feat_num rank`");
    const int k = (int)rank + 1;
    std::vector<std::string> names(field_names, field_names + n_fields);

    Mapped g;                                                 // view of the body only
    g.owns = false; g.p = f.p + body; g.n = f.n - body;
    std::vector<Range> rs = cut(g, n_threads);
    count_ranges(g, rs);
    struct Part { std::vector<int64_t> feat; std::vector<int32_t> field; std::vector<double> w; };
    std::vector<Part> parts(rs.size());
    Err err;
    run_threads((int)rs.size(), [&](int t) {
        const Range& r = rs[t];
        Part& pt = parts[t];
        pt.feat.reserve(r.examples); pt.field.reserve(r.examples); pt.w.reserve((size_t)r.examples * k);
        std::vector<Tok> tok;
        size_t s = r.lo, e, nx; int64_t ln = r.line0 + 2;     // +1 header, 1-based
        for (; s < r.hi; s = nx, ++ln) {
            line_end(g.p, g.n, s, e, nx);
            if (blank(g.p, s, e)) continue;
            // `line.strip().split()`: whitespace runs only -- ':' stays inside the tag token
            tok.clear();
            for (size_t i = s; i < e;) {
                while (i < e && is_ws(g.p[i])) ++i;
                if (i >= e) break;
                const size_t b = i;
                while (i < e && !is_ws(g.p[i])) ++i;
                tok.push_back(Tok{g.p + b, g.p + i});
            }
            int64_t feat;                                     // python/data_fm.py:39-43, in that order
            if (!py_int(tok[0].b, tok[0].e, feat)) { err.set(ln, CTR_ERR_PARSE, "feature id is not an int"); return; }
            for (int j = 0; j < k; ++j) {
                double v;
                if (1 + j >= (int)tok.size()) { err.set(ln, CTR_ERR_INDEX, "line ends before its weights (IndexError)"); return; }
                if (!py_float(tok[1 + j].b, tok[1 + j].e, v)) { err.set(ln, CTR_ERR_PARSE, "weight is not a float"); return; }
                pt.w.push_back(v);
            }
            if (1 + k >= (int)tok.size()) { err.set(ln, CTR_ERR_INDEX, "line ends before its field tag (IndexError)"); return; }
            const Tok tag = tok[1 + k];
            const char* colon = static_cast<const char*>(memchr(tag.b, ':', tag.e - tag.b));
            if (!colon) { err.set(ln, CTR_ERR_PARSE, "field tag has no ':' (ValueError: substring not found)"); return; }
            int fld = -1;
            for (int q = 0; q < n_fields; ++q)
                if (names[q].size() == (size_t)(colon - tag.b) && memcmp(names[q].data(), tag.b, colon - tag.b) == 0) { fld = q; break; }
            if (fld < 0) { err.set(ln, CTR_ERR_KEY, "unknown field name '" + std::string(tag.b, colon) + "' (KeyError)"); return; }
            pt.feat.push_back(feat); pt.field.push_back(fld);
        }
    });
    if (err.code != CTR_OK) return fail(err.code, std::string(path) + ":" + std::to_string(err.line.load()) + ": " + err.msg);
    ctr_fm_model* m = new ctr_fm_model();
    m->k = k; m->n_fields = n_fields; m->w0 = w0; m->from_file = true;
    size_t total = 0;
    for (auto& pt : parts) total += pt.feat.size();
    if (total >= (size_t)INT32_MAX) { delete m; return fail(CTR_ERR_CAP, "more than 2^31 rows"); }
    m->map.init(total);
    m->feat.reserve(total); m->field.reserve(total); m->rows.reserve(total * k);
    for (auto& pt : parts) {                                  // file order; a later line of the same id overwrites (dict)
        for (size_t i = 0; i < pt.feat.size(); ++i) {
            bool found;
            int32_t* sl = m->map.slot(pt.feat[i], found);
            if (found) {
                m->field[*sl] = pt.field[i];
                std::copy(pt.w.begin() + i * k, pt.w.begin() + (i + 1) * k, m->rows.begin() + (size_t)*sl * k);
            } else {
                *sl = (int32_t)m->feat.size();
                m->feat.push_back(pt.feat[i]); m->field.push_back(pt.field[i]);
                m->rows.insert(m->rows.end(), pt.w.begin() + i * k, pt.w.begin() + (i + 1) * k);
            }
        }
    }
    m->finder.build(m->feat, m->field);
    *out = m;
    return CTR_OK;
}

int ctr_fm_model_from_arrays(const int64_t* feat_ids, const int32_t* field_of_row, int64_t n_rows, int k, int n_fields,
                             ctr_fm_model** out)
{
    if (!feat_ids || !field_of_row || n_rows < 1 || n_rows >= INT32_MAX || k < 1 || n_fields < 1 || !out) return fail(CTR_ERR_ARG, "bad argument");
    ctr_fm_model* m = new ctr_fm_model();
    m->k = k; m->n_fields = n_fields;
    m->feat.assign(feat_ids, feat_ids + n_rows); m->field.assign(field_of_row, field_of_row + n_rows);
    m->map.init((size_t)n_rows);
    for (int64_t i = 0; i < n_rows; ++i) {
        if (field_of_row[i] < 0 || field_of_row[i] >= n_fields) { delete m; return fail(CTR_ERR_ARG, "field_of_row outside [0, n_fields)"); }
        bool found; int32_t* sl = m->map.slot(feat_ids[i], found);
        if (found) { delete m; return fail(CTR_ERR_ARG, "duplicate feature id"); }
        *sl = (int32_t)i;
    }
    m->finder.build(m->feat, m->field);
    *out = m;
    return CTR_OK;
}

void ctr_fm_model_free(ctr_fm_model* m) { delete m; }
int64_t ctr_fm_model_n_rows(const ctr_fm_model* m) { return m ? (int64_t)m->feat.size() : 0; }
int ctr_fm_model_k(const ctr_fm_model* m) { return m ? m->k : 0; }
double ctr_fm_model_w0(const ctr_fm_model* m) { return m ? m->w0 : 0.0; }

int ctr_fm_model_copy(const ctr_fm_model* m, double* rows, int64_t* feat_ids, int32_t* field_of_row)
{
    if (!m) return fail(CTR_ERR_ARG, "null model");
    if (rows) { if (!m->from_file) return fail(CTR_ERR_ARG, "model has no rows (built from arrays)"); if (!m->rows.empty()) memcpy(rows, m->rows.data(), m->rows.size() * sizeof(double)); }
    if (feat_ids && !m->feat.empty()) memcpy(feat_ids, m->feat.data(), m->feat.size() * sizeof(int64_t));
    if (field_of_row && !m->field.empty()) memcpy(field_of_row, m->field.data(), m->field.size() * sizeof(int32_t));
    return CTR_OK;
}

int ctr_count_lines(const char* path, int n_threads, int64_t* n_lines, int64_t* n_examples)
{
    if (!path) return fail(CTR_ERR_ARG, "null path");
    Mapped f;
    int rc = f.open_file(path);
    if (rc != CTR_OK) return rc;
    std::vector<Range> rs = cut(f, n_threads);
    count_ranges(f, rs);
    if (n_lines) *n_lines = rs.back().line0 + rs.back().lines;
    if (n_examples) *n_examples = rs.back().ex0 + rs.back().examples;
    return CTR_OK;
}

int ctr_parse_examples(const char* path, int mode, const ctr_fm_model* m, int width, int n_threads, int64_t cap,
                       int32_t* ids_out, int32_t* vals_out, int32_t* y_out, int64_t* n_out)
{
    return ctr_parse_examples_ex(path, mode, m, width, n_threads, cap, ids_out, vals_out, y_out, n_out, 0, nullptr, nullptr);
}

int ctr_parse_examples_ex(const char* path, int mode, const ctr_fm_model* m, int width, int n_threads, int64_t cap,
                          int32_t* ids_out, int32_t* vals_out, int32_t* y_out, int64_t* n_out,
                          int64_t shadow_cap, int32_t* shadow_out, int64_t* n_shadow)
{
    if (!path || !ids_out || !y_out || !n_out || width < 1 || cap < 0) return fail(CTR_ERR_ARG, "null / bad argument");
    if (mode < CTR_MODE_FNN || mode > CTR_MODE_PAIRS) return fail(CTR_ERR_ARG, "bad mode");
    if (mode == CTR_MODE_FNN && (!m || width != m->n_fields)) return fail(CTR_ERR_ARG, "CTR_MODE_FNN needs the FM model and width == its n_fields");
    if (mode == CTR_MODE_PAIRS && !vals_out) return fail(CTR_ERR_ARG, "CTR_MODE_PAIRS needs vals_out");
    *n_out = 0;
    Mapped f;
    int rc = f.open_file(path);
    if (rc != CTR_OK) return rc;
    std::vector<Range> rs = cut(f, n_threads);
    count_ranges(f, rs);
    const int64_t total = rs.back().ex0 + rs.back().examples;
    if (total > cap) return fail(CTR_ERR_CAP, std::string(path) + ": " + std::to_string(total) + " examples, room for " + std::to_string(cap));
    Err err;
    if (n_shadow) *n_shadow = 0;
    if (shadow_cap < 0 || (shadow_cap > 0 && !shadow_out)) return fail(CTR_ERR_ARG, "shadow_cap without shadow_out");
    std::vector<std::vector<int64_t>> shadows(rs.size());      // per range, (example, field, row) in line order
    run_threads((int)rs.size(), [&](int t) {
        const Range& r = rs[t];
        std::vector<Tok> tok;
        std::vector<int64_t> fid;
        static const SepTable sep;
        std::vector<int64_t>& sh = shadows[t];
        size_t s = r.lo, e, nx; int64_t ln = r.line0 + 1, ex = r.ex0;
        for (; s < r.hi; s = nx, ++ln) {
            line_end(f.p, f.n, s, e, nx);
            if (blank(f.p, s, e)) continue;
            int32_t* row = ids_out + (size_t)ex * width;
            int32_t* vrow = vals_out ? vals_out + (size_t)ex * width : nullptr;
            for (int j = 0; j < width; ++j) row[j] = -1;
            if (vrow) for (int j = 0; j < width; ++j) vrow[j] = 0;
            int64_t y;
            if (mode == CTR_MODE_FNN) {
                // `.replace(':', ' ').split()` and int() of the label and of every second token after it, in ONE walk over the
                // bytes: a token of 1..18 digits is read as it is scanned, anything else (sign, junk, 19+ digits) goes to py_int.
                // Ids first (each prefetches its table entry), rows second: the misses of one line overlap.
                fid.clear();
                size_t i = s; int64_t nt = 0; bool have_y = false;
                while (i < e) {
                    while (i < e && sep.t[(unsigned char)f.p[i]]) ++i;
                    if (i >= e) break;
                    const size_t b = i; uint64_t v = 0; bool digits = true;
                    for (; i < e && !sep.t[(unsigned char)f.p[i]]; ++i) {
                        const unsigned d = (unsigned char)f.p[i] - (unsigned)'0';
                        digits &= d <= 9; v = v * 10 + d;
                    }
                    if (nt == 0 || (nt & 1)) {
                        int64_t val = (int64_t)v;
                        if ((!digits || i - b > 18) && !py_int(f.p + b, f.p + i, val)) {
                            err.set(ln, CTR_ERR_PARSE, nt == 0 ? "label is not an int" : "feature id is not an int"); return;
                        }
                        if (nt == 0) { y = val; have_y = true; }
                        else { m->finder.prefetch(val); fid.push_back(val); }
                    }
                    ++nt;
                }
                if (!have_y || y < INT32_MIN || y > INT32_MAX) { err.set(ln, CTR_ERR_PARSE, "label is not an int"); return; }
                y_out[ex] = (int32_t)y;
                for (int64_t feat : fid) {
                    int32_t rr, fld;
                    if (!m->finder.find(feat, rr, fld)) { err.set(ln, CTR_ERR_KEY, "feature " + std::to_string(feat) + " is not in the FM model (KeyError)"); return; }
                    int32_t& slot = row[fld];
                    // an earlier feature of the same field: the gather forgets it (the later one wins, data_fm.py:52-53), the
                    // update loop does not (python/FNN_wnzh.py:300-306 walks every feature of the line)
                    if (slot >= 0) { sh.push_back(ex); sh.push_back(fld); sh.push_back(slot); }
                    slot = rr;
                }
            } else {
                // the reference's order of evaluation, so that a line with several defects raises what it raises there:
                // python/SNN_RBM.py:249-253 -- per pair int(s[f + 1]) (IndexError when the value is missing, ValueError when malformed),
                // then int(s[f]) of an active pair; the label int(s[0]) LAST (:258).  get_batch_x (rbm_sparse.py:148-151): the value,
                // then the id, of every pair; s[0] is never read -- a label that is not an int is 0 here, not an error.
                tokenize(f.p, s, e, false, tok);
                int n = 0;
                for (size_t j = 1; j < tok.size(); j += 2) {
                    int64_t feat, val;
                    if (j + 1 >= tok.size()) { err.set(ln, CTR_ERR_INDEX, "id without a value (IndexError)"); return; }
                    if (!py_int(tok[j + 1].b, tok[j + 1].e, val)) { err.set(ln, CTR_ERR_PARSE, "value is not an int"); return; }
                    if (mode == CTR_MODE_SNN_ACTIVE && val != 1) continue;      // int(s[f]) is only evaluated for active features
                    if (!py_int(tok[j].b, tok[j].e, feat) || feat < INT32_MIN || feat > INT32_MAX) { err.set(ln, CTR_ERR_PARSE, "feature id is not an int32"); return; }
                    if (val < INT32_MIN || val > INT32_MAX) { err.set(ln, CTR_ERR_PARSE, "value does not fit int32"); return; }
                    if (n >= width) { err.set(ln, CTR_ERR_CAP, "more than " + std::to_string(width) + " features on the line"); return; }
                    row[n] = (int32_t)feat;
                    if (vrow) vrow[n] = (int32_t)val;
                    ++n;
                }
                const bool have_y = !tok.empty() && py_int(tok[0].b, tok[0].e, y) && y >= INT32_MIN && y <= INT32_MAX;
                if (!have_y && mode == CTR_MODE_SNN_ACTIVE) { err.set(ln, CTR_ERR_PARSE, "label is not an int"); return; }
                y_out[ex] = have_y ? (int32_t)y : 0;
            }
            ++ex;
        }
    });
    if (err.code != CTR_OK) return fail(err.code, std::string(path) + ":" + std::to_string(err.line.load()) + ": " + err.msg);
    *n_out = total;
    int64_t ns = 0;
    for (auto& sh : shadows) ns += (int64_t)sh.size() / 3;
    if (n_shadow) *n_shadow = ns;
    if (shadow_out) {
        if (ns > shadow_cap) return fail(CTR_ERR_CAP, std::string(path) + ": " + std::to_string(ns) + " shadowed features, room for " + std::to_string(shadow_cap));
        int64_t o = 0;
        for (auto& sh : shadows)                                 // ranges are in file order
            for (size_t i = 0; i < sh.size(); i += 3) {
                if (sh[i] > INT32_MAX) return fail(CTR_ERR_CAP, "example index of a shadowed feature does not fit int32");
                shadow_out[o++] = (int32_t)sh[i]; shadow_out[o++] = (int32_t)sh[i + 1]; shadow_out[o++] = (int32_t)sh[i + 2];
            }
    }
    return CTR_OK;
}

// yzx: `fields = line.strip().split()`, y = int(fields[0]), ids = int(tok.split(':')[0]) for fields[2:]
static inline bool yzx_line(const char* p, size_t s, size_t e, std::vector<Tok>& tok, int64_t& y, std::vector<int64_t>& ind)
{
    tok.clear(); ind.clear();
    for (size_t i = s; i < e;) {
        while (i < e && is_ws(p[i])) ++i;
        if (i >= e) break;
        const size_t b = i;
        while (i < e && !is_ws(p[i])) ++i;
        tok.push_back(Tok{p + b, p + i});
    }
    if (tok.empty() || !py_int(tok[0].b, tok[0].e, y)) return false;
    for (size_t j = 2; j < tok.size(); ++j) {
        const char* c = static_cast<const char*>(memchr(tok[j].b, ':', tok[j].e - tok[j].b));
        int64_t v;
        if (!py_int(tok[j].b, c ? c : tok[j].e, v)) return false;
        ind.push_back(v);
    }
    return true;
}

int ctr_yzx_stat(const char* path, int n_threads, int64_t* n_examples, int64_t* max_dim, int64_t* max_fea)
{
    if (!path || !max_dim || !max_fea) return fail(CTR_ERR_ARG, "null argument");
    Mapped f;
    int rc = f.open_file(path);
    if (rc != CTR_OK) return rc;
    std::vector<Range> rs = cut(f, n_threads);
    count_ranges(f, rs);
    std::vector<int64_t> md(rs.size(), 0), mf(rs.size(), 0);
    Err err;
    run_threads((int)rs.size(), [&](int t) {
        const Range& r = rs[t];
        std::vector<Tok> tok; std::vector<int64_t> ind;
        size_t s = r.lo, e, nx; int64_t ln = r.line0 + 1;
        for (; s < r.hi; s = nx, ++ln) {
            line_end(f.p, f.n, s, e, nx);
            if (blank(f.p, s, e)) { err.set(ln, CTR_ERR_PARSE, "blank line (IndexError in the reference's stat())"); return; }
            int64_t y;
            if (!yzx_line(f.p, s, e, tok, y, ind)) { err.set(ln, CTR_ERR_PARSE, "malformed yzx line"); return; }
            if (ind.empty()) { err.set(ln, CTR_ERR_PARSE, "no features (max() of an empty list)"); return; }
            mf[t] = std::max<int64_t>(mf[t], (int64_t)ind.size());
            md[t] = std::max<int64_t>(md[t], *std::max_element(ind.begin(), ind.end()));
        }
    });
    if (err.code != CTR_OK) return fail(err.code, std::string(path) + ":" + std::to_string(err.line.load()) + ": " + err.msg);
    *max_dim = *std::max_element(md.begin(), md.end());
    *max_fea = *std::max_element(mf.begin(), mf.end());
    if (n_examples) *n_examples = rs.back().line0 + rs.back().lines;
    return CTR_OK;
}

int ctr_parse_yzx(const char* path, int n_threads, int64_t cap, int64_t max_dim, int max_fea, int64_t* X_ind, int64_t* X_val,
                  int64_t* y_out, int64_t* n_out)
{
    if (!path || !X_ind || !X_val || !y_out || !n_out || max_fea < 1) return fail(CTR_ERR_ARG, "null / bad argument");
    *n_out = 0;
    Mapped f;
    int rc = f.open_file(path);
    if (rc != CTR_OK) return rc;
    std::vector<Range> rs = cut(f, n_threads);
    count_ranges(f, rs);
    const int64_t total = rs.back().line0 + rs.back().lines;          // the reference does not skip blank lines here
    if (total > cap) return fail(CTR_ERR_CAP, std::string(path) + ": " + std::to_string(total) + " lines, room for " + std::to_string(cap));
    Err err;
    run_threads((int)rs.size(), [&](int t) {
        const Range& r = rs[t];
        std::vector<Tok> tok; std::vector<int64_t> ind;
        size_t s = r.lo, e, nx; int64_t ln = r.line0 + 1;
        for (; s < r.hi; s = nx, ++ln) {
            line_end(f.p, f.n, s, e, nx);
            int64_t y;
            if (!yzx_line(f.p, s, e, tok, y, ind)) { err.set(ln, CTR_ERR_PARSE, "malformed yzx line"); return; }
            if ((int64_t)ind.size() > max_fea) { err.set(ln, CTR_ERR_CAP, "more than max_fea features"); return; }
            int64_t* xi = X_ind + (size_t)(ln - 1) * max_fea; int64_t* xv = X_val + (size_t)(ln - 1) * max_fea;
            for (int j = 0; j < max_fea; ++j) { const bool live = j < (int)ind.size(); xi[j] = live ? ind[j] : max_dim; xv[j] = live ? 1 : 0; }
            y_out[ln - 1] = y;
        }
    });
    if (err.code != CTR_OK) return fail(err.code, std::string(path) + ":" + std::to_string(err.line.load()) + ": " + err.msg);
    *n_out = total;
    return CTR_OK;
}

}  // extern "C"
