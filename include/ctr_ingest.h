/* ctr_ingest.h -- native text ingestion for the FNN / SNN hot path (host code of libfnn_hip.so).
 *
 * Replaces the per-line Python parsing that surrounds the reference's training loop and that it
 * repeats for every batch, every epoch and every evaluation pass:
 *   A1  fm.model.txt parser        python/FNN_wnzh.py:62-84  ==  python/data_fm.py:15-44
 *   A2  `y id:val id:val ...`       python/FNN_wnzh.py:224-253 (linecache.getline + get_fxy per line)
 *       SNN flavour                 python/SNN_RBM.py:238-262 (split on single spaces, value must be 1)
 *       RBM pre-training flavour    python/sampling_based_gaussian_binary_rbm_sparse.py:142-156,:423-437
 *   A12 yzx `y z idx:val ...`       python/ipinyou.py:23-65 (stat, load_ipinyou_data)
 * Files are mmap'ed, cut into byte ranges at line boundaries and parsed by n_threads std::threads
 * straight into the caller's int32 arrays: one pass per file instead of one per epoch.  Line
 * terminators are "\n", "\r\n" and "\r" (the reference reads with universal newlines); blank lines
 * (only whitespace) are skipped as the reference's `line.strip() != ''` does.
 *
 * Every function returns CTR_OK or a negative code; ctr_last_error() (thread-local) names the file,
 * the 1-based line and the reason.  The codes mirror the Python exception the reference raises.
 */
#ifndef CTR_INGEST_H
#define CTR_INGEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define CTR_OK          0
#define CTR_ERR_ARG    -1
#define CTR_ERR_IO     -2   /* IOError: cannot open / map the file                                   */
#define CTR_ERR_PARSE  -3   /* ValueError: int() or float() of a malformed token                       */
#define CTR_ERR_KEY    -4   /* KeyError: unknown feature id (FNN_wnzh.py:95) or field name (:82)      */
#define CTR_ERR_CAP    -5   /* the caller's arrays are too small for the file                         */
#define CTR_ERR_INDEX  -6   /* IndexError: a token the reference indexes is not there (s[f + 1] of an  */
                            /* id without a value; a model line that ends before its field tag)       */

/* token rules of the three readers of `y id:val id:val ...` */
#define CTR_MODE_FNN         0   /* get_fxy: ':' -> ' ', split on whitespace runs, ids = tokens 1,3,5,..;
                                    values never parsed; id -> (row, field) through the FM model;
                                    ids_out [n][n_fields] int32, slot = field, -1 = empty, later id of a
                                    field wins (data_fm.py:52-53)                                       */
#define CTR_MODE_SNN_ACTIVE  1   /* get_fi_h1_y: strip, ':' -> ' ', split on SINGLE spaces; a feature
                                    counts when int(value) == 1; ids_out [n][width] raw feature ids in
                                    line order, -1 padded                                              */
#define CTR_MODE_PAIRS       2   /* same split; ids_out [n][width] all ids, vals_out [n][width] all
                                    values, -1 / 0 padded (RBM pre-training builds its visibles from it) */

const char* ctr_last_error(void);

/* ---- A1: the FM model ------------------------------------------------------------------------ */
typedef struct ctr_fm_model ctr_fm_model;
/* field_names[n_fields]: the keys of name_field in index order (python/FNN_wnzh.py:51-53). */
int ctr_fm_model_load(const char* path, const char* const* field_names, int n_fields, int n_threads,
                      ctr_fm_model** out);
void ctr_fm_model_free(ctr_fm_model* m);
int64_t ctr_fm_model_n_rows(const ctr_fm_model* m);      /* distinct feature ids, file order          */
int ctr_fm_model_k(const ctr_fm_model* m);               /* rank + 1                                  */
double ctr_fm_model_w0(const ctr_fm_model* m);
/* rows [n_rows][k] float64 (the parsed text, as the reference's float()), feat_ids [n_rows],
 * field_of_row [n_rows]; any pointer may be NULL. */
int ctr_fm_model_copy(const ctr_fm_model* m, double* rows, int64_t* feat_ids, int32_t* field_of_row);
/* A model made from arrays (no file): the id map of a table built elsewhere. */
int ctr_fm_model_from_arrays(const int64_t* feat_ids, const int32_t* field_of_row, int64_t n_rows, int k,
                             int n_fields, ctr_fm_model** out);

/* ---- A2: example files ------------------------------------------------------------------------ */
/* Number of lines and of non-blank lines (= examples). */
int ctr_count_lines(const char* path, int n_threads, int64_t* n_lines, int64_t* n_examples);
/* Parse the whole file.  `m` is needed for CTR_MODE_FNN only.  width: n_fields of the model
 * (CTR_MODE_FNN) or the caller's row width.  cap: rows available in ids_out / vals_out / y_out.
 * vals_out: CTR_MODE_PAIRS only (NULL otherwise).  n_out: examples written. */
int ctr_parse_examples(const char* path, int mode, const ctr_fm_model* m, int width, int n_threads,
                       int64_t cap, int32_t* ids_out, int32_t* vals_out, int32_t* y_out, int64_t* n_out);

/* The same, also reporting what CTR_MODE_FNN's "later id of a field wins" drops: shadow_out [shadow_cap][3] int32 =
 * (example, field, row) of every feature that a later feature of the same field replaced in ids_out, in file order
 * (python/FNN_wnzh.py:300-306 still updates those rows: see fnn_set_shadowed in fnn_hip.h).  *n_shadow = how many there are
 * (written also when shadow_out is NULL: count first, or retry after CTR_ERR_CAP). */
int ctr_parse_examples_ex(const char* path, int mode, const ctr_fm_model* m, int width, int n_threads,
                          int64_t cap, int32_t* ids_out, int32_t* vals_out, int32_t* y_out, int64_t* n_out,
                          int64_t shadow_cap, int32_t* shadow_out, int64_t* n_shadow);

/* ---- A12: yzx ----------------------------------------------------------------------------------- */
/* stat(): max index and max number of features per line over tokens 2.. (python/ipinyou.py:23-39). */
int ctr_yzx_stat(const char* path, int n_threads, int64_t* n_examples, int64_t* max_dim, int64_t* max_fea);
/* load_ipinyou_data() for the whole file, in file order (the reference shuffles each buffer with the
 * global NumPy RNG afterwards -- that stays with the caller): X_ind [n][max_fea] (pad = max_dim),
 * X_val [n][max_fea] (1 present, 0 pad), y [n]. */
int ctr_parse_yzx(const char* path, int n_threads, int64_t cap, int64_t max_dim, int max_fea,
                  int64_t* X_ind, int64_t* X_val, int64_t* y, int64_t* n_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CTR_INGEST_H */
