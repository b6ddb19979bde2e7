/* fnn_hip.h -- C ABI of libfnn_hip.so, the MI355X (gfx950) FNN hot path.
 *
 * Drop-in boundary for the compiled Theano callables of Atomu2014/deep-ctr and
 * the two Python loops either side of them (paths relative to the reference):
 *
 *   train   = theano.function([x, y], [gx, w1, w2, w3, b1, b2, b3], updates=SGD)
 *                                                   python/FNN_wnzh.py:177-182
 *   predict = theano.function([x], [p_1])           python/FNN_wnzh.py:183
 *   gather  : feats_to_layer_one_array / get_batch_data
 *                                                   python/FNN_wnzh.py:87-96,224-237
 *                                                   python/data_fm.py:46-70
 *   scatter : the sparse-row SGD loop               python/FNN_wnzh.py:299-306
 *
 * Plain C: pointers, sizes, ints.  No C++ or torch types cross this boundary.
 * Every function returns 0 (FNN_OK) or a negative error class; the message is
 * available from fnn_last_error().  A handle is bound to one device and one
 * HIP stream; calls on one handle are not re-entrant; work is asynchronous on
 * the stream except where a host output pointer forces a synchronisation.
 *
 * Buffers passed in are owned by the caller.  `memkind` says whether the
 * pointers of that call are host (FNN_MEM_HOST) or device (FNN_MEM_DEVICE,
 * e.g. torch.Tensor.data_ptr()) addresses.
 *
 * Layouts (row-major, contiguous): ids int32 [B, F] with slot f = field f and
 * -1 = no feature in that field (FNN_MODE_FM: ids[t][f] must be a row of field f -- fnn_set_table's field_of_row --
 * which DataFM / ctr_parse_examples guarantee; the sparse-row update groups keys per column without atomics, so one row
 * under two columns of a batch is a data race there.  FNN_MODE_BAG detects such rows and adds atomically);
 * y float32 [B]; masks uint8 [H1], [H2];
 * x / gx float32 [B, 1 + F*K] in the reference's layer-one layout
 * (x[0] = w_0, x[1 + f*K + l] = row(ids[f])[l],  python/FNN_wnzh.py:87-96).
 */
#ifndef FNN_HIP_H
#define FNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define FNN_OK          0
#define FNN_ERR_ARG    -1   /* bad argument / unsupported shape              */
#define FNN_ERR_HIP    -2   /* HIP runtime error (message has the HIP text)  */
#define FNN_ERR_STATE  -3   /* call order (table or weights not set, ...)    */
#define FNN_ERR_RANGE  -4   /* an id was outside [-1, n_rows)  (the reference
                               raises KeyError, python/FNN_wnzh.py:95)       */
#define FNN_ERR_NOMEM  -5

#define FNN_PREC_F32    0   /* exact-f32 MFMA; the parity mode               */
#define FNN_PREC_BF16   1   /* bf16 MFMA inputs, f32 accumulate, f32 masters */
#define FNN_PREC_BF16X3 2   /* operands as bf16 pairs (hi + lo, 16 significant bits), three bf16 MFMAs per product, f32
                               accumulate, f32 masters: 4.4x the exact-f32 MFMA rate, error ~7x f32's (430x below bf16's);
                               the FNN / SNN engine only (fnn_create) */

#define FNN_ACT_TANH    0   /* acti_type, python/FNN_wnzh.py:23,148-163      */
#define FNN_ACT_SIGMOID 1
#define FNN_ACT_LINEAR  2

#define FNN_MEM_HOST    0
#define FNN_MEM_DEVICE  1

/* Input layer.  FM: x = [w_0 | rows of the F active features]  (python/FNN_wnzh.py:87-96).
 * BAG: x = sigmoid(sum of the rows of ww0 of the active features + bb0), the SNN fine-tune of
 * python/SNN_RBM.py:238-291; then the table is ww0 [n_rows, h0], layer 1 is W [h0, H1],
 * fnn_gather returns x [B, h0], gx_out is [B, h0], `k` is ignored, lambda1 usually comes with
 * reg_all = 1 (:141-143), and the sparse update is ww0[f] -= lr*gx*x*(1-x) (no decay) with
 * bb0 -= lr * sum_t gx*x*(1-x). */
#define FNN_MODE_FM     0
#define FNN_MODE_BAG    1

typedef struct fnn_cfg {
    int32_t n_fields;     /* F: 16 for iPinYou (python/FNN_wnzh.py:51-53)     */
    int32_t k;            /* K = rank + 1: row = [w, v_1..v_rank]  (:76)      */
    int32_t hidden1;      /* python/FNN_wnzh.py:21,46                         */
    int32_t hidden2;      /* python/FNN_wnzh.py:22,47                         */
    int32_t max_batch;    /* largest B of any call; sizes the workspaces      */
    int32_t precision;    /* FNN_PREC_*                                       */
    int32_t act;          /* FNN_ACT_* for h1 (and h2 in predict); the second
                             dropout layer is tanh regardless (:165)          */
    int32_t reg_all;      /* 0: lambda1 on w3,b3 only (python/FNN_wnzh.py:173)
                             1: on all six tensors (python/SNN_RBM.py:141)    */
    float   lr;           /* python/FNN_wnzh.py:19,44                         */
    float   lambda1;      /* :20,48                                           */
    float   lambda_fm;    /* :49                                              */
    int32_t device;       /* HIP device ordinal                               */
    void*   stream;       /* hipStream_t to run on, or NULL = create one      */
    int32_t mode;         /* FNN_MODE_FM (FNN) or FNN_MODE_BAG (SNN fine-tune) */
    int32_t h0;           /* FNN_MODE_BAG: width of the bag rows / of x
                             (hidden0, python/SNN_RBM.py:25,53: 300, or 200 for advertiser
                             2997); a multiple of 4 in [192, 316] */
} fnn_cfg;

typedef struct fnn_handle fnn_handle;

const char* fnn_version(void);
/* sizeof(fnn_cfg) as the library was compiled.  A binding declares the struct itself (ctypes, cgo, ...):
 * it must compare its own size with this before the first fnn_create, which reads every field. */
uint64_t fnn_cfg_size(void);
/* Message of the last failing call on `h` (or of the last failing fnn_create
 * when h == NULL).  Valid until the next call on that handle. */
const char* fnn_last_error(const fnn_handle* h);

int fnn_create(const fnn_cfg* cfg, fnn_handle** out);
int fnn_destroy(fnn_handle* h);
int fnn_set_hparams(fnn_handle* h, float lr, float lambda1, float lambda_fm);
void* fnn_stream(fnn_handle* h);               /* the hipStream_t in use      */
int fnn_sync(fnn_handle* h);                   /* wait + report async errors
                                                  (FNN_ERR_RANGE lands here)  */

/* FM table = feat_weights / feat_field / w_0 of python/FNN_wnzh.py:62-84.
 * rows [n_rows, K] float32, field_of_row [n_rows] int32 in [0, F). */
int fnn_set_table(fnn_handle* h, const float* rows, int64_t n_rows,
                  const int32_t* field_of_row, float w0, int memkind);
int fnn_get_table(fnn_handle* h, float* rows_out, int memkind);
int fnn_get_rows(fnn_handle* h, const int64_t* row_ids, int64_t n, float* out, int memkind);

/* Dense tensors, reference shapes (python/FNN_wnzh.py:106-140):
 * layer 1: W [1+F*K, H1], b [H1];  layer 2: W [H1, H2], b [H2];
 * layer 3: W [H2], b [1]. */
int fnn_set_dense(fnn_handle* h, int layer, const float* W, const float* b, int memkind);
int fnn_get_dense(fnn_handle* h, int layer, float* W, float* b, int memkind);

/* FNN_MODE_BAG only: the bias bb0 [h0] of the bag layer (python/SNN_RBM.py:78,255,289). */
int fnn_set_bag_bias(fnn_handle* h, const float* bb0, int memkind);
int fnn_get_bag_bias(fnn_handle* h, float* bb0_out, int memkind);

/* A3: x_out [B, 1+F*K] = layer-one array of every example. */
int fnn_gather(fnn_handle* h, const int32_t* ids, int B, float* x_out, int memkind);

/* One pass of the hot loop body (python/FNN_wnzh.py:296-306):
 * gather -> train(x, y) with the given dropout rows -> dense SGD -> sparse-row
 * SGD with decay 1 - 2*lambda_fm*lr/b_size.  b_size <= 0 means B (the data-
 * parallel caller passes the GLOBAL batch length).  Outputs are optional:
 * p_out [B] = p_drop, gx_out [B, 1+F*K] (what `train` returns first),
 * loss_sum_out = sum of the cross-entropy over the batch (HOST pointer; non-NULL
 * forces a stream synchronisation). */
int fnn_train_step(fnn_handle* h, const int32_t* ids, const float* y, int B,
                   const uint8_t* mask1, const uint8_t* mask2, int b_size,
                   float* p_out, float* gx_out, int memkind, float* loss_sum_out);

/* Features that a LATER feature of the same field shadows in the gather of the NEXT fnn_train_step / fnn_step_begin call.
 * The layer-one array keeps one feature per field, the last of the line (python/FNN_wnzh.py:91-96), which is what
 * ids [B, F] expresses; the reference's update loop nevertheless walks EVERY feature of the line (:300-306), so a shadowed
 * feature's row also takes `row * c - lr * gx[t][1 + field*K + l]`.  tfr [n][3] int32 = (example t in [0, B), field, row),
 * in any order; consumed by the next step (n = 0 clears).  FNN_MODE_FM; not combined with FNN_DP_SPARSE_EXCHANGE.  Such
 * steps take the layer-by-layer kernels (B + n <= 16384).  iPinYou lines hold one feature per field: n = 0 there.
 * `field` must be the row's own field (fnn_set_table's field_of_row; FNN_ERR_ARG otherwise): the update groups keys per field, a
 * row named under another field would sit in two groups of one launch.  Synchronises the handle's stream. */
int fnn_set_shadowed(fnn_handle* h, const int32_t* tfr, int n, int memkind);

/* Optional: hand the ids of an UPCOMING training batch to the library (DEVICE pointer, same
 * pointer and B as the later fnn_train_step / fnn_step_begin call).  The sparse-row update first
 * groups a batch's (row, example) pairs by row -- the device-side counterpart of the reference
 * walking `for feat in ft` in example order (python/FNN_wnzh.py:300-306) -- and that grouping
 * depends on the ids only, so it can run while the previous step still computes.  Purely a
 * scheduling hint: results are identical with or without it.  The ids must not change between
 * this call and the step that consumes them. */
int fnn_prefetch_ids(fnn_handle* h, const int32_t* ids, int B);

/* ---- Data parallelism (new: the reference has no distributed code).  One process per GPU; the global batch is cut into
 * contiguous shards; the loss is a batch SUM (python/FNN_wnzh.py:173), so the dense gradients of the shards add up exactly and
 * ONE collective per step is the whole exchange.  Tables are replicated.
 *
 * Native form: after fnn_dp_init the ordinary fnn_train_step IS the data-parallel step -- the same three launches as on one
 * GPU with the collective between the second and the third: launch 1 (strips), launch 2 (weight-gradient slabs | sparse-row
 * SGD level 1 | run sorts), all-reduce of the split-K slabs on the handle's stream, launch 3 (slab reduce + dense SGD | sparse-row
 * SGD level 2 | rank merge).  b_size must then be the GLOBAL batch length (the decay of the sparse update, :304), and the
 * dropout rows must be the same on every rank (they are per batch, not per example, :154,166).
 *   FNN_DP_SPARSE_LOCAL     every rank applies the sparse-row updates of its own shard (replicas drift apart on rows that
 *                           several ranks touch: the throughput mode)
 *   FNN_DP_SPARSE_EXCHANGE  the step also all-gathers (ids, gx') of every shard, each padded to max_batch rows, and every rank
 *                           applies the whole global batch's row updates: replicas stay identical (the parity mode; world *
 *                           max_batch <= 32768).  FNN_MODE_FM: in global example order, equal to a single-process run of the
 *                           global batch.  FNN_MODE_BAG (no decay, the update is a sum: python/SNN_RBM.py:285-291): the shards
 *                           one after the other in rank order -- replicas bit-identical to each other, equal to the
 *                           single-process run up to the rounding of `world` partial sums per row
 * The collectives are RCCL's: librccl.so.1 is opened at fnn_dp_init (no link-time dependency), the handle owns its communicator. */
#define FNN_DP_SPARSE_LOCAL    0
#define FNN_DP_SPARSE_EXCHANGE 1
/* Rank 0: a fresh 128-byte ncclUniqueId to hand to every rank (any side channel: torch.distributed, MPI, a file). */
int fnn_dp_unique_id(void* id128_out);
/* Every rank, collectively (blocks until all `world` ranks have called it). */
int fnn_dp_init(fnn_handle* h, int rank, int world, const void* id128, int sparse_mode);
/* ... or the caller's own collectives in place of RCCL (a gloo rehearsal, MPI, a test double that stands for several ranks).
 * Each callback must ENQUEUE the operation on `stream` (or complete it before returning) and return 0:
 *   allreduce(ctx, buf, n_floats, stream)                     in-place f32 sum over the ranks
 *   allgather(ctx, send, recv, bytes_per_rank, stream)        recv = the ranks' `send` blocks in rank order (EXCHANGE only) */
typedef int (*fnn_allreduce_fn)(void* ctx, float* buf, int64_t n_floats, void* stream);
typedef int (*fnn_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int fnn_dp_init_custom(fnn_handle* h, int rank, int world, fnn_allreduce_fn allreduce, fnn_allgather_fn allgather,
                       void* ctx, int sparse_mode);
/* Back to single-process steps (destroys the communicator).  fnn_destroy does this too. */
int fnn_dp_shutdown(fnn_handle* h);

/* What the dense collective of the native step carries, and who performs it.  Both are properties of the HANDLE, set the same
 * way on every rank: the count and kind of the step's collectives never depend on a rank's own shard (a shard that takes the
 * layer-by-layer kernels -- shadowed features, more than 4096 examples -- issues the same collective as its peers).
 *   FNN_DP_PAYLOAD_SLABS    the split-K slabs of the weight gradients, reduced in place between launch 2 and launch 3
 *                           (4 x 131,072 floats at the reference shape: 2 MB; three launches + the collective)
 *   FNN_DP_PAYLOAD_BUCKET   launch 3 first sums this rank's slabs into the flat bucket (w1 | w2 | w3 | bag bias, padded:
 *                           123,008 floats = 0.5 MB at the reference shape), the collective reduces the bucket, a fourth
 *                           launch applies the update (a quarter of the bytes, one more launch)
 * Default: $FNN_DP_PAYLOAD ("slabs" / "bucket") read at fnn_dp_init*, else slabs.  Call between steps. */
#define FNN_DP_PAYLOAD_SLABS   0
#define FNN_DP_PAYLOAD_BUCKET  1
int fnn_dp_set_payload(fnn_handle* h, int payload);
/*   FNN_DP_COLLECTIVE_CALLBACK  RCCL (fnn_dp_init) or the caller's all-reduce (fnn_dp_init_custom)
 *   FNN_DP_COLLECTIVE_P2P       one-shot all-reduce over peer pointers INSIDE the update launch: every rank publishes its
 *                               bucket in an exchange region the peers have mapped (hipIpc*), raises a flag in every peer's
 *                               region, waits for the peers' flags and sums all buckets in rank order (so every rank forms the
 *                               same sum, bit for bit).  No library collective on the dense path; the payload is the bucket.
 *                               xGMI is point to point: each of the 7 reads of a rank's 0.5 MB uses its own link.
 * Set-up, after fnn_dp_init / fnn_dp_init_custom (which fixes rank and world, and still serves the all-gathers of
 * FNN_DP_SPARSE_EXCHANGE): every rank calls fnn_dp_p2p_export, the 64-byte handles are exchanged through any side channel,
 * every rank calls fnn_dp_p2p_attach with all `world` handles in rank order (its own included), then
 * fnn_dp_set_collective(h, FNN_DP_COLLECTIVE_P2P).  same_process != 0: the ranks are handles of ONE process (tests): a
 * "handle" then holds the region's device pointer in its first 8 bytes and nothing is opened.
 * A peer that does not arrive within $FNN_P2P_TIMEOUT_MS (default 30 s) makes the step fail (fnn_sync / the next host read returns
 * FNN_ERR_HIP) instead of hanging; the dense tensors of that step are left untouched. */
#define FNN_DP_COLLECTIVE_CALLBACK 0
#define FNN_DP_COLLECTIVE_P2P      1
int fnn_dp_p2p_export(fnn_handle* h, void* handle64_out, int same_process);
int fnn_dp_p2p_attach(fnn_handle* h, const void* handles, int same_process);
int fnn_dp_set_collective(fnn_handle* h, int collective);
/* What is in force: payload, collective, and the kind of memory the exchange region lives in (0 none, 1 uncached,
 * 2 fine-grained, 3 plain hipMalloc). */
int fnn_dp_get_config(fnn_handle* h, int* payload, int* collective, int* region_kind);
/* FNN_DP_COLLECTIVE_P2P diagnostic: the longest time (microseconds) the update launch of any step so far has waited for a peer's
 * flag (rank skew + the flag's way over the fabric).  The wait is bounded by $FNN_P2P_TIMEOUT_MS (default 30,000).  Synchronises. */
int fnn_dp_p2p_max_wait_us(fnn_handle* h, double* us_out);

/* Portable form, for a caller that issues the collective itself between two calls: _begin runs everything except the
 * dense SGD and leaves the dense gradients (sum over this rank's examples) in one flat f32
 * bucket; the caller all-reduces the bucket on fnn_stream(); _end applies
 * theta <- theta - lr * bucket.  (Five launches instead of three: the native form above is the fast one.) */
int fnn_step_begin(fnn_handle* h, const int32_t* ids, const float* y, int B,
                   const uint8_t* mask1, const uint8_t* mask2, int b_size,
                   float* p_out, float* gx_out, int memkind);
int fnn_dense_grad_bucket(fnn_handle* h, float** dev_ptr, int64_t* n_floats);
/* Optional, between _begin and _end: enqueue the sparse-row half of the step now, so that it
 * overlaps an all-reduce the caller has started asynchronously; _end runs it if this was not
 * called.  (_begin leaves the bucket complete before the sparse half is enqueued.) */
int fnn_step_scatter(fnn_handle* h);
/* Exact data-parallel mode (FNN_MODE_FM): the sparse-row exchange.  After _begin the slot-layout
 * row gradients of this rank's shard sit in gx' [B, *row_floats] f32 (column 16 f + l = d cost /
 * d row(ids[t][f])[l]); fnn_sparse_grad exposes that buffer so that the caller can all-gather it
 * together with the ids.  fnn_step_scatter_global then applies, INSTEAD of fnn_step_scatter, the
 * sparse-row SGD of the whole global batch in global example order (python/FNN_wnzh.py:299-306 --
 * every replica's table stays identical to the single-process run): ids_g [B_g, F] int32 (-1 =
 * empty / padding), gxp_g [B_g, row_floats] f32, DEVICE pointers, B_g <= max_batch. */
int fnn_sparse_grad(fnn_handle* h, float** dev_ptr, int64_t* row_floats);
int fnn_step_scatter_global(fnn_handle* h, const int32_t* ids_g, const float* gxp_g, int B_g);
int fnn_step_end(fnn_handle* h, float* loss_sum_out);
/* Sum of the cross-entropy of the last step on this rank (synchronises). */
int fnn_last_loss(fnn_handle* h, float* loss_sum_out);

/* A4': p_out [B] = predict(x) -- no masks, no rescale (python/FNN_wnzh.py:183). */
int fnn_predict(fnn_handle* h, const int32_t* ids, int B, float* p_out, int memkind);

/* A10: one evaluation pass (python/FNN_wnzh.py:193-221 get_err_bat; python/SNN_RBM.py:162-198):
 * predict all N examples (any N; internally in max_batch chunks, predictions stay in HBM), then
 * roc_auc_score, sqrt(mean_squared_error) and log_loss (python/baseline.py:427-429) on the device.
 * ids [N, F] int32, y [N] int32 labels; p_out [N] float32 or NULL; metrics are HOST doubles.
 * Only one class in y: FNN_ERR_RANGE (the reference's roc_auc_score raises ValueError); rmse and
 * logloss are still written. */
int fnn_eval(fnn_handle* h, const int32_t* ids, const int32_t* y, int64_t N, int memkind,
             double* auc, double* rmse, double* logloss, float* p_out);

/* Timing hook for bench.py: average device time (ms) of the kernel named
 * `which` ("gather", "fwd1", "fwd2", "head", "bwd1", "gx", "wgrad", "reduce",
 * "update", "sort", "scatter", "finalize") over the steps since the last
 * fnn_prof_reset(); HIP events on the handle's own streams.  Profiling is off
 * (zero overhead) until fnn_prof_enable(h, 1). */
int fnn_prof_enable(fnn_handle* h, int on);
int fnn_prof_reset(fnn_handle* h);
int fnn_prof_get(fnn_handle* h, const char* which, double* avg_ms, int64_t* launches);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FNN_HIP_H */
