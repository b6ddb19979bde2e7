/* ipnn_hip.h -- C ABI of the inner-product FNN family (FNN_IP_L3 / L5 / L7) in libfnn_hip.so.
 *
 * Replaces the TensorFlow graph of Atomu2014/deep-ctr's python/FNN_IP_L7.py (and _L3 / _L5, same
 * pattern): `forward` :102-133 (embeddings, pair-wise inner products, z1 = [e | p | b], then
 * l_{t+1} = dropout(act(l_t)) W_t + b_t with activation and inverted dropout BEFORE every matmul),
 * the loss sum(sigmoid_cross_entropy_with_logits) :82-88 and the gradient step.  Categorical
 * fields only (iPinYou shape: one id per field); optimiser: plain SGD, Adam or FTRL (IPNN_OPT_*).  Dropout keep-masks are INPUTS (uint8, one per element,
 * reference column order), NULL = no dropout (`drop_out=False`).
 *
 * Error codes are the FNN_ERR_* of fnn_hip.h; ipnn_last_error() has the message.
 */
#ifndef IPNN_HIP_H
#define IPNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define IPNN_ACT_TANH    0      /* python/tf_util.py:32-38 `activate` */
#define IPNN_ACT_SIGMOID 1
#define IPNN_ACT_RELU    3

#define IPNN_MAX_HIDDEN  8

#define IPNN_OPT_SGD     0      /* python/tf_util.py:26-29 GradientDescentOptimizer                       */
#define IPNN_OPT_ADAM    1      /* python/tf_util.py:17-20 AdamOptimizer(learning_rate, epsilon): the
                                   reference's choice for this family (python/baseline.py:146, lr 1e-4,
                                   eps 1e-8).  TensorFlow's gradient of the embedding tables is dense, so
                                   EVERY row's moments decay and every row moves each step             */
#define IPNN_OPT_FTRL    2      /* python/tf_util.py:21-24 FtrlOptimizer(learning_rate): TensorFlow's
                                   defaults (learning_rate_power -0.5, initial accumulator 0.1, l1 = l2 =
                                   0); also a dense pass over the tables: a row no example has touched is
                                   re-derived from its (zero) linear term, i.e. drops to 0 at step 1   */

typedef struct ipnn_cfg {
    int32_t n_fields;                  /* X_feas                                             */
    int32_t k;                         /* rank + 1: embedding row [w | v]  (FNN_IP_L7.py:66) */
    int32_t n_hidden;                  /* 3, 5 or 7 (any 1..8)                               */
    int32_t hidden[IPNN_MAX_HIDDEN];   /* e.g. 1000,800,600,400,200,100,50 (baseline.py:139) */
    int32_t act;                       /* IPNN_ACT_*                                         */
    int32_t pairs;                     /* 1: z1 = [e | p | b] (FNN_IP_L*, FNN_IP_L7.py:108-114);
                                          0: z1 = [e | b], the plain `FNN` class (python/FNN.py:80) */
    int32_t max_batch;                 /* <= 4096                                            */
    int32_t precision;                 /* FNN_PREC_F32 / FNN_PREC_BF16                       */
    float   lr;
    float   keep_prob;                 /* _reg_argv[0]                                       */
    int32_t optimizer;                 /* IPNN_OPT_*                                         */
    float   adam_beta1, adam_beta2;    /* TensorFlow defaults 0.9, 0.999                     */
    float   adam_eps;                  /* _ptmzr_argv[2]                                     */
    int32_t device;
    void*   stream;
} ipnn_cfg;

typedef struct ipnn_handle ipnn_handle;

const char* ipnn_last_error(const ipnn_handle* h);
/* sizeof(ipnn_cfg) as the library was compiled (a binding checks its own struct against it). */
uint64_t ipnn_cfg_size(void);
int ipnn_create(const ipnn_cfg* cfg, ipnn_handle** out);
int ipnn_destroy(ipnn_handle* h);
int ipnn_sync(ipnn_handle* h);

/* HOST pointers.  table rows [n_rows, K] = concat(W, V) (fm_wv, :66); b: the scalar `fm_b`. */
int ipnn_set_table(ipnn_handle* h, const float* rows, int64_t n_rows);
int ipnn_get_rows(ipnn_handle* h, const int64_t* row_ids, int64_t n, float* out);
int ipnn_set_b(ipnn_handle* h, float b);
int ipnn_get_b(ipnn_handle* h, float* b);
/* layer i = 1 .. n_hidden+1: W [d_{i-1}, d_i], bias [d_i]; d_0 = F*K + F(F-1)/2 + 1 (`mbd_dim`,
 * FNN_IP_L3.py:18), d_{n_hidden+1} = 1.  Reference row order of h1_w: [e_0..e_{F-1} | pairs | b]. */
int ipnn_set_layer(ipnn_handle* h, int layer, const float* W, const float* bias);
int ipnn_get_layer(ipnn_handle* h, int layer, float* W, float* bias);

/* DEVICE pointers: ids int32 [B, F], y f32 [B], masks[t] uint8 [B, d_t] for t = 0..n_hidden
 * (array of n_hidden+1 device pointers held in HOST memory; NULL = no dropout).
 * One SGD step.  logits_out [B] (device, nullable); loss_sum_out (host, nullable: synchronises). */
int ipnn_train_step(ipnn_handle* h, const int32_t* ids, const float* y, int B,
                    const uint8_t* const* masks, float* logits_out, float* loss_sum_out);
/* Loss reduction of the following train steps: 0 (default) = tf.reduce_sum, 1 = tf.reduce_mean over the batch
 * (`_ptmzr_argv[-1]`, python/FNN_IP_L7.py:83-86): every gradient of a step is scaled by 1 / B.  loss_sum_out stays the
 * SUM of the per-example cross-entropies (divide by B on the host for the mean). */
int ipnn_set_loss_mean(ipnn_handle* h, int mean);
/* p_out [B] = sigmoid(logits) without dropout (`test_preds`, FNN_IP_L3.py:81-84). */
int ipnn_predict(ipnn_handle* h, const int32_t* ids, int B, float* p_out);

/* Evaluation pass (python/baseline.py:382-437 `test`): predict all N examples (DEVICE ids [N, F]
 * int32, y [N] int32; chunks of max_batch), then AUC / RMSE / logloss on the device.  Metrics are
 * HOST doubles.  One class only: FNN_ERR_RANGE. */
int ipnn_eval(ipnn_handle* h, const int32_t* ids, const int32_t* y, int64_t N, double* auc, double* rmse, double* logloss);

/* Measurement hook (bench.py): HIP events on the handle's stream around the segments of a train
 * step -- "sort", "ip_fwd", "fwd", "bwd", "wgrad", "ip_bwd", "scatter", "update".  enable(1) clears
 * earlier samples; get returns the average device time of one segment in ms (0 if none). */
int ipnn_prof_enable(ipnn_handle* h, int on);
int ipnn_prof_get(ipnn_handle* h, const char* which, double* avg_ms);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* IPNN_HIP_H */
