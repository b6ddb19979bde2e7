/* fm_hip.h -- C ABI of factorisation-machine pre-training in libfnn_hip.so (MI355X, gfx950).
 *
 * The step BEFORE the FNN hot path: the model whose rows [w_i, v_i1..v_ik]
 * `fm.model.txt` carries into python/FNN_wnzh.py:62-84.  Arithmetic of the reference's TensorFlow
 * class python/FM.py: `factorization` :55-64
 *      yhat = b + sum_i w_i x_i + 1/2 (|sum_i v_i x_i|^2 - sum_i |v_i|^2 x_i^2),
 * loss :36-41 (sigmoid cross-entropy, reduce_sum or reduce_mean, + lambda * (l2_loss(W) + l2_loss(V)
 * + l2_loss(b)) with tf.nn.l2_loss = sum(t^2)/2), plain SGD (python/tf_util.py:26-29), as driven by
 * python/ipinyou.py:136-173.  One feature per field with value 1 (iPinYou; `load_ipinyou_data`
 * returns X_val = 1); ids [B, F] int32 with -1 = absent.
 *
 * The L2 term makes TensorFlow's gradient DENSE: every step multiplies the whole table by
 * (1 - lr * lambda).  Here the table is kept as `scale * stored` -- the decay is one scalar
 * multiplication per step, touched rows get -lr * g / scale through the same sorted, atomics-free
 * sparse-row update as the FNN path, and the scale is folded back into the rows before it leaves
 * 2^-24 .. 1.  Same result as the dense update up to f32 rounding.
 *
 * Error codes: FNN_ERR_* of fnn_hip.h; fm_last_error() has the message.
 */
#ifndef FM_HIP_H
#define FM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct fm_handle fm_handle;

const char* fm_last_error(const fm_handle* h);
/* k = rank + 1 (row = [w | v_1..v_rank]), 1 <= k <= 16; max_batch <= 4096. */
int fm_create(int n_fields, int k, int max_batch, int device, void* stream, fm_handle** out);
int fm_destroy(fm_handle* h);
int fm_sync(fm_handle* h);

/* HOST pointers.  rows [n_rows, k] = concat(W, V) (python/FM.py:19-20); b = the scalar bias (:21). */
int fm_set_table(fm_handle* h, const float* rows, int64_t n_rows);
int fm_get_table(fm_handle* h, float* rows_out);
int fm_get_rows(fm_handle* h, const int64_t* row_ids, int64_t n, float* out);
int fm_set_b(fm_handle* h, float b);
int fm_get_b(fm_handle* h, float* b);

/* One SGD step on a mini-batch.  DEVICE pointers: ids [B, F] int32, y [B] f32; p_out [B] =
 * sigmoid(yhat) before the update (`train_preds`, :42; nullable).  reduce_mean != 0: loss =
 * mean(xent) (the driver's setting), else sum.  loss_out (HOST, nullable; synchronises): the data
 * term of the loss as reduced. */
int fm_train_step(fm_handle* h, const int32_t* ids, const float* y, int B, float lr, float lambda,
                  int reduce_mean, float* p_out, float* loss_out);
/* p_out [B] = sigmoid(yhat) (`test_preds`, :52). */
int fm_predict(fm_handle* h, const int32_t* ids, int B, float* p_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FM_HIP_H */
