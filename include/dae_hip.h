/* dae_hip.h -- C ABI of the SNN-DAE pre-training kernels in libfnn_hip.so (MI355X, gfx950).
 *
 * Replaces the Theano denoising autoencoders of Atomu2014/deep-ctr's
 * python/sampling_based_denosing_autoencoder.py, as `get_da_weights` (:347-371) runs them
 * (batch_size = 1, corruption_level = 0, learning_rate = 0.1, tied weights, sigmoid, cross-entropy):
 *
 *   sparse_da   :234-345   one sampled negative per feature, rows gathered per example  -> dae_sparse_epoch
 *   da          :116-232   dense upper layers, online SGD                               -> dae_dense_epoch
 *   the lower-layer propagation inside da() :163-187                                   -> dae_bag_cumsum_sigmoid
 *                                                                           (+ rbm_affine / rbm_sigmoid)
 * Both trainers are online (one example per step, every step reads what the previous one wrote):
 * ONE workgroup walks the examples in order.  Sampled negative ids are INPUTS (the reference draws
 * them from RandomState(123), :306); DEVICE pointers unless a parameter says "host".  Errors: the
 * FNN_ERR_* codes of fnn_hip.h; dae_last_error() has the message.
 */
#ifndef DAE_HIP_H
#define DAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

const char* dae_last_error(void);

/* One pass of sparse_da's loop over N examples (:295-332).  table [n_rows, H] f32 is CONSTANT: the
 * reference's `givens` makes the returned W the un-updated input, so only the biases learn.
 * idx [N, S] int32: the S sampled visibles of each example in line order (negative, positive, ...),
 * x [N, S] f32 their values; bhid [H] and bvis [S] (positional) are updated in place; bhid_prev [H]
 * receives bhid as it was BEFORE the last example's update (what the Theano call returns).
 * cost_sum_out (host, nullable): sum of the per-example costs.  H <= 256, S <= 32. */
int dae_sparse_epoch(const float* table, int64_t n_rows, float* bhid, float* bvis, float* bhid_prev,
                     const int32_t* idx, const float* x, int64_t N, int H, int S, float lr,
                     double* cost_sum_out, void* stream);

/* One pass of da()'s loop (:143-196) over X [N, row] f32 (already propagated through the lower
 * layers): W [row, col], bhid [col], bvis [row] updated in place by N online steps
 * W -= lr (x (x) dy + d (x) y).  skip_last_update != 0: the last example only contributes its cost
 * (the reference returns the parameters as they were before the last call's update).
 * W lives in the registers of one 1024-thread workgroup when [row][col] fits [304][128] or [208][320];
 * larger shapes (row <= 2048, col <= 1024) keep W in global memory. */
int dae_dense_epoch(float* W, float* bhid, float* bvis, const float* X, int64_t N, int row, int col,
                    float lr, int skip_last_update, double* cost_sum_out, void* stream);

/* Layer-0 propagation of da() (:166-187): out [n, H] = sigmoid(cumsum_k(sum of the rows W0[id] over
 * ALL ids of the example) + b0) -- the reference never resets its accumulator between hidden units,
 * so unit k receives the running sum over units 0..k.  ids [n, F] int32, -1 = none.  H <= 1024. */
int dae_bag_cumsum_sigmoid(const float* W0, const float* b0, int H, int64_t n_rows, const int32_t* ids,
                           int n, int F, float* out, void* stream);

/* The same three steps in float64 -- the reference's own precision (theano.config.floatX).  Its
 * lr = 0.1 online dynamics amplify a 1e-7 perturbation to O(1) within a few thousand steps at
 * 200/300/100 hidden units, so only an f64 run tracks the reference's trajectory end to end.  W of
 * the dense trainer stays in global memory here (any row <= 2048, col <= 1024); dae_dense_epoch
 * takes the same path for f32 shapes its register tilings do not hold. */
int dae_sparse_epoch_f64(const double* table, int64_t n_rows, double* bhid, double* bvis, double* bhid_prev,
                         const int32_t* idx, const double* x, int64_t N, int H, int S, double lr,
                         double* cost_sum_out, void* stream);
int dae_dense_epoch_f64(double* W, double* bhid, double* bvis, const double* X, int64_t N, int row, int col,
                        double lr, int skip_last_update, double* cost_sum_out, void* stream);
int dae_bag_cumsum_sigmoid_f64(const double* W0, const double* b0, int H, int64_t n_rows, const int32_t* ids,
                               int n, int F, double* out, void* stream);
/* out [n, b] = sigmoid(in [n, a] . W [a, b] + bias [b]): the propagation between dense layers (:183-187). */
int dae_affine_sigmoid_f64(const double* in, const double* W, const double* bias, int n, int a, int b,
                           double* out, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DAE_HIP_H */
