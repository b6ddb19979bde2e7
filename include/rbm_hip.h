/* rbm_hip.h -- C ABI of the SNN pre-training kernels in libfnn_hip.so (MI355X, gfx950).
 *
 * Replaces the NumPy CD-1 trainers of Atomu2014/deep-ctr's
 * python/sampling_based_gaussian_binary_rbm_sparse.py:
 *
 *   sparse_CDTrainer.train   :413-508   (online, one example at a time)   -> rbm_sparse_epoch
 *   CDTrainer.train          :168-291   (dense mini-batch CD-1)           -> rbm_dense_cd1
 *   the lower-layer propagation inside CDTrainer.train :198-218          -> rbm_bag_sum / rbm_affine /
 *                                                                             rbm_sigmoid
 *
 * Random numbers are INPUTS: the reference draws `rng.uniform(size=hid.shape)` from the global
 * legacy NumPy stream (:371-375, :84-90); the host draws the same numbers and passes them in, so
 * any generator can be plugged in and parity tests can replay the reference's stream.
 *
 * All pointers are DEVICE pointers unless a parameter says "host".  Functions return 0 or a
 * negative error class (the FNN_ERR_* values of fnn_hip.h); rbm_last_error() has the message.
 */
#ifndef RBM_HIP_H
#define RBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

const char* rbm_last_error(void);

/* One pass of sparse_CDTrainer.train over N examples, in file order (:423-505).
 *   W [n_vis, H] f32, visbias [n_vis], hidbias [H], wstep [S, H] (the positional momentum buffer
 *   of :411, carried across examples and epochs), all updated in place.
 *   vid [N, S] int32: the S sorted visible ids of each example, vval [N, S] uint8 their 0/1 values
 *   (the dict of :425-437), unif [N, H] f32: the uniforms of sample_hid (:441).
 *   rates = (visbias, hidbias, weight) as in :405-410; W[f] += 2 * wstep (the step is applied twice,
 *   :461-462).  sq_err_out (host, nullable): sum over examples of sum_j (vis_j - v_j)^2.
 * The trainer is online (batch = 1): ONE workgroup walks the examples in order.  H <= 256, S <= 32. */
int rbm_sparse_epoch(float* W, float* visbias, float* hidbias, float* wstep,
                     const int32_t* vid, const uint8_t* vval, const float* unif,
                     int64_t N, int H, int S, float weightcost, float rate_vis, float rate_hid,
                     float rate_w, float momentum, double* sq_err_out, void* stream);

/* Mini-batch variant of the sparse CD-1 pass -- NOT the reference's schedule (which is online; SURVEY 8d lists a
 * batched mode for throughput); for M = 1 it is the same arithmetic as rbm_sparse_epoch.  The N examples are taken in
 * mini-batches of M; every example of a mini-batch reads the parameters as they were at its start:
 *     step_e[j] as in :447-453;  W[f_ej] += 2 (momentum wstep[j] + step_e[j]);  wstep[j] = momentum wstep[j] + mean_e step_e[j];
 *     visbias[f_ej] += (v_ej - vis_ej) rate_vis;  hidbias += rate_hid sum_e (hid_e - hid2_e).
 * dW [n_vis, H] and dvis [n_vis] are scratch accumulators owned by the caller: all zero on entry, all zero on return (used by
 * the fallback below only).  Round 3: the row update sorts the mini-batch's (row, entry) pairs by row (the library's own stable
 * radix sort, a whole group of mini-batches ahead) and sums every row's run in example order, recomputing the deltas from the
 * examples' hid / hid2 / vis -- no float atomics, bit-reproducible.  H % 4 != 0 (or RBM_BATCH_ATOMICS=1) keeps the round-2 form:
 * row sums accumulated with float atomics, reproducible only up to the rounding of those sums. */
int rbm_sparse_batch(float* W, float* dW, float* visbias, float* dvis, float* hidbias, float* wstep,
                     const int32_t* vid, const uint8_t* vval, const float* unif,
                     int64_t N, int M, int H, int S, float weightcost, float rate_vis, float rate_hid,
                     float rate_w, float momentum, double* sq_err_out, void* stream);

/* Data-parallel form of rbm_sparse_batch (new: the reference is one process; SURVEY 8e "batched mode shards like A8").  Every rank
 * holds its contiguous shard of every GLOBAL mini-batch: N and M count THIS rank's examples, M_global the whole mini-batch.  Per
 * mini-batch the ranks' sums of the positional steps and of the hidden-bias terms -- one flat buffer of S * H + H floats -- are summed
 * over the ranks by `allreduce` (in place, enqueued on `stream` or completed before it returns; 0 = OK), so wstep and hidbias stay
 * identical on every rank (wstep[j] = momentum wstep[j] + sum / M_global).  Each rank applies the row updates (W, visbias) of its own
 * examples only: tables are replicated and rows that several ranks touch drift apart, as in FNN_DP_SPARSE_LOCAL.  sq_err_out is this
 * rank's share.  The online trainer (rbm_sparse_epoch) is sequential by definition and has no data-parallel form: replicas only. */
typedef int (*rbm_allreduce_fn)(void* ctx, float* buf, int64_t n_floats, void* stream);
int rbm_sparse_batch_dp(float* W, float* dW, float* visbias, float* dvis, float* hidbias, float* wstep,
                        const int32_t* vid, const uint8_t* vval, const float* unif,
                        int64_t N, int M, int M_global, int H, int S, float weightcost, float rate_vis, float rate_hid,
                        float rate_w, float momentum, rbm_allreduce_fn allreduce, void* ctx, double* sq_err_out, void* stream);

/* Dense CD-1 (RBM + CDTrainer).  The handle owns the parameters [W | visbias | hidbias]
 * (python :13-26), the momentum buffer (:166) and the work buffers for up to max_n rows. */
typedef struct rbm_handle rbm_handle;
int rbm_dense_create(int nvis, int nhid, int max_n, int precision /* FNN_PREC_* */, int device,
                     void* stream, rbm_handle** out);
int rbm_dense_destroy(rbm_handle* h);
/* HOST pointers: W [nvis, nhid], visbias [nvis], hidbias [nhid]; set also zeroes the momentum. */
int rbm_dense_set(rbm_handle* h, const float* W, const float* visbias, const float* hidbias);
int rbm_dense_get(rbm_handle* h, float* W, float* visbias, float* hidbias);
/* One mini-batch of CDTrainer.train (:219-281): X [n, nvis] f32 (already sigmoid-ed, :218),
 * unif [n, nhid] f32.  sq_err_out (host, nullable): sum (vis - X)^2 (:276-280). */
int rbm_dense_cd1(rbm_handle* h, const float* X, int n, const float* unif, float weightcost,
                  float rate_vis, float rate_hid, float rate_w, float momentum, double* sq_err_out);

/* Lower-layer propagation (:198-218).  out [n, H] = sum of the rows W0[id] of the ACTIVE ids of
 * each example (ids [n, F] int32, -1 = none) + b0;  out [n, b] = in [n, a] . W [a, b] + bias [b]
 * (no nonlinearity between stacked layers);  x = 1 / (1 + exp(-x)) in place. */
int rbm_bag_sum(const float* W0, const float* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F,
                float* out, void* stream);
int rbm_affine(const float* in, const float* W, const float* bias, int n, int a, int b, float* out,
               void* stream);
int rbm_sigmoid(float* x, int64_t count, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RBM_HIP_H */
