// metrics.hip.h -- device-side evaluation metrics shared by the C ABIs (internal, not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace fnn {
// AUC (ties at 1/2, the trapezoid of sklearn.metrics.roc_auc_score), RMSE and logloss
// (sklearn.metrics.log_loss: p clipped to [eps, 1 - eps], eps = 2^-52) of p [n] against labels
// y [n] (0 / non-zero), both DEVICE pointers.  out = {auc, rmse, logloss, n_pos}.  Synchronises `st`.
// Returns 0, -1 on a HIP error (err set), -2 when only one class is present (auc undefined; the
// other two are still written).
int device_metrics(hipStream_t st, const float* p, const int32_t* y, int64_t n, double out[4], std::string& err);
}  // namespace fnn
