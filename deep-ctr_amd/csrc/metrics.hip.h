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

// Grouping of a GLOBAL batch for the exact data-parallel mode (fnn_step_scatter_global beyond the 16,384 keys the one-workgroup
// bitonic sort holds in LDS): per field f, the (row, t) pairs ids[t][f], t < B, sorted by (row, t) into
// rec[f * N2 + pos] = {row, t, s, e} -- [s, e) = the positions of the row's run inside the field -- and {-1, 0, 0, 0} for the
// N2 - (valid entries) slots that follow (ids outside [0, n_rows), t >= B).  N2 >= B, N2 <= 2^20.  The F * N2 64-bit keys are
// ordered by radix_sort_segments below (round 2 used rocPRIM here).  `ws` / `ws_bytes`: workspace owned by the caller,
// grown here when too small (hipMalloc: the first call synchronises).  Also zeroes *owner_cnt.  Returns 0 or -1 (err set).
int group_global(hipStream_t st, const int32_t* ids, int B, int F, int64_t n_rows, int N2, int4* rec, int* owner_cnt,
                 void** ws, size_t* ws_bytes, std::string& err);

// Stable LSD radix sort (8 bits per pass, bits [lo_bit, hi_bit)) of `nseg` independent segments of `n` 64-bit keys each
// (segment g = keys[g * n .. (g + 1) * n)), ping-ponging between `keys` and `tmp` (same size); `hist`: radix_sort_hist_bytes(nseg, n)
// bytes of workspace.  Enqueues 3 launches per pass on `st`; returns the buffer that holds the result.  Stable: keys that agree on
// the sorted bits keep their order -- generate keys in arrival order and sort on the row bits alone.
unsigned long long* radix_sort_segments(hipStream_t st, unsigned long long* keys, unsigned long long* tmp, unsigned* hist, int nseg, int n,
                                        int lo_bit, int hi_bit);
size_t radix_sort_hist_bytes(int nseg, int n);
// Sorted keys `row << 20 | index` (row = 2^31 - 1: invalid, sorted last) of `nseg` segments of `n` -> rec[g * n + pos] =
// {row, index, s, e}, [s, e) = the positions of the row's run inside its segment; {-1, 0, 0, 0} for invalid entries.
constexpr int GROUP_INDEX_BITS = 20;
constexpr unsigned long long GROUP_INVALID_ROW = (1ull << 31) - 1;
void group_records(hipStream_t st, const unsigned long long* sorted, int nseg, int n, int4* rec);
}  // namespace fnn
