// metrics.hip -- A10, the evaluation metrics of python/FNN_wnzh.py:193-221 (get_err_bat) and
// python/SNN_RBM.py:162-198 (auc_rmse) on the device: roc_auc_score, sqrt(mean_squared_error) and
// the log_loss of python/baseline.py:427-429, over predictions that never leave HBM.
//
// AUC is exact integer arithmetic.  Keys (label << 31 | bits of p), p >= 0, sort to [negatives by
// p | positives by p]; for every positive, lb / ub = number of negatives with p' < p / p' <= p by
// binary search in the negative range, and  AUC = sum(lb + ub) / (2 n_pos n_neg)  -- the Mann-Whitney
// statistic with ties counted 1/2, which is what the trapezoid rule over distinct thresholds gives.
// The sums are 64-bit integer atomics: order-independent, bitwise reproducible.  RMSE and logloss
// accumulate in f64 with a fixed-shape tree per block and a fixed-order sum of the block partials.
// The key sort is rocPRIM's device radix sort (a library sort for the metric pass; no hot-path
// kernel goes through a library).
#include "metrics.hip.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <vector>

namespace fnn {
namespace {

constexpr int MB = 256;          // threads per block
constexpr int ITEMS = 8;         // examples per thread

__global__ __launch_bounds__(MB) void k_metric_keys(const float* __restrict__ p, const int32_t* __restrict__ y, int64_t n,
                                                    uint32_t* __restrict__ keys, double* __restrict__ part /*[nblk][2]*/,
                                                    unsigned long long* __restrict__ n_pos)
{
    __shared__ double s_se[MB], s_ll[MB];
    __shared__ unsigned s_np;
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    const double eps = 2.220446049250313e-16;
    double se = 0.0, ll = 0.0; unsigned np_ = 0;
    const int64_t base = (int64_t)blockIdx.x * MB * ITEMS;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t i = base + (int64_t)k * MB + threadIdx.x;
        if (i < n) {
            const float pf = p[i];
            const bool pos = y[i] != 0;
            keys[i] = (pos ? 0x80000000u : 0u) | (__float_as_uint(pf) & 0x7FFFFFFFu);
            const double pd = (double)pf, yd = pos ? 1.0 : 0.0;
            se += (yd - pd) * (yd - pd);
            const double pc = fmin(fmax(pd, eps), 1.0 - eps);
            ll -= pos ? log(pc) : log(1.0 - pc);
            np_ += pos ? 1u : 0u;
        }
    }
    s_se[threadIdx.x] = se; s_ll[threadIdx.x] = ll;
    atomicAdd(&s_np, np_);
    __syncthreads();
    for (int o = MB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_se[threadIdx.x] += s_se[threadIdx.x + o]; s_ll[threadIdx.x] += s_ll[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x] = s_se[0]; part[2 * (size_t)blockIdx.x + 1] = s_ll[0];
        if (s_np) atomicAdd(n_pos, (unsigned long long)s_np);
    }
}

__global__ __launch_bounds__(MB) void k_metric_auc(const uint32_t* __restrict__ sorted, int64_t n, int64_t n_neg,
                                                   unsigned long long* __restrict__ acc)
{
    __shared__ unsigned long long s_a[MB];
    unsigned long long a = 0;
    for (int64_t j = n_neg + (int64_t)blockIdx.x * MB + threadIdx.x; j < n; j += (int64_t)gridDim.x * MB) {
        const uint32_t v = sorted[j] & 0x7FFFFFFFu;
        int64_t lo = 0, hi = n_neg;                         // first negative with key >= v
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] < v) lo = mid + 1; else hi = mid; }
        const int64_t lb = lo;
        hi = n_neg;                                         // first negative with key > v
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] <= v) lo = mid + 1; else hi = mid; }
        a += (unsigned long long)(lb + lo);
    }
    s_a[threadIdx.x] = a;
    __syncthreads();
    for (int o = MB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_a[threadIdx.x] += s_a[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && s_a[0]) atomicAdd(acc, s_a[0]);
}

// ---- grouping of a global batch (see metrics.hip.h): key = f << 51 | row << 20 | t, invalid rows = 2^31 - 1
constexpr int G_TB = 20, G_RB = 31;
constexpr unsigned long long G_INV = (1ull << G_RB) - 1;

__global__ __launch_bounds__(256) void k_group_keys(const int32_t* __restrict__ ids, int B, int F, int64_t n_rows, int N2,
                                                    unsigned long long* __restrict__ keys, int* __restrict__ owner_cnt)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid == 0) *owner_cnt = 0;
    if (gid >= (size_t)F * N2) return;
    const int f = (int)(gid / N2), t = (int)(gid % N2);
    unsigned long long row = G_INV;
    if (t < B) {
        const int64_t id = ids[(size_t)t * F + f];
        if (id >= 0 && id < n_rows) row = (unsigned long long)id;
    }
    keys[gid] = ((unsigned long long)f << (G_RB + G_TB)) | (row << G_TB) | (unsigned long long)t;
}

__global__ __launch_bounds__(256) void k_group_rec(const unsigned long long* __restrict__ sorted, int F, int N2, int4* __restrict__ rec)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)F * N2) return;
    const int f = (int)(gid / N2), pos = (int)(gid % N2);
    const unsigned long long key = sorted[gid];
    const unsigned long long row = (key >> G_TB) & G_INV;
    int4 r = make_int4(-1, 0, 0, 0);
    if (row != G_INV) {
        const unsigned long long* q = sorted + (size_t)f * N2;
        const unsigned long long lo_key = key & ~((1ull << G_TB) - 1), hi_key = lo_key + (1ull << G_TB);
        int lo = 0, hi = pos;                                   // first index with key >= lo_key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < lo_key) lo = mid + 1; else hi = mid; }
        const int s = lo;
        lo = pos + 1; hi = N2;                                  // first index with key >= hi_key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < hi_key) lo = mid + 1; else hi = mid; }
        r = make_int4((int)row, (int)(key & ((1ull << G_TB) - 1)), s, lo);
    }
    rec[gid] = r;
}

}  // namespace

int group_global(hipStream_t st, const int32_t* ids, int B, int F, int64_t n_rows, int N2, int4* rec, int* owner_cnt,
                 void** ws, size_t* ws_bytes, std::string& err)
{
    if (B < 1 || N2 < B || N2 > (1 << G_TB) || F < 1 || F > 64 || n_rows >= (int64_t)G_INV) { err = "group_global: shape out of range"; return -1; }
    const size_t n = (size_t)F * N2;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0,
                                            G_TB + G_RB + 6, st);
    if (e != hipSuccess) { err = std::string("rocprim::radix_sort_keys (size query): ") + hipGetErrorString(e); return -1; }
    const size_t kb = (n * 8 + 255) / 256 * 256, need = 2 * kb + tmp_bytes + 256;
    if (*ws_bytes < need) {
        if (*ws) { hipStreamSynchronize(st); hipFree(*ws); *ws = nullptr; *ws_bytes = 0; }
        e = hipMalloc(ws, need);
        if (e != hipSuccess) { err = std::string("hipMalloc (grouping workspace): ") + hipGetErrorString(e); return -1; }
        *ws_bytes = need;
    }
    unsigned long long* keys = static_cast<unsigned long long*>(*ws);
    unsigned long long* sorted = reinterpret_cast<unsigned long long*>(static_cast<char*>(*ws) + kb);
    void* tmp = static_cast<char*>(*ws) + 2 * kb;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_group_keys, dim3(nb), dim3(256), 0, st, ids, B, F, n_rows, N2, keys, owner_cnt);
    e = rocprim::radix_sort_keys(tmp, tmp_bytes, keys, sorted, n, 0, G_TB + G_RB + 6, st);
    if (e != hipSuccess) { err = std::string("rocprim::radix_sort_keys: ") + hipGetErrorString(e); return -1; }
    hipLaunchKernelGGL(k_group_rec, dim3(nb), dim3(256), 0, st, sorted, F, N2, rec);
    e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("group_global launch: ") + hipGetErrorString(e); return -1; }
    return 0;
}

int device_metrics(hipStream_t st, const float* p, const int32_t* y, int64_t n, double out[4], std::string& err)
{
#define MK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); goto done; } } while (0)
    int rc = -1;
    uint32_t *keys = nullptr, *sorted = nullptr; double* part = nullptr; unsigned long long* cnt = nullptr; void* tmp = nullptr;
    size_t tmp_bytes = 0;
    const int64_t nblk = (n + (int64_t)MB * ITEMS - 1) / ((int64_t)MB * ITEMS);
    std::vector<double> hp((size_t)nblk * 2);
    unsigned long long hc[2] = {0, 0};
    if (n < 1 || n > (int64_t)1 << 31) { err = "n outside [1, 2^31]"; return -1; }
    MK(hipMalloc((void**)&keys, (size_t)n * 4)); MK(hipMalloc((void**)&sorted, (size_t)n * 4));
    MK(hipMalloc((void**)&part, (size_t)nblk * 16)); MK(hipMalloc((void**)&cnt, 16));
    MK(hipMemsetAsync(cnt, 0, 16, st));
    hipLaunchKernelGGL(k_metric_keys, dim3((unsigned)nblk), dim3(MB), 0, st, p, y, n, keys, part, cnt);
    MK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, sorted, (size_t)n, 0, 32, st));
    MK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    MK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, sorted, (size_t)n, 0, 32, st));
    MK(hipMemcpyAsync(hc, cnt, 8, hipMemcpyDeviceToHost, st));
    MK(hipStreamSynchronize(st));
    {
        const int64_t n_pos = (int64_t)hc[0], n_neg = n - n_pos;
        if (n_pos > 0 && n_neg > 0) {
            const int64_t want = (n_pos + MB - 1) / MB;
            hipLaunchKernelGGL(k_metric_auc, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(MB), 0, st, sorted, n, n_neg, cnt + 1);
        }
        MK(hipMemcpyAsync(hc + 1, cnt + 1, 8, hipMemcpyDeviceToHost, st));
        MK(hipMemcpyAsync(hp.data(), part, (size_t)nblk * 16, hipMemcpyDeviceToHost, st));
        MK(hipStreamSynchronize(st));
        double se = 0.0, ll = 0.0;
        for (int64_t b = 0; b < nblk; ++b) { se += hp[2 * b]; ll += hp[2 * b + 1]; }
        out[1] = std::sqrt(se / (double)n); out[2] = ll / (double)n; out[3] = (double)n_pos;
        if (n_pos > 0 && n_neg > 0) { out[0] = (double)hc[1] / (2.0 * (double)n_pos * (double)n_neg); rc = 0; }
        else { out[0] = std::nan(""); err = "only one class present in y (roc_auc_score raises ValueError)"; rc = -2; }
    }
done:
    for (void* q : {(void*)keys, (void*)sorted, (void*)part, (void*)cnt, tmp}) if (q) hipFree(q);
    return rc;
#undef MK
}

}  // namespace fnn
