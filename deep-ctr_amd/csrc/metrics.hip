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
// The key sort of the METRIC pass is rocPRIM's device radix sort (a library sort for the metric pass; no hot-path
// kernel goes through a library).  The groupings of the update paths use this file's own radix sort.
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

// ------------------------------------------------------------------------------------------
// Stable LSD radix sort of independent SEGMENTS of 64-bit keys, 8 bits per pass (metrics.hip.h: radix_sort_segments).  The
// groupings this library needs are "sort by row, keep the order of arrival inside a row" over 32,768 .. a few million keys:
// the keys are generated in arrival order, so a STABLE sort on the row bits alone is the whole job (3 passes for 937,670
// rows).  Per pass three launches over (tile, segment): digit histogram of every tile of 2,048 keys; exclusive scan of a
// segment's [digit][tile] counts (one workgroup per segment); scatter, in which a key's place inside its tile's share of a digit
// is its rank among the tile's keys of that digit IN TILE ORDER: a wave owns 512 consecutive keys, a wave instruction 64
// consecutive ones, and the lanes holding the same digit find each other with eight ballots (one per digit bit) -- the lowest
// such lane bumps the wave's counter of the digit, everyone else reads its rank off the ballot mask.  No atomics on global
// memory, no dependence on scheduling: the result is bit-reproducible.
// ------------------------------------------------------------------------------------------
constexpr int RS_TILE = 2048, RS_ITEMS = 8;

__global__ __launch_bounds__(256) void k_rs_hist(const unsigned long long* __restrict__ keys, int n, int shift, int nblk, unsigned* __restrict__ hist)
{
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long* seg = keys + (size_t)blockIdx.y * n;
    const int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(unsigned)(seg[i] >> shift) & 255u], 1u);          // LDS integer atomics: counts, order-free
    }
    __syncthreads();
    hist[((size_t)blockIdx.y * nblk + blockIdx.x) * 256 + threadIdx.x] = s_h[threadIdx.x];       // [segment][tile][digit]: a wave's 64 digits are one line
}

// one workgroup per segment, thread = digit: counts [tile][digit] -> exclusive prefix in (digit, tile) order.  A thread's counts are
// 1 KB apart, a wave's 64 digits one line of a tile; eight tiles' loads in flight at a time (one load, one add per trip paid a
// memory round trip per tile: 26 us per pass at 64 tiles, profiles/r03c_rbm_kernel_stats.csv)
__global__ __launch_bounds__(256) void k_rs_scan(unsigned* __restrict__ hist, int nblk)
{
    __shared__ unsigned s_t[256];
    unsigned* h = hist + (size_t)blockIdx.x * nblk * 256 + threadIdx.x;
    unsigned tot = 0;
    for (int b0 = 0; b0 < nblk; b0 += 8) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b0 + k < nblk ? h[(size_t)(b0 + k) * 256] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += v[k];
    }
    s_t[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                       // inclusive scan over the 256 digit totals
        const unsigned v = (int)threadIdx.x >= o ? s_t[threadIdx.x - o] : 0u;
        __syncthreads();
        s_t[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s_t[threadIdx.x] - tot;
    for (int b0 = 0; b0 < nblk; b0 += 8) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b0 + k < nblk ? h[(size_t)(b0 + k) * 256] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (b0 + k < nblk) { h[(size_t)(b0 + k) * 256] = run; run += v[k]; }
    }
}

__global__ __launch_bounds__(256) void k_rs_scatter(const unsigned long long* __restrict__ keys_in, unsigned long long* __restrict__ keys_out,
                                                    int n, int shift, int nblk, const unsigned* __restrict__ hist)
{
    __shared__ unsigned s_cnt[4][256];                         // per wave: keys of each digit seen so far (then: in all)
    __shared__ unsigned s_base[256];                           // where this tile's keys of a digit start in the segment
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long* seg = keys_in + (size_t)blockIdx.y * n;
    unsigned long long* out = keys_out + (size_t)blockIdx.y * n;
#pragma unroll
    for (int w = 0; w < 4; ++w) s_cnt[w][tid] = 0;
    s_base[tid] = hist[((size_t)blockIdx.y * nblk + blockIdx.x) * 256 + tid];
    __syncthreads();
    const int base = blockIdx.x * RS_TILE + wave * (RS_TILE / 4);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long key[RS_ITEMS]; unsigned pre[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) { const int i = base + k * 64 + lane; key[k] = i < n ? seg[i] : 0ull; }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const bool valid = base + k * 64 + lane < n;
        const unsigned d = (unsigned)(key[k] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned long long bb = __ballot((d >> b) & 1u); m &= ((d >> b) & 1u) ? bb : ~bb; }
        // m: the valid lanes of this wave instruction that hold my digit (me included)
        const unsigned before = s_cnt[wave][d];               // ... in this wave's earlier instructions (LDS ops of a wave run in order)
        pre[k] = before + (unsigned)__popcll(m & lt);
        if (valid && (m & lt) == 0ull) s_cnt[wave][d] = before + (unsigned)__popcll(m);
    }
    __syncthreads();
    {   // thread = digit: the waves' totals -> where each wave's keys of the digit start inside the tile's share
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        if (base + k * 64 + lane >= n) continue;
        const unsigned d = (unsigned)(key[k] >> shift) & 255u;
        out[s_base[d] + s_cnt[wave][d] + pre[k]] = key[k];
    }
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

void group_records(hipStream_t st, const unsigned long long* sorted, int nseg, int n, int4* rec)
{
    const size_t tot = (size_t)nseg * n;
    hipLaunchKernelGGL(k_group_rec, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sorted, nseg, n, rec);
}

unsigned long long* radix_sort_segments(hipStream_t st, unsigned long long* keys, unsigned long long* tmp, unsigned* hist, int nseg, int n,
                                        int lo_bit, int hi_bit)
{
    const int nblk = (n + RS_TILE - 1) / RS_TILE;
    unsigned long long *in = keys, *out = tmp;
    for (int shift = lo_bit; shift < hi_bit; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk, nseg), dim3(256), 0, st, in, n, shift, nblk, hist);
        hipLaunchKernelGGL(k_rs_scan, dim3(nseg), dim3(256), 0, st, hist, nblk);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk, nseg), dim3(256), 0, st, in, out, n, shift, nblk, hist);
        std::swap(in, out);
    }
    return in;
}
size_t radix_sort_hist_bytes(int nseg, int n) { return (size_t)nseg * 256 * ((n + RS_TILE - 1) / RS_TILE) * sizeof(unsigned); }

int group_global(hipStream_t st, const int32_t* ids, int B, int F, int64_t n_rows, int N2, int4* rec, int* owner_cnt,
                 void** ws, size_t* ws_bytes, std::string& err)
{
    if (B < 1 || N2 < B || N2 > (1 << G_TB) || F < 1 || F > 64 || n_rows >= (int64_t)G_INV) { err = "group_global: shape out of range"; return -1; }
    const size_t n = (size_t)F * N2;
    hipError_t e;
    const size_t kb = (n * 8 + 255) / 256 * 256, need = 2 * kb + radix_sort_hist_bytes(F, N2) + 256;
    if (*ws_bytes < need) {
        if (*ws) { hipStreamSynchronize(st); hipFree(*ws); *ws = nullptr; *ws_bytes = 0; }
        e = hipMalloc(ws, need);
        if (e != hipSuccess) { err = std::string("hipMalloc (grouping workspace): ") + hipGetErrorString(e); return -1; }
        *ws_bytes = need;
    }
    unsigned long long* keys = static_cast<unsigned long long*>(*ws);
    unsigned long long* other = reinterpret_cast<unsigned long long*>(static_cast<char*>(*ws) + kb);
    unsigned* hist = reinterpret_cast<unsigned*>(static_cast<char*>(*ws) + 2 * kb);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_group_keys, dim3(nb), dim3(256), 0, st, ids, B, F, n_rows, N2, keys, owner_cnt);
    // every field is a segment of N2 keys generated in example order: a stable sort on the row bits alone leaves (row, t) order.
    // Bits: enough to tell n_rows apart from every valid row -- the all-ones row of an invalid entry then sorts behind them.
    int rbits = 1;
    while (((int64_t)1 << rbits) <= n_rows) ++rbits;
    const unsigned long long* sorted = radix_sort_segments(st, keys, other, hist, F, N2, G_TB, G_TB + rbits);
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
