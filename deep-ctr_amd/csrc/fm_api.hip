// fm_api.hip -- factorisation-machine pre-training on gfx950 behind include/fm_hip.h (row N3):
// python/FM.py:55-64 (factorization), :36-41 (loss) and plain SGD, on the FNN path's building
// blocks: padded 64-byte rows, the split sort + two-level segmented sparse-row update.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fm_hip.h"
#include "../../include/fnn_hip.h"
#include "fnn_step_kernels.hip.h"

using namespace fnn;

namespace {

thread_local std::string g_fm_err;
inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// 16 lanes per example (lane = field): each lane loads its field's 64-byte row, the field sums
// S_l = sum_f v_f[l] are 4-step shuffle reductions inside the 16-lane group.
struct FmArgs {
    const int32_t* ids; const float* y; int B, F, K; const float* table16; int64_t n_rows; const float* b;
    float scale, dscale; int train; float* gxp; int K1p; float* p_out; float* loss_t; float* gb_part; int* err;
    bool wt;                   // gx' written through (see store4_wt in fnn_kernels.hip.h)
};

__device__ __forceinline__ void fm_body(const FmArgs& a, const int blk, float* s_gb)
{
    const int tid = threadIdx.x, f = tid & 15, grp = tid >> 4;
    const int t = blk * 16 + grp;
    int64_t id = -1;
    if (t < a.B && f < a.F) {
        id = a.ids[(size_t)t * a.F + f];
        if (id < -1 || id >= a.n_rows) { atomicOr(a.err, 1); id = -1; }
    }
    float r[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) v = *reinterpret_cast<const float4*>(a.table16 + (size_t)id * SLOT + 4 * q);
        r[4 * q] = v.x * a.scale; r[4 * q + 1] = v.y * a.scale; r[4 * q + 2] = v.z * a.scale; r[4 * q + 3] = v.w * a.scale;
    }
    // yhat = b + sum_f w_f + 1/2 (sum_l S_l^2 - sum_f sum_l v_f[l]^2)                     (:56-63)
    float lin = r[0], sq = 0.f, S[16];
#pragma unroll
    for (int l = 1; l < 16; ++l) { S[l] = r[l]; sq = fmaf(r[l], r[l], sq); }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        lin += __shfl_xor(lin, o, 16); sq += __shfl_xor(sq, o, 16);
#pragma unroll
        for (int l = 1; l < 16; ++l) S[l] += __shfl_xor(S[l], o, 16);
    }
    float ss = 0.f;
#pragma unroll
    for (int l = 1; l < 16; ++l) ss = fmaf(S[l], S[l], ss);
    const float z = *a.b + lin + 0.5f * (ss - sq);
    const float p = 1.0f / (1.0f + expf(-z));
    float delta = 0.f;
    if (t < a.B) {
        if (a.p_out && f == 0) a.p_out[t] = p;
        if (a.train) {
            const float yy = a.y[t];
            delta = (p - yy) * a.dscale;                         // dscale = 1 (sum) or 1/B (mean)
            if (f == 0) a.loss_t[t] = fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
        }
    } else if (a.train && f == 0) a.loss_t[t] = 0.f;
    if (!a.train) return;
    // d yhat / d w_f = 1 ; d yhat / d v_f[l] = S_l - v_f[l]   (x = 1)
    float g[16];
    g[0] = (id >= 0) ? delta : 0.f;
#pragma unroll
    for (int l = 1; l < 16; ++l) g[l] = (id >= 0 && l < a.K) ? delta * (S[l] - r[l]) : 0.f;
    float* out = a.gxp + (size_t)t * a.K1p + f * SLOT;
    if (f < a.F) {
#pragma unroll
        for (int q = 0; q < 4; ++q) store16_sel(a.wt, out + 4 * q, make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]));   // (written through: FM_WT=0 for plain stores)
    }
    if (f == 0) s_gb[grp] = delta;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int i = 0; i < 16; ++i) s += s_gb[i]; a.gb_part[blk] = s; }
}
__global__ __launch_bounds__(256) void k_fm(const FmArgs a)
{
    __shared__ float s_gb[16];
    fm_body(a, blockIdx.x, s_gb);
}
// A training step is four launches: run sorts of the batch's (row, t) keys; their rank merge BESIDE the forward + gradients
// (both need only the ids: the merge takes 16 F workgroups, the examples the rest); level-1 sparse-row update; level-2 update
// BESIDE the bias / loss tail.  As six launches in a row (sort, sort, forward, scatter, scatter, tail) the step took 49.6 us.
template <typename KT>
__global__ __launch_bounds__(256) void k_fm_merge_fwd(const SortArgs so, const FmArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float s_gb[16];
    if ((int)blockIdx.x < so.nblk) { sortB_body<KT>(so, blockIdx.x, smem); return; }
    fm_body(a, (int)blockIdx.x - so.nblk, s_gb);
}

// b <- b (1 - lr lambda) - lr sum(delta); loss sum (fixed-shape tree)
__device__ __forceinline__ void fm_tail_body(float* b, const float* gb_part, int n, float lr, float lambda, const float* loss_t, int Ba, float lscale,
                                             float* loss_out, float* sl)
{   // both sums as 256 strided partial sums and a fixed-shape tree (one thread walking the n partials paid a memory round trip
    // per element: 19 us for 256 of them)
    float v = strided_sum256(gb_part, n);
    sl[threadIdx.x] = v; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sl[threadIdx.x] += sl[threadIdx.x + o]; __syncthreads(); }
    const float gsum = sl[0];
    __syncthreads();
    v = strided_sum256(loss_t, Ba);
    sl[threadIdx.x] = v; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sl[threadIdx.x] += sl[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { *b = *b * (1.0f - lr * lambda) - lr * gsum; *loss_out = sl[0] * lscale; }
}
__global__ __launch_bounds__(256) void k_fm_scat2_tail(const ScatArgs sa, float* b, const float* gb_part, int n, float lr, float lambda,
                                                       const float* loss_t, int Ba, float lscale, float* loss_out)
{
    __shared__ double s_sum[16][16];
    if (blockIdx.x == 0) { fm_tail_body(b, gb_part, n, lr, lambda, loss_t, Ba, lscale, loss_out, reinterpret_cast<float*>(&s_sum[0][0])); return; }
    scat2_body(sa, (int)blockIdx.x - 1, (int)gridDim.x - 1, s_sum);
}

__global__ void k_fm_rescale(float* table16, size_t n, float s)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table16[i] *= s;
}

}  // namespace

struct fm_handle {
    std::string err; int dev = 0; hipStream_t st = nullptr; bool own_stream = false;
    int F = 0, K = 0, Bmax = 0, K1p = 0;
    float* table16 = nullptr; int64_t n_rows = 0; float* b = nullptr; double scale = 1.0;
    float *gxp = nullptr, *loss_t = nullptr, *gb_part = nullptr, *loss_dev = nullptr; int* err_flag = nullptr;
    int4* rec = nullptr; double* part = nullptr; int4* owners = nullptr; int* owner_cnt = nullptr; void* skeys = nullptr;
    double* cpow1 = nullptr; bool key64 = true;
};

#define MHK(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return FNN_ERR_HIP; } } while (0)
#define MFAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

namespace {

int fold_scale(fm_handle* h)          // fold the lazy decay back into the rows
{
    if (h->scale == 1.0 || !h->table16) return FNN_OK;
    const size_t n = (size_t)h->n_rows * SLOT;
    hipLaunchKernelGGL(k_fm_rescale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->table16, n, (float)h->scale);
    MHK(h, hipGetLastError());
    h->scale = 1.0;
    return FNN_OK;
}

int fm_run(fm_handle* h, const int32_t* ids, const float* y, int B, float lr, float lambda, int reduce_mean, float* p_out, bool train)
{
    const int Ba = rup(B, 16), F = h->F;
    FmArgs a{ids, y, B, F, h->K, h->table16, h->n_rows, h->b, (float)h->scale, reduce_mean ? 1.0f / (float)B : 1.0f, train ? 1 : 0,
             h->gxp, h->K1p, p_out, h->loss_t, h->gb_part, h->err_flag, !(getenv("FM_WT") && atoi(getenv("FM_WT")) == 0)};
    if (!train) {
        hipLaunchKernelGGL(k_fm, dim3(Ba / 16), dim3(256), 0, h->st, a);
        MHK(h, hipGetLastError());
        return FNN_OK;
    }
    {
        SortArgs so{ids, B, F, h->n_rows, h->rec, h->owner_cnt, 4 * F, h->skeys};
        SortArgs sb = so; sb.nblk = 16 * F;
        if (h->key64) {
            hipLaunchKernelGGL((k_sortA<unsigned long long>), dim3(4 * F), dim3(256), 0, h->st, so);
            hipLaunchKernelGGL((k_fm_merge_fwd<unsigned long long>), dim3(16 * F + Ba / 16), dim3(256), SORT_N * 8, h->st, sb, a);
        } else {
            hipLaunchKernelGGL((k_sortA<unsigned>), dim3(4 * F), dim3(256), 0, h->st, so);
            hipLaunchKernelGGL((k_fm_merge_fwd<unsigned>), dim3(16 * F + Ba / 16), dim3(256), SORT_N * 4, h->st, sb, a);
        }
    }
    // dense L2 decay of the whole table = one scalar; touched rows: stored -= lr * g / scale
    h->scale *= 1.0 - (double)lr * (double)lambda;
    ScatArgs sa{h->rec, SORT_N, F, h->K, h->gxp, h->K1p, h->cpow1, (double)lr / h->scale, h->table16, h->part, h->owner_cnt,
                h->owners, SLOT};
    hipLaunchKernelGGL(k_scat1, dim3(F * SORT_N / 256), dim3(256), 0, h->st, sa);
    hipLaunchKernelGGL(k_fm_scat2_tail, dim3(1 + 256), dim3(256), 0, h->st, sa, h->b, h->gb_part, Ba / 16, lr, lambda, h->loss_t, Ba,
                       reduce_mean ? 1.0f / (float)B : 1.0f, h->loss_dev);
    MHK(h, hipGetLastError());
    if (h->scale < 5.96e-8 || h->scale > 1.0) return fold_scale(h);
    return FNN_OK;
}

}  // namespace

extern "C" {

const char* fm_last_error(const fm_handle* h) { return h ? h->err.c_str() : g_fm_err.c_str(); }

int fm_create(int n_fields, int k, int max_batch, int device, void* stream, fm_handle** out)
{
    if (!out) { g_fm_err = "null argument"; return FNN_ERR_ARG; }
    *out = nullptr;
    if (n_fields < 1 || n_fields > 16 || k < 1 || k > 16 || max_batch < 1 || max_batch > SORT_N) {
        g_fm_err = "need 1 <= n_fields <= 16, 1 <= k <= 16, 1 <= max_batch <= 4096"; return FNN_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_fm_err = "no HIP device (libfnn_hip.so has no CPU fallback)"; return FNN_ERR_HIP; }
    fm_handle* h = new fm_handle();
    h->dev = device; h->F = n_fields; h->K = k; h->Bmax = max_batch; h->K1p = 16 * SLOT;
    auto fail = [&](int code) { g_fm_err = h->err; fm_destroy(h); return code; };
#define FK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e2_); return fail(FNN_ERR_HIP); } } while (0)
    FK(hipSetDevice(h->dev));
    if (stream) h->st = (hipStream_t)stream; else { FK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    auto al = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, h->st); return e; };
    const size_t Ba = rup(h->Bmax, 16);
    FK(al((void**)&h->gxp, Ba * h->K1p * 4)); FK(al((void**)&h->loss_t, Ba * 4)); FK(al((void**)&h->gb_part, (Ba / 16) * 4));
    FK(al((void**)&h->loss_dev, 4)); FK(al((void**)&h->b, 4)); FK(al((void**)&h->err_flag, 4));
    FK(al((void**)&h->rec, (size_t)h->F * SORT_N * sizeof(int4))); FK(al((void**)&h->part, (size_t)h->F * (SORT_N / 16) * 2 * SLOT * 8));
    FK(al((void**)&h->owners, (size_t)h->F * (SORT_N / 16) * sizeof(int4))); FK(al((void**)&h->owner_cnt, 4));
    FK(al(&h->skeys, (size_t)h->F * SORT_N * 8));
    {
        std::vector<double> ones(SORT_N + 1, 1.0);                   // no per-touch decay: every power is 1
        FK(hipMalloc((void**)&h->cpow1, ones.size() * 8));
        FK(hipMemcpy(h->cpow1, ones.data(), ones.size() * 8, hipMemcpyHostToDevice));
    }
    FK(hipStreamSynchronize(h->st));
#undef FK
    *out = h;
    return FNN_OK;
}

int fm_destroy(fm_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    hipSetDevice(h->dev);
    if (h->st) hipStreamSynchronize(h->st);
    void* ptrs[] = {h->table16, h->b, h->gxp, h->loss_t, h->gb_part, h->loss_dev, h->err_flag, h->rec, h->part, h->owners, h->owner_cnt,
                    h->skeys, h->cpow1};
    for (void* p : ptrs) if (p) hipFree(p);
    if (h->own_stream && h->st) hipStreamDestroy(h->st);
    delete h;
    return FNN_OK;
}

int fm_sync(fm_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    int flag = 0;
    MHK(h, hipMemcpyAsync(&flag, h->err_flag, 4, hipMemcpyDeviceToHost, h->st));
    MHK(h, hipStreamSynchronize(h->st));
    if (flag) { MHK(h, hipMemsetAsync(h->err_flag, 0, 4, h->st)); MFAIL(h, FNN_ERR_RANGE, "feature id outside [-1, n_rows)"); }
    return FNN_OK;
}

int fm_set_table(fm_handle* h, const float* rows, int64_t n_rows)
{
    if (!h || !rows || n_rows < 1 || n_rows >= (1ll << 31)) return FNN_ERR_ARG;
    MHK(h, hipSetDevice(h->dev));
    MHK(h, hipStreamSynchronize(h->st));
    if (h->table16) { hipFree(h->table16); h->table16 = nullptr; }
    MHK(h, hipMalloc((void**)&h->table16, (size_t)n_rows * SLOT * 4));
    float* tmp = nullptr;
    MHK(h, hipMalloc((void**)&tmp, (size_t)n_rows * h->K * 4));
    MHK(h, hipMemcpy(tmp, rows, (size_t)n_rows * h->K * 4, hipMemcpyHostToDevice));
    const size_t n = (size_t)n_rows * SLOT;
    hipLaunchKernelGGL(k_pack_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, tmp, n_rows, h->K, SLOT, h->table16);
    MHK(h, hipStreamSynchronize(h->st));
    hipFree(tmp);
    h->n_rows = n_rows; h->scale = 1.0;
    h->key64 = (unsigned long long)n_rows * SORT_N > 0xFFFFFFFFull;
    return FNN_OK;
}

static int fm_rows(fm_handle* h, const int64_t* row_ids, int64_t n, float* out)
{
    if (!h->table16) MFAIL(h, FNN_ERR_STATE, "fm_set_table has not been called");
    MHK(h, hipSetDevice(h->dev));
    int rc = fold_scale(h);
    if (rc != FNN_OK) return rc;
    int64_t* di = nullptr; float* dout = nullptr;
    if (row_ids) { MHK(h, hipMalloc((void**)&di, n * 8)); MHK(h, hipMemcpy(di, row_ids, n * 8, hipMemcpyHostToDevice)); }
    MHK(h, hipMalloc((void**)&dout, (size_t)n * h->K * 4));
    const size_t cnt = (size_t)n * h->K;
    hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->st, h->table16, di, n, h->n_rows, h->K, SLOT,
                       dout, h->err_flag);
    MHK(h, hipMemcpyAsync(out, dout, cnt * 4, hipMemcpyDeviceToHost, h->st));
    rc = fm_sync(h);
    if (di) hipFree(di);
    hipFree(dout);
    return rc;
}

int fm_get_table(fm_handle* h, float* rows_out) { if (!h || !rows_out) return FNN_ERR_ARG; return fm_rows(h, nullptr, h->n_rows, rows_out); }
int fm_get_rows(fm_handle* h, const int64_t* row_ids, int64_t n, float* out)
{
    if (!h || !row_ids || !out || n < 1) return FNN_ERR_ARG;
    return fm_rows(h, row_ids, n, out);
}

int fm_set_b(fm_handle* h, float b)
{
    if (!h) return FNN_ERR_ARG;
    MHK(h, hipSetDevice(h->dev)); MHK(h, hipStreamSynchronize(h->st));
    MHK(h, hipMemcpy(h->b, &b, 4, hipMemcpyHostToDevice));
    return FNN_OK;
}
int fm_get_b(fm_handle* h, float* b)
{
    if (!h || !b) return FNN_ERR_ARG;
    MHK(h, hipSetDevice(h->dev)); MHK(h, hipStreamSynchronize(h->st));
    MHK(h, hipMemcpy(b, h->b, 4, hipMemcpyDeviceToHost));
    return FNN_OK;
}

int fm_train_step(fm_handle* h, const int32_t* ids, const float* y, int B, float lr, float lambda, int reduce_mean, float* p_out,
                  float* loss_out)
{
    if (!h || !ids || !y) return FNN_ERR_ARG;
    if (B < 1 || B > h->Bmax) MFAIL(h, FNN_ERR_ARG, "B must be in [1, max_batch]");
    if (!h->table16) MFAIL(h, FNN_ERR_STATE, "fm_set_table has not been called");
    if (!(lr * lambda < 1.0f) || lambda < 0.f) MFAIL(h, FNN_ERR_ARG, "need 0 <= lr * lambda < 1");
    MHK(h, hipSetDevice(h->dev));
    int rc = fm_run(h, ids, y, B, lr, lambda, reduce_mean, p_out, true);
    if (rc != FNN_OK) return rc;
    if (loss_out) {
        MHK(h, hipMemcpyAsync(loss_out, h->loss_dev, 4, hipMemcpyDeviceToHost, h->st));
        return fm_sync(h);
    }
    return FNN_OK;
}

int fm_predict(fm_handle* h, const int32_t* ids, int B, float* p_out)
{
    if (!h || !ids || !p_out) return FNN_ERR_ARG;
    if (B < 1 || B > h->Bmax) MFAIL(h, FNN_ERR_ARG, "B must be in [1, max_batch]");
    if (!h->table16) MFAIL(h, FNN_ERR_STATE, "fm_set_table has not been called");
    MHK(h, hipSetDevice(h->dev));
    return fm_run(h, ids, nullptr, B, 0.f, 0.f, 0, p_out, false);
}

}  // extern "C"
