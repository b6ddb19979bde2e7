// dae_api.hip -- SNN-DAE pre-training on gfx950 behind include/dae_hip.h: the online (batch = 1)
// denoising-autoencoder trainers of python/sampling_based_denosing_autoencoder.py.
//
// Both trainers are sequential by definition (example n reads the parameters example n-1 wrote), so
// each is ONE persistent workgroup that keeps the parameters on chip for the whole pass:
//   k_dae_sparse  thread = hidden unit; the S gathered rows of the (constant) table are registers,
//                 the S reconstructions reduce through LDS; the next example's rows are fetched while
//                 the current one computes.
//   k_dae_dense   1024 threads hold W [row][col] in registers -- wave w owns rows w*RPW.., lane l owns
//                 columns l, l+64, ..  -- so the row sums (z = W y) are wave reductions and the column
//                 sums (x W, d W) are a 16-partial LDS reduction; four barriers per example.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/dae_hip.h"
#include "../../include/fnn_hip.h"

namespace {

thread_local std::string g_err;
#define DHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e_); return FNN_ERR_HIP; } } while (0)
#define DFAIL(code, msg) do { g_err = (msg); return (code); } while (0)

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ inline float wave_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
// -x log z - (1-x) log(1-z), the terms with a zero factor dropped as Theano's 0 * log(.) = 0 would
// only differ at z in {0, 1}
__device__ inline float xent(float x, float z) { return -(x * logf(z) + (1.0f - x) * logf(1.0f - z)); }

struct SparseArgs {
    const float* table; int64_t n_rows; float *bhid, *bvis, *bhid_prev; const int32_t* idx; const float* x;
    int64_t N; int H, S; float lr; double* cost; int* err;
};

__global__ __launch_bounds__(256) void k_dae_sparse(const SparseArgs a)
{
    __shared__ float s_w[32][257];
    __shared__ float s_y[256], s_d[32], s_x[32], s_c[32];
    const int tid = threadIdx.x, H = a.H, S = a.S;
    const bool act = tid < H;
    float bh = act ? a.bhid[tid] : 0.f, bh_prev = bh;
    float bv = (tid < S) ? a.bvis[tid] : 0.f;                    // thread j < S also owns positional visible bias j
    double cost = 0.0;
    float wn[32];                                                 // rows of the NEXT example
    auto fetch = [&](int64_t n) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float v = 0.f;
            if (act && j < S && n < a.N) {
                const int64_t id = a.idx[n * S + j];
                if (id >= 0 && id < a.n_rows) v = a.table[(size_t)id * H + tid]; else if (tid == 0) atomicOr(a.err, 1);
            }
            wn[j] = v;
        }
    };
    fetch(0);
    for (int64_t n = 0; n < a.N; ++n) {
        float wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = wn[j];
        if (tid < 32) s_x[tid] = tid < S ? a.x[n * S + tid] : 0.f;
        fetch(n + 1);                                             // in flight under this example's arithmetic
        __syncthreads();
        float z = bh;                                             // y = sigmoid(x W + b)            (:94)
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fmaf(s_x[j], wc[j], z);
        const float y = act ? sigm(z) : 0.f;
        s_y[tid] = y;
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {   // z_j = sigmoid(y . W[j,:] + bvis_j)                                                    (:97)
            const int j = tid >> 3, seg = tid & 7;
            float acc = 0.f;
            for (int i = seg; i < H; i += 8) acc = fmaf(s_y[i], s_w[j][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) s_c[j] = acc;
        }
        __syncthreads();
        if (tid < 32) {
            float d = 0.f;
            if (tid < S) {
                const float zj = sigm(s_c[tid] + bv), xj = s_x[tid];
                d = zj - xj;
                s_c[tid] = xent(xj, zj);
                bv -= a.lr * d;                                   // b' <- b' - lr (z - x)
            } else s_c[tid] = 0.f;
            s_d[tid] = d;
        }
        __syncthreads();
        float dy = 0.f;                                           // (d W) * y (1 - y)
#pragma unroll
        for (int j = 0; j < 32; ++j) dy = fmaf(s_d[j], wc[j], dy);
        dy *= y * (1.0f - y);
        bh_prev = bh;
        bh -= a.lr * dy;
        if (tid == 0) { float c = 0.f; for (int j = 0; j < 32; ++j) c += s_c[j]; cost += (double)c; }
    }
    if (act) { a.bhid[tid] = bh; a.bhid_prev[tid] = bh_prev; }
    if (tid < S) a.bvis[tid] = bv;
    if (tid == 0 && a.cost) *a.cost = cost;
}

struct DenseArgs { float *W, *bhid, *bvis; const float* X; int64_t N; int row, col; float lr; int skip_last; double* cost; };

template <int RPW, int CPL>
__global__ __launch_bounds__(1024) void k_dae_dense(const DenseArgs a)
{
    constexpr int RP = 16 * RPW, CP = 64 * CPL;
    __shared__ float s_x[2][RP];
    __shared__ float s_d[RP], s_bv[RP];                           // per-row values, owned by the row's wave
    __shared__ float s_y[CP], s_dy[CP];                           // per-column values, owned by thread j < CP
    __shared__ float s_part[16][CP];
    __shared__ double s_cost[16];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, row = a.row, col = a.col, r0 = w * RPW;
    float Wr[RPW][CPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int i = r0 + r;
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = l + 64 * c; Wr[r][c] = (i < row && j < col) ? a.W[(size_t)i * col + j] : 0.f; }
    }
    float bh = (tid < col) ? a.bhid[tid] : 0.f;                   // thread j < col owns hidden bias j
    double cost = 0.0;
    if (tid < RP) { s_x[0][tid] = (tid < row && a.N > 0) ? a.X[tid] : 0.f; s_bv[tid] = tid < row ? a.bvis[tid] : 0.f; }
    __syncthreads();
    for (int64_t n = 0; n < a.N; ++n) {
        const float* sx = s_x[n & 1];
        float xn = 0.f;                                           // next example's x: in flight under this step
        if (tid < row && n + 1 < a.N) xn = a.X[(size_t)(n + 1) * row + tid];
        {   // partial of x W over this wave's rows
            float p[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) p[c] = 0.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float xr = sx[r0 + r];
#pragma unroll
                for (int c = 0; c < CPL; ++c) p[c] = fmaf(xr, Wr[r][c], p[c]);
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) s_part[w][l + 64 * c] = p[c];
        }
        __syncthreads();
        float ymine = 0.f;
        if (tid < CP) {                                           // y_j = sigmoid(sum of the 16 partials + b_j)
            float s = bh;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) s += s_part[q][tid];
            ymine = tid < col ? sigm(s) : 0.f;
            s_y[tid] = ymine;
        }
        __syncthreads();
        float yv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) yv[c] = s_y[l + 64 * c];
        float cw = 0.f;
        float p[CPL];                                             // partial of d W over this wave's rows
#pragma unroll
        for (int c = 0; c < CPL; ++c) p[c] = 0.f;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {                           // z_i = sigmoid(W[i,:] . y + b'_i): a wave reduction
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc = fmaf(yv[c], Wr[r][c], acc);
            acc = wave_sum(acc);
            const int i = r0 + r;
            const float xr = sx[i], zi = sigm(acc + s_bv[i]);
            const float d = i < row ? zi - xr : 0.f;              // the same value in every lane of the wave
            cw += i < row ? xent(xr, zi) : 0.f;
            if (l == 0) s_d[i] = d;                               // read back only after the next barrier
#pragma unroll
            for (int c = 0; c < CPL; ++c) p[c] = fmaf(d, Wr[r][c], p[c]);
        }
        cost += (double)cw;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s_part[w][l + 64 * c] = p[c];
        if (tid < RP) s_x[(n & 1) ^ 1][tid] = xn;
        __syncthreads();
        const bool upd = !(a.skip_last && n + 1 == a.N);
        const float lr = upd ? a.lr : 0.f;
        if (tid < CP) {                                           // dy_j = (d W)_j y_j (1 - y_j); b_j <- b_j - lr dy_j
            float s = 0.f;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) s += s_part[q][tid];
            const float dy = s * ymine * (1.0f - ymine);
            s_dy[tid] = dy;
            bh -= lr * dy;
        }
        if (l < RPW) s_bv[r0 + l] -= lr * s_d[r0 + l];            // b'_i <- b'_i - lr (z_i - x_i), this wave's rows
        __syncthreads();
        float dyv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) dyv[c] = s_dy[l + 64 * c];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float xr = sx[r0 + r], d = s_d[r0 + r];
#pragma unroll
            for (int c = 0; c < CPL; ++c) Wr[r][c] -= lr * fmaf(xr, dyv[c], d * yv[c]);     // x (x) dy + d (x) y
        }
        // the next example writes s_part after reading nothing shared that is still in use; its first barrier
        // orders s_y / s_dy / s_d reuse
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int i = r0 + r;
        if (i < row) {
            if (l == 0) a.bvis[i] = s_bv[i];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const int j = l + 64 * c; if (j < col) a.W[(size_t)i * col + j] = Wr[r][c]; }
        }
    }
    if (tid < col) a.bhid[tid] = bh;
    if (l == 0) s_cost[w] = cost;
    __syncthreads();
    if (tid == 0 && a.cost) { double t = 0.0; for (int q = 0; q < 16; ++q) t += s_cost[q]; *a.cost = t; }
}

__global__ __launch_bounds__(1024) void k_dae_bag_cumsum(const float* __restrict__ W0, const float* __restrict__ b0, int H,
                                                          int64_t n_rows, const int32_t* __restrict__ ids, int n, int F,
                                                          float* __restrict__ out, int* __restrict__ err)
{
    __shared__ float s[2][1024];
    const int t = blockIdx.x, k = threadIdx.x;
    float v = 0.f;
    if (k < H) {
        for (int f = 0; f < F; ++f) {
            const int64_t id = ids[(size_t)t * F + f];
            if (id < -1 || id >= n_rows) { if (k == 0) atomicOr(err, 1); continue; }
            if (id >= 0) v += W0[(size_t)id * H + k];
        }
    }
    int cur = 0;
    s[0][k] = v;
    __syncthreads();
    for (int o = 1; o < (int)blockDim.x; o <<= 1) {               // inclusive scan over the hidden units (Q3)
        const float add = k >= o ? s[cur][k - o] : 0.f;
        s[cur ^ 1][k] = s[cur][k] + add;
        cur ^= 1;
        __syncthreads();
    }
    if (k < H) out[(size_t)t * H + k] = sigm(s[cur][k] + b0[k]);
}

int* g_flag(hipStream_t st) {
    static thread_local int* flag = nullptr;
    if (!flag) { if (hipMalloc((void**)&flag, 4) != hipSuccess) return nullptr; }
    hipMemsetAsync(flag, 0, 4, st);
    return flag;
}
int read_flag(int* flag, hipStream_t st) {
    int h = 0;
    if (hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    return h;
}

}  // namespace

extern "C" {

const char* dae_last_error(void) { return g_err.c_str(); }

int dae_sparse_epoch(const float* table, int64_t n_rows, float* bhid, float* bvis, float* bhid_prev, const int32_t* idx,
                     const float* x, int64_t N, int H, int S, float lr, double* cost_sum_out, void* stream)
{
    if (!table || !bhid || !bvis || !bhid_prev || !idx || !x) DFAIL(FNN_ERR_ARG, "null pointer");
    if (N < 1 || H < 1 || H > 256 || S < 1 || S > 32 || n_rows < 1) DFAIL(FNN_ERR_ARG, "need N >= 1, 1 <= H <= 256, 1 <= S <= 32");
    hipStream_t st = (hipStream_t)stream;
    int* flag = g_flag(st);
    if (!flag) DFAIL(FNN_ERR_HIP, "hipMalloc failed");
    double* dcost = nullptr;
    DHK(hipMalloc((void**)&dcost, 8));
    SparseArgs a{table, n_rows, bhid, bvis, bhid_prev, idx, x, N, H, S, lr, dcost, flag};
    hipLaunchKernelGGL(k_dae_sparse, dim3(1), dim3(256), 0, st, a);
    DHK(hipGetLastError());
    double c = 0.0;
    DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
    const int bad = read_flag(flag, st);
    hipFree(dcost);
    if (bad < 0) DFAIL(FNN_ERR_HIP, "stream synchronisation failed");
    if (bad) DFAIL(FNN_ERR_RANGE, "visible id outside [0, n_rows)");
    if (cost_sum_out) *cost_sum_out = c;
    return FNN_OK;
}

int dae_dense_epoch(float* W, float* bhid, float* bvis, const float* X, int64_t N, int row, int col, float lr,
                    int skip_last_update, double* cost_sum_out, void* stream)
{
    if (!W || !bhid || !bvis || !X) DFAIL(FNN_ERR_ARG, "null pointer");
    if (N < 1 || row < 1 || col < 1) DFAIL(FNN_ERR_ARG, "need N, row, col >= 1");
    hipStream_t st = (hipStream_t)stream;
    double* dcost = nullptr;
    DHK(hipMalloc((void**)&dcost, 8));
    DenseArgs a{W, bhid, bvis, X, N, row, col, lr, skip_last_update, dcost};
    // register tilings (rows per wave x columns per lane); the smallest that holds [row][col]
    if (row <= 16 * 4 && col <= 64) hipLaunchKernelGGL((k_dae_dense<4, 1>), dim3(1), dim3(1024), 0, st, a);
    else if (row <= 16 * 8 && col <= 128) hipLaunchKernelGGL((k_dae_dense<8, 2>), dim3(1), dim3(1024), 0, st, a);
    else if (row <= 16 * 19 && col <= 128) hipLaunchKernelGGL((k_dae_dense<19, 2>), dim3(1), dim3(1024), 0, st, a);
    else if (row <= 16 * 13 && col <= 320) hipLaunchKernelGGL((k_dae_dense<13, 5>), dim3(1), dim3(1024), 0, st, a);
    else { hipFree(dcost); DFAIL(FNN_ERR_ARG, "dae_dense_epoch: [row][col] must fit [304][128] or [208][320]"); }
    DHK(hipGetLastError());
    double c = 0.0;
    DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
    DHK(hipStreamSynchronize(st));
    hipFree(dcost);
    if (cost_sum_out) *cost_sum_out = c;
    return FNN_OK;
}

int dae_bag_cumsum_sigmoid(const float* W0, const float* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F, float* out,
                           void* stream)
{
    if (!W0 || !b0 || !ids || !out) DFAIL(FNN_ERR_ARG, "null pointer");
    if (H < 1 || H > 1024 || n < 1 || F < 1) DFAIL(FNN_ERR_ARG, "need 1 <= H <= 1024, n >= 1, F >= 1");
    hipStream_t st = (hipStream_t)stream;
    int* flag = g_flag(st);
    if (!flag) DFAIL(FNN_ERR_HIP, "hipMalloc failed");
    int bs = 64; while (bs < H) bs <<= 1;
    hipLaunchKernelGGL(k_dae_bag_cumsum, dim3(n), dim3(bs), 0, st, W0, b0, H, n_rows, ids, n, F, out, flag);
    DHK(hipGetLastError());
    const int bad = read_flag(flag, st);
    if (bad < 0) DFAIL(FNN_ERR_HIP, "stream synchronisation failed");
    if (bad) DFAIL(FNN_ERR_RANGE, "feature id outside [-1, n_rows)");
    return FNN_OK;
}

}  // extern "C"
