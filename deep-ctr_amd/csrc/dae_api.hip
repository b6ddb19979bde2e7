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

#include <cstdlib>
#include <string>

#include "../../include/dae_hip.h"
#include "../../include/fnn_hip.h"

namespace {

thread_local std::string g_err;
#define DHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e_); return FNN_ERR_HIP; } } while (0)
#define DFAIL(code, msg) do { g_err = (msg); return (code); } while (0)

struct DevDouble {                       // one device double, freed on every exit path
    double* p = nullptr;
    ~DevDouble() { if (p) hipFree(p); }
};

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ inline double sigm(double z) { return 1.0 / (1.0 + exp(-z)); }
__device__ inline float wave_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
// -x log z - (1-x) log(1-z), the terms with a zero factor dropped as Theano's 0 * log(.) = 0 would
// only differ at z in {0, 1}
__device__ inline float xent(float x, float z) { return -(x * logf(z) + (1.0f - x) * logf(1.0f - z)); }
__device__ inline double xent(double x, double z) { return -(x * log(z) + (1.0 - x) * log(1.0 - z)); }
__device__ inline double wave_sum(double v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

// S = float (the fast mode) or double (the reference's own precision: its float64 trajectory is
// sensitive enough at lr = 0.1 that only an f64 run tracks it over thousands of steps).
template <typename S> struct SparseArgs {
    const S* table; int64_t n_rows; S *bhid, *bvis, *bhid_prev; const int32_t* idx; const S* x;
    int64_t N; int H, S_; S lr; double* cost; int* err;
};

template <typename S>
__global__ __launch_bounds__(256) void k_dae_sparse(const SparseArgs<S> a)
{
    extern __shared__ __align__(16) unsigned char dae_smem[];
    S (*s_w)[257] = reinterpret_cast<S (*)[257]>(dae_smem);                 // [32][257]
    S* s_y = reinterpret_cast<S*>(dae_smem) + 32 * 257;                      // [256]
    S* s_d = s_y + 256; S* s_x = s_d + 32; S* s_c = s_x + 32;               // [32] each
    const int tid = threadIdx.x, H = a.H, Sn = a.S_;
    const bool act = tid < H;
    S bh = act ? a.bhid[tid] : (S)0, bh_prev = bh;
    S bv = (tid < Sn) ? a.bvis[tid] : (S)0;                      // thread j < S also owns positional visible bias j
    double cost = 0.0;
    S wn[32];                                                     // rows of the NEXT example
    auto fetch = [&](int64_t n) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            S v = (S)0;
            if (act && j < Sn && n < a.N) {
                const int64_t id = a.idx[n * Sn + j];
                if (id >= 0 && id < a.n_rows) v = a.table[(size_t)id * H + tid]; else if (tid == 0) atomicOr(a.err, 1);
            }
            wn[j] = v;
        }
    };
    fetch(0);
    for (int64_t n = 0; n < a.N; ++n) {
        S wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = wn[j];
        if (tid < 32) s_x[tid] = tid < Sn ? a.x[n * Sn + tid] : (S)0;
        fetch(n + 1);                                             // in flight under this example's arithmetic
        __syncthreads();
        S z = bh;                                                 // y = sigmoid(x W + b)            (:94)
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fma(s_x[j], wc[j], z);
        const S y = act ? sigm(z) : (S)0;
        s_y[tid] = y;
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {   // z_j = sigmoid(y . W[j,:] + bvis_j)                                                    (:97)
            const int j = tid >> 3, seg = tid & 7;
            S acc = (S)0;
            for (int i = seg; i < H; i += 8) acc = fma(s_y[i], s_w[j][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) s_c[j] = acc;
        }
        __syncthreads();
        if (tid < 32) {
            S d = (S)0;
            if (tid < Sn) {
                const S zj = sigm(s_c[tid] + bv), xj = s_x[tid];
                d = zj - xj;
                s_c[tid] = xent(xj, zj);
                bv -= a.lr * d;                                   // b' <- b' - lr (z - x)
            } else s_c[tid] = (S)0;
            s_d[tid] = d;
        }
        __syncthreads();
        S dy = (S)0;                                              // (d W) * y (1 - y)
#pragma unroll
        for (int j = 0; j < 32; ++j) dy = fma(s_d[j], wc[j], dy);
        dy *= y * ((S)1 - y);
        bh_prev = bh;
        bh -= a.lr * dy;
        if (tid == 0) { double c = 0.0; for (int j = 0; j < 32; ++j) c += (double)s_c[j]; cost += c; }
    }
    if (act) { a.bhid[tid] = bh; a.bhid_prev[tid] = bh_prev; }
    if (tid < Sn) a.bvis[tid] = bv;
    if (tid == 0 && a.cost) *a.cost = cost;
}
template <typename S> constexpr size_t dae_sparse_lds() { return (size_t)(32 * 257 + 256 + 96) * sizeof(S); }

// The dense trainer with W in global memory (L2-resident, one workgroup): any [row][col] up to
// 1024 columns, float or double.  Four passes over W per example -- y = x W, z = W y, dy = d W, the
// update -- with the next example's x W accumulated inside the update pass, so three reads and one
// write of W per step.  The register-tiled k_dae_dense below is the fast f32 form of the same step.
template <typename S> struct DenseGArgs { S *W, *bhid, *bvis; const S* X; int64_t N; int row, col; S lr; int skip_last; double* cost; };

template <typename S>
__global__ __launch_bounds__(1024) void k_dae_dense_g(const DenseGArgs<S> a)
{
    extern __shared__ __align__(16) unsigned char dae_smem[];
    const int tid = threadIdx.x, row = a.row, col = a.col, w = tid >> 6, l = tid & 63;
    const int CP = (col + 63) / 64 * 64, G = 1024 / CP, g = tid / CP, j = tid % CP;
    const bool cact = g < G && j < col;
    S* s_part = reinterpret_cast<S*>(dae_smem);                  // [G][CP] <= 1024
    S* s_y = s_part + 1024; S* s_dy = s_y + CP;                  // [CP] each
    S* s_x = s_dy + CP; S* s_xn = s_x + row; S* s_d = s_xn + row; S* s_bv = s_d + row;    // [row] each
    double* s_cost = reinterpret_cast<double*>(s_bv + row + (row & 1));     // [16]
    S bh = (tid < col) ? a.bhid[tid] : (S)0;
    for (int i = tid; i < row; i += 1024) { s_x[i] = a.X[i]; s_bv[i] = a.bvis[i]; }
    double cost = 0.0;
    __syncthreads();
    {   // x W of the first example
        S p = (S)0;
        if (cact) for (int i = g; i < row; i += G) p = fma(s_x[i], a.W[(size_t)i * col + j], p);
        if (g < G) s_part[g * CP + j] = p;
    }
    __syncthreads();
    for (int64_t n = 0; n < a.N; ++n) {
        for (int i = tid; i < row; i += 1024) s_xn[i] = (n + 1 < a.N) ? a.X[(size_t)(n + 1) * row + i] : (S)0;
        S y = (S)0;
        if (tid < col) {
            S s = bh;
            for (int q = 0; q < G; ++q) s += s_part[q * CP + tid];
            y = sigm(s);
        }
        if (tid < CP) s_y[tid] = y;
        __syncthreads();
        for (int i = w; i < row; i += 16) {                       // z_i = sigmoid(W[i,:] . y + b'_i), one wave per row
            S acc = (S)0;
            for (int c = l; c < col; c += 64) acc = fma(s_y[c], a.W[(size_t)i * col + c], acc);
            acc = wave_sum(acc);
            const S zi = sigm(acc + s_bv[i]), xi = s_x[i];
            if (l == 0) s_d[i] = zi - xi;
            cost += (double)xent(xi, zi);
        }
        __syncthreads();
        {   // d W
            S p = (S)0;
            if (cact) for (int i = g; i < row; i += G) p = fma(s_d[i], a.W[(size_t)i * col + j], p);
            if (g < G) s_part[g * CP + j] = p;
        }
        __syncthreads();
        const S lr = (a.skip_last && n + 1 == a.N) ? (S)0 : a.lr;
        if (tid < col) {
            S s = (S)0;
            for (int q = 0; q < G; ++q) s += s_part[q * CP + tid];
            const S dy = s * y * ((S)1 - y);
            s_dy[tid] = dy;
            bh -= lr * dy;
        }
        __syncthreads();
        {   // W <- W - lr (x (x) dy + d (x) y); the next example's x W rides on the same pass
            S p = (S)0;
            if (cact) {
                const S dyj = s_dy[j], yj = s_y[j];
                for (int i = g; i < row; i += G) {
                    const size_t o = (size_t)i * col + j;
                    const S wv = a.W[o] - lr * fma(s_x[i], dyj, s_d[i] * yj);
                    a.W[o] = wv;
                    p = fma(s_xn[i], wv, p);
                }
            }
            if (g < G) s_part[g * CP + j] = p;
        }
        __syncthreads();
        for (int i = tid; i < row; i += 1024) { s_bv[i] -= lr * s_d[i]; s_x[i] = s_xn[i]; }
        __syncthreads();
    }
    for (int i = tid; i < row; i += 1024) a.bvis[i] = s_bv[i];
    if (tid < col) a.bhid[tid] = bh;
    if (l == 0) s_cost[w] = cost;
    __syncthreads();
    if (tid == 0 && a.cost) { double t = 0.0; for (int q = 0; q < 16; ++q) t += s_cost[q]; *a.cost = t; }
}

// ------------------------------------------------------------------------------------------
// The dense trainer SPLIT over 8 workgroups of one XCD (round 3; the float64 form the SNN-DAE script runs by default).
// W [row][col] in float64 is 480 KB at 200 x 300: it fits no single CU's registers or LDS, and k_dae_dense_g above moves it
// through one CU's port three and a half times per example (52 us per example).  Here workgroup c owns the hidden units
// [c cw, (c + 1) cw), cw = ceil(col / 8) <= 64, and keeps ITS columns of W in registers for the whole pass (wave w: rows w, w + 16,
// ...; lane: column) -- no global traffic for W at all.  Per example everything is local to the owner of a column (y = x W,
// dy = d W, the update) except the reconstruction z_i = sigmoid(sum_j y_j W[i][j] + b'_i), a sum over ALL hidden units: every
// workgroup publishes its partial row sums (row doubles, write-through stores), raises its flag, waits for the other seven and
// adds the eight partials in workgroup order (the same order everywhere: all eight hold the same z, d, b').  One hand-off per
// example (MI355X_MICROARCH.md "Valid forms": every handed-off byte stored and loaded sc1, stores drained before the flag, one
// lane polls, the others load behind a barrier); the eight workgroups sit on one XCD (block ids 8 apart) so the exchange stays in
// its L2.  Flags hold the example number + 1 and only grow; the exchange buffer has two parities (a workgroup rewrites parity p two
// examples later, after it has seen every peer's next flag, which a peer raises only once it has read parity p).  A peer that
// never arrives: bounded poll, error bit, every workgroup leaves.
// ------------------------------------------------------------------------------------------
constexpr int DSP_NS = 8;
template <typename S> struct DenseSArgs {
    S *W, *bhid, *bvis; const S* X; int64_t N; int row, col, cw; S lr; int skip_last; double* cost;
    S* xch;                        // [2][DSP_NS][rowp] partial row sums
    unsigned long long* flags;     // [DSP_NS] on 64-byte lines (stride 8)
    int rowp; int* err;
};

template <typename S, int RPW>
__global__ __launch_bounds__(1024) void k_dae_dense_split(const DenseSArgs<S> a)
{
    constexpr int RP = 16 * RPW;
    __shared__ S s_x[RP], s_xn[RP], s_d[RP], s_bv[RP], s_zp[RP];
    __shared__ S s_y[64], s_dy[64];
    __shared__ S s_part[16][64];
    __shared__ double s_cost[16];
    __shared__ int s_bad;
    if (blockIdx.x & 7) return;                                   // the eight working blocks are 8 apart: one XCD under round-robin placement
    const int wg = blockIdx.x >> 3, tid = threadIdx.x, w = tid >> 6, l = tid & 63, row = a.row, col = a.col;
    const int j = wg * a.cw + l;
    const bool cact = l < a.cw && j < col;
    S Wr[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) { const int i = w + 16 * r; Wr[r] = (cact && i < row) ? a.W[(size_t)i * col + j] : (S)0; }
    S bh = (w == 0 && cact) ? a.bhid[j] : (S)0;
    for (int i = tid; i < RP; i += 1024) { s_x[i] = i < row ? a.X[i] : (S)0; s_bv[i] = i < row ? a.bvis[i] : (S)0; s_xn[i] = (S)0; s_d[i] = (S)0; }
    if (tid == 0) s_bad = 0;
    double cost = 0.0;
    __syncthreads();
    {   // x W of the first example: a wave's share of the rows, then 16 partials per column
        S p = (S)0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) p = fma(s_x[w + 16 * r], Wr[r], p);
        s_part[w][l] = p;
    }
    __syncthreads();
    for (int64_t n = 0; n < a.N; ++n) {
        for (int i = tid; i < row; i += 1024) s_xn[i] = (n + 1 < a.N) ? a.X[(size_t)(n + 1) * row + i] : (S)0;
        S y = (S)0;
        if (w == 0) {
            S t = bh;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += s_part[q][l];
            y = cact ? sigm(t) : (S)0;
            s_y[l] = y;
        }
        __syncthreads();
        {   // this workgroup's share of z: row sums over ITS columns (wave reductions)
            const S yl = s_y[l];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const S v = wave_sum(yl * Wr[r]);
                if (l == 0) s_zp[w + 16 * r] = v;
            }
        }
        __syncthreads();
        {   // publish, signal, wait, gather
            S* mine = a.xch + ((size_t)(n & 1) * DSP_NS + wg) * a.rowp;
            for (int i = tid; i < row; i += 1024) __hip_atomic_store(mine + i, s_zp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + wg * 8, (unsigned long long)(n + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < DSP_NS) {
                int tries = 0;
                while (__hip_atomic_load(a.flags + tid * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)(n + 1)) {
                    if (++tries > (1 << 22)) { s_bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (s_bad) break;
            const S* all = a.xch + (size_t)(n & 1) * DSP_NS * a.rowp;
            for (int i = tid; i < row; i += 1024) {
                S zs = s_bv[i];
#pragma unroll
                for (int c = 0; c < DSP_NS; ++c) zs += __hip_atomic_load(all + (size_t)c * a.rowp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 loads
                const S zi = sigm(zs), xi = s_x[i];
                s_d[i] = zi - xi;
                if (wg == 0) cost += (double)xent(xi, zi);
            }
        }
        __syncthreads();
        {   // d W over the wave's rows
            S p = (S)0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) p = fma(s_d[w + 16 * r], Wr[r], p);
            s_part[w][l] = p;
        }
        __syncthreads();
        const S lr = (a.skip_last && n + 1 == a.N) ? (S)0 : a.lr;
        if (w == 0) {
            S t = (S)0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += s_part[q][l];
            const S dy = t * y * ((S)1 - y);
            s_dy[l] = dy;
            bh -= lr * dy;
        }
        __syncthreads();
        {   // W <- W - lr (x (x) dy + d (x) y); the next example's x W rides on the same pass
            const S dyl = s_dy[l], yl = s_y[l];
            S p = (S)0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = w + 16 * r;
                const S wv = Wr[r] - lr * fma(s_x[i], dyl, s_d[i] * yl);
                Wr[r] = cact ? wv : (S)0;
                p = fma(s_xn[i], Wr[r], p);
            }
            __syncthreads();                                       // every wave has read s_part / s_x of this example
            s_part[w][l] = p;
        }
        for (int i = tid; i < row; i += 1024) { s_bv[i] -= lr * s_d[i]; s_x[i] = s_xn[i]; }
        __syncthreads();
    }
    if (s_bad) { if (tid == 0) atomicOr(a.err, 4); return; }
#pragma unroll
    for (int r = 0; r < RPW; ++r) { const int i = w + 16 * r; if (cact && i < row) a.W[(size_t)i * col + j] = Wr[r]; }
    if (w == 0 && cact) a.bhid[j] = bh;
    if (wg == 0) {
        for (int i = tid; i < row; i += 1024) a.bvis[i] = s_bv[i];
        const double cwv = wave_sum(cost);                       // every thread i < row carried its row's terms
        if (l == 0) s_cost[w] = cwv;
        __syncthreads();
        if (tid == 0 && a.cost) { double t = 0.0; for (int q = 0; q < 16; ++q) t += s_cost[q]; *a.cost = t; }
    }
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

template <typename S>
__global__ __launch_bounds__(1024) void k_dae_bag_cumsum(const S* __restrict__ W0, const S* __restrict__ b0, int H,
                                                          int64_t n_rows, const int32_t* __restrict__ ids, int n, int F,
                                                          S* __restrict__ out, int* __restrict__ err)
{
    __shared__ S s[2][1024];
    const int t = blockIdx.x, k = threadIdx.x;
    S v = (S)0;
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
        const S add = k >= o ? s[cur][k - o] : (S)0;
        s[cur ^ 1][k] = s[cur][k] + add;
        cur ^= 1;
        __syncthreads();
    }
    if (k < H) out[(size_t)t * H + k] = sigm(s[cur][k] + b0[k]);
}

// out [n][b] = sigmoid(in [n][a] . W [a][b] + bias): the propagation between dense layers (:183-187), f64
__global__ void k_dae_affine_sigmoid_f64(const double* __restrict__ in, const double* __restrict__ W, const double* __restrict__ bias,
                                         int n, int a, int b, double* __restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * b) return;
    const int t = (int)(gid / b), c = (int)(gid % b);
    double s = bias[c];
    for (int i = 0; i < a; ++i) s = fma(in[(size_t)t * a + i], W[(size_t)i * b + c], s);
    out[gid] = sigm(s);
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

}  // extern "C" (templated bodies below)

namespace {

template <typename S>
int sparse_epoch_t(const S* table, int64_t n_rows, S* bhid, S* bvis, S* bhid_prev, const int32_t* idx, const S* x, int64_t N, int H,
                   int Sn, S lr, double* cost_sum_out, void* stream)
{
    if (!table || !bhid || !bvis || !bhid_prev || !idx || !x) DFAIL(FNN_ERR_ARG, "null pointer");
    if (N < 1 || H < 1 || H > 256 || Sn < 1 || Sn > 32 || n_rows < 1) DFAIL(FNN_ERR_ARG, "need N >= 1, 1 <= H <= 256, 1 <= S <= 32");
    hipStream_t st = (hipStream_t)stream;
    int* flag = g_flag(st);
    if (!flag) DFAIL(FNN_ERR_HIP, "hipMalloc failed");
    DevDouble dc;
    DHK(hipMalloc((void**)&dc.p, 8));
    double* dcost = dc.p;
    SparseArgs<S> a{table, n_rows, bhid, bvis, bhid_prev, idx, x, N, H, Sn, lr, dcost, flag};
    DHK(hipFuncSetAttribute((const void*)k_dae_sparse<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dae_sparse_lds<S>()));
    hipLaunchKernelGGL((k_dae_sparse<S>), dim3(1), dim3(256), dae_sparse_lds<S>(), st, a);
    DHK(hipGetLastError());
    double c = 0.0;
    DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
    const int bad = read_flag(flag, st);
    if (bad < 0) DFAIL(FNN_ERR_HIP, "stream synchronisation failed");
    if (bad) DFAIL(FNN_ERR_RANGE, "visible id outside [0, n_rows)");
    if (cost_sum_out) *cost_sum_out = c;
    return FNN_OK;
}

template <typename S>
int dense_g_epoch_t(S* W, S* bhid, S* bvis, const S* X, int64_t N, int row, int col, S lr, int skip_last, double* cost_sum_out,
                    void* stream)
{
    if (!W || !bhid || !bvis || !X) DFAIL(FNN_ERR_ARG, "null pointer");
    if (N < 1 || row < 1 || row > 2048 || col < 1 || col > 1024) DFAIL(FNN_ERR_ARG, "need N >= 1, 1 <= row <= 2048, 1 <= col <= 1024");
    hipStream_t st = (hipStream_t)stream;
    DevDouble dc;
    DHK(hipMalloc((void**)&dc.p, 8));
    double* dcost = dc.p;
    {   // float64, shapes the split form holds (col <= 512, row <= 512): eight workgroups with W in registers ($DAE_SPLIT=0: the one-workgroup form below)
        const char* ev = getenv("DAE_SPLIT");
        if (sizeof(S) == 8 && col <= 64 * DSP_NS && row <= 16 * 32 && !(ev && ev[0] == '0')) {
            const int rowp = (row + 15) / 16 * 16, cw = (col + DSP_NS - 1) / DSP_NS;
            struct Scratch { char* p = nullptr; ~Scratch() { if (p) hipFree(p); } } sc;
            const size_t xb = (size_t)2 * DSP_NS * rowp * sizeof(S), fb = (size_t)DSP_NS * 64, total = xb + fb + 64;
            DHK(hipMalloc((void**)&sc.p, total));
            DHK(hipMemsetAsync(sc.p, 0, total, st));
            DenseSArgs<S> sa{W, bhid, bvis, X, N, row, col, cw, lr, skip_last, dcost, reinterpret_cast<S*>(sc.p),
                             reinterpret_cast<unsigned long long*>(sc.p + xb), rowp, reinterpret_cast<int*>(sc.p + xb + fb)};
            const int rpw = (row + 15) / 16;
            if (rpw <= 8) hipLaunchKernelGGL((k_dae_dense_split<S, 8>), dim3(8 * DSP_NS), dim3(1024), 0, st, sa);
            else if (rpw <= 13) hipLaunchKernelGGL((k_dae_dense_split<S, 13>), dim3(8 * DSP_NS), dim3(1024), 0, st, sa);
            else if (rpw <= 19) hipLaunchKernelGGL((k_dae_dense_split<S, 19>), dim3(8 * DSP_NS), dim3(1024), 0, st, sa);
            else hipLaunchKernelGGL((k_dae_dense_split<S, 32>), dim3(8 * DSP_NS), dim3(1024), 0, st, sa);
            DHK(hipGetLastError());
            double c = 0.0; int bad = 0;
            DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
            DHK(hipMemcpyAsync(&bad, sa.err, 4, hipMemcpyDeviceToHost, st));
            DHK(hipStreamSynchronize(st));
            if (bad) DFAIL(FNN_ERR_HIP, "dae_dense_epoch_f64: a workgroup of the split trainer gave up waiting for its peers (DAE_SPLIT=0 selects the one-workgroup form); the parameters of this pass are invalid");
            if (cost_sum_out) *cost_sum_out = c;
            return FNN_OK;
        }
    }
    const int CP = (col + 63) / 64 * 64;
    const size_t lds = (size_t)(1024 + 2 * CP + 4 * row + 2) * sizeof(S) + 16 * 8;
    DenseGArgs<S> a{W, bhid, bvis, X, N, row, col, lr, skip_last, dcost};
    DHK(hipFuncSetAttribute((const void*)k_dae_dense_g<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_dae_dense_g<S>), dim3(1), dim3(1024), lds, st, a);
    DHK(hipGetLastError());
    double c = 0.0;
    DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
    DHK(hipStreamSynchronize(st));
    if (cost_sum_out) *cost_sum_out = c;
    return FNN_OK;
}

template <typename S>
int bag_cumsum_t(const S* W0, const S* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F, S* out, void* stream)
{
    if (!W0 || !b0 || !ids || !out) DFAIL(FNN_ERR_ARG, "null pointer");
    if (H < 1 || H > 1024 || n < 1 || F < 1) DFAIL(FNN_ERR_ARG, "need 1 <= H <= 1024, n >= 1, F >= 1");
    hipStream_t st = (hipStream_t)stream;
    int* flag = g_flag(st);
    if (!flag) DFAIL(FNN_ERR_HIP, "hipMalloc failed");
    int bs = 64; while (bs < H) bs <<= 1;
    hipLaunchKernelGGL((k_dae_bag_cumsum<S>), dim3(n), dim3(bs), 0, st, W0, b0, H, n_rows, ids, n, F, out, flag);
    DHK(hipGetLastError());
    const int bad = read_flag(flag, st);
    if (bad < 0) DFAIL(FNN_ERR_HIP, "stream synchronisation failed");
    if (bad) DFAIL(FNN_ERR_RANGE, "feature id outside [-1, n_rows)");
    return FNN_OK;
}

}  // namespace

extern "C" {

int dae_sparse_epoch(const float* table, int64_t n_rows, float* bhid, float* bvis, float* bhid_prev, const int32_t* idx,
                     const float* x, int64_t N, int H, int S, float lr, double* cost_sum_out, void* stream)
{
    return sparse_epoch_t<float>(table, n_rows, bhid, bvis, bhid_prev, idx, x, N, H, S, lr, cost_sum_out, stream);
}
int dae_sparse_epoch_f64(const double* table, int64_t n_rows, double* bhid, double* bvis, double* bhid_prev, const int32_t* idx,
                         const double* x, int64_t N, int H, int S, double lr, double* cost_sum_out, void* stream)
{
    return sparse_epoch_t<double>(table, n_rows, bhid, bvis, bhid_prev, idx, x, N, H, S, lr, cost_sum_out, stream);
}

int dae_dense_epoch(float* W, float* bhid, float* bvis, const float* X, int64_t N, int row, int col, float lr,
                    int skip_last_update, double* cost_sum_out, void* stream)
{
    if (!W || !bhid || !bvis || !X) DFAIL(FNN_ERR_ARG, "null pointer");
    if (N < 1 || row < 1 || col < 1) DFAIL(FNN_ERR_ARG, "need N, row, col >= 1");
    hipStream_t st = (hipStream_t)stream;
    // register tilings (rows per wave x columns per lane); the smallest that holds [row][col]; anything
    // larger takes the global-memory form of the same step
    const bool fits = (row <= 64 && col <= 64) || (row <= 128 && col <= 128) || (row <= 304 && col <= 128) || (row <= 208 && col <= 320);
    if (!fits) return dense_g_epoch_t<float>(W, bhid, bvis, X, N, row, col, lr, skip_last_update, cost_sum_out, stream);
    DevDouble dc;
    DHK(hipMalloc((void**)&dc.p, 8));
    double* dcost = dc.p;
    DenseArgs a{W, bhid, bvis, X, N, row, col, lr, skip_last_update, dcost};
    if (row <= 16 * 4 && col <= 64) hipLaunchKernelGGL((k_dae_dense<4, 1>), dim3(1), dim3(1024), 0, st, a);
    else if (row <= 16 * 8 && col <= 128) hipLaunchKernelGGL((k_dae_dense<8, 2>), dim3(1), dim3(1024), 0, st, a);
    else if (row <= 16 * 19 && col <= 128) hipLaunchKernelGGL((k_dae_dense<19, 2>), dim3(1), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((k_dae_dense<13, 5>), dim3(1), dim3(1024), 0, st, a);
    DHK(hipGetLastError());
    double c = 0.0;
    DHK(hipMemcpyAsync(&c, dcost, 8, hipMemcpyDeviceToHost, st));
    DHK(hipStreamSynchronize(st));
    if (cost_sum_out) *cost_sum_out = c;
    return FNN_OK;
}
int dae_dense_epoch_f64(double* W, double* bhid, double* bvis, const double* X, int64_t N, int row, int col, double lr,
                        int skip_last_update, double* cost_sum_out, void* stream)
{
    return dense_g_epoch_t<double>(W, bhid, bvis, X, N, row, col, lr, skip_last_update, cost_sum_out, stream);
}

int dae_bag_cumsum_sigmoid(const float* W0, const float* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F, float* out,
                           void* stream)
{
    return bag_cumsum_t<float>(W0, b0, H, n_rows, ids, n, F, out, stream);
}
int dae_bag_cumsum_sigmoid_f64(const double* W0, const double* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F, double* out,
                               void* stream)
{
    return bag_cumsum_t<double>(W0, b0, H, n_rows, ids, n, F, out, stream);
}

int dae_affine_sigmoid_f64(const double* in, const double* W, const double* bias, int n, int a, int b, double* out, void* stream)
{
    if (!in || !W || !bias || !out || n < 1 || a < 1 || b < 1) DFAIL(FNN_ERR_ARG, "null pointer or empty shape");
    const size_t cnt = (size_t)n * b;
    hipLaunchKernelGGL(k_dae_affine_sigmoid_f64, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, W, bias, n, a, b, out);
    DHK(hipGetLastError());
    return FNN_OK;
}

}  // extern "C"
