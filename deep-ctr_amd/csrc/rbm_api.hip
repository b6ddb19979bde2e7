// rbm_api.hip -- SNN pre-training kernels and their C ABI (include/rbm_hip.h), gfx950 only.
// Replaces the NumPy CD-1 trainers of python/sampling_based_gaussian_binary_rbm_sparse.py
// (Atomu2014/deep-ctr): sparse online CD-1 (:294-508) and dense mini-batch CD-1 (:10-291).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fnn_hip.h"
#include "../../include/rbm_hip.h"
#include "fnn_kernels.hip.h"
#include "metrics.hip.h"

using namespace fnn;

namespace {

thread_local std::string g_err;
#define RCK(expr)                                                                      \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e_); return FNN_ERR_HIP; } \
    } while (0)
#define RFAIL(code, msg) do { g_err = (msg); return (code); } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }

// ------------------------------------------------------------------------------------------
// A7  sparse online CD-1 (python :423-505).  The trainer is sequential by definition (every
// example reads the rows the previous one wrote), so ONE workgroup walks the examples; thread i
// owns hidden unit i: column i of the S gathered rows (registers), of the positional momentum
// buffer (registers) and hidbias[i].  The visible reconstruction is the only cross-thread step:
// 8 lanes per visible unit reduce hs . W[f,:] through LDS.
// ------------------------------------------------------------------------------------------
struct SparseArgs {
    float *W, *visbias, *hidbias, *wstep; const int32_t* vid; const uint8_t* vval; const float* unif;
    int64_t N; int H, S; float wcost, r_vis, r_hid, r_w, mom; double* sq_err;
};

__global__ __launch_bounds__(256) void k_rbm_sparse(const SparseArgs a)
{
    __shared__ float s_w[32][257];
    __shared__ float s_hs[256], s_vis[32], s_v[32], s_e[32];
    __shared__ int s_id[32];
    const int tid = threadIdx.x, H = a.H, S = a.S;
    const bool act = tid < H;
    float ws[32], hb = act ? a.hidbias[tid] : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) ws[j] = (act && j < S) ? a.wstep[(size_t)j * H + tid] : 0.f;
    double err = 0.0;
    for (int64_t n = 0; n < a.N; ++n) {
        if (tid < 32) {
            s_id[tid] = tid < S ? a.vid[n * S + tid] : 0;
            s_v[tid] = tid < S ? (float)a.vval[n * S + tid] : 0.f;
        }
        __syncthreads();
        float wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = (act && j < S) ? a.W[(size_t)s_id[j] * H + tid] : 0.f;
        float z = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fmaf(s_v[j], wc[j], z);
        const float hid = sigm(z);                                         // hid_activate(mf=True)
        const float u = act ? a.unif[n * H + tid] : 2.f;
        s_hs[tid] = act ? ((u < hid) ? 1.0f : floorf(hid)) : 0.f;          // sample_hid (:371-375)
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {   // vis_j = sigmoid(hs . W[f_j,:] + visbias[f_j])  (mean field, :356-366)
            const int j = tid >> 3, seg = tid & 7;
            float acc = 0.f;
            for (int i = seg; i < H; i += 8) acc = fmaf(s_hs[i], s_w[j][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) {
                const float vj = j < S ? sigm(acc + a.visbias[s_id[j]]) : 0.f;
                s_vis[j] = vj;
                const float d = vj - s_v[j];
                s_e[j] = j < S ? d * d : 0.f;
                if (j < S) a.visbias[s_id[j]] += (s_v[j] - vj) * a.r_vis;   // :472-484
            }
        }
        __syncthreads();
        float z2 = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z2 = fmaf(s_vis[j], wc[j], z2);
        const float hid2 = sigm(z2);                                        // mean-field hiddens
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float step = ((s_v[j] * hid - s_vis[j] * hid2) - a.wcost * wc[j]) * a.r_w;   // :447-453
            ws[j] = ws[j] * a.mom + step;                                   // :454-455
            if (act && j < S) a.W[(size_t)s_id[j] * H + tid] = wc[j] + 2.0f * ws[j];    // applied twice (:461-462)
        }
        hb += (hid - hid2) * a.r_hid;                                       // :492-495
        if (tid == 0) { float e = 0.f; for (int j = 0; j < 32; ++j) e += s_e[j]; err += (double)e; }
        __syncthreads();          // W / visbias stores are drained (vmcnt 0) before the next example reads
    }
    if (act) a.hidbias[tid] = hb;
#pragma unroll
    for (int j = 0; j < 32; ++j) if (act && j < S) a.wstep[(size_t)j * H + tid] = ws[j];
    if (tid == 0 && a.sq_err) *a.sq_err = err;
}

// The same pass for S = 32 (the only shape the reference's buffers admit, :388), written for instruction count -- the
// pass is bound by the instructions one wave per SIMD issues per example, not by memory.  Lane j of every wave loads
// id j and value j of the example; v_readlane turns them into scalars, so that the 32 row addresses are scalar
// arithmetic (a scalar base + the thread's column), the gathers and the row stores carry no per-element branch, and
// the value of a visible is a scalar operand of its fma.
__global__ __launch_bounds__(256) void k_rbm_sparse32(const SparseArgs a)
{
    __shared__ float s_w[32][257];
    __shared__ float s_hs[256];
    __shared__ __align__(16) float s_vis[32];
    __shared__ float s_e[32];
    const int tid = threadIdx.x, lane = tid & 63, H = a.H;
    const bool act = tid < H;
    const int col = act ? tid : 0;                          // idle threads (H < 256) shadow column 0 and never store
    float ws[32], hb = act ? a.hidbias[tid] : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) ws[j] = act ? a.wstep[(size_t)j * H + tid] : 0.f;
    const int jj = tid >> 3, seg = tid & 7;                 // 8 lanes reduce visible jj
    double err = 0.0;
    for (int64_t n = 0; n < a.N; ++n) {
        const int idv = a.vid[n * 32 + (lane & 31)];
        const float vv = (float)a.vval[n * 32 + (lane & 31)];
        const float u = act ? a.unif[n * H + tid] : 2.f;
        const int my_id = __shfl(idv, jj & 31);             // wave w reduces visibles 8 w .. 8 w + 7
        const float my_v = __shfl(vv, jj & 31);
        const float my_vb = a.visbias[my_id];
        int id[32]; float v[32], wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            id[j] = __builtin_amdgcn_readlane(idv, j);
            v[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), j));
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = a.W[(size_t)id[j] * H + col];
        float z = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fmaf(v[j], wc[j], z);
        const float hid = sigm(z);                                         // hid_activate(mf=True)
        s_hs[tid] = act ? ((u < hid) ? 1.0f : floorf(hid)) : 0.f;          // sample_hid (:371-375)
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {   // vis_j = sigmoid(hs . W[f_j,:] + visbias[f_j])  (mean field, :356-366)
            float acc = 0.f;
            for (int i = seg; i < H; i += 8) acc = fmaf(s_hs[i], s_w[jj][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) {
                const float vj = sigm(acc + my_vb);
                s_vis[jj] = vj;
                const float d = vj - my_v;
                s_e[jj] = d * d;
                a.visbias[my_id] = my_vb + (my_v - vj) * a.r_vis;          // :472-484
            }
        }
        __syncthreads();
        float vis[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(s_vis + 4 * q);
            vis[4 * q] = t4.x; vis[4 * q + 1] = t4.y; vis[4 * q + 2] = t4.z; vis[4 * q + 3] = t4.w;
        }
        float z2 = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z2 = fmaf(vis[j], wc[j], z2);
        const float hid2 = sigm(z2);                                        // mean-field hiddens
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float step = ((v[j] * hid - vis[j] * hid2) - a.wcost * wc[j]) * a.r_w;   // :447-453
            ws[j] = ws[j] * a.mom + step;                                   // :454-455
            wc[j] = wc[j] + 2.0f * ws[j];                                   // applied twice (:461-462)
        }
        if (act) {
#pragma unroll
            for (int j = 0; j < 32; ++j) a.W[(size_t)id[j] * H + col] = wc[j];
        }
        hb += (hid - hid2) * a.r_hid;                                       // :492-495
        if (tid == 0) { float e = 0.f; for (int j = 0; j < 32; ++j) e += s_e[j]; err += (double)e; }
        __syncthreads();          // W / visbias stores are drained (vmcnt 0) before the next example reads
    }
    if (act) {
        a.hidbias[tid] = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) a.wstep[(size_t)j * H + tid] = ws[j];
    }
    if (tid == 0 && a.sq_err) *a.sq_err = err;
}

// ------------------------------------------------------------------------------------------
// Mini-batch variant of A7 (SURVEY 8d config 5 "batched mode for throughput"; NOT the reference's schedule, which is
// online -- for a mini-batch of 1 the two coincide).  All M examples of a mini-batch read the parameters as they
// were at its start:
//     step_e[j]    = ((v_ej hid_e - vis_ej hid2_e) - weightcost W[f_ej]) rate_w          (as :447-453)
//     W[f_ej]     += 2 (momentum wstep[j] + step_e[j])                                    (:454-462, per example)
//     wstep[j]     = momentum wstep[j] + mean_e step_e[j]                                 (the positional buffer)
//     visbias[f_ej] += (v_ej - vis_ej) rate_vis ;  hidbias += rate_hid sum_e (hid_e - hid2_e)
// k_rbm_batch: one workgroup per example slot (thread = hidden unit), examples e = blockIdx, + gridDim, ...;
// row deltas are ACCUMULATED into dW / dvis (zero outside this function) with float atomics, so that no example
// reads a row another one has already moved; k_rbm_apply then moves every touched row once (grab-and-zero);
// k_rbm_batch_tail / k_rbm_tail_apply reduce the per-workgroup partials of wstep / hidbias / error in a fixed order.  The row sums
// depend on the order the atomics land in (rounding only): this mode is not bit-reproducible, the online one is.
// ------------------------------------------------------------------------------------------
struct BatchArgs {
    const float *W, *visbias, *hidbias, *wstep; float *dW, *dvis; const int32_t* vid; const uint8_t* vval; const float* unif;
    int M, H, S; float wcost, r_vis, r_w, mom; float* part_w; float* part_h; double* part_e;
    float* hbuf; float* visbuf;       // SORTED form: hid / hid2 of every example [M][2][H] and its reconstructions [M][S] (the row update recomputes the deltas)
    int* owner_cnt;                   // SORTED form: the row update's counter of multi-chunk runs, zeroed here
};

template <bool SORTED>
__global__ __launch_bounds__(256) void k_rbm_batch(const BatchArgs a)
{
    __shared__ float s_w[32][257];
    __shared__ float s_hs[256], s_vis[32], s_v[32], s_e[32];
    __shared__ int s_id[32];
    const int tid = threadIdx.x, H = a.H, S = a.S;
    const bool act = tid < H;
    if (SORTED && blockIdx.x == 0 && tid == 0) *a.owner_cnt = 0;
    const float hb = act ? a.hidbias[tid] : 0.f;
    float ws0[32], wsum[32], hacc = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) { ws0[j] = (act && j < S) ? a.mom * a.wstep[(size_t)j * H + tid] : 0.f; wsum[j] = 0.f; }
    double err = 0.0;
    for (int n = blockIdx.x; n < a.M; n += gridDim.x) {
        if (tid < 32) {
            s_id[tid] = tid < S ? a.vid[(size_t)n * S + tid] : 0;
            s_v[tid] = tid < S ? (float)a.vval[(size_t)n * S + tid] : 0.f;
        }
        __syncthreads();
        float wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = (act && j < S) ? a.W[(size_t)s_id[j] * H + tid] : 0.f;
        float z = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fmaf(s_v[j], wc[j], z);
        const float hid = sigm(z);
        const float u = act ? a.unif[(size_t)n * H + tid] : 2.f;
        s_hs[tid] = act ? ((u < hid) ? 1.0f : floorf(hid)) : 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {
            const int j = tid >> 3, seg = tid & 7;
            float acc = 0.f;
            for (int i = seg; i < H; i += 8) acc = fmaf(s_hs[i], s_w[j][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) {
                const float vj = j < S ? sigm(acc + a.visbias[s_id[j]]) : 0.f;
                s_vis[j] = vj;
                const float d = vj - s_v[j];
                s_e[j] = j < S ? d * d : 0.f;
                if (SORTED) { if (j < S) a.visbuf[(size_t)n * S + j] = vj; }
                else if (j < S) atomicAdd(a.dvis + s_id[j], (s_v[j] - vj) * a.r_vis);
            }
        }
        __syncthreads();
        float z2 = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z2 = fmaf(s_vis[j], wc[j], z2);
        const float hid2 = sigm(z2);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float step = ((s_v[j] * hid - s_vis[j] * hid2) - a.wcost * wc[j]) * a.r_w;
            wsum[j] += step;
            if (!SORTED && act && j < S) atomicAdd(a.dW + (size_t)s_id[j] * H + tid, 2.0f * (ws0[j] + step));
        }
        if (SORTED && act) { a.hbuf[((size_t)n * 2) * H + tid] = hid; a.hbuf[((size_t)n * 2 + 1) * H + tid] = hid2; }
        hacc += hid - hid2;
        if (tid == 0) { float e = 0.f; for (int j = 0; j < 32; ++j) e += s_e[j]; err += (double)e; }
        __syncthreads();
    }
    if (act) {
        a.part_h[(size_t)blockIdx.x * H + tid] = hacc;
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < S) a.part_w[((size_t)blockIdx.x * S + j) * H + tid] = wsum[j];
    }
    if (tid == 0) a.part_e[blockIdx.x] = err;
}

// every (example, slot) of the mini-batch: take the row's accumulated delta (the first taker gets it all, later ones 0)
__global__ __launch_bounds__(256) void k_rbm_apply(float* __restrict__ W, float* __restrict__ dW, float* __restrict__ visbias,
                                                   float* __restrict__ dvis, const int32_t* __restrict__ vid, int M, int H, int S)
{
    const int n = blockIdx.x / S, j = blockIdx.x % S;
    if (n >= M) return;
    const size_t f = (size_t)vid[(size_t)n * S + j];
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        const float d = atomicExch(dW + f * H + i, 0.f);
        if (d != 0.f) atomicAdd(W + f * H + i, d);
    }
    if (threadIdx.x == 0) {
        const float d = atomicExch(dvis + f, 0.f);
        if (d != 0.f) atomicAdd(visbias + f, d);
    }
}

// k_rbm_batch<true> for S = 32 (the reference's shape) with the online kernel's instruction diet (k_rbm_sparse32): ids and values of
// the example as scalars (v_readlane), 32 unconditional row gathers, no per-element branch.  Measured at M = 4096, H = 200:
// the generic body 60 us per mini-batch, this one in profiles/r03c_rbm_kernel_stats.csv.
__global__ __launch_bounds__(256) void k_rbm_batch32(const BatchArgs a)
{
    __shared__ float s_w[32][257];
    __shared__ float s_hs[256];
    __shared__ __align__(16) float s_vis[32];
    __shared__ float s_e[32];
    const int tid = threadIdx.x, lane = tid & 63, H = a.H;
    const bool act = tid < H;
    const int col = act ? tid : 0;                          // idle threads (H < 256) shadow column 0 and never store
    if (blockIdx.x == 0 && tid == 0) *a.owner_cnt = 0;
    const float hb = act ? a.hidbias[tid] : 0.f;
    float wsum[32], hacc = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) wsum[j] = 0.f;
    const int jj = tid >> 3, seg = tid & 7;                 // 8 lanes reduce visible jj
    double err = 0.0;
    for (int n = blockIdx.x; n < a.M; n += gridDim.x) {
        const int idv = a.vid[(size_t)n * 32 + (lane & 31)];
        const float vv = (float)a.vval[(size_t)n * 32 + (lane & 31)];
        const float u = act ? a.unif[(size_t)n * H + tid] : 2.f;
        const int my_id = __shfl(idv, jj & 31);
        const float my_v = __shfl(vv, jj & 31);
        const float my_vb = a.visbias[my_id];
        int id[32]; float v[32], wc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            id[j] = __builtin_amdgcn_readlane(idv, j);
            v[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), j));
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) wc[j] = a.W[(size_t)id[j] * H + col];
        float z = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z = fmaf(v[j], wc[j], z);
        const float hid = sigm(z);
        s_hs[tid] = act ? ((u < hid) ? 1.0f : floorf(hid)) : 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s_w[j][tid] = wc[j];
        __syncthreads();
        {
            float acc = 0.f;
            for (int i = seg; i < H; i += 8) acc = fmaf(s_hs[i], s_w[jj][i], acc);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (seg == 0) {
                const float vj = sigm(acc + my_vb);
                s_vis[jj] = vj;
                const float d = vj - my_v;
                s_e[jj] = d * d;
                a.visbuf[(size_t)n * 32 + jj] = vj;
            }
        }
        __syncthreads();
        float vis[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(s_vis + 4 * q);
            vis[4 * q] = t4.x; vis[4 * q + 1] = t4.y; vis[4 * q + 2] = t4.z; vis[4 * q + 3] = t4.w;
        }
        float z2 = hb;
#pragma unroll
        for (int j = 0; j < 32; ++j) z2 = fmaf(vis[j], wc[j], z2);
        const float hid2 = sigm(z2);
#pragma unroll
        for (int j = 0; j < 32; ++j) wsum[j] += ((v[j] * hid - vis[j] * hid2) - a.wcost * wc[j]) * a.r_w;
        if (act) { a.hbuf[((size_t)n * 2) * H + tid] = hid; a.hbuf[((size_t)n * 2 + 1) * H + tid] = hid2; }
        hacc += hid - hid2;
        if (tid == 0) { float e = 0.f; for (int j = 0; j < 32; ++j) e += s_e[j]; err += (double)e; }
        __syncthreads();
    }
    if (act) {
        a.part_h[(size_t)blockIdx.x * H + tid] = hacc;
#pragma unroll
        for (int j = 0; j < 32; ++j) a.part_w[((size_t)blockIdx.x * 32 + j) * H + tid] = wsum[j];
    }
    if (tid == 0) a.part_e[blockIdx.x] = err;
}

// ---- the row update of a mini-batch WITHOUT atomics (round 3).  The (row, entry) pairs of the mini-batch -- entry = e * S + j, in
// example order -- are sorted by row with the library's stable radix sort (metrics.hip: the whole epoch's mini-batches are grouped
// ahead, a few launches per 16 mini-batches), so a row's entries are one run of `rec`, in example order.  A thread owns a 16-byte
// quarter-column of a chunk of 32 sorted entries (the A8 update's shape, scatw1_body) and RECOMPUTES every entry's delta from what
// k_rbm_batch<true> left behind -- hid_e, hid2_e (16 B each), vis_ej, v_ej -- and the positional momentum row:
//     delta_ej = 2 (momentum wstep[j] + rate_w (v_ej hid_e - vis_ej hid2_e))         (f32, as the atomic form added it)
//     W[f] <- W[f] (1 - 2 rate_w weightcost m) + sum of the run's deltas              (m = entries of the run; f64 sum, one rounding)
//     visbias[f] += rate_vis sum (v_ej - vis_ej)
// Runs inside a chunk are finished by its thread; runs that cross chunks leave partial sums and an owner record, summed in chunk
// order by k_rbm_scat2.  Fixed order everywhere: two runs give the same bits (the atomic form did not).
struct RbmScatArgs {
    const int4* rec; int n;                   // the mini-batch's sorted records, n = M * S slots (invalid ones last)
    const float* hbuf; const float* visbuf; const uint8_t* vval; const float* wstep;
    float* W; float* visbias; int H, S; float wcost, r_vis, r_w, mom;
    double* part; int4* owners; int* owner_cnt;
};
constexpr int RCH = 32;                       // sorted entries per chunk

__global__ __launch_bounds__(256) void k_rbm_scat1(const RbmScatArgs a)
{
    const int H = a.H, nq = H >> 2, PW = H + 4, nchunk = (a.n + RCH - 1) / RCH;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int chunk = (int)(gid / nq), q = (int)(gid % nq);
    if (chunk >= nchunk) return;
    const int base = chunk * RCH;
    const float two_rw = 2.0f * a.r_w, two_mom = 2.0f * a.mom;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, av = 0;
    int4 rn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rn[j] = base + j < a.n ? a.rec[base + j] : make_int4(-1, 0, 0, 0);
    for (int sb = 0; sb < RCH; sb += 8) {
        int4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = rn[j];
        if (r[0].x < 0) break;                               // invalid entries sort to the end
        if (sb + 8 < RCH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rn[j] = base + sb + 8 + j < a.n ? a.rec[base + sb + 8 + j] : make_int4(-1, 0, 0, 0);
        }
        float4 hd[8], h2[8], wsj[8], wold[8]; float v[8], vi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = r[j].x >= 0;
            const int ent = live ? r[j].y : 0, ex = ent / a.S, slot = ent % a.S;
            hd[j] = *reinterpret_cast<const float4*>(a.hbuf + ((size_t)ex * 2) * H + 4 * q);
            h2[j] = *reinterpret_cast<const float4*>(a.hbuf + ((size_t)ex * 2 + 1) * H + 4 * q);
            wsj[j] = *reinterpret_cast<const float4*>(a.wstep + (size_t)slot * H + 4 * q);
            v[j] = (float)a.vval[ent]; vi[j] = a.visbuf[ent];
            // the old row is read only where it is written: at the last entry of a run that lies inside this chunk
            const int pos = base + sb + j;
            const bool need = live && pos + 1 == r[j].w && r[j].z >= base;
            wold[j] = *reinterpret_cast<const float4*>(a.W + (size_t)(need ? r[j].x : 0) * H + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (r[j].x < 0) continue;
            a0 += (double)(two_mom * wsj[j].x + two_rw * (v[j] * hd[j].x - vi[j] * h2[j].x));
            a1 += (double)(two_mom * wsj[j].y + two_rw * (v[j] * hd[j].y - vi[j] * h2[j].y));
            a2 += (double)(two_mom * wsj[j].z + two_rw * (v[j] * hd[j].z - vi[j] * h2[j].z));
            a3 += (double)(two_mom * wsj[j].w + two_rw * (v[j] * hd[j].w - vi[j] * h2[j].w));
            av += (double)((v[j] - vi[j]) * a.r_vis);
            const int pos = base + sb + j, s = r[j].z, e = r[j].w;
            if (pos + 1 != e && pos + 1 != base + RCH) continue;       // the run goes on inside this chunk
            if (s >= base && e <= base + RCH) {                        // the whole run lies in this chunk
                const double keep = 1.0 - (double)two_rw * (double)a.wcost * (double)(e - s);
                *reinterpret_cast<float4*>(a.W + (size_t)r[j].x * H + 4 * q) =
                    make_float4((float)((double)wold[j].x * keep + a0), (float)((double)wold[j].y * keep + a1),
                                (float)((double)wold[j].z * keep + a2), (float)((double)wold[j].w * keep + a3));
                if (q == 0) a.visbias[r[j].x] = (float)((double)a.visbias[r[j].x] + av);
            } else {
                const int which = (s < base) ? 0 : 1;
                double* pp = a.part + ((size_t)chunk * 2 + which) * PW;
                pp[4 * q] = a0; pp[4 * q + 1] = a1; pp[4 * q + 2] = a2; pp[4 * q + 3] = a3;
                if (q == 0) { pp[H] = av; if (which == 1) a.owners[atomicAdd(a.owner_cnt, 1)] = make_int4(0, s, e, r[j].x); }
            }
            a0 = a1 = a2 = a3 = av = 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_rbm_scat2(const RbmScatArgs a)
{
    __shared__ double s_w[4 * 264];
    const int H = a.H, nq = H >> 2, PW = H + 4, ngrp = 256 / nq < 4 ? 256 / nq : 4;
    const int grp = threadIdx.x / nq, q = threadIdx.x % nq;
    const int n = *a.owner_cnt;
    for (int o = blockIdx.x; o < n; o += gridDim.x) {
        const int4 ow = a.owners[o];                         // {0, s, e, row}
        const int q0 = ow.y / RCH, q1 = (ow.z - 1) / RCH;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, av = 0;
        float4* p = reinterpret_cast<float4*>(a.W + (size_t)ow.w * H + 4 * q);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp == 0) w = *p;
        if (grp < ngrp) {
            for (int qq = q0 + grp; qq <= q1; qq += ngrp) {   // a group takes every ngrp-th chunk of the run, in chunk order
                const double* pp = a.part + ((size_t)qq * 2 + (qq == q0 ? 1 : 0)) * PW;
                a0 += pp[4 * q]; a1 += pp[4 * q + 1]; a2 += pp[4 * q + 2]; a3 += pp[4 * q + 3];
                if (q == 0) av += pp[H];
            }
            double* d = s_w + ((size_t)grp * (nq + 1) + q) * 4;
            d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
            if (q == 0) s_w[((size_t)grp * (nq + 1) + nq) * 4] = av;
        }
        __syncthreads();
        if (grp == 0) {
            double t0 = 0, t1 = 0, t2 = 0, t3 = 0, tv = 0;
            for (int gI = 0; gI < ngrp; ++gI) {
                const double* d = s_w + ((size_t)gI * (nq + 1) + q) * 4;
                t0 += d[0]; t1 += d[1]; t2 += d[2]; t3 += d[3];
                if (q == 0) tv += s_w[((size_t)gI * (nq + 1) + nq) * 4];
            }
            const double keep = 1.0 - 2.0 * (double)a.r_w * (double)a.wcost * (double)(ow.z - ow.y);
            *p = make_float4((float)((double)w.x * keep + t0), (float)((double)w.y * keep + t1), (float)((double)w.z * keep + t2),
                             (float)((double)w.w * keep + t3));
            if (q == 0) a.visbias[ow.w] = (float)((double)a.visbias[ow.w] + tv);
        }
        __syncthreads();
    }
}

// keys of `nmb` mini-batches (segments of M * S): row << 20 | (e * S + j), example order; entries of examples past N: invalid
__global__ __launch_bounds__(256) void k_rbm_keys(const int32_t* __restrict__ vid, int64_t n_entries, int seg_n, int nmb, unsigned long long* __restrict__ keys)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)nmb * seg_n) return;
    const size_t idx = gid;                                     // mini-batch g = gid / seg_n holds entries [g * seg_n, ...)
    unsigned long long row = GROUP_INVALID_ROW;
    if ((int64_t)idx < n_entries) { const int id = vid[idx]; if (id >= 0) row = (unsigned long long)id; }
    keys[gid] = (row << GROUP_INDEX_BITS) | (unsigned long long)(gid % seg_n);
}

// The per-workgroup partials of wstep / hidbias (up to 1,024 x 6,600 floats = 27 MB at M = 4096) in two levels, fixed order
// (deterministic).  Level 1: block (bx, by) owns 64 elements and the partials w = by (mod TAIL_NG): a thread sums every
// (16 TAIL_NG)-th partial of its element with all its loads in flight, the 16 group sums meet in LDS -> part2[by][element].
// 104 x 8 blocks instead of round 2's 104 (which ran at 0.4 TB/s: 66 us, the longest kernel of the mini-batch after the atomics
// were gone).  Level 2 (k_rbm_tail_apply): the TAIL_NG sums of an element, then the update -- or, under data parallelism, the
// all-reduce first.
constexpr int TAIL_NG = 8;
__global__ __launch_bounds__(1024) void k_rbm_batch_tail(const float* __restrict__ part_w, const float* __restrict__ part_h, const double* __restrict__ part_e,
                                                         int nwg, int H, int S, double* __restrict__ err_acc, float* __restrict__ part2)
{
    __shared__ float s_p[16][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + el, nW = S * H, n = nW + H;
    float s = 0.f;
    if (i < n) {
        const float* base = i < nW ? part_w + i : part_h + (i - nW);
        const size_t stride = i < nW ? (size_t)nW : (size_t)H;
        for (int w0 = blockIdx.y * 16 + grp; w0 < nwg; w0 += 16 * TAIL_NG * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int w = w0 + 16 * TAIL_NG * k; v[k] = w < nwg ? base[(size_t)w * stride] : 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
    }
    s_p[grp][el] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += s_p[g][el];
        part2[(size_t)blockIdx.y * n + i] = t;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        // the workgroups' error sums: a fixed-shape tree over the block (one thread walking the 1,024 partials paid a memory round
        // trip per partial -- 63 us, the whole length of this launch, as profiles/r03_rbm_kernel_stats.csv showed)
        __shared__ double s_e[1024];
        __syncthreads();
        double e = 0.0;
        for (int w = threadIdx.x; w < nwg; w += 1024) e += part_e[w];
        s_e[threadIdx.x] = e;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_e[threadIdx.x] += s_e[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) *err_acc += s_e[0];
    }
}

// sums != null: the TAIL_NG level-1 sums of every element -> sums (what data parallelism all-reduces); apply = 1: wstep / hidbias from
// `src` (part2 with ng = TAIL_NG, or the all-reduced sums with ng = 1)
__global__ void k_rbm_tail_apply(float* __restrict__ wstep, float* __restrict__ hidbias, const float* __restrict__ src, int ng, int nW, int H, int M,
                                 float mom, float r_hid, float* __restrict__ sums, int apply)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = nW + H;
    if (i >= n) return;
    float t = 0.f;
    for (int g = 0; g < ng; ++g) t += src[(size_t)g * n + i];
    if (sums) sums[i] = t;
    if (!apply) return;
    if (i < nW) wstep[i] = mom * wstep[i] + t / (float)M;
    else hidbias[i - nW] += r_hid * t;
}

// ------------------------------------------------------------------------------------------
// A7'  dense CD-1 (python :219-281) on the MFMA kernels of the FNN path.  Parameters live in
// one padded matrix Wp [Kp][Hp]: Wp[r][c] = W, row nvis = hidbias, column nhid = visbias, so
// hid = sigmoid(X' Wp) and vis = sigmoid(hs' Wp^T) pick their biases up from a ones column, and
// the two correlation products X'^T hid' - vis'^T hid2' also deliver both bias steps
// (row nvis = posact - sum hid2, column nhid = batchsum - sum vis).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_rbm_prep(const float* __restrict__ X, int n, int nvis, int Na, int Kp, int ldT,
                           T* __restrict__ Xr, T* __restrict__ XT)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // (t quad, column)
    if (gid >= (size_t)(Na / 4) * Kp) return;
    const int c = (int)(gid % Kp), t0 = (int)(gid / Kp) * 4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + i;
        v[i] = (t < n) ? (c < nvis ? X[(size_t)t * nvis + c] : (c == nvis ? 1.0f : 0.0f)) : 0.0f;
        Xr[(size_t)t * Kp + c] = (T)v[i];
    }
    store4(XT + ft_off<T>(c, t0, ldT), v[0], v[1], v[2], v[3]);
}

template <typename T>
__global__ void k_rbm_binarise(const T* __restrict__ hid, const float* __restrict__ unif, int n, int nhid,
                               int Na, int Hp, T* __restrict__ hs)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)Na * Hp) return;
    const int t = (int)(gid / Hp), c = (int)(gid % Hp);
    float v = 0.f;
    if (t < n) {
        if (c < nhid) { const float p = (float)hid[gid]; v = (unif[(size_t)t * nhid + c] < p) ? 1.0f : floorf(p); }
        else if (c == nhid) v = 1.0f;
    }
    hs[gid] = (T)v;
}

template <typename T>
__global__ void k_rbm_sqerr(const T* __restrict__ vis, const float* __restrict__ X, int n, int nvis, int Kp,
                            double* __restrict__ part)
{
    __shared__ double s[256];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n * nvis; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / nvis), c = (int)(i % nvis);
        const double d = (double)(float)vis[(size_t)t * Kp + c] - (double)X[i];
        acc += d * d;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

template <typename T>
__global__ void k_rbm_update(const float* __restrict__ slab, int splitk, size_t nslab, size_t off_neg,
                             float* __restrict__ Wp, float* __restrict__ wsp, int nvis, int nhid, int Kp, int Hp,
                             float inv_n, float wcost, float r_vis, float r_hid, float r_w, float mom, int apply,
                             T* __restrict__ wf, T* __restrict__ wtf)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Kp * Hp) return;
    const int r = (int)(i / Hp), c = (int)(i % Hp);
    float w = Wp[i];
    if (apply) {
        float corr = 0.f;
#pragma unroll 8
        for (int z = 0; z < splitk; ++z) corr += slab[(size_t)z * nslab + i] - slab[(size_t)z * nslab + off_neg + i];
        if (r < nvis && c < nhid) {                                   // :240-249
            const float step = (corr * inv_n - wcost * w) * r_w;
            const float m = wsp[i] * mom + step;
            wsp[i] = m; w += m;
        } else if (r == nvis && c < nhid) w += corr * (r_hid * inv_n);   // hidbias  (:268-273)
        else if (c == nhid && r < nvis) w += corr * (r_vis * inv_n);     // visbias  (:262-266)
        Wp[i] = w;
    }
    wf[ft_off<T>(c, r, Kp)] = (T)w;        // forward:  output column c (hidden), contraction over r
    wtf[ft_off<T>(r, c, Hp)] = (T)w;       // backward: output column r (visible), contraction over c
}

__global__ void k_bag_sum(const float* __restrict__ W0, const float* __restrict__ b0, int H, int64_t n_rows,
                          const int32_t* __restrict__ ids, int n, int F, float* __restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * H) return;
    const int t = (int)(gid / H), c = (int)(gid % H);
    float s = 0.f;
    for (int f = 0; f < F; ++f) {
        const int64_t id = ids[(size_t)t * F + f];
        if (id >= 0 && id < n_rows) s += W0[(size_t)id * H + c];
    }
    out[gid] = s + b0[c];
}
__global__ void k_affine(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                         int n, int a, int b, float* __restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * b) return;
    const int t = (int)(gid / b), c = (int)(gid % b);
    float s = 0.f;
    for (int k = 0; k < a; ++k) s = fmaf(in[(size_t)t * a + k], W[(size_t)k * b + c], s);
    out[gid] = s + bias[c];
}
__global__ void k_sigmoid(float* __restrict__ x, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) x[i] = 1.0f / (1.0f + expf(-x[i]));
}

}  // namespace

struct rbm_handle {
    int nvis = 0, nhid = 0, Kp = 0, Hp = 0, max_n = 0, Na_max = 0, dev = 0, splitk = 8;
    bool bf16 = false;
    hipStream_t st = nullptr; bool own_stream = false;
    float *Wp = nullptr, *wsp = nullptr, *slab = nullptr;
    void *wf = nullptr, *wtf = nullptr;                                   // tiled shadows (T)
    void *Xr = nullptr, *XT = nullptr, *hid = nullptr, *hidT = nullptr, *hs = nullptr, *vis = nullptr, *visT = nullptr,
         *hid2 = nullptr, *hid2T = nullptr;
    uint8_t* ones = nullptr; double* errpart = nullptr;
    size_t nW = 0;
};

namespace {

template <typename T> void rbm_refresh(rbm_handle* h, int apply, float inv_n, float wcost, float rv, float rh, float rw, float mom) {
    const size_t n = h->nW;
    hipLaunchKernelGGL((k_rbm_update<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->slab, h->splitk,
                       2 * h->nW, h->nW, h->Wp, h->wsp, h->nvis, h->nhid, h->Kp, h->Hp, inv_n, wcost, rv, rh, rw, mom,
                       apply, (T*)h->wf, (T*)h->wtf);
}

template <typename T>
int rbm_step(rbm_handle* h, const float* X, int n, const float* unif, float wcost, float rv, float rh, float rw,
             float mom, double* sq_err_out)
{
    const int Na = rup(n, 256), Kp = h->Kp, Hp = h->Hp, ldT = h->Na_max;
    T *Xr = (T*)h->Xr, *XT = (T*)h->XT, *hid = (T*)h->hid, *hidT = (T*)h->hidT, *hs = (T*)h->hs, *vis = (T*)h->vis,
      *visT = (T*)h->visT, *hid2 = (T*)h->hid2, *hid2T = (T*)h->hid2T;
    {
        const size_t nt = (size_t)(Na / 4) * Kp;
        hipLaunchKernelGGL((k_rbm_prep<T>), dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->st, X, n, h->nvis, Na, Kp,
                           ldT, Xr, XT);
    }
    {   // hid = sigmoid(X' Wp)   (hid_activate, mean field)
        EpiFwd<T> e{hid, Hp, hidT, ldT, h->ones, ACT_SIGMOID, h->nhid, n};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiFwd<T>>), dim3(Na / 64, Hp / 64, 1), dim3(256), 0, h->st, Xr, Kp,
                           (const T*)h->wf, Kp, e);
    }
    {
        const size_t nt = (size_t)Na * Hp;
        hipLaunchKernelGGL((k_rbm_binarise<T>), dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->st, hid, unif, n,
                           h->nhid, Na, Hp, hs);
    }
    {   // vis = sigmoid(hs' Wp^T)   (mean-field visibles)
        EpiFwd<T> e{vis, Kp, visT, ldT, h->ones, ACT_SIGMOID, h->nvis, n};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiFwd<T>>), dim3(Na / 64, Kp / 64, 1), dim3(256), 0, h->st, hs, Hp,
                           (const T*)h->wtf, Hp, e);
    }
    {   // hid2 = sigmoid(vis' Wp)   (mean-field hiddens)
        EpiFwd<T> e{hid2, Hp, hid2T, ldT, h->ones, ACT_SIGMOID, h->nhid, n};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiFwd<T>>), dim3(Na / 64, Hp / 64, 1), dim3(256), 0, h->st, vis, Kp,
                           (const T*)h->wf, Kp, e);
    }
    {   // poscorr = X'^T hid',  negcorr = vis'^T hid2'   (contraction over the examples)
        WgradArgs wa;
        wa.p[0] = WgradProb{XT, hidT, h->slab, Kp / 64, Hp / 64, Hp};
        wa.p[1] = WgradProb{visT, hid2T, h->slab + h->nW, Kp / 64, Hp / 64, Hp};
        wa.p[2] = WgradProb{nullptr, nullptr, nullptr, 0, 1, 64};
        wa.p[3] = WgradProb{nullptr, nullptr, nullptr, 0, 1, 64};
        wa.ldT = ldT; wa.klen = Na / h->splitk; wa.zstride = 2 * h->nW;
        const int nb = 2 * (Kp / 64) * (Hp / 64);
        hipLaunchKernelGGL((k_wgrad<T>), dim3(nb, h->splitk), dim3(256), 0, h->st, wa);
    }
    if (sq_err_out)
        hipLaunchKernelGGL((k_rbm_sqerr<T>), dim3(256), dim3(256), 0, h->st, vis, X, n, h->nvis, Kp, h->errpart);
    rbm_refresh<T>(h, 1, 1.0f / (float)n, wcost, rv, rh, rw, mom);
    RCK(hipGetLastError());
    if (sq_err_out) {
        std::vector<double> part(256);
        RCK(hipMemcpyAsync(part.data(), h->errpart, 256 * sizeof(double), hipMemcpyDeviceToHost, h->st));
        RCK(hipStreamSynchronize(h->st));
        double s = 0; for (double v : part) s += v;
        *sq_err_out = s;
    }
    return FNN_OK;
}

}  // namespace

extern "C" {

const char* rbm_last_error(void) { return g_err.c_str(); }

int rbm_sparse_epoch(float* W, float* visbias, float* hidbias, float* wstep, const int32_t* vid,
                     const uint8_t* vval, const float* unif, int64_t N, int H, int S, float weightcost,
                     float rate_vis, float rate_hid, float rate_w, float momentum, double* sq_err_out, void* stream)
{
    if (!W || !visbias || !hidbias || !wstep || !vid || !vval || !unif) RFAIL(FNN_ERR_ARG, "null pointer");
    if (H < 1 || H > 256 || S < 1 || S > 32 || N < 1) RFAIL(FNN_ERR_ARG, "need 1 <= H <= 256, 1 <= S <= 32, N >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) RFAIL(FNN_ERR_HIP, "no HIP device (no CPU fallback)");
    hipStream_t st = (hipStream_t)stream;
    double* d_err = nullptr;
    RCK(hipMalloc((void**)&d_err, sizeof(double)));
    SparseArgs a{W, visbias, hidbias, wstep, vid, vval, unif, N, H, S, weightcost, rate_vis, rate_hid, rate_w,
                 momentum, d_err};
    static const bool generic = getenv("RBM_SPARSE_GENERIC") != nullptr;       // diagnostics: the general-S kernel at S = 32
    if (S == 32 && !generic) hipLaunchKernelGGL(k_rbm_sparse32, dim3(1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_rbm_sparse, dim3(1), dim3(256), 0, st, a);
    RCK(hipGetLastError());
    double e = 0;
    RCK(hipMemcpyAsync(&e, d_err, sizeof(double), hipMemcpyDeviceToHost, st));
    RCK(hipStreamSynchronize(st));
    hipFree(d_err);
    if (sq_err_out) *sq_err_out = e;
    return FNN_OK;
}

static int sparse_batch_impl(float* W, float* dW, float* visbias, float* dvis, float* hidbias, float* wstep, const int32_t* vid,
                             const uint8_t* vval, const float* unif, int64_t N, int M, int M_global, int H, int S, float weightcost, float rate_vis,
                             float rate_hid, float rate_w, float momentum, rbm_allreduce_fn allreduce, void* ctx, double* sq_err_out, void* stream)
{
    if (!W || !dW || !visbias || !dvis || !hidbias || !wstep || !vid || !vval || !unif) RFAIL(FNN_ERR_ARG, "null pointer");
    if (H < 1 || H > 256 || S < 1 || S > 32 || N < 1 || M < 1) RFAIL(FNN_ERR_ARG, "need 1 <= H <= 256, 1 <= S <= 32, N >= 1, M >= 1");
    if (allreduce && M_global < M) RFAIL(FNN_ERR_ARG, "M_global must be the GLOBAL mini-batch length (>= M)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) RFAIL(FNN_ERR_HIP, "no HIP device (no CPU fallback)");
    hipStream_t st = (hipStream_t)stream;
    int nwg_cap = 512;                                         // 2 workgroups per CU, several examples each (measured at M = 4096: 512 -> 0.137 ms per
                                                               // mini-batch, 1024 -> 0.141, 2048 -> 0.150: fewer partials for the tail to sum)
    if (const char* e = getenv("RBM_BATCH_WGS")) nwg_cap = std::max(64, std::min(4096, atoi(e)));     // tuning knob
    const int nwg_max = (int)std::min<int64_t>(M, nwg_cap);
    // the row update without atomics needs 16-byte quarter-columns (H % 4 == 0) and entry indices that fit the key (M * S <= 2^20);
    // anything else keeps the atomic form ($RBM_BATCH_ATOMICS=1 forces it: A/B measurements)
    const char* ev = getenv("RBM_BATCH_ATOMICS");
    const bool sorted = H % 4 == 0 && (int64_t)M * S <= ((int64_t)1 << GROUP_INDEX_BITS) && !(ev && ev[0] == '1');
    float *part_w = nullptr, *part_h = nullptr; double *part_e = nullptr, *d_err = nullptr;
    // scratch: ONE arena per host thread, grown on demand and kept between calls (a dozen hipMalloc / hipFree pairs per call cost
    // more than a mini-batch: 0.29 -> 0.2x ms per mini-batch of 4096 in the bench, which calls once per 16 mini-batches)
    struct Arena { char* base = nullptr; size_t cap = 0, used = 0; int dev = -1; ~Arena() { /* freed with the process */ } };
    static thread_local Arena arena;
    int cur_dev = 0; hipGetDevice(&cur_dev);
    size_t need_total = 0;
    auto rup256 = [](size_t b) { return (b + 255) / 256 * 256; };
    {   // everything this call will carve out of the arena (the same expressions as the RAL lines below)
        const int seg_n0 = M * S, nchunk0 = (seg_n0 + RCH - 1) / RCH, G0 = 16;
        need_total = rup256((size_t)nwg_max * S * H * 4) + rup256((size_t)nwg_max * H * 4) + rup256((size_t)nwg_max * 8) + rup256(8) + rup256((size_t)(S * H + H) * 4) + rup256((size_t)8 * (S * H + H) * 4);
        if (sorted) need_total += rup256((size_t)M * 2 * H * 4) + rup256((size_t)seg_n0 * 4) + rup256((size_t)nchunk0 * 2 * (H + 4) * 8) + rup256((size_t)nchunk0 * sizeof(int4)) +
                                  rup256(4) + 2 * rup256((size_t)G0 * seg_n0 * 8) + rup256((size_t)G0 * seg_n0 * sizeof(int4)) + rup256(radix_sort_hist_bytes(G0, seg_n0));
    }
    if (arena.dev != cur_dev || arena.cap < need_total) {
        if (arena.base) { hipDeviceSynchronize(); hipFree(arena.base); arena.base = nullptr; arena.cap = 0; }
        void* q = nullptr;
        if (hipMalloc(&q, need_total) != hipSuccess) { (void)hipGetLastError(); RFAIL(FNN_ERR_NOMEM, "rbm_sparse_batch: hipMalloc of the scratch arena failed"); }
        arena.base = static_cast<char*>(q); arena.cap = need_total; arena.dev = cur_dev;
    }
    arena.used = 0;
    auto cleanup = [&]() {};
#define RAL(ptr, bytes) do { ptr = reinterpret_cast<decltype(ptr)>(arena.base + arena.used); arena.used += rup256((size_t)(bytes)); } while (0)
    RAL(part_w, (size_t)nwg_max * S * H * 4); RAL(part_h, (size_t)nwg_max * H * 4); RAL(part_e, (size_t)nwg_max * 8); RAL(d_err, 8);
    float *sums = nullptr, *part2 = nullptr;
    if (allreduce) RAL(sums, (size_t)(S * H + H) * 4);
    RAL(part2, (size_t)TAIL_NG * (S * H + H) * 4);
    RCK(hipMemsetAsync(d_err, 0, 8, st));
    const int seg_n = M * S, nchunk = (seg_n + RCH - 1) / RCH, GROUP = 16;
    float *hbuf = nullptr, *visbuf = nullptr; double* spart = nullptr; int4 *owners = nullptr, *rec = nullptr; int* owner_cnt = nullptr;
    unsigned long long *keys = nullptr, *keys2 = nullptr; unsigned* hist = nullptr;
    if (sorted) {
        RAL(hbuf, (size_t)M * 2 * H * 4); RAL(visbuf, (size_t)seg_n * 4); RAL(spart, (size_t)nchunk * 2 * (H + 4) * 8);
        RAL(owners, (size_t)nchunk * sizeof(int4)); RAL(owner_cnt, 4);
        RAL(keys, (size_t)GROUP * seg_n * 8); RAL(keys2, (size_t)GROUP * seg_n * 8); RAL(rec, (size_t)GROUP * seg_n * sizeof(int4));
        RAL(hist, radix_sort_hist_bytes(GROUP, seg_n));
    }
    const int rbits = 31;                                       // the whole row field of the key (ids are int32; the all-ones row of an invalid entry sorts last): 4 passes
    int64_t mb = 0;
    for (int64_t n0 = 0; n0 < N; n0 += M, ++mb) {
        const int m = (int)std::min<int64_t>(M, N - n0), nwg = std::min(m, nwg_max);
        const int4* rec_mb = nullptr;
        if (sorted) {
            if (mb % GROUP == 0) {                              // group the next GROUP mini-batches' (row, entry) pairs by row
                const int nmb = (int)std::min<int64_t>(GROUP, (N - n0 + M - 1) / M);
                const size_t tot = (size_t)nmb * seg_n;
                hipLaunchKernelGGL(k_rbm_keys, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, vid + n0 * S, (N - n0) * S, seg_n, nmb, keys);
                const unsigned long long* srt = radix_sort_segments(st, keys, keys2, hist, nmb, seg_n, GROUP_INDEX_BITS, GROUP_INDEX_BITS + rbits);
                group_records(st, srt, nmb, seg_n, rec);
            }
            rec_mb = rec + (size_t)(mb % GROUP) * seg_n;
        }
        BatchArgs a{W, visbias, hidbias, wstep, dW, dvis, vid + n0 * S, vval + n0 * S, unif + n0 * H, m, H, S, weightcost, rate_vis, rate_w,
                    momentum, part_w, part_h, part_e, hbuf, visbuf, nullptr};
        if (sorted) {
            a.owner_cnt = owner_cnt;                               // zeroed by workgroup 0 of the compute launch (a fill launch costs 4.6 us)
            if (S == 32 && !getenv("RBM_BATCH_GENERIC")) hipLaunchKernelGGL(k_rbm_batch32, dim3(nwg), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_rbm_batch<true>, dim3(nwg), dim3(256), 0, st, a);
            RbmScatArgs sa{rec_mb, seg_n, hbuf, visbuf, vval + n0 * S, wstep, W, visbias, H, S, weightcost, rate_vis, rate_w, momentum,
                           spart, owners, owner_cnt};
            const long nthr = (long)nchunk * (H / 4);
            hipLaunchKernelGGL(k_rbm_scat1, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, sa);
            hipLaunchKernelGGL(k_rbm_scat2, dim3(256), dim3(256), 0, st, sa);
        } else {
            hipLaunchKernelGGL(k_rbm_batch<false>, dim3(nwg), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_rbm_apply, dim3((unsigned)(m * S)), dim3(64), 0, st, W, dW, visbias, dvis, vid + n0 * S, m, H, S);
        }
        const int nel = S * H + H;
        hipLaunchKernelGGL(k_rbm_batch_tail, dim3((unsigned)((nel + 63) / 64), TAIL_NG), dim3(1024), 0, st, part_w, part_h, part_e, nwg, H, S, d_err, part2);
        // level 2: the TAIL_NG sums of every element, then the update (single process) or the sums for the all-reduce
        hipLaunchKernelGGL(k_rbm_tail_apply, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, wstep, hidbias, part2, TAIL_NG, S * H, H, m, momentum,
                           rate_hid, sums, allreduce ? 0 : 1);
        if (allreduce) {
            // every rank runs the same number of mini-batches (its shard of each): the global count is what the mean divides by.
            // A short LAST mini-batch: the caller passes shards of the same global tail, so the ratio m / M carries over
            if (allreduce(ctx, sums, (int64_t)(S * H + H), (void*)st) != 0) { cleanup(); RFAIL(FNN_ERR_HIP, "rbm_sparse_batch_dp: the all-reduce callback failed"); }
            const int mg = (int)((int64_t)M_global * m / M);
            hipLaunchKernelGGL(k_rbm_tail_apply, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, wstep, hidbias, sums, 1, S * H, H, mg > 0 ? mg : 1,
                               momentum, rate_hid, (float*)nullptr, 1);
        }
    }
    hipError_t le = hipGetLastError();
    double e = 0;
    if (le == hipSuccess) le = hipMemcpyAsync(&e, d_err, sizeof(double), hipMemcpyDeviceToHost, st);
    if (le == hipSuccess) le = hipStreamSynchronize(st);
    cleanup();
#undef RAL
    if (le != hipSuccess) { g_err = std::string("rbm_sparse_batch: ") + hipGetErrorString(le); return FNN_ERR_HIP; }
    if (sq_err_out) *sq_err_out = e;
    return FNN_OK;
}

int rbm_sparse_batch(float* W, float* dW, float* visbias, float* dvis, float* hidbias, float* wstep, const int32_t* vid,
                     const uint8_t* vval, const float* unif, int64_t N, int M, int H, int S, float weightcost, float rate_vis,
                     float rate_hid, float rate_w, float momentum, double* sq_err_out, void* stream)
{
    return sparse_batch_impl(W, dW, visbias, dvis, hidbias, wstep, vid, vval, unif, N, M, M, H, S, weightcost, rate_vis, rate_hid, rate_w, momentum,
                             nullptr, nullptr, sq_err_out, stream);
}

int rbm_sparse_batch_dp(float* W, float* dW, float* visbias, float* dvis, float* hidbias, float* wstep, const int32_t* vid,
                        const uint8_t* vval, const float* unif, int64_t N, int M, int M_global, int H, int S, float weightcost, float rate_vis,
                        float rate_hid, float rate_w, float momentum, rbm_allreduce_fn allreduce, void* ctx, double* sq_err_out, void* stream)
{
    if (!allreduce) RFAIL(FNN_ERR_ARG, "rbm_sparse_batch_dp: null all-reduce callback");
    return sparse_batch_impl(W, dW, visbias, dvis, hidbias, wstep, vid, vval, unif, N, M, M_global, H, S, weightcost, rate_vis, rate_hid, rate_w,
                             momentum, allreduce, ctx, sq_err_out, stream);
}

int rbm_dense_create(int nvis, int nhid, int max_n, int precision, int device, void* stream, rbm_handle** out)
{
    if (!out) RFAIL(FNN_ERR_ARG, "null out");
    *out = nullptr;
    if (nvis < 1 || nvis > 4095 || nhid < 1 || nhid > 4095 || max_n < 1) RFAIL(FNN_ERR_ARG, "bad shape");
    if (precision != FNN_PREC_F32 && precision != FNN_PREC_BF16) RFAIL(FNN_ERR_ARG, "bad precision (FNN_PREC_F32 or FNN_PREC_BF16)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) RFAIL(FNN_ERR_HIP, "no HIP device (no CPU fallback)");
    if (device < 0 || device >= ndev) RFAIL(FNN_ERR_ARG, "device ordinal out of range");
    rbm_handle* h = new rbm_handle();
    h->nvis = nvis; h->nhid = nhid; h->Kp = rup(nvis + 1, 64); h->Hp = rup(nhid + 1, 64);
    h->max_n = max_n; h->Na_max = rup(max_n, 256); h->dev = device; h->bf16 = precision == FNN_PREC_BF16;
    h->nW = (size_t)h->Kp * h->Hp;
    RCK(hipSetDevice(device));
    if (stream) h->st = (hipStream_t)stream; else { RCK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    const size_t ts = h->bf16 ? 2 : 4, Na = h->Na_max;
    auto al = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, h->st); return e; };
    RCK(al((void**)&h->Wp, h->nW * 4)); RCK(al((void**)&h->wsp, h->nW * 4));
    RCK(al((void**)&h->slab, (size_t)h->splitk * 2 * h->nW * 4));
    RCK(al(&h->wf, h->nW * ts)); RCK(al(&h->wtf, h->nW * ts));
    RCK(al(&h->Xr, Na * h->Kp * ts)); RCK(al(&h->XT, Na * h->Kp * ts));
    RCK(al(&h->hid, Na * h->Hp * ts)); RCK(al(&h->hidT, Na * h->Hp * ts)); RCK(al(&h->hs, Na * h->Hp * ts));
    RCK(al(&h->vis, Na * h->Kp * ts)); RCK(al(&h->visT, Na * h->Kp * ts));
    RCK(al(&h->hid2, Na * h->Hp * ts)); RCK(al(&h->hid2T, Na * h->Hp * ts));
    RCK(hipMalloc((void**)&h->ones, (size_t)(h->Kp > h->Hp ? h->Kp : h->Hp)));
    RCK(hipMemsetAsync(h->ones, 1, (size_t)(h->Kp > h->Hp ? h->Kp : h->Hp), h->st));
    RCK(al((void**)&h->errpart, 256 * sizeof(double)));
    RCK(hipStreamSynchronize(h->st));
    *out = h;
    return FNN_OK;
}

int rbm_dense_destroy(rbm_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    hipSetDevice(h->dev);
    hipStreamSynchronize(h->st);
    void* ptrs[] = {h->Wp, h->wsp, h->slab, h->wf, h->wtf, h->Xr, h->XT, h->hid, h->hidT, h->hs, h->vis, h->visT,
                    h->hid2, h->hid2T, h->ones, h->errpart};
    for (void* p : ptrs) if (p) hipFree(p);
    if (h->own_stream) hipStreamDestroy(h->st);
    delete h;
    return FNN_OK;
}

int rbm_dense_set(rbm_handle* h, const float* W, const float* visbias, const float* hidbias)
{
    if (!h || !W || !visbias || !hidbias) RFAIL(FNN_ERR_ARG, "null pointer");
    RCK(hipSetDevice(h->dev));
    std::vector<float> p(h->nW, 0.f);
    for (int r = 0; r < h->nvis; ++r) {
        memcpy(&p[(size_t)r * h->Hp], &W[(size_t)r * h->nhid], (size_t)h->nhid * 4);
        p[(size_t)r * h->Hp + h->nhid] = visbias[r];
    }
    memcpy(&p[(size_t)h->nvis * h->Hp], hidbias, (size_t)h->nhid * 4);
    RCK(hipStreamSynchronize(h->st));
    RCK(hipMemcpy(h->Wp, p.data(), h->nW * 4, hipMemcpyHostToDevice));
    RCK(hipMemsetAsync(h->wsp, 0, h->nW * 4, h->st));
    if (h->bf16) rbm_refresh<bf16_t>(h, 0, 0, 0, 0, 0, 0, 0); else rbm_refresh<float>(h, 0, 0, 0, 0, 0, 0, 0);
    RCK(hipStreamSynchronize(h->st));
    return FNN_OK;
}

int rbm_dense_get(rbm_handle* h, float* W, float* visbias, float* hidbias)
{
    if (!h || !W || !visbias || !hidbias) RFAIL(FNN_ERR_ARG, "null pointer");
    RCK(hipSetDevice(h->dev));
    std::vector<float> p(h->nW);
    RCK(hipStreamSynchronize(h->st));
    RCK(hipMemcpy(p.data(), h->Wp, h->nW * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < h->nvis; ++r) {
        memcpy(&W[(size_t)r * h->nhid], &p[(size_t)r * h->Hp], (size_t)h->nhid * 4);
        visbias[r] = p[(size_t)r * h->Hp + h->nhid];
    }
    memcpy(hidbias, &p[(size_t)h->nvis * h->Hp], (size_t)h->nhid * 4);
    return FNN_OK;
}

int rbm_dense_cd1(rbm_handle* h, const float* X, int n, const float* unif, float weightcost, float rate_vis,
                  float rate_hid, float rate_w, float momentum, double* sq_err_out)
{
    if (!h || !X || !unif) RFAIL(FNN_ERR_ARG, "null pointer");
    if (n < 1 || n > h->max_n) RFAIL(FNN_ERR_ARG, "n must be in [1, max_n]");
    RCK(hipSetDevice(h->dev));
    return h->bf16 ? rbm_step<bf16_t>(h, X, n, unif, weightcost, rate_vis, rate_hid, rate_w, momentum, sq_err_out)
                   : rbm_step<float>(h, X, n, unif, weightcost, rate_vis, rate_hid, rate_w, momentum, sq_err_out);
}

int rbm_bag_sum(const float* W0, const float* b0, int H, int64_t n_rows, const int32_t* ids, int n, int F,
                float* out, void* stream)
{
    if (!W0 || !b0 || !ids || !out || n < 1) RFAIL(FNN_ERR_ARG, "bad argument");
    const size_t nt = (size_t)n * H;
    hipLaunchKernelGGL(k_bag_sum, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W0, b0, H, n_rows,
                       ids, n, F, out);
    RCK(hipGetLastError());
    return FNN_OK;
}

int rbm_affine(const float* in, const float* W, const float* bias, int n, int a, int b, float* out, void* stream)
{
    if (!in || !W || !bias || !out || n < 1) RFAIL(FNN_ERR_ARG, "bad argument");
    const size_t nt = (size_t)n * b;
    hipLaunchKernelGGL(k_affine, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, W, bias, n, a,
                       b, out);
    RCK(hipGetLastError());
    return FNN_OK;
}

int rbm_sigmoid(float* x, int64_t count, void* stream)
{
    if (!x || count < 1) RFAIL(FNN_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_sigmoid, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, count);
    RCK(hipGetLastError());
    return FNN_OK;
}

}  // extern "C"
