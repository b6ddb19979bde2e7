// fnn_step_kernels.hip.h -- the three launches of a training step (gfx950 only).
//
// One pass of the reference's hot loop body (python/FNN_wnzh.py:296-306) is three kernels on one
// stream, each a union of two independent roles that run side by side on different workgroups:
//
//   k_step1 = { MLP strip kernel: gather, forward, loss, backward-data }
//   k_step2 = { weight-gradient products } U { sparse-row SGD, level 1 } U { sort quarters of the NEXT batch's keys }
//   k_step3 = { slab reduce + dense SGD + shadow refresh } U { sparse-row SGD, level 2 } U { merge the quarters }
//
// The step is a few microseconds of math, so its cost is launches and dependent memory round
// trips; three fat launches and no cross-stream events are what that regime wants (measured:
// a 7-node hipGraph replay costs 46 us on this runtime, seven plain launches 18.5 us).
#pragma once
#include "fnn_kernels.hip.h"

namespace fnn {

// ------------------------------------------------------------------------------------------
// Grouping role: bitonic sort of one field's 4096 (row, t) keys by 256 threads x 16 keys.
// Strides below 16 are compare-exchanges inside a thread, strides 16..512 use wave shuffles,
// only strides 1024 and 2048 go through LDS.  32-bit keys (row << 12 | t) when n_rows * 4096
// fits, else 64-bit.  Segment bounds [s, e) come from a max-scan / min-scan of the head flags.
// ------------------------------------------------------------------------------------------
template <typename KT> struct KeyTraits;
template <> struct KeyTraits<unsigned> { static constexpr int SH = 12; };
template <> struct KeyTraits<unsigned long long> { static constexpr int SH = 32; };

struct SortArgs { const int32_t* ids; int B, F; int64_t n_rows; int4* rec; int* owner_cnt; int nblk; void* skeys; };

constexpr int SORT_N = 4096;     // keys per field handled by the union-kernel path (B <= 4096)

template <typename KT> __host__ __device__ constexpr size_t sort_lds_bytes() {
    return (size_t)SORT_N * sizeof(KT) + 256 * sizeof(KT) + 2 * 256 * sizeof(int);
}

// Phase A of the split sort: one workgroup sorts a quarter (1024 keys, 256 threads x 4) of a
// field's keys, ascending or descending as the full network would at k = 1024, and stores it.
template <typename KT>
__device__ __forceinline__ void sortA_body(const SortArgs& so, const int blk, unsigned char* smem)
{
    constexpr int SH = KeyTraits<KT>::SH;
    const KT INV = ~(KT)0;
    KT* s_key = reinterpret_cast<KT*>(smem);                 // [1024]
    const int tid = threadIdx.x, F = so.F, B = so.B;
    const int f = blk >> 2, base = (blk & 3) * 1024;
    KT key[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                             // coalesced over tid; initial order is free
        const int t = base + a * 256 + tid;
        KT kk = INV;
        if (t < B) {
            const int64_t id = so.ids[(size_t)t * F + f];
            if (id >= 0 && id < so.n_rows) kk = ((KT)id << SH) | (KT)t;
        }
        key[a] = kk;
    }
    const int i0 = base + tid * 4;                            // global position of key[0]
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 4) {
#pragma unroll
                for (int jj = 2; jj > 0; jj >>= 1) {
                    if (j == jj) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int b = a ^ jj;
                            if (b > a) {
                                const bool up = ((i0 + a) & k) == 0;
                                const KT x = key[a], y = key[b];
                                const KT mn = x < y ? x : y, mx = x < y ? y : x;
                                key[a] = up ? mn : mx; key[b] = up ? mx : mn;
                            }
                        }
                    }
                }
            } else {
                const bool keepmin = ((i0 & j) == 0) == ((i0 & k) == 0);
                if (j < 256) {
                    const int d = j >> 2;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const KT other = __shfl_xor(key[a], d);
                        const KT mine = key[a];
                        const KT mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                        key[a] = keepmin ? mn : mx;
                    }
                } else {
                    __syncthreads();
#pragma unroll
                    for (int a = 0; a < 4; ++a) s_key[tid * 4 + a] = key[a];
                    __syncthreads();
                    const int pt = (tid ^ (j >> 2)) * 4;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const KT other = s_key[pt + a];
                        const KT mine = key[a];
                        const KT mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                        key[a] = keepmin ? mn : mx;
                    }
                }
            }
        }
    }
    KT* out = static_cast<KT*>(so.skeys) + (size_t)f * SORT_N + i0;
#pragma unroll
    for (int a = 0; a < 4; ++a) out[a] = key[a];
}

// The whole sort (MERGE = false) or phase B of the split sort (MERGE = true: the keys come from
// phase A's four sorted quarters and only the merge levels k = 2048, 4096 remain), then the
// segment bounds.
template <typename KT, bool MERGE>
__device__ __forceinline__ void sort16_body(const SortArgs& so, const int f, unsigned char* smem)
{
    constexpr int SH = KeyTraits<KT>::SH;
    const KT INV = ~(KT)0;
    KT* s_key = reinterpret_cast<KT*>(smem);                 // [4096]
    KT* s_last = s_key + SORT_N;                              // [256]
    int* s_a = reinterpret_cast<int*>(s_last + 256);          // [256]
    int* s_b = s_a + 256;                                     // [256]
    const int tid = threadIdx.x, F = so.F, B = so.B;
    if (f == 0 && tid == 0) *so.owner_cnt = 0;
    KT key[16];
    if (MERGE) {
        const KT* in = static_cast<const KT*>(so.skeys) + (size_t)f * SORT_N + tid * 16;
#pragma unroll
        for (int a = 0; a < 16; ++a) key[a] = in[a];
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {                        // coalesced over tid; initial order is free
            const int t = a * 256 + tid;
            KT kk = INV;
            if (t < B) {
                const int64_t id = so.ids[(size_t)t * F + f];
                if (id >= 0 && id < so.n_rows) kk = ((KT)id << SH) | (KT)t;
            }
            key[a] = kk;
        }
    }
    for (int k = MERGE ? 2048 : 2; k <= SORT_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 16) {
#pragma unroll
                for (int jj = 8; jj > 0; jj >>= 1) {
                    if (j == jj) {
#pragma unroll
                        for (int a = 0; a < 16; ++a) {
                            const int b = a ^ jj;
                            if (b > a) {
                                const bool up = ((tid * 16 + a) & k) == 0;
                                const KT x = key[a], y = key[b];
                                const KT mn = x < y ? x : y, mx = x < y ? y : x;
                                key[a] = up ? mn : mx; key[b] = up ? mx : mn;
                            }
                        }
                    }
                }
            } else if (j < 1024) {
                const int d = j >> 4;
                const bool keepmin = (((tid << 4) & j) == 0) == (((tid << 4) & k) == 0);
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const KT other = __shfl_xor(key[a], d);
                    const KT mine = key[a];
                    const KT mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    key[a] = keepmin ? mn : mx;
                }
            } else {
                const bool keepmin = (((tid << 4) & j) == 0) == (((tid << 4) & k) == 0);
                __syncthreads();
#pragma unroll
                for (int a = 0; a < 16; ++a) s_key[tid * 16 + a] = key[a];
                __syncthreads();
                const int pt = (tid ^ (j >> 4)) * 16;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const KT other = s_key[pt + a];
                    const KT mine = key[a];
                    const KT mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    key[a] = keepmin ? mn : mx;
                }
            }
        }
    }
    // ---- segment bounds: s = position of the last head at or before p, e = next head after p
    s_last[tid] = key[15];
    __syncthreads();
    const KT prev = tid > 0 ? s_last[tid - 1] : INV;
    const int p0 = tid * 16;
    unsigned headmask = 0;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const KT pk = a == 0 ? prev : key[a - 1];
        const bool head = (a == 0 && tid == 0) || ((pk >> SH) != (key[a] >> SH));
        headmask |= head ? (1u << a) : 0u;
    }
    const int last_head = headmask ? p0 + 31 - __builtin_clz(headmask) : -1;
    const int first_head = headmask ? p0 + __builtin_ctz(headmask) : SORT_N;
    // exclusive prefix-max of last_head and exclusive suffix-min of first_head over the 256 threads
    s_a[tid] = last_head; s_b[tid] = first_head;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int va = (tid >= o) ? s_a[tid - o] : -1;
        const int vb = (tid + o < 256) ? s_b[tid + o] : SORT_N;
        __syncthreads();
        s_a[tid] = max(s_a[tid], va); s_b[tid] = min(s_b[tid], vb);
        __syncthreads();
    }
    const int carry_s = tid > 0 ? s_a[tid - 1] : -1;
    const int carry_e = tid < 255 ? s_b[tid + 1] : SORT_N;
    int4* out = so.rec + (size_t)f * SORT_N + p0;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const unsigned below = headmask & ((2u << a) - 1u);           // heads at positions <= a
        const unsigned above = a == 15 ? 0u : (headmask >> (a + 1));  // heads at positions > a
        const int s = below ? p0 + 31 - __builtin_clz(below) : carry_s;
        const int e = above ? p0 + a + 1 + __builtin_ctz(above) : carry_e;
        int4 r = make_int4(-1, 0, 0, 0);
        if (key[a] != INV) r = make_int4((int)(key[a] >> SH), (int)(key[a] & (((KT)1 << SH) - 1)), s, e);
        out[a] = r;
    }
}

template <typename KT>
static __global__ __launch_bounds__(256) void k_sort16(const SortArgs so)
{
    extern __shared__ __align__(16) unsigned char smem[];
    sort16_body<KT, false>(so, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// step 1: MLP strips of this batch.  (At 256 VGPRs a strip workgroup fills its CU, so the sort
// roles ride on steps 2 and 3, whose workgroups are small enough to share a CU.)
// ------------------------------------------------------------------------------------------
template <typename T, int C1, int C2, int CX, bool BAG>
static __global__ __launch_bounds__(256) void k_step1(const MlpArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    mlp_body<T, C1, C2, CX, BAG>(a, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// step 2: quarter-sorts of the NEXT batch's keys  U  weight gradients  U  sparse-row SGD level 1
// ------------------------------------------------------------------------------------------
template <typename T, typename KT>
static __global__ __launch_bounds__(256) void k_step2(const SortArgs so, const WgradArgs wa, const int nwx,
                                               const int splitk, const ScatArgs sa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x, nw = nwx * splitk;
    if (b < so.nblk) sortA_body<KT>(so, b, smem);                              // so.nblk = 4 * F or 0
    else if (b < so.nblk + nw) wgrad_body<T>(wa, (b - so.nblk) % nwx, (b - so.nblk) / nwx);
    else if (sa.rw == SLOT) scat1_body(sa, b - so.nblk - nw);
    else scatw1_body(sa, b - so.nblk - nw);
}

// ------------------------------------------------------------------------------------------
// step 3: slab reduce (+ L2 term) -> bucket, optional dense SGD + shadow refresh, loss sum
//         U  sparse-row SGD level 2.  UPDATE = false under data parallelism: the bucket is
//         all-reduced first and k_update applies it.
// ------------------------------------------------------------------------------------------
struct TailArgs {
    const float* slab; int splitk; size_t nw_all, nw12, nslab; float* master; float lambda1; int reg_all;
    const float* loss_t; int Ba; float* bucket; float* loss_sum; float lr; int K1p, H1p, H2p;
    void *w1, *w1t, *w2, *w2t; int nblk_red;
    float* bb0; size_t nbag, off_bag;      // bag mode: bias vector, its length (K1p) and slab offset
};

template <typename T, bool UPDATE, typename KT>
static __global__ __launch_bounds__(256) void k_step3(const SortArgs so, const TailArgs ta, const ScatArgs sa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_sum[16][16];
    if ((int)blockIdx.x < so.nblk) { sort16_body<KT, true>(so, blockIdx.x, smem); return; }   // merge: F or 0 WGs
    const int b = (int)blockIdx.x - so.nblk;
    if (b >= ta.nblk_red) {
        const int nb = (int)gridDim.x - so.nblk - ta.nblk_red;
        if (sa.rw == SLOT) scat2_body(sa, b - ta.nblk_red, nb, s_sum);
        else scatw2_body(sa, b - ta.nblk_red, nb, reinterpret_cast<double*>(smem));
        return;
    }
    if (b == ta.nblk_red - 1) {                                // loss: fixed-shape tree
        float* s_l = reinterpret_cast<float*>(&s_sum[0][0]);
        float v = 0.f;
        for (int i = threadIdx.x; i < ta.Ba; i += 256) v += ta.loss_t[i];
        s_l[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_l[threadIdx.x] += s_l[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *ta.loss_sum = s_l[0];
        return;
    }
    const size_t i = (size_t)b * 256 + threadIdx.x;
    if (i >= ta.nw_all + ta.nbag) return;
    if (i >= ta.nw_all) {               // bag bias: bb0 -= lr * sum_t delta_t  (python/SNN_RBM.py:289)
        const size_t c = i - ta.nw_all;
        float g = 0.f;
#pragma unroll 8
        for (int z = 0; z < ta.splitk; ++z) g += ta.slab[(size_t)z * ta.nslab + ta.off_bag + c * 64];
        ta.bucket[i] = g;
        if (UPDATE) ta.bb0[c] -= ta.lr * g;
        return;
    }
    const size_t src = (i < ta.nw12) ? i : ta.nw12 + (i - ta.nw12) * 64;
    float g = 0.f;
#pragma unroll 8
    for (int z = 0; z < ta.splitk; ++z) g += ta.slab[(size_t)z * ta.nslab + src];
    float w = ta.master[i];
    ta.bucket[i] = g;                 // data term only (what data parallelism all-reduces)
    if (UPDATE) {
        if (ta.reg_all || i >= ta.nw12) g += 2.0f * ta.lambda1 * w;     // L2 term (:173)
        w -= ta.lr * g;
        ta.master[i] = w;
        const size_t n1 = (size_t)ta.K1p * ta.H1p, n2 = (size_t)ta.H1p * ta.H2p;
        if (i < n1) {
            const int r = (int)(i / ta.H1p), c = (int)(i % ta.H1p);
            static_cast<T*>(ta.w1t)[ft_off<T>(c, r, ta.K1p)] = (T)w;
            static_cast<T*>(ta.w1)[ft_off<T>(r, c, ta.H1p)] = (T)w;
        } else if (i < n1 + n2) {
            const size_t j = i - n1;
            const int r = (int)(j / ta.H2p), c = (int)(j % ta.H2p);
            static_cast<T*>(ta.w2t)[ft_off<T>(c, r, ta.H1p)] = (T)w;
            static_cast<T*>(ta.w2)[ft_off<T>(r, c, ta.H2p)] = (T)w;
        }
    }
}

}  // namespace fnn
