// fnn_step_kernels.hip.h -- the three launches of a training step (gfx950 only).
//
// One pass of the reference's hot loop body (python/FNN_wnzh.py:296-306) is three kernels on one
// stream, each a union of two independent roles that run side by side on different workgroups:
//
//   k_step1 = { MLP strip kernel: gather, forward, loss, backward-data }
//   k_step2 = { weight-gradient products } U { sparse-row SGD, level 1 } U { sort runs of the NEXT batch's keys }
//   k_step3 = { slab reduce + dense SGD + shadow refresh } U { sparse-row SGD, level 2 } U { rank-merge the runs }
//
// The step is a few microseconds of math, so its cost is launches and dependent memory round
// trips; three fat launches and no cross-stream events are what that regime wants (measured:
// a 7-node hipGraph replay costs 46 us on this runtime, seven plain launches 18.5 us).
#pragma once
#include "fnn_kernels.hip.h"

namespace fnn {

// ------------------------------------------------------------------------------------------
// Grouping role: a field's 4096 (row, t) keys sorted in two independent phases -- 16 runs of 256
// keys, each bitonic-sorted inside ONE wave's registers (phase A), then a merge by rank in which
// every key finds its final place and its segment [s, e) with binary searches over the runs
// (phase B).  32-bit keys (row << 12 | t) when n_rows * 4096 fits, else 64-bit.  Measured against
// the single-kernel bitonic network it replaced: 34 us -> 2 x ~4 us of role time.
// ------------------------------------------------------------------------------------------
template <typename KT> struct KeyTraits;
template <> struct KeyTraits<unsigned> { static constexpr int SH = 12; };
template <> struct KeyTraits<unsigned long long> { static constexpr int SH = 32; };

struct SortArgs {
    const int32_t* ids; int B, F; int64_t n_rows; int4* rec; int* owner_cnt; int nblk; void* skeys;
    // bag mode only (null otherwise): which rows of this batch sit in MORE THAN ONE column.  The grouping is per column, and a
    // row's update is one read-modify-write per column segment -- two columns holding the same row (python/SNN_RBM.py:248-253
    // lists a line's active features in line order, so a feature's column depends on the line) would race.  Every segment head
    // claims its row with atomicMax(tag_first[row], stamp << 6 | column); a head that finds this batch's stamp already there
    // under another column marks tag_shared[row] = stamp, and the update launches (at least one kernel boundary later) add
    // into such rows with float atomics instead (scatw1_body / scatw2_body).  Stamps grow with every grouping: no reset pass.
    int* tag_first; int* tag_shared; int stamp;
};

constexpr int SORT_N = 4096;     // keys per field handled by the union-kernel path (B <= 4096)

template <typename KT> __host__ __device__ constexpr size_t sort_lds_bytes() { return (size_t)SORT_N * sizeof(KT); }

// Invalid entries (t >= B, id outside the table) carry the all-ones row, so that every key of a
// field is distinct (the rank merges below need a strict total order) and they sort to the end.
template <typename KT> __device__ __forceinline__ KT inv_row() { return (~(KT)0) >> KeyTraits<KT>::SH; }

// first index in the ascending run q[0 .. 1 << LOG) whose key is >= v (branch-free; q in LDS)
template <typename KT, int LOG>
__device__ __forceinline__ int lower_bound_pow2(const KT* q, const KT v) {
    int base = 0;
#pragma unroll
    for (int s = 1 << (LOG - 1); s >= 1; s >>= 1) base += (q[base + s - 1] < v) ? s : 0;
    return base + ((q[base] < v) ? 1 : 0);
}

// Phase A of the split sort: every wave bitonic-sorts a run of 256 keys in registers (4 per lane:
// strides below 4 inside the lane, the rest wave shuffles -- no LDS, no barriers) and stores it.
// One workgroup = 4 runs = a quarter of a field; so.nblk = 4 F.
template <typename KT>
__device__ __forceinline__ void sortA_body(const SortArgs& so, const int blk, unsigned char*)
{
    constexpr int SH = KeyTraits<KT>::SH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, F = so.F, B = so.B;
    const int f = blk >> 2, base = (blk & 3) * 1024 + wave * 256;
    KT key[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                             // initial order inside a run is free
        const int t = base + a * 64 + lane;
        KT row = inv_row<KT>();
        if (t < B) {
            const int64_t id = so.ids[(size_t)t * F + f];
            if (id >= 0 && id < so.n_rows) row = (KT)id;
        }
        key[a] = (row << SH) | (KT)t;
    }
    const int i0 = lane * 4;                                  // position of key[0] in the run
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 4) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int b = a ^ j;
                    if (b > a) {
                        const bool up = ((i0 + a) & k) == 0;
                        const KT x = key[a], y = key[b];
                        const KT mn = x < y ? x : y, mx = x < y ? y : x;
                        key[a] = up ? mn : mx; key[b] = up ? mx : mn;
                    }
                }
            } else {
                const bool keepmin = ((i0 & j) == 0) == ((i0 & k) == 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const KT other = __shfl_xor(key[a], j >> 2);
                    const KT mine = key[a];
                    const KT mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    key[a] = keepmin ? mn : mx;
                }
            }
        }
    }
    KT* out = static_cast<KT*>(so.skeys) + (size_t)f * SORT_N + base + i0;
#pragma unroll
    for (int a = 0; a < 4; ++a) out[a] = key[a];
}

// Phase B of the split sort: merge by RANK.  A key's place in the field's final order is its index
// in its own run plus, for each of the 15 other runs, the number of keys below it (a 9-step binary
// search in LDS); its segment [s, e) comes the same way: s = keys below (row, 0), e = keys below
// (row + 1, 0).  Every key is independent -- one thread per key, 16 workgroups per field
// (so.nblk = 16 F), one barrier -- instead of a 15 us chain of dependent merge stages.
template <typename KT>
__device__ __forceinline__ void sortB_body(const SortArgs& so, const int blk, unsigned char* smem)
{
    constexpr int SH = KeyTraits<KT>::SH;
    KT* s_key = reinterpret_cast<KT*>(smem);                 // [4096] the 16 sorted runs
    const int tid = threadIdx.x, f = blk >> 4, run = blk & 15;
    if (blk == 0 && tid == 0) *so.owner_cnt = 0;
    const KT* in = static_cast<const KT*>(so.skeys) + (size_t)f * SORT_N;
#pragma unroll
    for (int a = 0; a < 16; ++a) s_key[a * 256 + tid] = in[a * 256 + tid];
    __syncthreads();
    const KT key = s_key[run * 256 + tid];
    const KT row = key >> SH, lo_key = row << SH, hi_key = (row + 1) << SH;    // row + 1 wraps only for invalid entries
    // 48 binary searches (16 runs x {key, lo_key, hi_key}) advance in lockstep, so that every step
    // issues 48 independent LDS reads instead of one dependent read at a time.  The search of the
    // key in its own run returns its own index, so no run is special.
    int bp[16], bs[16], be[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bp[r] = bs[r] = be[r] = r * 256;
#pragma unroll
    for (int st = 128; st >= 1; st >>= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const KT vp = s_key[bp[r] + st - 1], vs = s_key[bs[r] + st - 1], ve = s_key[be[r] + st - 1];
            bp[r] += vp < key ? st : 0; bs[r] += vs < lo_key ? st : 0; be[r] += ve < hi_key ? st : 0;
        }
    }
    int pos = 0, s = 0, e = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        pos += bp[r] - r * 256 + (s_key[bp[r]] < key ? 1 : 0);
        s += bs[r] - r * 256 + (s_key[bs[r]] < lo_key ? 1 : 0);
        e += be[r] - r * 256 + (s_key[be[r]] < hi_key ? 1 : 0);
    }
    int4 rr = make_int4(-1, 0, 0, 0);
    if (row != inv_row<KT>()) {
        rr = make_int4((int)row, (int)(key & (((KT)1 << SH) - 1)), s, e);
        if (so.tag_first && pos == s) {                            // head of its segment: one claim per (row, column)
            const int mine = (so.stamp << 6) | f;
            const int old = atomicMax(&so.tag_first[(size_t)row], mine);
            if ((old >> 6) == so.stamp && old != mine) so.tag_shared[(size_t)row] = so.stamp;
        }
    }
    so.rec[(size_t)f * SORT_N + pos] = rr;
}

// The split sort as two plain launches, for a batch nobody announced (fnn_prefetch_ids) and for
// the inner-product family: 4 F then 16 F workgroups, ~7 us each instead of the 34 us single-kernel
// bitonic network.
template <typename KT>
static __global__ __launch_bounds__(256) void k_sortA(const SortArgs so)
{
    extern __shared__ __align__(16) unsigned char smem[];
    sortA_body<KT>(so, blockIdx.x, smem);
}
template <typename KT>
static __global__ __launch_bounds__(256) void k_sortB(const SortArgs so)
{
    extern __shared__ __align__(16) unsigned char smem[];
    sortB_body<KT>(so, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// step 1: MLP strips of this batch.  (At 256 VGPRs a strip workgroup fills its CU, so the sort
// roles ride on steps 2 and 3, whose workgroups are small enough to share a CU.)
// ------------------------------------------------------------------------------------------
template <typename T, int C1, int C2, int CX, bool BAG, int NW = 4>
static __global__ __launch_bounds__(64 * NW) void k_step1(const MlpArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    mlp_body<T, C1, C2, CX, BAG, NW>(a, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// step 2: run-sorts of the NEXT batch's keys  U  weight gradients  U  sparse-row SGD level 1
// ------------------------------------------------------------------------------------------
template <typename T, typename KT>
static __global__ __launch_bounds__(256) void k_step2(const SortArgs so, const WgradArgs wa, const int nwx,
                                               const int splitk, const ScatArgs sa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x, nw = nwx * splitk;
    if (b < so.nblk) sortA_body<KT>(so, b, smem);                              // so.nblk = 4 * F or 0
    else if (b < so.nblk + nw) {
        // XCD-aware order: hardware block ids that differ by a multiple of 8 share an XCD (and its L2); deal the role's blocks so
        // that each of the 8 classes owns a contiguous run of (tile, K slice) pairs -- one or two K slices of the operands per XCD
        // instead of a little of every slice
        int w = b - so.nblk;
        if (((b ^ w) & 7) == 0) {                              // the role starts at a multiple of 8
            const int cls = w & 7, k = w >> 3, qd = nw >> 3, rm = nw & 7;
            w = (cls < rm ? cls * (qd + 1) : rm * (qd + 1) + (cls - rm) * qd) + k;
        }
        wgrad_body<T>(wa, w % nwx, w / nwx);
    }
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
    if ((int)blockIdx.x < so.nblk) { sortB_body<KT>(so, blockIdx.x, smem); return; }          // rank merge: 16 F or 0 WGs
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
        v = strided_sum256(ta.loss_t, ta.Ba);
        s_l[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_l[threadIdx.x] += s_l[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *ta.loss_sum = s_l[0];
        return;
    }
    // Dense tensors W1p, W2p: 4 consecutive elements per thread (same row, 4 columns: 16-byte slab / master
    // accesses, one 8-byte piece of the [in][out] shadow); the short tail (w3p, bag bias) one element each.
    const size_t nvec = ta.nw12 / 4;
    const int nvb = (int)((nvec + 255) / 256);
    if (b < nvb) {
        const size_t q = (size_t)b * 256 + threadIdx.x;
        if (q >= nvec) return;
        const size_t i = q * 4;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = 0; z < ta.splitk; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(ta.slab + (size_t)z * ta.nslab + i);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
        *reinterpret_cast<float4*>(ta.bucket + i) = g;          // data term only (what data parallelism all-reduces)
        if (UPDATE) {
            float4 w = *reinterpret_cast<const float4*>(ta.master + i);
            const float l2 = ta.reg_all ? 2.0f * ta.lambda1 : 0.0f;     // L2 term (:173; only w3, b3 unless reg_all)
            w.x -= ta.lr * (g.x + l2 * w.x); w.y -= ta.lr * (g.y + l2 * w.y);
            w.z -= ta.lr * (g.z + l2 * w.z); w.w -= ta.lr * (g.w + l2 * w.w);
            *reinterpret_cast<float4*>(ta.master + i) = w;
            const size_t n1 = (size_t)ta.K1p * ta.H1p;
            const bool first = i < n1;
            const size_t j = first ? i : i - n1;
            const int ld = first ? ta.H1p : ta.H2p, kin = first ? ta.K1p : ta.H1p;
            const int r = (int)(j / ld), c = (int)(j % ld);
            T* wn = static_cast<T*>(first ? ta.w1 : ta.w2);     // [in][out]: k = column, 4 consecutive
            T* wt = static_cast<T*>(first ? ta.w1t : ta.w2t);   // [out][in]: k = row
            store4(wn + ft_off<T>(r, c, ld), w.x, w.y, w.z, w.w);
            wt[ft_off<T>(c, r, kin)] = (T)w.x; wt[ft_off<T>(c + 1, r, kin)] = (T)w.y;
            wt[ft_off<T>(c + 2, r, kin)] = (T)w.z; wt[ft_off<T>(c + 3, r, kin)] = (T)w.w;
        }
        return;
    }
    const size_t i = ta.nw12 + (size_t)(b - nvb) * 256 + threadIdx.x;
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
    const size_t src = ta.nw12 + (i - ta.nw12) * 64;            // w3p: column 0 of an [H2p][64] tile
    float g = 0.f;
#pragma unroll 8
    for (int z = 0; z < ta.splitk; ++z) g += ta.slab[(size_t)z * ta.nslab + src];
    float w = ta.master[i];
    ta.bucket[i] = g;
    if (UPDATE) {
        g += 2.0f * ta.lambda1 * w;                              // w3, b3 are always regularised (:173)
        ta.master[i] = w - ta.lr * g;
    }
}

// ------------------------------------------------------------------------------------------
// Data parallelism, FNN_DP_COLLECTIVE_P2P: the update launch that IS the all-reduce.  Every rank's exchange region
// ([2 parities][nbp floats] buckets, then 8 flags on 64-byte lines of their own) is mapped by every peer (hipIpc*).
//   1. block 0, lane p: release at system scope (this rank's bucket of the step -- written by the launch before this one -- is
//      visible to every device), then store the step number into flag[rank] of PEER p's region;
//   2. every block, lane p: poll flag[p] of its OWN region (acquire, system scope) until it holds this step's number -- peer p
//      has published; bounded by the clock (default 30 s, $FNN_P2P_TIMEOUT_MS: a first version counted ~2.5 s of polls, which a peer
//      process that was merely late at start-up could exceed -- met in a two-process rehearsal): a peer that never arrives raises
//      bit 4 of the error word and the block leaves without touching the weights;
//   3. sum the world's buckets in RANK order (every rank forms the same sum, bit for bit) and apply
//      theta <- theta - lr * (sum + L2 term) with the shadow refresh of k_update.
// Reuse of a parity buffer is safe without a second signal: a rank rewrites bucket[n & 1] in step n + 2, after its step n + 1
// launch saw every peer's flag n + 1, which a peer raises only after its own step-n launch (the reader of bucket[n & 1]) ended.
// Flags only grow (64-bit step numbers): nothing is ever reset.
// ------------------------------------------------------------------------------------------
struct P2PArgs {
    float* peer[8]; int world, rank; unsigned long long step; size_t bucket_off, flag_off; int* err;
    unsigned long long timeout_ticks;      // of the constant 100 MHz clock (s_memrealtime): how long a workgroup waits for a peer's flag
    unsigned long long* wait_max;          // the longest wait any step has seen so far (block 0's pollers), same clock: a diagnostic
};

template <typename T>
static __global__ __launch_bounds__(256) void k_p2p_update(const P2PArgs pa, float* __restrict__ master, const float lr, const float lambda1,
                                                    const int reg_all, const int K1p, const int H1p, const int H2p, T* __restrict__ w1,
                                                    T* __restrict__ w1t, T* __restrict__ w2, T* __restrict__ w2t, float* __restrict__ bb0,
                                                    const size_t nw12, const size_t nw, const size_t nbag)
{
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    if (blockIdx.x == 0 && tid < pa.world) {
        unsigned long long* f = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(pa.peer[tid]) + pa.flag_off) + (size_t)pa.rank * 8;
        __hip_atomic_store(f, pa.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (tid < pa.world) {                                       // one wave per workgroup polls (and performs the system-scope acquire)
        const unsigned long long* f = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(pa.peer[pa.rank]) + pa.flag_off) + (size_t)tid * 8;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < pa.step) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > pa.timeout_ticks) { s_bad = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (blockIdx.x == 0) atomicMax(pa.wait_max, __builtin_amdgcn_s_memrealtime() - t0);
    }
    __syncthreads();                                            // the other waves read the buckets behind this barrier
    if (s_bad) { if (tid == 0) atomicOr(pa.err, 16); return; }
    // W1p | W2p: 4 consecutive elements per thread (16-byte bucket reads from every rank, rank order), as in k_step3's dense role
    const size_t nvec = nw12 / 4;
    const int nvb = (int)((nvec + 255) / 256);
    const int b = blockIdx.x;
    if (b < nvb) {
        const size_t q = (size_t)b * 256 + tid;
        if (q >= nvec) return;
        const size_t i = q * 4;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < pa.world; ++r) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa.peer[r] + pa.bucket_off + i));
            g.x += v[0]; g.y += v[1]; g.z += v[2]; g.w += v[3];
        }
        float4 w = *reinterpret_cast<const float4*>(master + i);
        const float l2 = reg_all ? 2.0f * lambda1 : 0.0f;
        w.x -= lr * (g.x + l2 * w.x); w.y -= lr * (g.y + l2 * w.y);
        w.z -= lr * (g.z + l2 * w.z); w.w -= lr * (g.w + l2 * w.w);
        *reinterpret_cast<float4*>(master + i) = w;
        const size_t n1 = (size_t)K1p * H1p;
        const bool first = i < n1;
        const size_t j = first ? i : i - n1;
        const int ld = first ? H1p : H2p, kin = first ? K1p : H1p;
        const int r = (int)(j / ld), c = (int)(j % ld);
        T* wn = first ? w1 : w2;
        T* wt = first ? w1t : w2t;
        store4(wn + ft_off<T>(r, c, ld), w.x, w.y, w.z, w.w);
        wt[ft_off<T>(c, r, kin)] = (T)w.x; wt[ft_off<T>(c + 1, r, kin)] = (T)w.y;
        wt[ft_off<T>(c + 2, r, kin)] = (T)w.z; wt[ft_off<T>(c + 3, r, kin)] = (T)w.w;
        return;
    }
    const size_t i = nw12 + (size_t)(b - nvb) * 256 + tid;      // w3p, then the bag bias: one element each
    if (i >= nw + nbag) return;
    float g = 0.f;
    for (int r = 0; r < pa.world; ++r) g += __builtin_nontemporal_load(pa.peer[r] + pa.bucket_off + i);
    if (i >= nw) { bb0[i - nw] -= lr * g; return; }             // python/SNN_RBM.py:289
    const float w = master[i];
    g += 2.0f * lambda1 * w;                                     // w3, b3 are always regularised (python/FNN_wnzh.py:173)
    master[i] = w - lr * g;
}

}  // namespace fnn
