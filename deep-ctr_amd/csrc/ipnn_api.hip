// ipnn_api.hip -- the inner-product FNN family (FNN_IP_L3 / L5 / L7) on gfx950: kernels + C ABI
// (include/ipnn_hip.h).  Replaces the TensorFlow graph of python/FNN_IP_L7.py:102-133 (forward),
// :82-88 (loss) and its gradient step (plain SGD).  Built from the FNN path's pieces: the
// fragment-tiled MFMA GEMM with fused epilogues (k_gemm_ft: forward, backward-data and the split-K
// weight-gradient product of every layer, 64 x 64 per wave, operands double-buffered in registers) and the
// sorted, atomics-free sparse-row update (k_sortA / k_sortB / k_scat1 / k_scat2) -- plus two kernels of its
// own for the inner-product layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fnn_hip.h"
#include "../../include/ipnn_hip.h"
#include "fnn_step_kernels.hip.h"
#include "metrics.hip.h"

using namespace fnn;

namespace {

thread_local std::string g_ip_err;
inline int rup(int x, int m) { return (x + m - 1) / m * m; }

constexpr int A_TANH = IPNN_ACT_TANH, A_SIG = IPNN_ACT_SIGMOID, A_RELU = IPNN_ACT_RELU;

__device__ inline float ip_act(float z, int act) {
    if (act == A_RELU) return fmaxf(z, 0.f);
    if (act == A_TANH) return tanhf(z);
    return 1.0f / (1.0f + expf(-z));
}
template <int ACT> __device__ inline float ip_act_c(float z) {
    if (ACT == A_RELU) return fmaxf(z, 0.f);
    if (ACT == A_TANH) return tanhf(z);
    return 1.0f / (1.0f + expf(-z));
}
template <int ACT> __device__ inline float ip_dact_u_c(float u) {
    if (ACT == A_RELU) return u > 0.f ? 1.0f : 0.0f;
    if (ACT == A_TANH) return 1.0f - u * u;
    return u * (1.0f - u);
}
// derivative of act at l, written in terms of u = act(l)
__device__ inline float ip_dact_u(float u, int act) {
    if (act == A_RELU) return u > 0.f ? 1.0f : 0.0f;
    if (act == A_TANH) return 1.0f - u * u;
    return u * (1.0f - u);
}

// ------------------------------------------------------------------------------------------
// Inner-product layer, forward (python/FNN_IP_L7.py:104-114 + the first act/dropout of :115):
// 16 examples per workgroup.  a0' [Ba][D0p] in "slot" layout: column 16 f + l = e_f[l],
// columns 16F .. 16F+P-1 the P = F(F-1)/2 pair products (row-major i < j), column CB = b,
// column CB+1 = 1 (carries h1_b); every real column goes through act and the keep-mask / keep.
// a0 is written fragment-tiled twice: a0F (rows = examples, k = columns: the next forward product's
// A operand) and a0T (rows = columns, k = examples: the weight-gradient product's A operand).
// ------------------------------------------------------------------------------------------
struct IpFwdArgs {
    int P;                      // pair products: F (F - 1) / 2 (FNN_IP_L*) or 0 (the plain FNN class, python/FNN.py:80)
    const int32_t* ids; int B, F, K; const float* table16; int64_t n_rows; const float* b;
    const uint8_t* mask; int d0; float inv_keep; int act; int D0p, ldT; int* err;
    int skip;                   // diagnostics (IPNN_FWD_SKIP bits: 1 gather, 2 embedding store, 4 compute, 8 output stores); 0 in production
    bool wt;                    // outputs written through (IPNN_WT=0: plain stores; see store4_wt in fnn_kernels.hip.h)
};

// gather of 16 examples' rows into an LDS tile [16][F*16]: all ids of a batch of 4 elements per thread first, then all
// rows (two dependent round trips per batch instead of two per element)
// STR: floats per field in the tile -- 16 (vector stores), or 17 for the forward's pair products: lanes that read slot l of
// DIFFERENT fields then fall on different banks (with 16 all of them share two banks: a 32-way conflict per read)
template <int STR, int NT = 256>
__device__ __forceinline__ void ip_gather16(float* se, const int32_t* __restrict__ ids, const float* __restrict__ table16,
                                            const int64_t n_rows, const int t0, const int B, const int F, int* err)
{
    const int n = 16 * F * 4, FS = F * STR;
    for (int e0 = threadIdx.x; e0 < n; e0 += NT * 4) {
        int64_t id[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + NT * j, f = (e >> 2) % F, t = t0 + (e >> 2) / F;
            id[j] = (e < n && t < B) ? (int64_t)ids[(size_t)t * F + f] : -1;
        }
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + NT * j;
            if (e < n && t0 + (e >> 2) / F < B && (id[j] < 0 || id[j] >= n_rows)) { if (err) atomicOr(err, 1); id[j] = -1; }
            v[j] = id[j] >= 0 ? *reinterpret_cast<const float4*>(table16 + (size_t)id[j] * SLOT + 4 * (e & 3)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + NT * j;
            if (e < n) {
                float* d = se + ((e >> 2) / F) * FS + ((e >> 2) % F) * STR + 4 * (e & 3);
                if (STR == SLOT) *reinterpret_cast<float4*>(d) = v[j];
                else { d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w; }
            }
        }
    }
}

constexpr int IPF_NT = 512;     // 8 waves: the kernel is bound by instruction issue, two waves per SIMD overlap it (IPNN_IPF_NT=1024: four)
template <typename T, int NT = IPF_NT>
static __global__ __launch_bounds__(NT) void k_ip_fwd(const IpFwdArgs a, T* __restrict__ a0, T* __restrict__ a0T, float* __restrict__ emb)
{
    typedef typename Traits<T>::frag frag;
    constexpr int EPL = Traits<T>::EPL;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int SP = SLOT + 1;                                // padded field stride of the embedding tile
    float* se = reinterpret_cast<float*>(smem);                 // [16][F*17] raw embeddings
    float* sa = se + 16 * a.F * SP;                             // [16][D0p]  a0 values
    const int tid = threadIdx.x, t0 = blockIdx.x * 16, F = a.F, K = a.K, B = a.B, FS = F * SLOT, FSP = F * SP;
    const int P = a.P, CB = FS + P;
    if (!(a.skip & 1)) ip_gather16<SP, NT>(se, a.ids, a.table16, a.n_rows, t0, B, F, a.err);
    __syncthreads();
    if (emb && !(a.skip & 2))                                                    // kept for the backward of the inner products (no second gather)
        for (int e = tid; e < 16 * FS / 4; e += NT) {
            const int r = (4 * e) / FS, c = (4 * e) % FS;
            const float* q = se + r * FSP + (c >> 4) * SP + (c & 15);
            store16_sel(a.wt, emb + (size_t)t0 * FS + 4 * e, make_float4(q[0], q[1], q[2], q[3]));
        }
    const float bval = *a.b;
    // a thread owns columns c = (tid & 63) + 64 k and rows r = (tid >> 6) + NWV i; 8 columns at a time: their pair indices,
    // then all 8 RPT keep-mask bytes (one round trip), then the values
    constexpr int NWV = NT / 64, RPT = 16 / NWV;
    for (int c0 = tid & 63; c0 < ((a.skip & 4) ? 0 : a.D0p); c0 += 512) {
        int ref[8], pi[8], pj[8];                               // ref: column in the reference's z1 order
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + 64 * k;
            ref[k] = -1; pi[k] = 0; pj[k] = 0;
            if (c < FS) { const int f = c / SLOT, l = c % SLOT; if (l < K) ref[k] = f * K + l; }
            else if (c < CB) {
                int n = c - FS, i = 0;                          // n-th pair (i, j), i < j, row-major
                while (n >= F - 1 - i) { n -= F - 1 - i; ++i; }
                pi[k] = i; pj[k] = i + 1 + n; ref[k] = F * K + (c - FS);
            } else if (c == CB) ref[k] = a.d0 - 1;
        }
        float mk[8][RPT];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int t = t0 + (tid >> 6) + NWV * i;
                mk[k][i] = (a.mask && ref[k] >= 0 && t < B) ? (float)a.mask[(size_t)t * a.d0 + ref[k]] * a.inv_keep : 1.0f;
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + 64 * k;
            if (c >= a.D0p) break;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = (tid >> 6) + NWV * i, t = t0 + r;
                float z = 0.f;
                if (c < FS) z = se[r * FSP + (c >> 4) * SP + (c & 15)];
                else if (c < CB) {
                    float s2 = 0.f;
                    for (int l = 0; l < K; ++l) s2 = fmaf(se[r * FSP + pi[k] * SP + l], se[r * FSP + pj[k] * SP + l], s2);
                    z = s2;
                } else if (c == CB) z = bval;
                float v = 0.f;
                if (t < B) {
                    if (ref[k] >= 0) v = ip_act(z, a.act) * mk[k][i];
                    else if (c == CB + 1) v = 1.0f;
                }
                sa[r * a.D0p + c] = v;
            }
        }
    }
    __syncthreads();
    // F layout: this workgroup's 16 rows are one row tile; a lane-slot of a fragment = EPL consecutive columns of one row
    if (a.skip & 8) return;
    for (int e = tid; e < 16 * a.D0p / EPL; e += NT) {
        const int g = e >> 4, r = e & 15;                        // column group, row
        frag fv;
#pragma unroll
        for (int x = 0; x < EPL; ++x) fv[x] = (T)sa[r * a.D0p + g * EPL + x];
        store16_sel(a.wt, a0 + ft_off<T>(t0 + r, g * EPL, a.D0p), fv);
    }
    for (int e = tid; e < a.D0p * 4; e += NT) {
        const int c = e >> 2, tq = e & 3;
        store4_sel(a.wt, a0T + ft_off<T>(c, t0 + 4 * tq, a.ldT), sa[(4 * tq) * a.D0p + c], sa[(4 * tq + 1) * a.D0p + c],
                  sa[(4 * tq + 2) * a.D0p + c], sa[(4 * tq + 3) * a.D0p + c]);
    }
}

// Inner-product layer, backward: dz1' [Ba][D0p] f32 (already times mask/keep and act') ->
// slot-layout embedding gradients gx' [Ba][D0p] (columns 16f + l) for the sparse-row update, and the
// per-workgroup partial of db = sum_t dz1[b].
struct IpBwdArgs { int P; const int32_t* ids; int B, F, K; const float* table16; int64_t n_rows; int D0p; const float* emb; };

static __global__ __launch_bounds__(256) void k_ip_bwd(const IpBwdArgs a, const float* __restrict__ dz, float* __restrict__ gxp,
                                                        float* __restrict__ gb_part)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float* se = reinterpret_cast<float*>(smem);                 // [16][F*16]
    float* sd = se + 16 * a.F * SLOT;                           // [16][D0p]
    const int tid = threadIdx.x, t0 = blockIdx.x * 16, F = a.F, K = a.K, B = a.B, FS = F * SLOT;
    const int P = a.P, CB = FS + P;
    // both tiles are contiguous in memory (16 consecutive rows): 16-byte pieces, eight loads in flight per thread before the
    // first LDS store (a copy loop of one load -> one store per trip pays one memory round trip per trip)
    auto fill = [&](float* dst, const float* __restrict__ src, const int n4) {
        for (int e0 = tid; e0 < n4; e0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int e = e0 + 256 * k; v[k] = e < n4 ? *reinterpret_cast<const float4*>(src + 4 * (size_t)e) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int e = e0 + 256 * k; if (e < n4) *reinterpret_cast<float4*>(dst + 4 * e) = v[k]; }
        }
    };
    if (a.emb) fill(se, a.emb + (size_t)t0 * FS, 16 * FS / 4);       // the raw embeddings the forward gathered
    else ip_gather16<SLOT>(se, a.ids, a.table16, a.n_rows, t0, B, F, nullptr);
    fill(sd, dz + (size_t)t0 * a.D0p, 16 * a.D0p / 4);
    __syncthreads();
    for (int c = tid; c < FS; c += 256) {                        // a thread owns (field f, slot l) for all 16 examples:
        const int f = c / SLOT, l = c % SLOT;                    // the pair index is computed once per partner field
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = l < K ? sd[r * a.D0p + c] : 0.f;
        if (l < K && P) {
            for (int j = 0; j < F; ++j) {
                // pair (i, j), i < j, sits at FS + i*(2F - i - 1)/2 + (j - i - 1); j == f contributes nothing
                const int i0 = f < j ? f : j, j0 = f < j ? j : f;
                const int n = j == f ? 0 : i0 * (2 * F - i0 - 1) / 2 + (j0 - i0 - 1);
                const float on = j == f ? 0.f : 1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = fmaf(sd[r * a.D0p + FS + n] * on, se[r * FS + j * SLOT + l], g[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gxp[(size_t)(t0 + r) * a.D0p + c] = g[r];
    }
    if (tid == 0) { float s = 0.f; for (int r = 0; r < 16; ++r) s += sd[r * a.D0p + CB]; gb_part[blockIdx.x] = s; }
}

// ------------------------------------------------------------------------------------------
// GEMM epilogues of the deep stack.
// ------------------------------------------------------------------------------------------
// All activation / delta matrices of the stack are FRAGMENT-TILED (ft_off) in both orientations:
// xF (rows = examples, k = units) feeds the next forward / backward-data product, xT (rows = units,
// k = examples) the weight-gradient product.  Only dz1 (f32, for the inner-product backward) is row-major.
// The epilogues compute the 4 values a lane owns (rows r0 .. r0+3 of one column); k_gemm_ft writes both
// layouts.  Keep-masks are read TRANSPOSED ([unit][example], k_mask_T below): a lane's 4 rows are 4
// consecutive bytes, one dword load; activations for act' come from xT the same way (one 8-byte load).
// Transposed keep-masks are tiled like a fragment-tiled operand of bytes: 16 units x 32 examples = one 512-byte
// block, so that a 32-example strip reads whole blocks (and a 128-row GEMM tile four of them per 16 units).
__host__ __device__ inline size_t mask_off(int col, int row, int ldT) {
    return ((size_t)(col >> 4) * (ldT >> 5) + (row >> 5)) * 512 + (col & 15) * 32 + (row & 31);
}
// Every epilogue is split in two: load() fetches what the lane's 4 values need from memory (it can be issued
// before the product's k-loop), apply() turns the accumulator into the values; pre() is both.
template <typename T> struct EpiIpFwd {      // a_t = mask/keep * act(l_t); ones column at d
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; const uint8_t* maskT; float inv_keep; int act, d, B;
    struct Aux { unsigned mb; };
    __device__ Aux load(int r0, int col) const {
        Aux x{0x01010101u};
        if (maskT && col < d) x.mb = *reinterpret_cast<const unsigned*>(maskT + mask_off(col, r0, ldT));
        return x;
    }
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const { apply(load(r0, col), r0, col, acc, v); }
    // the activation is a wave-uniform run-time value: branch on it once per fragment (or once per block, applyA)
    __device__ void apply(const Aux& ax, int r0, int col, const f32x4& acc, float v[4]) const {
        if (act == A_RELU) applyA<A_RELU>(ax, r0, col, acc, v);
        else if (act == A_TANH) applyA<A_TANH>(ax, r0, col, acc, v);
        else applyA<A_SIG>(ax, r0, col, acc, v);
    }
    // PLAIN: every row of the block is an example (< B) and every column a real unit (< d): no edge selects
    template <int ACT, bool PLAIN = false> __device__ void applyA(const Aux& ax, int r0, int col, const f32x4& acc, float v[4]) const {
        const unsigned mb = ax.mb;
        const float sc = maskT ? inv_keep : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = ip_act_c<ACT>(acc[r]) * ((float)((mb >> (8 * r)) & 0xffu) * sc);
            if (!PLAIN) {
                x = col < d ? x : (col == d ? 1.0f : 0.f);
                x = (r0 + r < B) ? x : 0.f;
            }
            v[r] = x;
        }
    }
    __device__ bool plain(int row_end, int col_end) const { return row_end <= B && col_end <= d; }
};
template <typename T> struct EpiIpOut {      // logits (column 0), loss, delta = sigmoid(logit) - y
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; const float* y; float* logits; float* loss_t; float* p_out; int B;
    float gscale;            // 1 for a summed loss, 1 / B for tf.reduce_mean (python/FNN_IP_L7.py:83-86): scales every gradient of the step
    struct Aux { };
    __device__ Aux load(int, int) const { return Aux{}; }
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const { apply(Aux{}, r0, col, acc, v); }
    __device__ void apply(const Aux&, int r0, int col, const f32x4& acc, float v[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = r0 + r;
            float x = 0.f;
            if (col == 0 && t < B) {
                const float z = acc[r], p = 1.0f / (1.0f + expf(-z));
                if (logits) logits[t] = z;
                if (p_out) p_out[t] = p;
                if (y) { x = (p - y[t]) * gscale; loss_t[t] = fmaxf(z, 0.f) - z * y[t] + log1pf(expf(-fabsf(z))); }
            } else if (col == 0 && loss_t) loss_t[t] = 0.f;
            v[r] = x;
        }
    }
};
template <typename T> struct EpiIpBwd {      // delta l_t = (delta l_{t+1} W^T) * mask/keep * act'(l_t)
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; float* out32; int ld32; const T* aT; const uint8_t* maskT; float inv_keep, keep; int act, d, B;
    const int* ref;          // layer 0 (out32 != null): slot column -> reference column, -1 = padding
    struct Aux { unsigned mb; float4 u; bool real; };
    __device__ Aux load(int r0, int col) const {
        Aux x{0x01010101u, make_float4(0.f, 0.f, 0.f, 0.f), ref ? ref[col] >= 0 : col < d};
        if (x.real) {
            if (maskT) x.mb = *reinterpret_cast<const unsigned*>(maskT + mask_off(col, r0, ldT));
            x.u = load4(aT + ft_off<T>(col, r0, ldT));
        }
        return x;
    }
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const { apply(load(r0, col), r0, col, acc, v); }
    __device__ void apply(const Aux& ax, int r0, int col, const f32x4& acc, float v[4]) const {
        if (act == A_RELU) applyA<A_RELU>(ax, r0, col, acc, v);
        else if (act == A_TANH) applyA<A_TANH>(ax, r0, col, acc, v);
        else applyA<A_SIG>(ax, r0, col, acc, v);
    }
    template <int ACT, bool PLAIN = false> __device__ void applyA(const Aux& ax, int r0, int col, const f32x4& acc, float v[4]) const {
        const unsigned mb = ax.mb;
        const float uu[4] = {ax.u.x, ax.u.y, ax.u.z, ax.u.w};
        const float sc = maskT ? inv_keep : 1.0f, ks = maskT ? keep : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = (float)((mb >> (8 * r)) & 0xffu);
            float x = acc[r] * m * sc * ip_dact_u_c<ACT>(uu[r] * ks);                      // act(l_t) where m = 1
            if (!PLAIN) x = (r0 + r < B && ax.real) ? x : 0.f;
            v[r] = x;
            if (out32) out32[(size_t)(r0 + r) * ld32 + col] = x;
        }
    }
    __device__ bool plain(int row_end, int col_end) const { return !ref && !out32 && row_end <= B && col_end <= d; }
};

// ------------------------------------------------------------------------------------------
// Strip kernels: the whole deep stack for a strip of 16 RT examples inside ONE workgroup, forward
// (k_ip_strip_fwd: L + 1 products, activations, loss / output delta) and backward-data
// (k_ip_strip_bwd: L + 1 products down to dz1).  As separate GEMM launches these products are bound by
// fixed costs -- per launch ~4 us of launch, ~1.5 us of pipeline fill, ~5 us of epilogue writing both
// operand layouts and ~5 us of cold operands, against 3-8 us of L2-bound main loop -- 16 times per step.
// Here a strip's activations stay in LDS (two ping-pong tiles in fragment order, so an A fragment is
// one conflict-free ds_read_b128 per lane), each wave owns 64-column blocks of a layer's output and
// streams the weights (fragment-tiled, L2-resident) straight into B fragments, and only what the
// weight-gradient products need goes to HBM: the transposed activations / deltas.  Every workgroup
// reads every weight once per pass: 16 RT FLOP per L2 byte, which is what bounds these kernels.
// ------------------------------------------------------------------------------------------
constexpr int STRIP_MAXP = IPNN_MAX_HIDDEN + 1;                  // products of the stack
// TWO workgroups per strip (h = blockIdx.x & 1; strip = blockIdx.x >> 1): a 32-example strip occupies one CU and at batch 4096
// that is 128 of the chip's 256, each streaming every weight at the ~75 GB/s one CU draws from L2 -- the bound of these kernels.
// The pair splits every WIDE product by output column blocks (block j belongs to workgroup j & 1), so each CU streams half the
// weights, and swaps halves after it: a workgroup pushes its blocks of the new activation tile to `xch` with write-through
// (sc1) 16-byte stores, drains them, raises its flag, polls its partner's and pulls the partner's blocks into its own LDS tile
// with sc1 loads (MI355X_MICROARCH.md, "Valid forms": every store and load of the handed-off bytes sc1, stores drained before
// the flag, one lane polls, the others load behind a workgroup barrier).  Narrow products (fewer than DUO_MIN_BLOCKS column
// blocks) are computed by BOTH workgroups -- their weights are a few hundred KB, a swap costs more -- and workgroup 0 alone
// writes their outputs.  Workgroups b and b + 8 share an XCD, so the even XCDs only ever stream the even blocks' weights and
// the odd XCDs the odd ones: half the weight footprint per L2.  Flags hold launch_epoch * 16 + (swap index + 1): they only
// grow, so nothing is reset between launches; a poll gives up after DUO_SPIN_LIMIT tries and raises `err` (no hang on a bug).
struct StripDuo { int on; unsigned long long* xch; int* flags; int epoch; int* err; size_t xch_wg; int min_blocks; };
constexpr int DUO_MIN_BLOCKS = 5;                                // default of StripDuo::min_blocks (IPNN_DUO_MIN)
constexpr int DUO_SPIN_LIMIT = 1 << 22;
template <typename T> struct StripFwdArgs {
    const T* a0; int n;                                          // a0: F layout [Ba][Dp0]; n = L + 1
    const T* W[STRIP_MAXP]; int Dp[STRIP_MAXP + 1];              // W[t-1] = wf[t-1]: fragment-tiled [Dp_t][Dp_{t-1}]
    EpiIpFwd<T> ef[STRIP_MAXP]; EpiIpOut<T> eo;                  // ef[t-1], t = 1..L; eo: the output unit
    long long* dbg;                                              // IPNN_STAMPS=1 (diagnostics): s_memtime per layer, 16 per workgroup
    int rot;                                                     // rotate the block order per workgroup (IPNN_STRIP_ROT=0: off)
    StripDuo duo;                                                // two workgroups per strip (see StripDuo); duo.on = 0: one
    int sel;                                                     // IPNN_STAMPS: the product whose first block of wave 0 is stamped in detail (slots 10..14)
    // A launch may cover a RANGE of the stack (round 3: the wide products as pairs of 32-example strips, the narrow tail as 16-example
    // strips on every CU).  has_out = 0: the last product of this launch is a hidden layer whose tile leaves for HBM in the F layout
    // (finalF: the next launch's a0), block by block as it is computed -- no swap, no barrier behind it.
    int has_out; T* finalF;
    // 16-example strips (RT = 1) only: products whose weights are copied into LDS behind the two tiles at the start of the launch
    // (element offset there, or -1: streamed from L2 as usual).  The narrow products are chains of a few k-steps, each an L2 round
    // trip long (~600 ticks with four in flight); from LDS a k-step costs what its MFMAs cost
    int wlds[STRIP_MAXP];
    int warm;                                                    // touch every line of these arguments at the start (see strip_warm_args)
};
template <typename T> struct StripBwdArgs {
    const T* dlast; int n;                                       // delta of the output layer, F layout [Ba][64]
    const T* W[STRIP_MAXP]; int Dp[STRIP_MAXP + 1];              // W[t-1] = wb[t-1]: fragment-tiled [Dp_{t-1}][Dp_t]
    EpiIpBwd<T> eb[STRIP_MAXP];                                  // eb[t-1]: product t -> delta l_{t-1}
    long long* dbg;
    int rot;
    StripDuo duo;
    // bottom = 1: product 1 of this launch is the stack's first (its output is dz1, f32, no tile).  bottom = 0: the launch stops inside
    // the stack and product 1's tile leaves for HBM in the F layout (finalF: the next launch's dlast)
    int bottom; T* finalF;
    int wlds[STRIP_MAXP];                                        // as in StripFwdArgs (index = product index t - 1 of this launch)
    int warm;
};
// The kernels read their per-product arguments (ef[l] / eb[t - 1]: a different 64-byte line of the argument segment per product) with
// scalar loads at the top of each product, behind the barrier: a scalar-cache miss on every wave's critical path, once per product.
// One dword of every line at the start of the launch instead -- all misses in flight together, under the strip's own load.
template <int BYTES> __device__ __forceinline__ void strip_warm_args()
{
    typedef const int __attribute__((address_space(4))) karg_int;
    karg_int* p = (karg_int*)__builtin_amdgcn_kernarg_segment_ptr();
    int s = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) s |= p[o / 4];
    asm volatile("" :: "s"(s));
}

// one 64-column block of one product: acc[m][n] = sum_k in[16 m ..][k] W[64 blk + 16 n ..][k].
// Weights: five register stages (four k-steps in flight while one multiplies: with two waves per SIMD that
// covers an L2 round trip at the rate the texture path delivers fragments); loads past the end are clamped
// to the last k-step, so the loop has no branch.  The strip's own fragments come from LDS.
// An ITEM of a product is NF of its 16-column fragments: NF = 4 (a 64-column block) everywhere but in the 16-example strips of
// the narrow tail, whose kernels may take finer items (k_ip_strip_*<T, 1, NF>): a 64- or 128-wide product then occupies 4 / 8
// waves instead of 1 / 2, and an item's epilogue is NF / 4 of a block's.  `blk` below is the item index: fragments blk NF + n.
template <typename T, int NF = 4> struct StripB { typename Traits<T>::frag s[4][NF]; };     // k-steps 0..3 of an item's weights, in flight

template <typename T, int NF = 4>
__device__ __forceinline__ void strip_prefetch(StripB<T, NF>& pb, const T* __restrict__ W, const int nkt, const int blk, const int lane)
{
    typedef typename Traits<T>::frag frag;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NF; ++n) pb.s[j][n] = *reinterpret_cast<const frag*>(ft_frag<T>(W, blk * NF + n, min(j, nkt - 1), nkt, lane));
}

template <typename T, int RT, int NF = 4>
__device__ __forceinline__ void strip_product(f32x4 (&acc)[RT][NF], StripB<T, NF>& pb, const T* in, const T* __restrict__ W, const int nkt,
                                              const int blk, const int lane)
{
    typedef typename Traits<T>::frag frag;
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag b4[NF];
    frag (&b0)[NF] = pb.s[0]; frag (&b1)[NF] = pb.s[1]; frag (&b2)[NF] = pb.s[2]; frag (&b3)[NF] = pb.s[3];
    const int last = nkt - 1;
    auto loadb = [&](frag* b, const int kt) {
        const int k = min(kt, last);
#pragma unroll
        for (int n = 0; n < NF; ++n) b[n] = *reinterpret_cast<const frag*>(ft_frag<T>(W, blk * NF + n, k, nkt, lane));
    };
    auto mul = [&](const int kt, const frag* b) {
        frag a[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) a[m] = *reinterpret_cast<const frag*>(ft_frag<T>(in, m, kt, nkt, lane));
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) mma(acc[m][n], a[m], b[n]);
    };
    int kt = 0;
    for (; kt + 5 <= nkt; kt += 5) {
        loadb(b4, kt + 4); mul(kt, b0);
        if (kt + 5 < nkt) loadb(b0, kt + 5);
        mul(kt + 1, b1);
        if (kt + 6 < nkt) loadb(b1, kt + 6);
        mul(kt + 2, b2);
        if (kt + 7 < nkt) loadb(b2, kt + 7);
        mul(kt + 3, b3);
        if (kt + 8 < nkt) loadb(b3, kt + 8);
        mul(kt + 4, b4);
    }
    if (kt < nkt) mul(kt, b0);
    if (kt + 1 < nkt) mul(kt + 1, b1);
    if (kt + 2 < nkt) mul(kt + 2, b2);
    if (kt + 3 < nkt) mul(kt + 3, b3);
}

// what the epilogue of a block reads from memory (keep-mask bytes, activations for act'), fetched before the k-loop
template <typename T, int RT, int NF, typename Epi>
__device__ __forceinline__ void strip_aux(typename Epi::Aux (&ax)[RT][NF], const Epi& epi, const int row0, const int blk, const int lane)
{
    const int rq = 4 * (lane >> 4), cl = lane & 15;
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) ax[m][n] = epi.load(row0 + m * 16 + rq, (blk * NF + n) * 16 + cl);
}

// epilogue of a block: the lane's values through `epi.apply`, the transposed layout to HBM (8-byte pieces),
// the strip's own layout to the LDS tile `out` (the next product's A operand) when there is a next product
template <int ACT, bool PLAIN, typename T, int RT, int NF, typename Epi>
__device__ __forceinline__ void strip_epilogue_a(f32x4 (&acc)[RT][NF], const typename Epi::Aux (&ax)[RT][NF], const Epi& epi, T* out, const int N,
                                                 const int row0, const int blk, const int lane)
{
    const int rq = 4 * (lane >> 4), cl = lane & 15, col0 = blk * NF * 16;
    T* const outT = epi.outT ? epi.outT + ft_off<T>(col0 + cl, row0 + rq, epi.ldT) : nullptr;   // + n * 16 units, + m * 16 examples
    // (the item's first column is a multiple of 16 NF: with NF < 4 offsets of n * 16 columns still add without a carry into the k-step)
    T* const outL = out ? out + ft_off<T>(rq, col0 + cl, N) : nullptr;
    const size_t tn = ft_off<T>(16, 0, epi.ldT), tm = ft_off<T>(0, 16, epi.ldT);    // strides: 16 units, 16 examples (ldT % KS == 0)
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            float v[4];
            const int col = col0 + n * 16 + cl;
            epi.template applyA<ACT, PLAIN>(ax[m][n], row0 + m * 16 + rq, col, acc[m][n], v);
            if (outT) store4(outT + n * tn + m * tm, v[0], v[1], v[2], v[3]);
            if (outL) {
                T* o = outL + ft_off<T>(m * 16, n * 16, N);       // (row tile m, column offset 16 n): additive in this layout
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r * Traits<T>::EPL] = (T)v[r];
            }
        }
}
template <typename T, int RT, int NF, typename Epi>
__device__ __forceinline__ void strip_epilogue(f32x4 (&acc)[RT][NF], const typename Epi::Aux (&ax)[RT][NF], const Epi& epi, T* out, const int N,
                                               const int row0, const int blk, const int lane)
{   // one branch on the activation and on "interior block" per block
    const bool pl = epi.plain(row0 + RT * 16, (blk + 1) * NF * 16);
#define STRIP_EPI(ACT) do { if (pl) strip_epilogue_a<ACT, true, T, RT, NF, Epi>(acc, ax, epi, out, N, row0, blk, lane); \
                            else strip_epilogue_a<ACT, false, T, RT, NF, Epi>(acc, ax, epi, out, N, row0, blk, lane); } while (0)
    if (epi.act == A_RELU) STRIP_EPI(A_RELU);
    else if (epi.act == A_TANH) STRIP_EPI(A_TANH);
    else STRIP_EPI(A_SIG);
#undef STRIP_EPI
}

template <typename T> __device__ __forceinline__ void strip_load(T* dst, const T* __restrict__ src, const int nfrag16)
{   // nfrag16 16-byte pieces, contiguous in both
    typedef typename Traits<T>::frag frag;
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < nfrag16; i0 += nt * 4) {     // four loads in flight per thread before the first LDS store
        frag v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + nt * k; if (i < nfrag16) v[k] = reinterpret_cast<const frag*>(src)[i]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + nt * k; if (i < nfrag16) reinterpret_cast<frag*>(dst)[i] = v[k]; }
    }
}

#define STRIP_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
constexpr int STRIP_NW = 8;                                      // waves per strip workgroup (2 per SIMD: 256 registers each)
// A wave's blocks of the stack in order: (product p, list index w), (p, w + NW), ... then the next product it has a
// block in.  The weights of the NEXT block are requested before the current block's epilogue (they do not
// depend on the strip), so the barrier between two products and the epilogue hide their L2 round trip.
// A workgroup's LIST of a product's column blocks: all nblk of them, or -- a wide product of a pair (StripDuo) -- the
// (nblk - h + 1) / 2 blocks j with j & 1 == h.
struct StripItem { int p, blk; };
__device__ __forceinline__ bool duo_split(const StripDuo& d, const int nblk) { return d.on && nblk >= d.min_blocks; }
__device__ __forceinline__ int duo_count(const bool split, const int nblk, const int h) { return split ? (nblk - h + 1) >> 1 : nblk; }
// Workgroups walk their list in rotated order (list entry i of a workgroup with rotation g is entry (i + g) mod cnt), so
// that the workgroups of an XCD do not all stream the same weights -- the same L2 channels -- at the same moment.
// (per items per 64-column block: entry r of the list is sub-item r % per of the workgroup's own block r / per)
__device__ __forceinline__ int strip_phys(const bool split, const int i, const int cnt, const int h, const int rot, const int per = 1) {
    const int r = (i + rot) % cnt, ob = r / per;
    return (split ? 2 * ob + h : ob) * per + r % per;
}
template <typename T> __device__ __forceinline__ bool strip_has_out(const StripFwdArgs<T>& a) { return a.has_out != 0; }
template <typename T> __device__ __forceinline__ bool strip_has_out(const StripBwdArgs<T>&) { return false; }
// items of a product in a workgroup's list: `per` items per 64-column block (4 / NF; pairs split whole blocks: per = 1 there);
// the output unit is ONE item whatever its padded width (column 0 is the only unit)
template <typename A> __device__ __forceinline__ int strip_items(const A& a, const int p, const bool fwd, const int h, const int per)
{
    const int nblk = fwd ? a.Dp[p + 1] / 64 : a.Dp[a.n - p - 1] / 64;
    if (fwd && strip_has_out(a) && p == a.n - 1) return 1;
    return duo_count(duo_split(a.duo, nblk), nblk, h) * per;
}
template <typename A> __device__ __forceinline__ StripItem strip_next(const A& a, StripItem it, const int wave, const bool fwd, const int h, const int per,
                                                                     const int nw = STRIP_NW)
{   // fwd: product p has Dp[p + 1] / 64 blocks (the output unit, p = n - 1: one); bwd: product index q = n - t, Dp[t - 1] / 64 blocks
    it.blk += nw;
    for (;;) {
        if (it.p >= a.n) return it;
        if (it.blk < strip_items(a, it.p, fwd, h, per)) return it;
        it.p += 1; it.blk = wave;
    }
}

// ---- the swap of a pair (StripDuo).  xch of workgroup (strip, h): [2 parities][own block r = j >> 1][RT][256] 8-byte words,
// a block's RT x 2 KB in the order they have in the LDS tile (row tile m: k-steps 2 j and 2 j + 1 are adjacent there).
typedef unsigned int duo_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte write-through stores / L1-bypassing loads of the swap buffer (raw buffer instructions with the sc1 cache policy: the
// compiler keeps their wait counts, unlike inline assembly; 8-byte accesses run at 0.54-0.70 of the 16-byte rate)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t duo_rsrc(const StripDuo& d) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)d.xch, 0, (int)(d.xch_wg * 8 * (size_t)gridDim.x), 0x00020000);
}
template <typename T, int RT, int NF = 4>
__device__ __forceinline__ void duo_push(const StripDuo& d, const T* out, const int N, const int item, const int parity, const int lane)
{   // by the wave that has just written item `item` (NF fragments of block j = item NF / 4) of the tile `out` (LDS operations of one
    // wave complete in order); a block's RT x 2 KB keep the order they have in the LDS tile, so an item's 16-byte pieces are a run of it
    constexpr int PER = 4 / NF, PIECES = 16 * NF * (int)sizeof(T);       // 16-byte pieces per row tile: 128 (a block), 64, 32
    const int j = item / PER, sub = item % PER;
    const __amdgpu_buffer_rsrc_t rs = duo_rsrc(d);
    const size_t dst = ((size_t)blockIdx.x * d.xch_wg + (size_t)parity * (d.xch_wg >> 1) + (size_t)(j >> 1) * (RT * 256)) * 8 + (size_t)sub * PIECES * 16;     // bytes
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const duo_u32x4* src = reinterpret_cast<const duo_u32x4*>(out + ft_off<T>(m * 16, item * NF * 16, N));
        if (PIECES >= 128) {
            const duo_u32x4 v0 = src[lane], v1 = src[lane + 64];
            __builtin_amdgcn_raw_buffer_store_b128(v0, rs, (int)(dst + m * 2048 + lane * 16), 0, 16);                // aux 16 = sc1: write-through
            __builtin_amdgcn_raw_buffer_store_b128(v1, rs, (int)(dst + m * 2048 + (lane + 64) * 16), 0, 16);
        } else if (lane < PIECES) {
            const duo_u32x4 v0 = src[lane];
            __builtin_amdgcn_raw_buffer_store_b128(v0, rs, (int)(dst + m * 2048 + lane * 16), 0, 16);
        }
    }
}
// block j of the tile `out`, just written by this wave, to its place in an F-layout operand in HBM (plain 16-byte stores; the
// strip's row tiles are contiguous there, so the tile's own offsets apply): the hand-over between two launches of a split stack
template <typename T, int RT, int NF = 4>
__device__ __forceinline__ void strip_final_push(T* __restrict__ dstF, const T* out, const int N, const int sidx, const int j, const int lane)
{   // item j = columns 16 NF j ..: per row tile 16 x 16 NF elements, contiguous in the fragment-tiled layout (NF = 1, bf16: half a fragment)
    constexpr int PIECES = 16 * NF * (int)sizeof(T);             // 16-byte pieces per row tile
    T* base = dstF + (size_t)sidx * RT * 16 * N;
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const size_t o = ft_off<T>(m * 16, j * NF * 16, N);
        const duo_u32x4* src = reinterpret_cast<const duo_u32x4*>(out + o);
        duo_u32x4* dst = reinterpret_cast<duo_u32x4*>(base + o);
        if (PIECES >= 128) {
            duo_u32x4 v[PIECES / 64 > 0 ? PIECES / 64 : 1];
#pragma unroll
            for (int i = 0; i < PIECES / 64; ++i) v[i] = src[lane + 64 * i];
#pragma unroll
            for (int i = 0; i < PIECES / 64; ++i) dst[lane + 64 * i] = v[i];
        } else if (lane < PIECES) dst[lane] = src[lane];
    }
}
// every wave has drained its pushes -> flag -> the partner's flag -> the partner's blocks into `out`.  `pull` = false: this
// workgroup has nothing left to compute (it only publishes).  All 64 NW threads call it.
template <typename T, int RT, int NW = STRIP_NW>
__device__ __forceinline__ void duo_swap(const StripDuo& d, T* out, const int N, const int nblk, const int h, const int parity, const int seq,
                                         const bool pull, long long* dbg)
{
    long long ta = 0, tb = 0;
    if (dbg && threadIdx.x == 0) ta = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's sc1 stores have left (and whatever else it had in flight)
    lds_barrier();
    const int want = d.epoch * 16 + seq;
    if (dbg && threadIdx.x == 0) { tb = (long long)__builtin_amdgcn_s_memtime(); dbg[11] += tb - ta; }    // drain + barrier
    if (threadIdx.x == 0) {
        __hip_atomic_store(d.flags + blockIdx.x, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pull) {
            int n = 0;
            while (__hip_atomic_load(d.flags + (blockIdx.x ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++n > DUO_SPIN_LIMIT) { atomicOr(d.err, 2); break; }      // a partner that never arrives: report, do not hang
            }
        }
    }
    if (!pull) return;
    lds_barrier();
    if (dbg && threadIdx.x == 0) { ta = (long long)__builtin_amdgcn_s_memtime(); dbg[12] += ta - tb; }    // flag + the partner's
    // a block = RT x 2 KB = RT x 128 pieces of 16 bytes: threads 0..255 take the partner's blocks 0, 2, ..., threads 256..511 the odd ones
    const int ph = h ^ 1, cntp = (nblk - ph + 1) >> 1, t = threadIdx.x, pc = t & 255, m = pc >> 7, idx = pc & 127;
    constexpr int G = NW / 4;                                        // groups of 256 threads: group g takes the partner's blocks g, g + G, ...
    const __amdgpu_buffer_rsrc_t rs = duo_rsrc(d);
    const size_t src = ((size_t)(blockIdx.x ^ 1) * d.xch_wg + (size_t)parity * (d.xch_wg >> 1)) * 8 + (size_t)pc * 16;                 // bytes
    for (int r0 = 0; r0 < cntp && m < RT; r0 += 4 * G) {             // four loads in flight per thread
        duo_u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + (t >> 8) + G * k;
            if (r < cntp) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(src + (size_t)r * (RT * 2048)), 0, 16);           // sc1: from L2
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + (t >> 8) + G * k;
            if (r < cntp) reinterpret_cast<duo_u32x4*>(out + ft_off<T>(m * 16, (2 * r + ph) * 64, N))[idx] = v[k];
        }
    }
    if (dbg && threadIdx.x == 0) dbg[13] += (long long)__builtin_amdgcn_s_memtime() - ta;                  // pull (thread 0's share)
}

extern __shared__ __align__(16) unsigned char strip_smem[];   // the strip kernels' dynamic LDS: two tiles [RT * 16][maxD] (+ LDS weight copies)
template <typename T, int RT, int NF, int NW>
__device__ __forceinline__ void strip_fwd_body(const StripFwdArgs<T>& a, const int maxD)
{
    constexpr int EPL = Traits<T>::EPL, KS = Traits<T>::KS, PER = 4 / NF;       // NF < 4: no pairs (the host launches those without StripDuo)
    T* in = reinterpret_cast<T*>(strip_smem);
    T* out = in + (size_t)RT * 16 * maxD;
    T* wl = out + (size_t)RT * 16 * maxD;                        // RT = 1: LDS copies of the small products' weights (a.wlds)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = a.duo.on ? (int)(blockIdx.x & 1) : 0, sidx = a.duo.on ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int row0 = sidx * RT * 16;
    STRIP_STAMP(0);
    if (a.warm) strip_warm_args<(int)sizeof(StripFwdArgs<T>)>();
    if (a.dbg && threadIdx.x == 0) { for (int i = 10; i < 16; ++i) a.dbg[(size_t)blockIdx.x * 16 + i] = 0; }
    StripB<T, NF> pb;
    StripItem nx = strip_next(a, StripItem{0, wave - NW}, wave, true, h, PER, NW);
    const int rot = (a.rot == 1) ? (sidx >> 3) : (a.rot == 2 ? sidx : 0);   // workgroups g, g + 8, ... share an XCD
    // the last product this workgroup computes: the second of a pair stops behind the last product the pair splits
    int last = a.n - 1;
    if (h == 1) { last = -1; for (int p = 0; p < a.n; ++p) if (duo_split(a.duo, a.Dp[p + 1] / 64)) last = p; }
    auto blocks = [&](const int p, bool& split, int& cnt) { split = duo_split(a.duo, a.Dp[p + 1] / 64); cnt = strip_items(a, p, true, h, PER); };
    auto prefetch_next = [&]() {
        if (nx.p < a.n && nx.p <= last) {
            bool sp; int cn; blocks(nx.p, sp, cn);
            const int wo = RT == 1 ? a.wlds[nx.p] : -1;
            // (spelled from the LDS array itself: through the captured pointer the address space was lost and the loads came out flat_)
            if (wo >= 0) strip_prefetch<T, NF>(pb, reinterpret_cast<const T*>(strip_smem) + (size_t)2 * RT * 16 * maxD + wo, a.Dp[nx.p] / KS, strip_phys(sp, nx.blk, cn, h, rot, PER), lane);
            else strip_prefetch<T, NF>(pb, a.W[nx.p], a.Dp[nx.p] / KS, strip_phys(sp, nx.blk, cn, h, rot, PER), lane);
        }
    };
    if (RT == 1) {                                               // the small products' weights -> LDS (contiguous, fragment-tiled: offsets carry over)
        const bool first_lds = nx.p < a.n && a.wlds[nx.p] >= 0;
        if (!first_lds) prefetch_next();                         // a first item streamed from L2 (the big first product) does not wait for the copy
        bool any = false;
        for (int p = 0; p < a.n; ++p)
            if (a.wlds[p] >= 0) { strip_load<T>(wl + a.wlds[p], a.W[p], a.Dp[p] * a.Dp[p + 1] / EPL); any = true; }
        if (any) lds_barrier();                                  // (uniform: `any` comes from the arguments)
        if (first_lds) prefetch_next();
    } else prefetch_next();
    // the strip of a0: row tiles RT sidx .. of an F-layout operand are contiguous
    strip_load<T>(in, a.a0 + (size_t)sidx * RT * 16 * a.Dp[0], RT * 16 * a.Dp[0] / EPL);
    lds_barrier();
    STRIP_STAMP(1);
    f32x4 acc[RT][NF];
    int seq = 0;
    for (int l = 0; l <= last; ++l) {
        const int nkt = a.Dp[l] / KS, N = a.Dp[l + 1];
        const bool hidden = !a.has_out || l + 1 < a.n;
        const bool fin = !a.has_out && l + 1 == a.n;            // the launch stops here: blocks go to HBM, nobody swaps
        bool split; int cnt; blocks(l, split, cnt);
        while (nx.p == l) {
            const int blk = strip_phys(split, nx.blk, cnt, h, rot, PER);
            const bool det = a.dbg && l == a.sel && threadIdx.x == 0;
#define DET(i) do { if (det) a.dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
            DET(10);
            if (hidden) {
                EpiIpFwd<T> ef = a.ef[l];                       // the layer's epilogue parameters: scalar registers, loaded once
                if (a.duo.on && !split && h == 1) ef.outT = nullptr;     // (not reached: the second workgroup computes split products only)
                typename EpiIpFwd<T>::Aux ax[RT][NF];
                strip_aux<T, RT, NF>(ax, ef, row0, blk, lane);
                if (RT == 1 && a.wlds[l] >= 0) strip_product<T, RT, NF>(acc, pb, in, wl + a.wlds[l], nkt, blk, lane);
                else strip_product<T, RT, NF>(acc, pb, in, a.W[l], nkt, blk, lane);
                DET(11);
                nx = strip_next(a, nx, wave, true, h, PER, NW);
                prefetch_next();
                DET(12);
                strip_epilogue<T, RT, NF>(acc, ax, ef, out, N, row0, blk, lane);
                DET(13);
                if (fin) { if (split || h == 0) strip_final_push<T, RT, NF>(a.finalF, out, N, sidx, blk, lane); }
                else if (split) duo_push<T, RT, NF>(a.duo, out, N, blk, seq & 1, lane);
            } else {                                              // the output unit: logits, loss, delta (column 0): item 0 = fragment 0 ..
                if (RT == 1 && a.wlds[l] >= 0) strip_product<T, RT, NF>(acc, pb, in, wl + a.wlds[l], nkt, blk, lane);
                else strip_product<T, RT, NF>(acc, pb, in, a.W[l], nkt, blk, lane);
                nx.p = a.n;
                const EpiIpOut<T> eo = a.eo;
                const int rq = 4 * (lane >> 4), cl = lane & 15;
                // only column 0 is a unit: columns 1..63 of the delta (both layouts) are zero and stay zero -- the
                // buffers are zero-filled at creation and these kernels are their only writers -- so one fragment per row tile
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    float v[4];
                    const int r0 = row0 + m * 16 + rq;
                    eo.pre(r0, cl, acc[m][0], v);
                    if (eo.outT) store4(eo.outT + ft_off<T>(cl, r0, eo.ldT), v[0], v[1], v[2], v[3]);
                    if (eo.outF) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) eo.outF[ft_off<T>(r0 + r, cl, eo.ld)] = (T)v[r];
                    }
                }
            }
        }
        if (split && !fin) { duo_swap<T, RT, NW>(a.duo, out, N, N / 64, h, seq & 1, seq + 1, l < last, a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr); ++seq; }
        if (hidden && !fin) lds_barrier();
        STRIP_STAMP(2 + l);
        T* t = in; in = out; out = t;
    }
}
template <typename T, int RT, int NF = 4, int NW = STRIP_NW>
static __global__ __launch_bounds__(64 * NW) void k_ip_strip_fwd(const StripFwdArgs<T> a, const int maxD) { strip_fwd_body<T, RT, NF, NW>(a, maxD); }

template <typename T, int RT, int NF, int NW>
__device__ __forceinline__ void strip_bwd_body(const StripBwdArgs<T>& a, const int maxD)
{
    constexpr int EPL = Traits<T>::EPL, KS = Traits<T>::KS, PER = 4 / NF;
    T* in = reinterpret_cast<T*>(strip_smem);
    T* out = in + (size_t)RT * 16 * maxD;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = a.duo.on ? (int)(blockIdx.x & 1) : 0, sidx = a.duo.on ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int row0 = sidx * RT * 16;
    STRIP_STAMP(0);
    if (a.warm) strip_warm_args<(int)sizeof(StripBwdArgs<T>)>();
    if (a.dbg && threadIdx.x == 0) { for (int i = 10; i < 16; ++i) a.dbg[(size_t)blockIdx.x * 16 + i] = 0; }
    T* wl = out + (size_t)RT * 16 * maxD;                        // RT = 1: LDS copies of the small products' weights (a.wlds)
    StripB<T, NF> pb;                                             // item index q = n - t: product t = n - q
    StripItem nx = strip_next(a, StripItem{0, wave - NW}, wave, false, h, PER, NW);
    const int rot = (a.rot == 1) ? (sidx >> 3) : (a.rot == 2 ? sidx : 0);   // workgroups g, g + 8, ... share an XCD
    auto blocks = [&](const int q, bool& split, int& cnt) { split = duo_split(a.duo, a.Dp[a.n - q - 1] / 64); cnt = strip_items(a, q, false, h, PER); };
    auto prefetch_next = [&]() {
        if (nx.p < a.n) {
            bool sp; int cn; blocks(nx.p, sp, cn);
            const int wo = RT == 1 ? a.wlds[a.n - nx.p - 1] : -1;
            if (wo >= 0) strip_prefetch<T, NF>(pb, reinterpret_cast<const T*>(strip_smem) + (size_t)2 * RT * 16 * maxD + wo, a.Dp[a.n - nx.p] / KS, strip_phys(sp, nx.blk, cn, h, rot, PER), lane);
            else strip_prefetch<T, NF>(pb, a.W[a.n - nx.p - 1], a.Dp[a.n - nx.p] / KS, strip_phys(sp, nx.blk, cn, h, rot, PER), lane);
        }
    };
    if (RT == 1) {                                               // the small products' weights -> LDS; in the backward launch they are the FIRST products
        bool any = false;
        for (int p = 0; p < a.n; ++p)
            if (a.wlds[p] >= 0) { strip_load<T>(wl + a.wlds[p], a.W[p], a.Dp[p] * a.Dp[p + 1] / EPL); any = true; }
        if (any) lds_barrier();                                  // the first item's prefetch reads them
    }
    prefetch_next();
    strip_load<T>(in, a.dlast + (size_t)sidx * RT * 16 * a.Dp[a.n], RT * 16 * a.Dp[a.n] / EPL);
    lds_barrier();
    STRIP_STAMP(1);
    f32x4 acc[RT][NF];
    int seq = 0;
    for (int t = a.n; t >= 1; --t) {                              // delta l_{t-1} = (delta l_t . W_t^T) * mask * act'
        const int nkt = a.Dp[t] / KS, N = a.Dp[t - 1], q = a.n - t;
        bool split; int cnt; blocks(q, split, cnt);
        while (nx.p == q) {
            const int blk = strip_phys(split, nx.blk, cnt, h, rot, PER);
            EpiIpBwd<T> eb = a.eb[t - 1];
            if (a.duo.on && !split && h == 1) { eb.outT = nullptr; eb.out32 = nullptr; }   // a narrow product of a pair: both compute it, the first one stores it
            typename EpiIpBwd<T>::Aux ax[RT][NF];
            strip_aux<T, RT, NF>(ax, eb, row0, blk, lane);
            if (RT == 1 && a.wlds[t - 1] >= 0) strip_product<T, RT, NF>(acc, pb, in, wl + a.wlds[t - 1], nkt, blk, lane);
            else strip_product<T, RT, NF>(acc, pb, in, a.W[t - 1], nkt, blk, lane);
            nx = strip_next(a, nx, wave, false, h, PER, NW);
            prefetch_next();
            strip_epilogue<T, RT, NF>(acc, ax, eb, (t > 1 || !a.bottom) ? out : nullptr, N, row0, blk, lane);
            if (split && t > 1) duo_push<T, RT, NF>(a.duo, out, N, blk, seq & 1, lane);
            if (t == 1 && !a.bottom && (split || h == 0)) strip_final_push<T, RT, NF>(a.finalF, out, N, sidx, blk, lane);
        }
        if (split && t > 1) { duo_swap<T, RT, NW>(a.duo, out, N, N / 64, h, seq & 1, seq + 1, true, a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr); ++seq; }
        if (t > 1) lds_barrier();
        STRIP_STAMP(2 + q);
        T* x = in; in = out; out = x;
    }
}
template <typename T, int RT, int NF = 4, int NW = STRIP_NW>
static __global__ __launch_bounds__(64 * NW) void k_ip_strip_bwd(const StripBwdArgs<T> a, const int maxD) { strip_bwd_body<T, RT, NF, NW>(a, maxD); }
// The narrow tail of a training step, forward then backward, in ONE launch of 16-example strips: what the backward half reads of the
// forward half -- the output delta (F layout, 16 x 64) and the transposed activations its act' needs -- was stored by this same
// workgroup, so a drained store queue and a workgroup barrier order them (this CU's L1 holds none of those lines: the forward half
// never read them; stores write through to the L2).  Saves a launch and the second tile load's cold start.
template <typename T, int NF, int NW>
static __global__ __launch_bounds__(64 * NW) void k_ip_strip_tail(const StripFwdArgs<T> fa, const StripBwdArgs<T> ba, const int maxD)
{
    strip_fwd_body<T, 1, NF, NW>(fa, maxD);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    strip_bwd_body<T, 1, NF, NW>(ba, maxD);
}

// Keep-masks [B][d] uint8 (the ABI's layout, reference column order) -> [Dp][ldT] uint8, zero padded,
// for every layer in one launch; layer 0's columns are mapped to the slot layout on the way.
struct MaskTArgs {
    const uint8_t* src[IPNN_MAX_HIDDEN + 1]; uint8_t* dst[IPNN_MAX_HIDDEN + 1]; int d[IPNN_MAX_HIDDEN + 1], Dp[IPNN_MAX_HIDDEN + 1];
    int tile0[IPNN_MAX_HIDDEN + 2]; int n; const int* ref0; int B, Ba, ldT; bool wt;
};
static __global__ __launch_bounds__(256) void k_mask_T(const MaskTArgs a)
{
    __shared__ __align__(4) uint8_t s[64][68];          // 17 dwords per row: the 64 lanes of a byte store (one row each) spread over all banks
    int t = 0;
#pragma unroll
    for (int q = 1; q <= IPNN_MAX_HIDDEN; ++q) t += (q < a.n && (int)blockIdx.x >= a.tile0[q]) ? 1 : 0;
    const int local = (int)blockIdx.x - a.tile0[t], ntx = a.Ba / 64;
    const int t0 = (local % ntx) * 64, c0 = (local / ntx) * 64;
    {   // a thread's column is the same in all 16 of its elements (i = tid + 256 k): its source column once, then all
        // 16 bytes in one round trip, then the LDS transposition
        // (four such tiles per workgroup, 64 loads in flight per thread, changed nothing: 12.0 against 12.2 us -- not a latency chain)
        const int cx = threadIdx.x & 63, c = c0 + cx, d = a.d[t];
        int sc = c < a.Dp[t] ? c : -1;
        if (t == 0) sc = sc >= 0 ? a.ref0[sc] : -1; else if (sc >= d) sc = -1;
        const uint8_t* __restrict__ src = a.src[t] + (sc >= 0 ? sc : 0);
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ex = t0 + (threadIdx.x >> 6) + 4 * k;
            v[k] = (ex < a.B && sc >= 0) ? src[(size_t)ex * d] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s[cx][(threadIdx.x >> 6) + 4 * k] = v[k];
    }
    __syncthreads();
    const int cc = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (c0 + cc < a.Dp[t])
    {
        const unsigned* w = reinterpret_cast<const unsigned*>(&s[cc][16 * q]);
        store16_sel(a.wt, a.dst[t] + mask_off(c0 + cc, t0 + 16 * q, a.ldT), make_uint4(w[0], w[1], w[2], w[3]));
    }
}

// W_t <- W_t - lr * sum of slabs; refresh both tiled shadows.  Layer 1 rows are in slot layout.
template <typename T>
static __global__ void k_ip_update(float* __restrict__ W, const float* __restrict__ slab, int splitk, size_t zstride,
                                   float lr, int Din, int Dout, T* __restrict__ wf, T* __restrict__ wb)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Din * Dout) return;
    float w = W[i];
    if (slab) {
        float g = 0.f;
        for (int z = 0; z < splitk; ++z) g += slab[(size_t)z * zstride + i];
        w -= lr * g; W[i] = w;
    }
    const int r = (int)(i / Dout), c = (int)(i % Dout);
    wf[ft_off<T>(c, r, Din)] = (T)w;
    wb[ft_off<T>(r, c, Dout)] = (T)w;
}
__device__ inline float adam_step(float w, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    return w - lr_t * m / (sqrtf(v) + eps);
}

// TensorFlow's FtrlOptimizer(learning_rate) as python/tf_util.py:21-24 builds it (learning_rate_power -0.5, initial
// accumulator 0.1, l1 = l2 = 0; the ApplyFtrl kernel): state = (accum, linear).  A variable with a zero gradient keeps its
// accumulator and linear term, and is RE-DERIVED from them: w = -linear lr / sqrt(accum) -- with the dense table gradient
// of this graph, rows no example has touched yet drop to 0 at the first step, as they do in the reference.
__device__ inline float ftrl_step(float w, float g, float& accum, float& linear, float lr) {
    const float na = accum + g * g, sa = sqrtf(na);
    linear += g - g * g / (sa + sqrtf(accum)) / lr * w;         // sqrt(na) - sqrt(accum), written without the cancellation
    accum = na;
    return linear != 0.f ? -linear / (sa / lr) : 0.f;
}
__device__ inline float opt_step(int opt, float w, float g, float& s0, float& s1, float lr, float b1, float b2, float eps) {
    return opt == IPNN_OPT_FTRL ? ftrl_step(w, g, s0, s1, lr) : adam_step(w, g, s0, s1, lr, b1, b2, eps);
}

static __global__ void k_fill_f32(float* __restrict__ p, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Adam on the embedding table: TensorFlow's gradient through concat / slice is DENSE (zero for
// untouched rows), so every row's moments decay and every row moves each step -- one streaming pass
// over table, m, v and the per-row gradient sums G (which it zeroes again for the next step).
// (FTRL: the same pass with (accum, linear) in place of the moments.)
static __global__ void k_adam_table(float* __restrict__ tab, float* __restrict__ m, float* __restrict__ v, float* __restrict__ G,
                                    size_t n, float lr_t, float b1, float b2, float eps, int opt)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 w = *reinterpret_cast<float4*>(tab + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
    const float4 g = *reinterpret_cast<const float4*>(G + i);
    w.x = opt_step(opt, w.x, g.x, mm.x, vv.x, lr_t, b1, b2, eps); w.y = opt_step(opt, w.y, g.y, mm.y, vv.y, lr_t, b1, b2, eps);
    w.z = opt_step(opt, w.z, g.z, mm.z, vv.z, lr_t, b1, b2, eps); w.w = opt_step(opt, w.w, g.w, mm.w, vv.w, lr_t, b1, b2, eps);
    *reinterpret_cast<float4*>(tab + i) = w; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
    *reinterpret_cast<float4*>(G + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// One launch for the whole stack: W_t <- W_t - lr * (sum of its split-K slabs), both tiled shadows
// refreshed; the last workgroup applies the bias-of-z1 gradient and reduces the per-example losses.
struct IpUpdArgs {
    bool wt;                                         // weights and shadows written through (IPNN_WT)
    float* W[IPNN_MAX_HIDDEN + 1]; void* wf[IPNN_MAX_HIDDEN + 1]; void* wb[IPNN_MAX_HIDDEN + 1];
    size_t off[IPNN_MAX_HIDDEN + 2];                 // element offset of every layer in a slab; off[n] = total
    int Din[IPNN_MAX_HIDDEN + 1], Dout[IPNN_MAX_HIDDEN + 1], sk[IPNN_MAX_HIDDEN + 1]; int n;
    const float* slab; size_t zstride; float lr;
    float* b; const float* gb_part; int ngb; const float* loss_t; int Ba; float* loss_sum;
    // Adam (python/tf_util.py:17-20, TensorFlow's AdamOptimizer): first / second moments beside every tensor;
    // lr is then lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) of this step
    int adam;                                        // 0 = SGD, else IPNN_OPT_ADAM / IPNN_OPT_FTRL (state = (accum, linear))
    float beta1, beta2, eps; float* Wm[IPNN_MAX_HIDDEN + 1]; float* Wv[IPNN_MAX_HIDDEN + 1]; float* bmv;
    const int* err;                                  // bit 1 set (a strip pair gave up its swap): the step's activations are invalid -- nothing is applied
};

// The scalar b of z1 (python/FNN_IP_L7.py:104-114): its gradient is the sum of the per-workgroup partials of k_ip_bwd.  A launch
// of its own on the side stream, right behind k_ip_bwd, so that the dense update on the main stream depends on nothing there.
static __global__ __launch_bounds__(256) void k_ip_b_update(float* __restrict__ b, const float* __restrict__ gb_part, int ngb, int opt,
                                                            float* __restrict__ bmv, float lr, float beta1, float beta2, float eps, const int* __restrict__ err)
{
    __shared__ float sg[256];
    if (*err & 2) return;
    sg[threadIdx.x] = strided_sum256(gb_part, ngb);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sg[threadIdx.x] += sg[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (opt) *b = opt_step(opt, *b, sg[0], bmv[0], bmv[1], lr, beta1, beta2, eps);
        else *b -= lr * sg[0];
    }
}

template <typename T>
static __global__ __launch_bounds__(256) void k_ip_update_all(const IpUpdArgs u)
{
    if (blockIdx.x == gridDim.x - 1) {               // scalar tail: the loss, a fixed-shape tree (b: k_ip_b_update)
        __shared__ float sl[256];
        sl[threadIdx.x] = strided_sum256(u.loss_t, u.Ba); __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sl[threadIdx.x] += sl[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *u.loss_sum = sl[0];
        return;
    }
    // A workgroup owns a 64 x 64 tile of one layer's W (row lengths are multiples of 64); a thread 4 consecutive elements
    // of 4 rows: 16-byte slab / master / state accesses and 8-byte pieces of the backward shadow (k = column).  The forward
    // shadow has k = row: the tile goes through LDS transposed and leaves as whole 16-byte lane slots.
    typedef typename Traits<T>::frag frag;
    constexpr int EPL = Traits<T>::EPL;
    __shared__ __align__(16) T sT[64][64 + EPL];                 // [column][row], padded
    const size_t tile = blockIdx.x;
    if (tile * 4096 >= u.off[u.n]) return;
    if (*u.err & 2) return;                                      // a strip pair gave up (StripDuo): the gradients are garbage, the weights stay
    int t = 0;
#pragma unroll
    for (int q = 1; q <= IPNN_MAX_HIDDEN; ++q) t += (q < u.n && tile * 4096 >= u.off[q]) ? 1 : 0;
    const int Dout = u.Dout[t], Din = u.Din[t], ntc = Dout / 64;
    const int lt = (int)(tile - u.off[t] / 4096), r0 = (lt / ntc) * 64, c0 = (lt % ntc) * 64;
    const int cq = (threadIdx.x & 15) * 4, sk = u.sk[t];
    T* wb = static_cast<T*>(u.wb[t]);
    // the split-K slabs of the thread's 4 x 4 elements, summed in slab order: the loads of four slabs x four rows are issued
    // together (a `for z: load, add` loop pays one memory round trip per slab and row: 21 -> 17 us for the whole launch)
    float4 gs[4], ws[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ws[k] = *reinterpret_cast<const float4*>(u.W[t] + (size_t)(r0 + (threadIdx.x >> 4) + 16 * k) * Dout + c0 + cq);
    }
    for (int z0 = 0; z0 < sk; z0 += 4) {
        float4 v[4][4];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t i = u.off[t] + (size_t)(r0 + (threadIdx.x >> 4) + 16 * k) * Dout + c0 + cq;
                v[zz][k] = z0 + zz < sk ? *reinterpret_cast<const float4*>(u.slab + (size_t)(z0 + zz) * u.zstride + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
            for (int k = 0; k < 4; ++k) { gs[k].x += v[zz][k].x; gs[k].y += v[zz][k].y; gs[k].z += v[zz][k].z; gs[k].w += v[zz][k].w; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + (threadIdx.x >> 4) + 16 * k, c = c0 + cq;
        const size_t j = (size_t)r * Dout + c;
        const float4 g = gs[k];
        float4 w = ws[k];
        if (u.adam) {
            float4 m = *reinterpret_cast<const float4*>(u.Wm[t] + j), v = *reinterpret_cast<const float4*>(u.Wv[t] + j);
            w.x = opt_step(u.adam, w.x, g.x, m.x, v.x, u.lr, u.beta1, u.beta2, u.eps);
            w.y = opt_step(u.adam, w.y, g.y, m.y, v.y, u.lr, u.beta1, u.beta2, u.eps);
            w.z = opt_step(u.adam, w.z, g.z, m.z, v.z, u.lr, u.beta1, u.beta2, u.eps);
            w.w = opt_step(u.adam, w.w, g.w, m.w, v.w, u.lr, u.beta1, u.beta2, u.eps);
            *reinterpret_cast<float4*>(u.Wm[t] + j) = m; *reinterpret_cast<float4*>(u.Wv[t] + j) = v;
        } else { w.x -= u.lr * g.x; w.y -= u.lr * g.y; w.z -= u.lr * g.z; w.w -= u.lr * g.w; }
        store16_sel(u.wt, u.W[t] + j, w);                       // (written through: see store4_wt)
        store4_sel(u.wt, wb + ft_off<T>(r, c, Dout), w.x, w.y, w.z, w.w);
        const int rl = r - r0;
        sT[cq][rl] = (T)w.x; sT[cq + 1][rl] = (T)w.y; sT[cq + 2][rl] = (T)w.z; sT[cq + 3][rl] = (T)w.w;
    }
    __syncthreads();
    T* wf = static_cast<T*>(u.wf[t]);
    for (int e = threadIdx.x; e < 64 * (64 / EPL); e += 256) {      // (column, group of EPL rows): one lane slot of wf
        const int cl = e & 63, g8 = e >> 6;
        store16_sel(u.wt, wf + ft_off<T>(c0 + cl, r0 + g8 * EPL, Din), *reinterpret_cast<const frag*>(&sT[cl][g8 * EPL]));
    }
}

}  // namespace

struct ipnn_handle {
    ipnn_cfg cfg{}; std::string err; int dev = 0; hipStream_t st = nullptr; bool own_stream = false;
    int F = 0, K = 0, L = 0, P = 0, CB = 0, Bmax = 0, ldT = 0; bool bf16 = false; int splitk = 8;   // splitk: slab capacity
    std::vector<int> sk;                         // split-K of each layer's weight-gradient product
    bool adam = false; int64_t adam_t = 0;       // Adam / FTRL: two state tensors beside every variable, dense row-gradient table; step count
    bool ftrl = false;
    bool loss_mean = false;                      // ipnn_set_loss_mean: the loss is the batch MEAN (gradients scaled by 1 / B)
    std::vector<float*> Wm, Wv; float *tm = nullptr, *tv = nullptr, *tG = nullptr, *bmv = nullptr;
    std::vector<int> d, Dp;                      // d[0..L+1], padded
    float* table16 = nullptr; int64_t n_rows = 0; float* b = nullptr;
    std::vector<float*> W; std::vector<void*> wf, wb;           // W[t], t = 1..L+1 (index t-1)
    std::vector<void*> a, aT, dl, dlT;                           // a[t] t=0..L ; dl[t] t=1..L+1 (index t-1)
    std::vector<uint8_t*> maskT;                                 // keep-masks of a step, transposed [Dp_t][ldT], t = 0..L
    hipStream_t st2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mask = nullptr, ev_bwd = nullptr;
    // IPNN_UPDATE_SIDE=1 (default 0, measured slower -- DESIGN.md section 4): the dense update runs at the END of the side chain (behind ev_wg: the weight gradients) and the main
    // stream does NOT wait for it at the end of the step: the next call's mask transposition and gather -- which read neither the dense
    // weights nor the slabs -- run beside it, and the wait (ev_join) sits in front of the first product.  ev_tab: the side chain's table /
    // bias half is done (in front of the next gather).  Every other entry point joins first (ip_join).
    hipEvent_t ev_tab = nullptr, ev_wg = nullptr;
    bool upd_side = false, tab_pending = false, upd_pending = false;
    // side stream (IPNN_SIDE_STREAM=0: everything in line): the id grouping from the start of the step; the inner-product backward,
    // the scalar b and the sparse-row update beside the weight gradients.  ev_fork / ev_bwd: main -> side; ev_join: side -> main at
    // the end of the step (ev_mask only with IPNN_MASK_SIDE=1)
    float* emb = nullptr;                            // [ldT][F*16] raw embeddings of the step's examples (forward -> backward)
    float *dz0 = nullptr, *gxp = nullptr, *gb_part = nullptr, *loss_t = nullptr, *loss_dev = nullptr, *slab = nullptr;
    int* ref0 = nullptr; int* err_flag = nullptr;
    int4* rec = nullptr; double* part = nullptr; int4* owners = nullptr; int* owner_cnt = nullptr; void* skeys = nullptr;
    double* cpow1 = nullptr; bool key64 = true;
    size_t slab_stride = 0;
    bool prof = false;                               // HIP-event timing of the step's segments
    long long* stamps = nullptr;                     // IPNN_STAMPS=1: [2][Ba/16][16] time stamps of the strip kernels
    bool group_wgrad = true;                         // IPNN_GROUP_WGRAD=0: one launch per weight-gradient product
    bool strip_attr = false;
    int mask_side = 0;                               // IPNN_MASK_SIDE=1: the mask transposition on the side stream
    int group_xcd = 1;                               // IPNN_GROUP_XCD=0: tiles in launch order
    int strip_rot = 1, fwd_skip = 0;                 // IPNN_STRIP_ROT (0: every workgroup walks the blocks in the same order), IPNN_FWD_SKIP (diagnostics)
    int wide_nf = 2, wide_nw = 8;                    // IPNN_WIDE=nf:nw -- items (16-column fragments) and waves of the wide (pair) launches: 2:8 (default), 4:8, 2:12
    bool wt = true;                                  // IPNN_WT=0: plain stores where the launches write through by default
    int tail_fuse = 1;                               // IPNN_TAIL_FUSE: training steps run the forward and the backward tail in one launch
    int tail_nf = 1;                                 // IPNN_TAIL_NF: 16-column fragments per item in the 16-example strips of the narrow tail (1 / 2 / 4)
    int tail_nw = 16;                                // IPNN_TAIL_NW: waves per workgroup there (16 with IPNN_TAIL_NF=1 only: 128 registers per lane)
    int ipf_nt = 1024;                               // IPNN_IPF_NT: threads per workgroup of the gather + inner-product launch (512 / 1024)
    int strip_warm = 1;                              // IPNN_STRIP_WARM: the strip kernels touch their argument lines at the start (strip_warm_args)
    bool strip = true;                               // IPNN_STRIP=0: one GEMM launch per product instead of the strip kernels
    int duo = 1, duo_min = DUO_MIN_BLOCKS;           // IPNN_STRIP_DUO=0: one workgroup per strip (StripDuo); IPNN_DUO_MIN: narrowest product a pair splits
    unsigned long long* duo_xch = nullptr; int* duo_flags = nullptr; int duo_epoch = 0; size_t duo_xch_wg = 0; int n_cu = 256;
    bool stamp_tail = false;                         // IPNN_STAMPS=2: the stamps of the 16-example-strip launches instead of the wide ones
    int tail_split = 1;                              // IPNN_TAIL_SPLIT=0: the whole stack in one strip launch per direction (round 2's form)
    bool duo_failed = false;                         // a pair gave up once: every later train step is refused until the handle is re-created
    bool gemm_lds = false;                           // IPNN_GEMM_LDS=1: LDS-staged k_gemm_lds for the wide products (measured equal to k_gemm_ft: both L2-bound)
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_ev;
};

namespace {
struct IpProf {                                      // one segment of ip_run on the handle's stream
    ipnn_handle* h; const char* name; hipStream_t s; hipEvent_t b = nullptr, e = nullptr;
    IpProf(ipnn_handle* h_, const char* n, hipStream_t s_ = nullptr) : h(h_), name(n), s(s_ ? s_ : h_->st) {
        if (!h->prof) return;
        hipEventCreate(&b); hipEventCreate(&e); hipEventRecord(b, s);
    }
    ~IpProf() { if (!h->prof) return; hipEventRecord(e, s); h->prof_ev[name].emplace_back(b, e); }
};
}

#define IHK(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return FNN_ERR_HIP; } } while (0)
#define IFAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

namespace {

size_t ts(const ipnn_handle* h) { return h->bf16 ? 2 : 4; }

template <typename T> void ip_refresh(ipnn_handle* h, int t, const float* slab, float lr) {      // t = 1..L+1
    const int Din = h->Dp[t - 1], Dout = h->Dp[t];
    const size_t n = (size_t)Din * Dout;
    hipLaunchKernelGGL((k_ip_update<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->W[t - 1], slab, h->sk[t - 1],
                       h->slab_stride, lr, Din, Dout, (T*)h->wf[t - 1], (T*)h->wb[t - 1]);
}

inline int maxD2x(const int* Dp, int n) { int m = 0; for (int t = 0; t <= n; ++t) m = std::max(m, Dp[t]); return m; }

// the main stream waits for what the previous step left running on the side stream (see upd_side)
static int ip_join_table(ipnn_handle* h)
{
    if (h->tab_pending) { IHK(h, hipStreamWaitEvent(h->st, h->ev_tab, 0)); h->tab_pending = false; }
    return FNN_OK;
}
static int ip_join(ipnn_handle* h)
{
    const int rc = ip_join_table(h);
    if (rc != FNN_OK) return rc;
    if (h->upd_pending) { IHK(h, hipStreamWaitEvent(h->st, h->ev_join, 0)); h->upd_pending = false; }
    return FNN_OK;
}

template <typename T>
int ip_run(ipnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* const* masks, float* logits_out,
           float* p_out, bool train)
{
    const int Ba = rup(B, 256), L = h->L, F = h->F, ldT = h->ldT;
    const float keep = h->cfg.keep_prob, inv_keep = 1.0f / keep;
    const size_t lds_ip = (size_t)16 * (F * (SLOT + 1) + h->Dp[0]) * sizeof(float);    // k_ip_fwd pads its embedding tile
    const bool drop = train && masks;
    if (drop) for (int t = 0; t <= L; ++t) if (!masks[t]) IFAIL(h, FNN_ERR_ARG, "masks: null entry");
    { const int jrc = ip_join_table(h); if (jrc != FNN_OK) return jrc; }      // the previous step's sparse rows / bias (read by the gather)
    if (train) {
        // beside the stack, on the side stream: the grouping of the batch's ids for the sparse-row update (needed by the scatter);
        // the transposed keep-masks (needed from the first product on) go first on the main stream
        hipStream_t ss = h->st2 ? h->st2 : h->st;
        if (h->st2) { IHK(h, hipEventRecord(h->ev_fork, h->st)); IHK(h, hipStreamWaitEvent(h->st2, h->ev_fork, 0)); }
        if (drop) {   // keep-masks of all layers -> transposed, tiled, zero padded, slot-ordered for layer 0
            MaskTArgs ma{};
            int tiles = 0;
            for (int t = 0; t <= L; ++t) {
                ma.src[t] = masks[t]; ma.dst[t] = h->maskT[t]; ma.d[t] = h->d[t]; ma.Dp[t] = h->Dp[t]; ma.tile0[t] = tiles;
                tiles += (Ba / 64) * (h->Dp[t] / 64);
            }
            ma.tile0[L + 1] = tiles; ma.n = L + 1; ma.ref0 = h->ref0; ma.B = B; ma.Ba = Ba; ma.ldT = ldT; ma.wt = h->wt;
            // on the MAIN stream by default: a cross-stream event on the way into the first strip kernel costs more (10-20 us of
            // wait resolution, measured on the kernel trace) than the 10 us the transposition takes in line.  (Round 3 also tried
            // transposing the NEXT step's masks ahead, on the side stream -- beside the strips: 0.265 -> 0.270 ms per step, at the end
            // of the side chain beside the weight gradients: 0.286 -> 0.302 on a slower box: every kernel of this step is bound by
            // the CUs' ports or the L2, and a co-running launch takes what it saves.  Not kept.)
            IpProf pm(h, "mask_t", h->mask_side ? ss : h->st);
            hipLaunchKernelGGL(k_mask_T, dim3(tiles), dim3(256), 0, h->mask_side ? ss : h->st, ma);
            if (h->st2 && h->mask_side) IHK(h, hipEventRecord(h->ev_mask, h->st2));
        }
        SortArgs so{ids, B, F, h->n_rows, h->rec, h->owner_cnt, F, h->skeys};
        if (h->key64) {
            hipLaunchKernelGGL((k_sortA<unsigned long long>), dim3(4 * F), dim3(256), 0, ss, so);
            hipLaunchKernelGGL((k_sortB<unsigned long long>), dim3(16 * F), dim3(256), SORT_N * 8, ss, so);
        } else {
            hipLaunchKernelGGL((k_sortA<unsigned>), dim3(4 * F), dim3(256), 0, ss, so);
            hipLaunchKernelGGL((k_sortB<unsigned>), dim3(16 * F), dim3(256), SORT_N * 4, ss, so);
        }
    }
    {
        IpProf ps(h, "ip_fwd");
        IpFwdArgs fa{h->P, ids, B, F, h->K, h->table16, h->n_rows, h->b, (train && masks) ? masks[0] : nullptr, h->d[0],
                     (train && masks) ? inv_keep : 1.0f, h->cfg.act, h->Dp[0], ldT, h->err_flag, h->fwd_skip, h->wt};
        if (h->ipf_nt == 1024) hipLaunchKernelGGL((k_ip_fwd<T, 1024>), dim3(Ba / 16), dim3(1024), lds_ip, h->st, fa, (T*)h->a[0], (T*)h->aT[0], train ? h->emb : nullptr);
        else hipLaunchKernelGGL((k_ip_fwd<T>), dim3(Ba / 16), dim3(IPF_NT), lds_ip, h->st, fa, (T*)h->a[0], (T*)h->aT[0], train ? h->emb : nullptr);
    }
    // one product: C [M][N] = A . B^T on fragment-tiled operands; narrow problems take smaller wave tiles
    auto gemm = [&](const T* A, const T* Bm, int M, int N, int nkt_all, int nkt, int splitk, auto epi) {
        typedef decltype(epi) E;
        const int mt16 = M / 16, nt16 = N / 16;
        const long wg44 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
        const size_t lds4 = E::TILE ? gemm_ft_lds<T, 4>() : 0, lds2 = E::TILE ? gemm_ft_lds<T, 2>() : 0;
        if (wg44 >= 160 && h->gemm_lds) {
            constexpr size_t ldsb = gemm_lds_bytes<T, E::TILE>();
            // > 64 KiB of dynamic LDS needs the opt-in (opt-in path: set on every launch, a host-side call)
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_lds<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            hipLaunchKernelGGL((k_gemm_lds<T, E>), dim3((M + 127) / 128, (N + 127) / 128, splitk), dim3(256), ldsb, h->st, A, Bm,
                               mt16, nt16, nkt_all, nkt, epi);
        } else if (wg44 >= 160)
            hipLaunchKernelGGL((k_gemm_ft<T, 4, 4, E>), dim3((M + 127) / 128, (N + 127) / 128, splitk), dim3(256), lds4, h->st, A, Bm,
                               mt16, nt16, nkt_all, nkt, epi);
        else if (N > 64)
            hipLaunchKernelGGL((k_gemm_ft<T, 2, 4, E>), dim3((M + 63) / 64, (N + 127) / 128, splitk), dim3(256), lds4, h->st, A, Bm,
                               mt16, nt16, nkt_all, nkt, epi);
        else
            hipLaunchKernelGGL((k_gemm_ft<T, 2, 2, E>), dim3((M + 63) / 64, (N + 63) / 64, splitk), dim3(256), lds2, h->st, A, Bm,
                               mt16, nt16, nkt_all, nkt, epi);
    };
    if (drop && h->st2 && h->mask_side) IHK(h, hipStreamWaitEvent(h->st, h->ev_mask, 0));     // the strips / GEMMs read the transposed masks
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }            // the previous step's dense update (read from here on)
    constexpr int KS = Traits<T>::KS;
    int maxD = 0;
    for (int t = 0; t <= L + 1; ++t) maxD = std::max(maxD, h->Dp[t]);
    constexpr int RT = sizeof(T) == 2 ? 2 : 1;                   // strip = 32 (bf16) / 16 (f32) examples: two LDS tiles of 64 KiB at 1024 units
    const size_t strip_lds = (size_t)2 * RT * 16 * maxD * sizeof(T);
    const bool strip = h->strip && strip_lds <= 128 * 1024;
    // two workgroups per strip (StripDuo) where the pairs fit the chip at one workgroup per CU: both halves of a pair must be
    // resident to swap (bf16 strips of 32 examples: 256 workgroups at batch 4096)
    const int nstrips = Ba / (16 * RT);
    const bool duo = strip && h->duo && RT == 2 && 2 * nstrips <= h->n_cu;
    if (duo && !h->duo_xch) {
        h->duo_xch_wg = (size_t)2 * ((maxD / 64 + 1) / 2) * RT * 256;       // 8-byte words: [2 parities][own blocks][RT][256]
        IHK(h, hipMalloc((void**)&h->duo_xch, (size_t)(h->ldT / (16 * RT)) * 2 * h->duo_xch_wg * 8));
        IHK(h, hipMalloc((void**)&h->duo_flags, (size_t)(h->ldT / (16 * RT)) * 2 * sizeof(int)));
        IHK(h, hipMemsetAsync(h->duo_flags, 0, (size_t)(h->ldT / (16 * RT)) * 2 * sizeof(int), h->st));
    }
    if (duo && h->duo_epoch >= (1 << 26)) {
        // flags hold epoch * 16 + swap index in 32 bits: long before that wraps (2^27 launches), drain the stream and start over at 0
        IHK(h, hipStreamSynchronize(h->st));
        IHK(h, hipMemsetAsync(h->duo_flags, 0, (size_t)(h->ldT / (16 * RT)) * 2 * sizeof(int), h->st));
        h->duo_epoch = 0;
    }
    if (strip) {
        if (!h->strip_attr) {                               // > 64 KiB of dynamic LDS needs the opt-in (once per handle = per device)
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            if constexpr (RT == 2) {
                IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, RT, 2, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, RT, 2, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, RT, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, RT, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            }
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_fwd<T, 1, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_bwd<T, 1, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            IHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_strip_tail<T, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            h->strip_attr = true;
        }
    }
    // Round 3: the narrow tail of the stack in launches of its own.  The products a pair does not split (fewer than duo_min column
    // blocks: 448 -> 256 -> 128 -> 64 -> 1 of FNN_IP_L7) are a chain of short k-loops, each a barrier, an epilogue and a cold weight
    // fetch long -- stamped at 4.5-7 us per product, 23 us forward and 29 us backward for 4 % of the FLOPs, on HALF the CUs forward
    // (the second workgroup of a pair has nothing to do there).  They run as 16-example strips instead: 256 workgroups, one per CU, no
    // pairs; the wide products keep the pairs of 32-example strips.  The hand-over is the tile of layer `cut` in the F layout (a[cut]
    // forward, dl[cut - 1] backward), written block by block by the waves that computed it.
    // small products of a 16-example-strip launch whose weights go to LDS behind the tiles: each up to 64 KB, all of them within what is
    // left of 152 KB (IPNN_TAIL_LDS=0: none); returns the bytes they take, fills the element offsets
    // Measured (profiles/r03d_ipnn_tail_lds_ab.json, A/B on one box): with the 88 KB copy 0.2700 / 0.2704 ms per step against 0.2629 /
    // 0.2667 without -- the copy costs more than the k-steps it shortens (the tails are not bound by their weight fetches): OFF by default
    const int tail_w_lds = getenv("IPNN_TAIL_LDS") ? atoi(getenv("IPNN_TAIL_LDS")) : 0;
    auto lds_weights = [&](const int* Dp, int n, int* wlds) -> size_t {
        size_t used = 0;
        const size_t budget = (size_t)152 * 1024 > (size_t)2 * 16 * maxD2x(Dp, n) * sizeof(T) ? (size_t)152 * 1024 - (size_t)2 * 16 * maxD2x(Dp, n) * sizeof(T) : 0;
        for (int p = 0; p < n; ++p) {
            const size_t b = (size_t)Dp[p] * Dp[p + 1] * sizeof(T);
            if (tail_w_lds && b <= 64 * 1024 && used + b <= budget) { wlds[p] = (int)(used / sizeof(T)); used += b; }
            else wlds[p] = -1;
        }
        return used;
    };
    int cut = L;
    while (cut >= 1 && h->Dp[cut] / 64 < h->duo_min) --cut;
    const bool tsplit = strip && duo && h->tail_split && cut >= 1 && cut < L;
    int maxD2 = 0;
    for (int t = cut; t <= L + 1; ++t) maxD2 = std::max(maxD2, h->Dp[t]);
    const size_t tail_lds = (size_t)2 * 16 * maxD2 * sizeof(T);
    // training steps: the forward tail rides in the backward tail's launch (k_ip_strip_tail; IPNN_TAIL_FUSE=0: two launches)
    const bool fuse_tail = tsplit && train && h->tail_fuse && h->tail_nf == 1 && h->tail_nw == 16 && !tail_w_lds;
    StripFwdArgs<T> s2_fused{};
    if (strip) {
        IpProf ps(h, "fwd");
        StripFwdArgs<T> sa{};
        sa.a0 = (const T*)h->a[0]; sa.n = L + 1;
        for (int t = 0; t <= L + 1; ++t) sa.Dp[t] = h->Dp[t];
        for (int t = 1; t <= L + 1; ++t) sa.W[t - 1] = (const T*)h->wf[t - 1];
        for (int t = 1; t <= L; ++t)
            sa.ef[t - 1] = EpiIpFwd<T>{nullptr, h->Dp[t], (T*)h->aT[t], ldT, drop ? h->maskT[t] : nullptr, drop ? inv_keep : 1.0f, h->cfg.act,
                                       h->d[t], B};
        sa.eo = EpiIpOut<T>{train ? (T*)h->dl[L] : nullptr, h->Dp[L + 1], train ? (T*)h->dlT[L] : nullptr, ldT, train ? y : nullptr,
                            logits_out, h->loss_t, p_out, B, h->loss_mean ? 1.0f / (float)B : 1.0f};
        sa.dbg = h->stamps; sa.rot = h->strip_rot; sa.sel = getenv("IPNN_STAMP_SEL") ? atoi(getenv("IPNN_STAMP_SEL")) : -1;
        sa.has_out = 1; sa.finalF = nullptr; sa.warm = h->strip_warm;
        for (int t = 0; t < STRIP_MAXP; ++t) sa.wlds[t] = -1;
        if (!tsplit) {
            sa.duo = StripDuo{duo ? 1 : 0, h->duo_xch, h->duo_flags, ++h->duo_epoch, h->err_flag, h->duo_xch_wg, h->duo_min};
            hipLaunchKernelGGL((k_ip_strip_fwd<T, RT>), dim3(nstrips * (duo ? 2 : 1)), dim3(64 * STRIP_NW), strip_lds, h->st, sa, maxD);
        } else {
            StripFwdArgs<T> s1 = sa;                                  // products 1 .. cut: pairs of 32-example strips; a[cut] -> HBM
            s1.n = cut; s1.has_out = 0; s1.finalF = (T*)h->a[cut];
            s1.duo = StripDuo{1, h->duo_xch, h->duo_flags, ++h->duo_epoch, h->err_flag, h->duo_xch_wg, h->duo_min};
            StripFwdArgs<T> s2{};                                     // products cut + 1 .. L + 1: 16-example strips, every CU
            s2.a0 = (const T*)h->a[cut]; s2.n = L + 1 - cut;
            for (int t = cut; t <= L + 1; ++t) s2.Dp[t - cut] = h->Dp[t];
            for (int t = cut + 1; t <= L + 1; ++t) s2.W[t - cut - 1] = sa.W[t - 1];
            for (int t = cut + 1; t <= L; ++t) s2.ef[t - cut - 1] = sa.ef[t - 1];
            s2.eo = sa.eo; s2.dbg = h->stamp_tail ? h->stamps : nullptr; s2.rot = h->strip_rot; s2.sel = -1; s2.has_out = 1; s2.finalF = nullptr; s2.warm = sa.warm;
            if (h->stamp_tail) s1.dbg = nullptr;
            s2.duo = StripDuo{0, nullptr, nullptr, 0, h->err_flag, 0, h->duo_min};
            bool wide_done = false;
            if constexpr (RT == 2) {                              // half-block items in the wide launch (IPNN_WIDE=nf:nw; measured: forward 59.8 -> 56.4 us at 2:8)
                if (h->wide_nf == 2 && h->wide_nw == 12) { hipLaunchKernelGGL((k_ip_strip_fwd<T, RT, 2, 12>), dim3(nstrips * 2), dim3(64 * 12), strip_lds, h->st, s1, maxD); wide_done = true; }
                else if (h->wide_nf == 2 && h->wide_nw == 8) { hipLaunchKernelGGL((k_ip_strip_fwd<T, RT, 2, 8>), dim3(nstrips * 2), dim3(64 * 8), strip_lds, h->st, s1, maxD); wide_done = true; }
            }
            if (!wide_done)
            hipLaunchKernelGGL((k_ip_strip_fwd<T, RT>), dim3(nstrips * 2), dim3(64 * STRIP_NW), strip_lds, h->st, s1, maxD);
            const size_t wl2 = lds_weights(s2.Dp, s2.n, s2.wlds);
            if (fuse_tail) s2_fused = s2;
            else if (h->tail_nf == 1 && h->tail_nw == 16) hipLaunchKernelGGL((k_ip_strip_fwd<T, 1, 1, 16>), dim3(Ba / 16), dim3(64 * 16), tail_lds + wl2, h->st, s2, maxD2);
            else if (h->tail_nf == 1) hipLaunchKernelGGL((k_ip_strip_fwd<T, 1, 1>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wl2, h->st, s2, maxD2);
            else if (h->tail_nf == 2) hipLaunchKernelGGL((k_ip_strip_fwd<T, 1, 2>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wl2, h->st, s2, maxD2);
            else hipLaunchKernelGGL((k_ip_strip_fwd<T, 1>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wl2, h->st, s2, maxD2);
        }
    } else {
    IpProf ps(h, "fwd");
    for (int t = 1; t <= L; ++t) {       // l_t = a_{t-1} W_t ; a_t = drop(act(l_t))
        EpiIpFwd<T> e{(T*)h->a[t], h->Dp[t], (T*)h->aT[t], ldT, drop ? h->maskT[t] : nullptr, drop ? inv_keep : 1.0f, h->cfg.act,
                      h->d[t], B};
        gemm((const T*)h->a[t - 1], (const T*)h->wf[t - 1], Ba, h->Dp[t], h->Dp[t - 1] / KS, h->Dp[t - 1] / KS, 1, e);
    }
    {
        EpiIpOut<T> e{train ? (T*)h->dl[L] : nullptr, h->Dp[L + 1], train ? (T*)h->dlT[L] : nullptr, ldT, train ? y : nullptr,
                      logits_out, h->loss_t, p_out, B, h->loss_mean ? 1.0f / (float)B : 1.0f};
        gemm((const T*)h->a[L], (const T*)h->wf[L], Ba, h->Dp[L + 1], h->Dp[L] / KS, h->Dp[L] / KS, 1, e);
    }
    }
    if (!train) { IHK(h, hipGetLastError()); return FNN_OK; }
    if (strip) {
        IpProf ps(h, "bwd");
        StripBwdArgs<T> sb{};
        sb.dlast = (const T*)h->dl[L]; sb.n = L + 1;
        for (int t = 0; t <= L + 1; ++t) sb.Dp[t] = h->Dp[t];
        for (int t = 1; t <= L + 1; ++t) {
            const bool first = (t == 1);
            sb.W[t - 1] = (const T*)h->wb[t - 1];
            sb.eb[t - 1] = EpiIpBwd<T>{nullptr, h->Dp[t - 1], first ? nullptr : (T*)h->dlT[t - 2], ldT, first ? h->dz0 : nullptr, h->Dp[0],
                                       (const T*)h->aT[t - 1], drop ? h->maskT[t - 1] : nullptr, inv_keep, keep, h->cfg.act, h->d[t - 1], B,
                                       first ? h->ref0 : nullptr};
        }
        sb.dbg = h->stamps ? h->stamps + (size_t)(h->ldT / 16) * 16 : nullptr; sb.rot = h->strip_rot;
        sb.bottom = 1; sb.finalF = nullptr; sb.warm = h->strip_warm;
        for (int t = 0; t < STRIP_MAXP; ++t) sb.wlds[t] = -1;
        if (!tsplit) {
            sb.duo = StripDuo{duo ? 1 : 0, h->duo_xch, h->duo_flags, ++h->duo_epoch, h->err_flag, h->duo_xch_wg, h->duo_min};
            hipLaunchKernelGGL((k_ip_strip_bwd<T, RT>), dim3(nstrips * (duo ? 2 : 1)), dim3(64 * STRIP_NW), strip_lds, h->st, sb, maxD);
        } else {
            StripBwdArgs<T> sA{};                                     // products L + 1 .. cut + 1 (the narrow ones come first): 16-example strips
            sA.dlast = (const T*)h->dl[L]; sA.n = L + 1 - cut;
            for (int t = cut; t <= L + 1; ++t) sA.Dp[t - cut] = h->Dp[t];
            for (int t = cut + 1; t <= L + 1; ++t) { sA.W[t - cut - 1] = sb.W[t - 1]; sA.eb[t - cut - 1] = sb.eb[t - 1]; }
            sA.dbg = h->stamp_tail ? sb.dbg : nullptr; sA.rot = h->strip_rot; sA.bottom = 0; sA.finalF = (T*)h->dl[cut - 1]; sA.warm = sb.warm;
            sA.duo = StripDuo{0, nullptr, nullptr, 0, h->err_flag, 0, h->duo_min};
            const size_t wlA = lds_weights(sA.Dp, sA.n, sA.wlds);
            if (fuse_tail) hipLaunchKernelGGL((k_ip_strip_tail<T, 1, 16>), dim3(Ba / 16), dim3(64 * 16), tail_lds, h->st, s2_fused, sA, maxD2);
            else if (h->tail_nf == 1 && h->tail_nw == 16) hipLaunchKernelGGL((k_ip_strip_bwd<T, 1, 1, 16>), dim3(Ba / 16), dim3(64 * 16), tail_lds + wlA, h->st, sA, maxD2);
            else if (h->tail_nf == 1) hipLaunchKernelGGL((k_ip_strip_bwd<T, 1, 1>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wlA, h->st, sA, maxD2);
            else if (h->tail_nf == 2) hipLaunchKernelGGL((k_ip_strip_bwd<T, 1, 2>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wlA, h->st, sA, maxD2);
            else hipLaunchKernelGGL((k_ip_strip_bwd<T, 1>), dim3(Ba / 16), dim3(64 * STRIP_NW), tail_lds + wlA, h->st, sA, maxD2);
            StripBwdArgs<T> sB = sb;                                  // products cut .. 1: pairs of 32-example strips, from delta l_cut in HBM
            sB.dlast = (const T*)h->dl[cut - 1]; sB.n = cut;
            if (h->stamp_tail) sB.dbg = nullptr;
            sB.duo = StripDuo{1, h->duo_xch, h->duo_flags, ++h->duo_epoch, h->err_flag, h->duo_xch_wg, h->duo_min};
            bool wide_done = false;
            if constexpr (RT == 2) {
                if (h->wide_nf == 2 && h->wide_nw == 12) { hipLaunchKernelGGL((k_ip_strip_bwd<T, RT, 2, 12>), dim3(nstrips * 2), dim3(64 * 12), strip_lds, h->st, sB, maxD); wide_done = true; }
                else if (h->wide_nf == 2 && h->wide_nw == 8) { hipLaunchKernelGGL((k_ip_strip_bwd<T, RT, 2, 8>), dim3(nstrips * 2), dim3(64 * 8), strip_lds, h->st, sB, maxD); wide_done = true; }
            }
            if (!wide_done)
            hipLaunchKernelGGL((k_ip_strip_bwd<T, RT>), dim3(nstrips * 2), dim3(64 * STRIP_NW), strip_lds, h->st, sB, maxD);
        }
    } else {
    IpProf ps(h, "bwd");
    for (int t = L + 1; t >= 1; --t) {   // delta l_{t-1} from delta l_t ; then gW_t = a_{t-1}^T delta l_t
        const bool first = (t == 1);
        EpiIpBwd<T> e{first ? nullptr : (T*)h->dl[t - 2], h->Dp[t - 1], first ? nullptr : (T*)h->dlT[t - 2], ldT,
                      first ? h->dz0 : nullptr, h->Dp[0], (const T*)h->aT[t - 1], drop ? h->maskT[t - 1] : nullptr, inv_keep, keep,
                      h->cfg.act, h->d[t - 1], B, first ? h->ref0 : nullptr};
        gemm((const T*)h->dl[t - 1], (const T*)h->wb[t - 1], Ba, h->Dp[t - 1], h->Dp[t] / KS, h->Dp[t] / KS, 1, e);
    }
    }
    // Beside the weight gradients, on the side stream (after the id grouping, in stream order): the backward of the inner
    // products and the sparse-row update -- they need dz1 only, the weight gradients need the transposed operands only.
    float lr_step = h->cfg.lr;
    {
        hipStream_t ss = h->st2 ? h->st2 : h->st;
        if (h->st2) { IHK(h, hipEventRecord(h->ev_bwd, h->st)); IHK(h, hipStreamWaitEvent(h->st2, h->ev_bwd, 0)); }
        if (h->adam) {
            h->adam_t += 1;
            if (!h->ftrl)
                lr_step = (float)((double)h->cfg.lr * std::sqrt(1.0 - std::pow((double)h->cfg.adam_beta2, (double)h->adam_t)) /
                                  (1.0 - std::pow((double)h->cfg.adam_beta1, (double)h->adam_t)));
        }
        {
            IpProf ps(h, "ip_bwd", ss);
            IpBwdArgs ba{h->P, ids, B, F, h->K, h->table16, h->n_rows, h->Dp[0], h->emb};
            hipLaunchKernelGGL(k_ip_bwd, dim3(Ba / 16), dim3(256), lds_ip, ss, ba, h->dz0, h->gxp, h->gb_part);
            hipLaunchKernelGGL(k_ip_b_update, dim3(1), dim3(256), 0, ss, h->b, h->gb_part, Ba / 16, h->adam ? (int)h->cfg.optimizer : 0, h->bmv,
                               lr_step, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->err_flag);
        }
        {   // sparse rows: row -= lr * sum of its gradients (c = 1: the table of powers is all ones)
            IpProf ps(h, "scatter", ss);
            // Adam / FTRL: the same sorted sums land in the (zero) gradient table instead: G[row] = 0 * 1 - (-1) * sum
            ScatArgs sa{h->rec, SORT_N, F, h->K, h->gxp, h->Dp[0], h->cpow1, h->adam ? -1.0 : (double)h->cfg.lr,
                        h->adam ? h->tG : h->table16, h->part, h->owner_cnt, h->owners, SLOT};
            hipLaunchKernelGGL(k_scat1, dim3(F * SORT_N / 256), dim3(256), 0, ss, sa);
            hipLaunchKernelGGL(k_scat2, dim3(256), dim3(256), 0, ss, sa);
        }
        if (h->adam) {
            IpProf ps(h, "adam_table", ss);
            const size_t n = (size_t)h->n_rows * SLOT;
            hipLaunchKernelGGL(k_adam_table, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ss, h->table16, h->tm, h->tv, h->tG, n,
                               lr_step, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, (int)h->cfg.optimizer);
        }
        if (h->st2) IHK(h, hipEventRecord(h->st2 && h->upd_side ? h->ev_tab : h->ev_join, h->st2));
    }
    {   // all weight gradients: gW_t [Dp_{t-1}][Dp_t] = a_{t-1}^T . delta l_t, contraction over the examples,
        // split-K slabs (the split chosen per layer so that every product fills the chip)
        IpProf ps(h, "wgrad");
        if (h->group_wgrad && L + 1 <= GEMM_GROUP_MAX) {   // ONE launch for the whole stack: 128 x 128 tiles of every product, widest first
            GemmGroupArgs g{};
            std::vector<size_t> offs(L + 2, 0);
            for (int t = 1; t <= L + 1; ++t) offs[t] = offs[t - 1] + (size_t)h->Dp[t - 1] * h->Dp[t];
            std::vector<int> order(L + 1);
            for (int t = 1; t <= L + 1; ++t) order[t - 1] = t;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return (size_t)h->Dp[x - 1] * h->Dp[x] > (size_t)h->Dp[y - 1] * h->Dp[y]; });
            int wg = 0;
            for (int i = 0; i <= L; ++i) {
                const int t = order[i], sk = h->sk[t - 1], M = h->Dp[t - 1], N = h->Dp[t];
                GemmGroupProb& pr = g.p[i];
                pr.A = h->aT[t - 1]; pr.B = h->dlT[t - 1]; pr.out = h->slab + offs[t - 1]; pr.mt16 = M / 16; pr.nt16 = N / 16;
                pr.nkt_all = ldT / KS; pr.nkt = Ba / KS / sk; pr.ldo = N; pr.gx = (M + 127) / 128; pr.gy = (N + 127) / 128;
                g.wg0[i] = wg; wg += pr.gx * pr.gy * sk;
            }
            g.wg0[L + 1] = wg; g.n = L + 1; g.zstride = h->slab_stride; g.xcd = h->group_xcd; g.wt = h->wt;
            hipLaunchKernelGGL((k_gemm_group<T, 4, 4>), dim3(wg), dim3(256), gemm_f32w_lds(), h->st, g);
        } else {
        size_t off = 0;
        for (int t = 1; t <= L + 1; ++t) {
            const int sk = h->sk[t - 1];
            EpiF32 e{h->slab + off, h->Dp[t], h->slab_stride};
            gemm((const T*)h->aT[t - 1], (const T*)h->dlT[t - 1], h->Dp[t - 1], h->Dp[t], ldT / KS, Ba / KS / sk, sk, e);
            off += (size_t)h->Dp[t - 1] * h->Dp[t];
        }
        }
    }
    const bool side_upd = h->st2 && h->upd_side;
    if (side_upd) { IHK(h, hipEventRecord(h->ev_wg, h->st)); IHK(h, hipStreamWaitEvent(h->st2, h->ev_wg, 0)); }
    {
        hipStream_t us = side_upd ? h->st2 : h->st;
        IpProf ps(h, "update", us);
        size_t off = 0;
        IpUpdArgs u{};
        u.wt = h->wt;
        for (int t = 1; t <= L + 1; ++t) {
            u.W[t - 1] = h->W[t - 1]; u.wf[t - 1] = h->wf[t - 1]; u.wb[t - 1] = h->wb[t - 1]; u.off[t - 1] = off;
            u.Din[t - 1] = h->Dp[t - 1]; u.Dout[t - 1] = h->Dp[t]; u.sk[t - 1] = h->sk[t - 1];
            off += (size_t)h->Dp[t - 1] * h->Dp[t];
        }
        u.n = L + 1; u.off[L + 1] = off; u.slab = h->slab; u.zstride = h->slab_stride; u.lr = lr_step;
        u.adam = h->adam ? (int)h->cfg.optimizer : 0; u.beta1 = h->cfg.adam_beta1; u.beta2 = h->cfg.adam_beta2; u.eps = h->cfg.adam_eps; u.bmv = h->bmv;
        if (h->adam) for (int t = 0; t <= L; ++t) { u.Wm[t] = h->Wm[t]; u.Wv[t] = h->Wv[t]; }
        u.b = h->b; u.gb_part = h->gb_part; u.ngb = Ba / 16; u.loss_t = h->loss_t; u.Ba = Ba; u.loss_sum = h->loss_dev; u.err = h->err_flag;
        hipLaunchKernelGGL((k_ip_update_all<T>), dim3((unsigned)(off / 4096 + 1)), dim3(256), 0, us, u);
    }
    if (side_upd) { IHK(h, hipEventRecord(h->ev_join, h->st2)); h->tab_pending = h->upd_pending = true; }   // joined by the next call (ip_join)
    else if (h->st2) IHK(h, hipStreamWaitEvent(h->st, h->ev_join, 0));      // the step ends when the side chain has
    IHK(h, hipGetLastError());
    return FNN_OK;
}

}  // namespace

extern "C" {

const char* ipnn_last_error(const ipnn_handle* h) { return h ? h->err.c_str() : g_ip_err.c_str(); }

uint64_t ipnn_cfg_size(void) { return (uint64_t)sizeof(ipnn_cfg); }

int ipnn_create(const ipnn_cfg* cfg, ipnn_handle** out)
{
    if (!cfg || !out) { g_ip_err = "null argument"; return FNN_ERR_ARG; }
    *out = nullptr;
    if (cfg->n_fields < 2 || cfg->n_fields > 32 || cfg->k < 1 || cfg->k > 16 || cfg->n_hidden < 1 || cfg->n_hidden > IPNN_MAX_HIDDEN ||
        cfg->max_batch < 1 || cfg->max_batch > 4096 || !(cfg->keep_prob > 0.f && cfg->keep_prob <= 1.f)) {
        g_ip_err = "bad shape (2..32 fields, k <= 16, 1..8 hidden layers, batch <= 4096, 0 < keep_prob <= 1)"; return FNN_ERR_ARG; }
    if (cfg->act != A_TANH && cfg->act != A_SIG && cfg->act != A_RELU) { g_ip_err = "bad act"; return FNN_ERR_ARG; }
    if (cfg->precision != FNN_PREC_F32 && cfg->precision != FNN_PREC_BF16) { g_ip_err = "bad precision (FNN_PREC_F32 or FNN_PREC_BF16; FNN_PREC_BF16X3 is the FNN / SNN engine's)"; return FNN_ERR_ARG; }
    if (cfg->optimizer != IPNN_OPT_SGD && cfg->optimizer != IPNN_OPT_ADAM && cfg->optimizer != IPNN_OPT_FTRL) { g_ip_err = "bad optimizer"; return FNN_ERR_ARG; }
    if (cfg->optimizer == IPNN_OPT_ADAM && !(cfg->adam_eps > 0.f)) { g_ip_err = "Adam needs adam_eps > 0"; return FNN_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_ip_err = "no HIP device (libfnn_hip.so has no CPU fallback)"; return FNN_ERR_HIP; }
    ipnn_handle* h = new ipnn_handle();
    h->cfg = *cfg; h->dev = cfg->device; h->F = cfg->n_fields; h->K = cfg->k; h->L = cfg->n_hidden;
    h->P = cfg->pairs ? h->F * (h->F - 1) / 2 : 0; h->CB = h->F * SLOT + h->P; h->bf16 = cfg->precision == FNN_PREC_BF16;
    h->Bmax = cfg->max_batch; h->ldT = rup(h->Bmax, 256);
    if (const char* e = getenv("IPNN_GEMM_LDS")) h->gemm_lds = atoi(e) != 0;
    if (const char* e = getenv("IPNN_STRIP")) h->strip = atoi(e) != 0;
    if (const char* e = getenv("IPNN_STRIP_DUO")) h->duo = atoi(e);
    if (const char* e = getenv("IPNN_DUO_MIN")) h->duo_min = std::max(2, atoi(e));
    if (const char* e = getenv("IPNN_TAIL_SPLIT")) h->tail_split = atoi(e);
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) h->n_cu = ncu; }
    if (const char* e = getenv("IPNN_STRIP_ROT")) h->strip_rot = atoi(e);
    if (const char* e = getenv("IPNN_STRIP_WARM")) h->strip_warm = atoi(e);
    if (const char* e = getenv("IPNN_UPDATE_SIDE")) h->upd_side = atoi(e) != 0;
    if (const char* e = getenv("IPNN_IPF_NT")) h->ipf_nt = atoi(e) == 1024 ? 1024 : 512;
    if (const char* e = getenv("IPNN_TAIL_FUSE")) h->tail_fuse = atoi(e) != 0;
    if (const char* e = getenv("IPNN_WT")) h->wt = atoi(e) != 0;
    if (const char* e = getenv("IPNN_WIDE")) { int nf = 4, nw = 8; if (sscanf(e, "%d:%d", &nf, &nw) == 2 && ((nf == 2 && (nw == 8 || nw == 12)) || (nf == 4 && nw == 8))) { h->wide_nf = nf; h->wide_nw = nw; } }
    if (const char* e = getenv("IPNN_TAIL_NW")) h->tail_nw = atoi(e) == 16 ? 16 : 8;
    if (const char* e = getenv("IPNN_TAIL_NF")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) h->tail_nf = v; }
    if (const char* e = getenv("IPNN_GROUP_XCD")) h->group_xcd = atoi(e);
    if (const char* e = getenv("IPNN_MASK_SIDE")) h->mask_side = atoi(e);
    if (const char* e = getenv("IPNN_FWD_SKIP")) h->fwd_skip = atoi(e);
    const char* side = getenv("IPNN_SIDE_STREAM");
    if (const char* e = getenv("IPNN_GROUP_WGRAD")) h->group_wgrad = atoi(e) != 0;
    auto fail = [&](int code) { g_ip_err = h->err; ipnn_destroy(h); return code; };
#define IK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e2_); return fail(FNN_ERR_HIP); } } while (0)
    IK(hipSetDevice(h->dev));
    if (cfg->stream) h->st = (hipStream_t)cfg->stream; else { IK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    if (!side || atoi(side) != 0) {
        IK(hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
        IK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)); IK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        IK(hipEventCreateWithFlags(&h->ev_mask, hipEventDisableTiming));
        IK(hipEventCreateWithFlags(&h->ev_bwd, hipEventDisableTiming));
        IK(hipEventCreateWithFlags(&h->ev_tab, hipEventDisableTiming)); IK(hipEventCreateWithFlags(&h->ev_wg, hipEventDisableTiming));
    }
    h->d.resize(h->L + 2); h->Dp.resize(h->L + 2);
    h->d[0] = h->F * h->K + h->P + 1; h->Dp[0] = rup(h->CB + 2, 64);
    for (int t = 1; t <= h->L; ++t) { h->d[t] = cfg->hidden[t - 1]; h->Dp[t] = rup(h->d[t] + 1, 64); if (h->d[t] < 1 || h->d[t] > 4095) { h->err = "hidden size out of range"; return fail(FNN_ERR_ARG); } }
    h->d[h->L + 1] = 1; h->Dp[h->L + 1] = 64;
    h->sk.resize(h->L + 1);
    for (int t = 1; t <= h->L + 1; ++t) {                          // ~256 workgroups of 128 x 128 per product
        const int tiles = ((h->Dp[t - 1] + 127) / 128) * ((h->Dp[t] + 127) / 128);
        int sk = 1;
        // one launch per product wants ~256 workgroups each; the grouped launch runs all products side by side and is
        // served best by half of that (measured: 0.353 -> 0.345 ms/step; a quarter: 0.355)
        int want = h->group_wgrad ? 144 : 288;
        if (const char* e = getenv("IPNN_WGRAD_WANT")) want = std::max(16, atoi(e));       // tuning knob
        while (sk < h->splitk && tiles * sk * 2 <= want) sk *= 2;
        h->sk[t - 1] = sk;
    }
    auto al = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, h->st); return e; };
    const size_t Ba = h->ldT, tsz = ts(h);
    size_t nw = 0;
    h->W.assign(h->L + 1, nullptr); h->wf.assign(h->L + 1, nullptr); h->wb.assign(h->L + 1, nullptr);
    h->a.assign(h->L + 1, nullptr); h->aT.assign(h->L + 1, nullptr); h->dl.assign(h->L + 1, nullptr); h->dlT.assign(h->L + 1, nullptr);
    for (int t = 1; t <= h->L + 1; ++t) {
        const size_t n = (size_t)h->Dp[t - 1] * h->Dp[t]; nw += n;
        IK(al((void**)&h->W[t - 1], n * 4)); IK(al(&h->wf[t - 1], n * tsz)); IK(al(&h->wb[t - 1], n * tsz));
        IK(al(&h->dl[t - 1], Ba * h->Dp[t] * tsz)); IK(al(&h->dlT[t - 1], Ba * h->Dp[t] * tsz));
    }
    for (int t = 0; t <= h->L; ++t) { IK(al(&h->a[t], Ba * h->Dp[t] * tsz)); IK(al(&h->aT[t], Ba * h->Dp[t] * tsz)); }
    h->ftrl = cfg->optimizer == IPNN_OPT_FTRL;
    h->adam = cfg->optimizer == IPNN_OPT_ADAM || h->ftrl;          // both: per-variable state + dense table pass
    if (h->adam) {
        h->Wm.assign(h->L + 1, nullptr); h->Wv.assign(h->L + 1, nullptr);
        for (int t = 1; t <= h->L + 1; ++t) {
            const size_t n = (size_t)h->Dp[t - 1] * h->Dp[t];
            IK(al((void**)&h->Wm[t - 1], n * 4)); IK(al((void**)&h->Wv[t - 1], n * 4));
            if (h->ftrl) hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->Wm[t - 1], n, 0.1f);   // initial_accumulator_value
        }
        IK(al((void**)&h->bmv, 8));
        if (h->ftrl) hipLaunchKernelGGL(k_fill_f32, dim3(1), dim3(64), 0, h->st, h->bmv, (size_t)1, 0.1f);
    }
    h->maskT.assign(h->L + 1, nullptr);
    for (int t = 0; t <= h->L; ++t) IK(al((void**)&h->maskT[t], Ba * h->Dp[t]));
    h->slab_stride = nw;
    IK(al((void**)&h->slab, (size_t)h->splitk * nw * 4));
    IK(al((void**)&h->emb, Ba * h->F * SLOT * 4));
    IK(al((void**)&h->dz0, Ba * h->Dp[0] * 4)); IK(al((void**)&h->gxp, Ba * h->Dp[0] * 4));
    IK(al((void**)&h->gb_part, (Ba / 16) * 4)); IK(al((void**)&h->loss_t, Ba * 4)); IK(al((void**)&h->loss_dev, 4));
    IK(al((void**)&h->b, 4)); IK(al((void**)&h->err_flag, 4));
    IK(al((void**)&h->rec, (size_t)h->F * SORT_N * sizeof(int4))); IK(al((void**)&h->part, (size_t)h->F * (SORT_N / 16) * 2 * SLOT * 8));
    IK(al((void**)&h->owners, (size_t)h->F * (SORT_N / 16) * sizeof(int4))); IK(al((void**)&h->owner_cnt, 4));
    IK(al(&h->skeys, (size_t)h->F * SORT_N * 8));
    {   // c = 1: every power is 1
        std::vector<double> ones(SORT_N + 1, 1.0);
        IK(hipMalloc((void**)&h->cpow1, ones.size() * 8));
        IK(hipMemcpy(h->cpow1, ones.data(), ones.size() * 8, hipMemcpyHostToDevice));
        std::vector<int> ref(h->Dp[0], -1);                       // slot column -> reference z1 column
        for (int f = 0; f < h->F; ++f) for (int l = 0; l < h->K; ++l) ref[f * SLOT + l] = f * h->K + l;
        for (int n = 0; n < h->P; ++n) ref[h->F * SLOT + n] = h->F * h->K + n;
        ref[h->CB] = h->d[0] - 1;
        IK(hipMalloc((void**)&h->ref0, ref.size() * 4));
        IK(hipMemcpy(h->ref0, ref.data(), ref.size() * 4, hipMemcpyHostToDevice));
    }
    if (getenv("IPNN_STAMPS")) { IK(al((void**)&h->stamps, (size_t)2 * (h->ldT / 16) * 16 * 8)); h->stamp_tail = atoi(getenv("IPNN_STAMPS")) == 2; }
    IK(hipStreamSynchronize(h->st));
#undef IK
    *out = h;
    return FNN_OK;
}

int ipnn_destroy(ipnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    hipSetDevice(h->dev);
    if (h->st) { ip_join(h); hipStreamSynchronize(h->st); }
    for (auto v : {&h->wf, &h->wb, &h->a, &h->aT, &h->dl, &h->dlT}) for (void* p : *v) if (p) hipFree(p);
    for (float* p : h->W) if (p) hipFree(p);
    for (uint8_t* p : h->maskT) if (p) hipFree(p);
    for (float* p : h->Wm) if (p) hipFree(p);
    for (float* p : h->Wv) if (p) hipFree(p);
    for (float* p : {h->tm, h->tv, h->tG, h->bmv}) if (p) hipFree(p);
    void* ptrs[] = {h->table16, h->b, h->emb, h->dz0, h->gxp, h->gb_part, h->loss_t, h->loss_dev, h->slab, h->ref0, h->err_flag, h->rec,
                    h->part, h->owners, h->owner_cnt, h->skeys, h->cpow1, h->duo_xch, h->duo_flags};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& kv : h->prof_ev) for (auto& p : kv.second) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (h->st2) { hipStreamSynchronize(h->st2); hipStreamDestroy(h->st2); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev_mask) hipEventDestroy(h->ev_mask);
    if (h->ev_bwd) hipEventDestroy(h->ev_bwd);
    if (h->ev_tab) hipEventDestroy(h->ev_tab);
    if (h->ev_wg) hipEventDestroy(h->ev_wg);
    if (h->own_stream && h->st) hipStreamDestroy(h->st);
    delete h;
    return FNN_OK;
}

int ipnn_sync(ipnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    int flag = 0;
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipMemcpyAsync(&flag, h->err_flag, 4, hipMemcpyDeviceToHost, h->st));
    IHK(h, hipStreamSynchronize(h->st));
    if (flag) {
        IHK(h, hipMemsetAsync(h->err_flag, 0, 4, h->st));
        if (flag & 2) {
            h->duo_failed = true;
            IFAIL(h, FNN_ERR_HIP, "a strip workgroup gave up waiting for its partner (StripDuo swap): that step's outputs are invalid and its dense update was not applied; the handle refuses further train steps (IPNN_STRIP_DUO=0 selects one workgroup per strip)");
        }
        IFAIL(h, FNN_ERR_RANGE, "feature id outside [0, n_rows)");
    }
    return FNN_OK;
}

int ipnn_set_table(ipnn_handle* h, const float* rows, int64_t n_rows)
{
    if (!h || !rows || n_rows < 1) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    if (h->table16) hipFree(h->table16);
    IHK(h, hipMalloc((void**)&h->table16, (size_t)n_rows * SLOT * 4));
    float* tmp = nullptr;
    IHK(h, hipMalloc((void**)&tmp, (size_t)n_rows * h->K * 4));
    IHK(h, hipMemcpy(tmp, rows, (size_t)n_rows * h->K * 4, hipMemcpyHostToDevice));
    const size_t n = (size_t)n_rows * SLOT;
    hipLaunchKernelGGL(k_pack_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, tmp, n_rows, h->K, SLOT, h->table16);
    IHK(h, hipStreamSynchronize(h->st));
    hipFree(tmp);
    h->n_rows = n_rows;
    h->key64 = (unsigned long long)n_rows * SORT_N > 0xFFFFFFFFull;
    if (h->adam) {                                              // fresh moments and gradient table for the new rows
        for (float** p : {&h->tm, &h->tv, &h->tG}) {
            if (*p) { hipFree(*p); *p = nullptr; }
            IHK(h, hipMalloc((void**)p, n * 4));
            IHK(h, hipMemset(*p, 0, n * 4));
        }
        if (h->ftrl) { hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->tm, n, 0.1f); IHK(h, hipStreamSynchronize(h->st)); }
        h->adam_t = 0;
    }
    return FNN_OK;
}

int ipnn_get_rows(ipnn_handle* h, const int64_t* row_ids, int64_t n, float* out)
{
    if (!h || !row_ids || !out || n < 1 || !h->table16) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    int64_t* di = nullptr; float* dout = nullptr;
    IHK(h, hipMalloc((void**)&di, n * 8)); IHK(h, hipMalloc((void**)&dout, n * h->K * 4));
    IHK(h, hipMemcpy(di, row_ids, n * 8, hipMemcpyHostToDevice));
    const size_t cnt = (size_t)n * h->K;
    hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->st, h->table16, di, n, h->n_rows, h->K, SLOT,
                       dout, h->err_flag);
    IHK(h, hipMemcpyAsync(out, dout, cnt * 4, hipMemcpyDeviceToHost, h->st));
    IHK(h, hipStreamSynchronize(h->st));
    hipFree(di); hipFree(dout);
    return FNN_OK;
}

int ipnn_set_b(ipnn_handle* h, float b)
{
    if (!h) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    IHK(h, hipMemcpy(h->b, &b, 4, hipMemcpyHostToDevice));
    return FNN_OK;
}
int ipnn_get_b(ipnn_handle* h, float* b)
{
    if (!h || !b) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    IHK(h, hipMemcpy(b, h->b, 4, hipMemcpyDeviceToHost));
    return FNN_OK;
}

// reference row r of layer `layer` -> padded row
static int ip_row_of(const ipnn_handle* h, int layer, int r)
{
    if (layer > 1) return r;
    const int FK = h->F * h->K;
    if (r < FK) return (r / h->K) * SLOT + r % h->K;
    if (r < FK + h->P) return h->F * SLOT + (r - FK);
    return h->CB;
}

int ipnn_set_layer(ipnn_handle* h, int layer, const float* W, const float* bias)
{
    if (!h || !W || !bias || layer < 1 || layer > h->L + 1) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    const int din = h->d[layer - 1], dout = h->d[layer], Din = h->Dp[layer - 1], Dout = h->Dp[layer];
    std::vector<float> p((size_t)Din * Dout, 0.f);
    for (int r = 0; r < din; ++r) memcpy(&p[(size_t)ip_row_of(h, layer, r) * Dout], &W[(size_t)r * dout], (size_t)dout * 4);
    const int ones_row = layer == 1 ? h->CB + 1 : din;
    memcpy(&p[(size_t)ones_row * Dout], bias, (size_t)dout * 4);
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    IHK(h, hipMemcpy(h->W[layer - 1], p.data(), p.size() * 4, hipMemcpyHostToDevice));
    if (h->bf16) ip_refresh<bf16_t>(h, layer, nullptr, 0.f); else ip_refresh<float>(h, layer, nullptr, 0.f);
    IHK(h, hipStreamSynchronize(h->st));
    return FNN_OK;
}

int ipnn_get_layer(ipnn_handle* h, int layer, float* W, float* bias)
{
    if (!h || !W || !bias || layer < 1 || layer > h->L + 1) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
    const int din = h->d[layer - 1], dout = h->d[layer], Din = h->Dp[layer - 1], Dout = h->Dp[layer];
    std::vector<float> p((size_t)Din * Dout);
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    IHK(h, hipMemcpy(p.data(), h->W[layer - 1], p.size() * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < din; ++r) memcpy(&W[(size_t)r * dout], &p[(size_t)ip_row_of(h, layer, r) * Dout], (size_t)dout * 4);
    const int ones_row = layer == 1 ? h->CB + 1 : din;
    memcpy(bias, &p[(size_t)ones_row * Dout], (size_t)dout * 4);
    return FNN_OK;
}

int ipnn_train_step(ipnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* const* masks,
                    float* logits_out, float* loss_sum_out)
{
    if (!h || !ids || !y) return FNN_ERR_ARG;
    if (B < 1 || B > h->Bmax) IFAIL(h, FNN_ERR_ARG, "B must be in [1, max_batch]");
    if (!h->table16) IFAIL(h, FNN_ERR_STATE, "ipnn_set_table has not been called");
    if (h->duo_failed) IFAIL(h, FNN_ERR_STATE, "an earlier step failed (StripDuo swap timed out): re-create the handle");
    IHK(h, hipSetDevice(h->dev));
    int rc = h->bf16 ? ip_run<bf16_t>(h, ids, y, B, masks, logits_out, nullptr, true)
                     : ip_run<float>(h, ids, y, B, masks, logits_out, nullptr, true);
    if (rc != FNN_OK) return rc;
    if (loss_sum_out) {                                      // (the loss is summed in the update launch: join it)
        { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
        IHK(h, hipMemcpyAsync(loss_sum_out, h->loss_dev, 4, hipMemcpyDeviceToHost, h->st));
        return ipnn_sync(h);
    }
    return FNN_OK;
}

int ipnn_set_loss_mean(ipnn_handle* h, int mean)
{
    if (!h) return FNN_ERR_ARG;
    h->loss_mean = mean != 0;
    return FNN_OK;
}

int ipnn_predict(ipnn_handle* h, const int32_t* ids, int B, float* p_out)
{
    if (!h || !ids || !p_out) return FNN_ERR_ARG;
    if (B < 1 || B > h->Bmax) IFAIL(h, FNN_ERR_ARG, "B must be in [1, max_batch]");
    if (!h->table16) IFAIL(h, FNN_ERR_STATE, "ipnn_set_table has not been called");
    IHK(h, hipSetDevice(h->dev));
    return h->bf16 ? ip_run<bf16_t>(h, ids, nullptr, B, nullptr, nullptr, p_out, false)
                   : ip_run<float>(h, ids, nullptr, B, nullptr, nullptr, p_out, false);
}

int ipnn_eval(ipnn_handle* h, const int32_t* ids, const int32_t* y, int64_t N, double* auc, double* rmse, double* logloss)
{
    if (!h || !ids || !y || N < 1) return FNN_ERR_ARG;
    if (!h->table16) IFAIL(h, FNN_ERR_STATE, "ipnn_set_table has not been called");
    IHK(h, hipSetDevice(h->dev));
    float* p_d = nullptr;
    IHK(h, hipMalloc((void**)&p_d, (size_t)N * 4));
    for (int64_t lo = 0; lo < N; lo += h->Bmax) {
        const int B = (int)(N - lo < h->Bmax ? N - lo : h->Bmax);
        const int rc = h->bf16 ? ip_run<bf16_t>(h, ids + lo * h->F, nullptr, B, nullptr, nullptr, p_d + lo, false)
                               : ip_run<float>(h, ids + lo * h->F, nullptr, B, nullptr, nullptr, p_d + lo, false);
        if (rc != FNN_OK) { hipFree(p_d); return rc; }
    }
    double out[4] = {0, 0, 0, 0};
    std::string merr;
    const int mrc = device_metrics(h->st, p_d, y, N, out, merr);
    hipFree(p_d);
    if (mrc == -1) IFAIL(h, FNN_ERR_HIP, merr);
    if (auc) *auc = out[0];
    if (rmse) *rmse = out[1];
    if (logloss) *logloss = out[2];
    const int rc = ipnn_sync(h);
    if (rc != FNN_OK) return rc;
    if (mrc == -2) IFAIL(h, FNN_ERR_RANGE, merr);
    return FNN_OK;
}

int ipnn_prof_enable(ipnn_handle* h, int on)
{
    if (!h) return FNN_ERR_ARG;
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    if (on) { for (auto& kv : h->prof_ev) for (auto& p : kv.second) { hipEventDestroy(p.first); hipEventDestroy(p.second); } h->prof_ev.clear(); }
    h->prof = on != 0;
    return FNN_OK;
}

int ipnn_prof_get(ipnn_handle* h, const char* which, double* avg_ms)
{
    if (!h || !which || !avg_ms) return FNN_ERR_ARG;
    { const int jrc = ip_join(h); if (jrc != FNN_OK) return jrc; }
    IHK(h, hipStreamSynchronize(h->st));
    *avg_ms = 0.0;
    if (h->stamps && (!strcmp(which, "fwd") || !strcmp(which, "bwd"))) {      // IPNN_STAMPS=1: the last step's per-layer stamps
        const bool duo = h->duo_xch != nullptr && !h->stamp_tail;  // pairs: even workgroups run the whole stack, odd ones the wide products only
        const int nwg = h->stamp_tail ? h->ldT / 16 : h->ldT / (h->bf16 ? 32 : 16) * (duo ? 2 : 1), np = h->L + 3;
        std::vector<long long> st((size_t)nwg * 16);
        IHK(h, hipMemcpy(st.data(), h->stamps + (strcmp(which, "bwd") ? 0 : (size_t)(h->ldT / 16) * 16), st.size() * 8, hipMemcpyDeviceToHost));
        for (int par = 0; par < (duo ? 2 : 1); ++par) {
            long long t0 = st[(size_t)par * 16], t1 = 0; int cnt = 0;
            for (int w = par; w < nwg; w += duo ? 2 : 1) {
                t0 = std::min(t0, st[(size_t)w * 16]); ++cnt;
                for (int i = 0; i < np; ++i) t1 = std::max(t1, st[(size_t)w * 16 + i]);
            }
            fprintf(stderr, "[ipnn stamps %s%s] %d workgroups, first start -> last stamp %lld ticks; avg ticks per phase (load, then products):", which,
                    duo ? (par ? " odd" : " even") : "", cnt, t1 - t0);
            for (int i = 1; i < np; ++i) {
                double s2 = 0; int c2 = 0;
                for (int w = par; w < nwg; w += duo ? 2 : 1)
                    if (st[(size_t)w * 16 + i] > st[(size_t)w * 16 + i - 1]) { s2 += (double)(st[(size_t)w * 16 + i] - st[(size_t)w * 16 + i - 1]); ++c2; }
                fprintf(stderr, " %.0f", c2 ? s2 / c2 : 0.0);
            }
            double sk = 0; for (int w = par; w < nwg; w += duo ? 2 : 1) sk += (double)(st[(size_t)w * 16] - t0);
            fprintf(stderr, " | avg start skew %.0f", sk / cnt);
            if (getenv("IPNN_STAMP_SEL") && !strcmp(which, "fwd")) {
                double dd[3] = {0, 0, 0};
                for (int w = par; w < nwg; w += duo ? 2 : 1) for (int i = 0; i < 3; ++i) dd[i] += (double)(st[(size_t)w * 16 + 11 + i] - st[(size_t)w * 16 + 10 + i]);
                fprintf(stderr, " | product %s, wave 0, last block: aux+product %.0f, next+prefetch %.0f, epilogue %.0f", getenv("IPNN_STAMP_SEL"), dd[0] / cnt, dd[1] / cnt, dd[2] / cnt);
            } else if (duo) {
                double acc3[3] = {0, 0, 0};
                for (int w = par; w < nwg; w += 2) for (int i = 0; i < 3; ++i) acc3[i] += (double)st[(size_t)w * 16 + 11 + i];
                fprintf(stderr, " | swaps in all: drain+barrier %.0f, flag wait %.0f, pull %.0f", acc3[0] / cnt, acc3[1] / cnt, acc3[2] / cnt);
            }
            fprintf(stderr, "\n");
        }
    }
    auto it = h->prof_ev.find(which);
    if (it == h->prof_ev.end() || it->second.empty()) return FNN_OK;
    double tot = 0.0;
    for (auto& p : it->second) { float ms = 0.f; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) tot += ms; }
    *avg_ms = tot / (double)it->second.size();
    return FNN_OK;
}

}  // extern "C"
