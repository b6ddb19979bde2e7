// ipnn_api.hip -- the inner-product FNN family (FNN_IP_L3 / L5 / L7) on gfx950: kernels + C ABI
// (include/ipnn_hip.h).  Replaces the TensorFlow graph of python/FNN_IP_L7.py:102-133 (forward),
// :82-88 (loss) and its gradient step (plain SGD).  Built from the FNN path's pieces: the
// fragment-tiled MFMA GEMM with fused epilogues (k_gemm_ft: forward, backward-data and the split-K
// weight-gradient product of every layer, 64 x 64 per wave, operands double-buffered in registers) and the
// sorted, atomics-free sparse-row update (k_sortA / k_sortB / k_scat1 / k_scat2) -- plus two kernels of its
// own for the inner-product layer.
#include <hip/hip_runtime.h>

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
};

template <typename T>
static __global__ __launch_bounds__(256) void k_ip_fwd(const IpFwdArgs a, T* __restrict__ a0, T* __restrict__ a0T)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float* se = reinterpret_cast<float*>(smem);                 // [16][F*16] raw embeddings
    float* sa = se + 16 * a.F * SLOT;                           // [16][D0p]  a0 values
    const int tid = threadIdx.x, t0 = blockIdx.x * 16, F = a.F, K = a.K, B = a.B, FS = F * SLOT;
    const int P = a.P, CB = FS + P;
    for (int e = tid; e < 16 * F * 4; e += 256) {               // gather: (example, field, quarter)
        const int q = e & 3, f = (e >> 2) % F, r = (e >> 2) / F, t = t0 + r;
        int64_t id = -1;
        if (t < B) { id = a.ids[(size_t)t * F + f]; if (id < 0 || id >= a.n_rows) { atomicOr(a.err, 1); id = -1; } }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) v = *reinterpret_cast<const float4*>(a.table16 + (size_t)id * SLOT + 4 * q);
        *reinterpret_cast<float4*>(se + r * FS + f * SLOT + 4 * q) = v;
    }
    __syncthreads();
    const float bval = *a.b;
    for (int e = tid; e < 16 * a.D0p; e += 256) {
        const int r = e / a.D0p, c = e % a.D0p, t = t0 + r;
        float z = 0.f; int ref = -1;                            // ref: column in the reference's z1 order
        if (c < FS) { const int f = c / SLOT, l = c % SLOT; if (l < K) { z = se[r * FS + c]; ref = f * K + l; } }
        else if (c < CB) {
            int n = c - FS, i = 0;                              // n-th pair (i, j), i < j, row-major
            while (n >= F - 1 - i) { n -= F - 1 - i; ++i; }
            const int j = i + 1 + n;
            float s = 0.f;
            for (int l = 0; l < K; ++l) s = fmaf(se[r * FS + i * SLOT + l], se[r * FS + j * SLOT + l], s);
            z = s; ref = F * K + (c - FS);
        } else if (c == CB) { z = bval; ref = a.d0 - 1; }
        float v = 0.f;
        if (t < B) {
            if (ref >= 0) {
                const float m = a.mask ? (float)a.mask[(size_t)t * a.d0 + ref] * a.inv_keep : 1.0f;
                v = ip_act(z, a.act) * m;
            } else if (c == CB + 1) v = 1.0f;
        }
        sa[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < 16 * a.D0p; e += 256) a0[ft_off<T>(t0 + e / a.D0p, e % a.D0p, a.D0p)] = (T)sa[e];
    for (int e = tid; e < a.D0p * 4; e += 256) {
        const int c = e >> 2, tq = e & 3;
        store4(a0T + ft_off<T>(c, t0 + 4 * tq, a.ldT), sa[(4 * tq) * a.D0p + c], sa[(4 * tq + 1) * a.D0p + c],
               sa[(4 * tq + 2) * a.D0p + c], sa[(4 * tq + 3) * a.D0p + c]);
    }
}

// Inner-product layer, backward: dz1' [Ba][D0p] f32 (already times mask/keep and act') ->
// slot-layout embedding gradients gx' [Ba][D0p] (columns 16f + l) for the sparse-row update, and the
// per-workgroup partial of db = sum_t dz1[b].
struct IpBwdArgs { int P; const int32_t* ids; int B, F, K; const float* table16; int64_t n_rows; int D0p; };

static __global__ __launch_bounds__(256) void k_ip_bwd(const IpBwdArgs a, const float* __restrict__ dz, float* __restrict__ gxp,
                                                        float* __restrict__ gb_part)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float* se = reinterpret_cast<float*>(smem);                 // [16][F*16]
    float* sd = se + 16 * a.F * SLOT;                           // [16][D0p]
    const int tid = threadIdx.x, t0 = blockIdx.x * 16, F = a.F, K = a.K, B = a.B, FS = F * SLOT;
    const int P = a.P, CB = FS + P;
    for (int e = tid; e < 16 * F * 4; e += 256) {
        const int q = e & 3, f = (e >> 2) % F, r = (e >> 2) / F, t = t0 + r;
        int64_t id = -1;
        if (t < B) { id = a.ids[(size_t)t * F + f]; if (id < 0 || id >= a.n_rows) id = -1; }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) v = *reinterpret_cast<const float4*>(a.table16 + (size_t)id * SLOT + 4 * q);
        *reinterpret_cast<float4*>(se + r * FS + f * SLOT + 4 * q) = v;
    }
    for (int e = tid; e < 16 * a.D0p; e += 256) sd[e] = dz[(size_t)(t0 + e / a.D0p) * a.D0p + e % a.D0p];
    __syncthreads();
    for (int e = tid; e < 16 * FS; e += 256) {                   // (example r, field f, slot l)
        const int r = e / FS, c = e % FS, f = c / SLOT, l = c % SLOT;
        float g = 0.f;
        if (l < K) {
            g = sd[r * a.D0p + c];
            // pair (i, j), i < j, sits at FS + i*(2F - i - 1)/2 + (j - i - 1)
            for (int j = 0; j < (P ? F : 0); ++j) {
                if (j == f) continue;
                const int i0 = f < j ? f : j, j0 = f < j ? j : f;
                const int n = i0 * (2 * F - i0 - 1) / 2 + (j0 - i0 - 1);
                g = fmaf(sd[r * a.D0p + FS + n], se[r * FS + j * SLOT + l], g);
            }
        }
        gxp[(size_t)(t0 + r) * a.D0p + c] = g;
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
template <typename T> struct EpiIpFwd {      // a_t = mask/keep * act(l_t); ones column at d
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; const uint8_t* maskT; float inv_keep; int act, d, B;
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const {
        unsigned mb = 0x01010101u;
        if (maskT && col < d) mb = *reinterpret_cast<const unsigned*>(maskT + (size_t)col * ldT + r0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = r0 + r;
            float x = 0.f;
            if (t < B) {
                if (col < d) x = ip_act(acc[r], act) * ((float)((mb >> (8 * r)) & 0xffu) * (maskT ? inv_keep : 1.0f));
                else if (col == d) x = 1.0f;
            }
            v[r] = x;
        }
    }
};
template <typename T> struct EpiIpOut {      // logits (column 0), loss, delta = sigmoid(logit) - y
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; const float* y; float* logits; float* loss_t; float* p_out; int B;
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = r0 + r;
            float x = 0.f;
            if (col == 0 && t < B) {
                const float z = acc[r], p = 1.0f / (1.0f + expf(-z));
                if (logits) logits[t] = z;
                if (p_out) p_out[t] = p;
                if (y) { x = p - y[t]; loss_t[t] = fmaxf(z, 0.f) - z * y[t] + log1pf(expf(-fabsf(z))); }
            } else if (col == 0 && loss_t) loss_t[t] = 0.f;
            v[r] = x;
        }
    }
};
template <typename T> struct EpiIpBwd {      // delta l_t = (delta l_{t+1} W^T) * mask/keep * act'(l_t)
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; float* out32; int ld32; const T* aT; const uint8_t* maskT; float inv_keep, keep; int act, d, B;
    const int* ref;          // layer 0 (out32 != null): slot column -> reference column, -1 = padding
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const {
        const bool real = ref ? ref[col] >= 0 : col < d;
        unsigned mb = 0x01010101u;
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        if (real) {
            if (maskT) mb = *reinterpret_cast<const unsigned*>(maskT + (size_t)col * ldT + r0);
            u = load4(aT + ft_off<T>(col, r0, ldT));
        }
        const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = r0 + r;
            float x = 0.f;
            if (t < B && real) {
                const float m = (float)((mb >> (8 * r)) & 0xffu);
                x = acc[r] * m * (maskT ? inv_keep : 1.0f) * ip_dact_u(uu[r] * (maskT ? keep : 1.0f), act);   // act(l_t) where m = 1
            }
            v[r] = x;
            if (out32) out32[(size_t)t * ld32 + col] = x;
        }
    }
};

// Keep-masks [B][d] uint8 (the ABI's layout, reference column order) -> [Dp][ldT] uint8, zero padded,
// for every layer in one launch; layer 0's columns are mapped to the slot layout on the way.
struct MaskTArgs {
    const uint8_t* src[IPNN_MAX_HIDDEN + 1]; uint8_t* dst[IPNN_MAX_HIDDEN + 1]; int d[IPNN_MAX_HIDDEN + 1], Dp[IPNN_MAX_HIDDEN + 1];
    int tile0[IPNN_MAX_HIDDEN + 2]; int n; const int* ref0; int B, Ba, ldT;
};
static __global__ __launch_bounds__(256) void k_mask_T(const MaskTArgs a)
{
    __shared__ uint8_t s[64][80];
    int t = 0;
#pragma unroll
    for (int q = 1; q <= IPNN_MAX_HIDDEN; ++q) t += (q < a.n && (int)blockIdx.x >= a.tile0[q]) ? 1 : 0;
    const int local = (int)blockIdx.x - a.tile0[t], ntx = a.Ba / 64;
    const int t0 = (local % ntx) * 64, c0 = (local / ntx) * 64;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int tt = i >> 6, cc = i & 63, ex = t0 + tt, c = c0 + cc;
        int sc = c < a.Dp[t] ? c : -1;
        if (t == 0) sc = sc >= 0 ? a.ref0[sc] : -1; else if (sc >= a.d[t]) sc = -1;
        s[cc][tt] = (ex < a.B && sc >= 0) ? a.src[t][(size_t)ex * a.d[t] + sc] : (uint8_t)0;
    }
    __syncthreads();
    const int cc = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (c0 + cc < a.Dp[t])
        *reinterpret_cast<uint4*>(a.dst[t] + (size_t)(c0 + cc) * a.ldT + t0 + 16 * q) = *reinterpret_cast<const uint4*>(&s[cc][16 * q]);
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

// Adam on the embedding table: TensorFlow's gradient through concat / slice is DENSE (zero for
// untouched rows), so every row's moments decay and every row moves each step -- one streaming pass
// over table, m, v and the per-row gradient sums G (which it zeroes again for the next step).
static __global__ void k_adam_table(float* __restrict__ tab, float* __restrict__ m, float* __restrict__ v, float* __restrict__ G,
                                    size_t n, float lr_t, float b1, float b2, float eps)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 w = *reinterpret_cast<float4*>(tab + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
    const float4 g = *reinterpret_cast<const float4*>(G + i);
    w.x = adam_step(w.x, g.x, mm.x, vv.x, lr_t, b1, b2, eps); w.y = adam_step(w.y, g.y, mm.y, vv.y, lr_t, b1, b2, eps);
    w.z = adam_step(w.z, g.z, mm.z, vv.z, lr_t, b1, b2, eps); w.w = adam_step(w.w, g.w, mm.w, vv.w, lr_t, b1, b2, eps);
    *reinterpret_cast<float4*>(tab + i) = w; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
    *reinterpret_cast<float4*>(G + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// One launch for the whole stack: W_t <- W_t - lr * (sum of its split-K slabs), both tiled shadows
// refreshed; the last workgroup applies the bias-of-z1 gradient and reduces the per-example losses.
struct IpUpdArgs {
    float* W[IPNN_MAX_HIDDEN + 1]; void* wf[IPNN_MAX_HIDDEN + 1]; void* wb[IPNN_MAX_HIDDEN + 1];
    size_t off[IPNN_MAX_HIDDEN + 2];                 // element offset of every layer in a slab; off[n] = total
    int Din[IPNN_MAX_HIDDEN + 1], Dout[IPNN_MAX_HIDDEN + 1], sk[IPNN_MAX_HIDDEN + 1]; int n;
    const float* slab; size_t zstride; float lr;
    float* b; const float* gb_part; int ngb; const float* loss_t; int Ba; float* loss_sum;
    // Adam (python/tf_util.py:17-20, TensorFlow's AdamOptimizer): first / second moments beside every tensor;
    // lr is then lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) of this step
    int adam; float beta1, beta2, eps; float* Wm[IPNN_MAX_HIDDEN + 1]; float* Wv[IPNN_MAX_HIDDEN + 1]; float* bmv;
};

template <typename T>
static __global__ __launch_bounds__(256) void k_ip_update_all(const IpUpdArgs u)
{
    if (blockIdx.x == gridDim.x - 1) {               // scalar tail: b and the loss, fixed-shape trees
        __shared__ float sl[256], sg[256];
        float v = 0.f, g = 0.f;
        for (int i = threadIdx.x; i < u.Ba; i += 256) v += u.loss_t[i];
        for (int i = threadIdx.x; i < u.ngb; i += 256) g += u.gb_part[i];
        sl[threadIdx.x] = v; sg[threadIdx.x] = g; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sg[threadIdx.x] += sg[threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            *u.loss_sum = sl[0];
            if (u.adam) *u.b = adam_step(*u.b, sg[0], u.bmv[0], u.bmv[1], u.lr, u.beta1, u.beta2, u.eps);
            else *u.b -= u.lr * sg[0];
        }
        return;
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= u.off[u.n]) return;
    int t = 0;
#pragma unroll
    for (int q = 1; q <= IPNN_MAX_HIDDEN; ++q) t += (q < u.n && i >= u.off[q]) ? 1 : 0;
    const size_t j = i - u.off[t];
    float g = 0.f;
    for (int z = 0; z < u.sk[t]; ++z) g += u.slab[(size_t)z * u.zstride + i];
    float w;
    if (u.adam) {
        float m = u.Wm[t][j], v = u.Wv[t][j];
        w = adam_step(u.W[t][j], g, m, v, u.lr, u.beta1, u.beta2, u.eps);
        u.Wm[t][j] = m; u.Wv[t][j] = v;
    } else w = u.W[t][j] - u.lr * g;
    u.W[t][j] = w;
    const int r = (int)(j / u.Dout[t]), c = (int)(j % u.Dout[t]);
    static_cast<T*>(u.wf[t])[ft_off<T>(c, r, u.Din[t])] = (T)w;
    static_cast<T*>(u.wb[t])[ft_off<T>(r, c, u.Dout[t])] = (T)w;
}

}  // namespace

struct ipnn_handle {
    ipnn_cfg cfg{}; std::string err; int dev = 0; hipStream_t st = nullptr; bool own_stream = false;
    int F = 0, K = 0, L = 0, P = 0, CB = 0, Bmax = 0, ldT = 0; bool bf16 = false; int splitk = 8;   // splitk: slab capacity
    std::vector<int> sk;                         // split-K of each layer's weight-gradient product
    bool adam = false; int64_t adam_t = 0;       // Adam state: step count, moments of every tensor, dense row-gradient table
    std::vector<float*> Wm, Wv; float *tm = nullptr, *tv = nullptr, *tG = nullptr, *bmv = nullptr;
    std::vector<int> d, Dp;                      // d[0..L+1], padded
    float* table16 = nullptr; int64_t n_rows = 0; float* b = nullptr;
    std::vector<float*> W; std::vector<void*> wf, wb;           // W[t], t = 1..L+1 (index t-1)
    std::vector<void*> a, aT, dl, dlT;                           // a[t] t=0..L ; dl[t] t=1..L+1 (index t-1)
    std::vector<uint8_t*> maskT;                                 // keep-masks of a step, transposed [Dp_t][ldT], t = 0..L
    float *dz0 = nullptr, *gxp = nullptr, *gb_part = nullptr, *loss_t = nullptr, *loss_dev = nullptr, *slab = nullptr;
    int* ref0 = nullptr; int* err_flag = nullptr;
    int4* rec = nullptr; double* part = nullptr; int4* owners = nullptr; int* owner_cnt = nullptr; void* skeys = nullptr;
    double* cpow1 = nullptr; bool key64 = true;
    size_t slab_stride = 0;
    bool prof = false;                               // HIP-event timing of the step's segments
    bool gemm_lds = false;                           // IPNN_GEMM_LDS=1: LDS-staged k_gemm_lds for the wide products (measured equal to k_gemm_ft: both L2-bound)
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_ev;
};

namespace {
struct IpProf {                                      // one segment of ip_run on the handle's stream
    ipnn_handle* h; const char* name; hipEvent_t b = nullptr, e = nullptr;
    IpProf(ipnn_handle* h_, const char* n) : h(h_), name(n) {
        if (!h->prof) return;
        hipEventCreate(&b); hipEventCreate(&e); hipEventRecord(b, h->st);
    }
    ~IpProf() { if (!h->prof) return; hipEventRecord(e, h->st); h->prof_ev[name].emplace_back(b, e); }
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

template <typename T>
int ip_run(ipnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* const* masks, float* logits_out,
           float* p_out, bool train)
{
    const int Ba = rup(B, 256), L = h->L, F = h->F, ldT = h->ldT;
    const float keep = h->cfg.keep_prob, inv_keep = 1.0f / keep;
    const size_t lds_ip = (size_t)16 * (F * SLOT + h->Dp[0]) * sizeof(float);
    if (train) {
        IpProf ps(h, "sort");
        SortArgs so{ids, B, F, h->n_rows, h->rec, h->owner_cnt, F, h->skeys};
        if (h->key64) {
            hipLaunchKernelGGL((k_sortA<unsigned long long>), dim3(4 * F), dim3(256), 0, h->st, so);
            hipLaunchKernelGGL((k_sortB<unsigned long long>), dim3(16 * F), dim3(256), SORT_N * 8, h->st, so);
        } else {
            hipLaunchKernelGGL((k_sortA<unsigned>), dim3(4 * F), dim3(256), 0, h->st, so);
            hipLaunchKernelGGL((k_sortB<unsigned>), dim3(16 * F), dim3(256), SORT_N * 4, h->st, so);
        }
    }
    {
        IpProf ps(h, "ip_fwd");
        IpFwdArgs fa{h->P, ids, B, F, h->K, h->table16, h->n_rows, h->b, (train && masks) ? masks[0] : nullptr, h->d[0],
                     (train && masks) ? inv_keep : 1.0f, h->cfg.act, h->Dp[0], ldT, h->err_flag};
        hipLaunchKernelGGL((k_ip_fwd<T>), dim3(Ba / 16), dim3(256), lds_ip, h->st, fa, (T*)h->a[0], (T*)h->aT[0]);
    }
    // one product: C [M][N] = A . B^T on fragment-tiled operands; narrow problems take smaller wave tiles
    auto gemm = [&](const T* A, const T* Bm, int M, int N, int nkt_all, int nkt, int splitk, auto epi) {
        typedef decltype(epi) E;
        const int mt16 = M / 16, nt16 = N / 16;
        const long wg44 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
        const size_t lds4 = E::TILE ? gemm_ft_lds<T, 4>() : 0, lds2 = E::TILE ? gemm_ft_lds<T, 2>() : 0;
        if (wg44 >= 160 && h->gemm_lds) {
            constexpr size_t ldsb = gemm_lds_bytes<T, E::TILE>();
            static bool attr_set = false;                  // per instantiation (T, E): > 64 KiB of dynamic LDS needs the opt-in
            if (!attr_set) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_lds<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
                attr_set = true;
            }
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
    const bool drop = train && masks;
    if (drop) {   // keep-masks of all layers -> transposed, padded, slot-ordered for layer 0
        MaskTArgs ma{};
        int tiles = 0;
        for (int t = 0; t <= L; ++t) {
            if (!masks[t]) IFAIL(h, FNN_ERR_ARG, "masks: null entry");
            ma.src[t] = masks[t]; ma.dst[t] = h->maskT[t]; ma.d[t] = h->d[t]; ma.Dp[t] = h->Dp[t]; ma.tile0[t] = tiles;
            tiles += (Ba / 64) * (h->Dp[t] / 64);
        }
        ma.tile0[L + 1] = tiles; ma.n = L + 1; ma.ref0 = h->ref0; ma.B = B; ma.Ba = Ba; ma.ldT = ldT;
        hipLaunchKernelGGL(k_mask_T, dim3(tiles), dim3(256), 0, h->st, ma);
    }
    constexpr int KS = Traits<T>::KS;
    {
    IpProf ps(h, "fwd");
    for (int t = 1; t <= L; ++t) {       // l_t = a_{t-1} W_t ; a_t = drop(act(l_t))
        EpiIpFwd<T> e{(T*)h->a[t], h->Dp[t], (T*)h->aT[t], ldT, drop ? h->maskT[t] : nullptr, drop ? inv_keep : 1.0f, h->cfg.act,
                      h->d[t], B};
        gemm((const T*)h->a[t - 1], (const T*)h->wf[t - 1], Ba, h->Dp[t], h->Dp[t - 1] / KS, h->Dp[t - 1] / KS, 1, e);
    }
    {
        EpiIpOut<T> e{train ? (T*)h->dl[L] : nullptr, h->Dp[L + 1], train ? (T*)h->dlT[L] : nullptr, ldT, train ? y : nullptr,
                      logits_out, h->loss_t, p_out, B};
        gemm((const T*)h->a[L], (const T*)h->wf[L], Ba, h->Dp[L + 1], h->Dp[L] / KS, h->Dp[L] / KS, 1, e);
    }
    }
    if (!train) { IHK(h, hipGetLastError()); return FNN_OK; }
    {
    IpProf ps(h, "bwd");
    for (int t = L + 1; t >= 1; --t) {   // delta l_{t-1} from delta l_t ; then gW_t = a_{t-1}^T delta l_t
        const bool first = (t == 1);
        EpiIpBwd<T> e{first ? nullptr : (T*)h->dl[t - 2], h->Dp[t - 1], first ? nullptr : (T*)h->dlT[t - 2], ldT,
                      first ? h->dz0 : nullptr, h->Dp[0], (const T*)h->aT[t - 1], drop ? h->maskT[t - 1] : nullptr, inv_keep, keep,
                      h->cfg.act, h->d[t - 1], B, first ? h->ref0 : nullptr};
        gemm((const T*)h->dl[t - 1], (const T*)h->wb[t - 1], Ba, h->Dp[t - 1], h->Dp[t] / KS, h->Dp[t] / KS, 1, e);
    }
    }
    {   // all weight gradients: gW_t [Dp_{t-1}][Dp_t] = a_{t-1}^T . delta l_t, contraction over the examples,
        // split-K slabs (the split chosen per layer so that every product fills the chip)
        IpProf ps(h, "wgrad");
        size_t off = 0;
        for (int t = 1; t <= L + 1; ++t) {
            const int sk = h->sk[t - 1];
            EpiF32 e{h->slab + off, h->Dp[t], h->slab_stride};
            gemm((const T*)h->aT[t - 1], (const T*)h->dlT[t - 1], h->Dp[t - 1], h->Dp[t], ldT / KS, Ba / KS / sk, sk, e);
            off += (size_t)h->Dp[t - 1] * h->Dp[t];
        }
    }
    {
        IpProf ps(h, "ip_bwd");
        IpBwdArgs ba{h->P, ids, B, F, h->K, h->table16, h->n_rows, h->Dp[0]};
        hipLaunchKernelGGL(k_ip_bwd, dim3(Ba / 16), dim3(256), lds_ip, h->st, ba, h->dz0, h->gxp, h->gb_part);
    }
    {   // sparse rows: row -= lr * sum of its gradients (c = 1: the table of powers is all ones)
        IpProf ps(h, "scatter");
        // Adam: the same sorted sums land in the (zero) gradient table instead: G[row] = 0 * 1 - (-1) * sum
        ScatArgs sa{h->rec, SORT_N, F, h->K, h->gxp, h->Dp[0], h->cpow1, h->adam ? -1.0 : (double)h->cfg.lr,
                    h->adam ? h->tG : h->table16, h->part, h->owner_cnt, h->owners, SLOT};
        hipLaunchKernelGGL(k_scat1, dim3(F * SORT_N / 256), dim3(256), 0, h->st, sa);
        hipLaunchKernelGGL(k_scat2, dim3(256), dim3(256), 0, h->st, sa);
    }
    float lr_step = h->cfg.lr;
    if (h->adam) {
        h->adam_t += 1;
        lr_step = (float)((double)h->cfg.lr * std::sqrt(1.0 - std::pow((double)h->cfg.adam_beta2, (double)h->adam_t)) /
                          (1.0 - std::pow((double)h->cfg.adam_beta1, (double)h->adam_t)));
        IpProf ps(h, "adam_table");
        const size_t n = (size_t)h->n_rows * SLOT;
        hipLaunchKernelGGL(k_adam_table, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, h->st, h->table16, h->tm, h->tv, h->tG, n,
                           lr_step, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps);
    }
    {
        IpProf ps(h, "update");
        size_t off = 0;
        IpUpdArgs u{};
        for (int t = 1; t <= L + 1; ++t) {
            u.W[t - 1] = h->W[t - 1]; u.wf[t - 1] = h->wf[t - 1]; u.wb[t - 1] = h->wb[t - 1]; u.off[t - 1] = off;
            u.Din[t - 1] = h->Dp[t - 1]; u.Dout[t - 1] = h->Dp[t]; u.sk[t - 1] = h->sk[t - 1];
            off += (size_t)h->Dp[t - 1] * h->Dp[t];
        }
        u.n = L + 1; u.off[L + 1] = off; u.slab = h->slab; u.zstride = h->slab_stride; u.lr = lr_step;
        u.adam = h->adam ? 1 : 0; u.beta1 = h->cfg.adam_beta1; u.beta2 = h->cfg.adam_beta2; u.eps = h->cfg.adam_eps; u.bmv = h->bmv;
        if (h->adam) for (int t = 0; t <= L; ++t) { u.Wm[t] = h->Wm[t]; u.Wv[t] = h->Wv[t]; }
        u.b = h->b; u.gb_part = h->gb_part; u.ngb = Ba / 16; u.loss_t = h->loss_t; u.Ba = Ba; u.loss_sum = h->loss_dev;
        hipLaunchKernelGGL((k_ip_update_all<T>), dim3((unsigned)((off + 255) / 256 + 1)), dim3(256), 0, h->st, u);
    }
    IHK(h, hipGetLastError());
    return FNN_OK;
}

}  // namespace

extern "C" {

const char* ipnn_last_error(const ipnn_handle* h) { return h ? h->err.c_str() : g_ip_err.c_str(); }

int ipnn_create(const ipnn_cfg* cfg, ipnn_handle** out)
{
    if (!cfg || !out) { g_ip_err = "null argument"; return FNN_ERR_ARG; }
    *out = nullptr;
    if (cfg->n_fields < 2 || cfg->n_fields > 32 || cfg->k < 1 || cfg->k > 16 || cfg->n_hidden < 1 || cfg->n_hidden > IPNN_MAX_HIDDEN ||
        cfg->max_batch < 1 || cfg->max_batch > 4096 || !(cfg->keep_prob > 0.f && cfg->keep_prob <= 1.f)) {
        g_ip_err = "bad shape (2..32 fields, k <= 16, 1..8 hidden layers, batch <= 4096, 0 < keep_prob <= 1)"; return FNN_ERR_ARG; }
    if (cfg->act != A_TANH && cfg->act != A_SIG && cfg->act != A_RELU) { g_ip_err = "bad act"; return FNN_ERR_ARG; }
    if (cfg->optimizer != IPNN_OPT_SGD && cfg->optimizer != IPNN_OPT_ADAM) { g_ip_err = "bad optimizer"; return FNN_ERR_ARG; }
    if (cfg->optimizer == IPNN_OPT_ADAM && !(cfg->adam_eps > 0.f)) { g_ip_err = "Adam needs adam_eps > 0"; return FNN_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_ip_err = "no HIP device (libfnn_hip.so has no CPU fallback)"; return FNN_ERR_HIP; }
    ipnn_handle* h = new ipnn_handle();
    h->cfg = *cfg; h->dev = cfg->device; h->F = cfg->n_fields; h->K = cfg->k; h->L = cfg->n_hidden;
    h->P = cfg->pairs ? h->F * (h->F - 1) / 2 : 0; h->CB = h->F * SLOT + h->P; h->bf16 = cfg->precision == FNN_PREC_BF16;
    h->Bmax = cfg->max_batch; h->ldT = rup(h->Bmax, 256);
    if (const char* e = getenv("IPNN_GEMM_LDS")) h->gemm_lds = atoi(e) != 0;
    auto fail = [&](int code) { g_ip_err = h->err; ipnn_destroy(h); return code; };
#define IK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e2_); return fail(FNN_ERR_HIP); } } while (0)
    IK(hipSetDevice(h->dev));
    if (cfg->stream) h->st = (hipStream_t)cfg->stream; else { IK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    h->d.resize(h->L + 2); h->Dp.resize(h->L + 2);
    h->d[0] = h->F * h->K + h->P + 1; h->Dp[0] = rup(h->CB + 2, 64);
    for (int t = 1; t <= h->L; ++t) { h->d[t] = cfg->hidden[t - 1]; h->Dp[t] = rup(h->d[t] + 1, 64); if (h->d[t] < 1 || h->d[t] > 4095) { h->err = "hidden size out of range"; return fail(FNN_ERR_ARG); } }
    h->d[h->L + 1] = 1; h->Dp[h->L + 1] = 64;
    h->sk.resize(h->L + 1);
    for (int t = 1; t <= h->L + 1; ++t) {                          // ~256 workgroups of 128 x 128 per product
        const int tiles = ((h->Dp[t - 1] + 127) / 128) * ((h->Dp[t] + 127) / 128);
        int sk = 1;
        while (sk < h->splitk && tiles * sk * 2 <= 288) sk *= 2;
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
    h->adam = cfg->optimizer == IPNN_OPT_ADAM;
    if (h->adam) {
        h->Wm.assign(h->L + 1, nullptr); h->Wv.assign(h->L + 1, nullptr);
        for (int t = 1; t <= h->L + 1; ++t) {
            const size_t n = (size_t)h->Dp[t - 1] * h->Dp[t];
            IK(al((void**)&h->Wm[t - 1], n * 4)); IK(al((void**)&h->Wv[t - 1], n * 4));
        }
        IK(al((void**)&h->bmv, 8));
    }
    h->maskT.assign(h->L + 1, nullptr);
    for (int t = 0; t <= h->L; ++t) IK(al((void**)&h->maskT[t], Ba * h->Dp[t]));
    h->slab_stride = nw;
    IK(al((void**)&h->slab, (size_t)h->splitk * nw * 4));
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
    IK(hipStreamSynchronize(h->st));
#undef IK
    *out = h;
    return FNN_OK;
}

int ipnn_destroy(ipnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    hipSetDevice(h->dev);
    if (h->st) hipStreamSynchronize(h->st);
    for (auto v : {&h->wf, &h->wb, &h->a, &h->aT, &h->dl, &h->dlT}) for (void* p : *v) if (p) hipFree(p);
    for (float* p : h->W) if (p) hipFree(p);
    for (uint8_t* p : h->maskT) if (p) hipFree(p);
    for (float* p : h->Wm) if (p) hipFree(p);
    for (float* p : h->Wv) if (p) hipFree(p);
    for (float* p : {h->tm, h->tv, h->tG, h->bmv}) if (p) hipFree(p);
    void* ptrs[] = {h->table16, h->b, h->dz0, h->gxp, h->gb_part, h->loss_t, h->loss_dev, h->slab, h->ref0, h->err_flag, h->rec,
                    h->part, h->owners, h->owner_cnt, h->skeys, h->cpow1};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& kv : h->prof_ev) for (auto& p : kv.second) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (h->own_stream && h->st) hipStreamDestroy(h->st);
    delete h;
    return FNN_OK;
}

int ipnn_sync(ipnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    int flag = 0;
    IHK(h, hipMemcpyAsync(&flag, h->err_flag, 4, hipMemcpyDeviceToHost, h->st));
    IHK(h, hipStreamSynchronize(h->st));
    if (flag) { IHK(h, hipMemsetAsync(h->err_flag, 0, 4, h->st)); IFAIL(h, FNN_ERR_RANGE, "feature id outside [0, n_rows)"); }
    return FNN_OK;
}

int ipnn_set_table(ipnn_handle* h, const float* rows, int64_t n_rows)
{
    if (!h || !rows || n_rows < 1) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
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
        h->adam_t = 0;
    }
    return FNN_OK;
}

int ipnn_get_rows(ipnn_handle* h, const int64_t* row_ids, int64_t n, float* out)
{
    if (!h || !row_ids || !out || n < 1 || !h->table16) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev));
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
    IHK(h, hipSetDevice(h->dev)); IHK(h, hipStreamSynchronize(h->st));
    IHK(h, hipMemcpy(h->b, &b, 4, hipMemcpyHostToDevice));
    return FNN_OK;
}
int ipnn_get_b(ipnn_handle* h, float* b)
{
    if (!h || !b) return FNN_ERR_ARG;
    IHK(h, hipSetDevice(h->dev)); IHK(h, hipStreamSynchronize(h->st));
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
    IHK(h, hipSetDevice(h->dev));
    int rc = h->bf16 ? ip_run<bf16_t>(h, ids, y, B, masks, logits_out, nullptr, true)
                     : ip_run<float>(h, ids, y, B, masks, logits_out, nullptr, true);
    if (rc != FNN_OK) return rc;
    if (loss_sum_out) {
        IHK(h, hipMemcpyAsync(loss_sum_out, h->loss_dev, 4, hipMemcpyDeviceToHost, h->st));
        return ipnn_sync(h);
    }
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
    IHK(h, hipStreamSynchronize(h->st));
    if (on) { for (auto& kv : h->prof_ev) for (auto& p : kv.second) { hipEventDestroy(p.first); hipEventDestroy(p.second); } h->prof_ev.clear(); }
    h->prof = on != 0;
    return FNN_OK;
}

int ipnn_prof_get(ipnn_handle* h, const char* which, double* avg_ms)
{
    if (!h || !which || !avg_ms) return FNN_ERR_ARG;
    IHK(h, hipStreamSynchronize(h->st));
    *avg_ms = 0.0;
    auto it = h->prof_ev.find(which);
    if (it == h->prof_ev.end() || it->second.empty()) return FNN_OK;
    double tot = 0.0;
    for (auto& p : it->second) { float ms = 0.f; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) tot += ms; }
    *avg_ms = tot / (double)it->second.size();
    return FNN_OK;
}

}  // extern "C"
