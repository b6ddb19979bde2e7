// ipnn_api.hip -- the inner-product FNN family (FNN_IP_L3 / L5 / L7) on gfx950: kernels + C ABI
// (include/ipnn_hip.h).  Replaces the TensorFlow graph of python/FNN_IP_L7.py:102-133 (forward),
// :82-88 (loss) and its gradient step (plain SGD).  Built from the FNN path's pieces: fragment-tiled
// MFMA GEMMs with fused epilogues (k_gemm), the split-K weight-gradient kernel (k_wgrad) and the
// sorted, atomics-free sparse-row update (k_sortA / k_sortB / k_scat1 / k_scat2) -- plus two kernels of its
// own for the inner-product layer.
#include <hip/hip_runtime.h>

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
// ------------------------------------------------------------------------------------------
struct IpFwdArgs {
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
    const int P = F * (F - 1) / 2, CB = FS + P;
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
    for (int e = tid; e < 16 * a.D0p; e += 256) a0[(size_t)(t0 + e / a.D0p) * a.D0p + e % a.D0p] = (T)sa[e];
    for (int e = tid; e < a.D0p * 4; e += 256) {
        const int c = e >> 2, tq = e & 3;
        store4(a0T + ft_off<T>(c, t0 + 4 * tq, a.ldT), sa[(4 * tq) * a.D0p + c], sa[(4 * tq + 1) * a.D0p + c],
               sa[(4 * tq + 2) * a.D0p + c], sa[(4 * tq + 3) * a.D0p + c]);
    }
}

// Inner-product layer, backward: dz1' [Ba][D0p] f32 (already times mask/keep and act') ->
// slot-layout embedding gradients gx' [Ba][D0p] (columns 16f + l) for the sparse-row update, and the
// per-workgroup partial of db = sum_t dz1[b].
struct IpBwdArgs { const int32_t* ids; int B, F, K; const float* table16; int64_t n_rows; int D0p; };

static __global__ __launch_bounds__(256) void k_ip_bwd(const IpBwdArgs a, const float* __restrict__ dz, float* __restrict__ gxp,
                                                        float* __restrict__ gb_part)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float* se = reinterpret_cast<float*>(smem);                 // [16][F*16]
    float* sd = se + 16 * a.F * SLOT;                           // [16][D0p]
    const int tid = threadIdx.x, t0 = blockIdx.x * 16, F = a.F, K = a.K, B = a.B, FS = F * SLOT;
    const int P = F * (F - 1) / 2, CB = FS + P;
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
            for (int j = 0; j < F; ++j) {
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
template <typename T> struct EpiIpFwd {      // a_t = mask/keep * act(l_t); ones column at d
    T* out; int ld; T* outT; int ldT; const uint8_t* mask; float inv_keep; int act, d, B;
    __device__ void operator()(int row0, int col, const f32x4& acc, int) const {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = row0 + r;
            float x = 0.f;
            if (t < B) {
                if (col < d) x = ip_act(acc[r], act) * (mask ? (float)mask[(size_t)t * d + col] * inv_keep : 1.0f);
                else if (col == d) x = 1.0f;
            }
            v[r] = x;
            out[(size_t)t * ld + col] = (T)x;
        }
        store4(outT + ft_off<T>(col, row0, ldT), v[0], v[1], v[2], v[3]);
    }
};
template <typename T> struct EpiIpOut {      // logits (column 0), loss, delta = sigmoid(logit) - y
    T* dl; int ld; T* dlT; int ldT; const float* y; float* logits; float* loss_t; float* p_out; int B;
    __device__ void operator()(int row0, int col, const f32x4& acc, int) const {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = row0 + r;
            float x = 0.f;
            if (col == 0 && t < B) {
                const float z = acc[r], p = 1.0f / (1.0f + expf(-z));
                if (logits) logits[t] = z;
                if (p_out) p_out[t] = p;
                if (y) { x = p - y[t]; loss_t[t] = fmaxf(z, 0.f) - z * y[t] + log1pf(expf(-fabsf(z))); }
            } else if (col == 0 && loss_t) loss_t[t] = 0.f;
            v[r] = x;
            if (dl) dl[(size_t)t * ld + col] = (T)x;
        }
        if (dlT) store4(dlT + ft_off<T>(col, row0, ldT), v[0], v[1], v[2], v[3]);
    }
};
template <typename T> struct EpiIpBwd {      // delta l_t = (delta l_{t+1} W^T) * mask/keep * act'(l_t)
    T* out; int ld; T* outT; int ldT; float* out32; const T* a; const uint8_t* mask; float inv_keep, keep; int act, d, B;
    // layer 0 (out32 != null): `mask` is indexed through `ref` (slot column -> reference column)
    const int* ref; int dref;
    __device__ void operator()(int row0, int col, const f32x4& acc, int) const {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = row0 + r;
            float x = 0.f;
            const int mc = ref ? ref[col] : (col < d ? col : -1);
            if (t < B && mc >= 0) {
                const float m = mask ? (float)mask[(size_t)t * (ref ? dref : d) + mc] : 1.0f;
                const float u = (float)a[(size_t)t * ld + col] * (mask ? keep : 1.0f);      // act(l_t) where m = 1
                x = acc[r] * m * (mask ? inv_keep : 1.0f) * ip_dact_u(u, act);
            }
            v[r] = x;
            if (out32) out32[(size_t)t * ld + col] = x; else out[(size_t)t * ld + col] = (T)x;
        }
        if (outT) store4(outT + ft_off<T>(col, row0, ldT), v[0], v[1], v[2], v[3]);
    }
};

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
static __global__ void k_ip_update_b(float* b, const float* gb_part, int n, float lr, const float* loss_t, int Ba, float* loss_sum)
{
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < n; ++i) s += gb_part[i]; if (lr != 0.f) *b -= lr * s; }
    __shared__ float sl[256];
    float v = 0.f;
    for (int i = threadIdx.x; i < Ba; i += 256) v += loss_t[i];
    sl[threadIdx.x] = v; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sl[threadIdx.x] += sl[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *loss_sum = sl[0];
}

}  // namespace

struct ipnn_handle {
    ipnn_cfg cfg{}; std::string err; int dev = 0; hipStream_t st = nullptr; bool own_stream = false;
    int F = 0, K = 0, L = 0, P = 0, CB = 0, Bmax = 0, ldT = 0; bool bf16 = false; int splitk = 8;
    std::vector<int> d, Dp;                      // d[0..L+1], padded
    float* table16 = nullptr; int64_t n_rows = 0; float* b = nullptr;
    std::vector<float*> W; std::vector<void*> wf, wb;           // W[t], t = 1..L+1 (index t-1)
    std::vector<void*> a, aT, dl, dlT;                           // a[t] t=0..L ; dl[t] t=1..L+1 (index t-1)
    float *dz0 = nullptr, *gxp = nullptr, *gb_part = nullptr, *loss_t = nullptr, *loss_dev = nullptr, *slab = nullptr;
    int* ref0 = nullptr; int* err_flag = nullptr;
    int4* rec = nullptr; double* part = nullptr; int4* owners = nullptr; int* owner_cnt = nullptr; void* skeys = nullptr;
    double* cpow1 = nullptr; bool key64 = true;
    size_t slab_stride = 0;
    bool prof = false;                               // HIP-event timing of the step's segments
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
    hipLaunchKernelGGL((k_ip_update<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->W[t - 1], slab, h->splitk,
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
        IpFwdArgs fa{ids, B, F, h->K, h->table16, h->n_rows, h->b, (train && masks) ? masks[0] : nullptr, h->d[0],
                     (train && masks) ? inv_keep : 1.0f, h->cfg.act, h->Dp[0], ldT, h->err_flag};
        hipLaunchKernelGGL((k_ip_fwd<T>), dim3(Ba / 16), dim3(256), lds_ip, h->st, fa, (T*)h->a[0], (T*)h->aT[0]);
    }
    {
    IpProf ps(h, "fwd");
    for (int t = 1; t <= L; ++t) {       // l_t = a_{t-1} W_t ; a_t = drop(act(l_t))
        EpiIpFwd<T> e{(T*)h->a[t], h->Dp[t], (T*)h->aT[t], ldT, (train && masks) ? masks[t] : nullptr,
                      (train && masks) ? inv_keep : 1.0f, h->cfg.act, h->d[t], B};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiIpFwd<T>>), dim3(Ba / 64, h->Dp[t] / 64, 1), dim3(256), 0, h->st, (const T*)h->a[t - 1],
                           h->Dp[t - 1], (const T*)h->wf[t - 1], h->Dp[t - 1], e);
    }
    {
        EpiIpOut<T> e{train ? (T*)h->dl[L] : nullptr, h->Dp[L + 1], train ? (T*)h->dlT[L] : nullptr, ldT, train ? y : nullptr,
                      logits_out, h->loss_t, p_out, B};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiIpOut<T>>), dim3(Ba / 64, h->Dp[L + 1] / 64, 1), dim3(256), 0, h->st, (const T*)h->a[L],
                           h->Dp[L], (const T*)h->wf[L], h->Dp[L], e);
    }
    }
    if (!train) { IHK(h, hipGetLastError()); return FNN_OK; }
    {
    IpProf ps(h, "bwd");
    for (int t = L + 1; t >= 1; --t) {   // delta l_{t-1} from delta l_t ; then gW_t = a_{t-1}^T delta l_t
        const bool first = (t == 1);
        EpiIpBwd<T> e{first ? nullptr : (T*)h->dl[t - 2], h->Dp[t - 1], first ? nullptr : (T*)h->dlT[t - 2], ldT,
                      first ? h->dz0 : nullptr, (const T*)h->a[t - 1], masks ? masks[t - 1] : nullptr, inv_keep, keep,
                      h->cfg.act, h->d[t - 1], B, first ? h->ref0 : nullptr, h->d[0]};
        hipLaunchKernelGGL((k_gemm<T, 4, EpiIpBwd<T>>), dim3(Ba / 64, h->Dp[t - 1] / 64, 1), dim3(256), 0, h->st, (const T*)h->dl[t - 1],
                           h->Dp[t], (const T*)h->wb[t - 1], h->Dp[t], e);
    }
    }
    {   // all weight gradients: one problem per layer, own slab region each
        IpProf ps(h, "wgrad");
        size_t off = 0;
        for (int t = 1; t <= L + 1; ++t) {
            WgradArgs wa;
            wa.p[0] = WgradProb{h->aT[t - 1], h->dlT[t - 1], h->slab + off, h->Dp[t - 1] / 64, h->Dp[t] / 64, h->Dp[t]};
            wa.p[1] = wa.p[2] = wa.p[3] = WgradProb{nullptr, nullptr, nullptr, 0, 1, 64};
            wa.ldT = ldT; wa.klen = Ba / h->splitk; wa.zstride = h->slab_stride;
            hipLaunchKernelGGL((k_wgrad<T>), dim3((h->Dp[t - 1] / 64) * (h->Dp[t] / 64), h->splitk), dim3(256), 0, h->st, wa);
            off += (size_t)h->Dp[t - 1] * h->Dp[t];
        }
    }
    {
        IpProf ps(h, "ip_bwd");
        IpBwdArgs ba{ids, B, F, h->K, h->table16, h->n_rows, h->Dp[0]};
        hipLaunchKernelGGL(k_ip_bwd, dim3(Ba / 16), dim3(256), lds_ip, h->st, ba, h->dz0, h->gxp, h->gb_part);
    }
    {   // sparse rows: row -= lr * sum of its gradients (c = 1: the table of powers is all ones)
        IpProf ps(h, "scatter");
        ScatArgs sa{h->rec, SORT_N, F, h->K, h->gxp, h->Dp[0], h->cpow1, (double)h->cfg.lr, h->table16, h->part, h->owner_cnt,
                    h->owners, SLOT};
        hipLaunchKernelGGL(k_scat1, dim3(F * SORT_N / 256), dim3(256), 0, h->st, sa);
        hipLaunchKernelGGL(k_scat2, dim3(256), dim3(256), 0, h->st, sa);
    }
    {
        IpProf ps(h, "update");
        size_t off = 0;
        for (int t = 1; t <= L + 1; ++t) { ip_refresh<T>(h, t, h->slab + off, h->cfg.lr); off += (size_t)h->Dp[t - 1] * h->Dp[t]; }
        hipLaunchKernelGGL(k_ip_update_b, dim3(1), dim3(256), 0, h->st, h->b, h->gb_part, Ba / 16, h->cfg.lr, h->loss_t, Ba, h->loss_dev);
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
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_ip_err = "no HIP device (libfnn_hip.so has no CPU fallback)"; return FNN_ERR_HIP; }
    ipnn_handle* h = new ipnn_handle();
    h->cfg = *cfg; h->dev = cfg->device; h->F = cfg->n_fields; h->K = cfg->k; h->L = cfg->n_hidden;
    h->P = h->F * (h->F - 1) / 2; h->CB = h->F * SLOT + h->P; h->bf16 = cfg->precision == FNN_PREC_BF16;
    h->Bmax = cfg->max_batch; h->ldT = rup(h->Bmax, 256);
    auto fail = [&](int code) { g_ip_err = h->err; ipnn_destroy(h); return code; };
#define IK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e2_); return fail(FNN_ERR_HIP); } } while (0)
    IK(hipSetDevice(h->dev));
    if (cfg->stream) h->st = (hipStream_t)cfg->stream; else { IK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    h->d.resize(h->L + 2); h->Dp.resize(h->L + 2);
    h->d[0] = h->F * h->K + h->P + 1; h->Dp[0] = rup(h->CB + 2, 64);
    for (int t = 1; t <= h->L; ++t) { h->d[t] = cfg->hidden[t - 1]; h->Dp[t] = rup(h->d[t] + 1, 64); if (h->d[t] < 1 || h->d[t] > 4095) { h->err = "hidden size out of range"; return fail(FNN_ERR_ARG); } }
    h->d[h->L + 1] = 1; h->Dp[h->L + 1] = 64;
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
