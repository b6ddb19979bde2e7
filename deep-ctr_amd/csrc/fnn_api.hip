// fnn_api.hip -- host side of libfnn_hip.so: the C ABI declared in include/fnn_hip.h.
// Replaces the compiled Theano callables `train` / `predict` (python/FNN_wnzh.py:177-183 of
// Atomu2014/deep-ctr) and the Python gather / sparse-update loops around them (:87-96, :299-306).
// gfx950 only; there is no CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is opened with dlopen at fnn_dp_init
#include <dlfcn.h>

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fnn_hip.h"
#include "fnn_step_kernels.hip.h"
#include "metrics.hip.h"

using namespace fnn;

namespace {

thread_local std::string g_create_err;

struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double ms = 0; int64_t n = 0; };

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

struct fnn_handle {
    fnn_cfg cfg{};
    std::string err;
    int dev = 0;
    hipStream_t st = nullptr;
    bool own_stream = false;
    // shapes
    int F = 0, K = 0, H1 = 0, H2 = 0, xdim = 0, K1p = 0, H1p = 0, H2p = 0;
    int Bmax = 0, ldT = 0, N2max = 0;
    size_t n1 = 0, n2 = 0, nw12 = 0, nw = 0;
    int splitk = 4;             // split-K of the weight-gradient products (measured: 4 -> 40.8 us per step, 8 -> 42.9, 2 -> 51.1)
    int scat2_wgs = 256;        // workgroups walking the multi-chunk segments in launch 3
    bool bf16 = false;          // FNN_PREC_BF16: 2-byte elements
    bool split = false;         // FNN_PREC_BF16X3: 4-byte elements (bs16_t), the f32 mode's layouts
    int step1_waves = 8;                                               // FNN_STEP1_WAVES=4: four waves per strip (the 2-byte element types at hidden 300 / 100 run eight)
    int wt_stores = 15;                                                // FNN_WT_STORES (MlpArgs::wt; 0: plain stores): how the strip kernel's training outputs leave
    bool bag = false; int rw = SLOT; size_t nbag = 0, off_bag = 0;     // FNN_MODE_BAG: bag rows rw floats wide
    float* bb0 = nullptr; void* dlxT = nullptr; void* onesT = nullptr; float* gx_raw = nullptr;
    bool fused = true;          // one k_mlp launch instead of gather/fwd1/fwd2/head/bwd1/gx
    int role_off = 0;           // diagnostics only (FNN_ROLE_OFF): 1 sort, 2 dense, 4 sparse roles of launches 2/3 skipped
    // FM table
    float* table16 = nullptr; int32_t* field_of_row = nullptr; int64_t n_rows = 0; float w0 = 0.f;
    // dense
    float* master = nullptr; float* bucket = nullptr; float* slab = nullptr;
    void *w1 = nullptr, *w1t = nullptr, *w2 = nullptr, *w2t = nullptr;   // shadows (T)
    bool dense_set[3] = {false, false, false};
    // activations (T) and f32 work buffers
    void *xp = nullptr, *xpT = nullptr, *d1 = nullptr, *d1T = nullptr, *d2 = nullptr, *dl2 = nullptr,
         *dl2T = nullptr, *dl1 = nullptr, *dl1T = nullptr;
    float *gxp = nullptr, *p_buf = nullptr, *loss_t = nullptr, *loss_dev = nullptr;
    void *d2T = nullptr, *dl3T = nullptr;
    size_t nslab = 0;
    // scatter
    // Grouping results (sorted (row, t) records + level-2 work lists), double-buffered: the slot of
    // the batch being trained and the slot the NEXT batch is grouped into during this step.
    // tag_shared / stamp: bag mode, the rows several columns of the slot's batch hold (SortArgs).  One array per slot: the rank
    // merge of batch n + 1 writes its marks in the same launch in which the update of batch n still reads its own.
    struct SortSlot { int4* rec = nullptr; double* part = nullptr; int4* owners = nullptr; int* owner_cnt = nullptr;
                      int* tag_shared = nullptr; int stamp = 0; };
    int* tag_first = nullptr; int tag_stamp = 0;
    SortSlot slot[2]; int cur = 0;
    const int32_t* sorted_ids = nullptr; int sorted_B = 0;      // what slot[cur] holds (nullptr: nothing)
    const int32_t* next_ids = nullptr; int next_B = 0;          // pending fnn_prefetch_ids request
    bool key64 = false;
    void* skeys = nullptr;              // phase-A output of the split sort
    double* cpow_dev = nullptr;
    std::vector<double> cpow_host; double cpow_c = -1.0; int cpow_n = 0;
    int* err_flag = nullptr;
    uint8_t* ones_u8 = nullptr;         // all-ones dropout rows for predict
    // host-pointer staging
    int32_t* st_ids = nullptr; float* st_y = nullptr; uint8_t* st_m1 = nullptr; uint8_t* st_m2 = nullptr;
    float* st_p = nullptr; float* st_x = nullptr;
    // step state
    bool in_step = false, update_pending = false, scatter_pending = false, pend_have_next = false;
    int step_B = 0, pend_Ba = 0;
    int prefetch_hits = 0, prefetch_misses = 0;
    // features shadowed in the gather that the next step's sparse-row update still visits (fnn_set_shadowed)
    int32_t* shadow_dev = nullptr; int n_shadow = 0, shadow_cap = 0;
    // data parallelism (fnn_dp_init / fnn_dp_init_custom)
    bool dp = false, dp_own_comm = false; int dp_rank = 0, dp_world = 1, dp_sparse = FNN_DP_SPARSE_LOCAL;
    fnn_allreduce_fn dp_allreduce = nullptr; fnn_allgather_fn dp_allgather = nullptr; void* dp_ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t* xg_ids_send = nullptr; int32_t* xg_ids = nullptr; float* xg_gxp = nullptr;    // EXCHANGE: padded shard ids, gathered ids / gx'
    SortSlot gsl; int gN2 = 0; void* gws = nullptr; size_t gws_bytes = 0;      // grouping of a GLOBAL batch (fnn_step_scatter_global), grown on demand
    int cpow_cap = 0;                   // entries allocated in cpow_dev
    bool step_native_dp = false;        // the step in flight runs its own collective (fnn_train_step after fnn_dp_init)
    // what the dense collective carries and who performs it (fnn_dp_set_payload / fnn_dp_set_collective): handle-wide, never per shard
    int dp_payload = FNN_DP_PAYLOAD_SLABS, dp_collective = FNN_DP_COLLECTIVE_CALLBACK;
    // one-shot all-reduce over peer pointers: this rank's exchange region = [2 parities][xr_nbp floats] buckets, then 8 flags of 64 B
    float* xr = nullptr; size_t xr_nbp = 0, xr_bytes = 0; int xr_kind = 0; bool xr_same_process = false;
    float* peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool peer_opened[8] = {false, false, false, false, false, false, false, false};
    bool p2p_attached = false; unsigned long long dp_step_no = 0;
    unsigned long long p2p_timeout_ticks = 3000000000ull;      // 30 s of the 100 MHz clock ($FNN_P2P_TIMEOUT_MS)
    unsigned long long* p2p_wait_max = nullptr;                // device word: the longest flag wait so far (diagnostic)
    int step_bsize = 0;                 // its b_size (the decay table is rebuilt with it when a global batch outgrows the table)
    // profiling
    bool prof = false;
    std::map<std::string, ProfSlot> prof_slots;
};

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return FNN_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
#define FAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

namespace {

struct ProfScope {
    fnn_handle* h; hipStream_t s; hipEvent_t b = nullptr, e = nullptr; const char* name;
    ProfScope(fnn_handle* h_, const char* n, hipStream_t s_) : h(h_), s(s_), name(n) {
        if (!h->prof) return;
        hipEventCreate(&b); hipEventCreate(&e); hipEventRecord(b, s);
    }
    ~ProfScope() {
        if (!h->prof) return;
        hipEventRecord(e, s);
        h->prof_slots[name].ev.emplace_back(b, e);
    }
};

template <typename T> int alloc_dev(fnn_handle* h, T** p, size_t n, bool zero = true) {
    HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
    if (zero) HIPCHK(h, hipMemsetAsync(*p, 0, n * sizeof(T), h->st));
    return FNN_OK;
}

size_t tsize(const fnn_handle* h) { return h->bf16 ? 2 : 4; }
// the element type of the handle's precision: FN<T> ARGS
#define BY_PREC(h, FN, ARGS) do { if ((h)->bf16) FN<bf16_t> ARGS; else if ((h)->split) FN<bs16_t> ARGS; else FN<float> ARGS; } while (0)
#define BY_PREC_RC(h, FN, ARGS) ((h)->bf16 ? FN<bf16_t> ARGS : ((h)->split ? FN<bs16_t> ARGS : FN<float> ARGS))

int check_async(fnn_handle* h) {
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (flag) {
        HIPCHK(h, hipMemsetAsync(h->err_flag, 0, sizeof(int), h->st));
        if (flag & 16) FAIL(h, FNN_ERR_HIP, "data-parallel p2p all-reduce: a peer's flag did not arrive in time; the dense tensors of that step were left untouched");
        FAIL(h, FNN_ERR_RANGE, "feature id outside [-1, n_rows) (reference: KeyError, python/FNN_wnzh.py:95)");
    }
    return FNN_OK;
}

int update_cpow(fnn_handle* h, int b_size, int B) {
    const double c = 1.0 - 2.0 * (double)h->cfg.lambda_fm * (double)h->cfg.lr / (double)b_size;
    if (c == h->cpow_c && h->cpow_n >= B + 1) return FNN_OK;
    // in-flight kernels may still read the old table
    HIPCHK(h, hipStreamSynchronize(h->st));
    const int n = h->cpow_cap;
    h->cpow_host.resize(n);
    for (int i = 0; i < n; ++i) h->cpow_host[i] = std::pow(c, (double)i);
    HIPCHK(h, hipMemcpy(h->cpow_dev, h->cpow_host.data(), n * sizeof(double), hipMemcpyHostToDevice));
    h->cpow_c = c; h->cpow_n = n;
    return FNN_OK;
}

template <typename T, int NT, typename Epi>
void launch_gemm(hipStream_t s, const T* A, int lda, const T* Bft, int M, int N, int klen, Epi epi) {
    dim3 grid(M / 64, N / (16 * NT), 1);
    hipLaunchKernelGGL((k_gemm<T, NT, Epi>), grid, dim3(256), 0, s, A, lda, Bft, klen, epi);
}

// refresh shadows from the masters without a gradient step
template <typename T> void launch_update(fnn_handle* h, const float* bucket, float lr) {
    const size_t n = h->nw;
    hipLaunchKernelGGL((k_update<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->master,
                       bucket, lr, h->cfg.lambda1, h->cfg.reg_all, h->K1p, h->H1p, h->H2p, (T*)h->w1, (T*)h->w1t,
                       (T*)h->w2, (T*)h->w2t);
    if (h->bag && bucket)
        hipLaunchKernelGGL(k_axpy, dim3((unsigned)((h->nbag + 255) / 256)), dim3(256), 0, h->st, h->bb0,
                           bucket + h->nw, -lr, (int)h->nbag);
}

// the strip kernel is instantiated for the padded shapes in use; anything else takes the
// layer-by-layer kernels
bool mlp_shape_ok(const fnn_handle* h) {
    const int c1 = h->H1p / 64, c2 = h->H2p / 64, cx = h->K1p / 64;
    if (cx == 5) return h->bag && c1 == 5 && c2 == 2;        // bag rows 256..316 wide: the reference's default hidden0 = 300 (python/SNN_RBM.py:25)
    return cx == 4 && ((c1 == 5 && c2 == 2) || (c1 == 1 && c2 == 1));
}
// a fresh stamp for a grouping about to be written into `sl` (bag mode; see SortArgs).  2^25 groupings, then the tags start over.
int next_stamp(fnn_handle* h, fnn_handle::SortSlot& sl) {
    if (!h->bag) return 0;
    if (h->tag_stamp >= (1 << 25) - 1) {
        hipMemsetAsync(h->tag_first, 0, (size_t)h->n_rows * sizeof(int), h->st);
        for (auto& q : h->slot) hipMemsetAsync(q.tag_shared, 0, (size_t)h->n_rows * sizeof(int), h->st);
        h->tag_stamp = 0;
    }
    sl.stamp = ++h->tag_stamp;
    return sl.stamp;
}
int sort_n2(int B) { int N2 = 256; while (N2 < B) N2 <<= 1; return N2; }
constexpr int GLOBAL_BATCH_MAX = 32768;        // fnn_step_scatter_global / FNN_DP_SPARSE_EXCHANGE: 8 ranks x 4096 examples

template <typename T> WgradArgs make_wgrad_args(fnn_handle* h, int Ba) {
    WgradArgs wa;
    wa.p[0] = WgradProb{h->xpT, h->dl1T, h->slab, h->K1p / 64, h->H1p / 64, h->H1p};
    wa.p[1] = WgradProb{h->d1T, h->dl2T, h->slab + h->n1, h->H1p / 64, h->H2p / 64, h->H2p};
    wa.p[2] = WgradProb{h->d2T, h->dl3T, h->slab + h->nw12, h->H2p / 64, 1, 64};     // gw3p = column 0
    wa.p[3] = WgradProb{h->dlxT, h->onesT, h->slab + h->off_bag, h->bag ? h->K1p / 64 : 0, 1, 64};   // sum_t delta_t
    wa.ldT = h->ldT; wa.klen = Ba / h->splitk; wa.zstride = h->nslab;
    return wa;
}
int wgrad_blocks(const WgradArgs& wa) {
    return wa.p[0].mt * wa.p[0].nt + wa.p[1].mt * wa.p[1].nt + wa.p[2].mt * wa.p[2].nt + wa.p[3].mt * wa.p[3].nt;
}
ScatArgs make_scat_args(fnn_handle* h, const fnn_handle::SortSlot& sl, int N2) {
    return ScatArgs{sl.rec, N2, h->F, h->K, h->gxp, h->K1p, h->cpow_dev, (double)h->cfg.lr, h->table16,
                    sl.part, sl.owner_cnt, sl.owners, h->rw, sl.tag_shared, sl.stamp};
}
template <typename T> MlpArgs<T> make_mlp_args(fnn_handle* h, const int32_t* ids, const float* y, int B,
                                                const uint8_t* m1, const uint8_t* m2, bool train, float* p_out) {
    return MlpArgs<T>{ids, y ? y : h->loss_t, B, h->F, h->K, h->table16, h->n_rows, h->w0,
                      (const T*)h->w1, (const T*)h->w1t, (const T*)h->w2, (const T*)h->w2t, h->master + h->nw12,
                      m1 ? m1 : h->ones_u8, m2 ? m2 : h->ones_u8, h->cfg.act, train ? ACT_TANH : h->cfg.act,
                      h->H1, h->H2, train ? 1 : 0,
                      (T*)h->xpT, (T*)h->d1T, (T*)h->d2T, (T*)h->dl1T, (T*)h->dl2T, (T*)h->dl3T, h->ldT,
                      h->gxp, p_out, h->loss_t, h->err_flag, h->bb0, h->rw, (T*)h->dlxT, nullptr, h->wt_stores};
}
template <typename T, int C1, int C2, int CX> size_t mlp_lds_bytes(bool bag, int F, int nw = 4) {
    constexpr int PAD = 16 / (int)sizeof(T);
    constexpr int LX = 64 * CX + PAD, L1 = 64 * C1 + PAD, L2 = 64 * C2 + PAD, LXM = LX > L1 ? LX : L1;
    size_t n = (size_t)16 * (LXM + L1 + L2) * sizeof(T) + (size_t)16 * nw * sizeof(float);
    if (bag) n += (size_t)16 * 64 * CX * sizeof(float) + (size_t)16 * F * sizeof(int);
    else n += (size_t)nw * 16 * 36 * sizeof(float);           // the waves' blocks for regrouping gx' into whole lines (mlp_body, P4)
    return n;
}
template <typename T> void launch_step1(fnn_handle* h, int nmlp, const MlpArgs<T>& a) {
    const bool big = h->H1p / 64 == 5;
    const dim3 g(nmlp), b(256);
    {
        // hidden 300 / 100: EIGHT waves per strip (two per SIMD), a layer's 16-column fragments in runs of
        // ceil(n / 8) per wave -- the strip is a chain of phases, each as long as its busiest wave's run (FNN_STEP1_WAVES=4: four waves)
        if (big && h->step1_waves == 8) {
            const dim3 b8(512);
            if (h->K1p / 64 == 5) {
                const size_t lds = mlp_lds_bytes<T, 5, 2, 5>(true, h->F, 8);
                hipLaunchKernelGGL((k_step1<T, 5, 2, 5, true, 8>), g, b8, lds, h->st, a);
            } else if (h->bag) {
                const size_t lds = mlp_lds_bytes<T, 5, 2, 4>(true, h->F, 8);
                hipLaunchKernelGGL((k_step1<T, 5, 2, 4, true, 8>), g, b8, lds, h->st, a);
            } else {
                const size_t lds = mlp_lds_bytes<T, 5, 2, 4>(false, h->F, 8);
                hipLaunchKernelGGL((k_step1<T, 5, 2, 4, false, 8>), g, b8, lds, h->st, a);
            }
            return;
        }
    }
    if (h->K1p / 64 == 5) {                                      // bag mode only (mlp_shape_ok)
        const size_t lds5 = mlp_lds_bytes<T, 5, 2, 5>(true, h->F);
        hipLaunchKernelGGL((k_step1<T, 5, 2, 5, true>), g, b, lds5, h->st, a);
        return;
    }
    const size_t lds = big ? mlp_lds_bytes<T, 5, 2, 4>(h->bag, h->F) : mlp_lds_bytes<T, 1, 1, 4>(h->bag, h->F);
    if (h->bag) {
        if (big) hipLaunchKernelGGL((k_step1<T, 5, 2, 4, true>), g, b, lds, h->st, a);
        else hipLaunchKernelGGL((k_step1<T, 1, 1, 4, true>), g, b, lds, h->st, a);
    } else {
        if (big) hipLaunchKernelGGL((k_step1<T, 5, 2, 4, false>), g, b, lds, h->st, a);
        else hipLaunchKernelGGL((k_step1<T, 1, 1, 4, false>), g, b, lds, h->st, a);
    }
}
void launch_sort16(fnn_handle* h, const SortArgs& so) {      // split sort: 256-key runs, then the rank merge
    const int F = so.nblk;
    if (h->key64) {
        hipLaunchKernelGGL((k_sortA<unsigned long long>), dim3(4 * F), dim3(256), 0, h->st, so);
        hipLaunchKernelGGL((k_sortB<unsigned long long>), dim3(16 * F), dim3(256), SORT_N * 8, h->st, so);
    } else {
        hipLaunchKernelGGL((k_sortA<unsigned>), dim3(4 * F), dim3(256), 0, h->st, so);
        hipLaunchKernelGGL((k_sortB<unsigned>), dim3(16 * F), dim3(256), SORT_N * 4, h->st, so);
    }
}

// Launches 2 and 3 of a step.  `dense`: weight gradients / slab reduce (+ update); `sparse`: the
// sparse-row SGD of this batch and the grouping of the next one.  Single-GPU and native data-parallel
// steps run both halves in the same two launches (the data-parallel one with its all-reduce of the slabs
// between them); the portable split API (fnn_step_begin / _scatter / _end) runs the halves separately.
template <typename T>
void launch_step2(fnn_handle* h, bool dense, bool sparse, const float* gxp_src = nullptr)
{
    const int Ba = h->pend_Ba, nxt = h->cur ^ 1;
    const bool have_next = sparse && h->pend_have_next && !(h->role_off & 1);
    if (h->role_off & 2) dense = false;                     // timing experiments: results are wrong by construction
    if (h->role_off & 4) sparse = false;
    ScatArgs sa = make_scat_args(h, h->slot[h->cur], SORT_N);
    if (gxp_src) sa.gxp = gxp_src;                          // the sparse role reads another shard's gathered gradients (bag EXCHANGE)
    ProfScope ps(h, dense && sparse ? "step2" : (dense ? "step2_dense" : "step2_sparse"), h->st);
    const WgradArgs wa = make_wgrad_args<T>(h, Ba);
    const int nwx = dense ? wgrad_blocks(wa) : 0;
    const int nsc = !sparse ? 0 : (h->bag ? (int)(((size_t)h->F * (SORT_N / WCH) * (h->rw / 4) + 255) / 256)
                                          : h->F * SORT_N / 256);
    SortArgs so{h->next_ids, h->next_B, h->F, h->n_rows, h->slot[nxt].rec, h->slot[nxt].owner_cnt,
                have_next ? 4 * h->F : 0, h->skeys, nullptr, nullptr, 0};
    const dim3 grid(so.nblk + nwx * h->splitk + nsc);
    if (grid.x == 0) return;
    if (h->key64)
        hipLaunchKernelGGL((k_step2<T, unsigned long long>), grid, dim3(256), 0, h->st, so, wa, nwx,
                           h->splitk, sa);
    else
        hipLaunchKernelGGL((k_step2<T, unsigned>), grid, dim3(256), 0, h->st, so, wa, nwx, h->splitk, sa);
}

template <typename T>
void launch_step3(fnn_handle* h, bool dense, bool sparse, bool update, float* bucket_dst = nullptr, const float* gxp_src = nullptr)
{
    const int Ba = h->pend_Ba, nxt = h->cur ^ 1;
    const bool have_next = sparse && h->pend_have_next && !(h->role_off & 1);
    if (h->role_off & 2) dense = false;
    if (h->role_off & 4) sparse = false;
    ScatArgs sa = make_scat_args(h, h->slot[h->cur], SORT_N);
    if (gxp_src) sa.gxp = gxp_src;
    {
        ProfScope ps(h, dense && sparse ? "step3" : (dense ? "step3_dense" : "step3_sparse"), h->st);
        const int nred = dense ? (int)((h->nw12 / 4 + 255) / 256) + (int)((h->nw - h->nw12 + h->nbag + 255) / 256) + 1 : 0;
        TailArgs ta{h->slab, h->splitk, h->nw, h->nw12, h->nslab, h->master, h->cfg.lambda1, h->cfg.reg_all,
                    h->loss_t, Ba, bucket_dst ? bucket_dst : h->bucket, h->loss_dev, h->cfg.lr, h->K1p, h->H1p, h->H2p,
                    h->w1, h->w1t, h->w2, h->w2t, nred, h->bb0, h->nbag, h->off_bag};
        const int stamp = have_next ? next_stamp(h, h->slot[nxt]) : 0;      // the rank merge of the NEXT batch marks its shared rows
        SortArgs so{h->next_ids, h->next_B, h->F, h->n_rows, h->slot[nxt].rec, h->slot[nxt].owner_cnt,
                    have_next ? 16 * h->F : 0, h->skeys, h->tag_first, h->slot[nxt].tag_shared, stamp};
        const dim3 grid(so.nblk + nred + (sparse ? h->scat2_wgs : 0));    // these workgroups walk the multi-chunk segments
        const size_t lds = h->key64 ? sort_lds_bytes<unsigned long long>() : sort_lds_bytes<unsigned>();
        if (grid.x > 0) {
            if (h->key64) {
                if (update) hipLaunchKernelGGL((k_step3<T, true, unsigned long long>), grid, dim3(256), lds, h->st, so, ta, sa);
                else hipLaunchKernelGGL((k_step3<T, false, unsigned long long>), grid, dim3(256), lds, h->st, so, ta, sa);
            } else {
                if (update) hipLaunchKernelGGL((k_step3<T, true, unsigned>), grid, dim3(256), lds, h->st, so, ta, sa);
                else hipLaunchKernelGGL((k_step3<T, false, unsigned>), grid, dim3(256), lds, h->st, so, ta, sa);
            }
        }
    }
    if (sparse) {
        if (have_next) { h->cur = nxt; h->sorted_ids = h->next_ids; h->sorted_B = h->next_B; }
        else { h->sorted_ids = nullptr; h->sorted_B = 0; }
        h->next_ids = nullptr; h->next_B = 0;
        h->scatter_pending = false;
    }
}

template <typename T>
void launch_steps23(fnn_handle* h, bool dense, bool sparse, bool update)
{
    launch_step2<T>(h, dense, sparse);
    launch_step3<T>(h, dense, sparse, update);
}

// ---- collectives of the native data-parallel step, enqueued on the handle's stream
int dp_allreduce(fnn_handle* h, float* buf, size_t n, const char* what)
{
    ProfScope ps(h, "allreduce", h->st);
    h->err.clear();                              // the RCCL callback leaves its message here; a caller's callback leaves nothing
    const int rc = h->dp_allreduce(h->dp_ctx, buf, (int64_t)n, (void*)h->st);
    if (rc != 0) FAIL(h, FNN_ERR_HIP, std::string("data-parallel all-reduce (") + what + ") failed" + (h->err.empty() ? std::string(": the callback returned ") + std::to_string(rc) : ": " + h->err));
    return FNN_OK;
}

// Bucket payload: the flat bucket holds THIS rank's dense gradients (launch 3 without its update, or the layer-by-layer
// k_reduce); sum it over the ranks and apply theta <- theta - lr * (sum + L2 term).  Callback collective: all-reduce in place,
// then k_update.  P2P: the bucket sits in (or is copied into) this rank's exchange region and ONE launch does the rest.
inline float* p2p_bucket(fnn_handle* h) { return h->xr + (size_t)(h->dp_step_no & 1) * h->xr_nbp; }
template <typename T> int dp_finish_bucket(fnn_handle* h, bool in_region)
{
    if (h->dp_collective == FNN_DP_COLLECTIVE_P2P) {
        if (!in_region)
            HIPCHK(h, hipMemcpyAsync(p2p_bucket(h), h->bucket, (h->nw + h->nbag) * sizeof(float), hipMemcpyDeviceToDevice, h->st));
        ProfScope ps(h, "p2p_update", h->st);
        P2PArgs pa;
        for (int r = 0; r < 8; ++r) pa.peer[r] = h->peer[r];
        pa.world = h->dp_world; pa.rank = h->dp_rank; pa.step = h->dp_step_no + 1;
        pa.bucket_off = (size_t)(h->dp_step_no & 1) * h->xr_nbp; pa.flag_off = 2 * h->xr_nbp * sizeof(float); pa.err = h->err_flag;
        pa.timeout_ticks = h->p2p_timeout_ticks; pa.wait_max = h->p2p_wait_max;
        const unsigned nblk = (unsigned)((h->nw12 / 4 + 255) / 256 + (h->nw - h->nw12 + h->nbag + 255) / 256);
        hipLaunchKernelGGL((k_p2p_update<T>), dim3(nblk), dim3(256), 0, h->st, pa, h->master, h->cfg.lr,
                           h->cfg.lambda1, h->cfg.reg_all, h->K1p, h->H1p, h->H2p, (T*)h->w1, (T*)h->w1t, (T*)h->w2, (T*)h->w2t,
                           h->bb0, h->nw12, h->nw, h->nbag);
        h->dp_step_no++;
        return FNN_OK;
    }
    int rc = dp_allreduce(h, h->bucket, h->nw + h->nbag, "dense-gradient bucket");
    if (rc != FNN_OK) return rc;
    ProfScope ps(h, "update", h->st);
    launch_update<T>(h, h->bucket, h->cfg.lr);
    return FNN_OK;
}
inline bool dp_bucket_mode(const fnn_handle* h) { return h->dp_collective == FNN_DP_COLLECTIVE_P2P || h->dp_payload == FNN_DP_PAYLOAD_BUCKET; }

int ensure_global_ws(fnn_handle* h, int B_g);
int scatter_global_impl(fnn_handle* h, const int32_t* ids_g, const float* gxp_g, int B_g);

// EXCHANGE: all-gather (ids, gx') of every shard, each padded to ldT rows (empty ids), then the sparse-row SGD of the whole
// global batch in global example order on this rank.
// Bag mode (python/SNN_RBM.py:285-291: ww0[f] -= delta_t, no decay -- a plain sum over the examples): the gathered shards are
// applied ONE AFTER THE OTHER in rank order with the step's own sparse role (grouping of the shard's ids, then the two levels of
// the segmented row update reading that shard's gathered deltas): every rank performs the same operations in the same order, so
// the replicas stay bit-identical to each other, and equal to the single-process step of the global batch up to the rounding of
// `world` partial sums per row instead of one.
template <typename T>
int dp_exchange_sparse(fnn_handle* h, const int32_t* ids, int B)
{
    const int rows = h->ldT, F = h->F;
    {
        ProfScope ps(h, "allgather", h->st);
        const size_t n = (size_t)rows * F;
        hipLaunchKernelGGL(k_pad_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, ids, B, F, rows, h->xg_ids_send);
        int rc = h->dp_allgather(h->dp_ctx, h->xg_ids_send, h->xg_ids, (int64_t)(n * 4), (void*)h->st);
        if (rc == 0) rc = h->dp_allgather(h->dp_ctx, h->gxp, h->xg_gxp, (int64_t)((size_t)rows * h->K1p * 4), (void*)h->st);
        if (rc != 0) FAIL(h, FNN_ERR_HIP, "data-parallel all-gather failed: " + h->err);
    }
    if (!h->bag) return scatter_global_impl(h, h->xg_ids, h->xg_gxp, rows * h->dp_world);
    h->pend_have_next = false;
    for (int r = 0; r < h->dp_world; ++r) {
        const int32_t* ids_r = h->xg_ids + (size_t)r * rows * F;
        fnn_handle::SortSlot& sl = h->slot[h->cur];
        {
            ProfScope ps(h, "sort_now", h->st);
            SortArgs so{ids_r, rows, F, h->n_rows, sl.rec, sl.owner_cnt, F, h->skeys, h->tag_first, sl.tag_shared, next_stamp(h, sl)};
            launch_sort16(h, so);
        }
        const float* gx_r = h->xg_gxp + (size_t)r * rows * h->K1p;
        launch_step2<T>(h, false, true, gx_r);
        launch_step3<T>(h, false, true, false, nullptr, gx_r);
    }
    HIPCHK(h, hipGetLastError());
    return FNN_OK;
}

// The fast path: three role-split launches on the main stream (fnn_step_kernels.hip.h).
template <typename T>
int run_step_fast(fnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* m1,
                  const uint8_t* m2, float* p_out, float* gx_out_dev, bool update)
{
    const int Ba = rup(B, 256);
    const bool exchange = update && h->dp && h->dp_sparse == FNN_DP_SPARSE_EXCHANGE;   // the rows are grouped over the GLOBAL batch later
    // grouping of THIS batch: done by the previous step (fnn_prefetch_ids) or right now
    if (exchange) { h->next_ids = nullptr; h->next_B = 0; }
    else if (!(h->sorted_ids == ids && h->sorted_B == B)) {
        h->prefetch_misses++;
        ProfScope ps(h, "sort_now", h->st);
        SortArgs so{ids, B, h->F, h->n_rows, h->slot[h->cur].rec, h->slot[h->cur].owner_cnt, h->F, h->skeys,
                    h->tag_first, h->slot[h->cur].tag_shared, next_stamp(h, h->slot[h->cur])};
        launch_sort16(h, so);
    } else h->prefetch_hits++;
    const bool have_next = h->next_ids != nullptr && !(h->next_ids == ids && h->next_B == B);
    const int nxt = h->cur ^ 1;
    { ProfScope pe(h, "empty", h->st); }         // a back-to-back event pair: what the event mechanism itself adds to every slot
    {
        ProfScope ps(h, "step1", h->st);
        MlpArgs<T> ma = make_mlp_args<T>(h, ids, y, B, m1, m2, true, p_out);
        if (h->bag && gx_out_dev) ma.gx_raw = h->gx_raw;
        launch_step1<T>(h, Ba / 16, ma);
    }
    if (gx_out_dev) {
        const size_t n = (size_t)B * h->xdim;
        if (h->bag) hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->gx_raw,
                                       h->K1p, B, h->xdim, gx_out_dev);
        else hipLaunchKernelGGL(k_gx_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->gxp, h->K1p,
                                B, h->F, h->K, gx_out_dev);
    }
    h->pend_Ba = Ba; h->pend_have_next = have_next;
    if (update && h->dp) {
        // native data-parallel step: the same launches with ONE collective between the second and the third -- the split-K
        // slabs of the weight gradients are all-reduced in place, the third launch then sums GLOBAL slabs
        const bool local = h->dp_sparse == FNN_DP_SPARSE_LOCAL;
        launch_step2<T>(h, true, local);
        int rc;
        if (!dp_bucket_mode(h)) {
            rc = dp_allreduce(h, h->slab, (size_t)h->splitk * h->nslab, "weight-gradient slabs");
            if (rc != FNN_OK) return rc;
            launch_step3<T>(h, true, local, true);
        } else {
            // bucket payload: launch 3 sums this rank's slabs (no update), the collective carries a quarter of the bytes, a fourth
            // launch applies the update -- with the p2p collective that launch IS the all-reduce
            const bool p2p = h->dp_collective == FNN_DP_COLLECTIVE_P2P;
            launch_step3<T>(h, true, local, false, p2p ? p2p_bucket(h) : nullptr);
            rc = dp_finish_bucket<T>(h, p2p);
            if (rc != FNN_OK) return rc;
        }
        if (!local) {
            h->pend_have_next = false;
            rc = dp_exchange_sparse<T>(h, ids, B);
            if (rc != FNN_OK) return rc;
        }
    }
    else if (update) launch_steps23<T>(h, true, true, true);     // dense and sparse roles share the launches
    else { launch_steps23<T>(h, true, false, false); h->scatter_pending = true; }   // split API: sparse half deferred
    HIPCHK(h, hipGetLastError());
    return FNN_OK;
}

// The generic path: one kernel per layer / stage, any padded shape, B up to 16384.
template <typename T>
int run_step(fnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* m1,
             const uint8_t* m2, bool train, float* p_out, float* gx_out_dev)
{
    const int Ba = rup(B, 256);
    const int F = h->F, K = h->K, K1p = h->K1p, H1p = h->H1p, H2p = h->H2p, ldT = h->ldT;
    T *xp = (T*)h->xp, *xpT = (T*)h->xpT, *d1 = (T*)h->d1, *d1T = (T*)h->d1T, *d2 = (T*)h->d2,
      *dl2 = (T*)h->dl2, *dl2T = (T*)h->dl2T, *dl1 = (T*)h->dl1, *dl1T = (T*)h->dl1T;
    // shadowed features join their field's keys behind the B regular ones: room for all of them in any one field
    const int nsh = train ? h->n_shadow : 0;
    const int N2 = sort_n2(B + nsh);
    T *d2T = (T*)h->d2T, *dl3T = (T*)h->dl3T;
    if (nsh > 0) {
        if (B + nsh > 16384) FAIL(h, FNN_ERR_ARG, "batch plus shadowed features exceed 16384 keys per field");
        int rc = ensure_global_ws(h, B + nsh);                 // the per-batch slots hold max_batch keys per field
        if (rc != FNN_OK) return rc;
    }
    fnn_handle::SortSlot& sl = nsh > 0 ? h->gsl : h->slot[h->cur];
    h->sorted_ids = nullptr; h->next_ids = nullptr;            // this path keeps no grouping across steps
    if (h->bag && !(h->fused && mlp_shape_ok(h))) FAIL(h, FNN_ERR_ARG, "FNN_MODE_BAG needs the strip kernel (hidden sizes 300/100 or <=63/<=63)");
    if (h->bag && train) FAIL(h, FNN_ERR_ARG, "FNN_MODE_BAG trains through the three-launch path only (B <= 4096)");
    if (train && h->step_native_dp && h->dp_sparse == FNN_DP_SPARSE_EXCHANGE)
        FAIL(h, FNN_ERR_ARG, "FNN_DP_SPARSE_EXCHANGE runs on the three-launch path only (B <= 4096, hidden sizes the strip kernel is built for)");

    if (train) {   // A6 part 1: group the (row, t) pairs
        ProfScope ps(h, "sort", h->st);
        const int kpt = N2 <= 4096 ? 4 : (N2 == 8192 ? 8 : 16);
        const dim3 blk(N2 / kpt);
        if (kpt == 4)
            hipLaunchKernelGGL(k_sort<4>, dim3(F), blk, (size_t)N2 * 8, h->st, ids, B, F, h->n_rows, N2, sl.rec, sl.owner_cnt, h->shadow_dev, h->n_shadow, h->err_flag);
        else if (kpt == 8)
            hipLaunchKernelGGL(k_sort<8>, dim3(F), blk, (size_t)N2 * 8, h->st, ids, B, F, h->n_rows, N2, sl.rec, sl.owner_cnt, h->shadow_dev, h->n_shadow, h->err_flag);
        else
            hipLaunchKernelGGL(k_sort<16>, dim3(F), blk, (size_t)N2 * 8, h->st, ids, B, F, h->n_rows, N2, sl.rec, sl.owner_cnt, h->shadow_dev, h->n_shadow, h->err_flag);
    }
    if (h->fused && mlp_shape_ok(h)) {
        ProfScope ps(h, "mlp", h->st);
        launch_step1<T>(h, Ba / 16, make_mlp_args<T>(h, ids, y, B, m1, m2, train, p_out));
        if (!train) return FNN_OK;
    } else {
        {   // A3
            ProfScope ps(h, "gather", h->st);
            const int nthreads = Ba * F;
            hipLaunchKernelGGL((k_gather<T>), dim3((nthreads + 255) / 256), dim3(256), 0, h->st, ids, B, Ba,
                               F, K, h->table16, h->n_rows, h->w0, xp, K1p, xpT, ldT, h->err_flag);
        }
        {   // A4 layer 1: d1 = act(x' W1p) * r1
            ProfScope ps(h, "fwd1", h->st);
            EpiFwd<T> e{d1, H1p, train ? d1T : nullptr, ldT, m1, h->cfg.act, h->H1, B};
            launch_gemm<T, 4>(h->st, xp, K1p, (const T*)h->w1t, Ba, H1p, K1p, e);
        }
        {   // A4 layer 2: d2 = tanh(d1 W2p) * r2   (predict: acti_type, no mask)
            ProfScope ps(h, "fwd2", h->st);
            EpiFwd<T> e{d2, H2p, train ? d2T : nullptr, ldT, m2, train ? ACT_TANH : h->cfg.act, h->H2, B};
            launch_gemm<T, 4>(h->st, d1, H1p, (const T*)h->w2t, Ba, H2p, H1p, e);
        }
        {   // output unit, loss, delta2
            ProfScope ps(h, "head", h->st);
            hipLaunchKernelGGL((k_head<T>), dim3(Ba / 64), dim3(256), 0, h->st, d2, H2p, h->H2,
                               h->master + h->nw12, m2, y, B, train ? 1 : 0, p_out, dl2, dl2T, ldT, dl3T,
                               h->loss_t);
        }
        if (!train) return FNN_OK;
        {   // A5: delta1 = (delta2 W2p^T) * r1 * act'(d1)
            ProfScope ps(h, "bwd1", h->st);
            EpiBwd<T> e{dl1, H1p, dl1T, ldT, d1, m1, h->cfg.act, h->H1, B};
            launch_gemm<T, 4>(h->st, dl2, H2p, (const T*)h->w2, Ba, H1p, H2p, e);
        }
        {   // A5: gx' = delta1 W1p^T
            ProfScope ps(h, "gx", h->st);
            EpiF32 e{h->gxp, K1p, 0};
            launch_gemm<T, 4>(h->st, dl1, H1p, (const T*)h->w1, Ba, K1p, H1p, e);
        }
    }
    {   // A5: dense gradients, contraction over the examples, split-K slabs
        ProfScope ps(h, "wgrad", h->st);
        const WgradArgs wa = make_wgrad_args<T>(h, Ba);
        hipLaunchKernelGGL((k_wgrad<T>), dim3(wgrad_blocks(wa), h->splitk), dim3(256), 0, h->st, wa);
    }
    if (train && h->step_native_dp && !dp_bucket_mode(h)) {
        // native data parallelism, slabs payload: the SAME collective as the three-launch path issues, so that a rank whose shard
        // takes these kernels (shadowed features, B > 4096) pairs with peers that do not (round-2 advisor: mismatched counts)
        int rc = dp_allreduce(h, h->slab, (size_t)h->splitk * h->nslab, "weight-gradient slabs");
        if (rc != FNN_OK) return rc;
    }
    {
        ProfScope ps(h, "reduce", h->st);
        hipLaunchKernelGGL(k_reduce, dim3((unsigned)((h->nw + 255) / 256 + 1)), dim3(256), 0, h->st, h->slab,
                           h->splitk, h->nw, h->nw12, h->nslab, h->master, h->cfg.lambda1, h->cfg.reg_all,
                           h->loss_t, Ba, h->bucket, h->loss_dev);
    }
    if (gx_out_dev) {
        const size_t n = (size_t)B * h->xdim;
        hipLaunchKernelGGL(k_gx_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->gxp, K1p,
                           B, F, K, gx_out_dev);
    }
    const ScatArgs sa = make_scat_args(h, sl, N2);           // A6 part 2
    {
        ProfScope ps(h, "scatter", h->st);
        const size_t nthr = (size_t)F * N2;                  // 16 lanes per chunk of 16 entries
        hipLaunchKernelGGL(k_scat1, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, h->st, sa);
    }
    {
        ProfScope ps(h, "finalize", h->st);
        hipLaunchKernelGGL(k_scat2, dim3(64), dim3(256), 0, h->st, sa);
    }
    HIPCHK(h, hipGetLastError());
    return FNN_OK;
}

// Workspaces for the grouping and the two-level update of a GLOBAL batch of up to N2 (row, t) pairs per field: the per-rank
// slots are sized for max_batch, a global batch is world times that.  Grown on demand (the first call synchronises).
int ensure_global_ws(fnn_handle* h, int B_g)
{
    int N2 = sort_n2(B_g);
    if (N2 <= h->gN2 && h->cpow_cap >= B_g + 1) return FNN_OK;
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (N2 > h->gN2) {
        fnn_handle::SortSlot& sl = h->gsl;
        for (void* q : {(void*)sl.rec, (void*)sl.part, (void*)sl.owners, (void*)sl.owner_cnt}) if (q) hipFree(q);
        sl = fnn_handle::SortSlot();
        int rc;
        const size_t nchunk = (size_t)N2 / 16;
        if ((rc = alloc_dev(h, &sl.rec, (size_t)h->F * N2)) != FNN_OK) return rc;
        if ((rc = alloc_dev(h, &sl.part, (size_t)h->F * nchunk * 2 * SLOT)) != FNN_OK) return rc;
        if ((rc = alloc_dev(h, &sl.owners, (size_t)h->F * nchunk)) != FNN_OK) return rc;
        if ((rc = alloc_dev(h, &sl.owner_cnt, (size_t)1)) != FNN_OK) return rc;
        h->gN2 = N2;
    }
    if (h->cpow_cap < N2 + 1) {             // a row can be hit by every example of the global batch
        hipFree(h->cpow_dev); h->cpow_dev = nullptr;
        h->cpow_cap = N2 + 1;
        int rc = alloc_dev(h, &h->cpow_dev, (size_t)h->cpow_cap);
        if (rc != FNN_OK) return rc;
        h->cpow_c = -1.0; h->cpow_n = 0;
        if (h->step_bsize > 0 && (rc = update_cpow(h, h->step_bsize, B_g)) != FNN_OK) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->st));
    return FNN_OK;                              // (the LDS attributes of k_sort<8> / <16> are set at fnn_create)
}

// sparse-row SGD of a global batch in global example order (python/FNN_wnzh.py:299-306): one grouping per field over the
// global (row, t) keys -- the one-workgroup bitonic sort up to 16,384 keys, the library's own radix sort (metrics.hip) beyond -- then the
// two-level segmented update reading the gathered gradients
int scatter_global_impl(fnn_handle* h, const int32_t* ids_g, const float* gxp_g, int B_g)
{
    int rc = ensure_global_ws(h, B_g);
    if (rc != FNN_OK) return rc;
    if (h->cpow_n < B_g + 1) FAIL(h, FNN_ERR_STATE, "decay table shorter than the global batch");
    fnn_handle::SortSlot& sl = h->gsl;
    const int N2 = sort_n2(B_g), F = h->F;
    {
        ProfScope ps(h, "sort_global", h->st);
        if (N2 > 16384) {
            std::string err;
            if (group_global(h->st, ids_g, B_g, F, h->n_rows, N2, sl.rec, sl.owner_cnt, &h->gws, &h->gws_bytes, err) != 0)
                FAIL(h, FNN_ERR_HIP, err);
        } else {
            const int kpt = N2 <= 4096 ? 4 : (N2 == 8192 ? 8 : 16);
            const dim3 blk(N2 / kpt);
            if (kpt == 4) hipLaunchKernelGGL(k_sort<4>, dim3(F), blk, (size_t)N2 * 8, h->st, ids_g, B_g, F, h->n_rows, N2, sl.rec, sl.owner_cnt, nullptr, 0, h->err_flag);
            else if (kpt == 8) hipLaunchKernelGGL(k_sort<8>, dim3(F), blk, (size_t)N2 * 8, h->st, ids_g, B_g, F, h->n_rows, N2, sl.rec, sl.owner_cnt, nullptr, 0, h->err_flag);
            else hipLaunchKernelGGL(k_sort<16>, dim3(F), blk, (size_t)N2 * 8, h->st, ids_g, B_g, F, h->n_rows, N2, sl.rec, sl.owner_cnt, nullptr, 0, h->err_flag);
        }
    }
    {
        ProfScope ps(h, "scatter_global", h->st);
        ScatArgs sa = make_scat_args(h, sl, N2);
        sa.gxp = gxp_g;
        hipLaunchKernelGGL(k_scat1, dim3((unsigned)(((size_t)F * N2 + 255) / 256)), dim3(256), 0, h->st, sa);
        hipLaunchKernelGGL(k_scat2, dim3(N2 > 4096 ? 256 : 64), dim3(256), 0, h->st, sa);
    }
    HIPCHK(h, hipGetLastError());
    h->next_ids = nullptr; h->next_B = 0;       // no grouping of a later batch rides on this step
    h->scatter_pending = false;
    return FNN_OK;
}

// ---- RCCL, opened at run time: librccl.so.1 is already in the process under PyTorch-ROCm (RTLD_NOLOAD finds that copy: one
// RCCL per process), otherwise the loader's search path / this library's RUNPATH (/opt/rocm) finds it.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;

int rccl_load(std::string& err)
{
    static std::mutex mu;                          // handles of several host threads may set up data parallelism at once
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.lib) return FNN_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { err = std::string("cannot open librccl.so.1: ") + dlerror(); return FNN_ERR_HIP; }
    RcclApi a; a.lib = lib;
#define RSYM(f) do { a.f = reinterpret_cast<decltype(a.f)>(dlsym(lib, "nccl" #f)); if (!a.f) { err = "librccl: symbol nccl" #f " not found"; return FNN_ERR_HIP; } } while (0)
    RSYM(GetUniqueId); RSYM(CommInitRank); RSYM(AllReduce); RSYM(AllGather); RSYM(CommDestroy); RSYM(GetErrorString);
#undef RSYM
    g_rccl = a;
    return FNN_OK;
}

int rccl_allreduce_cb(void* ctx, float* buf, int64_t n, void* stream)
{
    fnn_handle* h = static_cast<fnn_handle*>(ctx);
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, h->comm, (hipStream_t)stream);
    if (r != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r); return -1; }
    return 0;
}
int rccl_allgather_cb(void* ctx, const void* send, void* recv, int64_t bytes, void* stream)
{
    fnn_handle* h = static_cast<fnn_handle*>(ctx);
    const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)bytes, ncclInt8, h->comm, (hipStream_t)stream);
    if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return -1; }
    return 0;
}

void p2p_release(fnn_handle* h)
{
    for (int r = 0; r < 8; ++r) {
        if (h->peer_opened[r] && h->peer[r]) hipIpcCloseMemHandle(h->peer[r]);
        h->peer[r] = nullptr; h->peer_opened[r] = false;
    }
    if (h->xr) hipFree(h->xr);
    if (h->p2p_wait_max) { hipFree(h->p2p_wait_max); h->p2p_wait_max = nullptr; }
    h->xr = nullptr; h->xr_kind = 0; h->p2p_attached = false; h->dp_step_no = 0;
    if (h->dp_collective == FNN_DP_COLLECTIVE_P2P) h->dp_collective = FNN_DP_COLLECTIVE_CALLBACK;
}

int dp_setup(fnn_handle* h, int rank, int world, int sparse_mode)
{
    if (world < 1 || rank < 0 || rank >= world) FAIL(h, FNN_ERR_ARG, "fnn_dp_init: rank / world out of range");
    if (sparse_mode != FNN_DP_SPARSE_LOCAL && sparse_mode != FNN_DP_SPARSE_EXCHANGE) FAIL(h, FNN_ERR_ARG, "fnn_dp_init: bad sparse_mode");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_dp_init inside a step");
    if (sparse_mode == FNN_DP_SPARSE_EXCHANGE) {
        if ((int64_t)world * h->ldT > GLOBAL_BATCH_MAX) FAIL(h, FNN_ERR_ARG, "FNN_DP_SPARSE_EXCHANGE: world * max_batch (rounded up to 256) must be <= 32768");
        if (!(h->fused && mlp_shape_ok(h)) || h->Bmax > SORT_N) FAIL(h, FNN_ERR_ARG, "FNN_DP_SPARSE_EXCHANGE runs on the three-launch path only (max_batch <= 4096, hidden sizes the strip kernel is built for)");
        int rc;
        const size_t rows = (size_t)h->ldT;
        if (!h->xg_ids_send && (rc = alloc_dev(h, &h->xg_ids_send, rows * h->F)) != FNN_OK) return rc;
        if (h->xg_ids) { hipFree(h->xg_ids); h->xg_ids = nullptr; }
        if (h->xg_gxp) { hipFree(h->xg_gxp); h->xg_gxp = nullptr; }
        if ((rc = alloc_dev(h, &h->xg_ids, rows * world * h->F)) != FNN_OK) return rc;
        if ((rc = alloc_dev(h, &h->xg_gxp, rows * world * h->K1p)) != FNN_OK) return rc;
        if (!h->bag && (rc = ensure_global_ws(h, (int)rows * world)) != FNN_OK) return rc;
    }
    h->dp_rank = rank; h->dp_world = world; h->dp_sparse = sparse_mode;
    h->sorted_ids = nullptr; h->next_ids = nullptr;
    h->dp_payload = FNN_DP_PAYLOAD_SLABS; h->dp_collective = FNN_DP_COLLECTIVE_CALLBACK;
    if (const char* ev = getenv("FNN_DP_PAYLOAD")) {
        if (!strcmp(ev, "bucket")) h->dp_payload = FNN_DP_PAYLOAD_BUCKET;
        else if (strcmp(ev, "slabs") != 0) FAIL(h, FNN_ERR_ARG, "FNN_DP_PAYLOAD must be slabs or bucket");
    }
    return FNN_OK;
}

int check_ready(fnn_handle* h, int B) {
    if (!h) return FNN_ERR_ARG;
    if (B <= 0 || B > h->Bmax) FAIL(h, FNN_ERR_ARG, "B must be in [1, max_batch]");
    if (!h->table16) FAIL(h, FNN_ERR_STATE, "fnn_set_table has not been called");
    if (!(h->dense_set[0] && h->dense_set[1] && h->dense_set[2]))
        FAIL(h, FNN_ERR_STATE, "fnn_set_dense has not been called for all three layers");
    return FNN_OK;
}

}  // namespace

extern "C" {

const char* fnn_version(void) { return "fnn_hip 0.2 (gfx950)"; }

uint64_t fnn_cfg_size(void) { return (uint64_t)sizeof(fnn_cfg); }

const char* fnn_last_error(const fnn_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int fnn_create(const fnn_cfg* cfg, fnn_handle** out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return FNN_ERR_ARG; }
    *out = nullptr;
    if (cfg->n_fields < 2 || cfg->n_fields > 64) { g_create_err = "n_fields must be in [2, 64]"; return FNN_ERR_ARG; }
    if (cfg->mode != FNN_MODE_FM && cfg->mode != FNN_MODE_BAG) { g_create_err = "bad mode"; return FNN_ERR_ARG; }
    if (cfg->mode == FNN_MODE_BAG && (cfg->h0 < 192 || cfg->h0 > 316 || cfg->h0 % 4 != 0)) {
        g_create_err = "FNN_MODE_BAG: h0 must be a multiple of 4 in [192, 316] (the strip kernel is built for bag rows of 256 or 320 padded floats; the reference uses 200 and 300, python/SNN_RBM.py:25,53)"; return FNN_ERR_ARG; }
    if (cfg->mode == FNN_MODE_FM && (cfg->k < 1 || cfg->k > 15)) { g_create_err = "k = rank+1 must be in [1, 15] (two pad slots of the 16-float row carry w_0 and the bias)"; return FNN_ERR_ARG; }
    if (cfg->hidden1 < 1 || cfg->hidden1 > 4095 || cfg->hidden2 < 1 || cfg->hidden2 > 255) { g_create_err = "hidden1 must be in [1, 4095], hidden2 in [1, 255]"; return FNN_ERR_ARG; }
    if (cfg->max_batch < 1 || cfg->max_batch > 16384) { g_create_err = "max_batch must be in [1, 16384] (per-field LDS sort)"; return FNN_ERR_ARG; }
    if (cfg->precision != FNN_PREC_F32 && cfg->precision != FNN_PREC_BF16 && cfg->precision != FNN_PREC_BF16X3) { g_create_err = "bad precision"; return FNN_ERR_ARG; }
    if (cfg->act < 0 || cfg->act > 2) { g_create_err = "bad act"; return FNN_ERR_ARG; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device: ") + hipGetErrorString(e) + " (libfnn_hip.so has no CPU fallback)";
        return FNN_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "device ordinal out of range"; return FNN_ERR_ARG; }
    fnn_handle* h = new fnn_handle();
    h->cfg = *cfg; h->dev = cfg->device;
    auto fail = [&](int code) { g_create_err = h->err; fnn_destroy(h); return code; };
#define CK(expr) do { int rc_ = (expr); if (rc_ != FNN_OK) return fail(rc_); } while (0)
#define HK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e2_); return fail(FNN_ERR_HIP); } } while (0)
    HK(hipSetDevice(h->dev));
    if (cfg->stream) { h->st = (hipStream_t)cfg->stream; h->own_stream = false; }
    else { HK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); h->own_stream = true; }
    h->F = cfg->n_fields; h->K = cfg->k; h->H1 = cfg->hidden1; h->H2 = cfg->hidden2;
    h->xdim = 1 + h->F * h->K;
    h->K1p = rup(h->F * SLOT, 64); h->H1p = rup(h->H1 + 1, 64); h->H2p = rup(h->H2 + 1, 64);
    h->bag = cfg->mode == FNN_MODE_BAG;
    if (const char* e = getenv("FNN_WT_STORES")) h->wt_stores = atoi(e);
    if (const char* e = getenv("FNN_STEP1_WAVES")) h->step1_waves = atoi(e) == 4 ? 4 : 8;
    if (h->bag) { h->rw = cfg->h0; h->K = cfg->h0; h->xdim = cfg->h0; h->K1p = rup(cfg->h0 + 1, 64); }
    h->Bmax = cfg->max_batch; h->ldT = rup(h->Bmax, 256);
    h->N2max = SORT_N; while (h->N2max < h->Bmax) h->N2max <<= 1;    // the three-launch path groups SORT_N slots per field whatever max_batch is
    h->n1 = (size_t)h->K1p * h->H1p; h->n2 = (size_t)h->H1p * h->H2p;
    h->nw12 = h->n1 + h->n2; h->nw = h->nw12 + h->H2p;
    h->bf16 = cfg->precision == FNN_PREC_BF16; h->split = cfg->precision == FNN_PREC_BF16X3;
    if (const char* ev = getenv("FNN_NO_FUSE")) h->fused = !(ev[0] == '1');
    if (const char* ev = getenv("FNN_ROLE_OFF")) h->role_off = atoi(ev);
    if (const char* ev = getenv("FNN_SPLITK")) { const int v = atoi(ev); if (v == 2 || v == 4 || v == 8 || v == 16) h->splitk = v; }   // tuning knob
    if (const char* ev = getenv("FNN_SCAT2_WGS")) { const int v = atoi(ev); if (v >= 16 && v <= 1024) h->scat2_wgs = v; }
    const size_t ts = tsize(h), Ba = h->ldT;
    CK(alloc_dev(h, &h->master, h->nw));
    h->off_bag = h->nw12 + (size_t)h->H2p * 64;
    h->nbag = h->bag ? (size_t)h->K1p : 0;
    CK(alloc_dev(h, &h->bucket, h->nw + h->nbag));
    h->nslab = h->off_bag + h->nbag * 64;
    CK(alloc_dev(h, &h->slab, (size_t)h->splitk * h->nslab));
    CK(alloc_dev(h, (char**)&h->w1, h->n1 * ts));   CK(alloc_dev(h, (char**)&h->w1t, h->n1 * ts));
    CK(alloc_dev(h, (char**)&h->w2, h->n2 * ts));   CK(alloc_dev(h, (char**)&h->w2t, h->n2 * ts));
    CK(alloc_dev(h, (char**)&h->xp, Ba * h->K1p * ts));  CK(alloc_dev(h, (char**)&h->xpT, Ba * h->K1p * ts));
    CK(alloc_dev(h, (char**)&h->d1, Ba * h->H1p * ts));  CK(alloc_dev(h, (char**)&h->d1T, Ba * h->H1p * ts));
    CK(alloc_dev(h, (char**)&h->dl1, Ba * h->H1p * ts)); CK(alloc_dev(h, (char**)&h->dl1T, Ba * h->H1p * ts));
    CK(alloc_dev(h, (char**)&h->d2, Ba * h->H2p * ts));
    CK(alloc_dev(h, (char**)&h->dl2, Ba * h->H2p * ts)); CK(alloc_dev(h, (char**)&h->dl2T, Ba * h->H2p * ts));
    CK(alloc_dev(h, &h->gxp, Ba * h->K1p));
    CK(alloc_dev(h, &h->p_buf, Ba));
    CK(alloc_dev(h, (char**)&h->d2T, Ba * h->H2p * ts));
    CK(alloc_dev(h, (char**)&h->dl3T, Ba * 64 * ts));
    CK(alloc_dev(h, &h->loss_t, Ba));
    CK(alloc_dev(h, &h->loss_dev, (size_t)1));
    for (auto& sl : h->slot) {
        CK(alloc_dev(h, &sl.rec, (size_t)h->F * h->N2max));
        const size_t nchunk = h->bag ? (size_t)h->N2max / WCH : (size_t)h->N2max / 16;
        CK(alloc_dev(h, &sl.part, (size_t)h->F * nchunk * 2 * h->rw));
        CK(alloc_dev(h, &sl.owners, (size_t)h->F * nchunk));
        CK(alloc_dev(h, &sl.owner_cnt, (size_t)1));
    }
    if (h->bag) {
        if (!mlp_shape_ok(h)) { h->err = "FNN_MODE_BAG needs hidden sizes the strip kernel is built for (hidden1 257..319 with hidden2 65..127, e.g. 300/100; h0 <= 252 also <=63/<=63)"; return fail(FNN_ERR_ARG); }
        CK(alloc_dev(h, &h->bb0, (size_t)h->K1p));
        CK(alloc_dev(h, (char**)&h->dlxT, Ba * h->K1p * ts));
        CK(alloc_dev(h, (char**)&h->onesT, Ba * 64 * ts));
        CK(alloc_dev(h, &h->gx_raw, Ba * h->K1p));
        // onesT: fragment-tiled [64][ldT] matrix whose row 0 is all ones (column sums via MFMA)
        std::vector<unsigned char> ones(Ba * 64 * ts, 0);
        for (size_t t = 0; t < Ba; ++t) {
            if (h->bf16) { const unsigned short one = 0x3F80; memcpy(&ones[ft_off<bf16_t>(0, (int)t, (int)Ba) * 2], &one, 2); }
            else if (h->split) { const bs16_t one(1.0f); memcpy(&ones[ft_off<bs16_t>(0, (int)t, (int)Ba) * 4], &one, 4); }
            else { const float one = 1.0f; memcpy(&ones[ft_off<float>(0, (int)t, (int)Ba) * 4], &one, 4); }
        }
        // alloc_dev zeroes with hipMemsetAsync on the handle's (non-blocking) stream, which a null-stream hipMemcpy does not
        // wait for: without this sync the zeroing could land AFTER the copy (seen as a bag bias that never moved, one run in a few)
        HK(hipStreamSynchronize(h->st));
        HK(hipMemcpy(h->onesT, ones.data(), ones.size(), hipMemcpyHostToDevice));
    }
    h->key64 = true;                                                  // refined when the table is set
    CK(alloc_dev(h, (char**)&h->skeys, (size_t)h->F * SORT_N * 8));
    h->cpow_cap = h->N2max + 1;
    CK(alloc_dev(h, &h->cpow_dev, (size_t)h->cpow_cap));
    CK(alloc_dev(h, &h->err_flag, (size_t)1));
    CK(alloc_dev(h, &h->ones_u8, (size_t)(h->H1p + h->H2p), false));
    HK(hipMemsetAsync(h->ones_u8, 1, (size_t)(h->H1p + h->H2p), h->st));
    CK(alloc_dev(h, &h->st_ids, (size_t)h->Bmax * h->F));
    CK(alloc_dev(h, &h->st_y, (size_t)h->Bmax));
    CK(alloc_dev(h, &h->st_m1, (size_t)h->H1p)); CK(alloc_dev(h, &h->st_m2, (size_t)h->H2p));
    CK(alloc_dev(h, &h->st_p, (size_t)h->Bmax));
    CK(alloc_dev(h, &h->st_x, (size_t)h->Bmax * h->xdim));
    // dynamic LDS beyond the 64 KB default, set once and for every size a later call can ask for (round-2 advisor: the workspace
    // of a global batch grows monotonically, so a smaller batch after a larger one used to launch k_sort<8> without its attribute;
    // with the kernel's static counter k_sort<8> at 8192 keys is 65,540 bytes)
    HK(hipFuncSetAttribute((const void*)k_sort<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    HK(hipFuncSetAttribute((const void*)k_sort<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    if (!h->bag) {
        const size_t gl = (size_t)GR_EX * (h->xdim + h->F) * sizeof(float);          // 65,600 bytes at F = 64, K = 15
        if (gl > 160 * 1024) { h->err = "n_fields * k too large for the reference-layout gather tile"; return fail(FNN_ERR_ARG); }
        HK(hipFuncSetAttribute((const void*)k_gather_ref, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(gl, (size_t)65536)));
    }
    HK(hipStreamSynchronize(h->st));
#undef CK
#undef HK
    *out = h;
    return FNN_OK;
}

int fnn_destroy(fnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    hipSetDevice(h->dev);
    if (h->st) hipStreamSynchronize(h->st);
    if (h->comm && h->dp_own_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    p2p_release(h);
    for (void* q : {(void*)h->tag_first, (void*)h->slot[0].tag_shared, (void*)h->slot[1].tag_shared, (void*)h->shadow_dev, (void*)h->xg_ids_send, (void*)h->xg_ids, (void*)h->xg_gxp, (void*)h->gsl.rec, (void*)h->gsl.part, (void*)h->gsl.owners,
                    (void*)h->gsl.owner_cnt, h->gws}) if (q) hipFree(q);
    for (auto& kv : h->prof_slots) for (auto& p : kv.second.ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    void* ptrs[] = {h->table16, h->field_of_row, h->master, h->bucket, h->slab, h->w1, h->w1t, h->w2, h->w2t,
                    h->xp, h->xpT, h->d1, h->d1T, h->d2, h->dl2, h->dl2T, h->dl1, h->dl1T, h->gxp, h->p_buf,
                    h->loss_t, h->d2T, h->dl3T, h->loss_dev, h->cpow_dev, h->err_flag, h->ones_u8, h->skeys, h->bb0, h->dlxT, h->onesT, h->gx_raw,
                    h->st_ids, h->st_y, h->st_m1, h->st_m2, h->st_p, h->st_x};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& sl : h->slot) {
        for (void* p : {(void*)sl.rec, (void*)sl.part, (void*)sl.owners, (void*)sl.owner_cnt}) if (p) hipFree(p);
    }
    if (h->own_stream && h->st) hipStreamDestroy(h->st);
    delete h;
    return FNN_OK;
}

int fnn_set_hparams(fnn_handle* h, float lr, float lambda1, float lambda_fm)
{
    if (!h) return FNN_ERR_ARG;
    h->cfg.lr = lr; h->cfg.lambda1 = lambda1; h->cfg.lambda_fm = lambda_fm;
    return FNN_OK;
}

void* fnn_stream(fnn_handle* h) { return h ? (void*)h->st : nullptr; }

int fnn_sync(fnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->dev));
    return check_async(h);
}

int fnn_set_table(fnn_handle* h, const float* rows, int64_t n_rows, const int32_t* field_of_row,
                  float w0, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (!rows || n_rows <= 0 || n_rows >= (1ll << 31)) FAIL(h, FNN_ERR_ARG, "rows null or n_rows out of range");
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (h->table16) { hipFree(h->table16); h->table16 = nullptr; }
    if (h->field_of_row) { hipFree(h->field_of_row); h->field_of_row = nullptr; }
    HIPCHK(h, hipMalloc((void**)&h->table16, (size_t)n_rows * h->rw * sizeof(float)));
    HIPCHK(h, hipMalloc((void**)&h->field_of_row, (size_t)n_rows * sizeof(int32_t)));
    const size_t nbytes = (size_t)n_rows * h->K * sizeof(float);
    const float* src = rows; float* tmp = nullptr;
    if (memkind == FNN_MEM_HOST) {
        HIPCHK(h, hipMalloc((void**)&tmp, nbytes));
        HIPCHK(h, hipMemcpy(tmp, rows, nbytes, hipMemcpyHostToDevice));
        src = tmp;
    }
    const size_t n = (size_t)n_rows * h->rw;
    hipLaunchKernelGGL(k_pack_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, src, n_rows, h->K, h->rw, h->table16);
    if (field_of_row)
        HIPCHK(h, hipMemcpyAsync(h->field_of_row, field_of_row, (size_t)n_rows * sizeof(int32_t),
                                 memkind == FNN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->st));
    else
        HIPCHK(h, hipMemsetAsync(h->field_of_row, 0, (size_t)n_rows * sizeof(int32_t), h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (tmp) hipFree(tmp);
    if (h->bag) {
        for (int** q : {&h->tag_first, &h->slot[0].tag_shared, &h->slot[1].tag_shared}) {
            if (*q) { hipFree(*q); *q = nullptr; }
            HIPCHK(h, hipMalloc((void**)q, (size_t)n_rows * sizeof(int)));
            HIPCHK(h, hipMemset(*q, 0, (size_t)n_rows * sizeof(int)));
        }
        h->tag_stamp = 0;
    }
    h->n_rows = n_rows; h->w0 = w0;
    h->key64 = (unsigned long long)n_rows * SORT_N > 0xFFFFFFFFull;     // else 32-bit (row << 12 | t) keys
    h->sorted_ids = nullptr; h->next_ids = nullptr;
    return FNN_OK;
}

static int get_rows_impl(fnn_handle* h, const int64_t* row_ids, int64_t n, float* out, int memkind)
{
    if (!h->table16) FAIL(h, FNN_ERR_STATE, "fnn_set_table has not been called");
    if (!out || n <= 0) FAIL(h, FNN_ERR_ARG, "out null or n <= 0");
    HIPCHK(h, hipSetDevice(h->dev));
    const size_t cnt = (size_t)n * h->K;
    const int64_t* ids_dev = row_ids; float* out_dev = out;
    int64_t* tmp_ids = nullptr; float* tmp_out = nullptr;
    if (memkind == FNN_MEM_HOST) {
        if (row_ids) {
            HIPCHK(h, hipMalloc((void**)&tmp_ids, (size_t)n * sizeof(int64_t)));
            HIPCHK(h, hipMemcpy(tmp_ids, row_ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
            ids_dev = tmp_ids;
        }
        HIPCHK(h, hipMalloc((void**)&tmp_out, cnt * sizeof(float)));
        out_dev = tmp_out;
    }
    hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->st, h->table16,
                       ids_dev, n, h->n_rows, h->K, h->rw, out_dev, h->err_flag);
    int rc = FNN_OK;
    if (memkind == FNN_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(out, tmp_out, cnt * sizeof(float), hipMemcpyDeviceToHost, h->st));
        rc = check_async(h);
        if (tmp_ids) hipFree(tmp_ids);
        hipFree(tmp_out);
    }
    return rc;
}

int fnn_get_table(fnn_handle* h, float* rows_out, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    return get_rows_impl(h, nullptr, h->n_rows, rows_out, memkind);
}

int fnn_get_rows(fnn_handle* h, const int64_t* row_ids, int64_t n, float* out, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (!row_ids) FAIL(h, FNN_ERR_ARG, "row_ids null");
    return get_rows_impl(h, row_ids, n, out, memkind);
}

// Reference shapes <-> padded slot layout.  Host staging; called rarely.
int fnn_set_dense(fnn_handle* h, int layer, const float* W, const float* b, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (layer < 1 || layer > 3 || !W || !b) FAIL(h, FNN_ERR_ARG, "layer must be 1..3, W and b non-null");
    HIPCHK(h, hipSetDevice(h->dev));
    const int F = h->F, K = h->K, H1 = h->H1, H2 = h->H2, H1p = h->H1p, H2p = h->H2p;
    const size_t nW = layer == 1 ? (size_t)h->xdim * H1 : layer == 2 ? (size_t)H1 * H2 : (size_t)H2;
    const size_t nb = layer == 1 ? H1 : layer == 2 ? H2 : 1;
    std::vector<float> hw(nW), hb(nb);
    if (memkind == FNN_MEM_HOST) { memcpy(hw.data(), W, nW * 4); memcpy(hb.data(), b, nb * 4); }
    else {
        HIPCHK(h, hipMemcpy(hw.data(), W, nW * 4, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(hb.data(), b, nb * 4, hipMemcpyDeviceToHost));
    }
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (layer == 1) {
        std::vector<float> p(h->n1, 0.f);
        if (h->bag) {                                                    // W [h0][H1]; b1 on the ones column h0
            for (int i = 0; i < h->rw; ++i) memcpy(&p[(size_t)i * H1p], &hw[(size_t)i * H1], H1 * 4);
            memcpy(&p[(size_t)h->rw * H1p], hb.data(), H1 * 4);
        } else {
            for (int f = 0; f < F; ++f)
                for (int l = 0; l < K; ++l)
                    memcpy(&p[(size_t)(f * SLOT + l) * H1p], &hw[(size_t)(1 + f * K + l) * H1], H1 * 4);
            memcpy(&p[(size_t)K * H1p], &hw[0], H1 * 4);                 // w1[0,:] rides on the w_0 slot
            memcpy(&p[(size_t)(SLOT + K) * H1p], hb.data(), H1 * 4);     // b1 rides on the ones slot
        }
        HIPCHK(h, hipMemcpy(h->master, p.data(), h->n1 * 4, hipMemcpyHostToDevice));
    } else if (layer == 2) {
        std::vector<float> p(h->n2, 0.f);
        for (int i = 0; i < H1; ++i) memcpy(&p[(size_t)i * H2p], &hw[(size_t)i * H2], H2 * 4);
        memcpy(&p[(size_t)H1 * H2p], hb.data(), H2 * 4);
        HIPCHK(h, hipMemcpy(h->master + h->n1, p.data(), h->n2 * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<float> p(H2p, 0.f);
        memcpy(p.data(), hw.data(), H2 * 4); p[H2] = hb[0];
        HIPCHK(h, hipMemcpy(h->master + h->nw12, p.data(), (size_t)H2p * 4, hipMemcpyHostToDevice));
    }
    BY_PREC(h, launch_update, (h, nullptr, 0.f));
    HIPCHK(h, hipStreamSynchronize(h->st));
    h->dense_set[layer - 1] = true;
    return FNN_OK;
}

int fnn_get_dense(fnn_handle* h, int layer, float* W, float* b, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (layer < 1 || layer > 3 || !W || !b) FAIL(h, FNN_ERR_ARG, "layer must be 1..3, W and b non-null");
    HIPCHK(h, hipSetDevice(h->dev));
    const int F = h->F, K = h->K, H1 = h->H1, H2 = h->H2, H1p = h->H1p, H2p = h->H2p;
    std::vector<float> m(h->nw);
    HIPCHK(h, hipStreamSynchronize(h->st));
    HIPCHK(h, hipMemcpy(m.data(), h->master, h->nw * 4, hipMemcpyDeviceToHost));
    const size_t nW = layer == 1 ? (size_t)h->xdim * H1 : layer == 2 ? (size_t)H1 * H2 : (size_t)H2;
    const size_t nb = layer == 1 ? H1 : layer == 2 ? H2 : 1;
    std::vector<float> hw(nW), hb(nb);
    if (layer == 1) {
        if (h->bag) {
            for (int i = 0; i < h->rw; ++i) memcpy(&hw[(size_t)i * H1], &m[(size_t)i * H1p], H1 * 4);
            memcpy(hb.data(), &m[(size_t)h->rw * H1p], H1 * 4);
        } else {
            memcpy(&hw[0], &m[(size_t)K * H1p], H1 * 4);
            for (int f = 0; f < F; ++f)
                for (int l = 0; l < K; ++l)
                    memcpy(&hw[(size_t)(1 + f * K + l) * H1], &m[(size_t)(f * SLOT + l) * H1p], H1 * 4);
            memcpy(hb.data(), &m[(size_t)(SLOT + K) * H1p], H1 * 4);
        }
    } else if (layer == 2) {
        const float* p = &m[h->n1];
        for (int i = 0; i < H1; ++i) memcpy(&hw[(size_t)i * H2], &p[(size_t)i * H2p], H2 * 4);
        memcpy(hb.data(), &p[(size_t)H1 * H2p], H2 * 4);
    } else {
        memcpy(hw.data(), &m[h->nw12], H2 * 4); hb[0] = m[h->nw12 + H2];
    }
    if (memkind == FNN_MEM_HOST) { memcpy(W, hw.data(), nW * 4); memcpy(b, hb.data(), nb * 4); }
    else {
        HIPCHK(h, hipMemcpy(W, hw.data(), nW * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(b, hb.data(), nb * 4, hipMemcpyHostToDevice));
    }
    return FNN_OK;
}

int fnn_set_bag_bias(fnn_handle* h, const float* bb0, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (!h->bag || !bb0) FAIL(h, FNN_ERR_ARG, "fnn_set_bag_bias: FNN_MODE_BAG handle and non-null pointer required");
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->st));
    HIPCHK(h, hipMemcpy(h->bb0, bb0, (size_t)h->rw * 4, memkind == FNN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
    return FNN_OK;
}

int fnn_get_bag_bias(fnn_handle* h, float* bb0_out, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (!h->bag || !bb0_out) FAIL(h, FNN_ERR_ARG, "fnn_get_bag_bias: FNN_MODE_BAG handle and non-null pointer required");
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->st));
    HIPCHK(h, hipMemcpy(bb0_out, h->bb0, (size_t)h->rw * 4, memkind == FNN_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return FNN_OK;
}

int fnn_gather(fnn_handle* h, const int32_t* ids, int B, float* x_out, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    // any B: the reference's get_batch_data serves 100,000-line evaluation chunks (python/FNN_wnzh.py:193-221); device pointers
    // go through one launch, host pointers through the max_batch-sized staging buffers chunk by chunk
    if (B <= 0 || (size_t)B * h->xdim > (size_t)1 << 40) FAIL(h, FNN_ERR_ARG, "B must be positive");
    if (!h->table16) FAIL(h, FNN_ERR_STATE, "fnn_set_table has not been called");
    if (!ids || !x_out) FAIL(h, FNN_ERR_ARG, "null pointer");
    HIPCHK(h, hipSetDevice(h->dev));
    const bool host = memkind == FNN_MEM_HOST;
    for (int lo = 0; lo < B; lo += host ? h->Bmax : B) {
        const int nb = host ? std::min(h->Bmax, B - lo) : B;
        const int32_t* ids_dev = ids + (size_t)lo * h->F; float* x_dev = x_out + (size_t)lo * h->xdim;
        if (host) {
            HIPCHK(h, hipMemcpyAsync(h->st_ids, ids + (size_t)lo * h->F, (size_t)nb * h->F * 4, hipMemcpyHostToDevice, h->st));
            ids_dev = h->st_ids; x_dev = h->st_x;
        }
        const size_t n = (size_t)nb * h->xdim;
        {
            ProfScope ps(h, "gather_ref", h->st);
            if (h->bag)
                hipLaunchKernelGGL(k_bag_ref, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, h->st, ids_dev, nb, h->F,
                                   h->rw, h->table16, h->n_rows, h->bb0, x_dev, h->err_flag, (h->wt_stores & 16) != 0);   // (measured: 1 % slower written through -- off)
            else
                hipLaunchKernelGGL(k_gather_ref, dim3((unsigned)((nb + GR_EX - 1) / GR_EX)), dim3(256), (size_t)GR_EX * (h->xdim + h->F) * sizeof(float), h->st, ids_dev, nb, h->F,
                                   h->K, h->table16, h->n_rows, h->w0, x_dev, h->err_flag, (h->wt_stores & 8) != 0);
        }
        if (host) {
            HIPCHK(h, hipMemcpyAsync(x_out + (size_t)lo * h->xdim, x_dev, n * 4, hipMemcpyDeviceToHost, h->st));
            const int rc = check_async(h);              // (synchronises: the staging buffers are free for the next chunk)
            if (rc != FNN_OK) return rc;
        }
    }
    return FNN_OK;
}

static int step_impl(fnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* mask1,
                     const uint8_t* mask2, int b_size, float* p_out, float* gx_out, int memkind,
                     bool inline_update)
{
    int rc = check_ready(h, B);
    if (rc != FNN_OK) return rc;
    if (!ids || !y || !mask1 || !mask2) FAIL(h, FNN_ERR_ARG, "ids, y, mask1, mask2 must be non-null");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_step_begin called twice without fnn_step_end");
    HIPCHK(h, hipSetDevice(h->dev));
    if (b_size <= 0) b_size = B;
    rc = update_cpow(h, b_size, B);
    if (rc != FNN_OK) return rc;
    const int32_t* ids_d = ids; const float* y_d = y; const uint8_t *m1 = mask1, *m2 = mask2;
    float* p_d = p_out; float* gx_d = gx_out;
    if (memkind == FNN_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(h->st_ids, ids, (size_t)B * h->F * 4, hipMemcpyHostToDevice, h->st));
        HIPCHK(h, hipMemcpyAsync(h->st_y, y, (size_t)B * 4, hipMemcpyHostToDevice, h->st));
        HIPCHK(h, hipMemcpyAsync(h->st_m1, mask1, (size_t)h->H1, hipMemcpyHostToDevice, h->st));
        HIPCHK(h, hipMemcpyAsync(h->st_m2, mask2, (size_t)h->H2, hipMemcpyHostToDevice, h->st));
        ids_d = h->st_ids; y_d = h->st_y; m1 = h->st_m1; m2 = h->st_m2;
        if (p_out) p_d = h->st_p;
        if (gx_out) gx_d = h->st_x;
        h->sorted_ids = nullptr; h->next_ids = nullptr;     // staging buffers are reused: no carried grouping
    }
    if (h->n_shadow > 0 && h->bag) FAIL(h, FNN_ERR_ARG, "fnn_set_shadowed: FNN_MODE_FM only");
    if (h->n_shadow > 0 && h->dp && inline_update && h->dp_sparse == FNN_DP_SPARSE_EXCHANGE)
        FAIL(h, FNN_ERR_ARG, "shadowed features are not carried through FNN_DP_SPARSE_EXCHANGE");
    // shadowed features: the layer-by-layer path, whose per-field sort has room for the extra keys
    const bool fast = h->fused && mlp_shape_ok(h) && B <= SORT_N && h->n_shadow == 0;
    h->update_pending = !(fast && inline_update);
    h->step_native_dp = inline_update && h->dp;
    h->step_bsize = b_size;
    if (fast)
        rc = BY_PREC_RC(h, run_step_fast, (h, ids_d, y_d, B, m1, m2, p_d, gx_d, inline_update));
    else
        rc = BY_PREC_RC(h, run_step, (h, ids_d, y_d, B, m1, m2, true, p_d, gx_d));
    h->n_shadow = 0;                                            // consumed (also by a failing step)
    if (rc != FNN_OK) return rc;
    if (memkind == FNN_MEM_HOST) {
        if (p_out) HIPCHK(h, hipMemcpyAsync(p_out, p_d, (size_t)B * 4, hipMemcpyDeviceToHost, h->st));
        if (gx_out) HIPCHK(h, hipMemcpyAsync(gx_out, gx_d, (size_t)B * h->xdim * 4, hipMemcpyDeviceToHost, h->st));
        if (p_out || gx_out) HIPCHK(h, hipStreamSynchronize(h->st));
    }
    h->in_step = true; h->step_B = B;
    return FNN_OK;
}

int fnn_step_begin(fnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* mask1,
                   const uint8_t* mask2, int b_size, float* p_out, float* gx_out, int memkind)
{
    return step_impl(h, ids, y, B, mask1, mask2, b_size, p_out, gx_out, memkind, false);
}

int fnn_set_shadowed(fnn_handle* h, const int32_t* tfr, int n, int memkind)
{
    if (!h) return FNN_ERR_ARG;
    if (n < 0 || (n > 0 && !tfr)) FAIL(h, FNN_ERR_ARG, "fnn_set_shadowed: null list or n < 0");
    if (h->bag) FAIL(h, FNN_ERR_ARG, "fnn_set_shadowed: FNN_MODE_FM only");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_set_shadowed inside a step");
    HIPCHK(h, hipSetDevice(h->dev));
    if (n > h->shadow_cap) {
        HIPCHK(h, hipStreamSynchronize(h->st));
        if (h->shadow_dev) { hipFree(h->shadow_dev); h->shadow_dev = nullptr; h->shadow_cap = 0; }
        const int cap = std::max(1024, n);
        HIPCHK(h, hipMalloc((void**)&h->shadow_dev, (size_t)cap * 3 * sizeof(int32_t)));
        h->shadow_cap = cap;
    }
    if (n > 0) {
        if (!h->table16) FAIL(h, FNN_ERR_STATE, "fnn_set_shadowed before fnn_set_table");
        HIPCHK(h, hipMemcpyAsync(h->shadow_dev, tfr, (size_t)n * 3 * sizeof(int32_t),
                                 memkind == FNN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->st));
        // a row belongs to ONE field (fnn_set_table's field_of_row): the sparse-row update groups keys per field, so an entry that
        // names a row under another field would put the row into two groups of one launch -- two unordered read-modify-writes.
        // Checked here (a launch and a read of the flag; this call is rare), not left to show as a lost update one run in four.
        hipLaunchKernelGGL(k_check_shadowed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->shadow_dev, n, h->field_of_row,
                           h->n_rows, h->F, h->err_flag);
        int flag = 0;
        HIPCHK(h, hipMemcpyAsync(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipStreamSynchronize(h->st));                       // (also: the caller's host buffer may go away)
        if (flag & 8) {
            HIPCHK(h, hipMemsetAsync(h->err_flag, 0, sizeof(int), h->st));
            h->n_shadow = 0;
            FAIL(h, FNN_ERR_ARG, "fnn_set_shadowed: an entry's row is outside the table or does not belong to the entry's field (field_of_row)");
        }
    }
    h->n_shadow = n;
    return FNN_OK;
}

int fnn_prefetch_ids(fnn_handle* h, const int32_t* ids, int B)
{
    int rc = check_ready(h, B);
    if (rc != FNN_OK) return rc;
    if (!ids) FAIL(h, FNN_ERR_ARG, "ids null");
    h->next_ids = ids; h->next_B = B;       // grouped by the next step's first launch
    return FNN_OK;
}

int fnn_step_scatter(fnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    if (!h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_step_scatter without fnn_step_begin");
    HIPCHK(h, hipSetDevice(h->dev));
    if (h->scatter_pending) {
        BY_PREC(h, launch_steps23, (h, false, true, false));
        HIPCHK(h, hipGetLastError());
    }
    return FNN_OK;
}

int fnn_sparse_grad(fnn_handle* h, float** dev_ptr, int64_t* row_floats)
{
    if (!h || !dev_ptr || !row_floats) return FNN_ERR_ARG;
    if (h->bag) FAIL(h, FNN_ERR_ARG, "fnn_sparse_grad: FNN_MODE_FM only");
    *dev_ptr = h->gxp; *row_floats = h->K1p;
    return FNN_OK;
}

int fnn_step_scatter_global(fnn_handle* h, const int32_t* ids_g, const float* gxp_g, int B_g)
{
    if (!h) return FNN_ERR_ARG;
    if (!h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_step_scatter_global without fnn_step_begin");
    if (h->bag) FAIL(h, FNN_ERR_ARG, "fnn_step_scatter_global: FNN_MODE_FM only");
    if (!ids_g || !gxp_g || B_g < 1 || B_g > GLOBAL_BATCH_MAX) FAIL(h, FNN_ERR_ARG, "ids_g / gxp_g null or B_g outside [1, 32768]");
    HIPCHK(h, hipSetDevice(h->dev));
    return scatter_global_impl(h, ids_g, gxp_g, B_g);
}

int fnn_dp_unique_id(void* id128_out)
{
    if (!id128_out) { g_create_err = "fnn_dp_unique_id: null pointer"; return FNN_ERR_ARG; }
    int rc = rccl_load(g_create_err);
    if (rc != FNN_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return FNN_ERR_HIP; }
    memcpy(id128_out, &id, sizeof(id));
    return FNN_OK;
}

int fnn_dp_shutdown(fnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_dp_shutdown inside a step");
    hipSetDevice(h->dev);
    if (h->st) hipStreamSynchronize(h->st);
    if (h->comm && h->dp_own_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->dp_own_comm = false;
    h->dp = false; h->dp_allreduce = nullptr; h->dp_allgather = nullptr; h->dp_ctx = nullptr;
    h->dp_rank = 0; h->dp_world = 1; h->dp_sparse = FNN_DP_SPARSE_LOCAL;
    p2p_release(h);
    h->dp_payload = FNN_DP_PAYLOAD_SLABS; h->dp_collective = FNN_DP_COLLECTIVE_CALLBACK;
    return FNN_OK;
}

int fnn_dp_set_payload(fnn_handle* h, int payload)
{
    if (!h) return FNN_ERR_ARG;
    if (payload != FNN_DP_PAYLOAD_SLABS && payload != FNN_DP_PAYLOAD_BUCKET) FAIL(h, FNN_ERR_ARG, "fnn_dp_set_payload: bad payload");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_dp_set_payload inside a step");
    h->dp_payload = payload;
    return FNN_OK;
}

int fnn_dp_p2p_export(fnn_handle* h, void* handle64_out, int same_process)
{
    if (!h || !handle64_out) return FNN_ERR_ARG;
    if (!h->dp) FAIL(h, FNN_ERR_STATE, "fnn_dp_p2p_export before fnn_dp_init / fnn_dp_init_custom");
    if (h->dp_world > 8) FAIL(h, FNN_ERR_ARG, "the p2p all-reduce serves up to 8 ranks (one node)");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_dp_p2p_export inside a step");
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->st));
    p2p_release(h);
    h->xr_nbp = (size_t)rup((int)(h->nw + h->nbag), 64);
    h->xr_bytes = 2 * h->xr_nbp * sizeof(float) + 8 * 64;
    // memory that stays coherent between devices: uncached, else fine-grained ($FNN_P2P_REGION = uncached | finegrained | plain
    // picks one; plain hipMalloc only where the ranks share a process and a device)
    const char* want = getenv("FNN_P2P_REGION");
    void* q = nullptr; h->xr_kind = 0;
    if ((!want || !strcmp(want, "uncached")) && hipExtMallocWithFlags(&q, h->xr_bytes, hipDeviceMallocUncached) == hipSuccess) h->xr_kind = 1;
    else if ((!want || !strcmp(want, "finegrained")) && hipExtMallocWithFlags(&q, h->xr_bytes, hipDeviceMallocFinegrained) == hipSuccess) h->xr_kind = 2;
    else if (((!want && same_process) || (want && !strcmp(want, "plain"))) && hipMalloc(&q, h->xr_bytes) == hipSuccess) h->xr_kind = 3;
    (void)hipGetLastError();
    if (!h->xr_kind) FAIL(h, FNN_ERR_NOMEM, "fnn_dp_p2p_export: no uncached / fine-grained device memory for the exchange region");
    h->xr = static_cast<float*>(q);
    HIPCHK(h, hipMemset(h->xr, 0, h->xr_bytes));
    if (!h->p2p_wait_max) HIPCHK(h, hipMalloc((void**)&h->p2p_wait_max, 8));
    HIPCHK(h, hipMemset(h->p2p_wait_max, 0, 8));
    if (const char* ev2 = getenv("FNN_P2P_TIMEOUT_MS")) { const long long ms = atoll(ev2); if (ms > 0) h->p2p_timeout_ticks = (unsigned long long)ms * 100000ull; }
    h->xr_same_process = same_process != 0;
    memset(handle64_out, 0, 64);
    if (same_process) memcpy(handle64_out, &h->xr, sizeof(void*));
    else {
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
        hipIpcMemHandle_t mh;
        HIPCHK(h, hipIpcGetMemHandle(&mh, h->xr));
        memcpy(handle64_out, &mh, 64);
    }
    return FNN_OK;
}

int fnn_dp_p2p_attach(fnn_handle* h, const void* handles, int same_process)
{
    if (!h || !handles) return FNN_ERR_ARG;
    if (!h->dp || !h->xr) FAIL(h, FNN_ERR_STATE, "fnn_dp_p2p_attach before fnn_dp_p2p_export");
    if ((same_process != 0) != h->xr_same_process) FAIL(h, FNN_ERR_ARG, "fnn_dp_p2p_attach: same_process differs from fnn_dp_p2p_export");
    HIPCHK(h, hipSetDevice(h->dev));
    const char* hs = static_cast<const char*>(handles);
    for (int r = 0; r < h->dp_world; ++r) {
        if (r == h->dp_rank) { h->peer[r] = h->xr; continue; }
        if (same_process) { void* q; memcpy(&q, hs + 64 * r, sizeof(void*)); h->peer[r] = static_cast<float*>(q); continue; }
        hipIpcMemHandle_t mh; memcpy(&mh, hs + 64 * r, 64);
        void* q = nullptr;
        HIPCHK(h, hipIpcOpenMemHandle(&q, mh, hipIpcMemLazyEnablePeerAccess));
        h->peer[r] = static_cast<float*>(q); h->peer_opened[r] = true;
    }
    for (int r = 0; r < h->dp_world; ++r) if (!h->peer[r]) FAIL(h, FNN_ERR_ARG, "fnn_dp_p2p_attach: a null region");
    h->p2p_attached = true; h->dp_step_no = 0;
    return FNN_OK;
}

int fnn_dp_set_collective(fnn_handle* h, int collective)
{
    if (!h) return FNN_ERR_ARG;
    if (collective != FNN_DP_COLLECTIVE_CALLBACK && collective != FNN_DP_COLLECTIVE_P2P) FAIL(h, FNN_ERR_ARG, "fnn_dp_set_collective: bad collective");
    if (h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_dp_set_collective inside a step");
    if (collective == FNN_DP_COLLECTIVE_P2P && !(h->dp && h->p2p_attached)) FAIL(h, FNN_ERR_STATE, "FNN_DP_COLLECTIVE_P2P needs fnn_dp_p2p_export + fnn_dp_p2p_attach on every rank first");
    h->dp_collective = collective;
    return FNN_OK;
}

int fnn_dp_p2p_max_wait_us(fnn_handle* h, double* us_out)
{
    if (!h || !us_out) return FNN_ERR_ARG;
    *us_out = 0.0;
    if (!h->p2p_wait_max) return FNN_OK;
    HIPCHK(h, hipSetDevice(h->dev));
    unsigned long long t = 0;
    HIPCHK(h, hipMemcpyAsync(&t, h->p2p_wait_max, 8, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    *us_out = (double)t / 100.0;
    return FNN_OK;
}

int fnn_dp_get_config(fnn_handle* h, int* payload, int* collective, int* region_kind)
{
    if (!h) return FNN_ERR_ARG;
    if (payload) *payload = dp_bucket_mode(h) ? FNN_DP_PAYLOAD_BUCKET : FNN_DP_PAYLOAD_SLABS;
    if (collective) *collective = h->dp_collective;
    if (region_kind) *region_kind = h->xr_kind;
    return FNN_OK;
}

int fnn_dp_init(fnn_handle* h, int rank, int world, const void* id128, int sparse_mode)
{
    if (!h) return FNN_ERR_ARG;
    if (!id128) FAIL(h, FNN_ERR_ARG, "fnn_dp_init: null unique id");
    if (h->dp) { int rc = fnn_dp_shutdown(h); if (rc != FNN_OK) return rc; }
    HIPCHK(h, hipSetDevice(h->dev));
    int rc = rccl_load(h->err);
    if (rc != FNN_OK) return rc;
    rc = dp_setup(h, rank, world, sparse_mode);
    if (rc != FNN_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    HIPCHK(h, hipStreamSynchronize(h->st));
    const ncclResult_t r = g_rccl.CommInitRank(&h->comm, world, id, rank);
    if (r != ncclSuccess) { h->comm = nullptr; FAIL(h, FNN_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    h->dp_own_comm = true;
    h->dp_allreduce = rccl_allreduce_cb; h->dp_allgather = rccl_allgather_cb; h->dp_ctx = h;
    h->dp = true;
    return FNN_OK;
}

int fnn_dp_init_custom(fnn_handle* h, int rank, int world, fnn_allreduce_fn allreduce, fnn_allgather_fn allgather, void* ctx,
                       int sparse_mode)
{
    if (!h) return FNN_ERR_ARG;
    if (!allreduce || (sparse_mode == FNN_DP_SPARSE_EXCHANGE && !allgather)) FAIL(h, FNN_ERR_ARG, "fnn_dp_init_custom: null callback");
    if (h->dp) { int rc = fnn_dp_shutdown(h); if (rc != FNN_OK) return rc; }
    HIPCHK(h, hipSetDevice(h->dev));
    int rc = dp_setup(h, rank, world, sparse_mode);
    if (rc != FNN_OK) return rc;
    h->dp_allreduce = allreduce; h->dp_allgather = allgather; h->dp_ctx = ctx;
    h->dp = true;
    return FNN_OK;
}

int fnn_dense_grad_bucket(fnn_handle* h, float** dev_ptr, int64_t* n_floats)
{
    if (!h || !dev_ptr || !n_floats) return FNN_ERR_ARG;
    *dev_ptr = h->bucket; *n_floats = (int64_t)(h->nw + h->nbag);
    return FNN_OK;
}

int fnn_step_end(fnn_handle* h, float* loss_sum_out)
{
    if (!h) return FNN_ERR_ARG;
    if (!h->in_step) FAIL(h, FNN_ERR_STATE, "fnn_step_end without fnn_step_begin");
    HIPCHK(h, hipSetDevice(h->dev));
    if (h->scatter_pending) { int rc = fnn_step_scatter(h); if (rc != FNN_OK) return rc; }
    if (h->update_pending) {
        if (h->step_native_dp && dp_bucket_mode(h)) {     // layer-by-layer path under native data parallelism, bucket payload
            int rc = BY_PREC_RC(h, dp_finish_bucket, (h, false));
            if (rc != FNN_OK) { h->in_step = false; return rc; }
        } else {                                           // (slabs payload: k_reduce already summed GLOBAL slabs)
            ProfScope ps(h, "update", h->st);
            BY_PREC(h, launch_update, (h, h->bucket, h->cfg.lr));
        }
    }
    HIPCHK(h, hipGetLastError());
    h->in_step = false;
    if (loss_sum_out) return fnn_last_loss(h, loss_sum_out);
    return FNN_OK;
}

int fnn_last_loss(fnn_handle* h, float* loss_sum_out)
{
    if (!h || !loss_sum_out) return FNN_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipMemcpyAsync(loss_sum_out, h->loss_dev, sizeof(float), hipMemcpyDeviceToHost, h->st));
    return check_async(h);
}

int fnn_train_step(fnn_handle* h, const int32_t* ids, const float* y, int B, const uint8_t* mask1,
                   const uint8_t* mask2, int b_size, float* p_out, float* gx_out, int memkind,
                   float* loss_sum_out)
{
    int rc = step_impl(h, ids, y, B, mask1, mask2, b_size, p_out, gx_out, memkind, true);
    if (rc != FNN_OK) return rc;
    return fnn_step_end(h, loss_sum_out);
}

int fnn_predict(fnn_handle* h, const int32_t* ids, int B, float* p_out, int memkind)
{
    int rc = check_ready(h, B);
    if (rc != FNN_OK) return rc;
    if (!ids || !p_out) FAIL(h, FNN_ERR_ARG, "null pointer");
    HIPCHK(h, hipSetDevice(h->dev));
    const int32_t* ids_d = ids; float* p_d = p_out;
    if (memkind == FNN_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(h->st_ids, ids, (size_t)B * h->F * 4, hipMemcpyHostToDevice, h->st));
        ids_d = h->st_ids; p_d = h->st_p;
    }
    rc = BY_PREC_RC(h, run_step, (h, ids_d, nullptr, B, nullptr, nullptr, false, p_d, nullptr));
    if (rc != FNN_OK) return rc;
    HIPCHK(h, hipGetLastError());
    if (memkind == FNN_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(p_out, p_d, (size_t)B * 4, hipMemcpyDeviceToHost, h->st));
        return check_async(h);
    }
    return FNN_OK;
}

int fnn_eval(fnn_handle* h, const int32_t* ids, const int32_t* y, int64_t N, int memkind, double* auc, double* rmse,
             double* logloss, float* p_out)
{
    int rc = check_ready(h, 1);
    if (rc != FNN_OK) return rc;
    if (!ids || !y || N < 1) FAIL(h, FNN_ERR_ARG, "ids / y null or N < 1");
    HIPCHK(h, hipSetDevice(h->dev));
    const bool host = memkind == FNN_MEM_HOST;
    int32_t *ids_d = nullptr, *y_d = nullptr; float* p_d = nullptr;
    auto cleanup = [&]() { if (host) { if (ids_d) hipFree(ids_d); if (y_d) hipFree(y_d); } if (p_d && p_d != p_out) hipFree(p_d); };
#define EK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return FNN_ERR_HIP; } } while (0)
    if (host) {
        EK(hipMalloc((void**)&ids_d, (size_t)N * h->F * 4)); EK(hipMalloc((void**)&y_d, (size_t)N * 4));
        EK(hipMemcpyAsync(ids_d, ids, (size_t)N * h->F * 4, hipMemcpyHostToDevice, h->st));
        EK(hipMemcpyAsync(y_d, y, (size_t)N * 4, hipMemcpyHostToDevice, h->st));
    } else { ids_d = const_cast<int32_t*>(ids); y_d = const_cast<int32_t*>(y); }
    if (p_out && !host) p_d = p_out; else EK(hipMalloc((void**)&p_d, (size_t)N * 4));
    for (int64_t lo = 0; lo < N; lo += h->Bmax) {
        const int B = (int)std::min<int64_t>(h->Bmax, N - lo);
        rc = BY_PREC_RC(h, run_step, (h, ids_d + lo * h->F, nullptr, B, nullptr, nullptr, false, p_d + lo, nullptr));
        if (rc != FNN_OK) { cleanup(); return rc; }
    }
    EK(hipGetLastError());
    double out[4] = {0, 0, 0, 0};
    std::string merr;
    const int mrc = device_metrics(h->st, p_d, y_d, N, out, merr);
    if (mrc == -1) { h->err = merr; cleanup(); return FNN_ERR_HIP; }
    if (p_out && host) EK(hipMemcpy(p_out, p_d, (size_t)N * 4, hipMemcpyDeviceToHost));
#undef EK
    cleanup();
    if (auc) *auc = out[0];
    if (rmse) *rmse = out[1];
    if (logloss) *logloss = out[2];
    rc = check_async(h);
    if (rc != FNN_OK) return rc;
    if (mrc == -2) FAIL(h, FNN_ERR_RANGE, merr);
    return FNN_OK;
}

int fnn_prof_enable(fnn_handle* h, int on) { if (!h) return FNN_ERR_ARG; h->prof = on != 0; return FNN_OK; }

static int prof_collect(fnn_handle* h)
{
    HIPCHK(h, hipStreamSynchronize(h->st));
    for (auto& kv : h->prof_slots) {
        for (auto& p : kv.second.ev) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { kv.second.ms += ms; kv.second.n += 1; }
            hipEventDestroy(p.first); hipEventDestroy(p.second);
        }
        kv.second.ev.clear();
    }
    return FNN_OK;
}

int fnn_prof_reset(fnn_handle* h)
{
    if (!h) return FNN_ERR_ARG;
    int rc = prof_collect(h);
    h->prof_slots.clear();
    return rc;
}

int fnn_prof_get(fnn_handle* h, const char* which, double* avg_ms, int64_t* launches)
{
    if (!h || !which || !avg_ms) return FNN_ERR_ARG;
    int rc = prof_collect(h);
    if (rc != FNN_OK) return rc;
    auto it = h->prof_slots.find(which);
    if (it == h->prof_slots.end() || it->second.n == 0) { *avg_ms = 0.0; if (launches) *launches = 0; return FNN_OK; }
    *avg_ms = it->second.ms / (double)it->second.n;
    if (launches) *launches = it->second.n;
    return FNN_OK;
}

}  // extern "C"
