// fnn_kernels.hip.h -- device code of libfnn_hip.so (gfx950 / CDNA4 only).
//
// Data layout in HBM:
//   table16  float [n_rows][16]   FM rows padded 44 B -> 64 B (one aligned line piece per row;
//                                 slots >= K stay 0).  python/FNN_wnzh.py:76 feat_weights.
//   x'       T [Ba][K1p]          layer-one array in "slot" layout: column 16*f + l holds
//                                 row(ids[t][f])[l]; column K (a pad slot of field 0) holds w_0
//                                 (the reference's x[0], python/FNN_wnzh.py:93) and column 16+K
//                                 holds the constant 1 that turns b1 into a row of W1p.
//   W1p      [K1p][H1p], W2p [H1p][H2p], w3p [H2p]: dense tensors padded the same way, the
//                                 bias of each layer stored as the weight row of the "ones"
//                                 column of its input (b1 = W1p[16+K], b2 = W2p[H1], b3 = w3p[H2]).
//   Global MFMA operands (weight shadows in both orientations, and the transposed activations
//   that feed the weight-gradient products) are stored FRAGMENT-TILED: the 16 rows x KS k-values
//   one MFMA fragment covers are 1 KiB contiguous in lane order (ft_off below), so a wave's
//   fragment load is one fully coalesced 16 B/lane read instead of 16 separate 64-B row pieces.
//
// All matrix products run on the matrix cores: v_mfma_f32_16x16x32_bf16 (FNN_PREC_BF16) or the
// exact-f32 v_mfma_f32_16x16x4_f32 (FNN_PREC_F32, the parity mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fnn {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SLOT = 16;            // padded row width (floats) of the FM table
constexpr int ACT_TANH = 0, ACT_SIGMOID = 1, ACT_LINEAR = 2;
constexpr double FIX_SCALE = 17592186044416.0;  // 2^44: fixed-point scale of the scatter sums

// FNN_PREC_BF16X3: every operand element is a PAIR of bf16 values, hi = bf16(v) and lo = bf16(v - hi) (16 significant bits),
// and a product is three bf16 MFMAs, hi*hi + hi*lo + lo*hi with f32 accumulation -- 4.4x the rate of the exact-f32 MFMA on this
// part (tools/exp/mma_split_bench.hip: 144 against 640 ticks per five 16x16 tiles of k = 16) at an error 430x below plain bf16's
// (6.7e-5 against 2.9e-2 on a K = 256 dot product of uniform values whose f32-MFMA error is 9.5e-6).  The element is 4 bytes, so
// every layout of the f32 mode (EPL = 4, KS = 16, ft_off, LDS tiles) holds unchanged; only `mma` differs.
__host__ __device__ inline unsigned bs_pack(const float v) {              // hi in the low half, lo in the high half
    const bf16_t h = (bf16_t)v;
    const bf16_t l = (bf16_t)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | (unsigned)__builtin_bit_cast(unsigned short, l) << 16;
}
__host__ __device__ inline float bs_unpack(const unsigned u) {
    return __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xffff0000u);
}
struct alignas(4) bs16_t {
    unsigned short hi, lo;
    bs16_t() = default;
    __host__ __device__ explicit bs16_t(float v) { const unsigned u = bs_pack(v); hi = (unsigned short)(u & 0xffffu); lo = (unsigned short)(u >> 16); }
    __host__ __device__ explicit operator float() const { return bs_unpack((unsigned)hi | (unsigned)lo << 16); }
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Traits;
template <> struct Traits<bs16_t> {
    typedef u32x4 frag;             // four elements: dword = hi | lo << 16
    static constexpr int EPL = 4;
    static constexpr int KS = 16;
};
template <> struct Traits<float> {
    typedef f32x4 frag;
    static constexpr int EPL = 4;   // elements per lane per fragment (16 B)
    static constexpr int KS = 16;   // k covered by one fragment pair
};
template <> struct Traits<bf16_t> {
    typedef bf16x8 frag;
    static constexpr int EPL = 8;
    static constexpr int KS = 32;
};

// One fragment pair -> accumulate.  Operand maps (cdna_hip_programming.md section 3):
//   16x16x32 bf16: lane l holds A[row l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][col l&15].
//   16x16x4  f32 : lane l holds A[row l&15][k = l>>4],        B[k = l>>4][col l&15]; here each
//   lane loads 4 consecutive k (k = 4*(l>>4) + m) and MFMA m consumes element m of both
//   operands -- a permutation of k shared by A and B, so the sum over k is complete.
__device__ inline void mma(f32x4& acc, const bf16x8 a, const bf16x8 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ inline void mma(f32x4& acc, const f32x4 a, const f32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// bs16_t: the same lane map as the f32 form (lane l holds k = 4*(l>>4) + j, j < 4) is the operand map of
// v_mfma_f32_16x16x16_bf16; the hi and lo planes of a fragment are two v_perm_b32 each.  The small terms are added first.
__device__ inline void mma(f32x4& acc, const u32x4 a, const u32x4 b) {
    union { unsigned u[2]; s16x4 s; } ah, al, bh, bl;
    ah.u[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100); ah.u[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100);
    al.u[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302); al.u[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302);
    bh.u[0] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100); bh.u[1] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100);
    bl.u[0] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302); bl.u[1] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al.s, bh.s, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah.s, bl.s, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah.s, bh.s, acc, 0, 0, 0);
}

// One A fragment against C B fragments (C independent accumulators).  The bf16-pair form issues its three terms TERM-major --
// the small terms of all C tiles, then the large one -- so that an MFMA never waits for the one before it (at one wave per
// SIMD the three dependent MFMAs of `mma` cost 56 ticks per tile, 29 when other work sits between them).
template <int C, typename F>
__device__ __forceinline__ void mma_row(f32x4 (&acc)[C], const F a, const F (&b)[C]) {
#pragma unroll
    for (int i = 0; i < C; ++i) mma(acc[i], a, b[i]);
}
template <int C>
__device__ __forceinline__ void mma_row(f32x4 (&acc)[C], const u32x4 a, const u32x4 (&b)[C]) {
    union P { unsigned u[2]; s16x4 s; };
    P ah, al, bh[C], bl[C];
    ah.u[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100); ah.u[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100);
    al.u[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302); al.u[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        bh[i].u[0] = __builtin_amdgcn_perm(b[i][1], b[i][0], 0x05040100); bh[i].u[1] = __builtin_amdgcn_perm(b[i][3], b[i][2], 0x05040100);
        bl[i].u[0] = __builtin_amdgcn_perm(b[i][1], b[i][0], 0x07060302); bl[i].u[1] = __builtin_amdgcn_perm(b[i][3], b[i][2], 0x07060302);
    }
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al.s, bh[i].s, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah.s, bl[i].s, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah.s, bh[i].s, acc[i], 0, 0, 0);
}

// Two k-steps of the bf16-pair form at once: the hi (lo) planes of fragments kk and kk + 1 side by side are an operand of
// v_mfma_f32_16x16x32_bf16 (which k a lane's eight values stand for does not matter as long as A and B agree), at twice the
// FLOP rate of the 16x16x16 form -- the matrix pipe's share of the strip kernel was 4.8 us of 27 (profiles/r02d_pmc_mfma_*).
template <int C>
__device__ __forceinline__ void mma_row2(f32x4 (&acc)[C], const u32x4 a0, const u32x4 a1, const u32x4 (&b0)[C], const u32x4 (&b1)[C]) {
    union P8 { unsigned u[4]; bf16x8 v; };
    P8 ah, al, bh[C], bl[C];
    ah.u[0] = __builtin_amdgcn_perm(a0[1], a0[0], 0x05040100); ah.u[1] = __builtin_amdgcn_perm(a0[3], a0[2], 0x05040100);
    ah.u[2] = __builtin_amdgcn_perm(a1[1], a1[0], 0x05040100); ah.u[3] = __builtin_amdgcn_perm(a1[3], a1[2], 0x05040100);
    al.u[0] = __builtin_amdgcn_perm(a0[1], a0[0], 0x07060302); al.u[1] = __builtin_amdgcn_perm(a0[3], a0[2], 0x07060302);
    al.u[2] = __builtin_amdgcn_perm(a1[1], a1[0], 0x07060302); al.u[3] = __builtin_amdgcn_perm(a1[3], a1[2], 0x07060302);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        bh[i].u[0] = __builtin_amdgcn_perm(b0[i][1], b0[i][0], 0x05040100); bh[i].u[1] = __builtin_amdgcn_perm(b0[i][3], b0[i][2], 0x05040100);
        bh[i].u[2] = __builtin_amdgcn_perm(b1[i][1], b1[i][0], 0x05040100); bh[i].u[3] = __builtin_amdgcn_perm(b1[i][3], b1[i][2], 0x05040100);
        bl[i].u[0] = __builtin_amdgcn_perm(b0[i][1], b0[i][0], 0x07060302); bl[i].u[1] = __builtin_amdgcn_perm(b0[i][3], b0[i][2], 0x07060302);
        bl[i].u[2] = __builtin_amdgcn_perm(b1[i][1], b1[i][0], 0x07060302); bl[i].u[3] = __builtin_amdgcn_perm(b1[i][3], b1[i][2], 0x07060302);
    }
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, bh[i].v, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bl[i].v, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bh[i].v, acc[i], 0, 0, 0);
}

// Fragment-tiled operand layout for a [rows][Ktot] operand whose k index is the contraction index:
// element (row, k) lives at ((row/16 * Ktot/KS + k/KS) * 64 + lane) * EPL + k % EPL with
// lane = row%16 + 16 * ((k % KS) / EPL)  -- exactly the MFMA operand map above.
template <typename T> __host__ __device__ inline size_t ft_off(int row_, int k_, int Ktot_) {
    constexpr unsigned EPL = Traits<T>::EPL, KS = Traits<T>::KS;
    const unsigned row = (unsigned)row_, k = (unsigned)k_, Ktot = (unsigned)Ktot_;     // non-negative: shifts and masks, no sign fix-ups
    return ((size_t)((row >> 4) * (Ktot / KS) + k / KS) * 64 + (row & 15) + 16 * ((k % KS) / EPL)) * EPL + (k % EPL);
}
// pointer to the fragment of row-tile `rt`, k-step `kt` for lane `lane`
template <typename T> __device__ inline const T* ft_frag(const T* base, int rt, int kt, int nkt, int lane) {
    return base + ((size_t)(rt * nkt + kt) * 64 + lane) * Traits<T>::EPL;
}

// sum of x[tid], x[tid + 256], ... (< n) in that order, with 8 loads in flight at a time: a plain `v += x[i]` loop pays one
// memory round trip per element
__device__ inline float strided_sum256(const float* __restrict__ x, const int n) {
    float v = 0.f;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = i0 + 256 * k; t[k] = i < n ? x[i] : 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
    }
    return v;
}

__device__ inline void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ inline void store4(bf16_t* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ inline void store4(bs16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<u32x4*>(p) = u32x4{bs_pack(a), bs_pack(b), bs_pack(c), bs_pack(d)};
}
// The same stores written THROUGH the XCD's L2 (sc1): what a kernel leaves dirty in its L2s is written back at its end, before the
// next launch can start (MI355X_MICROARCH.md, cost cell `boundary`: + bytes / 6 TB/s); the strip kernel's 13.6 MB of transposed
// activations and gx' leave while it still computes instead (MlpArgs::wt; measured on one box each: 40.9 -> 39.6 us per step; the same
// treatment of the split-K slabs and of step 3's outputs changed nothing, of the updated table rows -- scalar 4-byte stores, and the
// next launch re-reads hot rows -- cost 1.8 us)
__device__ inline void store4_wt(float* p, float a, float b, float c, float d) {
    const f32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ inline void store4_wt(bf16_t* p, float a, float b, float c, float d) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    u2 w; __builtin_memcpy(&w, &v, 8);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
__device__ inline void store4_wt(bs16_t* p, float a, float b, float c, float d) {
    const u32x4 v = {bs_pack(a), bs_pack(b), bs_pack(c), bs_pack(d)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
template <typename V> __device__ inline void store16_wt(void* p, const V& v) {          // any 16-byte value
    static_assert(sizeof(V) == 16, "store16_wt: 16-byte values");
    u32x4 w; __builtin_memcpy(&w, &v, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
// the same with the policy as a (wave-uniform) argument: every launch that writes through can be run with plain stores, and a test
// holds the two forms bit-equal (a write-through store that the next launch does not see in time would show there)
template <typename V> __device__ inline void store16_sel(const bool wt, void* p, const V& v) {
    if (wt) store16_wt(p, v); else { u32x4 w; __builtin_memcpy(&w, &v, 16); *reinterpret_cast<u32x4*>(p) = w; }
}
template <typename P> __device__ inline void store4_sel(const bool wt, P* p, float a, float b, float c, float d) {
    if (wt) store4_wt(p, a, b, c, d); else store4(p, a, b, c, d);
}
__device__ inline void store1_wt(float* p, float a) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(a) : "memory"); }
// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also drains every global
// load and store in flight (s_waitcnt vmcnt(0)), which would serialise the strip kernel's
// weight prefetch and its activation stores behind each phase barrier; the waves of a strip
// exchange data through LDS alone, so lgkmcnt(0) + s_barrier is the ordering they need.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline float4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

__device__ inline float4 load4(const bs16_t* p) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);         // (plain integer unpacking: a bit_cast of a vector ELEMENT to the
    return make_float4(bs_unpack(v[0]), bs_unpack(v[1]), bs_unpack(v[2]), bs_unpack(v[3]));   // struct read element 0 four times)
}

// tanh for the bf16 mode: 1 - 2/(exp(2z)+1) with the hardware exp (abs. error ~1e-7, far below
// bf16's 2^-9); the f32 parity mode keeps libm's tanhf.
template <typename T> __device__ inline float tanh_t(float z) { return tanhf(z); }
template <> __device__ inline float tanh_t<bf16_t>(float z) {
    const float e = __expf(2.0f * fminf(fmaxf(z, -15.f), 15.f));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);     // v_exp + v_rcp, no IEEE divide
}
template <> __device__ inline float tanh_t<bs16_t>(float z) { return tanh_t<bf16_t>(z); }      // 1e-7 is below 2^-17 as well
// logistic function: hardware exp + reciprocal (rel. error ~1e-7; z clamped so exp stays finite)
__device__ inline float sigmoid_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-fminf(fmaxf(z, -80.f), 80.f)));
}
// Branch-free activations for the strip kernel.  tanh(z) = 2*sigmoid(2z) - 1, so one hardware
// exp + reciprocal serves tanh and sigmoid alike: act(z) = a * 1/(1 + exp(-k z)) + b, with
// (k, a, b) = (2, 2, -1) for tanh and (1, 1, 0) for sigmoid; linear bypasses it with a select.
// The derivative in terms of the output d is the polynomial c0 + c1 d + c2 d^2:
// tanh 1 - d^2, sigmoid d - d^2, linear 1.  (Abs. error of the forward form ~2e-7.)
struct ActCoef { float k, a, b, c0, c1, c2; bool lin; };
__device__ inline ActCoef act_coef(int act) {
    ActCoef c;
    c.lin = act == ACT_LINEAR;
    c.k = act == ACT_TANH ? 2.0f : 1.0f;
    c.a = act == ACT_TANH ? 2.0f : 1.0f;
    c.b = act == ACT_TANH ? -1.0f : 0.0f;
    c.c0 = act == ACT_SIGMOID ? 0.0f : 1.0f;
    c.c1 = act == ACT_SIGMOID ? 1.0f : 0.0f;
    c.c2 = act == ACT_LINEAR ? 0.0f : -1.0f;
    return c;
}
__device__ inline float act_apply(const ActCoef& c, float z) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-c.k * z));
    return c.lin ? z : fmaf(c.a, s, c.b);
}
__device__ inline float dact_apply(const ActCoef& c, float d) { return fmaf(fmaf(c.c2, d, c.c1), d, c.c0); }
template <typename T> __device__ inline float act_fn_t(float z, int act) {
    if (act == ACT_TANH) return tanh_t<T>(z);
    if (act == ACT_SIGMOID) return sigmoid_fast(z);
    return z;
}
__device__ inline float act_fn(float z, int act) {
    if (act == ACT_TANH) return tanhf(z);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-z));
    return z;
}
// derivative of the activation written in terms of its (possibly masked) output d
__device__ inline float dact_fn(float d, int act) {
    if (act == ACT_TANH) return 1.0f - d * d;
    if (act == ACT_SIGMOID) return d * (1.0f - d);
    return 1.0f;
}

// ------------------------------------------------------------------------------------------
// A3  embedding gather: ids [B][F] -> x' [Ba][K1p] and x'^T [K1p][ldT]
//     (python/FNN_wnzh.py:87-96 feats_to_layer_one_array, batched as :224-237).
// One thread = 4 consecutive examples x one field x one 16-byte quarter of the 64-B row, so the
// four lanes of a row read one aligned 64-B piece and both output layouts get 4-element stores.
// ------------------------------------------------------------------------------------------
template <typename T>
static __global__ __launch_bounds__(256) void k_gather(const int32_t* __restrict__ ids, int B, int Ba, int F,
                                                int K, const float* __restrict__ table16,
                                                int64_t n_rows, float w0, T* __restrict__ xp, int K1p,
                                                T* __restrict__ xpT, int ldT, int* __restrict__ err)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = gid & 3;
    const int f = (gid >> 2) % F;
    const int t0 = ((gid >> 2) / F) * 4;
    if (t0 >= Ba) return;
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + i;
        int64_t id = -1;
        if (t < B) {
            id = ids[(size_t)t * F + f];
            if (id < -1 || id >= n_rows) { atomicOr(err, 1); id = -1; }
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) r = *reinterpret_cast<const float4*>(table16 + (size_t)id * SLOT + 4 * q);
        v[i][0] = r.x; v[i][1] = r.y; v[i][2] = r.z; v[i][3] = r.w;
        if (t < B && q == (K >> 2)) {
            if (f == 0) v[i][K & 3] = w0;       // x[0] = w_0               (:93)
            if (f == 1) v[i][K & 3] = 1.0f;     // ones column: b1 is a row of W1p
        }
    }
    const int c0 = f * SLOT + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        store4(xp + (size_t)(t0 + i) * K1p + c0, v[i][0], v[i][1], v[i][2], v[i][3]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        store4(xpT + ft_off<T>(c0 + j, t0, ldT), v[0][j], v[1][j], v[2][j], v[3][j]);
}

// Reference-layout gather for fnn_gather(): x [B][1+F*K] float (python/FNN_wnzh.py:91-96).  A workgroup writes GR_EX
// examples; a thread owns one column (its field and slot are worked out once, not per element -- the first version spent its time
// in the two integer divisions of a flat index: 47 us for 100,000 examples) and keeps 8 examples' ids, then rows, in flight.
// The tile of GR_EX x xdim floats goes through LDS and leaves as 16-byte stores over the workgroup's flat range of x (rows of
// 177 floats start on a 16-byte boundary only every fourth row, the range of 16 rows always does).
constexpr int GR_EX = 16;
static __global__ __launch_bounds__(256) void k_gather_ref(const int32_t* __restrict__ ids, int B, int F, int K,
                             const float* __restrict__ table16, int64_t n_rows, float w0,
                             float* __restrict__ x, int* __restrict__ err, const bool wt)
{
    extern __shared__ __align__(16) unsigned char gr_smem[];
    float* sx = reinterpret_cast<float*>(gr_smem);                  // [GR_EX][xdim]
    const int xdim = 1 + F * K;
    const int t0 = blockIdx.x * GR_EX;
    int* sids = reinterpret_cast<int*>(sx + GR_EX * xdim);            // [GR_EX][F]: the strip's ids, read once (contiguous)
    for (int e = threadIdx.x; e < GR_EX * F; e += 256) {
        int id = -1;
        if (t0 + e / F < B) {
            id = ids[(size_t)t0 * F + e];
            if (id < -1 || id >= n_rows) { atomicOr(err, 1); id = -1; }
        }
        sids[e] = id;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < xdim; c += 256) {
        const int f = c ? (c - 1) / K : 0, l = c ? (c - 1) % K : 0;
#pragma unroll
        for (int e0 = 0; e0 < GR_EX; e0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int id = c ? sids[(e0 + u) * F + f] : -1;
                v[u] = id >= 0 ? table16[(size_t)id * SLOT + l] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sx[(e0 + u) * xdim + c] = c ? v[u] : w0;          // x[0] = w_0 (:93)
        }
    }
    __syncthreads();
    const int nex = min(GR_EX, B - t0);
    const size_t base = (size_t)t0 * xdim;                          // multiple of 16 floats: 16-byte aligned
    const int nfl = nex * xdim, n4 = nfl >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) store16_sel(wt, reinterpret_cast<float4*>(x + base) + i, reinterpret_cast<const float4*>(sx)[i]);   // (written through: the output is the caller's)
    for (int i = 4 * n4 + threadIdx.x; i < nfl; i += 256) x[base + i] = sx[i];
}

// gx' [B][K1p] (slot layout) -> gx [B][1+F*K] (what `train` returns, python/FNN_wnzh.py:179).
static __global__ void k_gx_ref(const float* __restrict__ gxp, int K1p, int B, int F, int K,
                         float* __restrict__ gx)
{
    const int xdim = 1 + F * K;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)B * xdim) return;
    const int t = (int)(gid / xdim), i = (int)(gid % xdim);
    int c = K;                                   // d cost / d x[0] lives in the w_0 slot
    if (i > 0) c = ((i - 1) / K) * SLOT + (i - 1) % K;
    gx[gid] = gxp[(size_t)t * K1p + c];
}

// ------------------------------------------------------------------------------------------
// MFMA GEMM  C[M][N] = A[M][K] * Bt[N][K]^T  with a fused epilogue.
// One wave owns a 16 x (16*NT) output strip and streams its operands straight from L2 into
// fragment registers (16 B per lane, k-contiguous in both operands; no LDS, no barriers).
// grid = (M/64, N/(16*NT), splitK); block = 4 waves = 4 consecutive 16-row strips.
// ------------------------------------------------------------------------------------------
template <typename T> struct EpiFwd {        // A4: act(z) * mask, ones column, both layouts
    T* out; int ld; T* outT; int ldT; const uint8_t* mask; int act; int H; int B;
    __device__ void operator()(int row0, int col, const f32x4& acc, int) const {
        float m = 0.f;
        if (col < H) m = mask ? (float)mask[col] : 1.0f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = (col < H) ? act_fn(acc[r], act) * m : (col == H ? 1.0f : 0.0f);
            v[r] = (row0 + r < B) ? x : 0.0f;
            out[(size_t)(row0 + r) * ld + col] = (T)v[r];
        }
        if (outT) store4(outT + ft_off<T>(col, row0, ldT), v[0], v[1], v[2], v[3]);
    }
};
template <typename T> struct EpiBwd {        // A5: delta = (delta_next * W^T) * mask * act'(d)
    T* out; int ld; T* outT; int ldT; const T* d; const uint8_t* mask; int act; int H; int B;
    __device__ void operator()(int row0, int col, const f32x4& acc, int) const {
        float m = 0.f;
        if (col < H) m = mask ? (float)mask[col] : 1.0f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dv = (float)d[(size_t)(row0 + r) * ld + col];
            float x = acc[r] * m * dact_fn(dv, act);
            v[r] = (row0 + r < B && col < H) ? x : 0.0f;
            out[(size_t)(row0 + r) * ld + col] = (T)v[r];
        }
        store4(outT + ft_off<T>(col, row0, ldT), v[0], v[1], v[2], v[3]);
    }
};
struct EpiF32 {                               // plain float output (gx', split-K slabs)
    static constexpr bool TILE = false;
    float* out; int ld; size_t zstride;
    __device__ void operator()(int row0, int col, const f32x4& acc, int z) const {
        float* o = out + (size_t)z * zstride;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(size_t)(row0 + r) * ld + col] = acc[r];
    }
};

// The split-K slabs of a grouped launch (k_gemm_group): the same values, but a wave's 16 x 64 block of four fragments goes
// through a wave-private LDS tile and leaves as 16-byte write-through stores of whole 256-byte row pieces (the MFMA layout has a
// lane on 4 rows of one column: sixteen 4-byte stores per fragment row tile, 64-byte pieces, and 30 MB left dirty in the L2s for
// the end of the launch to write back).  Needs gemm_f32w_lds() bytes of dynamic LDS and TN = 4.
struct EpiF32W {
    static constexpr bool TILE = false;
    float* out; int ld; size_t zstride; bool wt;
};
constexpr size_t gemm_f32w_lds() { return (size_t)4 * 16 * 68 * sizeof(float); }

template <typename T, int NT, typename Epi>
static __global__ __launch_bounds__(256) void k_gemm(const T* __restrict__ A, int lda,
                                              const T* __restrict__ Bft, int klen, Epi epi)
{
    typedef typename Traits<T>::frag frag;
    constexpr int EPL = Traits<T>::EPL, KS = Traits<T>::KS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    const int ct0 = blockIdx.y * NT;                  // first 16-column tile of this strip
    const T* ap = A + (size_t)(row0 + (lane & 15)) * lda + (lane >> 4) * EPL;
    const int nkt = klen / KS;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kt = 0; kt < nkt; ++kt) {
        const frag a = *reinterpret_cast<const frag*>(ap + kt * KS);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const frag b = *reinterpret_cast<const frag*>(ft_frag<T>(Bft, ct0 + n, kt, nkt, lane));
            mma(acc[n], a, b);
        }
    }
    // C/D map of the 16x16 shapes: col = lane & 15, row = 4*(lane >> 4) + reg.
    const int r0 = row0 + 4 * (lane >> 4);
#pragma unroll
    for (int n = 0; n < NT; ++n) epi(r0, (ct0 + n) * 16 + (lane & 15), acc[n], 0);
}

// Epilogue of a wave's TM x TN accumulator fragments (shared by k_gemm_ft and k_gemm_lds); `smem`: LDS of
// gemm_ft_lds<T, TN>() bytes for a tile epilogue (unused otherwise).
template <typename T, int TM, int TN, typename Epi>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], const int rt0, const int ct0, const int wave, const int lane,
                                              unsigned char* gemm_smem, const Epi& epi, const int z)
{
    const int rq = 4 * (lane >> 4), cl = lane & 15;              // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
    if constexpr (std::is_same<Epi, EpiF32W>::value) {
        static_assert(TN == 4, "EpiF32W: four fragments per row tile");
        float* tile = reinterpret_cast<float*>(gemm_smem) + (size_t)wave * 16 * 68;     // [16][64 + 4]: conflict-free both ways
        float* o = epi.out + (size_t)z * epi.zstride;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(rq + r) * 68 + n * 16 + cl] = acc[m][n][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 4, c4 = idx & 15;
                const float4 v = *reinterpret_cast<const float4*>(tile + row * 68 + 4 * c4);
                store16_sel(epi.wt, o + (size_t)((rt0 + m) * 16 + row) * epi.ld + ct0 * 16 + 4 * c4, v);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else
    if constexpr (Epi::TILE) {
        // Activation-like output in both fragment-tiled orientations.  xT (rows = columns of C,
        // k = rows of C): the lane's 4 accumulator rows are 4 consecutive k -> one 8-byte store.
        // xF (rows = rows of C, k = columns): 16 rows at a time go through a wave-private LDS tile and
        // leave as whole 1-KiB fragments (16 B per lane, fully coalesced) instead of 2-byte stores.
        typedef typename Traits<T>::frag frag_t;
        constexpr int EPL = Traits<T>::EPL, KS = Traits<T>::KS, LDP = 16 * TN + 8, KT = 16 * TN / KS;
        T* tile = reinterpret_cast<T*>(gemm_smem) + (size_t)wave * 16 * LDP;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                float v[4];
                const int r0 = (rt0 + m) * 16 + rq, col = (ct0 + n) * 16 + cl;
                epi.pre(r0, col, acc[m][n], v);
                if (epi.outT) store4(epi.outT + ft_off<T>(col, r0, epi.ldT), v[0], v[1], v[2], v[3]);
                if (epi.outF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile[(rq + r) * LDP + n * 16 + cl] = (T)v[r];
                }
            }
            if (epi.outF) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {
                    const frag_t f = *reinterpret_cast<const frag_t*>(tile + (lane & 15) * LDP + kk * KS + (lane >> 4) * EPL);
                    T* dst = epi.outF + ((size_t)((rt0 + m) * (epi.ld / KS) + (ct0 * 16) / KS + kk) * 64 + lane) * EPL;
                    *reinterpret_cast<frag_t*>(dst) = f;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) epi((rt0 + m) * 16 + rq, (ct0 + n) * 16 + cl, acc[m][n], z);
    }
}

// ------------------------------------------------------------------------------------------
// MFMA GEMM on two FRAGMENT-TILED operands:  C[M][N] = sum_k A[M][k] * B[N][k], both stored with
// ft_off (16 rows x KS k-values = one contiguous 1-KiB fragment in lane order), so every operand
// load of a wave is one coalesced 16 B/lane read straight into MFMA registers: no LDS, no
// barriers.  A wave owns a (16 TM) x (16 TN) block of C (64 x 64 at TM = TN = 4: 16 MFMAs per
// 8 fragment loads), a workgroup 2 x 2 waves; the next k-step's fragments are in flight while the
// current one multiplies.  grid = (ceil(M / 32 TM), ceil(N / 32 TN), splitK); rows / columns are
// multiples of 16 TM / 16 TN.  The deep stack of the inner-product family runs all three of its
// products per layer on it (forward, backward-data, weight gradient).
// ------------------------------------------------------------------------------------------
template <typename T, int TM, int TN, typename Epi>
__device__ __forceinline__ void gemm_ft_body(const T* __restrict__ A, const T* __restrict__ Bm, const int mt16, const int nt16,
                                             const int nkt_all, const int nkt, const Epi& epi, const int bx, const int by, const int bz)
{
    typedef typename Traits<T>::frag frag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt0 = (bx * 2 + (wave & 1)) * TM, ct0 = (by * 2 + (wave >> 1)) * TN;
    if (rt0 >= mt16 || ct0 >= nt16) return;
    const int kt0 = bz * nkt;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // three register stages: the fragments of k-steps kt + 1 and kt + 2 are in flight while kt multiplies
    // (one L2 round trip is ~2-3 k-steps of MFMA time at two waves per SIMD)
    frag a0[TM], b0[TN], a1[TM], b1[TN], a2[TM], b2[TN];
    auto load = [&](frag* a, frag* b, int kt) {
        if (kt >= nkt) return;
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = *reinterpret_cast<const frag*>(ft_frag<T>(A, rt0 + m, kt0 + kt, nkt_all, lane));
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const frag*>(ft_frag<T>(Bm, ct0 + n, kt0 + kt, nkt_all, lane));
    };
    auto mul = [&](const frag* a, const frag* b) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) mma(acc[m][n], a[m], b[n]);
    };
    load(a0, b0, 0);
    load(a1, b1, 1);
    for (int kt = 0; kt < nkt; kt += 3) {
        load(a2, b2, kt + 2);
        mul(a0, b0);
        if (kt + 1 >= nkt) break;
        load(a0, b0, kt + 3);
        mul(a1, b1);
        if (kt + 2 >= nkt) break;
        load(a1, b1, kt + 4);
        mul(a2, b2);
    }
    extern __shared__ __align__(16) unsigned char gemm_smem_ft[];
    gemm_epilogue<T, TM, TN, Epi>(acc, rt0, ct0, wave, lane, gemm_smem_ft, epi, bz);
}

template <typename T, int TM, int TN, typename Epi>
static __global__ __launch_bounds__(256) void k_gemm_ft(const T* __restrict__ A, const T* __restrict__ Bm, int mt16, int nt16,
                                                  int nkt_all, int nkt, Epi epi)
{
    gemm_ft_body<T, TM, TN, Epi>(A, Bm, mt16, nt16, nkt_all, nkt, epi, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several independent products with plain f32 split-K output (the weight gradients of a whole stack) in ONE launch:
// problem q owns workgroups wg0[q] .. wg0[q+1]-1, laid out (x fastest, then y, then the K split).
constexpr int GEMM_GROUP_MAX = 9;
struct GemmGroupProb { const void* A; const void* B; float* out; int mt16, nt16, nkt_all, nkt, ldo, gx, gy; };
struct GemmGroupArgs { GemmGroupProb p[GEMM_GROUP_MAX]; int wg0[GEMM_GROUP_MAX + 1]; int n; size_t zstride; int xcd; int wt; };   // wt: slabs written through

template <typename T, int TM, int TN>
static __global__ __launch_bounds__(256) void k_gemm_group(const GemmGroupArgs g)
{
    int q = 0;
#pragma unroll
    for (int i = 1; i < GEMM_GROUP_MAX; ++i) q += (i < g.n && (int)blockIdx.x >= g.wg0[i]) ? 1 : 0;
    const GemmGroupProb& pr = g.p[q];
    int local = (int)blockIdx.x - g.wg0[q];
    if (g.xcd) {
        // XCD-aware order inside a problem: hardware workgroup ids that differ by a multiple of 8 share an XCD; the n tiles of
        // the problem are dealt so that each of those 8 classes owns one contiguous run of logical tiles (x fastest: a run
        // shares its B panel and a few A panels).  Across problems the launch order (round robin) keeps the XCDs balanced.
        const int n = g.wg0[q + 1] - g.wg0[q], cls = local & 7, k = local >> 3, qd = n >> 3, rm = n & 7;
        local = (cls < rm ? cls * (qd + 1) : rm * (qd + 1) + (cls - rm) * qd) + k;
    }
    const int bx = local % pr.gx, by = (local / pr.gx) % pr.gy, bz = local / (pr.gx * pr.gy);
    const EpiF32W e{pr.out, pr.ldo, g.zstride, g.wt != 0};
    gemm_ft_body<T, TM, TN, EpiF32W>(static_cast<const T*>(pr.A), static_cast<const T*>(pr.B), pr.mt16, pr.nt16, pr.nkt_all, pr.nkt, e, bx, by, bz);
}

// dynamic LDS of k_gemm_ft for a tile epilogue: 4 wave-private [16][16 TN + 8] tiles
template <typename T, int TN> constexpr size_t gemm_ft_lds() { return (size_t)4 * 16 * (16 * TN + 8) * sizeof(T); }

// ------------------------------------------------------------------------------------------
// The same product, operands staged through LDS: a workgroup (2 x 2 waves, 64 x 64 of C each) owns
// 128 x 128 of C; one k-step of it is 8 + 8 fragments = 16 KiB, which the four waves fetch with four
// global_load_lds_dwordx4 each (a fragment is 1 KiB contiguous in global memory AND lane-linear in
// LDS, so the DMA needs no swizzle and ds_read_b128 of a fragment is conflict-free).  GEMM_NS k-steps
// are in flight in a ring of LDS slots (one L2 round trip is ~6 k-steps of MFMA time and these
// products are only 5-32 k-steps long, so the depth of the prefetch is what sets their speed); a
// wave's next fragments are read from LDS into a second register set while the current ones
// multiply.  One raw s_barrier per k-step with counted vmcnt waits (cdna_hip_programming.md
// section 5, "Pipelining across barriers"): a slot is read only after the wait that retires its
// loads AND a barrier, and re-filled only after a barrier that follows its last read.
// grid = (ceil(M / 128), ceil(N / 128), splitK); M, N multiples of 64.
// ------------------------------------------------------------------------------------------
constexpr int GEMM_NS = 8;
#ifdef GEMM_STAMPS                    // diagnostic builds only (tools/exp/gemm_bench.hip): per-workgroup phase time stamps
__device__ long long* g_gemm_dbg;
#define GEMM_STAMP(i) do { if (threadIdx.x == 0 && g_gemm_dbg) g_gemm_dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_STAMP(i) do { } while (0)
#endif
template <typename T, bool TILE> constexpr size_t gemm_lds_bytes() {
    return (size_t)GEMM_NS * 16 * 1024 + (TILE ? gemm_ft_lds<T, 4>() : 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    // gfx9 s_waitcnt: vmcnt = simm16[15:14] : simm16[3:0]; expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}
// wait until at most `groups` groups of 4 loads are outstanding (wave-uniform argument)
__device__ __forceinline__ void wait_groups(const int groups) {
    switch (groups) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<4>(); break;
        case 2: wait_vmcnt<8>(); break;
        case 3: wait_vmcnt<12>(); break;
        case 4: wait_vmcnt<16>(); break;
        case 5: wait_vmcnt<20>(); break;
        default: wait_vmcnt<24>(); break;
    }
}

template <typename T, typename Epi>
static __global__ __launch_bounds__(256) void k_gemm_lds(const T* __restrict__ A, const T* __restrict__ Bm, int mt16, int nt16,
                                                   int nkt_all, int nkt, Epi epi)
{
    typedef typename Traits<T>::frag frag;
    constexpr int EPL = Traits<T>::EPL, NS = GEMM_NS;
    static_assert(NS - 2 <= 6, "wait_groups covers up to 6 groups");
    extern __shared__ __align__(16) unsigned char gemm_smem_l[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = wave & 1, wc = wave >> 1;
    const int rt0 = (blockIdx.x * 2 + wr) * 4, ct0 = (blockIdx.y * 2 + wc) * 4;
    const int kt0 = blockIdx.z * nkt;
    // the four fragments this wave stages per k-step: waves 0, 1 -> A row tiles, waves 2, 3 -> B column tiles
    // (indices past the edge are clamped: they land in slots nobody multiplies)
    const T* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = (wave & 1) * 4 + j;
        const int tile = (wave < 2) ? min((int)blockIdx.x * 8 + f, mt16 - 1) : min((int)blockIdx.y * 8 + f, nt16 - 1);
        src[j] = ft_frag<T>(wave < 2 ? A : Bm, tile, kt0, nkt_all, lane);
    }
    auto stage = [&](const int kt) {                       // k-step kt -> slot kt % NS
        unsigned char* slot = gemm_smem_l + (size_t)(kt % NS) * 16384 + (size_t)wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)kt * 64 * EPL),
                                             (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, 0, 0);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag a0[4], b0[4], a1[4], b1[4];
    auto fetch = [&](frag* a, frag* b, const int kt) {     // this wave's fragments of k-step kt: LDS -> registers
        const unsigned char* slot = gemm_smem_l + (size_t)(kt % NS) * 16384 + lane * 16;
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const frag*>(slot + (wr * 4 + m) * 1024);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const frag*>(slot + 8192 + (wc * 4 + n) * 1024);
    };
    auto mul = [&](const frag* a, const frag* b) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) mma(acc[m][n], a[m], b[n]);
    };
    GEMM_STAMP(0);
    // prologue: k-steps 0 .. NS-2 in flight; k-step 0 landed, visible to all waves and in registers
    const int npro = min(NS - 1, nkt);
    for (int kt = 0; kt < npro; ++kt) stage(kt);
    wait_groups(npro - 1);
    __builtin_amdgcn_s_barrier();
    fetch(a0, b0, 0);
    GEMM_STAMP(1);
    // iteration kt: [wait k-step kt+1] [barrier: also, every wave has read slot kt] [refill the slot of kt-1... with
    // k-step kt+NS-1] [read kt+1 into the other register set] [multiply kt]
    auto iter = [&](const frag* ca, const frag* cb, frag* na, frag* nb, const int kt) {
        if (kt + 1 < nkt) {
            // issued so far: k-steps < min(kt + NS - 1, nkt); k-step kt + 1 must have landed
            wait_groups(min(kt + NS - 1, nkt) - (kt + 2));
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave's reads of slot kt are complete
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nkt) stage(kt + NS - 1);              // into the slot of k-step kt - 1 (read in iteration kt - 2)
            fetch(na, nb, kt + 1);
        }
        mul(ca, cb);
    };
    // steady state (k-step kt + NS - 1 exists): no branches, fixed counts; in pairs for the register ping-pong
    auto steady = [&](const frag* ca, const frag* cb, frag* na, frag* nb, const int kt) {
        wait_vmcnt<4 * (NS - 3)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        stage(kt + NS - 1);
        fetch(na, nb, kt + 1);
        mul(ca, cb);
        __builtin_amdgcn_sched_barrier(0);                      // the products stay in front of the next step's wait
    };
    int kt = 0;
    for (; kt + NS < nkt; kt += 2) {
        steady(a0, b0, a1, b1, kt);
        steady(a1, b1, a0, b0, kt + 1);
    }
    for (; kt < nkt; kt += 2) {
        iter(a0, b0, a1, b1, kt);
        if (kt + 1 < nkt) iter(a1, b1, a0, b0, kt + 1);
    }
    GEMM_STAMP(2);
    if (rt0 < mt16 && ct0 < nt16)
        gemm_epilogue<T, 4, 4, Epi>(acc, rt0, ct0, wave, lane, gemm_smem_l + (size_t)NS * 16384, epi, (int)blockIdx.z);
    GEMM_STAMP(3);
}

// Weight gradients: gw = P^T Q with the contraction over the examples t (python/FNN_wnzh.py:174),
// for the three products of a step in ONE launch: x'^T delta1, d1^T delta2, d2^T delta3.  Both
// operands are fragment-tiled transposed activations.  A workgroup owns a 64 x 64 output block of
// one product and one K slice (blockIdx.y); partial sums go to that slice's f32 slab.
struct WgradProb { const void* A; const void* B; float* out; int mt, nt, ldo; };   // mt, nt: 64-blocks
struct WgradArgs { WgradProb p[4]; int ldT, klen; size_t zstride; };      // p[3]: bag-bias gradient (mt = 0 if unused)

template <typename T>
__device__ __forceinline__ void wgrad_tile(const WgradProb& pr, const int b, const int by, const int ldT,
                                           const int klen, const size_t zstride)
{
    typedef typename Traits<T>::frag frag;
    constexpr int KS = Traits<T>::KS;
    const int bm = b / pr.nt, bn = b % pr.nt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt = bm * 4 + wave, ct0 = bn * 4;
    const int nkt_all = ldT / KS, kt0 = by * (klen / KS), nkt = klen / KS;
    const T* A = static_cast<const T*>(pr.A);
    const T* B = static_cast<const T*>(pr.B);
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef FNN_WGRAD_RING_BF16
#define FNN_WGRAD_RING_BF16 0      // measured with 1: the bf16 step's second launch 16.5 -> 17.9 us (SNN 24.6 -> 26.3): bf16 keeps the unroll-4 loop
#endif
    if constexpr (sizeof(T) == 4 || FNN_WGRAD_RING_BF16) {
        // the 4-byte element types walk twice as many k-steps as bf16 for the same slice: a ring of D k-steps of both operands in
        // registers (loads past the end re-read the last k-step: no branch in the loop), instead of one round trip per 4 k-steps
#ifndef FNN_WGRAD_RING
#define FNN_WGRAD_RING 6
#endif
        constexpr int D = FNN_WGRAD_RING;
        frag ra[D], rb[D][4];
        auto ld = [&](const int j, const int kt) {
            const int k = kt0 + min(kt, nkt - 1);
            ra[j] = *reinterpret_cast<const frag*>(ft_frag<T>(A, rt, k, nkt_all, lane));
#pragma unroll
            for (int n = 0; n < 4; ++n) rb[j][n] = *reinterpret_cast<const frag*>(ft_frag<T>(B, ct0 + n, k, nkt_all, lane));
        };
#pragma unroll
        for (int j = 0; j < D; ++j) ld(j, j);
        int kt = 0;
#ifndef FNN_WGRAD_PAIRED
#define FNN_WGRAD_PAIRED 0      // measured with 1: the second launch 18.4 -> 20.4 us (the union launch's register budget); the strip kernel keeps it
#endif
        if constexpr (std::is_same<T, bs16_t>::value && FNN_WGRAD_PAIRED && (D % 2 == 0)) {
            // bf16 pairs: two k-steps per product (mma_row2): three 16x16x32 MFMAs per tile and pair of k-steps instead of six 16x16x16
            for (; kt + D <= nkt; kt += D) {
#pragma unroll
                for (int j = 0; j < D; j += 2) {
                    mma_row2<4>(acc, ra[j], ra[j + 1], rb[j], rb[j + 1]);
                    ld(j, kt + D + j); ld(j + 1, kt + D + j + 1);
                }
            }
#pragma unroll
            for (int j = 0; j < D; j += 2) {
                if (kt + j + 1 < nkt) mma_row2<4>(acc, ra[j], ra[j + 1], rb[j], rb[j + 1]);
                else if (kt + j < nkt) mma_row<4>(acc, ra[j], rb[j]);
            }
        } else {
        for (; kt + D <= nkt; kt += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
#pragma unroll
                for (int n = 0; n < 4; ++n) mma(acc[n], ra[j], rb[j][n]);      // (term-major issue, mma_row, measured slower here: 18.2 -> 22.4 us)
                ld(j, kt + D + j);
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (kt + j < nkt) {
#pragma unroll
                for (int n = 0; n < 4; ++n) mma(acc[n], ra[j], rb[j][n]);
            }
        }
        }
    } else {
#pragma unroll 4
    for (int kt = 0; kt < nkt; ++kt) {
        const frag af = *reinterpret_cast<const frag*>(ft_frag<T>(A, rt, kt0 + kt, nkt_all, lane));
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const frag bf = *reinterpret_cast<const frag*>(ft_frag<T>(B, ct0 + n, kt0 + kt, nkt_all, lane));
            mma(acc[n], af, bf);
        }
    }
    }
    float* o = pr.out + (size_t)by * zstride;
    const int r0 = rt * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = (ct0 + n) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(size_t)(r0 + r) * pr.ldo + col] = acc[n][r];
    }
}

// the product is picked with constant indices into the argument struct (a runtime index spills it)
template <typename T>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bx, const int by)
{
    const int n0 = a.p[0].mt * a.p[0].nt, n1 = a.p[1].mt * a.p[1].nt, n2 = a.p[2].mt * a.p[2].nt;
    if (bx < n0) wgrad_tile<T>(a.p[0], bx, by, a.ldT, a.klen, a.zstride);
    else if (bx < n0 + n1) wgrad_tile<T>(a.p[1], bx - n0, by, a.ldT, a.klen, a.zstride);
    else if (bx < n0 + n1 + n2) wgrad_tile<T>(a.p[2], bx - n0 - n1, by, a.ldT, a.klen, a.zstride);
    else wgrad_tile<T>(a.p[3], bx - n0 - n1 - n2, by, a.ldT, a.klen, a.zstride);
}

template <typename T>
static __global__ __launch_bounds__(256) void k_wgrad(const WgradArgs a) { wgrad_body<T>(a, blockIdx.x, blockIdx.y); }

// ------------------------------------------------------------------------------------------
// Fused MLP strip kernel: A3 gather -> A4 forward -> loss -> A5 backward-data for 16 examples per
// workgroup, in ONE launch (python/FNN_wnzh.py:144-174 + :87-96).  The six products that the
// reference's Theano graph runs one after another (x.w1, d1.w2, d2.w3, delta2.w2^T, delta1.w1^T)
// never leave the CU: the strip's activations live in LDS (padded rows, conflict-free
// ds_read_b128 fragments), each of the workgroup's waves (8; 4 with FNN_STEP1_WAVES=4) owns a contiguous run of every layer's output fragments,
// and the weights stream from L2 straight into B fragments.  What goes back to HBM is only what
// later kernels need: the transposed activations for the weight-gradient GEMMs, gx' for the
// sparse-row update, p / loss.  Dimensions are compile-time: K1p = 64*CX, H1p = 64*C1,
// H2p = 64*C2.
// ------------------------------------------------------------------------------------------
template <typename T> struct MlpArgs {
    const int32_t* ids; const float* y; int B, F, K; const float* table16; int64_t n_rows; float w0;
    const T *w1, *w1t, *w2, *w2t; const float* w3p; const uint8_t *m1, *m2;
    int act1, act2, H1, H2, train;
    T *xpT, *d1T, *d2T, *dl1T, *dl2T, *dl3T; int ldT;
    float *gxp, *p_out, *loss_t; int* err;
    // embedding-bag input layer (SNN fine-tune, python/SNN_RBM.py:238-291); unused in FM mode
    const float* bb0; int rw; T* dlxT; float* gx_raw;
    int wt;                         // bits: 1 the transposed activations by write-through stores (store4_wt), 2 gx' too, 4 gx' regrouped into whole lines
                                    // (bit 8 of FNN_WT_STORES: the reference-shaped outputs of fnn_gather, k_gather_ref / k_bag_ref)
#ifdef FNN_STAMPS
    long long* dbg;                 // diagnostic build only: per-workgroup phase time stamps
#endif
};
#ifdef FNN_STAMPS
#define FNN_STAMP(i) do { if (threadIdx.x == 0) a.dbg[(size_t)blk * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define FNN_STAMP_RT(i) do { if (threadIdx.x == 0) a.dbg[(size_t)blk * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)   // 100 MHz
#else
#define FNN_STAMP(i) do { } while (0)
#define FNN_STAMP_RT(i) do { } while (0)
#endif

// BAG = false: layer one is the concatenation of the F gathered FM rows (FNN, :87-96).
// Weight stream of one product phase for the 4-byte element types (f32, bf16 pairs), whose whole-phase register prefetch (PF
// below) would not fit: a ring of D k-steps of B fragments in registers.  The first D k-steps are requested one phase AHEAD
// (`head`, like PF), the others D k-steps ahead inside the fully unrolled loop -- without it every k-step of these modes paid an
// L2 round trip for its C fragments (step1: 68 us in f32 against 20.5 in bf16 for twice the bytes and 1/16 of the MFMA rate).
template <typename T, int C, int D> struct WRing { typename Traits<T>::frag s[D][C]; };
#ifndef FNN_WRING_FRAGS
#define FNN_WRING_FRAGS 30            // fragments of a ring: D = FNN_WRING_FRAGS / C k-steps in flight
#endif
template <int C> constexpr int wring_depth(int nk) { return (FNN_WRING_FRAGS / C) < nk ? (FNN_WRING_FRAGS / C) : nk; }
// (rtmax: the operand's last row tile -- a wave whose run of C fragments is clipped re-reads that one, its products are discarded)
template <typename T, int NK, int C, int D>
__device__ __forceinline__ void wring_head(WRing<T, C, D>& r, const T* __restrict__ W, const int rt0, const int lane, const int rtmax) {
    typedef typename Traits<T>::frag frag;
#pragma unroll
    for (int kk = 0; kk < D; ++kk)
#pragma unroll
        for (int i = 0; i < C; ++i) r.s[kk][i] = *reinterpret_cast<const frag*>(ft_frag<T>(W, min(rt0 + i, rtmax), kk, NK, lane));
}
template <typename T, int NK, int C, int D>
__device__ __forceinline__ void wring_product(f32x4 (&acc)[C], WRing<T, C, D>& r, const T* ap, const T* __restrict__ W, const int rt0, const int lane, const int rtmax) {
    typedef typename Traits<T>::frag frag;
    constexpr int KS = Traits<T>::KS;
#ifndef FNN_SPLIT_PAIRED
#define FNN_SPLIT_PAIRED 1
#endif
    if constexpr (std::is_same<T, bs16_t>::value && FNN_SPLIT_PAIRED && D >= 2) {
#pragma unroll
        for (int kk = 0; kk + 1 < NK; kk += 2) {
            const frag af0 = *reinterpret_cast<const frag*>(ap + kk * KS), af1 = *reinterpret_cast<const frag*>(ap + (kk + 1) * KS);
            mma_row2<C>(acc, af0, af1, r.s[kk % D], r.s[(kk + 1) % D]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (kk + u + D < NK) {
#pragma unroll
                    for (int i = 0; i < C; ++i) r.s[(kk + u) % D][i] = *reinterpret_cast<const frag*>(ft_frag<T>(W, min(rt0 + i, rtmax), kk + u + D, NK, lane));
                }
        }
        if (NK & 1) {
            const frag af = *reinterpret_cast<const frag*>(ap + (NK - 1) * KS);
            mma_row<C>(acc, af, r.s[(NK - 1) % D]);
        }
    } else {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const frag af = *reinterpret_cast<const frag*>(ap + kk * KS);
        mma_row<C>(acc, af, r.s[kk % D]);
        if (kk + D < NK) {
#pragma unroll
            for (int i = 0; i < C; ++i) r.s[kk % D][i] = *reinterpret_cast<const frag*>(ft_frag<T>(W, min(rt0 + i, rtmax), kk + D, NK, lane));
        }
    }
    }
}

// BAG = true : layer one is x = sigmoid(sum of the F gathered rows of ww0 + bb0), rw floats wide
//              (SNN, python/SNN_RBM.py:248-256), and the kernel returns
//              delta = gx * x * (1 - x) (:288-290) instead of gx.
// D1, D2, DX: 64-column blocks of H1p, H2p, K1p.  NW waves share a layer's 16-column fragments in contiguous runs of
// C = ceil(fragments / NW) per wave (NW = 4: the blocks' own quarters, C = D; NW = 8: the last waves' runs are clipped -- `ok`).
template <typename T, int D1, int D2, int DX, bool BAG = false, int NW = 4>
__device__ __forceinline__ void mlp_body(const MlpArgs<T>& a, const int blk, unsigned char* smem)
{
    constexpr int NF1 = 4 * D1, NF2 = 4 * D2, NFX = 4 * DX, NT = 64 * NW;
    constexpr int C1 = (NF1 + NW - 1) / NW, C2 = (NF2 + NW - 1) / NW, CX = (NFX + NW - 1) / NW;
    typedef typename Traits<T>::frag frag;
    constexpr int EPL = Traits<T>::EPL, KS = Traits<T>::KS;
    constexpr int K1p = 64 * DX, H1p = 64 * D1, H2p = 64 * D2;
    constexpr int PAD = 16 / (int)sizeof(T);
    constexpr int LX = K1p + PAD, L1 = H1p + PAD, L2 = H2p + PAD;
    constexpr int LXM = LX > L1 ? LX : L1;
    T* sx = reinterpret_cast<T*>(smem);          // [16][LXM]  x' tile, later delta1 (stride L1)
    T* sd1 = sx + 16 * LXM;                      // [16][L1]   d1
    T* sdl2 = sd1 + 16 * L1;                     // [16][L2]   delta2
    float* sz = reinterpret_cast<float*>(sdl2 + 16 * L2);   // [NW][16]
    float* sxf = sz + 16 * NW;                   // BAG: [16][K1p] f32 copy of x for delta
    int* sids = reinterpret_cast<int*>(sxf + 16 * K1p);     // BAG: [16][F] ids of the strip

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int t0 = blk * 16;
    const int F = a.F, K = a.K, B = a.B, ldT = a.ldT;
    const bool wt = (a.wt & 1) != 0, wtg = (a.wt & 2) != 0;
#define ST4(p, x0, x1, x2, x3) do { if (wt) store4_wt(p, x0, x1, x2, x3); else store4(p, x0, x1, x2, x3); } while (0)
    FNN_STAMP_RT(14);
    FNN_STAMP(0);

    // small per-column operands first, so that no later load has to wait behind a store
    float m1v[C1], m2v[C2], w3v[C2];
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        const int col = (wave * C1 + i) * 16 + lr;
        const float mv = (float)a.m1[col < a.H1 ? col : 0];      // m1 is never null (predict: ones)
        m1v[i] = (col < a.H1) ? mv : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < C2; ++i) {
        const int col = (wave * C2 + i) * 16 + lr;
        const float mv = (float)a.m2[col < a.H2 ? col : 0];
        m2v[i] = (col < a.H2) ? mv : 0.0f;
        w3v[i] = (wave * C2 + i < NF2) ? a.w3p[col] : 0.0f;
    }

    float rv[4];                                   // 1 for rows of the batch, 0 for the padding rows
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = (t0 + 4 * lq + r < B) ? 1.0f : 0.0f;
    float yv[4];                                   // labels of this lane's 4 rows (y is never null)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int t = t0 + 4 * lq + r; yv[r] = a.y[t < B ? t : B - 1]; }

    // bf16 mode: the weight fragments of a whole phase are fetched into registers one phase AHEAD
    // (they do not depend on the activations), so the L2 latency of the weight stream hides under
    // the previous phase instead of being paid 8 loads at a time inside the MFMA loop.
    constexpr bool PF = sizeof(T) == 2;
    constexpr int NK1 = K1p / KS, NKH1 = H1p / KS, NKH2 = H2p / KS;
    frag b1[PF ? NK1 : 1][PF ? C1 : 1];
    constexpr int DR1 = wring_depth<C1>(NK1), DR2 = wring_depth<C2>(NKH1), DR3 = wring_depth<C1>(NKH2), DR4 = wring_depth<CX>(NKH1);
    WRing<T, C1, PF ? 1 : DR1> r1;
    if constexpr (PF) {
#pragma unroll
        for (int kk = 0; kk < NK1; ++kk)
#pragma unroll
            for (int i = 0; i < C1; ++i)
                b1[kk][i] = *reinterpret_cast<const frag*>(ft_frag<T>(a.w1t, min(wave * C1 + i, NF1 - 1), kk, NK1, lane));
    } else wring_head<T, NK1, C1, DR1>(r1, a.w1t, wave * C1, lane, NF1 - 1);

    if constexpr (BAG) {
        // ---- P0 (bag): x = sigmoid(sum_f ww0[id_f] + bb0).  ids of the strip first (one per thread),
        // then every thread sums the 16-byte quarter-columns it owns over the F rows: F independent
        // loads in flight per item, rw/4 items per example.
        constexpr int BAGL = 16;             // row loads in flight per item (all 16 fields of an example at once)
        const int rw = a.rw, nq = rw >> 2;
        for (int e = tid; e < 16 * F; e += NT) {
            const int t = t0 + e / F;
            int id = -1;
            if (t < B) {
                id = a.ids[(size_t)t * F + e % F];
                if (id < -1 || id >= a.n_rows) { atomicOr(a.err, 1); id = -1; }
            }
            sids[e] = id;
        }
        lds_barrier();
        for (int e = tid; e < 16 * nq; e += NT) {
            const int r = e / nq, c4 = e % nq;
            float4 acc = *reinterpret_cast<const float4*>(a.bb0 + 4 * c4);
            for (int f0 = 0; f0 < F; f0 += BAGL) {
                float4 v[BAGL]; float w[BAGL];
#pragma unroll
                for (int u = 0; u < BAGL; ++u) {
                    const int id = (f0 + u < F) ? sids[r * F + f0 + u] : -1;
                    w[u] = id >= 0 ? 1.0f : 0.0f;
                    v[u] = *reinterpret_cast<const float4*>(a.table16 + (size_t)(id < 0 ? 0 : id) * rw + 4 * c4);
                }
#pragma unroll
                for (int u = 0; u < BAGL; ++u) {
                    acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
                    acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
                }
            }
            const float live = (t0 + r < B) ? 1.0f : 0.0f;
            const float x0 = sigmoid_fast(acc.x) * live, x1 = sigmoid_fast(acc.y) * live,
                        x2 = sigmoid_fast(acc.z) * live, x3 = sigmoid_fast(acc.w) * live;
            store4(sx + r * LX + 4 * c4, x0, x1, x2, x3);
            *reinterpret_cast<float4*>(sxf + r * K1p + 4 * c4) = make_float4(x0, x1, x2, x3);
        }
        for (int e = tid; e < 16 * (K1p - rw); e += NT) {       // ones column (b1 row) + padding
            const int r = e / (K1p - rw), c = rw + e % (K1p - rw);
            const float v = (c == rw && t0 + r < B) ? 1.0f : 0.0f;
            sx[r * LX + c] = (T)v; sxf[r * K1p + c] = 0.0f;
        }
        lds_barrier();
        if (a.train) {
            for (int e = tid; e < K1p * 4; e += NT) {
                const int c = e >> 2, tq = e & 3;
                ST4(a.xpT + ft_off<T>(c, t0 + 4 * tq, ldT), (float)sx[(4 * tq) * LX + c], (float)sx[(4 * tq + 1) * LX + c],
                    (float)sx[(4 * tq + 2) * LX + c], (float)sx[(4 * tq + 3) * LX + c]);
            }
        }
    } else {
    // ---- P0: gather 16 examples x F rows (64 B each) into the x' tile and x'^T (:87-96).
    // All ids first, then all rows: two dependent round trips for the whole strip.
    for (int e = tid; e < 16 * F; e += NT) {
        const int q = e & 3, f = (e >> 2) % F, tq = (e >> 2) / F;
        int64_t id[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + 4 * tq + i;
            id[i] = a.ids[(size_t)(t < B ? t : B - 1) * F + f];
        }
        bool bad = false;
        float4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + 4 * tq + i;
            if (t >= B) id[i] = -1;
            if (id[i] < -1 || id[i] >= a.n_rows) { bad = true; id[i] = -1; }
            r[i] = *reinterpret_cast<const float4*>(a.table16 + (size_t)(id[i] < 0 ? 0 : id[i]) * SLOT + 4 * q);
        }
        if (bad) atomicOr(a.err, 1);
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + 4 * tq + i;
            const bool live = id[i] >= 0;
            v[i][0] = live ? r[i].x : 0.f; v[i][1] = live ? r[i].y : 0.f;
            v[i][2] = live ? r[i].z : 0.f; v[i][3] = live ? r[i].w : 0.f;
            if (t < B && q == (K >> 2)) {
                if (f == 0) v[i][K & 3] = a.w0;
                if (f == 1) v[i][K & 3] = 1.0f;
            }
        }
        const int c0 = f * SLOT + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) store4(sx + (4 * tq + i) * LX + c0, v[i][0], v[i][1], v[i][2], v[i][3]);
        if (a.train) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ST4(a.xpT + ft_off<T>(c0 + j, t0 + 4 * tq, ldT), v[0][j], v[1][j], v[2][j], v[3][j]);
        }
    }
    for (int e = tid; e < 16 * (K1p - F * SLOT); e += NT) {       // pad columns of the tile
        const int r = e / (K1p - F * SLOT), c = F * SLOT + e % (K1p - F * SLOT);
        sx[r * LX + c] = (T)0.f;
    }
    }
    lds_barrier();
    FNN_STAMP(1);

    // ---- P1: d1 = act(x' W1p) * r1   (:147-155)
    float d1v[C1][4];
    frag b2[PF ? NKH1 : 1][PF ? C2 : 1];
    WRing<T, C2, PF ? 1 : DR2> r2;
    if constexpr (PF) {
#pragma unroll
        for (int kk = 0; kk < NKH1; ++kk)
#pragma unroll
            for (int i = 0; i < C2; ++i)
                b2[kk][i] = *reinterpret_cast<const frag*>(ft_frag<T>(a.w2t, min(wave * C2 + i, NF2 - 1), kk, NKH1, lane));
    } else wring_head<T, NKH1, C2, DR2>(r2, a.w2t, wave * C2, lane, NF2 - 1);
    {
        f32x4 acc[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* ap = sx + lr * LX + lq * EPL;
        if constexpr (PF) {
#pragma unroll
            for (int kk = 0; kk < NK1; ++kk) {
                const frag af = *reinterpret_cast<const frag*>(ap + kk * KS);
#pragma unroll
                for (int i = 0; i < C1; ++i) mma(acc[i], af, b1[kk][i]);
            }
        } else wring_product<T, NK1, C1, DR1>(acc, r1, ap, a.w1t, wave * C1, lane, NF1 - 1);
        FNN_STAMP(2);
        // m1v is 0 outside the real columns, so act(z)*m + [col == H1] is the ones column / padding too
        const ActCoef ac1 = act_coef(a.act1);
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int col = (wave * C1 + i) * 16 + lr;
            const bool ok = wave * C1 + i < NF1;
            const float m = m1v[i], one = (col == a.H1) ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaf(act_apply(ac1, acc[i][r]), m, one) * rv[r];
                d1v[i][r] = v;
                if (ok) sd1[(4 * lq + r) * L1 + col] = (T)v;
            }
        }
        if (a.train) {
#pragma unroll
            for (int i = 0; i < C1; ++i)
                if (wave * C1 + i < NF1) ST4(a.d1T + ft_off<T>((wave * C1 + i) * 16 + lr, t0 + 4 * lq, ldT), d1v[i][0], d1v[i][1], d1v[i][2], d1v[i][3]);
        }
    }
    lds_barrier();
    FNN_STAMP(3);

    // ---- P2: d2 = act2(d1 W2p) * r2, z3 = d2 . w3p   (:164-169)
    float d2v[C2][4], zp[4] = {0.f, 0.f, 0.f, 0.f};
    frag b3[PF ? NKH2 : 1][PF ? C1 : 1];
    WRing<T, C1, PF ? 1 : DR3> r3;
    if constexpr (PF) {
        if (a.train) {
#pragma unroll
            for (int kk = 0; kk < NKH2; ++kk)
#pragma unroll
                for (int i = 0; i < C1; ++i)
                    b3[kk][i] = *reinterpret_cast<const frag*>(ft_frag<T>(a.w2, min(wave * C1 + i, NF1 - 1), kk, NKH2, lane));
        }
    } else { if (a.train) wring_head<T, NKH2, C1, DR3>(r3, a.w2, wave * C1, lane, NF1 - 1); }
    {
        f32x4 acc[C2];
#pragma unroll
        for (int i = 0; i < C2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* ap = sd1 + lr * L1 + lq * EPL;
        if constexpr (PF) {
#pragma unroll
            for (int kk = 0; kk < NKH1; ++kk) {
                const frag af = *reinterpret_cast<const frag*>(ap + kk * KS);
#pragma unroll
                for (int i = 0; i < C2; ++i) mma(acc[i], af, b2[kk][i]);
            }
        } else wring_product<T, NKH1, C2, DR2>(acc, r2, ap, a.w2t, wave * C2, lane, NF2 - 1);
        FNN_STAMP(4);
        const ActCoef ac2 = act_coef(a.act2);
#pragma unroll
        for (int i = 0; i < C2; ++i) {
            const int col = (wave * C2 + i) * 16 + lr;
            const float m = m2v[i], w3 = w3v[i], one = (col == a.H2) ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaf(act_apply(ac2, acc[i][r]), m, one) * rv[r];
                d2v[i][r] = v;
                zp[r] = fmaf(v, w3, zp[r]);
            }
        }
        if (a.train) {
#pragma unroll
            for (int i = 0; i < C2; ++i)
                if (wave * C2 + i < NF2) ST4(a.d2T + ft_off<T>((wave * C2 + i) * 16 + lr, t0 + 4 * lq, ldT), d2v[i][0], d2v[i][1], d2v[i][2], d2v[i][3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        zp[r] += __shfl_xor(zp[r], 1); zp[r] += __shfl_xor(zp[r], 2);
        zp[r] += __shfl_xor(zp[r], 4); zp[r] += __shfl_xor(zp[r], 8);
        if (lr == 0) sz[wave * 16 + 4 * lq + r] = zp[r];
    }
    lds_barrier();
    float d3[4], zr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r, t = t0 + row;
        float zs = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) zs += sz[16 * w + row];
        zr[r] = zs;
        const float p = sigmoid_fast(zr[r]);
        d3[r] = (a.train && t < B) ? p - yv[r] : 0.f;
        if (a.p_out && wave == 0 && lr == 0 && t < B) a.p_out[t] = p;
    }
    FNN_STAMP(5);
    if (!a.train) return;
    if (wave == 0 && lr == 0) {          // one lane per 4 rows writes delta3 and the per-example loss
        float ls[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // -y log p - (1-y) log(1-p) = softplus(z) - y z
            const float z = zr[r];
            ls[r] = (t0 + 4 * lq + r < B) ? fmaxf(z, 0.f) + __logf(1.0f + __expf(-fabsf(z))) - yv[r] * z : 0.f;
        }
        store4(a.dl3T + ft_off<T>(0, t0 + 4 * lq, ldT), d3[0], d3[1], d3[2], d3[3]);
        store4(a.loss_t + t0 + 4 * lq, ls[0], ls[1], ls[2], ls[3]);
    }
    // delta2 = delta3 * w3 * r2 * (1 - d2^2)   (layer 2 is tanh on the dropout path, :165)
#pragma unroll
    for (int i = 0; i < C2; ++i) {
        const int col = (wave * C2 + i) * 16 + lr;
        if (wave * C2 + i >= NF2) continue;
        const float w3m = w3v[i] * m2v[i];
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = d3[r] * w3m * (1.0f - d2v[i][r] * d2v[i][r]);
            sdl2[(4 * lq + r) * L2 + col] = (T)v[r];
        }
        ST4(a.dl2T + ft_off<T>(col, t0 + 4 * lq, ldT), v[0], v[1], v[2], v[3]);
    }
    lds_barrier();
    FNN_STAMP(6);

    // ---- P3: delta1 = (delta2 W2p^T) * r1 * act'(d1)
    T* sdl1 = sx;                                       // the x' tile is dead since P1
    frag b4[PF ? NKH1 : 1][PF ? CX : 1];
    WRing<T, CX, PF ? 1 : DR4> r4;
    if constexpr (PF) {
#pragma unroll
        for (int kk = 0; kk < NKH1; ++kk)
#pragma unroll
            for (int i = 0; i < CX; ++i)
                b4[kk][i] = *reinterpret_cast<const frag*>(ft_frag<T>(a.w1, min(wave * CX + i, NFX - 1), kk, NKH1, lane));
    } else wring_head<T, NKH1, CX, DR4>(r4, a.w1, wave * CX, lane, NFX - 1);
    {
        f32x4 acc[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* ap = sdl2 + lr * L2 + lq * EPL;
        if constexpr (PF) {
#pragma unroll
            for (int kk = 0; kk < NKH2; ++kk) {
                const frag af = *reinterpret_cast<const frag*>(ap + kk * KS);
#pragma unroll
                for (int i = 0; i < C1; ++i) mma(acc[i], af, b3[kk][i]);
            }
        } else wring_product<T, NKH2, C1, DR3>(acc, r3, ap, a.w2, wave * C1, lane, NF1 - 1);
        FNN_STAMP(7);
        const ActCoef ac1 = act_coef(a.act1);
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int col = (wave * C1 + i) * 16 + lr;
            if (wave * C1 + i >= NF1) continue;
            const float m = m1v[i];
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][r] * m * dact_apply(ac1, d1v[i][r]) * rv[r];
                sdl1[(4 * lq + r) * L1 + col] = (T)v[r];
            }
            ST4(a.dl1T + ft_off<T>(col, t0 + 4 * lq, ldT), v[0], v[1], v[2], v[3]);
        }
    }
    lds_barrier();
    FNN_STAMP(8);

    // ---- P4: gx' = delta1 W1p^T   (what `train` returns first, :174,:179)
    {
        f32x4 acc[CX];
#pragma unroll
        for (int i = 0; i < CX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* ap = sdl1 + lr * L1 + lq * EPL;
        if constexpr (PF) {
#pragma unroll
            for (int kk = 0; kk < NKH1; ++kk) {
                const frag af = *reinterpret_cast<const frag*>(ap + kk * KS);
#pragma unroll
                for (int i = 0; i < CX; ++i) mma(acc[i], af, b4[kk][i]);
            }
        } else wring_product<T, NKH1, CX, DR4>(acc, r4, ap, a.w1, wave * CX, lane, NFX - 1);
        FNN_STAMP(9);
        if constexpr (!BAG && CX % 2 == 0) {
            if (a.wt & 4) {
                // gx' [example][K1p] f32 in whole 128-byte lines: two fragments (32 columns) at a time through a wave-private LDS
                // block -- the MFMA layout has a lane on 4 ROWS of one column (sixteen 4-byte stores per lane,
                // 64-byte pieces per instruction); regrouped, a lane holds 4 columns of one row (four 16-byte stores per lane)
                float* sg = sz + 16 * NW + wave * (16 * 36);           // behind the tiles (mlp_lds_bytes: FM mode keeps 4 x 2,304 bytes there)
#pragma unroll
                for (int ip = 0; ip < CX / 2; ++ip) {
                    if (wave * CX + 2 * ip >= NFX) break;                 // a clipped run (CX is even: pairs never straddle the end)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sg[(4 * lq + r) * 36 + f * 16 + lr] = acc[2 * ip + f][r];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = hh * 8 + (lane >> 3), c4 = lane & 7;
                        const float4 v = *reinterpret_cast<const float4*>(sg + row * 36 + 4 * c4);
                        float* dst = a.gxp + (size_t)(t0 + row) * K1p + (wave * CX + 2 * ip) * 16 + 4 * c4;
                        if (wtg) store4_wt(dst, v.x, v.y, v.z, v.w); else store4(dst, v.x, v.y, v.z, v.w);
                    }
                }
                FNN_STAMP(10);
                FNN_STAMP_RT(15);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < CX; ++i) {
            const int col = (wave * CX + i) * 16 + lr;
            if (wave * CX + i >= NFX) continue;
            if constexpr (BAG) {       // delta = gx * x * (1 - x)  (python/SNN_RBM.py:288-290, lr applied later)
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * lq + r;
                    const float xv = sxf[row * K1p + col];
                    v[r] = acc[i][r] * xv * (1.0f - xv);          // sxf is 0 beyond column rw and row B
                    if (wtg) store1_wt(a.gxp + (size_t)(t0 + row) * K1p + col, v[r]); else a.gxp[(size_t)(t0 + row) * K1p + col] = v[r];
                    if (a.gx_raw) a.gx_raw[(size_t)(t0 + row) * K1p + col] = acc[i][r];
                }
                ST4(a.dlxT + ft_off<T>(col, t0 + 4 * lq, ldT), v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { if (wtg) store1_wt(a.gxp + (size_t)(t0 + 4 * lq + r) * K1p + col, acc[i][r]); else a.gxp[(size_t)(t0 + 4 * lq + r) * K1p + col] = acc[i][r]; }
            }
        }
    }
    FNN_STAMP(10);
    FNN_STAMP_RT(15);
#undef ST4
}

template <typename T, int C1, int C2, int CX>
static __global__ __launch_bounds__(256) void k_mlp(const MlpArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    mlp_body<T, C1, C2, CX>(a, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// Output unit + loss + delta_2 (python/FNN_wnzh.py:169,172 and the first two lines of the
// closed form in SURVEY 8a/A5):  z3 = d2.w3p (b3 rides on the ones column), p = sigmoid(z3),
// xent, delta3 = p - y, delta2 = delta3 * w3 * r2 * (1 - d2^2).  delta3 is also written as row 0
// of a [16][ldT] matrix so that gw3p = d2^T delta3 is one more MFMA product of the weight-
// gradient launch; the per-example loss goes to loss_t.  16 lanes share 4 rows; a lane owns 4
// columns of every 64-column chunk.  block = 256 threads = 64 rows.
// ------------------------------------------------------------------------------------------
template <typename T>
static __global__ __launch_bounds__(256) void k_head(const T* __restrict__ d2, int H2p, int H2,
                                              const float* __restrict__ w3p,
                                              const uint8_t* __restrict__ mask2,
                                              const float* __restrict__ y, int B, int train,
                                              float* __restrict__ p_out, T* __restrict__ dl2,
                                              T* __restrict__ dl2T, int ldT, T* __restrict__ dl3T,
                                              float* __restrict__ loss_t)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c16 = lane & 15;
    const int row0 = blockIdx.x * 64 + wave * 16 + g * 4;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < H2p; cb += 64) {
        const int c = cb + c16 * 4;
        const float4 w = *reinterpret_cast<const float4*>(w3p + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 d = load4(d2 + (size_t)(row0 + r) * H2p + c);
            z[r] += d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        z[r] += __shfl_xor(z[r], 1); z[r] += __shfl_xor(z[r], 2);
        z[r] += __shfl_xor(z[r], 4); z[r] += __shfl_xor(z[r], 8);
    }
    float d3[4], ls[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = row0 + r;
        const float p = 1.0f / (1.0f + expf(-z[r]));
        d3[r] = 0.f; ls[r] = 0.f;
        if (t < B) {
            if (p_out && c16 == 0) p_out[t] = p;
            if (train) {
                const float yy = y[t];
                d3[r] = p - yy;
                // -y log p - (1-y) log(1-p) = softplus(z) - y z
                ls[r] = fmaxf(z[r], 0.f) + log1pf(expf(-fabsf(z[r]))) - yy * z[r];
            }
        }
    }
    if (!train) return;
    if (c16 == 0) {
        store4(dl3T + ft_off<T>(0, row0, ldT), d3[0], d3[1], d3[2], d3[3]);
        store4(loss_t + row0, ls[0], ls[1], ls[2], ls[3]);
    }
    for (int cb = 0; cb < H2p; cb += 64) {
        const int c = cb + c16 * 4;
        const float4 w = *reinterpret_cast<const float4*>(w3p + c);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = (c + j < H2) ? (mask2 ? (float)mask2[c + j] : 1.f) : 0.f;
        float o[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 d = load4(d2 + (size_t)(row0 + r) * H2p + c);
            const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[r][j] = d3[r] * wv[j] * m[j] * (1.0f - dv[j] * dv[j]);   // layer 2 is tanh (:165)
            store4(dl2 + (size_t)(row0 + r) * H2p + c, o[r][0], o[r][1], o[r][2], o[r][3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store4(dl2T + ft_off<T>(c + j, row0, ldT), o[0][j], o[1][j], o[2][j], o[3][j]);
    }
}

// ------------------------------------------------------------------------------------------
// Dense gradient bucket: sum the split-K slabs in a fixed order (the per-example part of the
// gradient, which is what data parallelism all-reduces); the last block sums the per-example losses.
// Slab z = [W1p grads n1 | W2p grads n2 | gw3p as column 0 of an [H2p][64] tile].
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_reduce(const float* __restrict__ slab, int splitk,
                                                size_t nw_all, size_t nw12, size_t nslab,
                                                const float* __restrict__ master, float lambda1,
                                                int reg_all, const float* __restrict__ loss_t, int Ba,
                                                float* __restrict__ bucket, float* __restrict__ loss_sum)
{
    if (blockIdx.x == gridDim.x - 1) {                       // loss: fixed-shape tree
        __shared__ float s_l[256];
        float v = 0.f;
        v = strided_sum256(loss_t, Ba);
        s_l[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_l[threadIdx.x] += s_l[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *loss_sum = s_l[0];
        return;
    }
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw_all) return;
    const size_t src = (i < nw12) ? i : nw12 + (i - nw12) * 64;
    float g = 0.f;
#pragma unroll 8
    for (int z = 0; z < splitk; ++z) g += slab[(size_t)z * nslab + src];
    bucket[i] = g;       // data term only: the L2 term is added where theta is updated (k_update)
}

// theta <- theta - lr * (g + L2 term)  (python/FNN_wnzh.py:173,179-182) on the f32 masters, then refresh the
// compute-precision shadows in both orientations, fragment-tiled (ft_off).
template <typename T>
static __global__ void k_update(float* __restrict__ master, const float* __restrict__ bucket, float lr,
                         float lambda1, int reg_all, int K1p, int H1p, int H2p, T* __restrict__ w1,
                         T* __restrict__ w1t, T* __restrict__ w2, T* __restrict__ w2t)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n1 = (size_t)K1p * H1p, n2 = (size_t)H1p * H2p;
    if (i >= n1 + n2 + H2p) return;
    float w = master[i];
    if (bucket) {      // + L2 term 2*lambda1*theta on w3,b3 (python/FNN_wnzh.py:173) or on all six (SNN)
        float g = bucket[i];
        if (reg_all || i >= n1 + n2) g += 2.0f * lambda1 * w;
        w -= lr * g; master[i] = w;
    }
    if (i < n1) {           // W1p[r = x' slot][c = h1 unit]
        const int r = (int)(i / H1p), c = (int)(i % H1p);
        w1t[ft_off<T>(c, r, K1p)] = (T)w;      // forward:  output column c, contraction over r
        w1[ft_off<T>(r, c, H1p)] = (T)w;       // gx:       output column r, contraction over c
    } else if (i < n1 + n2) {                  // W2p[r = h1 unit][c = h2 unit]
        const size_t j = i - n1;
        const int r = (int)(j / H2p), c = (int)(j % H2p);
        w2t[ft_off<T>(c, r, H1p)] = (T)w;      // forward
        w2[ft_off<T>(r, c, H2p)] = (T)w;       // delta1
    }
}

// fnn_set_shadowed: every (example, field, row) entry must name a row of that field (bit 3 of the flag otherwise)
static __global__ void k_check_shadowed(const int32_t* __restrict__ tfr, int n, const int32_t* __restrict__ field_of_row, int64_t n_rows,
                                        int F, int* __restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = tfr[3 * i + 1]; const int64_t r = tfr[3 * i + 2];
    if (f < 0 || f >= F || r < 0 || r >= n_rows || field_of_row[r] != f) atomicOr(err, 8);
}

// ------------------------------------------------------------------------------------------
// A6  sparse-row SGD with the reference's sequential duplicate semantics
// (python/FNN_wnzh.py:299-306): a row hit by m examples (in example order) with slot gradients
// g_1..g_m ends at  row*c^m - lr * sum_j g_j * c^(m-j),  c = 1 - 2*lambda_fm*lr/b_size.
//   k_sort    per field: bitonic sort of the (row, t) keys -- in registers for strides inside a
//             thread, with wave shuffles inside a wave, through LDS only for the few strides that
//             cross waves -- then every sorted entry learns its segment [s, e) by binary search
//             -> rec {row, t, s, e}.  Independent of the gradients: runs on a side stream under
//             the MLP.
//   k_scat1   a 16-lane group (lane = slot of the row) owns 16 consecutive sorted entries and adds
//             g * c^(e-1-pos) in f64 per run of equal rows.  The weight is absolute inside the
//             segment, so partial sums of a segment cut by chunk borders simply add up.  Runs
//             that lie inside the chunk are written back at once; the others leave a partial and
//             the run that opens a multi-chunk segment registers its owner.
//   k_scat2   one workgroup per registered owner adds the partials of its segment in a fixed
//             order and writes the row.  No float atomics anywhere: the result is bitwise
//             reproducible.
// ------------------------------------------------------------------------------------------
// `extra` [n_extra][3] = (example t, field, row): features of a line that a LATER feature of the same field shadowed in the
// gather (python/FNN_wnzh.py:91-96 keeps the last) but that the reference's update loop still visits (:300-306 walks every
// feature of the line).  They join their field's keys as (row, t) pairs behind the B regular ones, so that a row's decay
// powers and gradient terms count every visit in example order; N2 >= B + (extras of any one field).
template <int KPT>
static __global__ __launch_bounds__(1024) void k_sort(const int32_t* __restrict__ ids, int B, int F,
                                               int64_t n_rows, int N2, int4* __restrict__ rec,
                                               int* __restrict__ owner_cnt, const int32_t* __restrict__ extra, int n_extra,
                                               int* __restrict__ err)
{
    extern __shared__ unsigned long long s_key[];
    __shared__ int s_cnt;
    const int f = blockIdx.x, tid = threadIdx.x;       // blockDim.x == N2 / KPT
    if (f == 0 && tid == 0) *owner_cnt = 0;
    int cnt = 0;
    if (n_extra > 0) {                                 // wave-uniform: the common case pays one compare
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int j = tid; j < n_extra; j += blockDim.x) {
            const int t = extra[3 * j], ff = extra[3 * j + 1]; const int64_t r = extra[3 * j + 2];
            if (ff < 0 || ff >= F || t < 0 || t >= B || r < 0 || r >= n_rows) { if (f == 0) atomicOr(err, 1); continue; }
            if (ff != f) continue;
            const int p = atomicAdd(&s_cnt, 1);
            if (B + p < N2) s_key[B + p] = ((unsigned long long)r << 32) | (unsigned)t;
        }
        __syncthreads();
        cnt = min(s_cnt, N2 - B);
    }
    unsigned long long key[KPT];
#pragma unroll
    for (int a = 0; a < KPT; ++a) {
        const int i = tid * KPT + a;
        unsigned long long kk = ~0ull;
        if (i < B) {
            const int64_t id = ids[(size_t)i * F + f];
            if (id >= 0 && id < n_rows) kk = ((unsigned long long)id << 32) | (unsigned)i;
        } else if (i < B + cnt) kk = s_key[i];
        key[a] = kk;
    }
    if (n_extra > 0) __syncthreads();                  // s_key is reused by the exchange stages
    for (int k = 2; k <= N2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < KPT) {                              // both elements live in this thread
#pragma unroll
                for (int jj = KPT >> 1; jj > 0; jj >>= 1) {
                    if (j == jj) {
#pragma unroll
                        for (int a = 0; a < KPT; ++a) {
                            const int b = a ^ jj;
                            if (b > a) {
                                const bool up = ((tid * KPT + a) & k) == 0;
                                const unsigned long long x = key[a], y = key[b];
                                if ((x > y) == up) { key[a] = y; key[b] = x; }
                            }
                        }
                    }
                }
            } else if (j < 64 * KPT) {                  // partner lane of the same wave
                const int d = j / KPT;
#pragma unroll
                for (int a = 0; a < KPT; ++a) {
                    const int i = tid * KPT + a;
                    const unsigned long long other = __shfl_xor(key[a], d);
                    const bool keepmin = ((i & j) == 0) == ((i & k) == 0);
                    const unsigned long long mine = key[a];
                    key[a] = keepmin ? (mine < other ? mine : other) : (mine > other ? mine : other);
                }
            } else {                                    // partner in another wave: through LDS
                __syncthreads();
#pragma unroll
                for (int a = 0; a < KPT; ++a) s_key[tid * KPT + a] = key[a];
                __syncthreads();
#pragma unroll
                for (int a = 0; a < KPT; ++a) {
                    const int i = tid * KPT + a;
                    const unsigned long long other = s_key[i ^ j];
                    const bool keepmin = ((i & j) == 0) == ((i & k) == 0);
                    const unsigned long long mine = key[a];
                    key[a] = keepmin ? (mine < other ? mine : other) : (mine > other ? mine : other);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < KPT; ++a) s_key[tid * KPT + a] = key[a];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < KPT; ++a) {
        const int pos = tid * KPT + a;
        const unsigned long long kk = key[a];
        int4 r = make_int4(-1, 0, 0, 0);
        if (kk != ~0ull) {
            const unsigned long long lo_key = kk & 0xffffffff00000000ull;
            const unsigned long long hi_key = lo_key + 0x100000000ull;
            int lo = 0, hi = pos;                     // first index with key >= lo_key
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_key[mid] < lo_key) lo = mid + 1; else hi = mid; }
            const int s = lo;
            lo = pos + 1; hi = N2;                    // first index with key >= hi_key
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_key[mid] < hi_key) lo = mid + 1; else hi = mid; }
            r = make_int4((int)(kk >> 32), (int)(kk & 0xffffffffu), s, lo);
        }
        rec[(size_t)f * N2 + pos] = r;
    }
}

struct ScatArgs {
    const int4* rec; int N2, F, K; const float* gxp; int K1p; const double* cpow; double lr;
    float* table16; double* part; int* owner_cnt; int4* owners;
    int rw;          // 16: FM rows (decayed update); otherwise the bag-table row width (plain sum)
    const int* tag_shared; int stamp;     // bag mode: tag_shared[row] == stamp <=> the row sits in several columns of this batch (SortArgs)
};
// bag rows held by several columns of a batch: every column adds its sum with float atomics (a row touched by one column
// only -- the rule on iPinYou lines -- keeps the plain read-modify-write, one rounding)
__device__ __forceinline__ void atomic_add4(float* p, float a, float b, float c, float d) {
    atomicAdd(p, a); atomicAdd(p + 1, b); atomicAdd(p + 2, c); atomicAdd(p + 3, d);
}
__device__ __forceinline__ void scat1_body(const ScatArgs& sa, const int blk)
{
    const int4* __restrict__ rec = sa.rec; const int N2 = sa.N2, F = sa.F, K = sa.K, K1p = sa.K1p;
    const float* __restrict__ gxp = sa.gxp; const double* __restrict__ cpow = sa.cpow; const double lr = sa.lr;
    float* __restrict__ table16 = sa.table16; double* __restrict__ part = sa.part;
    int* __restrict__ owner_cnt = sa.owner_cnt; int4* __restrict__ owners = sa.owners;
    const int l = threadIdx.x & 15;                       // slot of the row
    const int G = (blk * 256 + threadIdx.x) >> 4;         // chunk of 16 sorted entries
    const int NQ = N2 >> 4;
    if (G >= F * NQ) return;
    const int f = G / NQ, q = G % NQ, base = q * 16;
    const int4 mine = rec[(size_t)f * N2 + base + l];
    // both decay factors of an entry depend on its record only: its own weight c^(e-1-pos) and its
    // segment's c^(e-s) -- fetched together with the gradients and the old rows, not after them.
    // (The conditional loads below compile to a branch and a wait per entry.  Making all 34 of them unconditional puts them
    // in flight together and was measured SLOWER inside the launch, 16.4 -> 18.2 us, A/B on one box with tools/gpu_ab.sh:
    // the role shares its CUs' 64 B/clk texture path with the weight-gradient role, which the burst of dword loads starves.)
    const double wmine = (mine.x >= 0) ? cpow[mine.w - 1 - (base + l)] : 0.0;
    const double cmine = (mine.x >= 0) ? cpow[mine.w - mine.z] : 0.0;
    int row[16], sg[16], eg[16];
    float g[16], wold[16];
    double w[16], cs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        row[j] = __shfl(mine.x, j, 16);
        const int t = __shfl(mine.y, j, 16);
        sg[j] = __shfl(mine.z, j, 16);
        eg[j] = __shfl(mine.w, j, 16);
        w[j] = __shfl(wmine, j, 16);
        cs[j] = __shfl(cmine, j, 16);
        const bool live = row[j] >= 0 && l < K;
        g[j] = live ? gxp[(size_t)t * K1p + f * SLOT + l] : 0.f;
        wold[j] = live ? table16[(size_t)row[j] * SLOT + l] : 0.f;
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (row[j] < 0) continue;
        acc += (double)g[j] * w[j];
        const bool last = (j == 15) || (row[j + 1 < 16 ? j + 1 : 15] != row[j]);
        if (!last) continue;
        const int s = sg[j], e = eg[j];
        if (s >= base && e <= base + 16) {                 // the whole segment lies in this chunk
            if (l < K) table16[(size_t)row[j] * SLOT + l] = (float)((double)wold[j] * cs[j] - lr * acc);
        } else {
            const int which = (s < base) ? 0 : 1;          // 0: enters from the left; 1: opens here
            part[(((size_t)f * NQ + q) * 2 + which) * SLOT + l] = acc;
            if (which == 1 && l == 0) owners[atomicAdd(owner_cnt, 1)] = make_int4(f, s, e, row[j]);
        }
        acc = 0.0;
    }
}

static __global__ __launch_bounds__(256) void k_scat1(const ScatArgs sa) { scat1_body(sa, blockIdx.x); }

__device__ __forceinline__ void scat2_body(const ScatArgs& sa, const int blk, const int nblk, double (*s_sum)[16])
{
    const int4* __restrict__ owners = sa.owners; const int N2 = sa.N2, K = sa.K;
    const double* __restrict__ part = sa.part; const double* __restrict__ cpow = sa.cpow; const double lr = sa.lr;
    float* __restrict__ table16 = sa.table16;
    const int n = *sa.owner_cnt;
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int NQ = N2 >> 4;
    for (int o = blk; o < n; o += nblk) {
        const int4 ow = owners[o];                         // {f, s, e, row}
        const int q0 = ow.y >> 4, q1 = (ow.z - 1) >> 4;
        double sum = 0.0;
        // the row and its decay factor are requested with the partial sums, not after them (one round trip less)
        float* p = table16 + (size_t)ow.w * SLOT + l;
        float wold = 0.f; double cdec = 0.0;
        if (grp == 0 && l < K) { wold = *p; cdec = cpow[ow.z - ow.y]; }
        for (int qb = q0 + grp; qb <= q1; qb += 64) {            // four chunks' partials in flight at a time
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = qb + 16 * k;
                v[k] = q <= q1 ? part[(((size_t)ow.x * NQ + q) * 2 + (q == q0 ? 1 : 0)) * SLOT + l] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += v[k];
        }
        s_sum[grp][l] = sum;
        __syncthreads();
        if (grp == 0 && l < K) {
            double tot = 0.0;
#pragma unroll
            for (int gI = 0; gI < 16; ++gI) tot += s_sum[gI][l];
            *p = (float)((double)wold * cdec - lr * tot);
        }
        __syncthreads();
    }
}

static __global__ __launch_bounds__(256) void k_scat2(const ScatArgs sa)
{
    __shared__ double s_sum[16][16];
    scat2_body(sa, blockIdx.x, gridDim.x, s_sum);
}

// ------------------------------------------------------------------------------------------
// Sparse-row update of the bag table (python/SNN_RBM.py:285-291): ww0[f] -= lr * delta_t for
// every example t that has feature f; no decay, so a row's result is row - lr * (sum of its
// deltas) in example order.  Same sorted records as the FM path; rows are rw floats wide, so a
// thread owns one 16-byte quarter-column of a chunk of 8 sorted entries.
// ------------------------------------------------------------------------------------------
constexpr int WCH = 32;          // sorted entries per chunk on the wide path (4 sub-batches of 8 loads)

__device__ __forceinline__ void scatw1_body(const ScatArgs& sa, const int blk)
{
    const int rw = sa.rw, nq = rw >> 2, N2 = sa.N2, NQ = N2 / WCH;
    const long gid = (long)blk * 256 + threadIdx.x;
    const int chunk = (int)(gid / nq), q = (int)(gid % nq);
    if (chunk >= sa.F * NQ) return;
    const int f = chunk / NQ, qc = chunk % NQ, base = qc * WCH;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    // the records of the NEXT batch of 8 entries are requested together with the gradients / old rows of the current one:
    // one memory round trip per batch instead of two (records, then what they point at)
    int4 rn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rn[j] = sa.rec[(size_t)f * N2 + base + j];
    for (int sb = 0; sb < WCH; sb += 8) {
        int4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = rn[j];
        if (r[0].x < 0) break;                               // invalid keys sort to the end
        if (sb + 8 < WCH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rn[j] = sa.rec[(size_t)f * N2 + base + sb + 8 + j];
        }
        float4 g[8], wold[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = r[j].x >= 0;
            g[j] = *reinterpret_cast<const float4*>(sa.gxp + (size_t)(live ? r[j].y : 0) * sa.K1p + 4 * q);
            // the old row is read only where it is written: at the last entry of a segment that lies inside this chunk
            const int pos = base + sb + j;
            const bool need = live && pos + 1 == r[j].w && r[j].z >= base;
            // (branch-free: entries that do not need it read row 0, which stays in cache)
            wold[j] = *reinterpret_cast<const float4*>(sa.table16 + (size_t)(need ? r[j].x : 0) * rw + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (r[j].x < 0) continue;
            a0 += g[j].x; a1 += g[j].y; a2 += g[j].z; a3 += g[j].w;
            const int pos = base + sb + j, s = r[j].z, e = r[j].w;
            if (pos + 1 != e && pos + 1 != base + WCH) continue;       // the run goes on inside this chunk
            if (s >= base && e <= base + WCH) {                      // the whole segment lies in this chunk
                float* dst = sa.table16 + (size_t)r[j].x * rw + 4 * q;
                if (sa.tag_shared[r[j].x] == sa.stamp)
                    atomic_add4(dst, (float)(-sa.lr * a0), (float)(-sa.lr * a1), (float)(-sa.lr * a2), (float)(-sa.lr * a3));
                else
                    *reinterpret_cast<float4*>(dst) =
                        make_float4((float)(wold[j].x - sa.lr * a0), (float)(wold[j].y - sa.lr * a1),
                                    (float)(wold[j].z - sa.lr * a2), (float)(wold[j].w - sa.lr * a3));
            } else {
                const int which = (s < base) ? 0 : 1;
                double* pp = sa.part + (((size_t)f * NQ + qc) * 2 + which) * rw + 4 * q;
                pp[0] = a0; pp[1] = a1; pp[2] = a2; pp[3] = a3;
                if (which == 1 && q == 0) sa.owners[atomicAdd(sa.owner_cnt, 1)] = make_int4(f, s, e, r[j].x);
            }
            a0 = a1 = a2 = a3 = 0;
        }
    }
}

__device__ __forceinline__ void scatw2_body(const ScatArgs& sa, const int blk, const int nblk, double* s_w /*[1024]*/)
{
    const int rw = sa.rw, nq = rw >> 2, NQ = sa.N2 / WCH, ngrp = 256 / nq;
    const int grp = threadIdx.x / nq, q = threadIdx.x % nq;
    const int n = *sa.owner_cnt;
    for (int o = blk; o < n; o += nblk) {
        const int4 ow = sa.owners[o];                      // {f, s, e, row}
        const int q0 = ow.y / WCH, q1 = (ow.z - 1) / WCH;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        // the row the segment updates is requested with the partial sums, not after them (one round trip less)
        float4* p = reinterpret_cast<float4*>(sa.table16 + (size_t)ow.w * rw + 4 * q);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp == 0) w = *p;
        if (grp < ngrp) {
            for (int qq0 = q0 + grp; qq0 <= q1; qq0 += 4 * ngrp) {           // four chunks' partials in flight at a time
                double v[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int qq = qq0 + k * ngrp;
                    const double* pp = sa.part + (((size_t)ow.x * NQ + (qq <= q1 ? qq : q1)) * 2 + (qq == q0 ? 1 : 0)) * rw + 4 * q;
                    const bool on = qq <= q1;
                    v[k][0] = on ? pp[0] : 0.0; v[k][1] = on ? pp[1] : 0.0; v[k][2] = on ? pp[2] : 0.0; v[k][3] = on ? pp[3] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { a0 += v[k][0]; a1 += v[k][1]; a2 += v[k][2]; a3 += v[k][3]; }
            }
            double* d = s_w + ((size_t)grp * nq + q) * 4;
            d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
        }
        __syncthreads();
        if (grp == 0) {
            double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            for (int gI = 0; gI < ngrp; ++gI) {
                const double* d = s_w + ((size_t)gI * nq + q) * 4;
                t0 += d[0]; t1 += d[1]; t2 += d[2]; t3 += d[3];
            }
            if (sa.tag_shared[ow.w] == sa.stamp)
                atomic_add4(reinterpret_cast<float*>(p), (float)(-sa.lr * t0), (float)(-sa.lr * t1), (float)(-sa.lr * t2), (float)(-sa.lr * t3));
            else
                *p = make_float4((float)(w.x - sa.lr * t0), (float)(w.y - sa.lr * t1), (float)(w.z - sa.lr * t2),
                                 (float)(w.w - sa.lr * t3));
        }
        __syncthreads();
    }
}

// (A workgroup-per-16-examples form with the ids staged through LDS, which took the FM-row gather from 47 to 21 us per 100,000
// examples, was measured on this kernel too and is SLOWER here: 43.8 -> 51.8 us per 16,384 -- one item per thread keeps more row
// loads in flight.)
// reference-shaped output of the bag path: x [B][H0] = sigmoid(sum of the F rows + bb0)
// (python/SNN_RBM.py:248-256).  One thread = 16 bytes of one example's output: its F row pieces are
// F independent 16-byte loads in flight (the ids of an example are the same for all its threads:
// L1 broadcast); consecutive threads read consecutive pieces of the same 800-byte rows.
static __global__ __launch_bounds__(256) void k_bag_ref(const int32_t* __restrict__ ids, int B, int F, int rw,
                                                         const float* __restrict__ table, int64_t n_rows,
                                                         const float* __restrict__ bb0, float* __restrict__ x, int* __restrict__ err, const bool wt)
{
    const int nq = rw >> 2;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)B * nq) return;
    const int t = (int)(gid / nq), c4 = (int)(gid % nq);
    float4 acc = *reinterpret_cast<const float4*>(bb0 + 4 * c4);
    for (int f0 = 0; f0 < F; f0 += 16) {
        int64_t id[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            id[u] = (f0 + u < F) ? ids[(size_t)t * F + f0 + u] : -1;
            if (id[u] < -1 || id[u] >= n_rows) { atomicOr(err, 1); id[u] = -1; }
        }
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            v[u] = id[u] >= 0 ? *reinterpret_cast<const float4*>(table + (size_t)id[u] * rw + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    store16_sel(wt, x + (size_t)t * rw + 4 * c4,
                make_float4(1.0f / (1.0f + expf(-acc.x)), 1.0f / (1.0f + expf(-acc.y)), 1.0f / (1.0f + expf(-acc.z)), 1.0f / (1.0f + expf(-acc.w))));
}
static __global__ void k_axpy(float* __restrict__ y, const float* __restrict__ x, float a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
static __global__ void k_copy_cols(const float* __restrict__ src, int ld, int B, int n, float* __restrict__ dst)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)B * n) return;
    dst[gid] = src[(gid / n) * ld + gid % n];
}

// exact data-parallel mode: this rank's ids [B][F] followed by empty slots (-1) up to `rows` rows -- every rank contributes a
// block of the same size to the all-gather and the gathered order is the global example order
static __global__ void k_pad_ids(const int32_t* __restrict__ ids, int B, int F, int rows, int32_t* __restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)rows * F) return;
    out[gid] = gid < (size_t)B * F ? ids[gid] : -1;
}

// helpers for fnn_set_table / fnn_get_table / fnn_get_rows
static __global__ void k_pack_table(const float* __restrict__ rows, int64_t n_rows, int K, int stride,
                             float* __restrict__ table16)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n_rows * stride) return;
    const size_t r = gid / stride; const int l = (int)(gid % stride);
    table16[gid] = (l < K) ? rows[r * K + l] : 0.f;
}
static __global__ void k_unpack_rows(const float* __restrict__ table16, const int64_t* __restrict__ row_ids,
                              int64_t n, int64_t n_rows, int K, int stride, float* __restrict__ out,
                              int* __restrict__ err)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * K) return;
    const size_t i = gid / K; const int l = (int)(gid % K);
    int64_t r = row_ids ? row_ids[i] : (int64_t)i;
    if (r < 0 || r >= n_rows) { atomicOr(err, 1); out[gid] = 0.f; return; }
    out[gid] = table16[(size_t)r * stride + l];
}

}  // namespace fnn
