// diagnostic harness (never part of the product build): the deep stack of FNN_IP_L7 (batch 4096, bf16, relu) as ONE persistent
// launch of 128 x 128 GEMM tiles with ROW-GROUP dependencies instead of strips or one launch per layer.
//
// Why: the strip kernels (ipnn_api.hip) stream every weight through every strip's CU -- bound by the 64 B/clk a CU ingests -- and
// pay ~5 us of fixed cost per product and strip; per-layer GEMM launches run their loops at 840 TFLOP/s but pay ~10 us of launch,
// fill, epilogue and cold operands each.  Here a tile (layer l, row group i of 128 examples, column tile j) waits only for the
// tiles of (l - 1, i): a counter per (layer, row group), raised by every finished tile behind write-through (sc1) stores of its
// output, polled by the consumer, whose loads of the produced activations are sc1 too (MI355X_MICROARCH.md, valid forms).
// Tiles are dealt layer-major to 256 resident workgroups (tile t to workgroup t mod 256), so a tile's producers always come
// earlier in some workgroup's list: no deadlock while all workgroups are resident.
//
// Prints: time of the chain with and without the dependency waits (the latter computes garbage, it prices the hand-offs), of the
// same tiles launched layer by layer, and a checksum comparison of the two correct variants.
#include "../../deep-ctr_amd/csrc/fnn_kernels.hip.h"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace fnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 8, RG = 32;                                 // products; row groups of 128 examples (batch 4096)
struct ChainArgs {
    const bf16_t* W[NL]; bf16_t* a[NL + 1]; int Dp[NL + 1]; int tile0[NL + 1]; int nct[NL];
    int* cnt; int epoch; int sync; int lo, hi;                  // tiles [lo, hi) of the list; sync: wait for the producers
    int* err;
};

__device__ __forceinline__ bf16x8 ld_sc1(const bf16_t* base, size_t elem_off, size_t bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem_off * 2), 0, 16);      // aux 16 = sc1: served by L2, never by a stale L1 line
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void st_sc1(bf16_t* base, size_t elem_off, size_t bytes, bf16x8 v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(elem_off * 2), 0, 16);   // write-through
}

static __global__ __launch_bounds__(256) void k_chain(const ChainArgs g)
{
    __shared__ __align__(16) bf16_t s_tile[4][16][64 + 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wr = wave & 1, wc = wave >> 1;
    for (int t = g.lo + (int)blockIdx.x; t < g.hi; t += (int)gridDim.x) {
        int l = 0;
#pragma unroll
        for (int q = 1; q < NL; ++q) l += t >= g.tile0[q] ? 1 : 0;
        const int local = t - g.tile0[l], j = local % g.nct[l], i = local / g.nct[l];
        const int K = g.Dp[l], N = g.Dp[l + 1], nkt = K / 32;
        if (l > 0 && g.sync) {
            if (threadIdx.x == 0) {
                const int want = g.epoch * g.nct[l - 1];
                int n = 0;
                while (__hip_atomic_load(g.cnt + (l - 1) * RG + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++n > (1 << 22)) { atomicOr(g.err, 1); break; }
                }
            }
            __syncthreads();
        }
        const int rt0 = i * 8 + wr * 4, ct0 = j * 8 + wc * 4;          // 16-row fragments of A (examples) and of W (output units)
        const bool active = ct0 * 16 < N;
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) {
            const bf16_t* A = g.a[l];
            const bf16_t* Wm = g.W[l];
            const size_t abytes = (size_t)4096 * K * 2;
            const int nct16 = N / 16;
            bf16x8 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
            auto load = [&](bf16x8* a, bf16x8* b, int kt) {
                if (kt >= nkt) return;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const size_t off = ((size_t)((rt0 + m) * nkt + kt) * 64 + lane) * 8;
                    a[m] = (l > 0) ? ld_sc1(A, off, abytes) : *reinterpret_cast<const bf16x8*>(A + off);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int ct = min(ct0 + n, nct16 - 1);
                    b[n] = *reinterpret_cast<const bf16x8*>(ft_frag<bf16_t>(Wm, ct, kt, nkt, lane));
                }
            };
            auto mul = [&](const bf16x8* a, const bf16x8* b) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma(acc[m][n], a[m], b[n]);
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
            // epilogue: relu, F layout (rows = examples, k = units) through a wave-private LDS tile, whole 1-KiB fragments out
            bf16_t* O = g.a[l + 1];
            const size_t obytes = (size_t)4096 * N * 2;
            const int rq = 4 * (lane >> 4), cl = lane & 15, nkto = N / 32;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_tile[wave][rq + r][n * 16 + cl] = (bf16_t)fmaxf(acc[m][n][r] * 0.05f, 0.f);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if ((ct0 * 16) / 32 + kk >= nkto) continue;
                    const bf16x8 f = *reinterpret_cast<const bf16x8*>(&s_tile[wave][lane & 15][kk * 32 + (lane >> 4) * 8]);
                    st_sc1(O, ((size_t)((rt0 + m) * nkto + (ct0 * 16) / 32 + kk) * 64 + lane) * 8, obytes, f);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have left
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(g.cnt + l * RG + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main()
{
    const int Dp[NL + 1] = {320, 1024, 832, 640, 448, 256, 128, 64, 64};
    const int B = 4096;
    std::mt19937 rng(1);
    ChainArgs g{};
    auto dev = [&](size_t bytes) { void* p = nullptr; hipMalloc(&p, bytes); hipMemset(p, 0, bytes); return p; };
    double flops = 0;
    int ntiles = 0;
    for (int l = 0; l < NL; ++l) {
        g.Dp[l] = Dp[l];
        std::vector<unsigned short> w((size_t)Dp[l + 1] * Dp[l]);
        for (auto& v : w) v = (unsigned short)(0x3c00 + (rng() % 128) - ((rng() & 1) ? 0x8000 : 0));    // small +- bf16 values
        bf16_t* W = (bf16_t*)dev(w.size() * 2);
        CK(hipMemcpy(W, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        g.W[l] = W;
        g.tile0[l] = ntiles; g.nct[l] = (Dp[l + 1] + 127) / 128;
        ntiles += RG * g.nct[l];
        flops += 2.0 * B * Dp[l] * Dp[l + 1];
    }
    g.Dp[NL] = Dp[NL]; g.tile0[NL] = ntiles;
    for (int l = 0; l <= NL; ++l) g.a[l] = (bf16_t*)dev((size_t)B * Dp[l] * 2);
    {
        std::vector<unsigned short> x((size_t)B * Dp[0]);
        for (auto& v : x) v = (unsigned short)(0x3c00 + (rng() % 64));
        CK(hipMemcpy(g.a[0], x.data(), x.size() * 2, hipMemcpyHostToDevice));
    }
    g.cnt = (int*)dev(NL * RG * 4); g.err = (int*)dev(4);
    printf("%d tiles of 128 x 128, %.2f GFLOP per pass\n", ntiles, flops * 1e-9);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto checksum = [&](double& s) { std::vector<unsigned short> o((size_t)B * Dp[NL]); hipMemcpy(o.data(), g.a[NL], o.size() * 2, hipMemcpyDeviceToHost);
                                      s = 0; for (size_t k = 0; k < o.size(); ++k) s += (double)(o[k] >> 7) * ((k % 97) + 1); };
    int epoch = 0;
    auto chain = [&](int sync, int wgs) { g.sync = sync; g.lo = 0; g.hi = ntiles; g.epoch = ++epoch; hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), 0, 0, g); };
    auto layered = [&]() { ++epoch; for (int l = 0; l < NL; ++l) { g.sync = 0; g.lo = g.tile0[l]; g.hi = g.tile0[l + 1]; g.epoch = epoch;
                                                              hipLaunchKernelGGL(k_chain, dim3(std::min(256, g.hi - g.lo)), dim3(256), 0, 0, g); } };
    double c_layer = 0, c_chain = 0;
    layered(); CK(hipDeviceSynchronize()); checksum(c_layer);
    CK(hipMemset(g.a[NL], 0, (size_t)B * Dp[NL] * 2));
    chain(1, 256); CK(hipDeviceSynchronize()); checksum(c_chain);
    int herr = 0; CK(hipMemcpy(&herr, g.err, 4, hipMemcpyDeviceToHost));
    printf("checksum layer-by-layer %.0f, chain %.0f  (%s)  spin-limit flag %d\n", c_layer, c_chain, c_layer == c_chain ? "equal" : "DIFFERENT", herr);
    for (int variant = 0; variant < 4; ++variant) {
        const char* name[4] = {"layer by layer (8 launches)", "chain, dependency waits, 256 WGs", "chain WITHOUT waits (garbage; prices the hand-offs)", "chain, dependency waits, 512 WGs"};
        for (int it = 0; it < 5; ++it) { if (variant == 0) layered(); else chain(variant != 2, variant == 3 ? 512 : 256); }
        hipEventRecord(e0, 0);
        const int n = 30;
        for (int it = 0; it < n; ++it) { if (variant == 0) layered(); else chain(variant != 2, variant == 3 ? 512 : 256); }
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-56s %7.1f us per pass = %6.0f TFLOP/s\n", name[variant], ms * 1000 / n, flops / (ms * 1e-3 / n) * 1e-12);
    }
    CK(hipMemcpy(&herr, g.err, 4, hipMemcpyDeviceToHost));
    printf("spin-limit flag at the end: %d\n", herr);
    return 0;
}
