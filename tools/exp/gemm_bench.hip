// diagnostic harness (never part of the product build): the inner-product family's GEMM kernels alone on the
// FNN_IP_L7 layer shapes -- time per launch, phase stamps of k_gemm_lds, and a check against a CPU product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DGEMM_STAMPS -o /tmp/gb tools/exp/gemm_bench.hip
#include "../../deep-ctr_amd/csrc/fnn_kernels.hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace fnn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T> struct EpiAct {           // like the forward epilogue: keep-mask (transposed bytes) * relu
    static constexpr bool TILE = true;
    T* outF; int ld; T* outT; int ldT; const uint8_t* maskT; float inv_keep;
    __device__ void pre(int r0, int col, const f32x4& acc, float v[4]) const {
        unsigned mb = 0x01010101u;
        if (maskT) mb = *reinterpret_cast<const unsigned*>(maskT + (size_t)col * ldT + r0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[r], 0.f) * ((float)((mb >> (8 * r)) & 0xffu) * inv_keep);
    }
};

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;          // 0 = k_gemm_lds, 1 = k_gemm_ft<4,4>
    const int M = 4096, ldT = 4096;
    struct Shape { int N, K; const char* name; } shapes[] = {{1024, 320, "fwd1"}, {832, 1024, "fwd2"}, {640, 832, "fwd3"}, {448, 640, "fwd4"}};
    std::mt19937 rng(1);
    auto dev = [&](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
    long long* dbg = (long long*)dev((size_t)4096 * 8 * 8);
#ifdef GEMM_STAMPS
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), &dbg, sizeof(dbg)));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const int N = sh.N, K = sh.K, nkt = K / 32;
        std::vector<unsigned short> A((size_t)M * K), Bw((size_t)N * K);       // fragment-tiled images
        std::vector<float> Ar((size_t)M * K), Br((size_t)N * K);
        for (int r = 0; r < M; ++r) for (int k = 0; k < K; ++k) { const float v = ((int)(rng() % 65) - 32) / 64.f; Ar[(size_t)r * K + k] = bf2f(f2bf(v)); A[ft_off<bf16_t>(r, k, K)] = f2bf(v); }
        for (int r = 0; r < N; ++r) for (int k = 0; k < K; ++k) { const float v = ((int)(rng() % 65) - 32) / 256.f; Br[(size_t)r * K + k] = bf2f(f2bf(v)); Bw[ft_off<bf16_t>(r, k, K)] = f2bf(v); }
        bf16_t* dA = (bf16_t*)dev(A.size() * 2); bf16_t* dB = (bf16_t*)dev(Bw.size() * 2);
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bw.data(), Bw.size() * 2, hipMemcpyHostToDevice));
        bf16_t* oF = (bf16_t*)dev((size_t)M * N * 2); bf16_t* oT = (bf16_t*)dev((size_t)N * ldT * 2);
        uint8_t* mk = (uint8_t*)dev((size_t)N * ldT); CK(hipMemset(mk, 1, (size_t)N * ldT));
        EpiAct<bf16_t> e{oF, N, oT, ldT, mk, 2.0f};
        const dim3 grid((M + 127) / 128, (N + 127) / 128, 1);
        auto launch = [&]() {
            if (variant == 0) {
                const size_t lds = gemm_lds_bytes<bf16_t, true>();
                static bool set = false;
                if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_lds<bf16_t, EpiAct<bf16_t>>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
                hipLaunchKernelGGL((k_gemm_lds<bf16_t, EpiAct<bf16_t>>), grid, dim3(256), lds, 0, dA, dB, M / 16, N / 16, nkt, nkt, e);
            } else {
                const size_t lds = gemm_ft_lds<bf16_t, 4>();
                hipLaunchKernelGGL((k_gemm_ft<bf16_t, 4, 4, EpiAct<bf16_t>>), grid, dim3(256), lds, 0, dA, dB, M / 16, N / 16, nkt, nkt, e);
            }
        };
        for (int it = 0; it < 10; ++it) launch();
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < 50; ++it) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / 50;
        // check 64 sampled outputs
        std::vector<unsigned short> hF((size_t)M * N), hT((size_t)N * ldT);
        CK(hipMemcpy(hF.data(), oF, hF.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hT.data(), oT, hT.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int s = 0; s < 4096; ++s) {
            const int r = rng() % M, c = rng() % N;
            double acc = 0; for (int k = 0; k < K; ++k) acc += (double)Ar[(size_t)r * K + k] * Br[(size_t)c * K + k];
            const double ref = std::max(acc, 0.0) * 2.0;
            const double gF = bf2f(hF[ft_off<bf16_t>(r, c, N)]), gT = bf2f(hT[ft_off<bf16_t>(c, r, ldT)]);
            maxerr = std::max(maxerr, std::max(std::abs(gF - ref), std::abs(gT - ref)) / (1.0 + std::abs(ref)));
        }
        printf("%s M=%d N=%d K=%d: %.2f us/launch  %.0f TFLOP/s  max rel err %.4f", sh.name, M, N, K, us, 2.0 * M * N * K / us * 1e-6, maxerr);
#ifdef GEMM_STAMPS
        if (variant == 0) {
            const int nwg = grid.x * grid.y;
            std::vector<long long> st((size_t)nwg * 8);
            CK(hipMemcpy(st.data(), dbg, st.size() * 8, hipMemcpyDeviceToHost));
            long long tmin = st[0]; for (int w = 0; w < nwg; ++w) tmin = std::min(tmin, st[(size_t)w * 8]);
            double ph[4] = {0, 0, 0, 0}; long long tend = 0;
            for (int w = 0; w < nwg; ++w) { for (int i = 0; i < 3; ++i) ph[i] += (double)(st[(size_t)w * 8 + i + 1] - st[(size_t)w * 8 + i]); ph[3] += (double)(st[(size_t)w * 8] - tmin); tend = std::max(tend, st[(size_t)w * 8 + 3]); }
            // s_memtime ticks at 100 MHz
            printf("   [avg per WG, us: start skew %.2f | prologue %.2f | loop %.2f | epilogue %.2f | first start -> last end %.2f]", ph[3] / nwg / 100, ph[0] / nwg / 100, ph[1] / nwg / 100, ph[2] / nwg / 100, (double)(tend - tmin) / 100);
        }
#endif
        printf("\n");
        hipFree(dA); hipFree(dB); hipFree(oF); hipFree(oT); hipFree(mk);
    }
    return 0;
}
