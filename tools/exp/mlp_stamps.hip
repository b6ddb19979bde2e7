// diagnostic harness: per-phase time stamps of the MLP strip kernel (never part of the product build)
#define FNN_STAMPS 1
// NOTE: k_step1 signature: (MlpArgs)
#include "../../deep-ctr_amd/csrc/fnn_step_kernels.hip.h"
#include <algorithm>
#include <cstdio>
#include <vector>
#include <random>
using namespace fnn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int B = 4096, F = 16, K = 11, H1 = 300, H2 = 100, K1p = 256, H1p = 320, H2p = 128, ldT = 4096;
    const int64_t D = 937670;
    std::mt19937 rng(1);
    std::vector<int32_t> ids(B * F); for (auto& v : ids) v = rng() % D;
    const char* ids_kind = "uniform over the table";
    if (FILE* f = fopen("gpurun_out/zipf_ids.bin", "rb")) {          // the bench's Zipf(1.1) ids of one batch (tools/step1_phases.sh writes them)
        if (fread(ids.data(), 4, ids.size(), f) == ids.size()) ids_kind = "Zipf(1.1) per field (synth.zipf_ids, the bench's batch 0)";
        fclose(f);
    }
    std::vector<float> tab(D * 16); for (auto& v : tab) v = (rng() % 1000) * 1e-4f - 0.05f;
    std::vector<float> y(B, 0.f);
    auto dev = [&](size_t bytes) { void* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes); return p; };
    int32_t* d_ids = (int32_t*)dev(ids.size() * 4); float* d_tab = (float*)dev(tab.size() * 4); float* d_y = (float*)dev(B * 4);
    CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> w(K1p * H1p); for (auto& v : w) v = 0x3c00 + (rng() % 64);   // small bf16 values
    bf16_t* w1 = (bf16_t*)dev(K1p * H1p * 2); bf16_t* w1t = (bf16_t*)dev(K1p * H1p * 2);
    bf16_t* w2 = (bf16_t*)dev(H1p * H2p * 2); bf16_t* w2t = (bf16_t*)dev(H1p * H2p * 2);
    CK(hipMemcpy(w1, w.data(), K1p * H1p * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w1t, w.data(), K1p * H1p * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, w.data(), H1p * H2p * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w2t, w.data(), H1p * H2p * 2, hipMemcpyHostToDevice));
    float* w3p = (float*)dev(H2p * 4);
    uint8_t* m = (uint8_t*)dev(512); hipMemset(m, 1, 512);
    bf16_t* xpT = (bf16_t*)dev((size_t)K1p * ldT * 2); bf16_t* d1T = (bf16_t*)dev((size_t)H1p * ldT * 2); bf16_t* d2T = (bf16_t*)dev((size_t)H2p * ldT * 2);
    bf16_t* dl1T = (bf16_t*)dev((size_t)H1p * ldT * 2); bf16_t* dl2T = (bf16_t*)dev((size_t)H2p * ldT * 2); bf16_t* dl3T = (bf16_t*)dev((size_t)64 * ldT * 2);
    float* gxp = (float*)dev((size_t)B * K1p * 4); float* p_out = (float*)dev(B * 4); float* loss_t = (float*)dev(B * 4); int* err = (int*)dev(4);
    long long* dbg = (long long*)dev(256 * 16 * 8);
    MlpArgs<bf16_t> a{d_ids, d_y, B, F, K, d_tab, D, -3.f, w1, w1t, w2, w2t, w3p, m, m, 0, 0, H1, H2, 1,
                      xpT, d1T, d2T, dl1T, dl2T, dl3T, ldT, gxp, p_out, loss_t, err, nullptr, 16, nullptr, nullptr, dbg};
    const size_t lds = 16 * (328 + 328 + 136) * 2 + 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k_step1<bf16_t, 5, 2, 4, false>), dim3(256), dim3(256), lds, 0, a);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL((k_step1<bf16_t, 5, 2, 4, false>), dim3(256), dim3(256), lds, 0, a);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("k_step1 (mlp only, with stamps) %.2f us/launch\n", ms * 1000 / 50);
    std::vector<long long> hd(256 * 16); CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
    const char* names[10] = {"init loads+P0 gather+barrier", "P1 mfma loop", "P1 epilogue+barrier", "P2 mfma loop", "P2 epi+head (2 barriers)",
                             "delta2+barrier", "P3 mfma loop", "P3 epilogue+barrier", "P4 mfma loop", "P4 stores"};
    for (int i = 0; i < 10; ++i) {
        std::vector<long long> d; for (int b = 0; b < 256; ++b) d.push_back(hd[b * 16 + i + 1] - hd[b * 16 + i]);
        std::sort(d.begin(), d.end());
        printf("  %-32s median %6lld  min %6lld  max %6lld  (ticks)\n", names[i], d[128], d[0], d[255]);
    }
    std::vector<long long> tot; for (int b = 0; b < 256; ++b) tot.push_back(hd[b * 16 + 10] - hd[b * 16]);
    std::sort(tot.begin(), tot.end()); printf("  total median %lld ticks; spread of block start: ", tot[128]);
    long long s0 = hd[0], s1 = hd[0]; for (int b = 0; b < 256; ++b) { s0 = std::min(s0, hd[b * 16]); s1 = std::max(s1, hd[b * 16]); }
    long long e_max = 0; for (int b = 0; b < 256; ++b) e_max = std::max(e_max, hd[b * 16 + 10]);
    printf("%lld ticks; first start -> last end %lld ticks\n", s1 - s0, e_max - s0);
    // ticks -> microseconds: s_memtime ticks of a workgroup's body over its s_memrealtime (100 MHz) span, median over workgroups
    std::vector<double> tpu; for (int b = 0; b < 256; ++b) { const long long rt = hd[b * 16 + 15] - hd[b * 16 + 14]; if (rt > 0) tpu.push_back((double)(hd[b * 16 + 10] - hd[b * 16]) / ((double)rt / 100.0)); }
    std::sort(tpu.begin(), tpu.end());
    const double ticks_per_us = tpu.empty() ? 2400.0 : tpu[tpu.size() / 2];
    if (FILE* f = fopen("gpurun_out/step1_phases.json", "w")) {
        fprintf(f, "{\"kernel\": \"k_step1<bf16, 5, 2, 4> (mlp_body), diagnostic build with s_memtime stamps (tools/exp/mlp_stamps.hip); the product kernel executes no stamp\", ");
        fprintf(f, "\"batch\": %d, \"workgroups\": 256, \"ids\": \"%s\", \"ticks_per_us\": %.1f, \"launch_us_with_stamps\": %.2f, ", B, ids_kind, ticks_per_us, ms * 1000 / 50);
        fprintf(f, "\"phases_us_median\": {");
        for (int i = 0; i < 10; ++i) {
            std::vector<long long> d; for (int b = 0; b < 256; ++b) d.push_back(hd[b * 16 + i + 1] - hd[b * 16 + i]);
            std::sort(d.begin(), d.end());
            fprintf(f, "%s\"%s\": %.3f", i ? ", " : "", names[i], d[128] / ticks_per_us);
        }
        std::vector<long long> d0; for (int b = 0; b < 256; ++b) d0.push_back(hd[b * 16 + 1] - hd[b * 16]);
        std::sort(d0.begin(), d0.end());
        fprintf(f, "}, \"gather_phase_us_median\": %.3f, \"gather_phase_us_min\": %.3f, \"gather_phase_us_max\": %.3f, \"body_us_median\": %.3f}\n",
                d0[128] / ticks_per_us, d0[0] / ticks_per_us, d0[255] / ticks_per_us, tot[128] / ticks_per_us);
        fclose(f);
    }
    return 0;
}
