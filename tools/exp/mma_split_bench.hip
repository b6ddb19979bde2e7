// Micro-benchmark: cost of one k16 "fragment pair -> accumulate" in three arithmetic forms, operands in registers:
//   f32   : 4 x v_mfma_f32_16x16x4_f32                     (FNN_PREC_F32 today)
//   split : {hi, lo} bf16 pairs per element, 3 x v_mfma_f32_16x16x16_bf16 (hi*hi + hi*lo + lo*hi) behind v_perm de-interleaves
//   bf16  : 1 x v_mfma_f32_16x16x32_bf16 per k32           (FNN_PREC_BF16 today; shown per k16 = half an instruction)
// and the error of the split form against f64 on random data.  hipcc --offload-arch=gfx950 -O3 mma_split_bench.hip -o mma_split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ inline void mma_f32(f32x4& acc, f32x4 a, f32x4 b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
}
union U2 { unsigned u[2]; s16x4 s; };
__device__ inline void split_planes(u32x4 v, s16x4& hi, s16x4& lo) {   // element = hi | lo << 16
    U2 h, l;
    h.u[0] = __builtin_amdgcn_perm(v[1], v[0], 0x05040100); h.u[1] = __builtin_amdgcn_perm(v[3], v[2], 0x05040100);
    l.u[0] = __builtin_amdgcn_perm(v[1], v[0], 0x07060302); l.u[1] = __builtin_amdgcn_perm(v[3], v[2], 0x07060302);
    hi = h.s; lo = l.s;
}
__device__ inline void mma_split(f32x4& acc, s16x4 ah, s16x4 al, s16x4 bh, s16x4 bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
}
template <int MODE> __global__ void k_time(const float* src, float* out, long long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[5];
    for (int n = 0; n < 5; ++n) acc[n] = f32x4{0, 0, 0, 0};
    f32x4 a = *reinterpret_cast<const f32x4*>(src + lane * 4);
    f32x4 b[5];
    for (int n = 0; n < 5; ++n) b[n] = *reinterpret_cast<const f32x4*>(src + 256 + (n * 64 + lane) * 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int n = 0; n < 5; ++n) mma_f32(acc[n], a, b[n]);
        } else if (MODE == 1) {
            s16x4 ah, al; split_planes(__builtin_bit_cast(u32x4, a), ah, al);
#pragma unroll
            for (int n = 0; n < 5; ++n) { s16x4 bh, bl; split_planes(__builtin_bit_cast(u32x4, b[n]), bh, bl); mma_split(acc[n], ah, al, bh, bl); }
        } else {
            const bf16x8 a8 = __builtin_bit_cast(bf16x8, a);
#pragma unroll
            for (int n = 0; n < 5; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, b[n]), acc[n], 0, 0, 0);
        }
        a[0] += 1e-30f;                      // loop-carried: the operands are not loop-invariant for the optimiser
        asm volatile("" : "+v"(a));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < 5; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// accuracy: C[16x16] = A[16xK] B[Kx16], K = 256, split form against f64
__device__ inline unsigned to_split(float v) {
    const __bf16 h = (__bf16)v; const __bf16 l = (__bf16)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | (unsigned)__builtin_bit_cast(unsigned short, l) << 16;
}
__global__ void k_acc(const float* A, const float* B, int K, float* C_split, float* C_f32, float* C_bf16) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 cs{0, 0, 0, 0}, cf{0, 0, 0, 0}, cb{0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 a, b; u32x4 as, bs;
        for (int j = 0; j < 4; ++j) { a[j] = A[r * K + k0 + 4 * q + j]; b[j] = B[(k0 + 4 * q + j) * 16 + r]; as[j] = to_split(a[j]); bs[j] = to_split(b[j]); }
        mma_f32(cf, a, b);
        s16x4 ah, al, bh, bl; split_planes(as, ah, al); split_planes(bs, bh, bl);
        mma_split(cs, ah, al, bh, bl);
        s16x4 z{0, 0, 0, 0};
        cb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, cb, 0, 0, 0);
        (void)z;
    }
    for (int j = 0; j < 4; ++j) { C_split[(4 * q + j) * 16 + r] = cs[j]; C_f32[(4 * q + j) * 16 + r] = cf[j]; C_bf16[(4 * q + j) * 16 + r] = cb[j]; }
}
int main() {
    const int K = 256;
    std::vector<float> A(16 * K), B(K * 16);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : B) v = (float)rand() / RAND_MAX * 2 - 1;
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 3 * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_acc<<<1, 64>>>(dA, dB, K, dC, dC + 256, dC + 512);
    std::vector<float> C(768); hipMemcpy(C.data(), dC, 768 * 4, hipMemcpyDeviceToHost);
    double es = 0, ef = 0, eb = 0, nr = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + j];
        es = fmax(es, fabs(C[i * 16 + j] - s)); ef = fmax(ef, fabs(C[256 + i * 16 + j] - s)); eb = fmax(eb, fabs(C[512 + i * 16 + j] - s)); nr = fmax(nr, fabs(s));
    }
    printf("{\"accuracy_K256_uniform\": {\"max_abs_err_split\": %.3e, \"max_abs_err_f32_mfma\": %.3e, \"max_abs_err_bf16\": %.3e, \"max_abs_value\": %.3f},\n", es, ef, eb, nr);
    float* src; hipMalloc(&src, 4096 * 4); hipMemset(src, 0, 4096 * 4);
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    long long* tk; hipMalloc(&tk, 1024 * 8);
    const int iters = 2000;
    const char* names[3] = {"f32_4x16x16x4", "split_3x16x16x16_bf16", "bf16_1x16x16x32_per_k32"};
    printf(" \"cycles_per_5_tiles\": {");
    for (int wpb = 4; wpb <= 8; wpb += 4)
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) k_time<0><<<256, 64 * wpb>>>(src, out, tk, iters);
            if (mode == 1) k_time<1><<<256, 64 * wpb>>>(src, out, tk, iters);
            if (mode == 2) k_time<2><<<256, 64 * wpb>>>(src, out, tk, iters);
            hipDeviceSynchronize();
            std::vector<long long> t(256); hipMemcpy(t.data(), tk, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : t) s += v;
            printf("%s\"%s_%dwaves_per_cu\": %.1f", (wpb == 4 && mode == 0) ? "" : ", ", names[mode], wpb, s / 256 / iters);
        }
    printf("}, \"note\": \"s_memtime ticks (100 MHz-domain scaled by the runtime: compare ratios) per loop trip of 5 n-tiles at k16 (bf16: k32)\"}\n");
    return 0;
}
