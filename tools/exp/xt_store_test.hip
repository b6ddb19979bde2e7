// Stand-alone check of the 16-byte transposed-operand store built from v_permlane16_swap (not product code): every lane holds 4 bf16
// "examples" of one unit; even 16-lane rows store {own 8 bytes, the next row's 8 bytes} written through.  Verifies the bytes on the
// host, over many launches, with a consumer kernel reading them right behind the producer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void store16_wt(void* p, u32x4 w) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory"); }
__global__ void produce(unsigned* out, unsigned salt, int mode)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned gid = blockIdx.x * 4 + wave;                 // one "fragment" of 64 lanes x 8 bytes per wave
    // lane (lq = lane >> 4, cl = lane & 15) holds dwords d0, d1 = f(unit cl, examples 4 lq ..)
    unsigned d0 = (gid * 64 + lane) * 2 + salt, d1 = d0 + 1;
    d0 = d0 * 2654435761u; d1 = d1 * 2654435761u;               // VALU writes right in front of the swap
    unsigned p0, p1;
    if (mode == 0) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(d0, d0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(d1, d1, false, false);
        p0 = s0[1]; p1 = s1[1];
    } else {
        p0 = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, d0); p1 = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, d1);
    }
    // slot layout: [gid][row pair (lq >> 1)][cl] of 16 bytes
    if (((lane >> 4) & 1) == 0) store16_wt(out + ((size_t)gid * 32 + (lane >> 5) * 16 + (lane & 15)) * 4, u32x4{d0, d1, p0, p1});
}
__global__ void consume(const unsigned* in, unsigned long long* sum, size_t n4)
{
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        s += (unsigned long long)v.x + v.y * 3ull + v.z * 5ull + v.w * 7ull;
    }
    atomicAdd(sum, s);
}
int main(int argc, char** argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 4096, nfrag = nblk * 4; const size_t n4 = (size_t)nfrag * 32, nword = n4 * 4;
    unsigned* buf; unsigned long long* sum;
    hipMalloc(&buf, nword * 4); hipMalloc(&sum, 8);
    std::vector<unsigned> host(nword);
    int bad_runs = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int it = 0; it < 200; ++it) {
            const unsigned salt = 1000003u * it + mode;
            hipMemsetAsync(sum, 0, 8, 0);
            hipLaunchKernelGGL(produce, dim3(nblk), dim3(256), 0, 0, buf, salt, mode);
            hipLaunchKernelGGL(consume, dim3(1024), dim3(256), 0, 0, buf, sum, n4);
            unsigned long long got = 0, want = 0;
            hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost);
            hipMemcpy(host.data(), buf, nword * 4, hipMemcpyDeviceToHost);
            size_t wrong = 0;
            for (int g = 0; g < nfrag; ++g)
                for (int pr = 0; pr < 2; ++pr)
                    for (int cl = 0; cl < 16; ++cl) {
                        const int lane = pr * 32 + cl;
                        unsigned e[4];
                        for (int h = 0; h < 2; ++h) {
                            const unsigned d0 = ((unsigned)(g * 64 + lane + 16 * h) * 2 + salt);
                            e[2 * h] = d0 * 2654435761u; e[2 * h + 1] = (d0 + 1) * 2654435761u;
                        }
                        const unsigned* v = &host[((size_t)g * 32 + pr * 16 + cl) * 4];
                        for (int k = 0; k < 4; ++k) wrong += v[k] != e[k];
                        want += (unsigned long long)e[0] + e[1] * 3ull + e[2] * 5ull + e[3] * 7ull;
                    }
            if (wrong || got != want) { ++bad_runs; if (bad_runs < 6) printf("mode %d it %d: %zu wrong words on the host, consumer sum %s\n", mode, it, wrong, got == want ? "ok" : "DIFFERS"); }
        }
    printf("bad runs: %d of 400\n", bad_runs);
    return bad_runs != 0;
}
