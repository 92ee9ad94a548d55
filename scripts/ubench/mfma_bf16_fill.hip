// Micro-benchmark: v_mfma_f32_32x32x16_bf16 on gfx950 -- issue rate of a dependent chain, how many independent VALU /
// LDS instructions of the SAME wave hide behind it, and whether the co-resident wave's VALU work overlaps with it
// (it does not for v_mfma_f32_32x32x2_f32, see mfma_fill.hip / mfma_valu_overlap.hip).
// Also: the VALU price of the three-way bf16 split of an fp32 register pair (round-to-nearest-even pieces).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x, m, k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k))
#define LDSR(v, p) asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(p))

// KIND 0: every wave runs the same stream (1 MFMA per chain step + NV VALU + NL LDS reads)
// KIND 1: waves 0-3 run the MFMA chain, waves 4-7 (512-thread launches only) run a pure VALU stream of n * NV FMAs
template <int NV, int NL, int CHAINS, int KIND>
__global__ __launch_bounds__(512) void k(int n, float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = lane * 1e-3f, m = 1.0001f, kk = 0.5f;
    i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane + 4, lane + 5, lane + 6, lane + 7};
    f32x16 c0 = {0}, c1 = {0};
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i;
    f32x4 l[4] = {};
    unsigned p = (unsigned)(lane * 16);
    if (KIND == 1 && wave >= 4) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) FMA(x[i % 16], m, kk);
        }
    } else {
        for (int it = 0; it < n; ++it) {
            MFMA(c0, a, b);
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) FMA(x[i % 16], m, kk);
#pragma unroll
                for (int i = 0; i < NL; ++i) LDSR(l[i % 4], p);
            }
            if (CHAINS == 2) {
                MFMA(c1, a, b);
                if (KIND == 0) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) FMA(x[i % 16], m, kk);
#pragma unroll
                    for (int i = 0; i < NL; ++i) LDSR(l[i % 4], p);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float r = c0[0] + c1[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += l[i][0];
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int NV, int NL, int CHAINS, int KIND>
void run(const char* name, int threads, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = CHAINS == 2 ? 1000 : 2000;         // 2000 MFMAs per MFMA wave either way
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NL, CHAINS, KIND>), dim3(256), dim3(threads), 0, 0, n, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const int mfma_waves = KIND == 1 ? 1 : threads / 256;
    printf("%-34s waves/SIMD %d : %7.1f us  (%.1f ns per MFMA per SIMD)\n", name, threads / 256, best * 1e3f,
           best * 1e6f / 2000.f / mfma_waves);
}

// split cost: 16 fp32 registers -> 3 x 8 packed bf16 registers (RNE pieces), repeated n times on changing data
__global__ __launch_bounds__(256) void split_k(int n, float* out) {
    const int lane = threadIdx.x & 63;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = lane * 1e-3f + i * 0.37f;
    unsigned accp = 0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            unsigned p0, p1, p2;
            float a = x[i], b = x[i + 1];
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
            float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
            float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(sa), "v"(sb));
            accp ^= p0 + p1 + p2;
            x[i] = a * 1.0001f; x[i + 1] = b * 0.9999f;      // 2 extra VALU per pair (subtracted below)
        }
    }
    if (accp == 0x12345u) out[threadIdx.x] = (float)accp;
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    for (int threads : {256, 512}) {
        run<0, 0, 1, 0>("bf16 mfma only", threads, out);
        run<0, 0, 2, 0>("2 chains: bf16 mfma only", threads, out);
        run<2, 0, 1, 0>("bf16 mfma + 2 valu", threads, out);
        run<4, 0, 1, 0>("bf16 mfma + 4 valu", threads, out);
        run<6, 0, 1, 0>("bf16 mfma + 6 valu", threads, out);
        run<8, 0, 1, 0>("bf16 mfma + 8 valu", threads, out);
        run<12, 0, 1, 0>("bf16 mfma + 12 valu", threads, out);
        run<16, 0, 1, 0>("bf16 mfma + 16 valu", threads, out);
        run<4, 0, 2, 0>("2 chains: bf16 mfma + 4 valu", threads, out);
        run<8, 0, 2, 0>("2 chains: bf16 mfma + 8 valu", threads, out);
        run<0, 1, 1, 0>("bf16 mfma + 1 ds_read_b128", threads, out);
        run<0, 2, 1, 0>("bf16 mfma + 2 ds_read_b128", threads, out);
        run<4, 1, 1, 0>("bf16 mfma + 4 valu + 1 lds", threads, out);
    }
    // co-resident waves: waves 0-3 MFMA chain (2000), waves 4-7 VALU (2000 * NV FMAs)
    run<4, 0, 1, 1>("A: bf16 mfma | B: 4 valu per mfma", 512, out);
    run<8, 0, 1, 1>("A: bf16 mfma | B: 8 valu per mfma", 512, out);
    run<16, 0, 1, 1>("A: bf16 mfma | B: 16 valu per mfma", 512, out);
    {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e9f;
        const int n = 2000;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(split_k, dim3(256), dim3(256), 0, 0, n, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("3-way bf16 split of 16 fp32 registers (+16 filler VALU): %.1f ns per 16 registers, one wave per SIMD\n",
               best * 1e6f / n);
    }
    return 0;
}
