// Micro-benchmark: how many independent VALU / LDS instructions of the SAME wave hide behind a dependent chain of
// v_mfma_f32_32x32x2_f32 (64 cycles each)?  One wave per SIMD (256-thread workgroups, 1 per CU) or two (512).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x, m, k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k))
#define LDSR(v, p) asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(p))

template <int NV, int NL, int CHAINS>
__global__ __launch_bounds__(512) void k(int n, float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a = lane * 1e-3f, b = a + 1.f, m = 1.0001f, kk = 0.5f;
    f32x16 c0 = {0}, c1 = {0};
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    f32x4 l[4] = {};
    unsigned p = (unsigned)(lane * 16);
    for (int it = 0; it < n; ++it) {
        MFMA(c0, a, b);
#pragma unroll
        for (int i = 0; i < NV; ++i) FMA(x[i % 16], m, kk);
#pragma unroll
        for (int i = 0; i < NL; ++i) LDSR(l[i % 4], p);
        if (CHAINS == 2) {
            MFMA(c1, a, b);
#pragma unroll
            for (int i = 0; i < NV; ++i) FMA(x[i % 16], m, kk);
#pragma unroll
            for (int i = 0; i < NL; ++i) LDSR(l[i % 4], p);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float r = c0[0] + c1[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += l[i][0];
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int NV, int NL, int CHAINS>
void run(const char* name, int threads, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = CHAINS == 2 ? 500 : 1000;         // 1000 MFMAs per wave either way
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NL, CHAINS>), dim3(256), dim3(threads), 0, 0, n, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s waves/SIMD %d : %7.1f us  (%.1f ns per MFMA)\n", name, threads / 256, best * 1e3f, best * 1e6f / 1000.f / (threads / 256));
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    for (int threads : {256, 512}) {
        run<0, 0, 1>("mfma only", threads, out);
        run<4, 0, 1>("mfma + 4 valu", threads, out);
        run<8, 0, 1>("mfma + 8 valu", threads, out);
        run<12, 0, 1>("mfma + 12 valu", threads, out);
        run<16, 0, 1>("mfma + 16 valu", threads, out);
        run<24, 0, 1>("mfma + 24 valu", threads, out);
        run<0, 2, 1>("mfma + 2 ds_read_b128", threads, out);
        run<0, 4, 1>("mfma + 4 ds_read_b128", threads, out);
        run<8, 2, 1>("mfma + 8 valu + 2 lds", threads, out);
        run<8, 0, 2>("2 chains: mfma + 8 valu", threads, out);
        run<0, 0, 2>("2 chains: mfma only", threads, out);
    }
    return 0;
}
