// Micro-benchmark: do fp32 MFMAs of one wave overlap with VALU work of the co-resident wave on the same SIMD?
// Workgroup = 512 threads = 8 waves = 2 per SIMD (one CU), grid = 256 workgroups.  mode bits per wave half:
//   waves 0-3 run `a_kind`, waves 4-7 run `b_kind`:  0 = idle, 1 = dependent MFMA chain, 2 = VALU (4 independent fma chains),
//   3 = LDS reads (ds_read_b128 stream), 4 = MFMA chain with independent VALU interleaved by the compiler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float run_mfma(int n, float a, float b) {
    f32x16 c = {0};
    for (int i = 0; i < n; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    return c[0] + c[5];
}
__device__ __forceinline__ float run_valu(int n, float x) {
    float a = x, b = x + 1.f, c = x + 2.f, d = x + 3.f;
    for (int i = 0; i < n; ++i) {
        a = a * 1.0001f + 0.5f; b = b * 0.9999f + 0.25f; c = c * 1.0002f - 0.5f; d = d * 0.9998f - 0.25f;
    }
    return a + b + c + d;
}
__device__ __forceinline__ float run_lds(int n, const float* lds, int lane) {
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256) & 8191));
        acc += v;
    }
    return acc[0] + acc[1] + acc[2] + acc[3];
}
__device__ __forceinline__ float run_mix(int n, float a, float b) {
    f32x16 c = {0};
    float p = a, q = b, r = a + 1.f, s = b + 1.f;
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { p = p * 1.0001f + 0.5f; q = q * 0.9999f + 0.25f; r = r * 1.0002f - 0.5f; s = s * 0.9998f - 0.25f; }
    }
    return c[0] + p + q + r + s;
}

__global__ __launch_bounds__(512) void k(int a_kind, int b_kind, int n_mfma, int n_valu, int n_lds, float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kind = wave < 4 ? a_kind : b_kind;
    float r = 0.f;
    const float x = (float)lane * 1e-3f;
    if (kind == 1) r = run_mfma(n_mfma, x, x + 1.f);
    else if (kind == 2) r = run_valu(n_valu, x);
    else if (kind == 3) r = run_lds(n_lds, lds, lane);
    else if (kind == 4) r = run_mix(n_mfma, x, x + 1.f);
    if (r == 123.456f) out[threadIdx.x] = r;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_mfma = 1000, n_valu = 4000, n_lds = 4000;   // 1000 MFMAs = 64k cycles; 16000 fma = ? cycles
    const int cfg[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {1, 3}, {2, 3}, {4, 0}, {4, 4}, {4, 2}};
    for (auto& c : cfg) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], n_mfma, n_valu, n_lds, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("A=%d B=%d : %.1f us\n", c[0], c[1], best * 1e3f);
    }
    return 0;
}
