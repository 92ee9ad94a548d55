// Micro-benchmark: the floor for a single-pass kernel over 2^20 (… 2^24) transitions of the GAE scan's shape -- the same
// grid (one 256-thread workgroup per 2,048 transitions), the same loads (v_s, v_next f32, rew f64, two flag bytes: 18 B)
// and stores (adv, returns f32: 8 B) per transition, NO scan and no hand-off.  Timed like bench.py times the scan (HIP
// events around back-to-back launches).  What is left between this and gae_single_pass is the scan itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void empty_kernel(float* out) {
    if (out == nullptr && threadIdx.x == 12345) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void stream_kernel(const float* __restrict__ v, const float* __restrict__ vn,
                                                     const double* __restrict__ rew, const uint8_t* __restrict__ te,
                                                     const uint8_t* __restrict__ tr, float* __restrict__ adv,
                                                     float* __restrict__ ret, int64_t n) {
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (base + 8 > n) return;
    const float4 a0 = reinterpret_cast<const float4*>(v + base)[0], a1 = reinterpret_cast<const float4*>(v + base)[1];
    const float4 b0 = reinterpret_cast<const float4*>(vn + base)[0], b1 = reinterpret_cast<const float4*>(vn + base)[1];
    double2 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = reinterpret_cast<const double2*>(rew + base)[k];
    const uint2 t8 = *reinterpret_cast<const uint2*>(te + base), u8 = *reinterpret_cast<const uint2*>(tr + base);
    const float m = (t8.x | t8.y | u8.x | u8.y) ? 0.f : 1.f;
    float4 o0, o1, p0, p1;
    o0.x = (float)r[0].x + a0.x * m - b0.x; o0.y = (float)r[0].y + a0.y * m - b0.y;
    o0.z = (float)r[1].x + a0.z * m - b0.z; o0.w = (float)r[1].y + a0.w * m - b0.w;
    o1.x = (float)r[2].x + a1.x * m - b1.x; o1.y = (float)r[2].y + a1.y * m - b1.y;
    o1.z = (float)r[3].x + a1.z * m - b1.z; o1.w = (float)r[3].y + a1.w * m - b1.w;
    p0.x = o0.x + a0.x; p0.y = o0.y + a0.y; p0.z = o0.z + a0.z; p0.w = o0.w + a0.w;
    p1.x = o1.x + a1.x; p1.y = o1.y + a1.y; p1.z = o1.z + a1.z; p1.w = o1.w + a1.w;
    reinterpret_cast<float4*>(adv + base)[0] = o0; reinterpret_cast<float4*>(adv + base)[1] = o1;
    reinterpret_cast<float4*>(ret + base)[0] = p0; reinterpret_cast<float4*>(ret + base)[1] = p1;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lg : {20, 22, 24}) {
        const int64_t n = 1ll << lg;
        float *v, *vn, *adv, *ret; double* rew; uint8_t *te, *tr;
        (void)hipMalloc(&v, n * 4); (void)hipMalloc(&vn, n * 4); (void)hipMalloc(&adv, n * 4); (void)hipMalloc(&ret, n * 4);
        (void)hipMalloc(&rew, n * 8); (void)hipMalloc(&te, n); (void)hipMalloc(&tr, n);
        (void)hipMemset(v, 0, n * 4); (void)hipMemset(vn, 0, n * 4); (void)hipMemset(rew, 0, n * 8); (void)hipMemset(te, 0, n); (void)hipMemset(tr, 0, n);
        const unsigned grid = (unsigned)(n / 2048);
        for (int which = 0; which < 2; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                const int iters = 50;
                (void)hipEventRecord(e0);
                for (int i = 0; i < iters; ++i) {
                    if (which == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, adv);
                    else hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, v, vn, rew, te, tr, adv, ret, n);
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms / iters < best) best = ms / iters;
            }
            printf("n = 2^%d  %-34s %7.2f us per launch%s\n", lg, which == 0 ? "empty kernel, same grid" : "loads + stores of the scan, no scan",
                   best * 1e3f, which ? "" : "");
            if (which) printf("           = %.2f TB/s of the 26 B / transition\n", 26.0 * n / (best * 1e-3) / 1e12);
        }
        (void)hipFree(v); (void)hipFree(vn); (void)hipFree(adv); (void)hipFree(ret); (void)hipFree(rew); (void)hipFree(te); (void)hipFree(tr);
    }
    return 0;
}
