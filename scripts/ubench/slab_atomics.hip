// Micro-benchmark: 512 workgroups each contribute an 11,400-float slab.  (a) plain write-through stores into 512 private
// slabs, (b) float atomic adds into 8 slabs chosen by the hardware XCC id, (c) atomic adds into blockIdx % 8 slabs,
// (d) atomic adds into ONE slab.  Checks that the atomically accumulated sums are exact (integer-valued floats).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int W = 11400;
__global__ __launch_bounds__(256) void k(float* slabs, int mode, int* xcc_out) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    if (threadIdx.x == 0) xcc_out[blockIdx.x] = (int)xcc;
    int s = mode == 0 ? blockIdx.x : (mode == 1 ? (int)xcc : (mode == 2 ? (int)(blockIdx.x & 7) : 0));
    float* p = slabs + (size_t)s * W;
    for (int c = threadIdx.x; c < W; c += 256) {
        const float v = (float)((c + blockIdx.x) & 7);
        if (mode == 0) __hip_atomic_store(p + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    float* slabs; int* xcc;
    (void)hipMalloc(&slabs, sizeof(float) * 512 * W); (void)hipMalloc(&xcc, 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"512 private slabs, sc1 stores", "8 slabs by XCC_ID, atomic add", "8 slabs by blockIdx % 8, atomic add",
                           "1 slab, atomic add"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipMemset(slabs, 0, sizeof(float) * 512 * W);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, slabs, mode, xcc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // verify: total over slabs of column c must be sum_b ((c + b) & 7)
        std::vector<float> h((size_t)512 * W);
        (void)hipMemcpy(h.data(), slabs, sizeof(float) * 512 * W, hipMemcpyDeviceToHost);
        const int ns = mode == 0 ? 512 : (mode == 3 ? 1 : 8);
        long bad = 0;
        for (int c = 0; c < W; ++c) {
            double got = 0, want = 0;
            for (int s = 0; s < ns; ++s) got += h[(size_t)s * W + c];
            for (int b = 0; b < 512; ++b) want += (double)((c + b) & 7);
            if (got != want) ++bad;
        }
        std::vector<int> hx(512);
        (void)hipMemcpy(hx.data(), xcc, 512 * 4, hipMemcpyDeviceToHost);
        int hist[8] = {0}, rr = 0;
        for (int b = 0; b < 512; ++b) { hist[hx[b] & 7]++; rr += (hx[b] == (b & 7)); }
        printf("%-40s: %7.1f us   wrong columns %ld   (blocks with xcc == b%%8: %d / 512; per-XCC %d %d %d %d %d %d %d %d)\n", names[mode],
               best * 1e3f, bad, rr, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    }
    return 0;
}
