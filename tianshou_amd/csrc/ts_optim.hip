// ts_optim.hip -- target-network updates (and other flat-vector optimizer helpers) for gfx950.
//
// Replaces polyak_parameter_update / LaggedNetworkCollection.full_parameter_update
// (tianshou/utils/lagged_network.py:8-18, 81-87), which loop over parameter tensors in Python.
// Roofline: HBM, 12 B per parameter (read src, read tgt, write tgt).
#include "ts_common.h"

#pragma clang fp contract(off)   // tau * src + (1 - tau) * tgt: two products, one sum, as torch

namespace {

__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ tgt, const float* __restrict__ src,
                                                     int64_t n, float tau, float one_minus_tau) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 s = *reinterpret_cast<const float4*>(src + i);
            float4 t = *reinterpret_cast<float4*>(tgt + i);
            t.x = tau * s.x + one_minus_tau * t.x;
            t.y = tau * s.y + one_minus_tau * t.y;
            t.z = tau * s.z + one_minus_tau * t.z;
            t.w = tau * s.w + one_minus_tau * t.w;
            *reinterpret_cast<float4*>(tgt + i) = t;
        } else {
            for (int64_t k = i; k < n; ++k) tgt[k] = tau * src[k] + one_minus_tau * tgt[k];
        }
    }
}

}  // namespace

extern "C" {

int ts_polyak_update(float* tgt, const float* src, int64_t n, double tau, ts_stream_t stream) {
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_polyak_update: negative n");
    if (n == 0) return TS_OK;
    TS_REQUIRE(tgt && src, TS_ERR_INVALID_ARG, "ts_polyak_update: NULL argument");
    TS_REQUIRE(((reinterpret_cast<uintptr_t>(tgt) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0,
               TS_ERR_INVALID_ARG, "ts_polyak_update: pointers must be 16-byte aligned");
    hipStream_t s = ts::as_stream(stream);
    if (tau == 1.0) {   // full_parameter_update: plain copy
        TS_HIP_CHECK(hipMemcpyAsync(tgt, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, s));
        return TS_OK;
    }
    int64_t blocks = ts::ceil_div(n, 1024);
    if (blocks > 2048) blocks = 2048;
    // torch evaluates `tau * src + (1 - tau) * tgt` with the Python scalars rounded to float32
    hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tgt, src, n, (float)tau,
                       (float)(1.0 - tau));
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
