// ts_optim.hip -- target-network updates (and other flat-vector optimizer helpers) for gfx950.
//
// Replaces polyak_parameter_update / LaggedNetworkCollection.full_parameter_update
// (tianshou/utils/lagged_network.py:8-18, 81-87), which loop over parameter tensors in Python.
// Roofline: HBM, 12 B per parameter (read src, read tgt, write tgt).
#include <cmath>

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

// ---- clip_grad_norm_ + Adam over one flat parameter vector ------------------------------------
// Optimizer.step (algorithm_base.py:484-500) with torch.optim.Adam's single-tensor arithmetic
// (optim.py:89-110).  28 B / parameter of HBM traffic (+4 for the norm pass when clipping).
constexpr int SUMSQ_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float* p; float* m; float* v; const float* g; int64_t n;
    const float* part; int n_part; float max_norm;
    float lr_step, beta1, beta2, bc2_sqrt, eps;
    float omb1, omb2;    // (1 - beta) rounded from double, as torch passes them
    const float* step_dev;      // {lr_step, bc2_sqrt} on the device instead (a captured launch replayed with a new step number)
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    if (a.step_dev) { a.lr_step = a.step_dev[0]; a.bc2_sqrt = a.step_dev[1]; }
    float scale = 1.f;
    if (a.part) {   // every workgroup re-reduces the partials in the same order -> identical scale
        __shared__ float red[4];
        float s = 0.f;
        for (int k = threadIdx.x; k < a.n_part; k += 256) s += a.part[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        scale = fminf(a.max_norm / (norm + 1e-6f), 1.f);     // clip_grad_norm_
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float gq = a.g[i] * scale;
    float m = a.m[i], v = a.v[i];
    m = m + (gq - m) * a.omb1;                           // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + a.omb2 * gq * gq;                  // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    a.p[i] = a.p[i] + (-a.lr_step * m) / denom;          // addcdiv_(m, denom, -step_size)
    a.m[i] = m;
    a.v[i] = v;
}

// ---- the general step: Adam or RMSprop, optional L2 weight decay -------------------------------
// torch/optim/adam.py `_single_tensor_adam` and torch/optim/rmsprop.py `_single_tensor_rmsprop`, operation by operation
// (the reference builds both through optim.py:89-140; examples/mujoco/mujoco_a2c.py:117 is RMSprop(eps=1e-5, alpha=0.99)).
struct OptimArgs {
    float* p; float* m; float* v; const float* g; int64_t n;
    const float* part; int n_part; float max_norm;
    int kind, centered;
    float wd;
    float lr, lr_step, beta2, bc2_sqrt, eps, omb1, omb2;     // Adam (lr_step = lr / bias_correction1)
    float alpha, oma, momentum;                              // RMSprop (oma = 1 - alpha rounded from double)
};

__global__ __launch_bounds__(256) void optim_kernel(OptimArgs a) {
    float scale = 1.f;
    if (a.part) {
        __shared__ float red[4];
        float s = 0.f;
        for (int k = threadIdx.x; k < a.n_part; k += 256) s += a.part[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        scale = fminf(a.max_norm / (norm + 1e-6f), 1.f);     // clip_grad_norm_
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float par = a.p[i];
    float gq = a.g[i] * scale;
    if (a.wd != 0.f) gq = gq + a.wd * par;                   // grad.add(param, alpha=weight_decay)
    if (a.kind == TS_OPT_RMSPROP) {
        float sq = a.v[i];
        sq = sq * a.alpha + a.oma * gq * gq;                 // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
        float avg;
        if (a.centered) {
            float ga = a.m[i];
            ga = ga + (gq - ga) * a.oma;                     // grad_avg.lerp_(grad, 1 - alpha)
            a.m[i] = ga;
            avg = sqrtf(sq + -1.f * ga * ga);                // square_avg.addcmul(grad_avg, grad_avg, value=-1).sqrt_()
        } else {
            avg = sqrtf(sq);
        }
        avg = avg + a.eps;
        if (a.momentum > 0.f) {
            float buf = a.m[i];
            buf = buf * a.momentum + gq / avg;               // buf.mul_(momentum).addcdiv_(grad, avg)
            a.m[i] = buf;
            a.p[i] = par + -a.lr * buf;                      // param.add_(buf, alpha=-lr)
        } else {
            a.p[i] = par + (-a.lr * gq) / avg;               // param.addcdiv_(grad, avg, value=-lr)
        }
        a.v[i] = sq;
        return;
    }
    float m = a.m[i], v = a.v[i];
    m = m + (gq - m) * a.omb1;
    v = v * a.beta2 + a.omb2 * gq * gq;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    a.p[i] = par + (-a.lr_step * m) / denom;
    a.m[i] = m;
    a.v[i] = v;
}

// The same step for up to three parameter vectors of equal length in one launch (blockIdx.y = vector), optionally
// followed by the Polyak update of each vector's lagged copy from the NEW parameters (lagged_network.py:17-18) --
// per element exactly adam_kernel (without clipping) and polyak_kernel.
struct AdamMultiArgs {
    float* p[3]; float* m[3]; float* v[3]; const float* g[3]; float* tgt[3];
    int64_t n;
    float lr_step, beta1, beta2, bc2_sqrt, eps, omb1, omb2, tau, one_minus_tau;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMultiArgs a) {
    const int k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float gq = a.g[k][i];
    float m = a.m[k][i], v = a.v[k][i];
    m = m + (gq - m) * a.omb1;
    v = v * a.beta2 + a.omb2 * gq * gq;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    const float pn = a.p[k][i] + (-a.lr_step * m) / denom;
    a.p[k][i] = pn;
    a.m[k][i] = m;
    a.v[k][i] = v;
    if (a.tgt[k]) a.tgt[k][i] = a.tau * pn + a.one_minus_tau * a.tgt[k][i];
}

}  // namespace

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
int adam_step_multi(hipStream_t s, int nvec, float* const* params, float* const* m, float* const* v,
                    const float* const* grad, float* const* lagged, int64_t n, int64_t step, double lr, double beta1,
                    double beta2, double eps, double tau) {
    TS_REQUIRE(nvec >= 1 && nvec <= 3, TS_ERR_INVALID_ARG, "adam_step_multi: 1..3 vectors");
    AdamMultiArgs a{};
    for (int k = 0; k < nvec; ++k) {
        a.p[k] = params[k]; a.m[k] = m[k]; a.v[k] = v[k]; a.g[k] = grad[k];
        a.tgt[k] = (lagged && tau > 0.0) ? lagged[k] : nullptr;
    }
    a.n = n;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.lr_step = (float)(lr / bc1);
    a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = (float)eps;
    a.tau = (float)tau; a.one_minus_tau = (float)(1.0 - tau);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)nvec), dim3(256), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int optim_step(hipStream_t s, const OptimDesc& o, float* params, float* m, float* v, const float* grad, int64_t n,
               int64_t step, double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch) {
    if (o.kind == TS_OPT_ADAM && o.weight_decay == 0.0)       // the plain step keeps its own (leaner) kernel
        return adam_step(s, params, m, v, grad, n, step, lr, beta1, beta2, eps, max_grad_norm, norm_scratch);
    TS_REQUIRE(o.kind == TS_OPT_ADAM || o.kind == TS_OPT_RMSPROP, TS_ERR_UNSUPPORTED, "optimizer kind %d", o.kind);
    TS_REQUIRE(!(o.kind == TS_OPT_RMSPROP && o.centered && o.momentum > 0.0), TS_ERR_UNSUPPORTED,
               "RMSprop: centered together with momentum needs two auxiliary vectors (one is provided)");
    OptimArgs a{};
    a.p = params; a.m = m; a.v = v; a.g = grad; a.n = n;
    if (max_grad_norm > 0) {
        hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, grad, n, norm_scratch);
        a.part = norm_scratch; a.n_part = SUMSQ_BLOCKS; a.max_norm = (float)max_grad_norm;
    }
    a.kind = o.kind; a.centered = o.centered; a.wd = (float)o.weight_decay;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.lr = (float)lr; a.lr_step = (float)(lr / bc1);
    a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)sqrt(bc2); a.eps = (float)eps;
    a.alpha = (float)o.alpha; a.oma = (float)(1.0 - o.alpha); a.momentum = (float)o.momentum;
    hipLaunchKernelGGL(optim_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

void adam_step_scalars(int64_t step, double lr, double beta1, double beta2, float out[2]) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    out[0] = (float)(lr / bc1);          // step_size
    out[1] = (float)sqrt(bc2);           // bias_correction2_sqrt
}

static int adam_launch(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, const float sc[2],
                       const float* step_dev, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch) {
    AdamArgs a{};
    a.p = params; a.m = m; a.v = v; a.g = grad; a.n = n;
    if (max_grad_norm > 0) {
        hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, grad, n, norm_scratch);
        a.part = norm_scratch; a.n_part = SUMSQ_BLOCKS; a.max_norm = (float)max_grad_norm;
    }
    a.lr_step = sc[0];
    a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.bc2_sqrt = sc[1];
    a.eps = (float)eps;
    a.step_dev = step_dev;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch) {
    float sc[2];
    adam_step_scalars(step, lr, beta1, beta2, sc);
    return adam_launch(s, params, m, v, grad, n, sc, nullptr, beta1, beta2, eps, max_grad_norm, norm_scratch);
}

int adam_step_dev(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, const float* step_dev,
                  double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch) {
    const float sc[2] = {0.f, 1.f};
    return adam_launch(s, params, m, v, grad, n, sc, step_dev, beta1, beta2, eps, max_grad_norm, norm_scratch);
}
}  // namespace ts

extern "C" {

int ts_adam_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, const float* grad, int64_t n,
                 int64_t step, double lr, double beta1, double beta2, double eps, double max_grad_norm,
                 ts_stream_t stream) {
    TS_REQUIRE(n >= 0 && step >= 1, TS_ERR_INVALID_ARG, "ts_adam_step: bad n / step");
    if (n == 0) return TS_OK;
    TS_REQUIRE(params && adam_m && adam_v && grad, TS_ERR_INVALID_ARG, "ts_adam_step: NULL argument");
    float* scratch = nullptr;
    if (max_grad_norm > 0) {
        TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_adam_step: clipping needs a workspace");
        if (int rc = ts::ws_reserve(ws, 4096)) return rc;
        scratch = static_cast<float*>(ws->base);
    }
    return ts::adam_step(ts::as_stream(stream), params, adam_m, adam_v, grad, n, step, lr, beta1, beta2, eps,
                         max_grad_norm, scratch);
}

int ts_optim_step(ts_workspace* ws, int32_t kind, float* params, float* state_m, float* state_v, const float* grad, int64_t n,
                  int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double rms_alpha,
                  double rms_momentum, int32_t rms_centered, double max_grad_norm, ts_stream_t stream) {
    TS_REQUIRE(n >= 0 && step >= 1, TS_ERR_INVALID_ARG, "ts_optim_step: bad n / step");
    if (n == 0) return TS_OK;
    TS_REQUIRE(params && state_m && state_v && grad, TS_ERR_INVALID_ARG, "ts_optim_step: NULL argument");
    float* scratch = nullptr;
    if (max_grad_norm > 0) {
        TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_optim_step: clipping needs a workspace");
        if (int rc = ts::ws_reserve(ws, 4096)) return rc;
        scratch = static_cast<float*>(ws->base);
    }
    ts::OptimDesc o;
    o.kind = kind; o.centered = rms_centered; o.weight_decay = weight_decay; o.alpha = rms_alpha; o.momentum = rms_momentum;
    return ts::optim_step(ts::as_stream(stream), o, params, state_m, state_v, grad, n, step, lr, beta1, beta2, eps,
                          max_grad_norm, scratch);
}

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
