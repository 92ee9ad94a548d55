// ts_npg.hip -- natural-gradient policy updates (NPG, TRPO) on the MuJoCo actor-critic for gfx950.
//
// Replaces, on device-resident float32 batches:
//   NPG._preprocess_batch's network passes      tianshou/algorithm/modelfree/npg.py:123-138 (V(s), V(s'), log pi_old)
//   NPG._update_with_batch (actor part)         npg.py:149-177: vanilla gradient, Fisher-vector products (_MVP :195-200),
//                                               conjugate gradients (:202-224), step of trust_region_size
//   TRPO._update_with_batch (actor part)        trpo.py:132-191: ratio surrogate, step size sqrt(2 max_kl / s^T F s),
//                                               backtracking line search
//   the critic iterations                       npg.py:179-183 (mse_loss + Optimizer.step, algorithm_base.py:484-500)
// Networks: examples/mujoco/mujoco_npg.py:103-128 = the PPO nets (Net[h, h] tanh, unbounded Gaussian actor with a
// state-independent log-sigma parameter, separate critic).
//
// The reference forms F v by differentiating the mean KL twice (autograd double backward).  At the expansion point the KL's
// first derivatives w.r.t. the distribution parameters vanish exactly, so its Hessian is the Gauss-Newton product
//     F v = (1/B) sum_b J_b^T diag(1 / sigma^2) J_b v   (mean part)   +   2 v_s   (log-sigma part),
// which is what is computed here: one forward-mode pass (J v), one reverse pass (J^T u).  Linear layers run on the fp32-MFMA
// GEMM kernels of ts_conv.hip; this file adds the tanh / Gaussian elementwise kernels, the single-workgroup conjugate-
// gradient update and the orchestration.  The conjugate-gradient early exit (npg.py:217-218) and TRPO's line search are
// decided on the device (no host round trip): every iteration / candidate is evaluated, a flag or a selection kernel keeps
// the reference's result.
//
// Flat layouts: actor  L1 [k0 + 1, h] | L2 [h + 1, h] | head [h + 1, 32] (columns [0, A) = mu) | log_sigma [32]
//               critic L1 | L2 | head [h + 1, 32] (column 0 = V)        (k0 = obs_dim rounded up to 32; last row = bias)
#include <algorithm>

#include "ts_common.h"
#include "ts_conv.h"

#pragma clang fp contract(off)

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
// ts_ppo.hip / ts_npg_q.h: the one-launch gradient / Fisher-vector product / candidate evaluation passes of the 64-64 actor
bool npg_fused_supported(int64_t obs_dim, int64_t hidden, int64_t act_dim);
size_t npg_fused_slab_floats(int64_t obs_dim, int64_t B);
size_t npg_fused_eval_floats(int64_t B, int n_cand);
int npg_fvp_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* v, const float* x, int obs, int k0, int act,
                  int64_t B, float damping, float* slabs, float* out);
int npg_grad_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* x, const float* actions, const float* adv,
                   const float* logp_old, int obs, int k0, int act, int64_t B, float* slabs, float* grad, float* loss_out,
                   float* mu);
int npg_infer_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* x, const float* actions, int obs, int k0,
                    int act, int64_t B, float* head_out, int head_stride, float* logp_out);
int npg_critic_grad_fused(hipStream_t s, ts_workspace* ws, const float* critic, const float* x, const float* returns, int obs,
                          int k0, int64_t B, float* slabs, float* grad, float* loss_out);
int npg_eval_fused(hipStream_t s, ts_workspace* ws, const float* theta_old, const float* cands, int64_t cand_stride, int n_cand,
                   const float* x, const float* actions, const float* adv, const float* logp_old, const float* mu, int obs, int k0,
                   int act, int64_t B, float* partial, float* res, float* apply_theta = nullptr, float* apply_stats3 = nullptr);
}

namespace {

constexpr int HEAD = 32;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

struct Net3 {
    ts::ConvGeom l[3];
    int64_t off[4];               // off[3] = end of the head block
    int obs, hid, k0;
};

int make_net3(int B, int64_t obs_dim, int64_t hidden, Net3* n) {
    TS_REQUIRE(obs_dim >= 1 && obs_dim <= 65536 && hidden >= 32 && hidden <= 1024 && hidden % 32 == 0, TS_ERR_INVALID_ARG,
               "npg: obs_dim >= 1, hidden a multiple of 32 in [32, 1024]");
    n->obs = (int)obs_dim; n->hid = (int)hidden; n->k0 = (n->obs + 31) / 32 * 32;
    const int dims[4] = {n->k0, n->hid, n->hid, HEAD};
    int64_t o = 0;
    for (int i = 0; i < 3; ++i) {
        n->l[i] = ts::ConvGeom{B, 1, 1, dims[i], 1, 1, 1, 1, 1, dims[i + 1]};
        n->off[i] = o;
        o += n->l[i].param_elems();
    }
    n->off[3] = o;
    return TS_OK;
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Carve {
    char* p;
    float* f(size_t n) { float* r = reinterpret_cast<float*>(p); p += al(4 * n); return r; }
};

struct Act3 { float* h1; float* h2; float* out; };

Act3 take_act(Carve& c, const Net3& n, int64_t B) { return Act3{c.f(B * n.hid), c.f(B * n.hid), c.f(B * HEAD)}; }

size_t act_bytes(const Net3& n, int64_t B) { return 2 * al(4 * B * n.hid) + al(4 * B * HEAD); }

size_t split_floats(const Net3& n) {
    size_t s = 4;
    for (int i = 0; i < 3; ++i) { const int ns = ts::conv_fwd_splits(n.l[i]); if (ns > 1) s = std::max(s, (size_t)ns * n.l[i].out_elems()); }
    return s;
}

size_t slab_floats(const Net3& n) {
    size_t s = 0;
    for (int i = 0; i < 3; ++i) s = std::max(s, (size_t)ts::conv_wgrad_splits(n.l[i]) * n.l[i].param_elems());
    return s;
}

// ---- elementwise kernels ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ obs, int64_t B, int obs_dim, int k0,
                                                       float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * k0) return;
    const int64_t b = i / k0;
    const int j = (int)(i - b * k0);
    x[i] = j < obs_dim ? obs[b * obs_dim + j] : 0.f;
}

__global__ __launch_bounds__(256) void tanh_kernel(float* __restrict__ h, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = tanhf(h[i]);
}

// dh *= 1 - h^2   (backward through tanh)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dh[i] = dh[i] * (1.f - h[i] * h[i]);
}

// forward-mode combine: t = (ta - bias_w[col]) + tb, where ta = dh_prev . W + b_w (the GEMM adds the layer's own bias) and
// tb = h_prev . V + b_v; out = t * (1 - h^2) for a tanh layer (h != NULL), t for the head
__global__ __launch_bounds__(256) void jvp_combine_kernel(const float* __restrict__ ta, const float* __restrict__ tb,
                                                          const float* __restrict__ bias_w, const float* __restrict__ h,
                                                          int64_t n, int cols, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = (ta[i] - bias_w[i % cols]) + tb[i];
    out[i] = h ? t * (1.f - h[i] * h[i]) : t;
}

// Independent(Normal(mu, exp(s)), 1).log_prob(act) for one sample (torch/distributions/normal.py)
// mu_bound > 0: ContinuousActorProbabilistic(unbounded=False), the constructor default (utils/net/continuous.py:230-231):
// the distribution's mean is max_action * tanh(head output)
__device__ __forceinline__ float bounded_mu(float raw, float mu_bound) { return mu_bound > 0.f ? mu_bound * tanhf(raw) : raw; }

__device__ __forceinline__ float gauss_logp(const float* mu, const float* act, const float* log_sigma, int A, float mu_bound = 0.f) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
        const float sigma = expf(log_sigma[j]), var = sigma * sigma, d = act[j] - bounded_mu(mu[j], mu_bound);
        lp += -(d * d) / (2.f * var) - logf(sigma) - LOG_SQRT_2PI;
    }
    return lp;
}

__global__ __launch_bounds__(256) void infer_out_kernel(const float* __restrict__ mu_head, const float* __restrict__ v_head,
                                                        const float* __restrict__ act, const float* __restrict__ log_sigma,
                                                        int64_t B, int A, float* __restrict__ v_out, float* __restrict__ logp_out,
                                                        float* __restrict__ mu_out, float mu_bound = 0.f) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (v_out) v_out[b] = v_head[b * HEAD];
    if (logp_out) logp_out[b] = gauss_logp(mu_head + b * HEAD, act + b * A, log_sigma, A, mu_bound);
    if (mu_out) for (int j = 0; j < A; ++j) mu_out[b * A + j] = bounded_mu(mu_head[b * HEAD + j], mu_bound);
}

// block-wide deterministic sum (256 threads): all threads get the result
__device__ float block_sum_256(float v, float* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

// surrogate loss and its gradient w.r.t. the head outputs and log_sigma (npg.py:152-155 / trpo.py:135-138):
//   NPG  (ratio_mode 0): loss = -mean(logp * adv)             d logp_b = -adv_b / B
//   TRPO (ratio_mode 1): loss = -mean(exp(logp - logp_old) adv)   d logp_b = -adv_b ratio_b / B
// per block: partial[blk * (1 + A) + 0] = sum of loss terms, [1 + j] = sum_b d logp_b ((a - mu)^2 / var - 1)
__global__ __launch_bounds__(256) void actor_loss_kernel(const float* __restrict__ mu_head, const float* __restrict__ act,
                                                         const float* __restrict__ adv, const float* __restrict__ logp_old,
                                                         const float* __restrict__ log_sigma, int ratio_mode, int64_t B, int A,
                                                         float* __restrict__ d_head, float* __restrict__ partial) {
    __shared__ float red[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float inv_b = 1.f / (float)B;
    float term = 0.f, dlogp = 0.f;
    if (b < B) {
        const float lp = gauss_logp(mu_head + b * HEAD, act + b * A, log_sigma, A);
        if (ratio_mode) {
            const float ratio = expf(lp - logp_old[b]);
            term = ratio * adv[b];
            dlogp = -adv[b] * ratio * inv_b;
        } else {
            term = lp * adv[b];
            dlogp = -adv[b] * inv_b;
        }
    }
    const float tot = block_sum_256(term, red);
    if (threadIdx.x == 0) partial[blockIdx.x * (1 + A)] = tot;
    for (int j = 0; j < HEAD; ++j) {
        float ds = 0.f, dm = 0.f;
        if (b < B && j < A) {
            const float sigma = expf(log_sigma[j]), var = sigma * sigma, d = act[b * A + j] - mu_head[b * HEAD + j];
            dm = dlogp * d / var;
            ds = dlogp * (d * d / var - 1.f);
        }
        if (b < B) d_head[b * HEAD + j] = dm;
        if (j < A) {
            const float t = block_sum_256(ds, red);
            if (threadIdx.x == 0) partial[blockIdx.x * (1 + A) + 1 + j] = t;
        }
    }
}

// loss = -sum(partial[., 0]) / B; grad[sigma_off + j] = sum(partial[., 1 + j]) (j < A), 0 for the padding entries
__global__ __launch_bounds__(256) void actor_loss_finish_kernel(const float* __restrict__ partial, int n_blocks, int64_t B, int A,
                                                                float* __restrict__ loss, float* __restrict__ g_sigma) {
    __shared__ float red[256];
    for (int k = 0; k <= A; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += 256) s += partial[i * (1 + A) + k];
        const float t = block_sum_256(s, red);
        if (threadIdx.x == 0) {
            if (k == 0) *loss = -(t / (float)B);
            else g_sigma[k - 1] = t;
        }
    }
    if ((int)threadIdx.x >= A && threadIdx.x < HEAD) g_sigma[threadIdx.x] = 0.f;
}

// Fisher upstream: u[b, j] = dmu[b, j] / (sigma_j^2 B) for j < A, 0 for the padding columns
__global__ __launch_bounds__(256) void fisher_upstream_kernel(const float* __restrict__ dmu, const float* __restrict__ log_sigma,
                                                              int64_t B, int A, float* __restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * HEAD) return;
    const int j = (int)(i % HEAD);
    float v = 0.f;
    if (j < A) { const float sigma = expf(log_sigma[j]); v = dmu[i] / (sigma * sigma) / (float)B; }
    u[i] = v;
}

// z += damping * v; the log-sigma block additionally gets the exact 2 v_s of the KL's Hessian
__global__ __launch_bounds__(256) void fvp_finish_kernel(float* __restrict__ z, const float* __restrict__ v, int64_t P,
                                                         int64_t sigma_off, int A, float damping) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float r = z[i] + v[i] * damping;
    if (i >= sigma_off) r = (i - sigma_off < A ? 2.f * v[i] : 0.f) + v[i] * damping;
    z[i] = r;
}

// One conjugate-gradient iteration (npg.py:213-223) in a single workgroup.  sc = {rdotr, done flag, p.z of the last call}.
// Once new_rdotr < tol the reference leaves the loop: later calls are no-ops.
__global__ __launch_bounds__(1024) void cg_update_kernel(float* __restrict__ x, float* __restrict__ r, float* __restrict__ p,
                                                         const float* __restrict__ z, int64_t P, float tol,
                                                         float* __restrict__ sc) {
    __shared__ float red[2][16];
    if (sc[1] != 0.f) return;
    // workgroup sum in a fixed order: butterfly inside each wavefront, then the sixteen wave sums in order -- two barriers
    // per sum (the kernel is ten iterations' worth of pure latency per minibatch; a 10-level LDS tree cost 22 barriers a sum)
    auto block_sum = [&](float v, int slot) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[slot][threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[slot][w];
        return t;
    };
    float pz = 0.f;
    for (int64_t i = threadIdx.x; i < P; i += 1024) pz += p[i] * z[i];
    pz = block_sum(pz, 0);
    const float rdotr = sc[0];
    const float alpha = rdotr / pz;
    float nr = 0.f;
    for (int64_t i = threadIdx.x; i < P; i += 1024) {
        x[i] += alpha * p[i];
        const float ri = r[i] - alpha * z[i];
        r[i] = ri;
        nr += ri * ri;
    }
    const float new_rdotr = block_sum(nr, 1);
    if (new_rdotr < tol) {
        if (threadIdx.x == 0) sc[1] = 1.f;
        return;
    }
    const float beta = new_rdotr / rdotr;
    for (int64_t i = threadIdx.x; i < P; i += 1024) p[i] = r[i] + beta * p[i];
    if (threadIdx.x == 0) { sc[0] = new_rdotr; sc[2] = pz; }
}

// r = p = g, x = 0, sc = {g.g, 0, 0}
__global__ __launch_bounds__(1024) void cg_init_kernel(const float* __restrict__ g, float* __restrict__ x, float* __restrict__ r,
                                                       float* __restrict__ p, int64_t P, float* __restrict__ sc) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < P; i += 1024) { const float v = g[i]; x[i] = 0.f; r[i] = v; p[i] = v; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { sc[0] = red[0]; sc[1] = 0.f; sc[2] = 0.f; }
}

// out[0] = sqrt(2 max_kl / (s . Fs)) with s = -x (trpo.py:153-160): the sign cancels in the product
__global__ __launch_bounds__(1024) void trpo_step_size_kernel(const float* __restrict__ x, const float* __restrict__ fx, int64_t P,
                                                              float max_kl, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < P; i += 1024) s += x[i] * fx[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sqrtf(2.f * max_kl / red[0]);
}

// cand = theta + scale * (-x); scale = fixed (NPG) or step[0] * coeff^k (TRPO candidate k)
__global__ __launch_bounds__(256) void candidate_kernel(const float* __restrict__ theta, const float* __restrict__ x, int64_t P,
                                                        const float* __restrict__ step, float fixed_scale, float coeff_pow,
                                                        float* __restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float scale = step ? step[0] * coeff_pow : fixed_scale;
    cand[i] = theta[i] + scale * (-x[i]);
}

// every backtracking candidate of TRPO's line search in one launch (blockIdx.y = k): candidate_kernel's arithmetic with
// coeff^k formed by the host loop's k multiplications
__global__ __launch_bounds__(256) void candidates_all_kernel(const float* __restrict__ theta, const float* __restrict__ x, int64_t P,
                                                             const float* __restrict__ step, float coeff, float* __restrict__ cands) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float cpow = 1.f;
    for (unsigned k = 0; k < blockIdx.y; ++k) cpow = cpow * coeff;
    const float scale = step[0] * cpow;
    cands[(int64_t)blockIdx.y * P + i] = theta[i] + scale * (-x[i]);
}

// per-block partial sums of kl(old || new) (torch/distributions/kl.py _kl_normal_normal, summed over the action dims) and of
// the TRPO surrogate term ratio * adv under the candidate parameters
__global__ __launch_bounds__(256) void kl_eval_kernel(const float* __restrict__ mu_old, const float* __restrict__ ls_old,
                                                      const float* __restrict__ mu_new, const float* __restrict__ ls_new,
                                                      const float* __restrict__ act, const float* __restrict__ adv,
                                                      const float* __restrict__ logp_old, int64_t B, int A,
                                                      float* __restrict__ partial) {
    __shared__ float red[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float kl = 0.f, term = 0.f;
    if (b < B) {
        for (int j = 0; j < A; ++j) {
            const float so = expf(ls_old[j]), sn = expf(ls_new[j]);
            const float q = so / sn, var_ratio = q * q;
            const float d = (mu_old[b * HEAD + j] - mu_new[b * HEAD + j]) / sn, t1 = d * d;
            kl += 0.5f * (var_ratio + t1 - 1.f - logf(var_ratio));
        }
        if (logp_old) term = expf(gauss_logp(mu_new + b * HEAD, act + b * A, ls_new, A) - logp_old[b]) * adv[b];
    }
    const float k = block_sum_256(kl, red);
    const float t = block_sum_256(term, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = k; partial[blockIdx.x * 2 + 1] = t; }
}

// res[cand * 2 + {0, 1}] = {mean kl, -mean(ratio adv)}
__global__ __launch_bounds__(256) void kl_finish_kernel(const float* __restrict__ partial, int n_blocks, int64_t B,
                                                        float* __restrict__ res) {
    __shared__ float red[256];
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += 256) s += partial[i * 2 + k];
        const float t = block_sum_256(s, red);
        if (threadIdx.x == 0) res[k] = k == 0 ? t / (float)B : -(t / (float)B);
    }
}

// TRPO's line search outcome (trpo.py:167-191) from the evaluated candidates: the first k with kl_k < max_kl and
// loss_k < loss_0; none: parameters restored, step size 0, kl of the last candidate.  stats = {loss_0 (kept), kl, step_size}
__global__ __launch_bounds__(256) void trpo_select_kernel(float* __restrict__ theta, const float* __restrict__ cands, int64_t P,
                                                          const float* __restrict__ res, int n_cand, float max_kl, float coeff,
                                                          const float* __restrict__ step, float* __restrict__ stats) {
    __shared__ int s_pick;
    if (threadIdx.x == 0) {
        int pick = -1;
        for (int k = 0; k < n_cand && pick < 0; ++k)
            if (res[2 * k] < max_kl && res[2 * k + 1] < stats[0]) pick = k;
        s_pick = pick;
    }
    __syncthreads();
    const int pick = s_pick;
    if (pick >= 0)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) theta[i] = cands[(int64_t)pick * P + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float st = step[0];
        for (int k = 0; k < pick; ++k) st = st * coeff;
        stats[1] = res[2 * (pick >= 0 ? pick : n_cand - 1)];
        stats[2] = pick >= 0 ? st : 0.f;
    }
}

// per-workgroup sum of one value per thread (256 threads), fixed order: wave shuffle trees, then the four wave sums in order
__device__ __forceinline__ float block_sum256(float v, float* red4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// out = scale * sum_i partial[i] in a fixed order (thread t: i = t, t + 256, ...; then block_sum256)
__global__ __launch_bounds__(256) void sum_finish_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    __shared__ float red4[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const float t = block_sum256(v, red4);
    if (threadIdx.x == 0) *out = t * scale;
}

// critic: vf_loss = mse_loss(returns, V); d_head[b, 0] = 2 (V - returns) / B.  One sample per thread, per-workgroup partial
// sums of the loss (sum_finish_kernel adds them up): as ONE workgroup over 65,536 samples -- which also writes their 8 MB of
// d_head rows -- this kernel was 412 us, a third of an NPG update.
__global__ __launch_bounds__(256) void critic_loss_kernel(const float* __restrict__ v_head, const float* __restrict__ ret, int64_t B,
                                                          float* __restrict__ d_head, float* __restrict__ partial) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    __shared__ float red4[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float inv_b = 1.f / (float)B;
    float ls = 0.f;
    if (b < B) {
        const float t = v_head[b * HEAD] - ret[b];
        ls = t * t;
        f32x4* row = reinterpret_cast<f32x4*>(d_head + b * HEAD);
        row[0] = f32x4{2.f * t * inv_b, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 1; j < HEAD / 4; ++j) row[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float tot = block_sum256(ls, red4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// ---- PPO / A2C on the GEMM path (any Net[h, h] tanh actor-critic: the shapes the fused kernels of ts_ppo.hip do not
// cover, e.g. Humanoid's obs 376 / act 17 / hidden 256) -----------------------------------------------------------------
// PPO._update_with_batch (ppo.py:181-211) per sample, same branch / tie semantics as ts_ppo.hip's net_fwd_bwd:
// term_b and d term_b / d logp_b; d_head = d loss / d mu, per-block partial sums of the loss terms and of d loss / d sigma.
struct WideLossP { float eps_clip, dual_clip, ent_coef, inv_b; int a2c, adv_norm; float mu_bound; };

__global__ __launch_bounds__(256) void ppo_wide_actor_loss_kernel(const float* __restrict__ mu_head, const float* __restrict__ act,
                                                                  const float* __restrict__ adv, const float* __restrict__ logp_old,
                                                                  const float* __restrict__ log_sigma,
                                                                  const float* __restrict__ adv_stats, WideLossP hp, int64_t B, int A,
                                                                  float* __restrict__ d_head, float* __restrict__ partial) {
    __shared__ float red[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float term = 0.f, dlogp = 0.f;
    if (b < B) {
        const float lp = gauss_logp(mu_head + b * HEAD, act + b * A, log_sigma, A, hp.mu_bound);
        float Ab = adv[b];
        if (hp.adv_norm) Ab = (Ab - adv_stats[0]) / (adv_stats[1] + 1e-8f);            // ppo.py:184-186
        if (hp.a2c) {                                                                  // a2c.py:266-267
            term = -lp * Ab;
            dlogp = -Ab * hp.inv_b;
        } else {
            const float ratio = expf(lp - logp_old[b]);                                // :187
            const float surr1 = ratio * Ab;
            const float surr2 = fminf(fmaxf(ratio, 1.f - hp.eps_clip), 1.f + hp.eps_clip) * Ab;   // :190
            const float clip1 = fminf(surr1, surr2);
            float basek = (surr1 <= surr2) ? Ab : 0.f;          // torch.min backward (ties: both branches equal)
            if (hp.dual_clip > 0.f && Ab < 0.f) {                                      // :191-194
                const float dA = hp.dual_clip * Ab;
                term = -fmaxf(clip1, dA);
                if (!(clip1 >= dA)) basek = 0.f;
            } else {
                term = -clip1;                                                         // :196
            }
            dlogp = -basek * ratio * hp.inv_b;
        }
    }
    // actor_loss_finish_kernel reports -sum / B: park the negated term so that it returns mean(term)
    const float tot = block_sum_256(-term, red);
    if (threadIdx.x == 0) partial[blockIdx.x * (1 + A)] = tot;
    for (int j = 0; j < HEAD; ++j) {
        float ds = 0.f, dm = 0.f;
        if (b < B && j < A) {
            const float raw = mu_head[b * HEAD + j], tb = hp.mu_bound > 0.f ? tanhf(raw) : 0.f;
            const float sigma = expf(log_sigma[j]), var = sigma * sigma;
            const float d = act[b * A + j] - (hp.mu_bound > 0.f ? hp.mu_bound * tb : raw);
            dm = dlogp * d / var;
            if (hp.mu_bound > 0.f) dm = (dm * hp.mu_bound) * (1.f - tb * tb);          // MulBackward, then TanhBackward
            ds = dlogp * (d * d / var - 1.f) - hp.ent_coef * hp.inv_b;                 // entropy: d / d log_sigma = 1
        }
        if (b < B) d_head[b * HEAD + j] = dm;
        if (j < A) {
            const float t = block_sum_256(ds, red);
            if (threadIdx.x == 0) partial[blockIdx.x * (1 + A) + 1 + j] = t;
        }
    }
}

// value loss (ppo.py:198-208, a2c.py:270) and its gradient w.r.t. the value head, scaled by vf_coef; one sample per thread,
// per-workgroup partial sums (sum_finish_kernel)
__global__ __launch_bounds__(256) void ppo_wide_critic_loss_kernel(const float* __restrict__ v_head, const float* __restrict__ ret,
                                                                   const float* __restrict__ v_old, float eps_clip, int value_clip,
                                                                   float vf_coef, int64_t B, float inv_b,
                                                                   float* __restrict__ d_head, float* __restrict__ partial) {
    // inv_b = 1 / (global minibatch size): the same scale as the actor part (data-parallel shards sum to the batch mean)
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    __shared__ float red4[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (b < B) {
        const float value = v_head[b * HEAD], r = ret[b];
        const float vf1 = (r - value) * (r - value);
        const float g1 = -2.f * (r - value);
        float dv = g1;
        term = vf1;
        if (value_clip) {
            const float vo = v_old[b], dvo = value - vo;
            const float vclip = vo + fminf(fmaxf(dvo, -eps_clip), eps_clip);
            const float vf2 = (r - vclip) * (r - vclip);
            const float g2 = (dvo >= -eps_clip && dvo <= eps_clip) ? -2.f * (r - vclip) : 0.f;
            term = fmaxf(vf1, vf2);                                                    // torch.max: ties split the gradient
            dv = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        }
        f32x4* row = reinterpret_cast<f32x4*>(d_head + b * HEAD);
        row[0] = f32x4{dv * vf_coef * inv_b, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 1; j < HEAD / 4; ++j) row[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float tot = block_sum256(term, red4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// losses[3] = Normal.entropy().sum(-1) (identical for every sample), losses[0] = clip + vf_coef vf - ent_coef ent (ppo.py:211)
__global__ void ppo_wide_total_kernel(float* __restrict__ losses, const float* __restrict__ log_sigma, int A, float vf_coef,
                                      float ent_coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ent = 0.f;
    for (int k = 0; k < A; ++k) ent += 0.5f + 0.5f * 1.8378770664093453f + logf(expf(log_sigma[k]));
    losses[3] = ent;
    losses[0] = losses[1] + vf_coef * losses[2] - ent_coef * ent;
}

// ---- conditioned sigma (ContinuousActorProbabilistic(conditioned_sigma=True), utils/net/continuous.py:212-234): the head's
// columns [CS_COL, CS_COL + A) hold Linear(h) = the un-clamped log sigma of every sample; sigma = exp(clamp(., -20, 2)).
constexpr int CS_COL = 16;
constexpr float CS_MIN = -20.f, CS_MAX = 2.f;

__device__ __forceinline__ float gauss_logp_cs(const float* head, const float* act, int A, float mu_bound = 0.f) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
        const float ls = fminf(fmaxf(head[CS_COL + j], CS_MIN), CS_MAX);
        const float sigma = expf(ls), var = sigma * sigma, d = act[j] - bounded_mu(head[j], mu_bound);
        lp += -(d * d) / (2.f * var) - logf(sigma) - LOG_SQRT_2PI;
    }
    return lp;
}

__global__ __launch_bounds__(256) void infer_out_cs_kernel(const float* __restrict__ mu_head, const float* __restrict__ v_head,
                                                           const float* __restrict__ act, int64_t B, int A,
                                                           float* __restrict__ v_out, float* __restrict__ logp_out,
                                                           float* __restrict__ mu_out, float mu_bound = 0.f) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (v_out) v_out[b] = v_head[b * HEAD];
    if (logp_out) logp_out[b] = gauss_logp_cs(mu_head + b * HEAD, act + b * A, A, mu_bound);
    if (mu_out) for (int j = 0; j < A; ++j) mu_out[b * A + j] = bounded_mu(mu_head[b * HEAD + j], mu_bound);
}

// The clipped-surrogate / A2C actor loss of ppo_wide_actor_loss_kernel with a per-sample sigma: the gradient w.r.t. log sigma
// goes back through the clamp (torch: passes where min <= x <= max) into the head's sigma columns instead of into a
// parameter; the entropy is a per-sample quantity (partial[., 1] = its per-block sums).  partial: [n_blocks][2].
__global__ __launch_bounds__(256) void ppo_net_actor_loss_cs_kernel(const float* __restrict__ head, const float* __restrict__ act,
                                                                    const float* __restrict__ adv, const float* __restrict__ logp_old,
                                                                    const float* __restrict__ adv_stats, WideLossP hp, int64_t B, int A,
                                                                    float* __restrict__ d_head, float* __restrict__ partial) {
    __shared__ float red[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float term = 0.f, dlogp = 0.f, ent = 0.f;
    if (b < B) {
        const float* hrow = head + b * HEAD;
        const float lp = gauss_logp_cs(hrow, act + b * A, A, hp.mu_bound);
        float Ab = adv[b];
        if (hp.adv_norm) Ab = (Ab - adv_stats[0]) / (adv_stats[1] + 1e-8f);
        if (hp.a2c) {
            term = -lp * Ab;
            dlogp = -Ab * hp.inv_b;
        } else {
            const float ratio = expf(lp - logp_old[b]);
            const float surr1 = ratio * Ab;
            const float surr2 = fminf(fmaxf(ratio, 1.f - hp.eps_clip), 1.f + hp.eps_clip) * Ab;
            const float clip1 = fminf(surr1, surr2);
            float basek = (surr1 <= surr2) ? Ab : 0.f;
            if (hp.dual_clip > 0.f && Ab < 0.f) {
                const float dA = hp.dual_clip * Ab;
                term = -fmaxf(clip1, dA);
                if (!(clip1 >= dA)) basek = 0.f;
            } else {
                term = -clip1;
            }
            dlogp = -basek * ratio * hp.inv_b;
        }
        float* drow = d_head + b * HEAD;
        for (int j = 0; j < HEAD; ++j) drow[j] = 0.f;
        for (int j = 0; j < A; ++j) {
            const float raw = hrow[CS_COL + j];
            const float ls = fminf(fmaxf(raw, CS_MIN), CS_MAX);
            const float tb = hp.mu_bound > 0.f ? tanhf(hrow[j]) : 0.f;
            const float sigma = expf(ls), var = sigma * sigma, d = act[b * A + j] - (hp.mu_bound > 0.f ? hp.mu_bound * tb : hrow[j]);
            ent += 0.5f + 0.5f * 1.8378770664093453f + logf(sigma);                     // Normal.entropy()
            drow[j] = dlogp * d / var;
            if (hp.mu_bound > 0.f) drow[j] = (drow[j] * hp.mu_bound) * (1.f - tb * tb);
            const float dls = dlogp * (d * d / var - 1.f) - hp.ent_coef * hp.inv_b;
            drow[CS_COL + j] = (raw >= CS_MIN && raw <= CS_MAX) ? dls : 0.f;
        }
    }
    const float tot = block_sum_256(-term, red);
    const float et = block_sum_256(ent, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = tot; partial[2 * blockIdx.x + 1] = et; }
}

// loss[1] (clip / actor loss) = -sum(partial[., 0]) / B, loss[3] (entropy) = sum(partial[., 1]) / B; the log_sigma block of the
// gradient (unused parameters) is zero
__global__ __launch_bounds__(256) void ppo_net_cs_finish_kernel(const float* __restrict__ partial, int n_blocks, int64_t B,
                                                                float* __restrict__ losses4, float* __restrict__ g_sigma) {
    __shared__ float red[256];
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += 256) s += partial[2 * i + k];
        const float t = block_sum_256(s, red);
        if (threadIdx.x == 0) {
            if (k == 0) losses4[1] = -(t / (float)B);
            else losses4[3] = t / (float)B;
        }
    }
    if (threadIdx.x < HEAD) g_sigma[threadIdx.x] = 0.f;
}

// losses[0] = clip + vf_coef vf - ent_coef ent with the entropy already in losses[3]
__global__ void ppo_net_total_cs_kernel(float* __restrict__ losses, float vf_coef, float ent_coef) {
    if (threadIdx.x == 0 && blockIdx.x == 0) losses[0] = losses[1] + vf_coef * losses[2] - ent_coef * losses[3];
}

// ---- network passes --------------------------------------------------------------------------------------------------
int forward(hipStream_t s, ts_workspace* ws, const Net3& n, const float* p, const float* x, const Act3& a, float* split, int64_t B) {
    const unsigned gh = (unsigned)ts::ceil_div(B * n.hid, 256);
    if (int rc = ts::conv_forward(s, n.l[0], x, p + n.off[0], a.h1, false, split, ws)) return rc;
    hipLaunchKernelGGL(tanh_kernel, dim3(gh), dim3(256), 0, s, a.h1, B * n.hid);
    if (int rc = ts::conv_forward(s, n.l[1], a.h1, p + n.off[1], a.h2, false, split, ws)) return rc;
    hipLaunchKernelGGL(tanh_kernel, dim3(gh), dim3(256), 0, s, a.h2, B * n.hid);
    TS_LAUNCH_CHECK();
    return ts::conv_forward(s, n.l[2], a.h2, p + n.off[2], a.out, false, split, ws);
}

struct Bwd { float* dh2; float* dh1; float* slabs; };

// grad[0 .. off[3]) = J^T d_out
int backward(hipStream_t s, ts_workspace* ws, const Net3& n, const float* p, const float* x, const Act3& a, const float* d_out,
             float* grad, const Bwd& sc, int64_t B) {
    const float* xin[3] = {x, a.h1, a.h2};
    const float* dy[3] = {sc.dh1, sc.dh2, d_out};
    float* dxl[3] = {nullptr, sc.dh1, sc.dh2};
    const unsigned gh = (unsigned)ts::ceil_div(B * n.hid, 256);
    for (int i = 2; i >= 0; --i) {
        if (int rc = ts::conv_wgrad(s, n.l[i], xin[i], dy[i], sc.slabs, ws)) return rc;
        if (int rc = ts::slab_sum(s, sc.slabs, ts::conv_wgrad_splits(n.l[i]), n.l[i].param_elems(), grad + n.off[i])) return rc;
        if (i > 0) {
            if (int rc = ts::conv_dgrad(s, n.l[i], dy[i], p + n.off[i], nullptr, dxl[i], ws)) return rc;
            hipLaunchKernelGGL(tanh_bwd_kernel, dim3(gh), dim3(256), 0, s, dxl[i], xin[i], B * n.hid);
            TS_LAUNCH_CHECK();
        }
    }
    return TS_OK;
}

struct Jvp { float* ta; float* tb; float* d1; float* d2; float* dmu; };

// dmu = J v for the direction v (same layout as the parameters), given the activations of the forward pass
int jvp(hipStream_t s, ts_workspace* ws, const Net3& n, const float* p, const float* v, const float* x, const Act3& a,
        const Jvp& j, float* split, int64_t B) {
    const int64_t nh = B * n.hid, no = B * HEAD;
    const unsigned gh = (unsigned)ts::ceil_div(nh, 256), go = (unsigned)ts::ceil_div(no, 256);
    // layer 1: t1 = x V1 + b1v
    if (int rc = ts::conv_forward(s, n.l[0], x, v + n.off[0], j.d1, false, split, ws)) return rc;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(gh), dim3(256), 0, s, j.d1, a.h1, nh);          // d1 = t1 (1 - h1^2)
    // layer 2: t2 = d1 W2 + h1 V2 + b2v
    if (int rc = ts::conv_forward(s, n.l[1], j.d1, p + n.off[1], j.ta, false, split, ws)) return rc;
    if (int rc = ts::conv_forward(s, n.l[1], a.h1, v + n.off[1], j.tb, false, split, ws)) return rc;
    hipLaunchKernelGGL(jvp_combine_kernel, dim3(gh), dim3(256), 0, s, j.ta, j.tb, p + n.off[1] + (int64_t)n.hid * n.hid, a.h2, nh,
                       n.hid, j.d2);
    // head: dmu = d2 W3 + h2 V3 + b3v
    if (int rc = ts::conv_forward(s, n.l[2], j.d2, p + n.off[2], j.ta, false, split, ws)) return rc;
    if (int rc = ts::conv_forward(s, n.l[2], a.h2, v + n.off[2], j.tb, false, split, ws)) return rc;
    hipLaunchKernelGGL(jvp_combine_kernel, dim3(go), dim3(256), 0, s, j.ta, j.tb, p + n.off[2] + (int64_t)n.hid * HEAD,
                       (const float*)nullptr, no, HEAD, j.dmu);
    TS_LAUNCH_CHECK();
    return TS_OK;
}


// ---- trunks of any depth / width / activation (ts_net_desc): the PPO / A2C step and the inference passes of ppo_wide on the
// same layer kernels.  Net / MLP: Sequential(Linear, act, Linear, act, ...) (utils/net/common.py:90-178), linear heads.
constexpr int MAXL = TS_NET_MAX_HIDDEN + 1;          // linear layers incl. the head

struct NetL {
    ts::ConvGeom l[MAXL];
    int64_t off[MAXL + 1];        // off[L] = end of the head block
    int64_t ln_off[MAXL];         // layer norm: offset of hidden layer i's gamma[width] | beta[width] (behind its wb block)
    int L, obs, k0, act_fn, wmax, csigma, ln;
    float ln_eps;
    int width[MAXL + 1];          // width[0] = k0, width[i] = padded width of hidden layer i - 1, width[L] = HEAD
    int tw[MAXL + 1];             // the same widths as configured (unpadded): what a LayerNorm normalises over
};

int make_netl(int B, const ts_net_desc* d, NetL* n) {
    TS_REQUIRE(d != nullptr, TS_ERR_INVALID_ARG, "net: descriptor is NULL");
    TS_REQUIRE(d->obs_dim >= 1 && d->obs_dim <= 65536 && d->n_hidden >= 1 && d->n_hidden <= TS_NET_MAX_HIDDEN, TS_ERR_INVALID_ARG,
               "net: obs_dim >= 1 and 1 .. %d hidden layers", TS_NET_MAX_HIDDEN);
    TS_REQUIRE(d->activation >= TS_NET_ACT_TANH && d->activation <= TS_NET_ACT_NONE, TS_ERR_UNSUPPORTED,
               "net: activation must be tanh, ReLU or none");
    n->L = d->n_hidden + 1; n->obs = (int)d->obs_dim; n->k0 = (n->obs + 31) / 32 * 32; n->act_fn = d->activation;
    n->csigma = (d->flags & TS_NET_CONDITIONED_SIGMA) ? 1 : 0;
    n->ln = (d->flags & TS_NET_LAYERNORM) ? 1 : 0;
    n->ln_eps = n->ln ? (float)(d->ln_eps > 0.0 ? d->ln_eps : 1e-5) : 0.f;
    n->width[0] = n->k0; n->tw[0] = n->obs;
    n->wmax = HEAD;
    for (int i = 0; i < d->n_hidden; ++i) {
        TS_REQUIRE(d->hidden[i] >= 1 && d->hidden[i] <= 1024, TS_ERR_UNSUPPORTED, "net: hidden widths must be in [1, 1024]");
        n->tw[i + 1] = (int)d->hidden[i];
        n->width[i + 1] = ((int)d->hidden[i] + 31) / 32 * 32;
        n->wmax = std::max(n->wmax, n->width[i + 1]);
    }
    n->width[n->L] = HEAD; n->tw[n->L] = HEAD;
    int64_t o = 0;
    for (int i = 0; i < n->L; ++i) {
        n->l[i] = ts::ConvGeom{B, 1, 1, n->width[i], 1, 1, 1, 1, 1, n->width[i + 1]};
        n->off[i] = o;
        o += n->l[i].param_elems();
        n->ln_off[i] = o;
        if (n->ln && i + 1 < n->L) o += 2 * (int64_t)n->width[i + 1];
    }
    n->off[n->L] = o;
    return TS_OK;
}

// h[i] = output of layer i (h[L - 1] = the head's 32 columns); layer norm: xh[i] = the normalised pre-activation of hidden layer
// i, rs[i] = 1 / sqrt(var + eps) per row (what the backward pass needs)
struct ActL { float* h[MAXL]; float* xh[MAXL]; float* rs[MAXL]; };

ActL take_actl(Carve& c, const NetL& n, int64_t B) {
    ActL a{};
    for (int i = 0; i < n.L; ++i) a.h[i] = c.f(B * n.width[i + 1]);
    if (n.ln)
        for (int i = 0; i + 1 < n.L; ++i) { a.xh[i] = c.f(B * n.width[i + 1]); a.rs[i] = c.f(B); }
    return a;
}
size_t actl_bytes(const NetL& n, int64_t B) {
    size_t s = 0;
    for (int i = 0; i < n.L; ++i) s += al(4 * B * n.width[i + 1]);
    if (n.ln)
        for (int i = 0; i + 1 < n.L; ++i) s += al(4 * B * n.width[i + 1]) + al(4 * B);
    return s;
}
size_t split_floats(const NetL& n) {
    size_t s = 4;
    for (int i = 0; i < n.L; ++i) { const int ns = ts::conv_fwd_splits(n.l[i]); if (ns > 1) s = std::max(s, (size_t)ns * n.l[i].out_elems()); }
    return s;
}
constexpr int LN_PART_BLOCKS = 256;            // workgroups (= partial sums of d gamma / d beta) of the layer-norm backward pass
size_t slab_floats(const NetL& n) {
    size_t s = 0;
    for (int i = 0; i < n.L; ++i) s = std::max(s, (size_t)ts::conv_wgrad_splits(n.l[i]) * n.l[i].param_elems());
    if (n.ln) s = std::max(s, (size_t)LN_PART_BLOCKS * 2 * n.wmax);      // (the slab area doubles as that pass's partial sums)
    return s;
}

// dh *= (h > 0)   (backward through ReLU; h is the layer's OUTPUT, torch's threshold_backward uses the same sign test)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dh[i] = h[i] > 0.f ? dh[i] : 0.f;
}

// ---- layer norm between a hidden Linear and its activation: MLP(norm_layer=nn.LayerNorm), utils/net/common.py:25-39 ----------
// One wavefront per row.  In place on the GEMM's output: h <- act(gamma * xhat + beta), xhat = (z - mean) * rstd over the layer's
// `n` configured features (torch.nn.LayerNorm: biased variance, eps inside the square root); columns [n, pitch) stay zero.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void ln_fwd_kernel(float* __restrict__ h, float* __restrict__ xhat, float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, int64_t B, int n,
                                                     int pitch, float eps, int act_fn) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float* r = h + row * pitch;
    float sm = 0.f;
    for (int j = lane; j < n; j += 64) sm += r[j];
    const float mean = wave_sum(sm) / (float)n;
    float q = 0.f;
    for (int j = lane; j < n; j += 64) { const float d = r[j] - mean; q += d * d; }
    const float rs = 1.f / sqrtf(wave_sum(q) / (float)n + eps);
    for (int j = lane; j < pitch; j += 64) {
        float xh = 0.f, out = 0.f;
        if (j < n) {
            xh = (r[j] - mean) * rs;
            const float y = xh * gamma[j] + beta[j];
            out = act_fn == TS_NET_ACT_TANH ? tanhf(y) : (act_fn == TS_NET_ACT_RELU ? fmaxf(y, 0.f) : y);
        }
        if (xhat) xhat[row * pitch + j] = xh;
        r[j] = out;
    }
    if (lane == 0 && rstd) rstd[row] = rs;
}

// Backward through the layer norm.  dy (in: gradient w.r.t. gamma * xhat + beta, i.e. behind the activation's derivative; out: the
// gradient w.r.t. the GEMM's output): dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.  Every workgroup also sums
// dy * xhat and dy over its rows (wave-strided, fixed order) into part[block][2][pitch]; ln_bwd_finish_kernel adds the workgroups'
// partial sums in order -> d gamma, d beta.  pitch <= 1024: at most 16 columns per lane.
__global__ __launch_bounds__(256) void ln_bwd_kernel(float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int64_t B, int n, int pitch,
                                                     float* __restrict__ part) {
    __shared__ float red[2][4][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float ag[16], ab[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < B; row += (int64_t)gridDim.x * 4) {
        float* d = dy + row * pitch;
        const float* xh = xhat + row * pitch;
        float g[16], x[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = lane + 64 * k;
            g[k] = 0.f; x[k] = 0.f;
            if (j < n) {
                const float dv = d[j];
                x[k] = xh[j];
                g[k] = dv * gamma[j];
                ag[k] += dv * x[k];
                ab[k] += dv;
                s1 += g[k];
                s2 += g[k] * x[k];
            }
        }
        const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n, rs = rstd[row];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = lane + 64 * k;
            if (j < pitch) d[j] = j < n ? rs * (g[k] - m1 - x[k] * m2) : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { red[0][wave][lane + 64 * k] = ag[k]; red[1][wave][lane + 64 * k] = ab[k]; }
    __syncthreads();
    for (int j = threadIdx.x; j < pitch; j += 256) {
        part[((int64_t)blockIdx.x * 2 + 0) * pitch + j] = ((red[0][0][j] + red[0][1][j]) + red[0][2][j]) + red[0][3][j];
        part[((int64_t)blockIdx.x * 2 + 1) * pitch + j] = ((red[1][0][j] + red[1][1][j]) + red[1][2][j]) + red[1][3][j];
    }
}
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* __restrict__ part, int n_blocks, int pitch,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= pitch) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < n_blocks; ++k) { a += part[((int64_t)k * 2 + 0) * pitch + j]; b += part[((int64_t)k * 2 + 1) * pitch + j]; }
    dgamma[j] = a; dbeta[j] = b;
}

int forward_l(hipStream_t s, ts_workspace* ws, const NetL& n, const float* p, const float* x, const ActL& a, float* split, int64_t B) {
    const float* in = x;
    for (int i = 0; i < n.L; ++i) {
        const bool hidden = i + 1 < n.L;
        const bool ln = hidden && n.ln;
        if (int rc = ts::conv_forward(s, n.l[i], in, p + n.off[i], a.h[i], hidden && !ln && n.act_fn == TS_NET_ACT_RELU, split, ws)) return rc;
        if (ln) {                                      // Linear -> LayerNorm -> activation (one kernel for the last two)
            const int w = n.width[i + 1];
            hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, a.h[i], a.xh[i], a.rs[i], p + n.ln_off[i],
                               p + n.ln_off[i] + w, B, n.tw[i + 1], w, n.ln_eps, n.act_fn);
            TS_LAUNCH_CHECK();
        } else if (hidden && n.act_fn == TS_NET_ACT_TANH) {
            const int64_t cnt = B * n.width[i + 1];
            hipLaunchKernelGGL(tanh_kernel, dim3((unsigned)ts::ceil_div(cnt, 256)), dim3(256), 0, s, a.h[i], cnt);
            TS_LAUNCH_CHECK();
        }
        in = a.h[i];
    }
    return TS_OK;
}

// grad[0 .. off[L]) = J^T d_out; dha / dhb: two [B, wmax] scratch buffers (the hidden gradients ping-pong between them)
int backward_l(hipStream_t s, ts_workspace* ws, const NetL& n, const float* p, const float* x, const ActL& a, const float* d_out,
               float* grad, float* dha, float* dhb, float* slabs, int64_t B) {
    const float* dy = d_out;
    for (int i = n.L - 1; i >= 0; --i) {
        const float* xin = i == 0 ? x : a.h[i - 1];
        if (int rc = ts::conv_wgrad(s, n.l[i], xin, dy, slabs, ws)) return rc;
        if (int rc = ts::slab_sum(s, slabs, ts::conv_wgrad_splits(n.l[i]), n.l[i].param_elems(), grad + n.off[i])) return rc;
        if (i > 0) {
            float* dx = (dy == dha) ? dhb : dha;
            if (int rc = ts::conv_dgrad(s, n.l[i], dy, p + n.off[i], nullptr, dx, ws)) return rc;
            const int64_t cnt = B * n.width[i];
            const unsigned g = (unsigned)ts::ceil_div(cnt, 256);
            if (n.act_fn == TS_NET_ACT_TANH) hipLaunchKernelGGL(tanh_bwd_kernel, dim3(g), dim3(256), 0, s, dx, a.h[i - 1], cnt);
            else if (n.act_fn == TS_NET_ACT_RELU) hipLaunchKernelGGL(relu_bwd_kernel, dim3(g), dim3(256), 0, s, dx, a.h[i - 1], cnt);
            TS_LAUNCH_CHECK();
            if (n.ln) {                                // ... and back through hidden layer i - 1's layer norm (slabs: its partial sums)
                const int w = n.width[i];
                const int nb = (int)std::min<int64_t>(LN_PART_BLOCKS, ts::ceil_div(B, 4));
                hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, dx, a.xh[i - 1], a.rs[i - 1], p + n.ln_off[i - 1], B,
                                   n.tw[i], w, slabs);
                hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((unsigned)ts::ceil_div(w, 256)), dim3(256), 0, s, slabs, nb, w,
                                   grad + n.ln_off[i - 1], grad + n.ln_off[i - 1] + w);
                TS_LAUNCH_CHECK();
            }
            dy = dx;
        }
    }
    return TS_OK;
}

// forward-mode combine of a generic trunk's layer: t = (ta - bias_w[col]) + tb (see jvp_combine_kernel); out = t * act'(h) with the
// layer's own output h: 1 - h^2 (tanh), [h > 0] (ReLU: torch's threshold_backward sign test), 1 (no activation / the head)
__global__ __launch_bounds__(256) void jvp_combine_act_kernel(const float* __restrict__ ta, const float* __restrict__ tb,
                                                              const float* __restrict__ bias_w, const float* __restrict__ h,
                                                              int64_t n, int cols, int act_fn, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = (ta ? ta[i] - bias_w[i % cols] : 0.f) + tb[i];
    float d = 1.f;
    if (h) d = act_fn == TS_NET_ACT_TANH ? 1.f - h[i] * h[i] : (act_fn == TS_NET_ACT_RELU ? (h[i] > 0.f ? 1.f : 0.f) : 1.f);
    out[i] = t * d;
}

// dmu = J v for a direction v (same layout as the parameters) of a generic trunk, given the forward pass's activations:
// d_0 = (x V_0 + bv_0) act'(h_0);  d_i = (d_{i-1} W_i + h_{i-1} V_i + bv_i) act'(h_i);  head: no activation.
// ta / tb: [B, wmax] scratch; d[2]: the directional derivatives ping-pong; dmu: [B, HEAD].
int jvp_l(hipStream_t s, ts_workspace* ws, const NetL& n, const float* p, const float* v, const float* x, const ActL& a, float* ta,
          float* tb, float* const* d, float* dmu, float* split, int64_t B) {
    const float* dprev = nullptr;
    for (int i = 0; i < n.L; ++i) {
        const bool head = i + 1 == n.L;
        const int64_t cnt = B * n.width[i + 1];
        const float* hin = i == 0 ? x : a.h[i - 1];
        if (dprev) if (int rc = ts::conv_forward(s, n.l[i], dprev, p + n.off[i], ta, false, split, ws)) return rc;
        if (int rc = ts::conv_forward(s, n.l[i], hin, v + n.off[i], tb, false, split, ws)) return rc;
        float* out = head ? dmu : d[i & 1];
        hipLaunchKernelGGL(jvp_combine_act_kernel, dim3((unsigned)ts::ceil_div(cnt, 256)), dim3(256), 0, s, dprev ? ta : (const float*)nullptr,
                           tb, p + n.off[i] + (int64_t)n.width[i] * n.width[i + 1], head ? (const float*)nullptr : a.h[i], cnt,
                           n.width[i + 1], n.act_fn, out);
        TS_LAUNCH_CHECK();
        dprev = out;
    }
    return TS_OK;
}

}  // namespace

extern "C" {

int ts_npg_layout(int64_t obs_dim, int64_t hidden, int64_t act_dim, int64_t* h_out3) {
    Net3 n;
    if (int rc = make_net3(1, obs_dim, hidden, &n)) return rc;
    TS_REQUIRE(act_dim >= 1 && act_dim <= HEAD && h_out3, TS_ERR_INVALID_ARG, "ts_npg_layout: act_dim must be in [1, 32]");
    h_out3[0] = n.k0; h_out3[1] = n.off[3] + HEAD; h_out3[2] = n.off[3];
    return TS_OK;
}

int ts_npg_infer(ts_workspace* ws, const float* actor, const float* critic, int64_t obs_dim, int64_t hidden, int64_t act_dim,
                 const float* obs, const float* act, int64_t B, float* v_out, float* logp_out, float* mu_out,
                 ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_infer: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_npg_infer: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(obs && act_dim >= 1 && act_dim <= HEAD && (!v_out || critic) && ((!logp_out && !mu_out) || actor) &&
                   (!logp_out || act), TS_ERR_INVALID_ARG, "ts_npg_infer: bad argument");
    Net3 n;
    if (int rc = make_net3((int)B, obs_dim, hidden, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    if (act_dim <= 8 && ts::npg_fused_supported(obs_dim, hidden, act_dim)) {      // one forward kernel per network (ts_npg_q.h)
        if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + 4096)) return rc;
        float* x = static_cast<float*>(ws->base);
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
        TS_LAUNCH_CHECK();
        if (logp_out || mu_out)
            if (int rc = ts::npg_infer_fused(s, ws, actor, x, act, n.obs, n.k0, (int)act_dim, B, mu_out, (int)act_dim, logp_out))
                return rc;
        if (v_out)
            if (int rc = ts::npg_infer_fused(s, ws, critic, x, nullptr, n.obs, n.k0, 1, B, v_out, 1, nullptr)) return rc;
        return TS_OK;
    }
    if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + 2 * act_bytes(n, B) + al(4 * split_floats(n)) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const Act3 aa = take_act(c, n, B), ac = take_act(c, n, B);
    float* split = c.f(split_floats(n));
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    if (logp_out || mu_out)
        if (int rc = forward(s, ws, n, actor, x, aa, split, B)) return rc;
    if (v_out)
        if (int rc = forward(s, ws, n, critic, x, ac, split, B)) return rc;
    hipLaunchKernelGGL(infer_out_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, aa.out, ac.out, act,
                       actor ? actor + n.off[3] : (const float*)nullptr, B, (int)act_dim, v_out, logp_out, mu_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_npg_actor_step(ts_workspace* ws, float* actor, int64_t obs_dim, int64_t hidden, int64_t act_dim, const float* obs,
                      const float* act, const float* adv, const float* logp_old, int64_t B, const ts_npg_hparams* hp,
                      float* stats_out3, float* dbg_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_actor_step: workspace is NULL");
    TS_REQUIRE(actor && obs && act && adv && hp && stats_out3 && B >= 1 && act_dim >= 1 && act_dim <= HEAD, TS_ERR_INVALID_ARG,
               "ts_npg_actor_step: bad argument");
    TS_REQUIRE(hp->algo == 0 || hp->algo == 1, TS_ERR_INVALID_ARG, "ts_npg_actor_step: algo must be 0 (NPG) or 1 (TRPO)");
    TS_REQUIRE(hp->algo == 0 || (logp_old && hp->max_backtracks >= 1 && hp->max_backtracks <= 32 && hp->max_kl > 0.0),
               TS_ERR_INVALID_ARG, "ts_npg_actor_step: TRPO needs logp_old, max_kl > 0 and 1 <= max_backtracks <= 32");
    TS_REQUIRE(hp->cg_iters >= 1 && hp->cg_iters <= 100, TS_ERR_INVALID_ARG, "ts_npg_actor_step: bad cg_iters");
    Net3 n;
    if (int rc = make_net3((int)B, obs_dim, hidden, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int A = (int)act_dim;
    const int64_t P = n.off[3] + HEAD, sig = n.off[3];
    const int n_blocks = (int)ts::ceil_div(B, 256);
    const int n_cand = hp->algo == 1 ? hp->max_backtracks : 1;
    const size_t bytes = al(4 * B * n.k0) + 2 * act_bytes(n, B) + 4 * al(4 * B * n.hid) + 3 * al(4 * B * HEAD) +
                         2 * al(4 * B * n.hid) + al(4 * slab_floats(n)) + al(4 * split_floats(n)) +
                         (size_t)(6 + n_cand) * al(4 * P) + al(4 * (size_t)n_blocks * (2 + A)) + al(4 * (8 + 2 * n_cand)) + 4096;
    // hidden 64, obs <= 32, act <= 8: gradient, Fisher-vector products and candidate evaluations are one kernel + one small
    // sum each (ts_npg_q.h); other shapes: per-layer GEMM passes
    const bool fused = ts::npg_fused_supported(obs_dim, hidden, act_dim);
    const size_t fvp_floats = fused ? ts::npg_fused_slab_floats(obs_dim, B) : 0;
    const size_t eval_floats = fused ? ts::npg_fused_eval_floats(B, n_cand) : 0;
    if (int rc = ts::ws_reserve(ws, bytes + al(4 * fvp_floats) + al(4 * eval_floats) + (fused ? al(4 * B * 8) : 0))) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const Act3 a0 = take_act(c, n, B), a1 = take_act(c, n, B);           // activations at theta; at a candidate
    Jvp j{c.f(B * n.hid), c.f(B * n.hid), c.f(B * n.hid), c.f(B * n.hid), c.f(B * HEAD)};
    float* d_head = c.f(B * HEAD);
    float* u = c.f(B * HEAD);
    Bwd bw{c.f(B * n.hid), c.f(B * n.hid), c.f(slab_floats(n))};
    float* split = c.f(split_floats(n));
    float* g = c.f(P); float* cx = c.f(P); float* cr = c.f(P); float* cp = c.f(P); float* cz = c.f(P); float* fx = c.f(P);
    float* cands = c.f((size_t)n_cand * P);
    float* partial = c.f((size_t)n_blocks * (2 + A));
    float* sc = c.f(8 + 2 * n_cand);                                       // {rdotr, done, p.z, step, -, -, -, -, res...}
    float* fvp_slabs = fused ? c.f(fvp_floats) : nullptr;
    float* eval_part = fused ? c.f(eval_floats) : nullptr;
    float* mu_old = fused ? c.f(B * 8) : nullptr;
    float* step = sc + 3;
    float* res = sc + 8;

    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    // vanilla gradient of the surrogate (npg.py:152-158 / trpo.py:135-141)
    if (fused) {
        if (int rc = ts::npg_grad_fused(s, ws, actor, x, act, adv, hp->algo == 1 ? logp_old : (const float*)nullptr, n.obs, n.k0, A, B,
                                        fvp_slabs, g, stats_out3, mu_old))
            return rc;
    } else {
        if (int rc = forward(s, ws, n, actor, x, a0, split, B)) return rc;
        hipLaunchKernelGGL(actor_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a0.out, act, adv, logp_old, actor + sig,
                           hp->algo, B, A, d_head, partial);
        hipLaunchKernelGGL(actor_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, B, A, stats_out3, g + sig);
        TS_LAUNCH_CHECK();
        if (int rc = backward(s, ws, n, actor, x, a0, d_head, g, bw, B)) return rc;
    }

    // F v (+ damping v) for a direction v -> out
    auto fvp = [&](const float* v, float* out) -> int {
        if (fused)
            return ts::npg_fvp_fused(s, ws, actor, v, x, n.obs, n.k0, A, B, (float)hp->damping, fvp_slabs, out);
        if (int rc = jvp(s, ws, n, actor, v, x, a0, j, split, B)) return rc;
        hipLaunchKernelGGL(fisher_upstream_kernel, dim3((unsigned)ts::ceil_div(B * HEAD, 256)), dim3(256), 0, s, j.dmu, actor + sig,
                           B, A, u);
        TS_LAUNCH_CHECK();
        if (int rc = backward(s, ws, n, actor, x, a0, u, out, bw, B)) return rc;
        hipLaunchKernelGGL(fvp_finish_kernel, dim3((unsigned)ts::ceil_div(P, 256)), dim3(256), 0, s, out, v, P, sig, A,
                           (float)hp->damping);
        TS_LAUNCH_CHECK();
        return TS_OK;
    };

    // conjugate gradients (npg.py:202-224): x ~ F^-1 g; search direction = -x
    hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, s, g, cx, cr, cp, P, sc);
    for (int it = 0; it < hp->cg_iters; ++it) {
        if (int rc = fvp(cp, cz)) return rc;
        hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, s, cx, cr, cp, cz, P, (float)hp->residual_tol, sc);
        TS_LAUNCH_CHECK();
    }
    if (dbg_out) {                                    // {gradient, search direction x (sign flipped by the caller), F g + damping g}
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out, g, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out + P, cx, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        if (int rc = fvp(g, fx)) return rc;
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out + 2 * P, fx, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
    }
    const unsigned gp = (unsigned)ts::ceil_div(P, 256);
    auto eval = [&](const float* cand, float* out2, bool with_loss) -> int {     // kl(old || cand) [, surrogate at cand]
        if (int rc = forward(s, ws, n, cand, x, a1, split, B)) return rc;
        hipLaunchKernelGGL(kl_eval_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a0.out, actor + sig, a1.out, cand + sig, act,
                           adv, with_loss ? logp_old : (const float*)nullptr, B, A, partial);
        hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, B, out2);
        TS_LAUNCH_CHECK();
        return TS_OK;
    };
    if (hp->algo == 0) {                              // npg.py:170-177
        hipLaunchKernelGGL(candidate_kernel, dim3(gp), dim3(256), 0, s, actor, cx, P, (const float*)nullptr,
                           (float)hp->trust_region_size, 1.f, cands);
        if (fused)          // the evaluation's finish kernel takes the step as well (theta <- candidate, kl, step size 0)
            return ts::npg_eval_fused(s, ws, actor, cands, P, 1, x, act, adv, nullptr, mu_old, n.obs, n.k0, A, B, eval_part, res, actor,
                                      stats_out3);
        if (int rc = eval(cands, res, false)) return rc;
        TS_HIP_CHECK(hipMemcpyAsync(actor, cands, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemcpyAsync(stats_out3 + 1, res, sizeof(float), hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemsetAsync(stats_out3 + 2, 0, sizeof(float), s));
        return TS_OK;
    }
    // TRPO: step size (trpo.py:153-160), then every backtracking candidate, then the reference's choice among them
    if (int rc = fvp(cx, fx)) return rc;
    hipLaunchKernelGGL(trpo_step_size_kernel, dim3(1), dim3(1024), 0, s, cx, fx, P, (float)hp->max_kl, step);
    float cpow = 1.f;
    for (int k = 0; k < n_cand && !fused; ++k) {
        hipLaunchKernelGGL(candidate_kernel, dim3(gp), dim3(256), 0, s, actor, cx, P, step, 0.f, cpow, cands + (size_t)k * P);
        if (int rc = eval(cands + (size_t)k * P, res + 2 * k, true)) return rc;
        cpow = cpow * (float)hp->backtrack_coeff;
    }
    if (fused) {                                      // every backtracking candidate: one launch to form them, one to evaluate them
        hipLaunchKernelGGL(candidates_all_kernel, dim3(gp, (unsigned)n_cand), dim3(256), 0, s, actor, cx, P, step,
                           (float)hp->backtrack_coeff, cands);
        TS_LAUNCH_CHECK();
    }
    if (fused)
        if (int rc = ts::npg_eval_fused(s, ws, actor, cands, P, n_cand, x, act, adv, logp_old, mu_old, n.obs, n.k0, A, B, eval_part, res))
            return rc;
    hipLaunchKernelGGL(trpo_select_kernel, dim3((unsigned)std::min<int64_t>(gp, 64)), dim3(256), 0, s, actor, cands, P, res, n_cand,
                       (float)hp->max_kl, (float)hp->backtrack_coeff, step, stats_out3);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_npg_actor_grad(ts_workspace* ws, const float* actor, int64_t obs_dim, int64_t hidden, int64_t act_dim, const float* obs,
                      const float* act, const float* weight, int64_t B, float* loss_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_actor_grad: workspace is NULL");
    TS_REQUIRE(actor && obs && act && weight && loss_out && grad_out && B >= 1 && act_dim >= 1 && act_dim <= HEAD,
               TS_ERR_INVALID_ARG, "ts_npg_actor_grad: bad argument");
    Net3 n;
    if (int rc = make_net3((int)B, obs_dim, hidden, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int A = (int)act_dim;
    const int n_blocks = (int)ts::ceil_div(B, 256);
    if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + act_bytes(n, B) + al(4 * B * HEAD) + 2 * al(4 * B * n.hid) +
                                        al(4 * slab_floats(n)) + al(4 * split_floats(n)) + al(4 * (size_t)n_blocks * (2 + A)) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const Act3 a = take_act(c, n, B);
    float* d_head = c.f(B * HEAD);
    Bwd bw{c.f(B * n.hid), c.f(B * n.hid), c.f(slab_floats(n))};
    float* split = c.f(split_floats(n));
    float* partial = c.f((size_t)n_blocks * (2 + A));
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    if (int rc = forward(s, ws, n, actor, x, a, split, B)) return rc;
    hipLaunchKernelGGL(actor_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a.out, act, weight, (const float*)nullptr,
                       actor + n.off[3], 0, B, A, d_head, partial);
    hipLaunchKernelGGL(actor_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, B, A, loss_out, grad_out + n.off[3]);
    TS_LAUNCH_CHECK();
    return backward(s, ws, n, actor, x, a, d_head, grad_out, bw, B);
}

int ts_npg_critic_steps(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                        int64_t hidden, const float* obs, const float* returns, int64_t B, int64_t iters, double lr, double beta1,
                        double beta2, double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_critic_steps: workspace is NULL");
    TS_REQUIRE(critic && adam_m && adam_v && obs && returns && loss_out && B >= 1 && adam_step >= 1 && iters >= 1 && iters <= 4096,
               TS_ERR_INVALID_ARG, "ts_npg_critic_steps: bad argument");
    Net3 n;
    if (int rc = make_net3((int)B, obs_dim, hidden, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int64_t P = n.off[3];
    // hidden 64, obs <= 32: one kernel + one small sum per iteration's gradient (ts_npg_q.h, CRITIC pass)
    const bool fused = ts::npg_fused_supported(obs_dim, hidden, 1);
    const size_t slab_fl = fused ? ts::npg_fused_slab_floats(obs_dim, B) : 0;
    const size_t bytes = fused ? al(4 * B * n.k0) + al(4 * slab_fl) + al(4 * P) + 8192
                               : al(4 * B * n.k0) + act_bytes(n, B) + al(4 * B * HEAD) + 2 * al(4 * B * n.hid) + al(4 * slab_floats(n)) +
                                     al(4 * split_floats(n)) + al(4 * P) + al(4 * ts::ceil_div(B, 256)) + 8192;
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    if (fused) {
        float* slabs = c.f(slab_fl);
        float* grad = c.f(P);
        float* norm_part = c.f(1024);
        if (grad_out) grad = grad_out;
        for (int64_t it = 0; it < iters; ++it) {
            if (int rc = ts::npg_critic_grad_fused(s, ws, critic, x, returns, n.obs, n.k0, B, slabs, grad, loss_out)) return rc;
            if (lr < 0.0) continue;
            if (int rc = ts::adam_step(s, critic, adam_m, adam_v, grad, P, adam_step + it, lr, beta1, beta2, adam_eps, max_grad_norm,
                                       norm_part))
                return rc;
        }
        return TS_OK;
    }
    const Act3 a = take_act(c, n, B);
    float* d_head = c.f(B * HEAD);
    Bwd bw{c.f(B * n.hid), c.f(B * n.hid), c.f(slab_floats(n))};
    float* split = c.f(split_floats(n));
    float* grad = c.f(P);
    float* norm_part = c.f(1024);
    const int n_blocks = (int)ts::ceil_div(B, 256);
    float* lpart = c.f(n_blocks);
    if (grad_out) grad = grad_out;
    for (int64_t it = 0; it < iters; ++it) {
        if (int rc = forward(s, ws, n, critic, x, a, split, B)) return rc;
        hipLaunchKernelGGL(critic_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a.out, returns, B, d_head, lpart);
        hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, s, lpart, n_blocks, 1.f / (float)B, loss_out);
        TS_LAUNCH_CHECK();
        if (int rc = backward(s, ws, n, critic, x, a, d_head, grad, bw, B)) return rc;
        if (lr < 0.0) continue;
        if (int rc = ts::adam_step(s, critic, adam_m, adam_v, grad, P, adam_step + it, lr, beta1, beta2, adam_eps, max_grad_norm,
                                   norm_part))
            return rc;
    }
    return TS_OK;
}

int ts_npg_critic_step(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                       int64_t hidden, const float* obs, const float* returns, int64_t B, double lr, double beta1, double beta2,
                       double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream) {
    return ts_npg_critic_steps(ws, critic, adam_m, adam_v, adam_step, obs_dim, hidden, obs, returns, B, 1, lr, beta1, beta2, adam_eps,
                               max_grad_norm, loss_out, grad_out, stream);
}

int ts_ppo_wide_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                     int64_t hidden, int64_t act_dim, const float* obs, const float* act, const float* adv, const float* returns,
                     const float* logp_old, const float* v_old, int64_t B, int64_t global_batch, const float* adv_stats,
                     const ts_ppo_hparams* hp, float* losses_out4, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_ppo_wide_step: workspace is NULL");
    TS_REQUIRE(params && obs && act && adv && returns && hp && losses_out4 && B >= 1 && global_batch >= B && act_dim >= 1 &&
                   act_dim <= HEAD, TS_ERR_INVALID_ARG, "ts_ppo_wide_step: bad argument");
    const bool a2c = hp->algo == 1;
    TS_REQUIRE(a2c || logp_old, TS_ERR_INVALID_ARG, "ts_ppo_wide_step: PPO needs logp_old");
    TS_REQUIRE(a2c || !hp->value_clip || v_old, TS_ERR_INVALID_ARG, "ts_ppo_wide_step: value_clip needs v_old");
    TS_REQUIRE(a2c || !hp->adv_norm || adv_stats, TS_ERR_INVALID_ARG, "ts_ppo_wide_step: adv_norm needs adv_stats");
    const bool apply = hp->lr >= 0.0;
    TS_REQUIRE(!apply || (adam_m && adam_v && adam_step >= 1), TS_ERR_INVALID_ARG, "ts_ppo_wide_step: Adam state missing");
    Net3 n;
    if (int rc = make_net3((int)B, obs_dim, hidden, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int A = (int)act_dim;
    const int64_t Pa = n.off[3] + HEAD, Pc = n.off[3], P = Pa + Pc;
    const int n_blocks = (int)ts::ceil_div(B, 256);
    if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + act_bytes(n, B) + al(4 * B * HEAD) + 2 * al(4 * B * n.hid) +
                                        al(4 * slab_floats(n)) + al(4 * split_floats(n)) + al(4 * P) +
                                        al(4 * (size_t)n_blocks * (2 + A)) + 8192))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const Act3 a = take_act(c, n, B);
    float* d_head = c.f(B * HEAD);
    Bwd bw{c.f(B * n.hid), c.f(B * n.hid), c.f(slab_floats(n))};
    float* split = c.f(split_floats(n));
    float* grad = c.f(P);
    float* partial = c.f((size_t)n_blocks * (2 + A));
    float* norm_part = c.f(1024);
    if (grad_out) grad = grad_out;
    const float* actor = params;
    const float* critic = params + Pa;
    const float inv_b = 1.0f / (float)global_batch;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    // ---- actor: forward, clipped surrogate (+ entropy gradient on log_sigma), backward
    if (int rc = forward(s, ws, n, actor, x, a, split, B)) return rc;
    WideLossP lp{};
    lp.eps_clip = (float)hp->eps_clip; lp.dual_clip = a2c ? 0.f : (float)(hp->dual_clip > 0.0 ? hp->dual_clip : 0.0);
    lp.ent_coef = (float)hp->ent_coef; lp.inv_b = inv_b; lp.a2c = a2c; lp.adv_norm = a2c ? 0 : hp->adv_norm;
    hipLaunchKernelGGL(ppo_wide_actor_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a.out, act, adv, logp_old,
                       actor + n.off[3], adv_stats, lp, B, A, d_head, partial);
    // loss = -sum(-term) / B with B = the LOCAL row count: rescale to the global mean below when they differ
    hipLaunchKernelGGL(actor_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, global_batch, A, losses_out4 + 1,
                       grad + n.off[3]);
    TS_LAUNCH_CHECK();
    if (int rc = backward(s, ws, n, actor, x, a, d_head, grad, bw, B)) return rc;
    // ---- critic
    if (int rc = forward(s, ws, n, critic, x, a, split, B)) return rc;
    hipLaunchKernelGGL(ppo_wide_critic_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a.out, returns, v_old, (float)hp->eps_clip,
                       a2c ? 0 : hp->value_clip, (float)hp->vf_coef, B, inv_b, d_head, partial);
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, inv_b, losses_out4 + 2);
    TS_LAUNCH_CHECK();
    if (int rc = backward(s, ws, n, critic, x, a, d_head, grad + Pa, bw, B)) return rc;
    hipLaunchKernelGGL(ppo_wide_total_kernel, dim3(1), dim3(64), 0, s, losses_out4, actor + n.off[3], A, (float)hp->vf_coef,
                       (float)hp->ent_coef);
    TS_LAUNCH_CHECK();
    if (!apply) return TS_OK;
    // joint clip_grad_norm_ over actor + critic (a2c.py:103-107) + Adam
    return ts::optim_step(s, ts::optim_from(hp), params, adam_m, adam_v, grad, P, adam_step, hp->lr, hp->beta1, hp->beta2, hp->adam_eps,
                         hp->max_grad_norm > 0.0 ? hp->max_grad_norm : 0.0, norm_part);
}

int ts_net_layout(const ts_net_desc* net, int64_t act_dim, int64_t* h_out3) {
    NetL n;
    if (int rc = make_netl(1, net, &n)) return rc;
    TS_REQUIRE(act_dim >= 1 && act_dim <= HEAD && h_out3, TS_ERR_INVALID_ARG, "ts_net_layout: act_dim must be in [1, 32]");
    h_out3[0] = n.k0; h_out3[1] = n.off[n.L] + HEAD; h_out3[2] = n.off[n.L];
    return TS_OK;
}

int ts_ppo_net_infer(ts_workspace* ws, const float* actor, const float* critic, const ts_net_desc* actor_net,
                     const ts_net_desc* critic_net, int64_t act_dim, const float* obs, const float* act, int64_t B,
                     float* v_out, float* logp_out, float* mu_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_ppo_net_infer: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_ppo_net_infer: negative batch");
    if (B == 0) return TS_OK;
    const bool want_a = logp_out || mu_out;
    TS_REQUIRE(obs && act_dim >= 1 && act_dim <= HEAD && (!v_out || (critic && critic_net)) && (!want_a || (actor && actor_net)) &&
                   (!logp_out || act), TS_ERR_INVALID_ARG, "ts_ppo_net_infer: bad argument");
    const float mu_bound = (want_a && actor_net->max_action > 0.0) ? (float)actor_net->max_action : 0.f;
    NetL na{}, nc{};
    if (want_a) if (int rc = make_netl((int)B, actor_net, &na)) return rc;
    if (v_out) if (int rc = make_netl((int)B, critic_net, &nc)) return rc;
    const NetL& any = want_a ? na : nc;
    TS_REQUIRE(!(want_a && v_out) || na.obs == nc.obs, TS_ERR_SHAPE, "ts_ppo_net_infer: actor and critic read different observations");
    hipStream_t s = ts::as_stream(stream);
    const size_t sp = std::max(want_a ? split_floats(na) : 4, v_out ? split_floats(nc) : 4);
    if (int rc = ts::ws_reserve(ws, al(4 * B * any.k0) + (want_a ? actl_bytes(na, B) : 0) + (v_out ? actl_bytes(nc, B) : 0) + al(4 * sp) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * any.k0);
    ActL aa{}, ac{};
    if (want_a) aa = take_actl(c, na, B);
    if (v_out) ac = take_actl(c, nc, B);
    float* split = c.f(sp);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * any.k0, 256)), dim3(256), 0, s, obs, B, any.obs, any.k0, x);
    TS_LAUNCH_CHECK();
    if (want_a) if (int rc = forward_l(s, ws, na, actor, x, aa, split, B)) return rc;
    if (v_out) if (int rc = forward_l(s, ws, nc, critic, x, ac, split, B)) return rc;
    if (want_a && na.csigma) {
        TS_REQUIRE(act_dim <= CS_COL, TS_ERR_UNSUPPORTED, "conditioned sigma: at most %d actions", CS_COL);
        hipLaunchKernelGGL(infer_out_cs_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, aa.h[na.L - 1],
                           v_out ? ac.h[nc.L - 1] : nullptr, act, B, (int)act_dim, v_out, logp_out, mu_out, mu_bound);
    } else {
        hipLaunchKernelGGL(infer_out_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, want_a ? aa.h[na.L - 1] : nullptr,
                           v_out ? ac.h[nc.L - 1] : nullptr, act, want_a ? actor + na.off[na.L] : (const float*)nullptr, B, (int)act_dim,
                           v_out, logp_out, mu_out, mu_bound);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_ppo_net_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step,
                    const ts_net_desc* actor_net, const ts_net_desc* critic_net, int64_t act_dim, const float* obs,
                    const float* act, const float* adv, const float* returns, const float* logp_old, const float* v_old,
                    int64_t B, int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp, float* losses_out4,
                    float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_ppo_net_step: workspace is NULL");
    TS_REQUIRE(params && obs && act && adv && returns && hp && losses_out4 && B >= 1 && global_batch >= B && act_dim >= 1 &&
                   act_dim <= HEAD, TS_ERR_INVALID_ARG, "ts_ppo_net_step: bad argument");
    const bool a2c = hp->algo == 1;
    TS_REQUIRE(a2c || logp_old, TS_ERR_INVALID_ARG, "ts_ppo_net_step: PPO needs logp_old");
    TS_REQUIRE(a2c || !hp->value_clip || v_old, TS_ERR_INVALID_ARG, "ts_ppo_net_step: value_clip needs v_old");
    TS_REQUIRE(a2c || !hp->adv_norm || adv_stats, TS_ERR_INVALID_ARG, "ts_ppo_net_step: adv_norm needs adv_stats");
    const bool apply = hp->lr >= 0.0;
    TS_REQUIRE(!apply || (adam_m && adam_v && adam_step >= 1), TS_ERR_INVALID_ARG, "ts_ppo_net_step: Adam state missing");
    NetL na, nc;
    if (int rc = make_netl((int)B, actor_net, &na)) return rc;
    if (int rc = make_netl((int)B, critic_net, &nc)) return rc;
    TS_REQUIRE(na.obs == nc.obs, TS_ERR_SHAPE, "ts_ppo_net_step: actor and critic read different observations");
    hipStream_t s = ts::as_stream(stream);
    const int A = (int)act_dim;
    const int64_t Pa = na.off[na.L] + HEAD, Pc = nc.off[nc.L], P = Pa + Pc;
    const int n_blocks = (int)ts::ceil_div(B, 256);
    const int wmax = std::max(na.wmax, nc.wmax);
    const size_t sp = std::max(split_floats(na), split_floats(nc)), sl = std::max(slab_floats(na), slab_floats(nc));
    if (int rc = ts::ws_reserve(ws, al(4 * B * na.k0) + std::max(actl_bytes(na, B), actl_bytes(nc, B)) + al(4 * B * HEAD) +
                                        2 * al(4 * B * wmax) + al(4 * sl) + al(4 * sp) + al(4 * P) +
                                        al(4 * (size_t)n_blocks * (2 + A)) + 8192))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * na.k0);
    char* act_base = c.p;                                   // the two networks run one after the other: same activation area
    const ActL aa = take_actl(c, na, B);
    Carve c2{act_base};
    const ActL ac = take_actl(c2, nc, B);
    if (c2.p > c.p) c.p = c2.p;
    float* d_head = c.f(B * HEAD);
    float* dha = c.f(B * wmax);
    float* dhb = c.f(B * wmax);
    float* slabs = c.f(sl);
    float* split = c.f(sp);
    float* grad = c.f(P);
    float* partial = c.f((size_t)n_blocks * (2 + A));
    float* norm_part = c.f(1024);
    if (grad_out) grad = grad_out;
    const float* actor = params;
    const float* critic = params + Pa;
    const float inv_b = 1.0f / (float)global_batch;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * na.k0, 256)), dim3(256), 0, s, obs, B, na.obs, na.k0, x);
    TS_LAUNCH_CHECK();
    // ---- actor: forward, clipped surrogate (+ entropy gradient on log_sigma), backward   (ppo.py:181-196, a2c.py:262-268)
    if (int rc = forward_l(s, ws, na, actor, x, aa, split, B)) return rc;
    WideLossP lp{};
    lp.eps_clip = (float)hp->eps_clip; lp.dual_clip = a2c ? 0.f : (float)(hp->dual_clip > 0.0 ? hp->dual_clip : 0.0);
    lp.ent_coef = (float)hp->ent_coef; lp.inv_b = inv_b; lp.a2c = a2c; lp.adv_norm = a2c ? 0 : hp->adv_norm;
    lp.mu_bound = (float)(actor_net->max_action > 0.0 ? actor_net->max_action : 0.0);      // ts_net_desc.max_action
    if (na.csigma) {
        TS_REQUIRE(A <= CS_COL, TS_ERR_UNSUPPORTED, "conditioned sigma: at most %d actions", CS_COL);
        hipLaunchKernelGGL(ppo_net_actor_loss_cs_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, aa.h[na.L - 1], act, adv, logp_old,
                           adv_stats, lp, B, A, d_head, partial);
        hipLaunchKernelGGL(ppo_net_cs_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, global_batch, losses_out4,
                           grad + na.off[na.L]);
    } else {
        hipLaunchKernelGGL(ppo_wide_actor_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, aa.h[na.L - 1], act, adv, logp_old,
                           actor + na.off[na.L], adv_stats, lp, B, A, d_head, partial);
        hipLaunchKernelGGL(actor_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, global_batch, A, losses_out4 + 1,
                           grad + na.off[na.L]);
    }
    TS_LAUNCH_CHECK();
    if (int rc = backward_l(s, ws, na, actor, x, aa, d_head, grad, dha, dhb, slabs, B)) return rc;
    // ---- critic   (ppo.py:198-208, a2c.py:270)
    if (int rc = forward_l(s, ws, nc, critic, x, ac, split, B)) return rc;
    hipLaunchKernelGGL(ppo_wide_critic_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, ac.h[nc.L - 1], returns, v_old,
                       (float)hp->eps_clip, a2c ? 0 : hp->value_clip, (float)hp->vf_coef, B, inv_b, d_head, partial);
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, inv_b, losses_out4 + 2);
    TS_LAUNCH_CHECK();
    if (int rc = backward_l(s, ws, nc, critic, x, ac, d_head, grad + Pa, dha, dhb, slabs, B)) return rc;
    if (na.csigma) hipLaunchKernelGGL(ppo_net_total_cs_kernel, dim3(1), dim3(64), 0, s, losses_out4, (float)hp->vf_coef, (float)hp->ent_coef);
    else hipLaunchKernelGGL(ppo_wide_total_kernel, dim3(1), dim3(64), 0, s, losses_out4, actor + na.off[na.L], A, (float)hp->vf_coef,
                            (float)hp->ent_coef);
    TS_LAUNCH_CHECK();
    if (!apply) return TS_OK;
    // joint clip_grad_norm_ over actor + critic (a2c.py:103-107) + Adam
    return ts::optim_step(s, ts::optim_from(hp), params, adam_m, adam_v, grad, P, adam_step, hp->lr, hp->beta1, hp->beta2, hp->adam_eps,
                         hp->max_grad_norm > 0.0 ? hp->max_grad_norm : 0.0, norm_part);
}


// ---- NPG / TRPO on trunks of any depth / widths / activation (ts_net_desc; round 6) -----------------------------------------
// The per-layer path of ts_npg_actor_step / ts_npg_critic_steps over forward_l / backward_l / jvp_l: the same loss, Fisher,
// conjugate-gradient, candidate and selection kernels on the head's 32 columns.
int ts_npg_net_actor_step(ts_workspace* ws, float* actor, const ts_net_desc* net, int64_t act_dim, const float* obs, const float* act,
                          const float* adv, const float* logp_old, int64_t B, const ts_npg_hparams* hp, float* stats_out3,
                          float* dbg_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_net_actor_step: workspace is NULL");
    TS_REQUIRE(actor && net && obs && act && adv && hp && stats_out3 && B >= 1 && act_dim >= 1 && act_dim <= HEAD, TS_ERR_INVALID_ARG,
               "ts_npg_net_actor_step: bad argument");
    TS_REQUIRE(hp->algo == 0 || hp->algo == 1, TS_ERR_INVALID_ARG, "ts_npg_net_actor_step: algo must be 0 (NPG) or 1 (TRPO)");
    TS_REQUIRE(hp->algo == 0 || (logp_old && hp->max_backtracks >= 1 && hp->max_backtracks <= 32 && hp->max_kl > 0.0),
               TS_ERR_INVALID_ARG, "ts_npg_net_actor_step: TRPO needs logp_old, max_kl > 0 and 1 <= max_backtracks <= 32");
    TS_REQUIRE(hp->cg_iters >= 1 && hp->cg_iters <= 100, TS_ERR_INVALID_ARG, "ts_npg_net_actor_step: bad cg_iters");
    TS_REQUIRE(!(net->flags & TS_NET_CONDITIONED_SIGMA) && !(net->max_action > 0.0), TS_ERR_UNSUPPORTED,
               "ts_npg_net_actor_step: an unbounded actor with a state-independent sigma_param is required");
    TS_REQUIRE(!(net->flags & TS_NET_LAYERNORM), TS_ERR_UNSUPPORTED, "ts_npg_net_actor_step: no forward-mode pass through a layer norm");
    NetL n;
    if (int rc = make_netl((int)B, net, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int A = (int)act_dim;
    const int64_t P = n.off[n.L] + HEAD, sig = n.off[n.L];
    const int n_blocks = (int)ts::ceil_div(B, 256);
    const int n_cand = hp->algo == 1 ? hp->max_backtracks : 1;
    const size_t sl = slab_floats(n), sp = split_floats(n);
    if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + 2 * actl_bytes(n, B) + 6 * al(4 * B * n.wmax) + 3 * al(4 * B * HEAD) + al(4 * sl) +
                                        al(4 * sp) + (size_t)(6 + n_cand) * al(4 * P) + al(4 * (size_t)n_blocks * (2 + A)) +
                                        al(4 * (8 + 2 * n_cand)) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const ActL a0 = take_actl(c, n, B), a1 = take_actl(c, n, B);         // activations at theta; at a candidate
    float* ta = c.f(B * n.wmax); float* tb = c.f(B * n.wmax);
    float* dd[2] = {c.f(B * n.wmax), c.f(B * n.wmax)};
    float* dha = c.f(B * n.wmax); float* dhb = c.f(B * n.wmax);
    float* dmu = c.f(B * HEAD); float* d_head = c.f(B * HEAD); float* u = c.f(B * HEAD);
    float* slabs = c.f(sl);
    float* split = c.f(sp);
    float* g = c.f(P); float* cx = c.f(P); float* cr = c.f(P); float* cp = c.f(P); float* cz = c.f(P); float* fx = c.f(P);
    float* cands = c.f((size_t)n_cand * P);
    float* partial = c.f((size_t)n_blocks * (2 + A));
    float* sc = c.f(8 + 2 * n_cand);                                       // {rdotr, done, p.z, step, -, -, -, -, res...}
    float* step = sc + 3;
    float* res = sc + 8;
    const float* out0 = a0.h[n.L - 1];

    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    // vanilla gradient of the surrogate (npg.py:152-158 / trpo.py:135-141)
    if (int rc = forward_l(s, ws, n, actor, x, a0, split, B)) return rc;
    hipLaunchKernelGGL(actor_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, out0, act, adv, logp_old, actor + sig, hp->algo, B, A,
                       d_head, partial);
    hipLaunchKernelGGL(actor_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, B, A, stats_out3, g + sig);
    TS_LAUNCH_CHECK();
    if (int rc = backward_l(s, ws, n, actor, x, a0, d_head, g, dha, dhb, slabs, B)) return rc;

    auto fvp = [&](const float* v, float* out) -> int {                   // F v (+ damping v) for a direction v
        if (int rc = jvp_l(s, ws, n, actor, v, x, a0, ta, tb, dd, dmu, split, B)) return rc;
        hipLaunchKernelGGL(fisher_upstream_kernel, dim3((unsigned)ts::ceil_div(B * HEAD, 256)), dim3(256), 0, s, dmu, actor + sig, B, A, u);
        TS_LAUNCH_CHECK();
        if (int rc = backward_l(s, ws, n, actor, x, a0, u, out, dha, dhb, slabs, B)) return rc;
        hipLaunchKernelGGL(fvp_finish_kernel, dim3((unsigned)ts::ceil_div(P, 256)), dim3(256), 0, s, out, v, P, sig, A, (float)hp->damping);
        TS_LAUNCH_CHECK();
        return TS_OK;
    };
    // conjugate gradients (npg.py:202-224): x ~ F^-1 g; search direction = -x
    hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, s, g, cx, cr, cp, P, sc);
    for (int it = 0; it < hp->cg_iters; ++it) {
        if (int rc = fvp(cp, cz)) return rc;
        hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, s, cx, cr, cp, cz, P, (float)hp->residual_tol, sc);
        TS_LAUNCH_CHECK();
    }
    if (dbg_out) {                                    // {gradient, search direction x (sign flipped by the caller), F g + damping g}
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out, g, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out + P, cx, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        if (int rc = fvp(g, fx)) return rc;
        TS_HIP_CHECK(hipMemcpyAsync(dbg_out + 2 * P, fx, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
    }
    const unsigned gp = (unsigned)ts::ceil_div(P, 256);
    auto eval = [&](const float* cand, float* out2, bool with_loss) -> int {     // kl(old || cand) [, surrogate at cand]
        if (int rc = forward_l(s, ws, n, cand, x, a1, split, B)) return rc;
        hipLaunchKernelGGL(kl_eval_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, out0, actor + sig, a1.h[n.L - 1], cand + sig, act, adv,
                           with_loss ? logp_old : (const float*)nullptr, B, A, partial);
        hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(256), 0, s, partial, n_blocks, B, out2);
        TS_LAUNCH_CHECK();
        return TS_OK;
    };
    if (hp->algo == 0) {                              // npg.py:170-177
        hipLaunchKernelGGL(candidate_kernel, dim3(gp), dim3(256), 0, s, actor, cx, P, (const float*)nullptr, (float)hp->trust_region_size,
                           1.f, cands);
        if (int rc = eval(cands, res, false)) return rc;
        TS_HIP_CHECK(hipMemcpyAsync(actor, cands, sizeof(float) * P, hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemcpyAsync(stats_out3 + 1, res, sizeof(float), hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemsetAsync(stats_out3 + 2, 0, sizeof(float), s));
        return TS_OK;
    }
    // TRPO: step size (trpo.py:153-160), then every backtracking candidate, then the reference's choice among them
    if (int rc = fvp(cx, fx)) return rc;
    hipLaunchKernelGGL(trpo_step_size_kernel, dim3(1), dim3(1024), 0, s, cx, fx, P, (float)hp->max_kl, step);
    float cpow = 1.f;
    for (int k = 0; k < n_cand; ++k) {
        hipLaunchKernelGGL(candidate_kernel, dim3(gp), dim3(256), 0, s, actor, cx, P, step, 0.f, cpow, cands + (size_t)k * P);
        if (int rc = eval(cands + (size_t)k * P, res + 2 * k, true)) return rc;
        cpow = cpow * (float)hp->backtrack_coeff;
    }
    hipLaunchKernelGGL(trpo_select_kernel, dim3((unsigned)std::min<int64_t>(gp, 64)), dim3(256), 0, s, actor, cands, P, res, n_cand,
                       (float)hp->max_kl, (float)hp->backtrack_coeff, step, stats_out3);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_npg_net_critic_steps(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, const ts_net_desc* net,
                            const float* obs, const float* returns, int64_t B, int64_t iters, double lr, double beta1, double beta2,
                            double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_npg_net_critic_steps: workspace is NULL");
    TS_REQUIRE(critic && adam_m && adam_v && net && obs && returns && loss_out && B >= 1 && adam_step >= 1 && iters >= 1 && iters <= 4096,
               TS_ERR_INVALID_ARG, "ts_npg_net_critic_steps: bad argument");
    TS_REQUIRE(!(net->flags & TS_NET_LAYERNORM), TS_ERR_UNSUPPORTED, "ts_npg_net_critic_steps: layer-norm trunks are a PPO / A2C feature");
    NetL n;
    if (int rc = make_netl((int)B, net, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const int64_t P = n.off[n.L];
    const int n_blocks = (int)ts::ceil_div(B, 256);
    const size_t sl = slab_floats(n), sp = split_floats(n);
    if (int rc = ts::ws_reserve(ws, al(4 * B * n.k0) + actl_bytes(n, B) + al(4 * B * HEAD) + 2 * al(4 * B * n.wmax) + al(4 * sl) + al(4 * sp) +
                                        al(4 * P) + al(4 * n_blocks) + 8192))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.f(B * n.k0);
    const ActL a = take_actl(c, n, B);
    float* d_head = c.f(B * HEAD);
    float* dha = c.f(B * n.wmax); float* dhb = c.f(B * n.wmax);
    float* slabs = c.f(sl);
    float* split = c.f(sp);
    float* grad = c.f(P);
    float* norm_part = c.f(1024);
    float* lpart = c.f(n_blocks);
    if (grad_out) grad = grad_out;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.k0, 256)), dim3(256), 0, s, obs, B, n.obs, n.k0, x);
    TS_LAUNCH_CHECK();
    for (int64_t it = 0; it < iters; ++it) {                              // npg.py:179-187: MSE to the returns, clip + Adam
        if (int rc = forward_l(s, ws, n, critic, x, a, split, B)) return rc;
        hipLaunchKernelGGL(critic_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, a.h[n.L - 1], returns, B, d_head, lpart);
        hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, s, lpart, n_blocks, 1.f / (float)B, loss_out);
        TS_LAUNCH_CHECK();
        if (int rc = backward_l(s, ws, n, critic, x, a, d_head, grad, dha, dhb, slabs, B)) return rc;
        if (lr < 0.0) continue;
        if (int rc = ts::adam_step(s, critic, adam_m, adam_v, grad, P, adam_step + it, lr, beta1, beta2, adam_eps, max_grad_norm, norm_part))
            return rc;
    }
    return TS_OK;
}

}  // extern "C"
