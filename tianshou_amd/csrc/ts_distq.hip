// ts_distq.hip -- distributional Q-learning (QRDQN, C51) on the Atari networks for gfx950.
//
// Replaces, on device-resident NHWC observations:
//   QRDQNet.forward / C51Net.forward          tianshou/env/atari/atari_network.py:227-235 / :141-151
//   QRDQNPolicy.compute_q_value               tianshou/algorithm/modelfree/qrdqn.py:19-21
//   C51Policy.compute_q_value                 modelfree/c51.py:66-67
//   QRDQN._target_q, C51._target_dist (the lagged net's distribution of the online net's greedy action)
//                                             qrdqn.py:93-104, c51.py:123-132
//   QRDQN._update_with_batch                  qrdqn.py:106-131 (quantile Huber loss, new priorities)
//   C51._update_with_batch                    c51.py:133-160  (projection, cross entropy, new priorities)
//   Optimizer.step                            algorithm_base.py:484-500 (clip_grad_norm_ + Adam)
// The network is DQNet with n_act * n_atoms outputs: the trunk, fc1 and the head all run on the fp32-MFMA
// implicit-GEMM kernels of ts_conv.hip; this file adds the per-sample distribution kernels and the orchestration.
// Flat parameter layout: the ts_dqn layout with the head matrix [513, W], W = n_act * n_atoms rounded up to a multiple
// of 32 (GEMM tile width); column a * n_atoms + j, the padding columns are and stay exactly zero.
#include <algorithm>

#include "ts_common.h"
#include "ts_conv.h"

#pragma clang fp contract(off)

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
}

namespace {

constexpr int HIDDEN = 512;
constexpr int MAX_ATOMS = 256;
constexpr int MAX_ACT = 64;

struct Net {
    ts::ConvGeom l[5];          // conv1, conv2, conv3, fc1, head (512 -> W)
    int64_t off[6];
    int n_act, n_atoms, ld;     // ld = W: row stride of the head output
};

int make_net(int B, int c, int h, int w, int n_act, int n_atoms, Net* n) {
    TS_REQUIRE(c >= 1 && h >= 1 && w >= 1 && n_act >= 1 && n_act <= MAX_ACT && n_atoms >= 2 && n_atoms <= MAX_ATOMS,
               TS_ERR_INVALID_ARG, "distq: bad network dimensions (n_act <= 64, 2 <= n_atoms <= 256)");
    static const int oc[3] = {32, 64, 64}, ks[3] = {8, 4, 3}, st[3] = {4, 2, 1};
    int ic = c, ih = h, iw = w;
    for (int i = 0; i < 3; ++i) {
        TS_REQUIRE(ih >= ks[i] && iw >= ks[i], TS_ERR_INVALID_ARG, "distq: observation too small for DQNet");
        n->l[i] = ts::ConvGeom{B, ih, iw, ic, ks[i], ks[i], st[i], (ih - ks[i]) / st[i] + 1, (iw - ks[i]) / st[i] + 1, oc[i]};
        ic = oc[i]; ih = n->l[i].OH; iw = n->l[i].OW;
    }
    n->l[3] = ts::ConvGeom{B, 1, 1, ic * ih * iw, 1, 1, 1, 1, 1, HIDDEN};
    n->ld = (n_act * n_atoms + 31) / 32 * 32;
    n->l[4] = ts::ConvGeom{B, 1, 1, HIDDEN, 1, 1, 1, 1, 1, n->ld};
    n->n_act = n_act;
    n->n_atoms = n_atoms;
    int64_t o = 0;
    for (int i = 0; i < 5; ++i) { n->off[i] = o; o += n->l[i].param_elems(); }
    n->off[5] = o;
    return TS_OK;
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Acts { float* h[5]; float* split; };

size_t split_floats(const Net& n) {
    size_t s = 4;
    for (int i = 0; i < 5; ++i) {
        const int ns = ts::conv_fwd_splits(n.l[i]);
        if (ns > 1) s = std::max(s, (size_t)ns * n.l[i].out_elems());
    }
    return s;
}

size_t acts_bytes(const Net& n) {
    size_t s = al(4 * split_floats(n));
    for (int i = 0; i < 5; ++i) s += al(4 * (size_t)n.l[i].out_elems());
    return s;
}

char* carve_acts(const Net& n, char* p, Acts* a) {
    for (int i = 0; i < 5; ++i) { a->h[i] = reinterpret_cast<float*>(p); p += al(4 * (size_t)n.l[i].out_elems()); }
    a->split = reinterpret_cast<float*>(p);
    return p + al(4 * split_floats(n));
}

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

__device__ __forceinline__ float wave_max(float s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s = fmaxf(s, __shfl_xor(s, off, 64));
    return s;
}

// ---- head: one wave per sample.  QR (kind 0): Q[b, a] = mean_j x[b, a, j].  C51 (kind 1): x[b, a, :] is replaced by
// softmax(x[b, a, :]) in place and Q[b, a] = sum_j p_j support_j.  act = first maximum of Q (torch.argmax, dqn.py:141).
__global__ __launch_bounds__(256) void distq_head_kernel(int kind, float* __restrict__ x, const float* __restrict__ support,
                                                         int64_t B, int A, int N, int ld, float* __restrict__ q_out,
                                                         int64_t* __restrict__ act_out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float best = 0.f;
    int best_a = 0;
    for (int a = 0; a < A; ++a) {
        float* row = x + b * ld + a * N;
        float q;
        if (kind == 0) {
            float s = 0.f;
            for (int j = lane; j < N; j += 64) s += row[j];
            q = wave_sum(s) / (float)N;
        } else {
            float m = -INFINITY;
            for (int j = lane; j < N; j += 64) m = fmaxf(m, row[j]);
            m = wave_max(m);
            float s = 0.f;
            for (int j = lane; j < N; j += 64) s += expf(row[j] - m);
            s = wave_sum(s);
            float qs = 0.f;
            for (int j = lane; j < N; j += 64) {
                const float p = expf(row[j] - m) / s;
                row[j] = p;
                qs += p * support[j];
            }
            q = wave_sum(qs);
        }
        if (q_out && lane == 0) q_out[b * A + a] = q;
        if (a == 0 || q > best) { best = q; best_a = a; }
    }
    if (act_out && lane == 0) act_out[b] = best_a;
}

// out[b, :] = dist[b, act[b], :]   (act == nullptr: out[b, k] = dist[b, k], k < N -- unpadding copy)
__global__ __launch_bounds__(256) void distq_select_kernel(const float* __restrict__ dist, const int64_t* __restrict__ act,
                                                           int64_t B, int N, int ld, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const int64_t b = i / N;
    const int j = (int)(i - b * N);
    out[i] = dist[b * ld + (act ? act[b] * N : 0) + j];
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {      // all threads get the sum (fixed order)
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- QRDQN loss (qrdqn.py:111-128), one workgroup per sample:
//   theta_i = x[b, act_b, i], T_j = returns[b, j], d_ij = T_j - theta_i
//   l_ij = smooth_l1(d_ij), w_ij = |tau_hat_i - 1{d_ij <= 0}|
//   huber_b = (1/N) sum_i sum_j l_ij w_ij,   prio_b = (1/N) sum_i sum_j l_ij,   loss = mean_b(huber_b weight_b)
//   d loss / d theta_i = -(weight_b / (B N)) sum_j w_ij clamp(d_ij, -1, 1);  all other head outputs get 0.
__global__ __launch_bounds__(256) void qr_loss_kernel(const float* __restrict__ x, const int64_t* __restrict__ act,
                                                      const float* __restrict__ ret, const float* __restrict__ weight,
                                                      const float* __restrict__ tau_hat, int64_t B, int N, int ld,
                                                      float* __restrict__ d_head, float* __restrict__ prio,
                                                      float* __restrict__ lw) {
    __shared__ float th[MAX_ATOMS], T[MAX_ATOMS], red[4];
    const int64_t b = blockIdx.x;
    const int a = (int)act[b];
    for (int j = threadIdx.x; j < N; j += 256) {
        th[j] = x[b * ld + a * N + j];
        T[j] = ret[b * N + j];
    }
    __syncthreads();
    const float wb = weight ? weight[b] : 1.f;
    const float scale = wb / ((float)B * (float)N);
    float* drow = d_head + b * ld;
    for (int k = threadIdx.x; k < ld; k += 256)
        if (k / N != a) drow[k] = 0.f;
    float wl = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float theta = th[i], tau = tau_hat[i];
        float li = 0.f, ai = 0.f, g = 0.f;
        for (int j = 0; j < N; ++j) {
            const float d = T[j] - theta, ad = fabsf(d);
            const bool quad = ad < 1.f;
            const float l = quad ? 0.5f * d * d : ad - 0.5f;
            const float w = fabsf(tau - (d <= 0.f ? 1.f : 0.f));
            li += l * w;
            ai += l;
            g += w * (quad ? d : (d > 0.f ? 1.f : -1.f));
        }
        drow[a * N + i] = -g * scale;
        wl += li;
        sl += ai;
    }
    wl = block_sum_256(wl, red);
    sl = block_sum_256(sl, red);
    if (threadIdx.x == 0) {
        const float huber = wl / (float)N;
        prio[b] = sl / (float)N;
        lw[b] = huber * wb;
    }
}

// ---- C51 loss (c51.py:133-158), one workgroup per sample; p = softmax probabilities of the taken action:
//   m_i = sum_j clamp(1 - |clamp(returns[b, j], v_min, v_max) - z_i| / delta_z, 0, 1) next_dist[b, j]
//   ce_b = -sum_i m_i log(p_i + 1e-8),  prio_b = ce_b,  loss = mean_b(ce_b weight_b)
//   d loss / d logit_k = p_k (g_k - sum_i p_i g_i),  g_i = -(weight_b / B) m_i / (p_i + 1e-8)
__global__ __launch_bounds__(256) void c51_loss_kernel(const float* __restrict__ p_all, const int64_t* __restrict__ act,
                                                       const float* __restrict__ ret, const float* __restrict__ next_dist,
                                                       const float* __restrict__ weight, const float* __restrict__ support,
                                                       float v_min, float v_max, float delta_z, int64_t B, int N, int ld,
                                                       float* __restrict__ d_head, float* __restrict__ prio,
                                                       float* __restrict__ lw, float* __restrict__ target_out) {
    __shared__ float ts_[MAX_ATOMS], nd[MAX_ATOMS], red[4];
    const int64_t b = blockIdx.x;
    const int a = (int)act[b];
    for (int j = threadIdx.x; j < N; j += 256) {
        ts_[j] = fminf(fmaxf(ret[b * N + j], v_min), v_max);
        nd[j] = next_dist[b * N + j];
    }
    __syncthreads();
    const float wb = weight ? weight[b] : 1.f;
    const float scale = wb / (float)B;
    float* drow = d_head + b * ld;
    for (int k = threadIdx.x; k < ld; k += 256)
        if (k / N != a) drow[k] = 0.f;
    const int i = threadIdx.x;
    float p = 0.f, g = 0.f, ce = 0.f;
    if (i < N) {
        const float z = support[i];
        float m = 0.f;
        for (int j = 0; j < N; ++j) m += fminf(fmaxf(1.f - fabsf(ts_[j] - z) / delta_z, 0.f), 1.f) * nd[j];
        if (target_out) target_out[b * N + i] = m;
        p = p_all[b * ld + a * N + i];
        ce = -(m * logf(p + 1e-8f));
        g = -(m / (p + 1e-8f)) * scale;
    }
    const float s = block_sum_256(p * g, red);
    const float ce_b = block_sum_256(ce, red);
    if (i < N) drow[a * N + i] = p * (g - s);
    if (threadIdx.x == 0) {
        prio[b] = ce_b;
        lw[b] = ce_b * wb;
    }
}

// loss = mean_b lw[b] (fixed order)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ v, int64_t B, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) s += v[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] / (float)B;
}

int net_forward(hipStream_t s, ts_workspace* ws, const Net& n, const float* params, const void* obs, bool obs_u8,
                const Acts& a) {
    const float* x = static_cast<const float*>(obs);
    for (int i = 0; i < 5; ++i) {
        if (int rc = ts::conv_forward(s, n.l[i], x, params + n.off[i], a.h[i], i < 4, a.split, ws, i == 0 && obs_u8))
            return rc;
        x = a.h[i];
    }
    return TS_OK;
}

int check_kind(int kind, const float* aux, const char* who) {
    TS_REQUIRE(kind == TS_DISTQ_QR || kind == TS_DISTQ_C51, TS_ERR_INVALID_ARG, "%s: kind must be TS_DISTQ_QR or TS_DISTQ_C51", who);
    TS_REQUIRE(kind == TS_DISTQ_QR || aux, TS_ERR_INVALID_ARG, "%s: C51 needs the support vector", who);
    return TS_OK;
}

// ================================================================================================================
// Rainbow (modelfree/rainbow.py on RainbowNet, env/atari/atari_network.py:154-208): C51 with NoisyLinear layers
// (utils/net/discrete.py:317-374) and dueling heads.
//   flat parameters: conv1 | conv2 | conv3 | Q0.mu [F+1, 512] | Q0.sigma | Q2.mu [513, ldq] | Q2.sigma | V0.mu | V0.sigma |
//                    V2.mu [513, ldv] | V2.sigma         (F = 64 * OH3 * OW3 in (h, w, c) order, ldq / ldv = n_act * n_atoms /
//                    n_atoms rounded up to 32; last row of a block = bias; padding zero)
//   noise of one network: Q0.eps_p [F] | Q0.eps_q [512] | Q2.eps_p [512] | Q2.eps_q [ldq] | V0.eps_p [F] | V0.eps_q [512] |
//                    V2.eps_p [512] | V2.eps_q [ldv];   NULL = eval mode (mu only)
// ================================================================================================================
struct RNet {
    ts::ConvGeom conv[3];
    ts::ConvGeom lin[4];         // Q0, Q2, V0, V2
    int64_t off_conv[3];
    int64_t off_lin[4];          // mu block; the sigma block follows it
    int64_t total;
    int64_t off_noise[8];
    int64_t noise_total;
    int F, n_act, n_atoms, ldq, ldv;
};

int make_rnet(int B, int c, int h, int w, int n_act, int n_atoms, RNet* n) {
    Net base;
    if (int rc = make_net(B, c, h, w, n_act, n_atoms, &base)) return rc;
    for (int i = 0; i < 3; ++i) { n->conv[i] = base.l[i]; n->off_conv[i] = base.off[i]; }
    n->F = base.l[3].IC;
    n->n_act = n_act; n->n_atoms = n_atoms;
    n->ldq = base.ld;
    n->ldv = (n_atoms + 31) / 32 * 32;
    const int dims[4][2] = {{n->F, HIDDEN}, {HIDDEN, n->ldq}, {n->F, HIDDEN}, {HIDDEN, n->ldv}};
    int64_t o = base.off[3], q = 0;
    for (int i = 0; i < 4; ++i) {
        n->lin[i] = ts::ConvGeom{B, 1, 1, dims[i][0], 1, 1, 1, 1, 1, dims[i][1]};
        n->off_lin[i] = o;
        o += 2 * n->lin[i].param_elems();
        n->off_noise[2 * i] = q; q += dims[i][0];
        n->off_noise[2 * i + 1] = q; q += dims[i][1];
    }
    n->total = o;
    n->noise_total = q;
    return TS_OK;
}

// NoisyLinear.forward's weights (discrete.py:366-374): eff[k, o] = mu + sigma * (eps_q[o] * eps_p[k]); bias row: mu + sigma * eps_q
__global__ __launch_bounds__(256) void noisy_eff_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                        const float* __restrict__ eps_p, const float* __restrict__ eps_q, int K,
                                                        int OC, float* __restrict__ eff) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)(K + 1) * OC) return;
    const int k = (int)(i / OC), o = (int)(i - (int64_t)k * OC);
    float v = mu[i];
    if (eps_p) v += sigma[i] * (k < K ? eps_q[o] * eps_p[k] : eps_q[o]);
    eff[i] = v;
}

// d mu = d eff; d sigma = d eff * (eps_q x eps_p) (bias row: * eps_q)
__global__ __launch_bounds__(256) void noisy_grad_kernel(const float* __restrict__ g_eff, const float* __restrict__ eps_p,
                                                         const float* __restrict__ eps_q, int K, int OC, float* __restrict__ g_mu,
                                                         float* __restrict__ g_sigma) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)(K + 1) * OC) return;
    const int k = (int)(i / OC), o = (int)(i - (int64_t)k * OC);
    const float g = g_eff[i];
    g_mu[i] = g;
    g_sigma[i] = g * (k < K ? eps_q[o] * eps_p[k] : eps_q[o]);
}

// logits[b, a, j] = q[b, a, j] - mean_a q[b, ., j] + v[b, j]  (atari_network.py:203), in place in q
__global__ __launch_bounds__(256) void dueling_kernel(float* __restrict__ q, const float* __restrict__ v, int64_t B, int A, int N,
                                                      int ldq, int ldv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const int64_t b = i / N;
    const int j = (int)(i - b * N);
    float* row = q + b * ldq;
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += row[a * N + j];
    const float shift = v[b * ldv + j] - s / (float)A;
    for (int a = 0; a < A; ++a) row[a * N + j] += shift;
}

// backward of dueling_kernel: dq = dl - mean_a dl (in place), dv = sum_a dl; padding columns of dv zeroed
__global__ __launch_bounds__(256) void dueling_bwd_kernel(float* __restrict__ dl, float* __restrict__ dv, int64_t B, int A, int N,
                                                          int ldq, int ldv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * ldv) return;
    const int64_t b = i / ldv;
    const int j = (int)(i - b * ldv);
    if (j >= N) { dv[i] = 0.f; return; }
    float* row = dl + b * ldq;
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += row[a * N + j];
    dv[i] = s;
    const float m = s / (float)A;
    for (int a = 0; a < A; ++a) row[a * N + j] -= m;
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

struct RActs {
    float* eff[4];               // effective weight matrices of the four noisy layers
    float* c[3];                 // conv activations; c[2] viewed as [B, F] = the features
    float* hq; float* hv;        // hidden activations of the two branches
    float* q; float* v;          // Q2 / V2 outputs; q becomes logits, then probabilities
    float* split;
};

size_t r_split_floats(const RNet& n) {
    size_t s = 4;
    for (int i = 0; i < 3; ++i) { const int ns = ts::conv_fwd_splits(n.conv[i]); if (ns > 1) s = std::max(s, (size_t)ns * n.conv[i].out_elems()); }
    for (int i = 0; i < 4; ++i) { const int ns = ts::conv_fwd_splits(n.lin[i]); if (ns > 1) s = std::max(s, (size_t)ns * n.lin[i].out_elems()); }
    return s;
}

size_t r_acts_bytes(const RNet& n) {
    size_t s = al(4 * r_split_floats(n));
    for (int i = 0; i < 4; ++i) s += al(4 * (size_t)n.lin[i].param_elems()) + al(4 * (size_t)n.lin[i].out_elems());
    for (int i = 0; i < 3; ++i) s += al(4 * (size_t)n.conv[i].out_elems());
    return s;
}

char* r_carve(const RNet& n, char* p, RActs* a) {
    auto take = [&](size_t floats) { float* r = reinterpret_cast<float*>(p); p += al(4 * floats); return r; };
    for (int i = 0; i < 4; ++i) a->eff[i] = take((size_t)n.lin[i].param_elems());
    for (int i = 0; i < 3; ++i) a->c[i] = take((size_t)n.conv[i].out_elems());
    a->hq = take((size_t)n.lin[0].out_elems()); a->q = take((size_t)n.lin[1].out_elems());
    a->hv = take((size_t)n.lin[2].out_elems()); a->v = take((size_t)n.lin[3].out_elems());
    a->split = take(r_split_floats(n));
    return p;
}

// One pass of RainbowNet.forward: probabilities in a.q ([B, ldq] rows), optional Q values / greedy action.
int r_forward(hipStream_t s, ts_workspace* ws, const RNet& n, const float* params, const float* noise, const void* obs,
              bool obs_u8, int64_t B, const float* support, const RActs& a, float* q_out, int64_t* act_out) {
    for (int i = 0; i < 4; ++i) {
        const int K = n.lin[i].IC, OC = n.lin[i].OC;
        const float* mu = params + n.off_lin[i];
        hipLaunchKernelGGL(noisy_eff_kernel, dim3((unsigned)ts::ceil_div((int64_t)(K + 1) * OC, 256)), dim3(256), 0, s, mu,
                           mu + n.lin[i].param_elems(), noise ? noise + n.off_noise[2 * i] : (const float*)nullptr,
                           noise ? noise + n.off_noise[2 * i + 1] : (const float*)nullptr, K, OC, a.eff[i]);
    }
    TS_LAUNCH_CHECK();
    const float* x = static_cast<const float*>(obs);
    for (int i = 0; i < 3; ++i) {
        if (int rc = ts::conv_forward(s, n.conv[i], x, params + n.off_conv[i], a.c[i], true, a.split, ws, i == 0 && obs_u8)) return rc;
        x = a.c[i];
    }
    if (int rc = ts::conv_forward(s, n.lin[0], a.c[2], a.eff[0], a.hq, true, a.split, ws)) return rc;
    if (int rc = ts::conv_forward(s, n.lin[1], a.hq, a.eff[1], a.q, false, a.split, ws)) return rc;
    if (int rc = ts::conv_forward(s, n.lin[2], a.c[2], a.eff[2], a.hv, true, a.split, ws)) return rc;
    if (int rc = ts::conv_forward(s, n.lin[3], a.hv, a.eff[3], a.v, false, a.split, ws)) return rc;
    hipLaunchKernelGGL(dueling_kernel, dim3((unsigned)ts::ceil_div(B * n.n_atoms, 256)), dim3(256), 0, s, a.q, a.v, B, n.n_act,
                       n.n_atoms, n.ldq, n.ldv);
    hipLaunchKernelGGL(distq_head_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, TS_DISTQ_C51, a.q, support, B,
                       n.n_act, n.n_atoms, n.ldq, q_out, act_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace

extern "C" {

int ts_rainbow_layout(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, int64_t* h_out20) {
    RNet n;
    if (int rc = make_rnet(1, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    TS_REQUIRE(h_out20, TS_ERR_INVALID_ARG, "ts_rainbow_layout: NULL output");
    h_out20[0] = n.F; h_out20[1] = n.ldq; h_out20[2] = n.ldv; h_out20[3] = n.total; h_out20[4] = n.noise_total;
    for (int i = 0; i < 3; ++i) h_out20[5 + i] = n.off_conv[i];
    for (int i = 0; i < 4; ++i) h_out20[8 + i] = n.off_lin[i];
    for (int i = 0; i < 8; ++i) h_out20[12 + i] = n.off_noise[i];
    return TS_OK;
}

int ts_rainbow_forward(ts_workspace* ws, const float* params, const float* noise, int64_t c, int64_t h, int64_t w,
                       int64_t n_act, int64_t n_atoms, const float* support, const void* obs_nhwc, int obs_u8, int64_t B,
                       float* dist_out, float* q_out, int64_t* act_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rainbow_forward: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_rainbow_forward: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && support && obs_nhwc, TS_ERR_INVALID_ARG, "ts_rainbow_forward: NULL argument");
    RNet n;
    if (int rc = make_rnet((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    if (int rc = ts::ws_reserve(ws, r_acts_bytes(n))) return rc;
    RActs a;
    r_carve(n, static_cast<char*>(ws->base), &a);
    hipStream_t s = ts::as_stream(stream);
    if (int rc = r_forward(s, ws, n, params, noise, obs_nhwc, obs_u8 != 0, B, support, a, q_out, act_out)) return rc;
    if (dist_out) {
        const int row = n.n_act * n.n_atoms;
        hipLaunchKernelGGL(distq_select_kernel, dim3((unsigned)ts::ceil_div(B * row, 256)), dim3(256), 0, s, a.q,
                           (const int64_t*)nullptr, B, row, n.ldq, dist_out);
        TS_LAUNCH_CHECK();
    }
    return TS_OK;
}

int ts_rainbow_next_dist(ts_workspace* ws, const float* params, const float* noise, const float* params_old,
                         const float* noise_old, int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms,
                         const float* support, const void* obs_next_nhwc, int obs_u8, int64_t B, float* out,
                         ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rainbow_next_dist: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_rainbow_next_dist: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && support && obs_next_nhwc && out, TS_ERR_INVALID_ARG, "ts_rainbow_next_dist: NULL argument");
    RNet n;
    if (int rc = make_rnet((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    const size_t one = r_acts_bytes(n);
    if (int rc = ts::ws_reserve(ws, 2 * one + al(8 * (size_t)B))) return rc;
    RActs ao, at;
    char* p = r_carve(n, static_cast<char*>(ws->base), &ao);
    p = r_carve(n, p, &at);
    int64_t* act = reinterpret_cast<int64_t*>(p);
    hipStream_t s = ts::as_stream(stream), side;
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    const bool two = params_old != nullptr;
    if (two) {
        if (int rc = ts::stream_wait(ws, s, side, 9)) return rc;
        if (int rc = r_forward(side, ws, n, params_old, noise_old, obs_next_nhwc, obs_u8 != 0, B, support, at, nullptr, nullptr))
            return rc;
    }
    if (int rc = r_forward(s, ws, n, params, noise, obs_next_nhwc, obs_u8 != 0, B, support, ao, nullptr, act)) return rc;
    if (two)
        if (int rc = ts::stream_wait(ws, side, s, 10)) return rc;
    hipLaunchKernelGGL(distq_select_kernel, dim3((unsigned)ts::ceil_div(B * n.n_atoms, 256)), dim3(256), 0, s,
                       two ? at.q : ao.q, act, B, n.n_atoms, n.ldq, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_rainbow_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, const float* noise,
                      int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, const float* support,
                      const void* obs_nhwc, int obs_u8, const int64_t* act, const float* returns, const float* next_dist,
                      const float* weight, int64_t B, const ts_distq_hparams* hp, float* prio_out, float* loss_out,
                      float* target_dist_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rainbow_update: workspace is NULL");
    TS_REQUIRE(B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG, "ts_rainbow_update: bad batch size / step");
    TS_REQUIRE(params && adam_m && adam_v && noise && support && obs_nhwc && act && returns && next_dist && hp && prio_out &&
                   loss_out && hp->v_max > hp->v_min, TS_ERR_INVALID_ARG, "ts_rainbow_update: bad argument");
    RNet n;
    if (int rc = make_rnet((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    // workspace: pass | dY buffers | wgrad slabs | effective-weight gradient | flat gradient | per-sample terms | norm
    // (one slab set per layer and one effective-weight gradient per noisy layer: their weight gradients run side by side)
    size_t slab_c[3], slab_l[4], geff_l[4], slab_all = 0;
    for (int i = 0; i < 3; ++i) slab_all += slab_c[i] = al(4 * (size_t)ts::conv_wgrad_splits(n.conv[i]) * n.conv[i].param_elems());
    for (int i = 0; i < 4; ++i) {
        slab_all += slab_l[i] = al(4 * (size_t)ts::conv_wgrad_splits(n.lin[i]) * n.lin[i].param_elems());
        slab_all += geff_l[i] = al(4 * (size_t)n.lin[i].param_elems());
    }
    size_t bytes = r_acts_bytes(n) + slab_all + al(4 * (size_t)n.total) + al(4 * (size_t)B) + 4096;
    for (int i = 0; i < 3; ++i) bytes += al(4 * (size_t)n.conv[i].out_elems());
    bytes += 2 * al(4 * (size_t)n.lin[0].out_elems()) + al(4 * (size_t)n.lin[1].out_elems()) + al(4 * (size_t)n.lin[3].out_elems()) +
             al(4 * (size_t)n.conv[2].out_elems());
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    RActs a;
    char* p = r_carve(n, static_cast<char*>(ws->base), &a);
    auto take = [&](size_t floats) { float* r = reinterpret_cast<float*>(p); p += al(4 * floats); return r; };
    float* dc[3];
    for (int i = 0; i < 3; ++i) dc[i] = take((size_t)n.conv[i].out_elems());
    float* dhq = take((size_t)n.lin[0].out_elems());
    float* dhv = take((size_t)n.lin[0].out_elems());
    float* dq = take((size_t)n.lin[1].out_elems());
    float* dv = take((size_t)n.lin[3].out_elems());
    float* dfeat2 = take((size_t)n.conv[2].out_elems());
    float *slabs_c[3], *slabs_l[4], *g_eff[4];
    for (int i = 0; i < 3; ++i) { slabs_c[i] = reinterpret_cast<float*>(p); p += slab_c[i]; }
    for (int i = 0; i < 4; ++i) { slabs_l[i] = reinterpret_cast<float*>(p); p += slab_l[i]; }
    for (int i = 0; i < 4; ++i) { g_eff[i] = reinterpret_cast<float*>(p); p += geff_l[i]; }
    float* grad = take((size_t)n.total);
    float* lw = take((size_t)B);
    float* norm_part = reinterpret_cast<float*>(p);
    if (grad_out) grad = grad_out;

    if (int rc = r_forward(s, ws, n, params, noise, obs_nhwc, obs_u8 != 0, B, support, a, nullptr, nullptr)) return rc;
    const double dz = (hp->v_max - hp->v_min) / (double)(n.n_atoms - 1);
    hipLaunchKernelGGL(c51_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, a.q, act, returns, next_dist, weight, support,
                       (float)hp->v_min, (float)hp->v_max, (float)dz, B, n.n_atoms, n.ldq, dq, prio_out, lw, target_dist_out);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, lw, B, loss_out);
    TS_LAUNCH_CHECK();
    if (int rc = ts::record_td(ws, s)) return rc;        // prio_out / loss_out are written: ts_dqn_wait_td
    hipLaunchKernelGGL(dueling_bwd_kernel, dim3((unsigned)ts::ceil_div(B * n.ldv, 256)), dim3(256), 0, s, dq, dv, B, n.n_act,
                       n.n_atoms, n.ldq, n.ldv);
    TS_LAUNCH_CHECK();
    // The four noisy layers: (input, upstream gradient, input-gradient target, ReLU mask of the input).  The advantage
    // branch (Q2 -> Q0) sends its input gradients down the caller's stream, the value branch (V2 -> V0) down the workspace's
    // first side stream (the two streams that own input-gradient scratch); the weight gradients of the advantage branch go
    // to the second side stream, those of the value branch follow its input gradients.  Same launches on the same
    // operands as the sequential order: only their placement in time differs.
    const float* xin[4] = {a.c[2], a.hq, a.c[2], a.hv};
    const float* dy[4] = {dhq, dq, dhv, dv};
    float* dxo[4] = {dc[2], dhq, dfeat2, dhv};
    hipStream_t sv, sw;
    if (int rc = ts::side_streams(ws, s, &sv, &sw)) return rc;
    auto noisy_wgrad = [&](hipStream_t st, int i) -> int {
        const int K = n.lin[i].IC, OC = n.lin[i].OC;
        if (int rc = ts::conv_wgrad(st, n.lin[i], xin[i], dy[i], slabs_l[i], ws)) return rc;
        if (int rc = ts::slab_sum(st, slabs_l[i], ts::conv_wgrad_splits(n.lin[i]), n.lin[i].param_elems(), g_eff[i])) return rc;
        float* g_mu = grad + n.off_lin[i];
        hipLaunchKernelGGL(noisy_grad_kernel, dim3((unsigned)ts::ceil_div((int64_t)(K + 1) * OC, 256)), dim3(256), 0, st,
                           g_eff[i], noise + n.off_noise[2 * i], noise + n.off_noise[2 * i + 1], K, OC, g_mu,
                           g_mu + n.lin[i].param_elems());
        TS_LAUNCH_CHECK();
        return TS_OK;
    };
    auto noisy_dgrad = [&](hipStream_t st, int i) -> int {
        return ts::conv_dgrad(st, n.lin[i], dy[i], a.eff[i], xin[i], dxo[i], ws);
    };
    if (int rc = ts::stream_wait(ws, s, sv, 11)) return rc;              // dq, dv
    if (int rc = ts::stream_wait(ws, s, sw, 12)) return rc;
    if (int rc = noisy_dgrad(sv, 3)) return rc;                          // V2: dv -> dhv
    if (int rc = noisy_dgrad(sv, 2)) return rc;                          // V0: dhv -> dfeat2
    if (int rc = noisy_wgrad(sw, 1)) return rc;                          // Q2
    if (int rc = noisy_dgrad(s, 1)) return rc;                           // Q2: dq -> dhq
    if (int rc = ts::stream_wait(ws, s, sw, 13)) return rc;              // dhq
    if (int rc = noisy_wgrad(sw, 0)) return rc;                          // Q0
    if (int rc = noisy_dgrad(s, 0)) return rc;                           // Q0: dhq -> dc[2]
    if (int rc = ts::stream_wait(ws, sv, s, 14)) return rc;              // dfeat2 (and dhv)
    if (int rc = noisy_wgrad(sv, 3)) return rc;                          // V2, V0 behind the value branch's input gradients
    if (int rc = noisy_wgrad(sv, 2)) return rc;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)ts::ceil_div(n.conv[2].out_elems(), 256)), dim3(256), 0, s, dc[2], dfeat2,
                       n.conv[2].out_elems());
    TS_LAUNCH_CHECK();
    {   // conv3, conv2, conv1 (ts::chain_backward joins both side streams into the caller's at its end)
        const float* x[3]; const float* wb[3]; float* g[3];
        for (int i = 0; i < 3; ++i) {
            x[i] = i == 0 ? static_cast<const float*>(obs_nhwc) : a.c[i - 1];
            wb[i] = params + n.off_conv[i];
            g[i] = grad + n.off_conv[i];
        }
        if (int rc = ts::chain_backward(s, ws, 3, n.conv, x, dc, wb, slabs_c, g, obs_u8 != 0)) return rc;
    }
    if (hp->lr < 0.0) return TS_OK;
    return ts::adam_step(s, params, adam_m, adam_v, grad, n.total, adam_step, hp->lr, hp->beta1, hp->beta2, hp->adam_eps,
                         hp->max_grad_norm, norm_part);
}

}  // extern "C"

namespace {

}  // namespace

extern "C" {

int64_t ts_distq_param_count(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms) {
    Net n;
    if (make_net(1, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n) != TS_OK) return -1;
    return n.off[5];
}

int ts_distq_forward(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                     int64_t n_atoms, int kind, const float* aux, const void* obs_nhwc, int obs_u8, int64_t B,
                     float* dist_out, float* q_out, int64_t* act_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_distq_forward: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_distq_forward: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs_nhwc, TS_ERR_INVALID_ARG, "ts_distq_forward: NULL argument");
    if (int rc = check_kind(kind, aux, "ts_distq_forward")) return rc;
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    if (int rc = ts::ws_reserve(ws, acts_bytes(n))) return rc;
    Acts a;
    carve_acts(n, static_cast<char*>(ws->base), &a);
    hipStream_t s = ts::as_stream(stream);
    if (int rc = net_forward(s, ws, n, params, obs_nhwc, obs_u8 != 0, a)) return rc;
    hipLaunchKernelGGL(distq_head_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, kind, a.h[4], aux, B,
                       n.n_act, n.n_atoms, n.ld, q_out, act_out);
    TS_LAUNCH_CHECK();
    if (dist_out) {
        const int row = n.n_act * n.n_atoms;
        hipLaunchKernelGGL(distq_select_kernel, dim3((unsigned)ts::ceil_div(B * row, 256)), dim3(256), 0, s, a.h[4],
                           (const int64_t*)nullptr, B, row, n.ld, dist_out);
        TS_LAUNCH_CHECK();
    }
    return TS_OK;
}

int ts_distq_next_dist(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                       int64_t w, int64_t n_act, int64_t n_atoms, int kind, const float* aux,
                       const void* obs_next_nhwc, int obs_u8, int64_t B, float* out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_distq_next_dist: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_distq_next_dist: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs_next_nhwc && out, TS_ERR_INVALID_ARG, "ts_distq_next_dist: NULL argument");
    if (int rc = check_kind(kind, aux, "ts_distq_next_dist")) return rc;
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    const size_t one = acts_bytes(n);
    if (int rc = ts::ws_reserve(ws, 2 * one + al(8 * (size_t)B))) return rc;
    Acts ao, at;
    char* p = carve_acts(n, static_cast<char*>(ws->base), &ao);
    p = carve_acts(n, p, &at);
    int64_t* act = reinterpret_cast<int64_t*>(p);
    hipStream_t s = ts::as_stream(stream), side;
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    const bool two = params_old != nullptr;
    if (two) {          // the lagged net's pass runs beside the online net's
        if (int rc = ts::stream_wait(ws, s, side, 9)) return rc;
        if (int rc = net_forward(side, ws, n, params_old, obs_next_nhwc, obs_u8 != 0, at)) return rc;
        if (kind == TS_DISTQ_C51) {
            hipLaunchKernelGGL(distq_head_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, side, kind, at.h[4],
                               aux, B, n.n_act, n.n_atoms, n.ld, (float*)nullptr, (int64_t*)nullptr);
            TS_LAUNCH_CHECK();
        }
    }
    if (int rc = net_forward(s, ws, n, params, obs_next_nhwc, obs_u8 != 0, ao)) return rc;
    hipLaunchKernelGGL(distq_head_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, kind, ao.h[4], aux, B,
                       n.n_act, n.n_atoms, n.ld, (float*)nullptr, act);
    TS_LAUNCH_CHECK();
    if (two)
        if (int rc = ts::stream_wait(ws, side, s, 10)) return rc;
    hipLaunchKernelGGL(distq_select_kernel, dim3((unsigned)ts::ceil_div(B * n.n_atoms, 256)), dim3(256), 0, s,
                       two ? at.h[4] : ao.h[4], act, B, n.n_atoms, n.ld, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_distq_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                    int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, int kind, const float* aux,
                    const void* obs_nhwc, int obs_u8, const int64_t* act, const float* returns, const float* next_dist,
                    const float* weight, int64_t B, const ts_distq_hparams* hp, float* prio_out, float* loss_out,
                    float* target_dist_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_distq_update: workspace is NULL");
    TS_REQUIRE(B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG, "ts_distq_update: bad batch size / step");
    TS_REQUIRE(params && adam_m && adam_v && obs_nhwc && act && returns && hp && prio_out && loss_out && aux,
               TS_ERR_INVALID_ARG, "ts_distq_update: NULL argument");
    if (int rc = check_kind(kind, aux, "ts_distq_update")) return rc;
    TS_REQUIRE(kind == TS_DISTQ_QR || (next_dist && hp->v_max > hp->v_min), TS_ERR_INVALID_ARG,
               "ts_distq_update: C51 needs next_dist and v_min < v_max");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, (int)n_atoms, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);

    // workspace: activations | dY of every layer | wgrad slabs | flat gradient | per-sample loss terms | norm partials
    size_t slab[5], slab_all = 0;
    for (int i = 0; i < 5; ++i) {          // one slab set per layer: the weight gradients run side by side (ts::chain_backward)
        slab[i] = al(4 * (size_t)ts::conv_wgrad_splits(n.l[i]) * n.l[i].param_elems());
        slab_all += slab[i];
    }
    size_t bytes = acts_bytes(n) + slab_all + al(4 * (size_t)n.off[5]) + al(4 * (size_t)B) + 4096;
    for (int i = 0; i < 5; ++i) bytes += al(4 * (size_t)n.l[i].out_elems());
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Acts a;
    char* p = carve_acts(n, static_cast<char*>(ws->base), &a);
    float* dy[5];
    for (int i = 0; i < 5; ++i) { dy[i] = reinterpret_cast<float*>(p); p += al(4 * (size_t)n.l[i].out_elems()); }
    float* slabs[5];
    for (int i = 0; i < 5; ++i) { slabs[i] = reinterpret_cast<float*>(p); p += slab[i]; }
    float* grad = reinterpret_cast<float*>(p); p += al(4 * (size_t)n.off[5]);
    float* lw = reinterpret_cast<float*>(p); p += al(4 * (size_t)B);
    float* norm_part = reinterpret_cast<float*>(p);
    if (grad_out) grad = grad_out;

    if (int rc = net_forward(s, ws, n, params, obs_nhwc, obs_u8 != 0, a)) return rc;
    if (kind == TS_DISTQ_QR) {
        hipLaunchKernelGGL(qr_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, a.h[4], act, returns, weight, aux, B,
                           n.n_atoms, n.ld, dy[4], prio_out, lw);
    } else {
        hipLaunchKernelGGL(distq_head_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, kind, a.h[4], aux, B,
                           n.n_act, n.n_atoms, n.ld, (float*)nullptr, (int64_t*)nullptr);
        const double dz = (hp->v_max - hp->v_min) / (double)(n.n_atoms - 1);
        hipLaunchKernelGGL(c51_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, a.h[4], act, returns, next_dist, weight,
                           aux, (float)hp->v_min, (float)hp->v_max, (float)dz, B, n.n_atoms, n.ld, dy[4], prio_out,
                           lw, target_dist_out);
    }
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, lw, B, loss_out);
    TS_LAUNCH_CHECK();
    if (int rc = ts::record_td(ws, s)) return rc;        // prio_out / loss_out are written: ts_dqn_wait_td

    // head, fc1, conv3, conv2, conv1: input gradients down the caller's stream, the weight gradients beside them on the
    // workspace's side streams (ts::chain_backward, as ts_dqn_update)
    {
        const float* x[5]; const float* wb[5]; float* g[5];
        for (int i = 0; i < 5; ++i) {
            x[i] = i == 0 ? static_cast<const float*>(obs_nhwc) : a.h[i - 1];
            wb[i] = params + n.off[i];
            g[i] = grad + n.off[i];
        }
        if (int rc = ts::chain_backward(s, ws, 5, n.l, x, dy, wb, slabs, g, obs_u8 != 0)) return rc;
    }
    if (hp->lr < 0.0) return TS_OK;
    return ts::adam_step(s, params, adam_m, adam_v, grad, n.off[5], adam_step, hp->lr, hp->beta1, hp->beta2,
                         hp->adam_eps, hp->max_grad_norm, norm_part);
}

}  // extern "C"
