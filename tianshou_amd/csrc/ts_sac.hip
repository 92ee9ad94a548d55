// ts_sac.hip -- the SAC learn() step (tanh-Gaussian actor, twin critics, auto alpha, Polyak) for gfx950.
//
// Replaces, on device-resident float32 batches:
//   SACPolicy.forward                    tianshou/algorithm/modelfree/sac.py:108-131
//   correct_log_prob_gaussian_tanh       sac.py:25-39
//   SAC._target_q_compute_value          sac.py:290-296, td3.py:94-102
//   _minimize_critic_squared_loss        ddpg.py:267-285
//   SAC._update_with_batch               sac.py:298-336 (three optimizer steps, AutoAlpha.update :203-209,
//                                        polyak_parameter_update utils/lagged_network.py:8-18)
//   nets: ContinuousActorProbabilistic (conditioned sigma, unbounded) continuous.py:220-238,
//         ContinuousCritic (concat) continuous.py:144-169, Net/MLP ReLU common.py:90-178
// All Linear layers run on the fp32-MFMA implicit-GEMM kernels of ts_conv.hip (1x1 case); input widths are
// zero-padded to a multiple of 32 and the narrow heads to 32-column blocks (the padding stays exactly zero
// under Adam: zero inputs / zero upstream gradients give zero weight gradients).
// Roofline: fp32 MFMA for the GEMMs; the sampling / loss kernels are launch-latency sized ([B, act_dim]).
#include <algorithm>
#include <cmath>

#include "ts_common.h"
#include "ts_conv.h"
#include "ts_mlp.h"

#pragma clang fp contract(off)   // the elementwise formulas follow torch's operation order

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
int adam_step_multi(hipStream_t s, int nvec, float* const* params, float* const* m, float* const* v,
                    const float* const* grad, float* const* lagged, int64_t n, int64_t step, double lr, double beta1,
                    double beta2, double eps, double tau);
}

namespace {

constexpr int HID = 256;
constexpr int SIG_COL = 32;          // sigma block of the actor head starts at column 32
constexpr float SIGMA_MIN = -20.f, SIGMA_MAX = 2.f;
constexpr float TANH_EPS = 1.1920928955078125e-07f;     // np.finfo(np.float32).eps
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

inline int pad32(int x) { return (x + 31) / 32 * 32; }

constexpr int MAXD = TS_MLP_MAX_HIDDEN_LAYERS;      // hidden layers of a trunk (ts_mlp_set_trunk)
constexpr int MAXL = MAXD + 1;                      // linear layers incl. the head

struct Mlp {                  // in -> hid x depth -> head, ReLU between (depth 2 / width 256: the examples' nets)
    ts::ConvGeom l[MAXL];
    int64_t off[MAXL + 1];
    int L;                    // linear layers = depth + 1; the head is l[L - 1]
    int act;                  // TS_NET_ACT_RELU (Net's default, the examples) or TS_NET_ACT_TANH after every hidden layer
    int64_t total() const { return off[L]; }
    const ts::ConvGeom& head() const { return l[L - 1]; }
    // the one-launch three-layer kernels of ts_mlp.hip are written for two ReLU hidden layers of one width
    bool three() const { return L == 3 && l[1].OC == l[0].OC && act == TS_NET_ACT_RELU; }
};

Mlp make_mlp(int B, int in_pad, int head_cols, int hid = HID, int depth = 2, int act = TS_NET_ACT_RELU) {
    Mlp m{};
    if (depth < 1 || depth > MAXD) depth = 2;       // (validated by ts_mlp_set_trunk / the layout entry points)
    m.L = depth + 1;
    m.act = act == TS_NET_ACT_TANH ? TS_NET_ACT_TANH : TS_NET_ACT_RELU;
    int64_t o = 0;
    for (int i = 0; i < m.L; ++i) {
        m.l[i] = ts::ConvGeom{B, 1, 1, i == 0 ? in_pad : hid, 1, 1, 1, 1, 1, i + 1 == m.L ? head_cols : hid};
        m.off[i] = o;
        o += m.l[i].param_elems();
    }
    m.off[m.L] = o;
    return m;
}

__global__ __launch_bounds__(256) void mlp_tanh_kernel(float* __restrict__ h, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = tanhf(h[i]);
}
// dh *= 1 - h^2   (backward through tanh; h = the layer's output)
__global__ __launch_bounds__(256) void mlp_tanh_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dh[i] = dh[i] * (1.f - h[i] * h[i]);
}

struct Act { float* h[MAXD]; float* out; };      // forward activations of one pass: h[i] = output of hidden layer i
inline Act act_of(float* h1, float* h2, float* out) { Act a{}; a.h[0] = h1; a.h[1] = h2; a.out = out; return a; }
inline float* act_out(const Mlp& m, const Act& a, int i) { return i + 1 == m.L ? a.out : a.h[i]; }      // output of layer i

int mlp_forward(hipStream_t s, ts_workspace* ws, const Mlp& m, const float* p, const float* x, const Act& a,
                float* split) {
    if (m.three() && ts::mlp3_supported(m.l[0].IC, m.l[0].OC, m.l[2].OC))       // one launch (ts_mlp.hip)
        return ts::mlp3_forward(s, x, m.l[0].B, m.l[0].IC, p + m.off[0], p + m.off[1], p + m.off[2], m.l[2].OC, a.h[0],
                                a.h[1], a.out, ws);
    const float* in = x;
    for (int i = 0; i < m.L; ++i) {
        float* out = act_out(m, a, i);
        const bool hidden = i + 1 < m.L;
        if (int rc = ts::conv_forward(s, m.l[i], in, p + m.off[i], out, hidden && m.act == TS_NET_ACT_RELU, split, ws)) return rc;
        if (hidden && m.act == TS_NET_ACT_TANH) {
            const int64_t cnt = (int64_t)m.l[i].B * m.l[i].OC;
            hipLaunchKernelGGL(mlp_tanh_kernel, dim3((unsigned)ts::ceil_div(cnt, 256)), dim3(256), 0, s, out, cnt);
            TS_LAUNCH_CHECK();
        }
        in = out;
    }
    return TS_OK;
}

// n <= MLP3_MAX_NETS networks of one shape on the same input and the same stream: one launch (blockIdx.y = network) on the
// fused path, n calls otherwise.
constexpr int MULTI_MAX = ts::MLP3_MAX_NETS;
int mlp_forward_multi(hipStream_t s, ts_workspace* ws, const Mlp& m, int n, const float* const* p, const float* x, const Act* a,
                      float* const* split) {
    if (n > 1 && n <= MULTI_MAX && m.three() && ts::mlp3_supported(m.l[0].IC, m.l[0].OC, m.l[2].OC)) {
        const float *w1[MULTI_MAX], *w2[MULTI_MAX], *w3[MULTI_MAX];
        float *h1[MULTI_MAX], *h2[MULTI_MAX], *out[MULTI_MAX];
        for (int k = 0; k < n; ++k) {
            w1[k] = p[k] + m.off[0]; w2[k] = p[k] + m.off[1]; w3[k] = p[k] + m.off[2];
            h1[k] = a[k].h[0]; h2[k] = a[k].h[1]; out[k] = a[k].out;
        }
        return ts::mlp3_forward_n(s, n, x, m.l[0].B, m.l[0].IC, w1, w2, w3, m.l[2].OC, h1, h2, out, ws);
    }
    for (int k = 0; k < n; ++k)
        if (int rc = mlp_forward(s, ws, m, p[k], x, a[k], split[k & 1])) return rc;
    return TS_OK;
}
int mlp_forward_twin(hipStream_t s, ts_workspace* ws, const Mlp& m, const float* const* p, const float* x, const Act* a,
                     float* const* split) {
    return mlp_forward_multi(s, ws, m, 2, p, x, a, split);
}

size_t split_floats(const Mlp& m) {
    size_t s = 4;
    for (int i = 0; i < m.L; ++i) {
        const int ns = ts::conv_fwd_splits(m.l[i]);
        if (ns > 1) s = std::max(s, (size_t)ns * m.l[i].out_elems());
    }
    return s;
}

struct BwdScratch { float* dh[MAXD]; float* slabs; };        // dh[i] = gradient w.r.t. the output of hidden layer i

bool fused_backward(const Mlp& m, bool want_dx, int col0, int col1) {
    return m.three() && ts::mlp3_backward_supported(m.l[0].IC, m.l[0].OC, m.l[2].OC, want_dx, col0, col1);
}

// Weight gradients of n <= 5 networks of the same shape whose input-gradient chains (mlp3_backward) have run on stream
// s: all 3 n GEMMs in one launch, all 3 n slab sets summed in one launch.
constexpr int WGRADS_MAX_NETS = 5;
int mlp_weight_grads(hipStream_t s, ts_workspace* ws, int n, const Mlp& m, const float* const* x, const Act* a,
                     const float* const* d_out, float* const* grad, const BwdScratch* sc) {
    ts::ConvGeom geoms[3 * WGRADS_MAX_NETS];
    const float* X[3 * WGRADS_MAX_NETS];
    const float* dY[3 * WGRADS_MAX_NETS];
    float* slabs[3 * WGRADS_MAX_NETS];
    ts::SlabSeg seg[3 * WGRADS_MAX_NETS];
    TS_REQUIRE(n >= 1 && n <= WGRADS_MAX_NETS && m.L == 3, TS_ERR_INVALID_ARG, "mlp_weight_grads: 1 .. %d three-layer networks",
               WGRADS_MAX_NETS);
    for (int k = 0; k < n; ++k) {
        const float* xin[3] = {x[k], a[k].h[0], a[k].h[1]};
        const float* dy[3] = {sc[k].dh[0], sc[k].dh[1], d_out[k]};
        size_t off = 0;
        for (int i = 2; i >= 0; --i) {
            const int j = 3 * k + (2 - i);
            const int ns = ts::conv_wgrad_splits(m.l[i]);
            geoms[j] = m.l[i]; X[j] = xin[i]; dY[j] = dy[i]; slabs[j] = sc[k].slabs + off;
            seg[j] = ts::SlabSeg{sc[k].slabs + off, ns, m.l[i].param_elems(), grad[k] + m.off[i]};
            off += (size_t)ns * m.l[i].param_elems();
        }
    }
    if (int rc = ts::conv_wgrad_group(s, 3 * n, geoms, X, dY, slabs, ws)) return rc;
    return ts::slab_sum_multi(s, seg, 3 * n);
}

// d_out = d loss / d head output.  grad (nullable) receives the flat parameter gradient; dx (nullable) the
// gradient w.r.t. the input columns [col0, col1).  `part`: 1 = the input-gradient chain, 2 = the weight gradients,
// 3 = both -- callers that run two networks on two streams enqueue part 1 of both before part 2 of either, so that
// the second stream has work while the host is still submitting the first one's GEMMs (on the per-layer path the two
// are interleaved and everything happens in part 1).
int mlp_backward(hipStream_t s, ts_workspace* ws, const Mlp& m, const float* p, const float* x, const Act& a,
                 const float* d_out, float* grad, float* dx, int col0, int col1, const BwdScratch& sc, int part = 3) {
    if (fused_backward(m, dx != nullptr, col0, col1)) {
        // all input gradients in one launch (ts_mlp.hip), then the three weight-gradient GEMMs in one launch
        if (part & 1)
            if (int rc = ts::mlp3_backward(s, d_out, m.l[0].B, m.l[0].IC, p + m.off[0], p + m.off[1], p + m.off[2],
                                           m.l[2].OC, a.h[0], a.h[1], sc.dh[0], sc.dh[1], dx, col0, col1, ws))
                return rc;
        if (!grad || !(part & 2)) return TS_OK;
        return mlp_weight_grads(s, ws, 1, m, &x, &a, &d_out, &grad, &sc);
    }
    if (!(part & 1)) return TS_OK;
    for (int i = m.L - 1; i >= 0; --i) {            // layer i: input xin, upstream gradient dy
        const float* xin = i == 0 ? x : a.h[i - 1];
        const float* dy = i + 1 == m.L ? d_out : sc.dh[i];
        if (grad) {
            if (int rc = ts::conv_wgrad(s, m.l[i], xin, dy, sc.slabs, ws)) return rc;
            if (int rc = ts::slab_sum(s, sc.slabs, ts::conv_wgrad_splits(m.l[i]), m.l[i].param_elems(),
                                      grad + m.off[i]))
                return rc;
        }
        if (i > 0) {                                  // (xin = the ReLU output the gradient is masked with / the tanh output)
            const bool relu = m.act == TS_NET_ACT_RELU;
            if (int rc = ts::conv_dgrad(s, m.l[i], dy, p + m.off[i], relu ? xin : nullptr, sc.dh[i - 1], ws)) return rc;
            if (!relu) {
                const int64_t cnt = (int64_t)m.l[i].B * m.l[i].IC;
                hipLaunchKernelGGL(mlp_tanh_bwd_kernel, dim3((unsigned)ts::ceil_div(cnt, 256)), dim3(256), 0, s, sc.dh[i - 1], xin, cnt);
                TS_LAUNCH_CHECK();
            }
        } else if (dx) {
            if (int rc = ts::conv_dgrad(s, m.l[0], dy, p + m.off[0], nullptr, dx, ws, col0, col1)) return rc;
        }
    }
    return TS_OK;
}

// The input-gradient chains (part 1 of mlp_backward) of n networks on one stream: one launch on the fused path.
int mlp_backward_multi(hipStream_t s, ts_workspace* ws, const Mlp& m, int n, const float* const* p, const float* x, const Act* a,
                       const float* const* d_out, float* const* dx, int col0, int col1, const BwdScratch* sc) {
    const bool want_dx = dx && dx[0];
    if (n > 1 && n <= MULTI_MAX && fused_backward(m, want_dx, col0, col1)) {
        const float *w1[MULTI_MAX], *w2[MULTI_MAX], *w3[MULTI_MAX], *h1[MULTI_MAX], *h2[MULTI_MAX];
        float *dh1[MULTI_MAX], *dh2[MULTI_MAX];
        for (int k = 0; k < n; ++k) {
            w1[k] = p[k] + m.off[0]; w2[k] = p[k] + m.off[1]; w3[k] = p[k] + m.off[2];
            h1[k] = a[k].h[0]; h2[k] = a[k].h[1]; dh1[k] = sc[k].dh[0]; dh2[k] = sc[k].dh[1];
        }
        return ts::mlp3_backward_n(s, n, d_out, m.l[0].B, m.l[0].IC, w1, w2, w3, m.l[2].OC, h1, h2, dh1, dh2, dx, col0, col1, ws);
    }
    for (int k = 0; k < n; ++k)
        if (int rc = mlp_backward(s, ws, m, p[k], x, a[k], d_out[k], nullptr, want_dx ? dx[k] : nullptr, col0, col1, sc[k], 1))
            return rc;
    return TS_OK;
}
int mlp_backward_twin(hipStream_t s, ts_workspace* ws, const Mlp& m, const float* const* p, const float* x, const Act* a,
                      const float* const* d_out, float* const* dx, int col0, int col1, const BwdScratch* sc) {
    return mlp_backward_multi(s, ws, m, 2, p, x, a, d_out, dx, col0, col1, sc);
}

size_t slab_floats(const Mlp& m) {      // the layers' slab sets side by side (one slab_sum_multi launch)
    size_t s = 0;
    for (int i = 0; i < m.L; ++i) s += (size_t)ts::conv_wgrad_splits(m.l[i]) * m.l[i].param_elems();
    return s;
}

// Second stream for the twin critics.  With the one-launch chains of ts_mlp.hip every kernel of a C5-shape critic fills
// the chip and the twin chains gain nothing side by side: measured on one stream SAC 1,820 vs 1,778 and TD3 2,382 vs
// 2,302 updates/s (DDPG equal), without the event record / wait pairs.  SAC / TD3 / DDPG therefore stay on the caller's
// stream when their networks take the fused path; REDQ (ten critics: +17 %) keeps both streams; DiscreteSAC left them
// for the multi-network launches (dsac_one_stream below).
int twin_stream(ts_workspace* ws, hipStream_t s, const Mlp& critic, hipStream_t* out) {
    static const bool force = getenv("TS_TWIN_STREAMS") != nullptr;       // experiments
    if (!force && critic.three() && ts::mlp3_supported(critic.l[0].IC, critic.l[0].OC, critic.l[2].OC)) { *out = s; return TS_OK; }
    return ts::side_stream(ws, s, out);
}

// DiscreteSAC: its three networks share one shape and one input, so on the fused path they go through the multi-network
// launches on the caller's stream (TS_TWIN_STREAMS keeps the two-stream per-network chains for comparison).
bool dsac_one_stream(const Mlp& m) {
    static const bool force = getenv("TS_TWIN_STREAMS") != nullptr;
    return !force && m.three() && ts::mlp3_supported(m.l[0].IC, m.l[0].OC, m.l[2].OC) && fused_backward(m, false, 0, 0);
}

// ---- elementwise kernels ----------------------------------------------------------------------------
// row of a 32-column head gradient: [v, 0 x 31] (the whole row is written: the buffer needs no memset)
__device__ __forceinline__ void store_head_row(float* __restrict__ row, float v) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4* r = reinterpret_cast<f32x4*>(row);
    r[0] = f32x4{v, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 1; q < 8; ++q) r[q] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// x_a[b] = [obs | 0], x_c[b] = [obs | act | 0]; x_p (nullable) = a second copy of x_c's observation columns (its
// action columns are written by the policy kernel).  Four output columns per thread (ka, kc are multiples of 32).
// `rows` (nullable): obs / act are whole replay-buffer columns and sample b is their row rows[b] -- the gather of
// ReplayBuffer.__getitem__ (buffer_base.py:605-649) happens here instead of in launches of its own.
__device__ __forceinline__ void sac_pack_item(int64_t i, const float* __restrict__ obs, const float* __restrict__ act,
                                              int64_t B, int obs_dim, int act_dim, int ka, int kc,
                                              float* __restrict__ x_a, float* __restrict__ x_c,
                                              float* __restrict__ x_p, const int64_t* __restrict__ rows) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int w4 = (ka + kc) / 4;
    if (i >= B * w4) return;
    const int64_t b = i / w4;
    const int j = (int)(i - b * w4) * 4;
    const int64_t src = rows ? rows[b] : b;
    const float* ob = obs + src * obs_dim;
    // observation rows of 4 k floats from a 16-byte aligned base: whole 16-byte groups
    const bool vec = (obs_dim & 3) == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0;
    if (j < ka) {
        if (!x_a) return;
        f32x4 v;
        if (vec && j + 4 <= obs_dim) v = *reinterpret_cast<const f32x4*>(ob + j);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = j + t < obs_dim ? ob[j + t] : 0.f;
        }
        *reinterpret_cast<f32x4*>(x_a + b * ka + j) = v;
    } else if (x_c) {
        const int k = j - ka;
        f32x4 v, vp;
        if (vec && k + 4 <= obs_dim) v = vp = *reinterpret_cast<const f32x4*>(ob + k);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = k + t;
                float e = 0.f;
                if (c < obs_dim) e = ob[c];
                vp[t] = e;
                if (c >= obs_dim && c < obs_dim + act_dim && act) e = act[src * act_dim + c - obs_dim];
                v[t] = e;
            }
        }
        *reinterpret_cast<f32x4*>(x_c + b * kc + k) = v;
        if (x_p) *reinterpret_cast<f32x4*>(x_p + b * kc + k) = vp;
    }
}

__global__ __launch_bounds__(256) void sac_pack_kernel(const float* __restrict__ obs, const float* __restrict__ act,
                                                       int64_t B, int obs_dim, int act_dim, int ka, int kc,
                                                       float* __restrict__ x_a, float* __restrict__ x_c,
                                                       float* __restrict__ x_p, const int64_t* __restrict__ rows = nullptr) {
    sac_pack_item((int64_t)blockIdx.x * 256 + threadIdx.x, obs, act, B, obs_dim, act_dim, ka, kc, x_a, x_c, x_p, rows);
}

// ts_sac_learn_rows: both packing passes of an update -- [obs | act] of the sampled rows for the critics / the actor, and
// obs_next of the same rows for the target pass -- and (nullable) the update's rsample() noise in ONE launch: blocks
// [0, pack_blocks) pack (obs, act) -> x_a, x_c, x_p (the observation columns again: the policy's action goes there while x_c
// still holds the buffer's); [pack_blocks, 2 pack_blocks) pack obs_next -> xn_a, xn_c (action columns left to the policy
// kernel, padding zero); the remaining blocks are ts_normal_fill(noise, noise_n, seed, offset).
struct Pack2Args {
    const float* obs; const float* act; const float* obs_next; const int64_t* rows;
    int64_t B; int obs_dim, act_dim, ka, kc;
    float* x_a; float* x_c; float* x_p; float* xn_a; float* xn_c;
    unsigned pack_blocks;
    float* noise; int64_t noise_n; uint64_t seed, offset; int noise_halves;
};
__global__ __launch_bounds__(256) void sac_pack2_kernel(Pack2Args a) {
    const unsigned blk = blockIdx.x;
    if (blk < a.pack_blocks) {
        sac_pack_item((int64_t)blk * 256 + threadIdx.x, a.obs, a.act, a.B, a.obs_dim, a.act_dim, a.ka, a.kc, a.x_a, a.x_c, a.x_p, a.rows);
    } else if (blk < 2 * a.pack_blocks) {
        sac_pack_item((int64_t)(blk - a.pack_blocks) * 256 + threadIdx.x, a.obs_next, nullptr, a.B, a.obs_dim, a.act_dim, a.ka, a.kc,
                      a.xn_a, a.xn_c, nullptr, a.rows);
    } else {
        // noise_halves = 1: ONE stream of noise_n elements at `offset`; 2: the two halves of `noise` are streams of their own at
        // offset and offset + 1 (noise_n elements each: two ts_normal_fill calls with consecutive counters, as the hooks draw)
        unsigned nb = blk - 2 * a.pack_blocks;
        const unsigned per = (unsigned)((((a.noise_n + 3) >> 2) + 255) >> 8);
        const unsigned half = nb >= per ? 1u : 0u;
        nb -= half * per;
        const int64_t q = (int64_t)nb * 256 + threadIdx.x;
        if (4 * q >= a.noise_n) return;
        float z[4];
        ts::normal4(q, a.seed, a.offset + half, z);
        float* out = a.noise + (int64_t)half * a.noise_n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < a.noise_n) out[4 * q + e] = z[e];
    }
}

// SACPolicy.forward after the actor MLP (sac.py:114-123): one thread per sample.
// head[b] = [mu (A) .. | raw log-sigma (A) at column 32 ..].  Writes the squashed action into x_c's action
// columns (nullable), act_out (nullable), logp_out; `keep` (nullable, [B, 3A]) stores {a - mu, sigma, squashed}
// for the backward pass.
__device__ __forceinline__ void sac_policy_item(int64_t t, const float* __restrict__ head, const float* __restrict__ noise,
                                                int64_t B, int A, int head_cols, int obs_dim, int kc, float bound,
                                                float* __restrict__ x_c, float* __restrict__ act_out,
                                                float* __restrict__ logp_out, float* __restrict__ keep,
                                                float* __restrict__ mu_out, float* __restrict__ sigma_out) {
    // half a wavefront per sample, lane j = action dimension j (A <= 32); the two sums over j are butterfly
    // reductions inside the half-wave (fixed order)
    const int64_t b = t >> 5;
    const int j = (int)(t & 31);
    float lp = 0.f, corr = 0.f;
    if (b < B && j < A) {
        const float* hb = head + b * head_cols;
        // `bound` > 0: ContinuousActorProbabilistic(unbounded=False), the class default -- mu = max_action * tanh(mu)
        // (continuous.py:230-231) in front of the Gaussian; 0 = the examples' unbounded actor
        const float mu = bound > 0.f ? bound * tanhf(hb[j]) : hb[j];
        const float sigma = expf(fminf(fmaxf(hb[SIG_COL + j], SIGMA_MIN), SIGMA_MAX));
        const float e = noise ? noise[b * A + j] : 0.f;
        const float a = mu + e * sigma;                              // Normal.rsample: loc + eps * scale
        const float d = a - mu;
        lp = -(d * d) / (2.f * (sigma * sigma)) - logf(sigma) - LOG_SQRT_2PI;      // Normal.log_prob
        const float sq = tanhf(a);
        corr = logf(1.f - sq * sq + TANH_EPS);                       // sac.py:38
        if (x_c) x_c[b * kc + obs_dim + j] = sq;
        if (act_out) act_out[b * A + j] = sq;
        if (keep) { keep[(b * 3 + 0) * A + j] = d; keep[(b * 3 + 1) * A + j] = sigma; keep[(b * 3 + 2) * A + j] = sq; }
        if (mu_out) mu_out[b * A + j] = mu;                          // `logits` of SACPolicy.forward (sac.py:114, 125)
        if (sigma_out) sigma_out[b * A + j] = sigma;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        lp += __shfl_xor(lp, o, 32);
        corr += __shfl_xor(corr, o, 32);
    }
    if (b < B && j == 0) logp_out[b] = lp - corr;
}
__global__ __launch_bounds__(256) void sac_policy_kernel(const float* __restrict__ head, const float* __restrict__ noise,
                                                         int64_t B, int A, int head_cols, int obs_dim, int kc, float bound,
                                                         float* __restrict__ x_c, float* __restrict__ act_out,
                                                         float* __restrict__ logp_out, float* __restrict__ keep,
                                                         float* __restrict__ mu_out = nullptr,
                                                         float* __restrict__ sigma_out = nullptr) {
    sac_policy_item((int64_t)blockIdx.x * 256 + threadIdx.x, head, noise, B, A, head_cols, obs_dim, kc, bound, x_c, act_out, logp_out,
                    keep, mu_out, sigma_out);
}
// ts_sac_learn_rows: the actor ran ONCE on [obs_next rows | obs rows] (its parameters do not change between the target pass and
// the actor pass of one update); blocks [0, nb) are the target pass's policy step (a' into xn_c, log pi(a'|s')), blocks
// [nb, 2 nb) the actor pass's (a into x_p, log pi(a|s), `keep` for the backward pass).  Workgroups never straddle the halves.
struct Policy2Args {
    const float* head; const float* noise_next; const float* noise; int64_t B; int A, obs_dim, kc; float bound; unsigned nb;
    float* xn_c; float* logp_n; float* x_p; float* logp; float* keep;
};
__global__ __launch_bounds__(256) void sac_policy2_kernel(Policy2Args a) {
    if (blockIdx.x < a.nb)
        sac_policy_item((int64_t)blockIdx.x * 256 + threadIdx.x, a.head, a.noise_next, a.B, a.A, 64, a.obs_dim, a.kc, a.bound, a.xn_c,
                        nullptr, a.logp_n, nullptr, nullptr, nullptr);
    else
        sac_policy_item((int64_t)(blockIdx.x - a.nb) * 256 + threadIdx.x, a.head + a.B * 64, a.noise, a.B, a.A, 64, a.obs_dim, a.kc,
                        a.bound, a.x_p, nullptr, a.logp, a.keep, nullptr, nullptr);
}

// SAC._target_q_compute_value: min(Q1_old, Q2_old) - alpha * log_prob  (q arrays are [B, 32], column 0)
// `rew` (nullable): also the 1-step return of compute_nstep_return (algorithm_base.py:785-817 with n_step = 1, in
// ts_returns.hip nstep_fused_kernel's arithmetic: the value mask multiplies in float32, gamma and the reward add in float64):
// out[b] = float(double(tq * mask) * gamma + rew[rows[b]]), mask = !terminated[rows[b]].
__device__ __forceinline__ float sac_target_value(float q1, float q2, float alpha, float logp) { return fminf(q1, q2) - alpha * logp; }
__device__ __forceinline__ float sac_one_step_return(float tq, uint8_t terminated, double gamma, double rew) {
    const float tqm = tq * (terminated ? 0.f : 1.f);
    const double q = (double)tqm * gamma;
    return (float)(q + rew);
}
__global__ __launch_bounds__(256) void sac_target_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                         const float* __restrict__ logp, const float* __restrict__ log_alpha,
                                                         float fixed_alpha, int64_t B, float* __restrict__ out,
                                                         const double* __restrict__ rew = nullptr,
                                                         const uint8_t* __restrict__ terminated = nullptr,
                                                         const int64_t* __restrict__ rows = nullptr, double gamma = 0.0) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float tq = sac_target_value(q1[b * 32], q2[b * 32], alpha, logp[b]);
    if (!rew) { out[b] = tq; return; }
    const int64_t r = rows ? rows[b] : b;
    out[b] = sac_one_step_return(tq, terminated[r], gamma, rew[r]);
}

// block-wide deterministic sum (1024 threads): butterfly inside each wavefront, then the 16 wave sums
__device__ float block_sum_1024(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                   // `red` may still be read by a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) r += red[w];
    return r;
}

// _minimize_critic_squared_loss (ddpg.py:279-284): td = Q - returns; loss = mean(td^2 w); d_out[b, 0] = 2 td w / B
__global__ __launch_bounds__(1024) void sac_critic_loss_kernel(const float* __restrict__ q, const float* __restrict__ ret,
                                                               const float* __restrict__ weight, int64_t B,
                                                               float* __restrict__ td, float* __restrict__ d_out,
                                                               float* __restrict__ loss) {
    __shared__ float red[1024];
    const float inv_b = 1.f / (float)B;
    float ls = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const float t = q[b * 32] - ret[b];
        const float w = weight ? weight[b] : 1.f;
        td[b] = t;
        ls += t * t * w;
        d_out[b * 32] = 2.f * t * w * inv_b;       // the other 31 columns of d_out stay zero
    }
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *loss = tot * inv_b;
}

// actor loss (sac.py:312-314) = mean(alpha log_prob - min(Q1, Q2)); upstream gradients for the two critics
__global__ __launch_bounds__(1024) void sac_actor_loss_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                              const float* __restrict__ logp,
                                                              const float* __restrict__ log_alpha, float fixed_alpha,
                                                              int64_t B, float* __restrict__ d_q1, float* __restrict__ d_q2,
                                                              float* __restrict__ loss) {
    __shared__ float red[1024];
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float inv_b = 1.f / (float)B;
    float ls = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const float a = q1[b * 32], c = q2[b * 32];
        ls += alpha * logp[b] - fminf(a, c);
        // torch.minimum backward: ties share the gradient
        store_head_row(d_q1 + b * 32, a < c ? -inv_b : (a == c ? -0.5f * inv_b : 0.f));
        store_head_row(d_q2 + b * 32, c < a ? -inv_b : (a == c ? -0.5f * inv_b : 0.f));
    }
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *loss = tot * inv_b;
}

// backward of sac_policy_kernel: d head[b] from d loss / d squashed action (sum of the two critics' input
// gradients, action columns of [B, kc]) and from the alpha * log_prob term.  Autograd's formulas, term by term.
__global__ __launch_bounds__(256) void sac_policy_bwd_kernel(const float* __restrict__ head, const float* __restrict__ noise,
                                                             const float* __restrict__ keep, const float* __restrict__ dx1,
                                                             const float* __restrict__ dx2,
                                                             const float* __restrict__ log_alpha, float fixed_alpha,
                                                             int64_t B, int A, int head_cols, int obs_dim, int kc, float bound,
                                                             float* __restrict__ d_head) {
    // thread (b, j < 32): columns j and 32 + j of row b; columns past the action dimension are written as zeros
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = i >> 5;
    const int j = (int)(i & 31);
    if (b >= B) return;
    if (j >= A) {
        d_head[b * head_cols + j] = 0.f;
        d_head[b * head_cols + SIG_COL + j] = 0.f;
        return;
    }
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float g_lp = alpha / (float)B;                              // d loss / d log_prob[b]
    const float d = keep[(b * 3 + 0) * A + j], sigma = keep[(b * 3 + 1) * A + j], sq = keep[(b * 3 + 2) * A + j];
    const float e = noise[b * A + j];
    const float var = sigma * sigma;
    const float one_m = 1.f - sq * sq;
    const float d_sq = dx1[b * kc + obs_dim + j] + dx2[b * kc + obs_dim + j];      // from -min(Q1, Q2)
    // log_prob = sum_j [-(a-mu)^2 / (2 var) - log sigma - c] - sum_j log(1 - sq^2 + eps)
    const float g_sq = d_sq + g_lp * (2.f * sq / (one_m + TANH_EPS));
    const float g_a = g_sq * one_m + g_lp * (-d / var);               // into a = mu + eps * sigma
    const float g_mu = g_a + g_lp * (d / var);
    const float g_sigma = g_a * e + g_lp * (d * d / (var * sigma) - 1.f / sigma);
    const float raw = head[b * head_cols + SIG_COL + j];
    const float g_raw = (raw >= SIGMA_MIN && raw <= SIGMA_MAX) ? g_sigma * sigma : 0.f;   // clamp().exp()
    float g_mu_raw = g_mu;
    if (bound > 0.f) {                                                // through mu = bound * tanh(raw mu)
        const float tm = tanhf(head[b * head_cols + j]);
        g_mu_raw = g_mu * (bound * (1.f - tm * tm));
    }
    d_head[b * head_cols + j] = g_mu_raw;
    d_head[b * head_cols + SIG_COL + j] = g_raw;
}

// The two loss kernels above run on ONE workgroup (4096 strided Q reads through a single CU: 10 - 17 us).  SAC's update
// uses these: one thread per sample over ceil(B / 256) workgroups, a partial loss sum per workgroup (`part`), summed in
// workgroup order by loss_finish (in sac_alpha_kernel at the end of a whole update, in sac_loss_finish_kernel after a
// single phase) -- a fixed order, so the loss statistics stay deterministic.
__device__ float block_sum_256(float v, float* red4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
    __syncthreads();
    return red4[0] + red4[1] + red4[2] + red4[3];
}

// blockIdx.y = critic (one launch for the twins when both run on one stream; gridDim.y = 1 otherwise)
struct CriticLossArgs {
    const float* q[2]; float* td[2]; float* d_out[2]; float* part[2];
    const float* ret; const float* weight; int64_t B;
    // ts_sac_learn_rows (rew != NULL): the 1-step return is formed here from the lagged critics' outputs instead of being read
    // -- sac_target_kernel's arithmetic (the same device functions); critic 0's blocks also write it to ret_out (nullable)
    const float* tq[2]; const float* logp_n; const float* log_alpha; float fixed_alpha;
    const double* rew; const uint8_t* terminated; const int64_t* rows; double gamma; float* ret_out;
};
__global__ __launch_bounds__(256) void sac_critic_loss_mb_kernel(CriticLossArgs a) {
    __shared__ float red[4];
    const int k = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float inv_b = 1.f / (float)a.B;
    float ls = 0.f;
    if (b < a.B) {
        float ret;
        if (a.rew) {
            const float alpha = a.log_alpha ? expf(*a.log_alpha) : a.fixed_alpha;
            const int64_t r = a.rows[b];
            ret = sac_one_step_return(sac_target_value(a.tq[0][b * 32], a.tq[1][b * 32], alpha, a.logp_n[b]), a.terminated[r], a.gamma, a.rew[r]);
            if (k == 0 && a.ret_out) a.ret_out[b] = ret;
        } else ret = a.ret[b];
        const float t = a.q[k][b * 32] - ret;
        const float w = a.weight ? a.weight[b] : 1.f;
        a.td[k][b] = t;
        ls = t * t * w;
        store_head_row(a.d_out[k] + b * 32, 2.f * t * w * inv_b);
    }
    const float tot = block_sum_256(ls, red);
    if (threadIdx.x == 0) a.part[k][blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sac_actor_loss_mb_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                                const float* __restrict__ logp,
                                                                const float* __restrict__ log_alpha, float fixed_alpha,
                                                                int64_t B, float* __restrict__ d_q1, float* __restrict__ d_q2,
                                                                float* __restrict__ part) {
    __shared__ float red[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float inv_b = 1.f / (float)B;
    float ls = 0.f;
    if (b < B) {
        const float a = q1[b * 32], c = q2[b * 32];
        ls = alpha * logp[b] - fminf(a, c);
        // torch.minimum backward: ties share the gradient
        store_head_row(d_q1 + b * 32, a < c ? -inv_b : (a == c ? -0.5f * inv_b : 0.f));
        store_head_row(d_q2 + b * 32, c < a ? -inv_b : (a == c ? -0.5f * inv_b : 0.f));
    }
    const float tot = block_sum_256(ls, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__device__ __forceinline__ float loss_finish(const float* part, int n, int64_t B) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    return s * (1.f / (float)B);
}

// mean losses from the partial sums (nullable pointers are skipped); one thread
__global__ void sac_loss_finish_kernel(const float* p0, float* o0, const float* p1, float* o1, int n, int64_t B) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (p0) *o0 = loss_finish(p0, n, B);
    if (p1) *o1 = loss_finish(p1, n, B);
}

// AutoAlpha.update (sac.py:203-209) + Adam on the scalar log_alpha; (td1 + td2) / 2 (sac.py:306)
struct AlphaArgs {
    const float* logp; int64_t B; float target_entropy;
    float* log_alpha; float* m; float* v;
    float lr_step, beta1, beta2, bc2_sqrt, eps, omb1, omb2;
    float* alpha_loss; float* alpha_out; float fixed_alpha;
    const float* td1; const float* td2; float* weight_out;
    const float* neg_mean_logp;      // data-parallel: -mean(log_prob) over the GLOBAL batch (replaces the local mean)
    const float* loss_part;          // whole update: partial sums of {actor, critic1, critic2} losses, n_part each
    float* losses; int n_part;       // -> losses[0 .. 2] (nullable loss_part: already finished by the phases)
};

// NT = 1024 threads, or 256 threads that each play four of the 1024 (thread t: t, t + 256, t + 512, t + 768 -- whole waves
// map to whole waves, so the butterfly sums and the order of the sixteen wave sums are those of the 1024-thread form)
template <int NT>
__device__ __forceinline__ void sac_alpha_body(const AlphaArgs& a, float* red) {     // red[16]
    constexpr int J = 1024 / NT;
    float s[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        s[j] = 0.f;
        for (int64_t b = threadIdx.x + NT * j; b < a.B; b += 1024) {
            s[j] += a.logp[b];
            if (a.weight_out) a.weight_out[b] = (a.td1[b] + a.td2[b]) / 2.f;
        }
    }
    // the three loss means: threads 64, 128, 192 (one per wave, beside the log-prob loads), same sequential sums as loss_finish
    if (a.loss_part && (threadIdx.x == 64 || threadIdx.x == 128 || threadIdx.x == 192)) {
        const int k = (threadIdx.x >> 6) - 1;
        a.losses[k] = loss_finish(a.loss_part + k * a.n_part, a.n_part, a.B);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j)
        if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) + (NT / 64) * j] = s[j];
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    if (threadIdx.x != 0) return;
    if (!a.log_alpha) { *a.alpha_out = a.fixed_alpha; *a.alpha_loss = 0.f; return; }
    // mean entropy deficit = mean(target - (-log_prob)); written so that the single-call and the phased update
    // (which receives -mean(log_prob) through the exchange buffer) evaluate the same float operations
    const float mean_def = a.target_entropy - (a.neg_mean_logp ? *a.neg_mean_logp : -(tot / (float)a.B));
    const float la = *a.log_alpha;
    *a.alpha_loss = -(la * mean_def);
    const float g = -mean_def;
    float m = *a.m, v = *a.v;
    m = m + (g - m) * a.omb1;
    v = v * a.beta2 + a.omb2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    const float nla = la + (-a.lr_step * m) / denom;
    *a.log_alpha = nla; *a.m = m; *a.v = v;
    *a.alpha_out = expf(nla);
}

__global__ __launch_bounds__(1024) void sac_alpha_kernel(AlphaArgs a) {
    __shared__ float red[16];
    sac_alpha_body<1024>(a, red);
}

// Weight-gradient slab sums + Adam (+ Polyak of the lagged copy) of up to two networks' three layers in ONE launch, and
// optionally the alpha step as one more workgroup: per element the operations of ts::slab_sum_multi (same slab order, same
// tree), ts_optim.hip's adam_multi_kernel and sac_alpha_kernel -- bit-identical to the three launches it replaces.
// Workgroups of 256 threads = slab_sum_multi's (many short workgroups in flight hide the Adam tail that only 16 of a
// workgroup's threads run); the alpha workgroup plays its 1024-thread form on 256.
constexpr int SLAB_ADAM_SEGS = 6;
struct SlabAdamArgs {
    const float* slabs[SLAB_ADAM_SEGS]; int64_t n[SLAB_ADAM_SEGS]; int nslab[SLAB_ADAM_SEGS];
    float* grad[SLAB_ADAM_SEGS]; float* p[SLAB_ADAM_SEGS]; float* m[SLAB_ADAM_SEGS]; float* v[SLAB_ADAM_SEGS];
    float* tgt[SLAB_ADAM_SEGS];
    unsigned first_vb[SLAB_ADAM_SEGS + 1];
    float lr_step, beta1, beta2, bc2_sqrt, eps, omb1, omb2, tau, one_minus_tau;
};

template <bool ALPHA>
__global__ __launch_bounds__(256) void slab_adam_kernel(SlabAdamArgs a, AlphaArgs al) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    __shared__ f32x4 red[256];
    if (ALPHA && blockIdx.x == 0) {               // the longest workgroup (a serial pass over the batch) starts first
        sac_alpha_body<256>(al, reinterpret_cast<float*>(red));
        return;
    }
    const int t = threadIdx.x;
    const unsigned vb = blockIdx.x - (ALPHA ? 1 : 0);
    int sg = 0;
#pragma unroll
    for (int k = 1; k < SLAB_ADAM_SEGS; ++k) sg += vb >= a.first_vb[k] ? 1 : 0;
    const float* __restrict__ slabs = a.slabs[sg];
    const int64_t n = a.n[sg];
    const int nslab = a.nslab[sg];
    const int col = t & 15, part = t >> 4;
    const int64_t i = ((int64_t)(vb - a.first_vb[sg]) * 16 + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 p4 = s, m4 = s, v4 = s, t4 = s;
    const bool owner = part == 0 && i < n;                    // the Adam operands of the chunk: in flight beside the slab loads
    if (owner) {
        p4 = *reinterpret_cast<const f32x4*>(a.p[sg] + i);
        m4 = *reinterpret_cast<const f32x4*>(a.m[sg] + i);
        v4 = *reinterpret_cast<const f32x4*>(a.v[sg] + i);
        if (a.tgt[sg]) t4 = *reinterpret_cast<const f32x4*>(a.tgt[sg] + i);
    }
    if (i < n)
        for (int k = part; k < nslab; k += 16) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * n + i);
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int st = 8; st > 0; st >>= 1) {
        if (part < st) red[threadIdx.x] += red[threadIdx.x + 16 * st];
        __syncthreads();
    }
    if (!owner) return;
    const f32x4 g4 = red[threadIdx.x];
    *reinterpret_cast<f32x4*>(a.grad[sg] + i) = g4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gq = g4[e];
        float m = m4[e], v = v4[e];
        m = m + (gq - m) * a.omb1;                           // exp_avg.lerp_(grad, 1 - beta1)
        v = v * a.beta2 + a.omb2 * gq * gq;                  // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        const float pn = p4[e] + (-a.lr_step * m) / denom;   // addcdiv_(m, denom, -step_size)
        p4[e] = pn; m4[e] = m; v4[e] = v;
        t4[e] = a.tau * pn + a.one_minus_tau * t4[e];        // lagged_network.py:17-18
    }
    *reinterpret_cast<f32x4*>(a.p[sg] + i) = p4;
    *reinterpret_cast<f32x4*>(a.m[sg] + i) = m4;
    *reinterpret_cast<f32x4*>(a.v[sg] + i) = v4;
    if (a.tgt[sg]) *reinterpret_cast<f32x4*>(a.tgt[sg] + i) = t4;
}

// mlp_weight_grads + the Adam step (+ Polyak, + alpha) behind it: the 3 n weight-gradient GEMMs in one launch, then
// slab_adam_kernel.  p / m / v / lag: the n networks' flat vectors (lag nullable); alpha (nullable): the alpha step rides along.
struct AdamSpec { int64_t step; double lr, beta1, beta2, eps, tau; };
int mlp_weight_grads_adam(hipStream_t s, ts_workspace* ws, int n, const Mlp& m, const float* const* x, const Act* a,
                          const float* const* d_out, float* const* grad, const BwdScratch* sc, float* const* p, float* const* pm,
                          float* const* pv, float* const* lag, const AdamSpec& sp, const AlphaArgs* alpha) {
    TS_REQUIRE(n >= 1 && 3 * n <= SLAB_ADAM_SEGS && m.L == 3, TS_ERR_INVALID_ARG, "mlp_weight_grads_adam: 1 .. 2 three-layer networks");
    ts::ConvGeom geoms[SLAB_ADAM_SEGS];
    const float* X[SLAB_ADAM_SEGS];
    const float* dY[SLAB_ADAM_SEGS];
    float* slabs[SLAB_ADAM_SEGS];
    SlabAdamArgs g{};
    unsigned vbs = 0;
    for (int j = 0; j < SLAB_ADAM_SEGS; ++j) g.first_vb[j] = 0xffffffffu;
    for (int k = 0; k < n; ++k) {
        const float* xin[3] = {x[k], a[k].h[0], a[k].h[1]};
        const float* dy[3] = {sc[k].dh[0], sc[k].dh[1], d_out[k]};
        size_t off = 0;
        for (int i = 2; i >= 0; --i) {
            const int j = 3 * k + (2 - i);
            const int ns = ts::conv_wgrad_splits(m.l[i]);
            const int64_t pe = m.l[i].param_elems();
            TS_REQUIRE(pe % 4 == 0, TS_ERR_INVALID_ARG, "mlp_weight_grads_adam: layer sizes must be multiples of 4");
            geoms[j] = m.l[i]; X[j] = xin[i]; dY[j] = dy[i]; slabs[j] = sc[k].slabs + off;
            g.slabs[j] = sc[k].slabs + off; g.nslab[j] = ns; g.n[j] = pe;
            g.grad[j] = grad[k] + m.off[i]; g.p[j] = p[k] + m.off[i]; g.m[j] = pm[k] + m.off[i]; g.v[j] = pv[k] + m.off[i];
            g.tgt[j] = (lag && lag[k] && sp.tau > 0.0) ? lag[k] + m.off[i] : nullptr;
            g.first_vb[j] = vbs;
            vbs += (unsigned)ts::ceil_div(pe, 64);
            off += (size_t)ns * pe;
        }
    }
    for (int j = 3 * n; j <= SLAB_ADAM_SEGS; ++j) g.first_vb[j] = j == SLAB_ADAM_SEGS ? vbs : 0xffffffffu;
    g.first_vb[SLAB_ADAM_SEGS] = vbs;
    const double bc1 = 1.0 - pow(sp.beta1, (double)sp.step), bc2 = 1.0 - pow(sp.beta2, (double)sp.step);
    g.lr_step = (float)(sp.lr / bc1);
    g.beta1 = (float)sp.beta1; g.beta2 = (float)sp.beta2;
    g.omb1 = (float)(1.0 - sp.beta1); g.omb2 = (float)(1.0 - sp.beta2);
    g.bc2_sqrt = (float)sqrt(bc2);
    g.eps = (float)sp.eps;
    g.tau = (float)sp.tau; g.one_minus_tau = (float)(1.0 - sp.tau);
    if (int rc = ts::conv_wgrad_group(s, 3 * n, geoms, X, dY, slabs, ws)) return rc;
    if (alpha) hipLaunchKernelGGL(slab_adam_kernel<true>, dim3(vbs + 1), dim3(256), 0, s, g, *alpha);
    else hipLaunchKernelGGL(slab_adam_kernel<false>, dim3(vbs), dim3(256), 0, s, g, AlphaArgs{});
    TS_LAUNCH_CHECK();
    return TS_OK;
}

__global__ __launch_bounds__(256) void polyak2_kernel(float* __restrict__ t1, const float* __restrict__ s1,
                                                      float* __restrict__ t2, const float* __restrict__ s2, int64_t n,
                                                      float tau, float omt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    t1[i] = tau * s1[i] + omt * t1[i];
    t2[i] = tau * s2[i] + omt * t2[i];
}

// ---- TD3 / DDPG (deterministic actor) ---------------------------------------------------------------------
// ContinuousActorDeterministic.forward (continuous.py:70-85): a = max_action * tanh(head); TD3's target policy
// smoothing (td3.py:195-199): a += clamp(noise * policy_noise, +-noise_clip) (no clamp when noise_clip <= 0)
__global__ __launch_bounds__(256) void det_policy_kernel(const float* __restrict__ head, const float* __restrict__ noise,
                                                         int64_t B, int A, float max_action, float policy_noise,
                                                         float noise_clip, int obs_dim, int kc, float* __restrict__ x_c,
                                                         float* __restrict__ act_out, float* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * A) return;
    const int64_t b = i / A;
    const int j = (int)(i - b * A);
    const float t = tanhf(head[b * 32 + j]);
    float a = max_action * t;
    if (noise) {
        float n = noise[i] * policy_noise;
        if (noise_clip > 0.f) n = fminf(fmaxf(n, -noise_clip), noise_clip);
        a += n;
    }
    if (x_c) x_c[b * kc + obs_dim + j] = a;
    if (act_out) act_out[i] = a;
    if (keep) keep[i] = t;
}

__global__ __launch_bounds__(256) void td3_target_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                         int64_t B, float* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    out[b] = q2 ? fminf(q1[b * 32], q2[b * 32]) : q1[b * 32];
}

// actor loss = -Q1(s, pi(s)).mean()  (ddpg.py:407, td3.py:216)
__global__ __launch_bounds__(1024) void det_actor_loss_kernel(const float* __restrict__ q1, int64_t B,
                                                              float* __restrict__ d_q1, float* __restrict__ loss) {
    __shared__ float red[1024];
    const float inv_b = 1.f / (float)B;
    float ls = 0.f;
    if (blockIdx.x > 0) {            // workgroups 1..: the d_q1 rows (workgroup 0 keeps the sum and its order)
        const int64_t b = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x;
        if (b < B) store_head_row(d_q1 + b * 32, -inv_b);
        return;
    }
    for (int64_t b = threadIdx.x; b < B; b += 1024) ls += q1[b * 32];
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *loss = -(tot * inv_b);
}

__global__ __launch_bounds__(256) void det_policy_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ keep,
                                                             int64_t B, int A, float max_action, int obs_dim, int kc,
                                                             float* __restrict__ d_head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // thread (b, j < 32): the whole 32-column row is written
    const int64_t b = i >> 5;
    const int j = (int)(i & 31);
    if (b >= B) return;
    if (j >= A) { d_head[b * 32 + j] = 0.f; return; }
    const float t = keep[b * A + j];
    d_head[b * 32 + j] = dx[b * kc + obs_dim + j] * max_action * (1.f - t * t);
}

__global__ __launch_bounds__(1024) void td3_weight_kernel(const float* __restrict__ td1, const float* __restrict__ td2,
                                                          int64_t B, float* __restrict__ out) {
    for (int64_t b = (int64_t)blockIdx.x * 1024 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 1024)
        out[b] = td2 ? (td1[b] + td2[b]) / 2.f : td1[b];
}

__global__ __launch_bounds__(256) void polyak1_kernel(float* __restrict__ t, const float* __restrict__ s, int64_t n, float tau,
                                                      float omt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) t[i] = tau * s[i] + omt * t[i];
}

// ---- DiscreteSAC kernels (one thread per sample; logits / Q rows are [hw] wide, the first A columns are real) -----------
__device__ __forceinline__ float row_logsumexp(const float* l, int A) {
    float m = l[0];
    for (int j = 1; j < A; ++j) m = fmaxf(m, l[j]);
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(l[j] - m);
    return m + logf(s);
}

// _target_q_compute_value (discrete_sac.py:147-155): sum_a p_a min(Q1_old, Q2_old)_a + alpha H(p)
__global__ __launch_bounds__(256) void dsac_target_kernel(const float* __restrict__ logits, const float* __restrict__ q1,
                                                          const float* __restrict__ q2, const float* __restrict__ log_alpha,
                                                          float fixed_alpha, int64_t B, int A, int hw,
                                                          float* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float* l = logits + b * hw;
    const float lse = row_logsumexp(l, A);
    float sq = 0.f, plogp = 0.f;
    for (int j = 0; j < A; ++j) {
        const float lp = l[j] - lse, p = expf(lp);
        sq += p * fminf(q1[b * hw + j], q2[b * hw + j]);
        plogp += lp * p;
    }
    out[b] = sq + alpha * -plogp;
}

// critic step (discrete_sac.py:162-172): td = Q(s)[a] - returns; loss = mean(td^2 w); d_out[b, a_b] = 2 td w / B
struct DsacLossArgs { const float* q[2]; float* td[2]; float* d_out[2]; float* loss[2]; };
__global__ __launch_bounds__(1024) void dsac_critic_loss_kernel(DsacLossArgs la, const int64_t* __restrict__ act,
                                                                const float* __restrict__ ret, const float* __restrict__ weight,
                                                                int64_t B, int hw) {      // blockIdx.x = critic
    // grid (critics, 1 + ceil(B hw / 1024)): row 0 sums the loss and writes td (the order of one 1,024-thread workgroup),
    // rows 1.. write d_out, one element per thread -- as one workgroup per critic whose threads each wrote whole 128-byte rows
    // this was a 43 us launch
    __shared__ float red[1024];
    const float* __restrict__ q = la.q[blockIdx.x];
    float* __restrict__ td = la.td[blockIdx.x];
    float* __restrict__ d_out = la.d_out[blockIdx.x];
    const float inv_b = 1.f / (float)B;
    if (blockIdx.y > 0) {
        const int64_t el = (int64_t)(blockIdx.y - 1) * 1024 + threadIdx.x;
        if (el >= B * hw) return;
        const int64_t b = el / hw;
        const int j = (int)(el - b * hw), a = (int)act[b];
        const float t = q[b * hw + a] - ret[b];
        const float w = weight ? weight[b] : 1.f;
        d_out[el] = j == a ? 2.f * t * w * inv_b : 0.f;
        return;
    }
    float ls = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const int a = (int)act[b];
        const float t = q[b * hw + a] - ret[b];
        const float w = weight ? weight[b] : 1.f;
        td[b] = t;
        ls += t * t * w;
    }
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *la.loss[blockIdx.x] = tot * inv_b;
}

// ---- the same two per-sample kernels with 32 lanes per sample (hw = 32: lane j = column j, two samples per wave): coalesced
// rows, the row reductions as five-step shuffles inside the half wave.  One thread per sample walked its 128-byte row alone
// (29 us for the actor step, 13 us for the target at B = 4096).
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 32));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    return v;
}
struct RowSoftmax { float lp, p, sq, plogp; };
__device__ __forceinline__ RowSoftmax row_softmax32(const float* __restrict__ logits, const float* __restrict__ q1,
                                                    const float* __restrict__ q2, int64_t b, int j, int A) {
    const bool live = j < A;
    const float l = live ? logits[b * 32 + j] : -3.0e38f;
    const float m = half_max(l);
    const float lse = m + logf(half_sum(live ? expf(l - m) : 0.f));
    RowSoftmax r;
    r.lp = live ? l - lse : 0.f;
    r.p = live ? expf(r.lp) : 0.f;
    const float qm = live ? fminf(q1[b * 32 + j], q2[b * 32 + j]) : 0.f;
    r.sq = half_sum(r.p * qm);
    r.plogp = half_sum(r.lp * r.p);
    return r;
}

__global__ __launch_bounds__(256) void dsac_target32_kernel(const float* __restrict__ logits, const float* __restrict__ q1,
                                                            const float* __restrict__ q2, const float* __restrict__ log_alpha,
                                                            float fixed_alpha, int64_t B, int A, float* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = g >> 5 < B ? g >> 5 : B - 1;
    const int j = (int)(g & 31);
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const RowSoftmax r = row_softmax32(logits, q1, q2, b, j, A);
    if (j == 0 && (g >> 5) < B) out[b] = r.sq + alpha * -r.plogp;
}

__global__ __launch_bounds__(256) void dsac_actor32_kernel(const float* __restrict__ logits, const float* __restrict__ q1,
                                                           const float* __restrict__ q2, const float* __restrict__ log_alpha,
                                                           float fixed_alpha, int64_t B, int A, float* __restrict__ d_head,
                                                           float* __restrict__ neg_ent, float* __restrict__ f_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = (g >> 5) < B;
    const int64_t b = in ? g >> 5 : B - 1;
    const int j = (int)(g & 31);
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float inv_b = 1.f / (float)B;
    const RowSoftmax r = row_softmax32(logits, q1, q2, b, j, A);
    const float H = -r.plogp;
    if (!in) return;
    float d = 0.f;
    if (j < A) d = -inv_b * (r.p * (fminf(q1[b * 32 + j], q2[b * 32 + j]) - r.sq) - alpha * r.p * (r.lp + H));
    d_head[b * 32 + j] = d;
    if (j == 0) { neg_ent[b] = -H; f_out[b] = alpha * H + r.sq; }
}

// actor step (discrete_sac.py:176-184): f_b = alpha H_b + sum_a p_a q_a, q = min(Q1, Q2) (no grad); loss = -mean f.
// d f / d logit_k = p_k (q_k - sum_a p_a q_a) - alpha p_k (log p_k + H)
__global__ __launch_bounds__(256) void dsac_actor_kernel(const float* __restrict__ logits, const float* __restrict__ q1,
                                                         const float* __restrict__ q2, const float* __restrict__ log_alpha,
                                                         float fixed_alpha, int64_t B, int A, int hw,
                                                         float* __restrict__ d_head, float* __restrict__ neg_ent,
                                                         float* __restrict__ f_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float inv_b = 1.f / (float)B;
    const float* l = logits + b * hw;
    const float lse = row_logsumexp(l, A);
    float sq = 0.f, plogp = 0.f;
    for (int j = 0; j < A; ++j) {
        const float lp = l[j] - lse, p = expf(lp);
        sq += p * fminf(q1[b * hw + j], q2[b * hw + j]);
        plogp += lp * p;
    }
    const float H = -plogp;
    for (int j = 0; j < hw; ++j) {
        float d = 0.f;
        if (j < A) {
            const float lp = l[j] - lse, p = expf(lp);
            d = -inv_b * (p * (fminf(q1[b * hw + j], q2[b * hw + j]) - sq) - alpha * p * (lp + H));
        }
        d_head[b * hw + j] = d;
    }
    neg_ent[b] = -H;
    f_out[b] = alpha * H + sq;
}

__global__ __launch_bounds__(1024) void neg_mean_kernel(const float* __restrict__ v, int64_t B, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) s += v[b];
    const float tot = block_sum_1024(s, red);
    if (threadIdx.x == 0) *out = -(tot / (float)B);
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, int64_t B, int A, int hw,
                                                         float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * A) return;
    const int64_t b = i / A;
    dst[i] = src[b * hw + (i - b * A)];
}

// ---- REDQ (redq.py): SAC's policy + an ensemble of critics ------------------------------------------------------------------
// _target_q_compute_value (redq.py:248-261): min or mean over the sampled subset (qs[k] = out of subset member k, [B, 32]
// rows, column 0) minus alpha * log_prob
__global__ __launch_bounds__(256) void redq_target_kernel(const float* __restrict__ qs, int S, int64_t stride,
                                                          const float* __restrict__ logp, const float* __restrict__ log_alpha,
                                                          float fixed_alpha, int mean_mode, int64_t B, float* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    float v = qs[b * 32];
    for (int k = 1; k < S; ++k) {
        const float q = qs[k * stride + b * 32];
        v = mean_mode ? v + q : fminf(v, q);
    }
    if (mean_mode) v = v / (float)S;
    out[b] = v - alpha * logp[b];
}

// one member's share of the ensemble loss (redq.py:266-270): td_e = Q_e - returns, loss = sum_e sum_b td^2 w / (E B)
// (blockIdx.x = member of the launch: Q, td, d_out and the loss part of member k sit k strides after the first one's)
// grid (members, 1 + ceil(B / 1024)): row 0 of the grid sums the loss and writes td (one workgroup per member: the summation
// order of a single 1,024-thread workgroup), rows 1.. write the 128-byte d_out rows -- as one workgroup per member those 512 KB
// of stores made this a 28 us launch.
__global__ __launch_bounds__(1024) void redq_critic_loss_kernel(const float* __restrict__ q, int64_t q_stride,
                                                                const float* __restrict__ ret, const float* __restrict__ weight,
                                                                int64_t B, float inv_eb, float* __restrict__ td,
                                                                float* __restrict__ d_out, int64_t d_stride,
                                                                float* __restrict__ loss_part) {
    __shared__ float red[1024];
    const int64_t e = blockIdx.x;
    q += e * q_stride; td += e * B; d_out += e * d_stride; loss_part += e;
    if (blockIdx.y > 0) {
        const int64_t b = (int64_t)(blockIdx.y - 1) * 1024 + threadIdx.x;
        if (b >= B) return;
        const float t = q[b * 32] - ret[b];
        const float w = weight ? weight[b] : 1.f;
        store_head_row(d_out + b * 32, 2.f * t * w * inv_eb);
        return;
    }
    float ls = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const float t = q[b * 32] - ret[b];
        const float w = weight ? weight[b] : 1.f;
        td[b] = t;
        ls += t * t * w;
    }
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *loss_part = tot * inv_eb;
}

// critic_loss = sum_e loss_part[e]; batch.weight = mean_e td_e (redq.py:272)
__global__ __launch_bounds__(256) void redq_finish_kernel(const float* __restrict__ tds, const float* __restrict__ loss_parts, int E,
                                                          int64_t B, float* __restrict__ critic_loss,
                                                          float* __restrict__ weight_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b == 0) {
        float t = 0.f;
        for (int e = 0; e < E; ++e) t += loss_parts[e];
        *critic_loss = t;
    }
    if (b >= B || !weight_out) return;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += tds[(int64_t)e * B + b];
    weight_out[b] = s / (float)E;
}

// actor loss (redq.py:280-282) = mean_b(alpha log_prob - mean_e Q_e); every member receives d Q_e[b] = -1 / (E B)
__global__ __launch_bounds__(1024) void redq_actor_loss_kernel(const float* __restrict__ qs, int E, int64_t stride,
                                                               const float* __restrict__ logp,
                                                               const float* __restrict__ log_alpha, float fixed_alpha, int64_t B,
                                                               float* __restrict__ d_q, float* __restrict__ loss) {
    __shared__ float red[1024];
    const float alpha = log_alpha ? expf(*log_alpha) : fixed_alpha;
    const float inv_b = 1.f / (float)B, g = -1.f / ((float)E * (float)B);
    float ls = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        float qa = 0.f;
        for (int e = 0; e < E; ++e) qa += qs[e * stride + b * 32];
        ls += alpha * logp[b] - qa / (float)E;
        d_q[b * 32] = g;
    }
    const float tot = block_sum_1024(ls, red);
    if (threadIdx.x == 0) *loss = tot * inv_b;
}

// dst[b, col0 + j] (+)= src[b, col0 + j], j < A: the action columns of an input gradient [B, kc]
__global__ __launch_bounds__(256) void acc_cols_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t B, int kc,
                                                       int col0, int A, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * A) return;
    const int64_t b = i / A;
    const int64_t o = b * kc + col0 + (i - b * A);
    dst[o] = first ? src[o] : dst[o] + src[o];
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Carve {
    char* p;
    template <class T> T* take(size_t n) { T* r = reinterpret_cast<T*>(p); p += al(sizeof(T) * n); return r; }
};

struct Dims { int obs, act, ka, kc, hid, depth, fn; float bound; };     // fn: TS_NET_ACT_* of the trunks

// hidden: width of the two hidden layers of every Net[h, h] of the SAC / TD3 / DDPG / REDQ entry points -- a property of
// the workspace (ts_mlp_set_hidden; 0 = the examples' 256).  Any multiple of 32 up to 1024 runs: 256 on the fused
// three-layer kernels of ts_mlp.hip, everything else on the per-layer GEMM kernels.
int make_dims_h(int64_t obs_dim, int64_t act_dim, int64_t hidden, Dims* d, int64_t depth = 0) {
    TS_REQUIRE(obs_dim >= 1 && obs_dim <= 65536 && act_dim >= 1 && act_dim <= 32, TS_ERR_INVALID_ARG,
               "sac: obs_dim must be >= 1 and act_dim in [1, 32]");
    if (hidden == 0) hidden = HID;
    TS_REQUIRE(hidden >= 32 && hidden <= 1024 && hidden % 32 == 0, TS_ERR_INVALID_ARG,
               "sac: hidden width must be a multiple of 32 in [32, 1024], got %lld", (long long)hidden);
    if (depth == 0) depth = 2;
    TS_REQUIRE(depth >= 1 && depth <= MAXD, TS_ERR_INVALID_ARG, "sac: 1 .. %d hidden layers, got %lld", MAXD, (long long)depth);
    d->obs = (int)obs_dim; d->act = (int)act_dim; d->hid = (int)hidden; d->depth = (int)depth;
    d->ka = pad32(d->obs); d->kc = pad32(d->obs + d->act);
    d->bound = 0.f;
    d->fn = TS_NET_ACT_RELU;
    return TS_OK;
}

// hidden width and depth: properties of the workspace (ts_mlp_set_hidden / ts_mlp_set_trunk)
int make_dims(const ts_workspace* ws, int64_t obs_dim, int64_t act_dim, Dims* d) {
    if (int rc = make_dims_h(obs_dim, act_dim, ws ? ws->mlp_hidden : 0, d, ws ? ws->mlp_depth : 0)) return rc;
    d->bound = ws ? ws->sac_actor_bound : 0.f;          // ts_sac_set_actor_bound (SAC's / REDQ's Gaussian actor only)
    d->fn = ws && ws->mlp_act_tanh ? TS_NET_ACT_TANH : TS_NET_ACT_RELU;       // ts_mlp_set_activation
    return TS_OK;
}

Act take_act(Carve& c, int64_t B, int head_cols, int hid = HID, int depth = 2) {
    Act a{};
    for (int i = 0; i < depth; ++i) a.h[i] = c.take<float>(B * hid);
    a.out = c.take<float>(B * head_cols);
    return a;
}
BwdScratch take_scratch(Carve& c, int64_t B, int hid, int depth, size_t slab) {
    BwdScratch sc{};
    for (int i = depth - 1; i >= 0; --i) sc.dh[i] = c.take<float>(B * hid);
    sc.slabs = c.take<float>(slab);
    return sc;
}
// bytes of one PAIR of hidden-width buffers' share in the workspace formulas below, which count two buffers per network and
// pass (h1, h2 / dh1, dh2): a trunk of `depth` hidden layers needs depth of them, i.e. ceil(depth / 2) per counted pair member
template <class D> inline size_t hbytes(int64_t B, const D& d) { return al(4 * B * d.hid) * (size_t)((d.depth + 1) / 2); }

// ---- DiscreteSAC (discrete_sac.py): three MLPs obs -> hid -> hid -> n_act, Categorical policy ------------------------
struct DDims { int obs, act, hid, ka, hw, depth, fn; int64_t P; };

int make_ddims(int64_t obs_dim, int64_t n_act, int64_t hidden, DDims* d, int64_t depth = 0) {
    TS_REQUIRE(obs_dim >= 1 && obs_dim <= 65536 && n_act >= 2 && n_act <= 64 && hidden >= 32 && hidden <= 2048 &&
                   hidden % 32 == 0, TS_ERR_INVALID_ARG,
               "dsac: obs_dim >= 1, n_act in [2, 64], hidden a multiple of 32 in [32, 2048]");
    if (depth == 0) depth = 2;
    TS_REQUIRE(depth >= 1 && depth <= MAXD, TS_ERR_INVALID_ARG, "dsac: 1 .. %d hidden layers, got %lld", MAXD, (long long)depth);
    d->obs = (int)obs_dim; d->act = (int)n_act; d->hid = (int)hidden; d->depth = (int)depth; d->fn = TS_NET_ACT_RELU;
    d->ka = pad32(d->obs); d->hw = pad32(d->act);
    d->P = make_mlp(1, d->ka, d->hw, d->hid, d->depth).total();
    return TS_OK;
}

}  // namespace

extern "C" {

int ts_sac_layout(int64_t obs_dim, int64_t act_dim, int64_t* h_out8) { return ts_sac_layout_h(obs_dim, act_dim, 0, h_out8); }

int ts_mlp_layout(int64_t in_dim, int64_t hidden, int64_t depth, int64_t head_cols, int64_t* h_out) {
    TS_REQUIRE(h_out, TS_ERR_INVALID_ARG, "ts_mlp_layout: NULL output");
    TS_REQUIRE(in_dim >= 1 && in_dim <= 65536 + 32 && (head_cols == 32 || head_cols == 64), TS_ERR_INVALID_ARG,
               "ts_mlp_layout: in_dim >= 1, head_cols 32 or 64");
    if (hidden == 0) hidden = HID;
    if (depth == 0) depth = 2;
    TS_REQUIRE(hidden >= 32 && hidden <= 2048 && hidden % 32 == 0 && depth >= 1 && depth <= MAXD, TS_ERR_INVALID_ARG,
               "ts_mlp_layout: hidden a multiple of 32, 1 .. %d hidden layers", MAXD);
    const Mlp m = make_mlp(1, pad32((int)in_dim), (int)head_cols, (int)hidden, (int)depth);
    h_out[0] = pad32((int)in_dim);
    for (int i = 0; i <= m.L; ++i) h_out[1 + i] = m.off[i];
    return TS_OK;
}

int ts_sac_layout_h(int64_t obs_dim, int64_t act_dim, int64_t hidden, int64_t* h_out8) {
    Dims d;
    if (int rc = make_dims_h(obs_dim, act_dim, hidden, &d)) return rc;
    TS_REQUIRE(h_out8, TS_ERR_INVALID_ARG, "ts_sac_layout: NULL output");
    const Mlp a = make_mlp(1, d.ka, 64, d.hid, d.depth, d.fn), c = make_mlp(1, d.kc, 32, d.hid, d.depth, d.fn);
    h_out8[0] = d.ka; h_out8[1] = d.kc; h_out8[2] = a.total(); h_out8[3] = c.total();
    h_out8[4] = a.off[1]; h_out8[5] = a.off[2]; h_out8[6] = c.off[1]; h_out8[7] = c.off[2];
    return TS_OK;
}

int ts_sac_policy_forward(ts_workspace* ws, const float* actor, const float* obs, const float* noise, int64_t B,
                          int64_t obs_dim, int64_t act_dim, float* act_out, float* logp_out, float* mu_sigma_out,
                          ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_policy_forward: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && obs && logp_out, TS_ERR_INVALID_ARG, "ts_sac_policy_forward: bad argument");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn);
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + 3 * hbytes(B, d) + al(4 * B * 3 * d.act) +
                                        al(4 * split_floats(ma)) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    const Act aa = take_act(c, B, 64, d.hid, d.depth);
    float* keep = c.take<float>(B * 3 * d.act);
    float* split = c.take<float>(split_floats(ma));
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, (float*)nullptr, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, ma, actor, x_a, aa, split)) return rc;
    hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B, d.act,
                       64, d.obs, d.kc, d.bound, (float*)nullptr, act_out, logp_out, mu_sigma_out ? keep : nullptr);
    TS_LAUNCH_CHECK();
    if (mu_sigma_out) {      // {a - mu, sigma, squashed} rows, for diagnostics / tests
        TS_HIP_CHECK(hipMemcpyAsync(mu_sigma_out, keep, sizeof(float) * B * 3 * d.act, hipMemcpyDeviceToDevice, s));
    }
    return TS_OK;
}

int ts_sac_policy_forward_logits(ts_workspace* ws, const float* actor, const float* obs, const float* noise, int64_t B,
                                 int64_t obs_dim, int64_t act_dim, float* act_out, float* logp_out, float* mu_out,
                                 float* sigma_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_policy_forward_logits: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && obs && logp_out, TS_ERR_INVALID_ARG, "ts_sac_policy_forward_logits: bad argument");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn);
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + 3 * hbytes(B, d) + al(4 * split_floats(ma)) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    const Act aa = take_act(c, B, 64, d.hid, d.depth);
    float* split = c.take<float>(split_floats(ma));
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, (float*)nullptr, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, ma, actor, x_a, aa, split)) return rc;
    hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B, d.act,
                       64, d.obs, d.kc, d.bound, (float*)nullptr, act_out, logp_out, (float*)nullptr, mu_out, sigma_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

static int sac_target_impl(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                           const float* log_alpha, double fixed_alpha, const float* obs_next, const float* noise, int64_t B,
                           int64_t obs_dim, int64_t act_dim, float* out, ts_stream_t stream, const int64_t* rows,
                           const double* rew, const uint8_t* terminated, double gamma) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_target_q: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && critic1_old && critic2_old && obs_next && noise && out, TS_ERR_INVALID_ARG,
               "ts_sac_target_q: bad argument");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    const size_t spl = std::max(split_floats(ma), split_floats(mc));
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + al(4 * B * d.kc) + 9 * hbytes(B, d) + 2 * al(4 * B) +
                                        2 * al(4 * spl) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    float* x_c = c.take<float>(B * d.kc);
    const Act aa = take_act(c, B, 64, d.hid, d.depth), a1 = take_act(c, B, 32, d.hid, d.depth), a2 = take_act(c, B, 32, d.hid, d.depth);
    float* logp = c.take<float>(B);
    float* split = c.take<float>(spl);
    float* split2 = c.take<float>(spl);
    hipStream_t side;
    if (int rc = twin_stream(ws, s, mc, &side)) return rc;
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs_next,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, x_c, (float*)nullptr, rows);
    if (int rc = mlp_forward(s, ws, ma, actor, x_a, aa, split)) return rc;
    hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B, d.act,
                       64, d.obs, d.kc, d.bound, x_c, (float*)nullptr, logp, (float*)nullptr);
    TS_LAUNCH_CHECK();
    if (side == s) {                                                    // one stream: both lagged critics in one launch
        const float* pp[2] = {critic1_old, critic2_old};
        const Act aa2[2] = {a1, a2};
        float* sp[2] = {split, split2};
        if (int rc = mlp_forward_twin(s, ws, mc, pp, x_c, aa2, sp)) return rc;
    } else {
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;        // the two lagged critics side by side
        if (int rc = mlp_forward(side, ws, mc, critic2_old, x_c, a2, split2)) return rc;
        if (int rc = mlp_forward(s, ws, mc, critic1_old, x_c, a1, split)) return rc;
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    }
    hipLaunchKernelGGL(sac_target_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, a1.out, a2.out, logp,
                       log_alpha, (float)fixed_alpha, B, out, rew, terminated, rows, gamma);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_sac_target_q(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                    const float* log_alpha, double fixed_alpha, const float* obs_next, const float* noise, int64_t B,
                    int64_t obs_dim, int64_t act_dim, float* out, ts_stream_t stream) {
    return sac_target_impl(ws, actor, critic1_old, critic2_old, log_alpha, fixed_alpha, obs_next, noise, B, obs_dim, act_dim, out,
                           stream, nullptr, nullptr, nullptr, 0.0);
}

int ts_sac_returns_rows(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                        const float* log_alpha, double fixed_alpha, const float* obs_next_buf, const double* rew_buf,
                        const uint8_t* terminated_buf, const int64_t* rows, const float* noise, int64_t B, int64_t obs_dim,
                        int64_t act_dim, double gamma, float* returns_out, ts_stream_t stream) {
    TS_REQUIRE(obs_next_buf && rew_buf && terminated_buf && rows && returns_out, TS_ERR_INVALID_ARG,
               "ts_sac_returns_rows: NULL argument");
    return sac_target_impl(ws, actor, critic1_old, critic2_old, log_alpha, fixed_alpha, obs_next_buf, noise, B, obs_dim, act_dim,
                           returns_out, stream, rows, rew_buf, terminated_buf, gamma);
}

}  // extern "C"

namespace {
// Phases of one SAC update.  ts_sac_update runs all four in one call; ts_sac_update_phase runs one at a time so
// that data-parallel replicas can all-reduce the gradients between "grad" and "apply" (the workspace keeps the packed
// inputs, TD errors, policy intermediates and log-probabilities between the calls).
enum : int { PH_CRITIC_GRAD = 1, PH_CRITIC_APPLY = 2, PH_ACTOR_GRAD = 4, PH_ACTOR_APPLY = 8, PH_ALL = 15 };

// ts_sac_learn_rows: the target pass of SAC._target_q / compute_nstep_return (n_step = 1) in front of the update, in the same
// launch sequence.  `returns` of sac_update_impl is then ignored (the critic-loss launch forms the return itself).
struct LearnExt {
    const float* obs_next; const double* rew; const uint8_t* terminated; double gamma;
    const float* noise_next;                                  // rsample() eps of a' ~ pi(s')
    float* noise_fill; int64_t noise_n; uint64_t seed, offset; // nullable: the packing launch draws ts_normal_fill(noise_fill, ...) first
    int noise_halves;                                         // 1: one stream of noise_n; 2: two streams of noise_n each (offset, offset + 1)
    float* returns_out;                                       // nullable
};

// `grads`: all phases in one call -> optional output [critic1 | critic2 | actor] (ts_sac_update's grads_out);
// single phases -> the exchange buffer: critic phases [critic1 | critic2], actor phases [actor | -mean(log_prob)].
int sac_update_impl(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const float* act,
                    const float* returns, const float* weight, const float* noise, int64_t B, int64_t obs_dim,
                    int64_t act_dim, const ts_sac_hparams* hp, float* stats_out5, float* weight_out, float* grads,
                    int phases, ts_stream_t stream, const int64_t* rows = nullptr, const LearnExt* ext = nullptr) {
    float* const grads_out = phases == PH_ALL ? grads : nullptr;
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_update: workspace is NULL");
    TS_REQUIRE(st && hp && obs && act && (returns || ext) && noise && stats_out5 && B >= 1 && adam_step >= 1,
               TS_ERR_INVALID_ARG, "ts_sac_update: bad argument");
    TS_REQUIRE(st->actor && st->critic1 && st->critic2 && st->critic1_old && st->critic2_old && st->actor_m &&
                   st->actor_v && st->critic1_m && st->critic1_v && st->critic2_m && st->critic2_v,
               TS_ERR_INVALID_ARG, "ts_sac_update: NULL state pointer");
    TS_REQUIRE(!hp->auto_alpha || (st->log_alpha && st->log_alpha_m && st->log_alpha_v), TS_ERR_INVALID_ARG,
               "ts_sac_update: auto alpha needs log_alpha and its Adam moments");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    const size_t slab = std::max(slab_floats(ma), slab_floats(mc));
    const int64_t pa = ma.total(), pc = mc.total();
    size_t bytes = al(4 * B * d.ka) + 4 * al(4 * B * d.kc) + 9 * hbytes(B, d) + 3 * al(4 * B * 64) +
                   2 * hbytes(B, d) + al(4 * slab) + al(4 * std::max(pa, pc)) + 6 * al(4 * B) +
                   al(4 * B * 3 * d.act) + 8192;
    const size_t spl = std::max(split_floats(ma), split_floats(mc));
    bytes += 2 * al(4 * spl) + 2 * hbytes(B, d) + al(4 * slab) + al(4 * pc) + 2 * al(4 * B * 32) +
             al(4 * 3 * ts::ceil_div(B, 256));
    if (ext) bytes += al(4 * B * d.ka) + al(4 * B * d.kc) + 2 * al(4 * B * 32) + al(4 * B) + 2 * hbytes(B, d) + al(4 * B * 64);
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    // (ext: the actor's two forward passes of an update -- on obs_next for the target, on obs for its own loss -- see the same
    // parameters, so they are ONE launch over [obs_next rows | obs rows]; x_a / aa are the second halves)
    float* x_a2 = c.take<float>((ext ? 2 : 1) * B * d.ka);
    float* x_a = ext ? x_a2 + B * d.ka : x_a2;
    float* x_c = c.take<float>(B * d.kc);          // [obs | buffer action]
    float* x_p = c.take<float>(B * d.kc);          // [obs | policy action]
    float* dx1 = c.take<float>(B * d.kc);
    const Act aa_all = take_act(c, ext ? 2 * B : B, 64, d.hid, d.depth);
    Act aa = aa_all;
    if (ext) {
        for (int i = 0; i < d.depth; ++i) aa.h[i] += B * d.hid;
        aa.out += B * 64;
    }
    const Act a1 = take_act(c, B, 32, d.hid, d.depth), a2 = take_act(c, B, 32, d.hid, d.depth);
    // upstream gradients of the head outputs: the loss kernels write the live columns, the zero padding of all five
    // comes from ONE memset at the start of the update
    float* zeroed = c.take<float>(B * 192);
    float* d_c1 = zeroed;                          // critic 1 loss      [B, 32]
    float* d_c2 = zeroed + B * 32;                 // critic 2 loss      [B, 32]
    float* d_q1 = zeroed + B * 64;                 // actor loss -> Q1   [B, 32]
    float* d_q2 = zeroed + B * 96;                 // actor loss -> Q2   [B, 32]
    float* d_head = zeroed + B * 128;              // policy backward    [B, 64]
    float* dx2 = c.take<float>(B * 64 > B * d.kc ? B * 64 : B * d.kc);
    BwdScratch sc, sc2;              // one set per stream (the two critics run concurrently)
    sc = take_scratch(c, B, d.hid, d.depth, slab);
    sc2 = take_scratch(c, B, d.hid, d.depth, slab);
    float* grad = c.take<float>(std::max(pa, pc));
    float* grad2 = c.take<float>(pc);
    float* split2 = c.take<float>(spl);
    float* td1 = c.take<float>(B); float* td2 = c.take<float>(B); float* logp = c.take<float>(B);
    float* keep = c.take<float>(B * 3 * d.act);
    float* norm_part = c.take<float>(1024);
    float* split = c.take<float>(spl);
    const unsigned gb = (unsigned)ts::ceil_div(B, 256);
    float* loss_part = c.take<float>(3 * (size_t)gb);      // {actor, critic1, critic2} x gb partial sums
    float* xn_a = x_a2; float* xn_c = nullptr; float* tq_out[2] = {nullptr, nullptr}; float* logp_n = nullptr;
    if (ext) {                                             // the target pass's inputs and results
        xn_c = c.take<float>(B * d.kc);
        tq_out[0] = c.take<float>(B * 32); tq_out[1] = c.take<float>(B * 32); logp_n = c.take<float>(B);
    }
    const float* log_alpha = hp->auto_alpha ? st->log_alpha : nullptr;
    float* g_out[3] = {grads_out, grads_out ? grads_out + pc : nullptr, grads_out ? grads_out + 2 * pc : nullptr};

    if (phases != PH_ALL) {      // exchange-buffer layout of the single phases
        g_out[0] = grads; g_out[1] = grads + pc; g_out[2] = grads;
    }
    // critic 1 & 2 (ddpg.py:279-285), each with its own Adam step.  The two chains are independent: critic 1 on the
    // caller's stream, critic 2 on the workspace's side stream (each of these GEMMs fills only part of the chip).
    hipStream_t side;
    if (int rc = twin_stream(ws, s, mc, &side)) return rc;
    // one stream: every reader of [obs | buffer action] (the critics' passes) is done before the policy kernel writes the
    // policy's action, so the actor pass reuses x_c instead of a second packed copy of the observations
    if (side == s && !ext) x_p = x_c;
    if (ext) {
        // one stream, one-launch chains (the caller checked): packing of both passes + the noise in one launch, then the target
        // pass -- actor on obs_next, a' and its log-probability, the two lagged critics in one launch; their outputs wait in
        // tq_out for the critic-loss launch
        TS_REQUIRE(side == s && phases == PH_ALL && rows, TS_ERR_UNSUPPORTED, "ts_sac_learn_rows: fused path needs one stream");
        Pack2Args pk{};
        pk.obs = obs; pk.act = act; pk.obs_next = ext->obs_next; pk.rows = rows; pk.B = B; pk.obs_dim = d.obs; pk.act_dim = d.act;
        pk.ka = d.ka; pk.kc = d.kc; pk.x_a = x_a; pk.x_c = x_c; pk.x_p = x_p; pk.xn_a = xn_a; pk.xn_c = xn_c;
        pk.pack_blocks = (unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256);
        pk.noise = ext->noise_fill; pk.noise_n = ext->noise_fill ? ext->noise_n : 0; pk.seed = ext->seed; pk.offset = ext->offset;
        pk.noise_halves = ext->noise_halves;
        const unsigned nz = (unsigned)ts::ceil_div(ts::ceil_div(pk.noise_n, 4), 256) * (unsigned)ext->noise_halves;
        hipLaunchKernelGGL(sac_pack2_kernel, dim3(2 * pk.pack_blocks + nz), dim3(256), 0, s, pk);
        const Mlp ma2 = make_mlp((int)(2 * B), d.ka, 64, d.hid, d.depth, d.fn);
        if (int rc = mlp_forward(s, ws, ma2, st->actor, x_a2, aa_all, split)) return rc;
        Policy2Args pa{};
        pa.head = aa_all.out; pa.noise_next = ext->noise_next; pa.noise = noise; pa.B = B; pa.A = d.act; pa.obs_dim = d.obs; pa.kc = d.kc;
        pa.bound = d.bound; pa.nb = (unsigned)ts::ceil_div(B * 32, 256);
        pa.xn_c = xn_c; pa.logp_n = logp_n; pa.x_p = x_p; pa.logp = logp; pa.keep = keep;
        hipLaunchKernelGGL(sac_policy2_kernel, dim3(2 * pa.nb), dim3(256), 0, s, pa);
        TS_LAUNCH_CHECK();
        // the lagged critics on (s', a') and the live critics on (s, a) in ONE launch of four networks (the lagged pair keeps no
        // hidden activations)
        const float* xs4[4] = {xn_c, xn_c, x_c, x_c};
        const float* w1[4]; const float* w2[4]; const float* w3[4];
        const float* p4[4] = {st->critic1_old, st->critic2_old, st->critic1, st->critic2};
        for (int k = 0; k < 4; ++k) { w1[k] = p4[k] + mc.off[0]; w2[k] = p4[k] + mc.off[1]; w3[k] = p4[k] + mc.off[2]; }
        float* h1[4] = {nullptr, nullptr, a1.h[0], a2.h[0]};
        float* h2[4] = {nullptr, nullptr, a1.h[1], a2.h[1]};
        float* o4[4] = {tq_out[0], tq_out[1], a1.out, a2.out};
        if (int rc = ts::mlp3_forward_nx(s, 4, xs4, mc.l[0].B, mc.l[0].IC, w1, w2, w3, mc.l[2].OC, h1, h2, o4, ws)) return rc;
    } else if (phases & PH_CRITIC_GRAD) {
        hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs, act,
                           B, d.obs, d.act, d.ka, d.kc, x_a, x_c, x_p == x_c ? (float*)nullptr : x_p, rows);
        // (`zeroed`: every kernel that fills a head-gradient buffer writes whole rows, padding columns included)
    }
    hipStream_t stq[2] = {s, side};
    float* crit[2] = {st->critic1, st->critic2};
    float* crit_m[2] = {st->critic1_m, st->critic2_m};
    float* crit_v[2] = {st->critic1_v, st->critic2_v};
    float* tds[2] = {td1, td2};
    const Act acts[2] = {a1, a2};
    float* dheads[2] = {d_c1, d_c2};
    float* splits[2] = {split, split2};
    float* gbuf[2] = {grad, grad2};
    const BwdScratch scs[2] = {sc, sc2};
    if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
    // both critics on one stream (the one-launch chains): one loss launch, the six weight-gradient GEMMs in one launch,
    // one Adam launch that also moves the lagged critics (nothing reads them again in this update)
    const bool twin_group = side == s && fused_backward(mc, false, 0, 0);
    const bool polyak_with_adam = twin_group && hp->critic_lr >= 0.0 && hp->tau > 0.0;
    // whole updates on the one-launch chains: each network group's slab sums + Adam (+ Polyak; + the alpha step behind the
    // actor's) are one launch (slab_adam_kernel; TS_SAC_SPLIT_ADAM=1 keeps the separate launches)
    const bool fuse_adam = twin_group && phases == PH_ALL && hp->critic_lr >= 0.0 && hp->actor_lr >= 0.0 &&
                           fused_backward(ma, false, 0, 0) && !getenv("TS_SAC_SPLIT_ADAM");
    auto critic_loss = [&](hipStream_t sk, int k0, int nk) {
        CriticLossArgs la{};
        for (int k = 0; k < nk; ++k) {
            la.q[k] = acts[k0 + k].out; la.td[k] = tds[k0 + k]; la.d_out[k] = dheads[k0 + k];
            la.part[k] = loss_part + (1 + k0 + k) * gb;
        }
        la.ret = returns; la.weight = weight; la.B = B;
        if (ext) {
            la.tq[0] = tq_out[0]; la.tq[1] = tq_out[1]; la.logp_n = logp_n; la.log_alpha = log_alpha; la.fixed_alpha = (float)hp->alpha;
            la.rew = ext->rew; la.terminated = ext->terminated; la.rows = rows; la.gamma = ext->gamma; la.ret_out = ext->returns_out;
        }
        hipLaunchKernelGGL(sac_critic_loss_mb_kernel, dim3(gb, (unsigned)nk), dim3(256), 0, sk, la);
    };
    if ((phases & PH_CRITIC_GRAD) && twin_group) {
        const float* dh2[2] = {dheads[0], dheads[1]};
        if (!ext)            // (ext: part of the four-network launch above)
            if (int rc = mlp_forward_twin(s, ws, mc, crit, x_c, acts, splits)) return rc;
        critic_loss(s, 0, 2);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward_twin(s, ws, mc, crit, x_c, acts, dh2, nullptr, 0, 0, scs)) return rc;
        const float* xs2[2] = {x_c, x_c};
        float* gk2[2] = {g_out[0] ? g_out[0] : gbuf[0], g_out[1] ? g_out[1] : gbuf[1]};
        if (fuse_adam) {       // slab sums, both Adam steps and the lagged critics' Polyak update in one launch
            float* lag[2] = {st->critic1_old, st->critic2_old};
            const AdamSpec sp{adam_step, hp->critic_lr, hp->beta1, hp->beta2, hp->adam_eps, hp->tau};
            if (int rc = mlp_weight_grads_adam(s, ws, 2, mc, xs2, acts, dh2, gk2, scs, crit, crit_m, crit_v, lag, sp, nullptr))
                return rc;
        } else if (int rc = mlp_weight_grads(s, ws, 2, mc, xs2, acts, dh2, gk2, scs)) return rc;
    }
    for (int k = 0; k < 2 && (phases & PH_CRITIC_GRAD) && !twin_group; ++k) {
        hipStream_t sk = stq[k];
        float* gk = g_out[k] ? g_out[k] : gbuf[k];
        if (int rc = mlp_forward(sk, ws, mc, crit[k], x_c, acts[k], splits[k])) return rc;
        critic_loss(sk, k, 1);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward(sk, ws, mc, crit[k], x_c, acts[k], dheads[k], gk, nullptr, 0, 0, scs[k], 1)) return rc;
    }
    if ((phases & PH_CRITIC_APPLY) && hp->critic_lr >= 0.0 && twin_group && !fuse_adam) {
        const float* gk2[2] = {g_out[0] ? g_out[0] : gbuf[0], g_out[1] ? g_out[1] : gbuf[1]};
        float* lag[2] = {st->critic1_old, st->critic2_old};
        if (int rc = ts::adam_step_multi(s, 2, crit, crit_m, crit_v, gk2, lag, pc, adam_step, hp->critic_lr, hp->beta1, hp->beta2,
                                         hp->adam_eps, hp->tau))
            return rc;
    }
    for (int k = 0; k < 2 && !twin_group; ++k) {
        hipStream_t sk = stq[k];
        float* gk = g_out[k] ? g_out[k] : gbuf[k];
        if (phases & PH_CRITIC_GRAD)
            if (int rc = mlp_backward(sk, ws, mc, crit[k], x_c, acts[k], dheads[k], gk, nullptr, 0, 0, scs[k], 2)) return rc;
        if ((phases & PH_CRITIC_APPLY) && hp->critic_lr >= 0.0)
            if (int rc = ts::adam_step(sk, crit[k], crit_m[k], crit_v[k], gk, pc, adam_step, hp->critic_lr, hp->beta1,
                                       hp->beta2, hp->adam_eps, 0.0, norm_part))
                return rc;
    }
    if (phases == PH_CRITIC_GRAD) {          // a single phase finishes its own loss statistics
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
        hipLaunchKernelGGL(sac_loss_finish_kernel, dim3(1), dim3(64), 0, s, loss_part + gb, stats_out5 + 1,
                           loss_part + 2 * gb, stats_out5 + 2, (int)gb, B);
        TS_LAUNCH_CHECK();
        return TS_OK;
    }
    if (!(phases & (PH_ACTOR_GRAD | PH_ACTOR_APPLY))) return ts::stream_wait(ws, side, s, 1);

    // actor (sac.py:308-315): a ~ pi(s) with the supplied noise, Q1(s, a), Q2(s, a) with the UPDATED critics
    float* ga = g_out[2] ? g_out[2] : grad;
    if (phases & PH_ACTOR_GRAD) {
        if (!ext) {          // (ext: done in front of the target pass, in the same two launches)
            if (int rc = mlp_forward(s, ws, ma, st->actor, x_a, aa, split)) return rc;
            hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B,
                               d.act, 64, d.obs, d.kc, d.bound, x_p, (float*)nullptr, logp, keep);
        }
        const Act a12[2] = {a1, a2};
        if (side == s) {                                                      // one stream: both critics in one launch
            if (int rc = mlp_forward_twin(s, ws, mc, crit, x_p, a12, splits)) return rc;
        } else {
            if (int rc = ts::stream_wait(ws, s, side, 1)) return rc;          // x_p ready; critic 2 is already updated there
            if (int rc = mlp_forward(side, ws, mc, st->critic2, x_p, a2, split2)) return rc;
            if (int rc = mlp_forward(s, ws, mc, st->critic1, x_p, a1, split)) return rc;
            if (int rc = ts::stream_wait(ws, side, s, 2)) return rc;
        }
        hipLaunchKernelGGL(sac_actor_loss_mb_kernel, dim3(gb), dim3(256), 0, s, a1.out, a2.out, logp, log_alpha,
                           (float)hp->alpha, B, d_q1, d_q2, loss_part);
        TS_LAUNCH_CHECK();
        if (side == s) {
            const float* dq[2] = {d_q1, d_q2};
            float* dxs[2] = {dx1, dx2};
            if (int rc = mlp_backward_twin(s, ws, mc, crit, x_p, a12, dq, dxs, d.obs, d.obs + d.act, scs)) return rc;
        } else {
            if (int rc = ts::stream_wait(ws, s, side, 3)) return rc;
            if (int rc = mlp_backward(side, ws, mc, st->critic2, x_p, a2, d_q2, nullptr, dx2, d.obs, d.obs + d.act, sc2)) return rc;
            if (int rc = mlp_backward(s, ws, mc, st->critic1, x_p, a1, d_q1, nullptr, dx1, d.obs, d.obs + d.act, sc)) return rc;
            if (int rc = ts::stream_wait(ws, side, s, 4)) return rc;
        }
        hipLaunchKernelGGL(sac_policy_bwd_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise,
                           keep, dx1, dx2, log_alpha, (float)hp->alpha, B, d.act, 64, d.obs, d.kc, d.bound, d_head);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward(s, ws, ma, st->actor, x_a, aa, d_head, ga, nullptr, 0, 0, sc, fuse_adam ? 1 : 3)) return rc;
        if (phases != PH_ALL) {      // the alpha step's only batch statistic, for the all-reduce; the actor loss
            hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(1024), 0, s, logp, B, grads + pa);
            hipLaunchKernelGGL(sac_loss_finish_kernel, dim3(1), dim3(64), 0, s, loss_part, stats_out5, (const float*)nullptr,
                               (float*)nullptr, (int)gb, B);
            TS_LAUNCH_CHECK();
        }
    }
    if (!(phases & PH_ACTOR_APPLY)) return TS_OK;
    if (hp->actor_lr >= 0.0 && !fuse_adam)
        if (int rc = ts::adam_step(s, st->actor, st->actor_m, st->actor_v, ga, pa, adam_step, hp->actor_lr, hp->beta1,
                                   hp->beta2, hp->adam_eps, 0.0, norm_part))
            return rc;

    // alpha (sac.py:317-319), batch.weight (sac.py:306), Polyak (sac.py:321)
    AlphaArgs aa2{};
    aa2.logp = logp; aa2.B = B; aa2.target_entropy = (float)hp->target_entropy;
    aa2.log_alpha = hp->auto_alpha ? st->log_alpha : nullptr; aa2.m = st->log_alpha_m; aa2.v = st->log_alpha_v;
    const double bc1 = 1.0 - pow(hp->beta1, (double)adam_step), bc2 = 1.0 - pow(hp->beta2, (double)adam_step);
    aa2.lr_step = (float)(hp->alpha_lr / bc1); aa2.beta1 = (float)hp->beta1; aa2.beta2 = (float)hp->beta2;
    aa2.omb1 = (float)(1.0 - hp->beta1); aa2.omb2 = (float)(1.0 - hp->beta2);
    aa2.bc2_sqrt = (float)sqrt(bc2); aa2.eps = (float)hp->adam_eps;
    aa2.alpha_loss = stats_out5 + 4; aa2.alpha_out = stats_out5 + 3; aa2.fixed_alpha = (float)hp->alpha;
    aa2.td1 = td1; aa2.td2 = td2; aa2.weight_out = weight_out;
    aa2.neg_mean_logp = phases != PH_ALL ? grads + pa : nullptr;
    aa2.loss_part = phases == PH_ALL ? loss_part : nullptr; aa2.losses = stats_out5; aa2.n_part = (int)gb;
    if (fuse_adam) {           // the actor's weight gradients, slab sums, Adam step and the alpha step: two launches
        const float* xa = x_a;
        const float* dh = d_head;
        float* pp[1] = {st->actor}; float* pm[1] = {st->actor_m}; float* pv[1] = {st->actor_v};
        const AdamSpec sp{adam_step, hp->actor_lr, hp->beta1, hp->beta2, hp->adam_eps, 0.0};
        if (int rc = mlp_weight_grads_adam(s, ws, 1, ma, &xa, &aa, &dh, &ga, &sc, pp, pm, pv, nullptr, sp, &aa2)) return rc;
    } else {
        hipLaunchKernelGGL(sac_alpha_kernel, dim3(1), dim3(1024), 0, s, aa2);
    }
    if (hp->tau > 0.0 && !polyak_with_adam)
        hipLaunchKernelGGL(polyak2_kernel, dim3((unsigned)ts::ceil_div(pc, 256)), dim3(256), 0, s, st->critic1_old,
                           st->critic1, st->critic2_old, st->critic2, pc, (float)hp->tau, (float)(1.0 - hp->tau));
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace

extern "C" {

int ts_sac_update(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const float* act,
                  const float* returns, const float* weight, const float* noise, int64_t B, int64_t obs_dim,
                  int64_t act_dim, const ts_sac_hparams* hp, float* stats_out5, float* weight_out, float* grads_out,
                  ts_stream_t stream) {
    return sac_update_impl(ws, st, adam_step, obs, act, returns, weight, noise, B, obs_dim, act_dim, hp, stats_out5,
                           weight_out, grads_out, PH_ALL, stream);
}

int ts_sac_update_rows(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs_buf, const float* act_buf,
                       const int64_t* rows, const float* returns, const float* weight, const float* noise, int64_t B,
                       int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp, float* stats_out5, float* weight_out,
                       ts_stream_t stream) {
    TS_REQUIRE(rows != nullptr, TS_ERR_INVALID_ARG, "ts_sac_update_rows: rows is NULL");
    return sac_update_impl(ws, st, adam_step, obs_buf, act_buf, returns, weight, noise, B, obs_dim, act_dim, hp, stats_out5,
                           weight_out, nullptr, PH_ALL, stream, rows);
}

int ts_sac_learn_rows(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const ts_sac_replay* replay,
                      const int64_t* rows, const float* weight, float* noise2, int fill_noise, uint64_t noise_seed,
                      uint64_t noise_offset, int64_t B, int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp, double gamma,
                      float* returns_out, float* stats_out5, float* weight_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_learn_rows: workspace is NULL");
    TS_REQUIRE(st && hp && replay && replay->obs && replay->act && replay->obs_next && replay->rew && replay->terminated && rows &&
                   noise2 && returns_out && stats_out5 && B >= 1 && adam_step >= 1,
               TS_ERR_INVALID_ARG, "ts_sac_learn_rows: bad argument");
    TS_REQUIRE(fill_noise >= 0 && fill_noise <= 2, TS_ERR_INVALID_ARG, "ts_sac_learn_rows: fill_noise must be 0, 1 or 2");
    TS_REQUIRE(st->actor && st->critic1_old && st->critic2_old && (!hp->auto_alpha || st->log_alpha), TS_ERR_INVALID_ARG,
               "ts_sac_learn_rows: NULL state pointer");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    const Mlp mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn), ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn);
    float* noise_next = noise2;
    float* noise_upd = noise2 + B * act_dim;
    // the fused sequence rides on the one-stream, one-launch chains; every other trunk (depth, tanh, widths outside the fused
    // kernels, TS_TWIN_STREAMS) runs the same two entry points one after the other -- same results, no launch saved
    static const bool twin_streams = getenv("TS_TWIN_STREAMS") != nullptr, unfused = getenv("TS_SAC_LEARN_UNFUSED") != nullptr;
    const bool fused = !unfused && !twin_streams && mc.three() && ts::mlp3_supported(mc.l[0].IC, mc.l[0].OC, mc.l[2].OC) &&
                       fused_backward(mc, false, 0, 0) && ma.three() && ts::mlp3_supported(ma.l[0].IC, ma.l[0].OC, ma.l[2].OC);
    if (!fused) {
        if (fill_noise == 1)
            if (int rc = ts_normal_fill(noise2, 2 * B * act_dim, noise_seed, noise_offset, stream)) return rc;
        if (fill_noise == 2) {
            if (int rc = ts_normal_fill(noise2, B * act_dim, noise_seed, noise_offset, stream)) return rc;
            if (int rc = ts_normal_fill(noise2 + B * act_dim, B * act_dim, noise_seed, noise_offset + 1, stream)) return rc;
        }
        if (int rc = sac_target_impl(ws, st->actor, st->critic1_old, st->critic2_old, hp->auto_alpha ? st->log_alpha : nullptr, hp->alpha,
                                     replay->obs_next, noise_next, B, obs_dim, act_dim, returns_out, stream, rows, replay->rew,
                                     replay->terminated, gamma))
            return rc;
        return sac_update_impl(ws, st, adam_step, replay->obs, replay->act, returns_out, weight, noise_upd, B, obs_dim, act_dim, hp,
                               stats_out5, weight_out, nullptr, PH_ALL, stream, rows);
    }
    LearnExt ext{};
    ext.obs_next = replay->obs_next; ext.rew = replay->rew; ext.terminated = replay->terminated; ext.gamma = gamma;
    ext.noise_next = noise_next; ext.noise_fill = fill_noise ? noise2 : nullptr;
    ext.noise_halves = fill_noise == 2 ? 2 : 1; ext.noise_n = (fill_noise == 2 ? 1 : 2) * B * act_dim;
    ext.seed = noise_seed; ext.offset = noise_offset; ext.returns_out = returns_out;
    return sac_update_impl(ws, st, adam_step, replay->obs, replay->act, nullptr, weight, noise_upd, B, obs_dim, act_dim, hp, stats_out5,
                           weight_out, nullptr, PH_ALL, stream, rows, &ext);
}

int ts_sac_update_phase(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const float* act,
                        const float* returns, const float* weight, const float* noise, int64_t B, int64_t obs_dim,
                        int64_t act_dim, const ts_sac_hparams* hp, int phase, float* stats_out5, float* weight_out,
                        float* grads, ts_stream_t stream) {
    TS_REQUIRE(phase == PH_CRITIC_GRAD || phase == PH_CRITIC_APPLY || phase == PH_ACTOR_GRAD || phase == PH_ACTOR_APPLY,
               TS_ERR_INVALID_ARG, "ts_sac_update_phase: phase must be 1, 2, 4 or 8");
    TS_REQUIRE(grads != nullptr, TS_ERR_INVALID_ARG, "ts_sac_update_phase: the exchange buffer is NULL");
    return sac_update_impl(ws, st, adam_step, obs, act, returns, weight, noise, B, obs_dim, act_dim, hp, stats_out5,
                           weight_out, grads, phase, stream);
}

// ---- TD3 / DDPG ------------------------------------------------------------------------------------------------
int ts_td3_layout(int64_t obs_dim, int64_t act_dim, int64_t* h_out4) { return ts_td3_layout_h(obs_dim, act_dim, 0, h_out4); }

int ts_td3_layout_h(int64_t obs_dim, int64_t act_dim, int64_t hidden, int64_t* h_out4) {
    Dims d;
    if (int rc = make_dims_h(obs_dim, act_dim, hidden, &d)) return rc;
    TS_REQUIRE(h_out4, TS_ERR_INVALID_ARG, "ts_td3_layout: NULL output");
    h_out4[0] = d.ka; h_out4[1] = d.kc;
    h_out4[2] = make_mlp(1, d.ka, 32, d.hid, d.depth, d.fn).total(); h_out4[3] = make_mlp(1, d.kc, 32, d.hid, d.depth, d.fn).total();
    return TS_OK;
}

int ts_td3_policy_forward(ts_workspace* ws, const float* actor, const float* obs, int64_t B, int64_t obs_dim,
                          int64_t act_dim, double max_action, float* act_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_td3_policy_forward: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && obs && act_out, TS_ERR_INVALID_ARG, "ts_td3_policy_forward: bad argument");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 32, d.hid, d.depth, d.fn);
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + 3 * hbytes(B, d) + al(4 * split_floats(ma)) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    const Act aa = take_act(c, B, 32, d.hid, d.depth);
    float* split = c.take<float>(split_floats(ma));
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, (float*)nullptr, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, ma, actor, x_a, aa, split)) return rc;
    hipLaunchKernelGGL(det_policy_kernel, dim3((unsigned)ts::ceil_div(B * d.act, 256)), dim3(256), 0, s, aa.out,
                       (const float*)nullptr, B, d.act, (float)max_action, 0.f, 0.f, d.obs, d.kc, (float*)nullptr, act_out,
                       (float*)nullptr);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_td3_target_q(ts_workspace* ws, const float* actor_old, const float* critic1_old, const float* critic2_old,
                    const float* obs_next, const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim,
                    double max_action, double policy_noise, double noise_clip, float* out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_td3_target_q: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor_old && critic1_old && obs_next && out, TS_ERR_INVALID_ARG, "ts_td3_target_q: bad argument");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream);
    const Mlp ma = make_mlp((int)B, d.ka, 32, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    const size_t spl = std::max(split_floats(ma), split_floats(mc));
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + al(4 * B * d.kc) + 9 * hbytes(B, d) + 2 * al(4 * spl) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    float* x_c = c.take<float>(B * d.kc);
    const Act aa = take_act(c, B, 32, d.hid, d.depth), a1 = take_act(c, B, 32, d.hid, d.depth), a2 = take_act(c, B, 32, d.hid, d.depth);
    float* split = c.take<float>(spl);
    float* split2 = c.take<float>(spl);
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs_next,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, x_c, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, ma, actor_old, x_a, aa, split)) return rc;
    hipLaunchKernelGGL(det_policy_kernel, dim3((unsigned)ts::ceil_div(B * d.act, 256)), dim3(256), 0, s, aa.out, noise, B,
                       d.act, (float)max_action, (float)policy_noise, (float)noise_clip, d.obs, d.kc, x_c, (float*)nullptr,
                       (float*)nullptr);
    TS_LAUNCH_CHECK();
    if (critic2_old) {                                                  // the two lagged critics side by side
        hipStream_t side;
        if (int rc = twin_stream(ws, s, mc, &side)) return rc;
        if (side == s) {
            const float* pp[2] = {critic1_old, critic2_old};
            const Act aa2[2] = {a1, a2};
            float* sp[2] = {split, split2};
            if (int rc = mlp_forward_twin(s, ws, mc, pp, x_c, aa2, sp)) return rc;
        } else {
            if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
            if (int rc = mlp_forward(side, ws, mc, critic2_old, x_c, a2, split2)) return rc;
            if (int rc = mlp_forward(s, ws, mc, critic1_old, x_c, a1, split)) return rc;
            if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
        }
    } else {
        if (int rc = mlp_forward(s, ws, mc, critic1_old, x_c, a1, split)) return rc;
    }
    hipLaunchKernelGGL(td3_target_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, a1.out,
                       critic2_old ? a2.out : (const float*)nullptr, B, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_td3_update(ts_workspace* ws, const ts_td3_state* st, int64_t critic_step, int64_t actor_step, const float* obs,
                  const float* act, const float* returns, const float* weight, int64_t B, int64_t obs_dim,
                  int64_t act_dim, const ts_td3_hparams* hp, float* stats_out3, float* weight_out, float* grads_out,
                  ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_td3_update: workspace is NULL");
    TS_REQUIRE(st && hp && obs && act && returns && stats_out3 && B >= 1 && critic_step >= 1 && actor_step >= 1,
               TS_ERR_INVALID_ARG, "ts_td3_update: bad argument");
    TS_REQUIRE(st->actor && st->actor_m && st->actor_v && st->critic1 && st->critic1_m && st->critic1_v &&
                   st->actor_old && st->critic1_old, TS_ERR_INVALID_ARG, "ts_td3_update: NULL state pointer");
    const bool twin = st->critic2 != nullptr;
    TS_REQUIRE(!twin || (st->critic2_m && st->critic2_v && st->critic2_old), TS_ERR_INVALID_ARG,
               "ts_td3_update: incomplete second critic");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream), side;
    const Mlp ma = make_mlp((int)B, d.ka, 32, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    if (int rc = twin_stream(ws, s, mc, &side)) return rc;
    const size_t slab = std::max(slab_floats(ma), slab_floats(mc));
    const size_t spl = std::max(split_floats(ma), split_floats(mc));
    const int64_t pa = ma.total(), pc = mc.total();
    const unsigned gb = (unsigned)ts::ceil_div(B, 256);
    const size_t bytes = al(4 * B * d.ka) + 3 * al(4 * B * d.kc) + 9 * hbytes(B, d) + 4 * al(4 * B * 32) + al(4 * 3 * gb) +
                         4 * hbytes(B, d) + 2 * al(4 * slab) + 2 * al(4 * std::max(pa, pc)) + 2 * al(4 * B) +
                         al(4 * B * d.act) + 2 * al(4 * spl) + 8192;
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    float* x_c = c.take<float>(B * d.kc);
    float* x_p = c.take<float>(B * d.kc);
    float* dx1 = c.take<float>(B * d.kc);
    const Act aa = take_act(c, B, 32, d.hid, d.depth), a1 = take_act(c, B, 32, d.hid, d.depth), a2 = take_act(c, B, 32, d.hid, d.depth);
    // upstream gradients of the head outputs [B, 32]: live column written by the loss kernels, zero padding from ONE
    // memset: {critic 1 loss, critic 2 loss, actor loss -> Q1, policy backward}
    float* zeroed = c.take<float>(B * 128);
    float* dheads[2] = {zeroed, zeroed + B * 32};
    float* d_q = zeroed + B * 64;
    float* d_pol = zeroed + B * 96;
    float* loss_part = c.take<float>(3 * (size_t)gb);      // {actor, critic1, critic2} x gb partial sums
    BwdScratch scs[2];
    for (int k = 0; k < 2; ++k) scs[k] = take_scratch(c, B, d.hid, d.depth, slab);
    float* gbuf[2] = {c.take<float>(std::max(pa, pc)), c.take<float>(std::max(pa, pc))};
    float* tds[2] = {c.take<float>(B), c.take<float>(B)};
    float* keep = c.take<float>(B * d.act);
    float* splits[2] = {c.take<float>(spl), c.take<float>(spl)};
    float* norm_part = c.take<float>(1024);
    float* g_out[3] = {grads_out, grads_out ? grads_out + pc : nullptr, grads_out ? grads_out + 2 * pc : nullptr};

    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs, act, B,
                       d.obs, d.act, d.ka, d.kc, x_a, x_c, hp->update_actor ? x_p : (float*)nullptr);
    // (`zeroed`: every kernel that fills a head-gradient buffer writes whole rows, padding columns included)
    // critics (ddpg.py:279-285): critic 1 on the caller's stream, critic 2 on the side stream
    hipStream_t stq[2] = {s, side};
    float* crit[2] = {st->critic1, st->critic2};
    float* crit_m[2] = {st->critic1_m, st->critic2_m};
    float* crit_v[2] = {st->critic1_v, st->critic2_v};
    const Act acts[2] = {a1, a2};
    if (twin)
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
    // both critics on one stream (the one-launch chains): one loss launch, six weight-gradient GEMMs in one launch, one
    // Adam launch that also moves the lagged critics on the updates that move lagged networks (td3.py:215-219)
    const bool twin_group = twin && side == s && fused_backward(mc, false, 0, 0);
    const bool polyak_with_adam = twin_group && hp->critic_lr >= 0.0 && hp->tau > 0.0 && hp->update_actor;
    auto critic_loss = [&](hipStream_t sk, int k0, int nk) {
        CriticLossArgs la{};
        for (int k = 0; k < nk; ++k) {
            la.q[k] = acts[k0 + k].out; la.td[k] = tds[k0 + k]; la.d_out[k] = dheads[k0 + k];
            la.part[k] = loss_part + (1 + k0 + k) * gb;
        }
        la.ret = returns; la.weight = weight; la.B = B;
        hipLaunchKernelGGL(sac_critic_loss_mb_kernel, dim3(gb, (unsigned)nk), dim3(256), 0, sk, la);
    };
    if (twin_group) {
        const float* dh2[2] = {dheads[0], dheads[1]};
        if (int rc = mlp_forward_twin(s, ws, mc, crit, x_c, acts, splits)) return rc;
        critic_loss(s, 0, 2);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward_twin(s, ws, mc, crit, x_c, acts, dh2, nullptr, 0, 0, scs)) return rc;
        const float* xs2[2] = {x_c, x_c};
        float* gk2[2] = {g_out[0] ? g_out[0] : gbuf[0], g_out[1] ? g_out[1] : gbuf[1]};
        if (int rc = mlp_weight_grads(s, ws, 2, mc, xs2, acts, dh2, gk2, scs)) return rc;
        if (hp->critic_lr >= 0.0) {
            const float* gc2[2] = {gk2[0], gk2[1]};
            float* lag[2] = {st->critic1_old, st->critic2_old};
            if (int rc = ts::adam_step_multi(s, 2, crit, crit_m, crit_v, gc2, lag, pc, critic_step, hp->critic_lr, hp->beta1,
                                             hp->beta2, hp->adam_eps, polyak_with_adam ? hp->tau : 0.0))
                return rc;
        }
    }
    for (int k = 0; k < (twin ? 2 : 1) && !twin_group; ++k) {
        hipStream_t sk = stq[k];
        if (int rc = mlp_forward(sk, ws, mc, crit[k], x_c, acts[k], splits[k])) return rc;
        critic_loss(sk, k, 1);
        TS_LAUNCH_CHECK();
        float* gk = g_out[k] ? g_out[k] : gbuf[k];
        if (int rc = mlp_backward(sk, ws, mc, crit[k], x_c, acts[k], dheads[k], gk, nullptr, 0, 0, scs[k])) return rc;
        if (hp->critic_lr >= 0.0)
            if (int rc = ts::adam_step(sk, crit[k], crit_m[k], crit_v[k], gk, pc, critic_step, hp->critic_lr, hp->beta1,
                                       hp->beta2, hp->adam_eps, 0.0, norm_part))
                return rc;
    }
    if (twin)
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    hipLaunchKernelGGL(sac_loss_finish_kernel, dim3(1), dim3(64), 0, s, loss_part + gb, stats_out3 + 1,
                       twin ? loss_part + 2 * gb : (const float*)nullptr, stats_out3 + 2, (int)gb, B);
    if (weight_out)
        hipLaunchKernelGGL(td3_weight_kernel, dim3(16), dim3(1024), 0, s, tds[0], twin ? tds[1] : (const float*)nullptr, B,
                           weight_out);                                          // td3.py:212 / ddpg.py:405
    if (hp->update_actor) {                                                      // td3.py:215-219, ddpg.py:406-409
        if (int rc = mlp_forward(s, ws, ma, st->actor, x_a, aa, splits[0])) return rc;
        hipLaunchKernelGGL(det_policy_kernel, dim3((unsigned)ts::ceil_div(B * d.act, 256)), dim3(256), 0, s, aa.out,
                           (const float*)nullptr, B, d.act, (float)hp->max_action, 0.f, 0.f, d.obs, d.kc, x_p,
                           (float*)nullptr, keep);
        if (int rc = mlp_forward(s, ws, mc, st->critic1, x_p, a1, splits[0])) return rc;
        hipLaunchKernelGGL(det_actor_loss_kernel, dim3(1 + (unsigned)ts::ceil_div(B, 1024)), dim3(1024), 0, s, a1.out, B, d_q, stats_out3);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward(s, ws, mc, st->critic1, x_p, a1, d_q, nullptr, dx1, d.obs, d.obs + d.act, scs[0]))
            return rc;
        hipLaunchKernelGGL(det_policy_bwd_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, dx1, keep, B,
                           d.act, (float)hp->max_action, d.obs, d.kc, d_pol);
        TS_LAUNCH_CHECK();
        float* ga = g_out[2] ? g_out[2] : gbuf[0];
        if (int rc = mlp_backward(s, ws, ma, st->actor, x_a, aa, d_pol, ga, nullptr, 0, 0, scs[0])) return rc;
        if (hp->actor_lr >= 0.0)
            if (int rc = ts::adam_step(s, st->actor, st->actor_m, st->actor_v, ga, pa, actor_step, hp->actor_lr, hp->beta1,
                                       hp->beta2, hp->adam_eps, 0.0, norm_part))
                return rc;
        if (hp->tau > 0.0) {                                                     // lagged_network.py:17-18
            const float tau = (float)hp->tau, omt = (float)(1.0 - hp->tau);
            hipLaunchKernelGGL(polyak1_kernel, dim3((unsigned)ts::ceil_div(pa, 256)), dim3(256), 0, s, st->actor_old,
                               st->actor, pa, tau, omt);
            if (twin && polyak_with_adam) {
            } else if (twin)
                hipLaunchKernelGGL(polyak2_kernel, dim3((unsigned)ts::ceil_div(pc, 256)), dim3(256), 0, s, st->critic1_old,
                                   st->critic1, st->critic2_old, st->critic2, pc, tau, omt);
            else
                hipLaunchKernelGGL(polyak1_kernel, dim3((unsigned)ts::ceil_div(pc, 256)), dim3(256), 0, s, st->critic1_old,
                                   st->critic1, pc, tau, omt);
        }
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// ---- DiscreteSAC ---------------------------------------------------------------------------------------------------
int ts_dsac_layout(int64_t obs_dim, int64_t n_act, int64_t hidden, int64_t* h_out3) {
    DDims d;
    if (int rc = make_ddims(obs_dim, n_act, hidden, &d, 0)) return rc;
    TS_REQUIRE(h_out3, TS_ERR_INVALID_ARG, "ts_dsac_layout: NULL output");
    h_out3[0] = d.ka; h_out3[1] = d.hw; h_out3[2] = d.P;
    return TS_OK;
}

int ts_dsac_policy_forward(ts_workspace* ws, const float* actor, const float* obs, int64_t B, int64_t obs_dim,
                           int64_t n_act, int64_t hidden, float* logits_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dsac_policy_forward: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && obs && logits_out, TS_ERR_INVALID_ARG, "ts_dsac_policy_forward: bad argument");
    DDims d;
    if (int rc = make_ddims(obs_dim, n_act, hidden, &d, ws->mlp_depth)) return rc;
    d.fn = ws->mlp_act_tanh ? TS_NET_ACT_TANH : TS_NET_ACT_RELU;
    hipStream_t s = ts::as_stream(stream);
    const Mlp m = make_mlp((int)B, d.ka, d.hw, d.hid, d.depth, d.fn);
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + 2 * hbytes(B, d) + al(4 * B * d.hw) + al(4 * split_floats(m)) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.take<float>(B * d.ka);
    const Act a = take_act(c, B, d.hw, d.hid, d.depth);
    float* split = c.take<float>(split_floats(m));
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * d.ka / 4, 256)), dim3(256), 0, s, obs,
                       (const float*)nullptr, B, d.obs, 0, d.ka, 0, x, (float*)nullptr, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, m, actor, x, a, split)) return rc;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)ts::ceil_div(B * d.act, 256)), dim3(256), 0, s, a.out, B, d.act, d.hw,
                       logits_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_dsac_target_q(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                     const float* log_alpha, double fixed_alpha, const float* obs_next, int64_t B, int64_t obs_dim,
                     int64_t n_act, int64_t hidden, float* out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dsac_target_q: workspace is NULL");
    TS_REQUIRE(B >= 1 && actor && critic1_old && critic2_old && obs_next && out, TS_ERR_INVALID_ARG,
               "ts_dsac_target_q: bad argument");
    DDims d;
    if (int rc = make_ddims(obs_dim, n_act, hidden, &d, ws->mlp_depth)) return rc;
    d.fn = ws->mlp_act_tanh ? TS_NET_ACT_TANH : TS_NET_ACT_RELU;
    hipStream_t s = ts::as_stream(stream), side;
    const Mlp m = make_mlp((int)B, d.ka, d.hw, d.hid, d.depth, d.fn);
    const size_t spl = split_floats(m);
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + 6 * hbytes(B, d) + 3 * al(4 * B * d.hw) + 2 * al(4 * spl) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.take<float>(B * d.ka);
    const Act aa = take_act(c, B, d.hw, d.hid, d.depth), a1 = take_act(c, B, d.hw, d.hid, d.depth), a2 = take_act(c, B, d.hw, d.hid, d.depth);
    float* split = c.take<float>(spl);
    float* split2 = c.take<float>(spl);
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * d.ka / 4, 256)), dim3(256), 0, s, obs_next,
                       (const float*)nullptr, B, d.obs, 0, d.ka, 0, x, (float*)nullptr, (float*)nullptr);
    TS_LAUNCH_CHECK();
    if (dsac_one_stream(m)) {        // the three networks read the same rows: one launch
        const float* p3[3] = {actor, critic1_old, critic2_old};
        const Act a3[3] = {aa, a1, a2};
        float* sp[2] = {split, split2};
        if (int rc = mlp_forward_multi(s, ws, m, 3, p3, x, a3, sp)) return rc;
    } else {
        if (int rc = ts::side_stream(ws, s, &side)) return rc;
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
        if (int rc = mlp_forward(side, ws, m, critic2_old, x, a2, split2)) return rc;
        if (int rc = mlp_forward(s, ws, m, actor, x, aa, split)) return rc;
        if (int rc = mlp_forward(s, ws, m, critic1_old, x, a1, split)) return rc;
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    }
    if (d.hw == 32)
        hipLaunchKernelGGL(dsac_target32_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, a1.out, a2.out,
                           log_alpha, (float)fixed_alpha, B, d.act, out);
    else
        hipLaunchKernelGGL(dsac_target_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, aa.out, a1.out, a2.out,
                           log_alpha, (float)fixed_alpha, B, d.act, d.hw, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_dsac_update(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const int64_t* act,
                   const float* returns, const float* weight, int64_t B, int64_t obs_dim, int64_t n_act, int64_t hidden,
                   const ts_sac_hparams* hp, float* stats_out5, float* weight_out, float* grads_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dsac_update: workspace is NULL");
    TS_REQUIRE(st && hp && obs && act && returns && stats_out5 && B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG,
               "ts_dsac_update: bad argument");
    TS_REQUIRE(st->actor && st->critic1 && st->critic2 && st->critic1_old && st->critic2_old && st->actor_m &&
                   st->actor_v && st->critic1_m && st->critic1_v && st->critic2_m && st->critic2_v,
               TS_ERR_INVALID_ARG, "ts_dsac_update: NULL state pointer");
    TS_REQUIRE(!hp->auto_alpha || (st->log_alpha && st->log_alpha_m && st->log_alpha_v), TS_ERR_INVALID_ARG,
               "ts_dsac_update: auto alpha needs log_alpha and its Adam moments");
    DDims d;
    if (int rc = make_ddims(obs_dim, n_act, hidden, &d, ws->mlp_depth)) return rc;
    d.fn = ws->mlp_act_tanh ? TS_NET_ACT_TANH : TS_NET_ACT_RELU;
    hipStream_t s = ts::as_stream(stream), side;
    const Mlp m = make_mlp((int)B, d.ka, d.hw, d.hid, d.depth, d.fn);
    const size_t slab = slab_floats(m), spl = split_floats(m);
    const int64_t P = d.P;
    const size_t bytes = al(4 * B * d.ka) + 10 * hbytes(B, d) + 6 * al(4 * B * d.hw) + 2 * al(4 * slab) +
                         2 * al(4 * P) + 2 * al(4 * spl) + 4 * al(4 * B) + al(4 * 1024) + 4096;
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x = c.take<float>(B * d.ka);
    const Act aa = take_act(c, B, d.hw, d.hid, d.depth);
    const Act acts[2] = {take_act(c, B, d.hw, d.hid, d.depth), take_act(c, B, d.hw, d.hid, d.depth)};
    float* dheads[3] = {c.take<float>(B * d.hw), c.take<float>(B * d.hw), c.take<float>(B * d.hw)};   // critic1, critic2, actor
    BwdScratch scs[2];
    for (int k = 0; k < 2; ++k) scs[k] = take_scratch(c, B, d.hid, d.depth, slab);
    float* gbuf[2] = {c.take<float>(P), c.take<float>(P)};
    float* splits[2] = {c.take<float>(spl), c.take<float>(spl)};
    float* tds[2] = {c.take<float>(B), c.take<float>(B)};
    float* neg_ent = c.take<float>(B);
    float* fval = c.take<float>(B);
    float* norm_part = c.take<float>(1024);
    const float* log_alpha = hp->auto_alpha ? st->log_alpha : nullptr;
    float* g_out[3] = {grads_out, grads_out ? grads_out + P : nullptr, grads_out ? grads_out + 2 * P : nullptr};

    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * d.ka / 4, 256)), dim3(256), 0, s, obs,
                       (const float*)nullptr, B, d.obs, 0, d.ka, 0, x, (float*)nullptr, (float*)nullptr);
    TS_LAUNCH_CHECK();
    float* crit[2] = {st->critic1, st->critic2};
    float* crit_m[2] = {st->critic1_m, st->critic2_m};
    float* crit_v[2] = {st->critic1_v, st->critic2_v};
    float* gk2[2] = {g_out[0] ? g_out[0] : gbuf[0], g_out[1] ? g_out[1] : gbuf[1]};
    const bool one_stream = dsac_one_stream(m);
    const bool polyak_with_adam = one_stream && hp->critic_lr >= 0.0 && hp->tau > 0.0;
    auto critic_loss = [&](hipStream_t sk, int k0, int nk) {
        DsacLossArgs la{};
        for (int k = 0; k < nk; ++k) {
            la.q[k] = acts[k0 + k].out; la.td[k] = tds[k0 + k]; la.d_out[k] = dheads[k0 + k];
            la.loss[k] = stats_out5 + 1 + k0 + k;
        }
        hipLaunchKernelGGL(dsac_critic_loss_kernel, dim3((unsigned)nk, 1 + (unsigned)ts::ceil_div(B * d.hw, 1024)), dim3(1024), 0, sk, la,
                           act, returns, weight, B, d.hw);
    };
    if (one_stream) {
        // both critics per launch: forward, loss, input-gradient chains, the six weight-gradient GEMMs, Adam (which also
        // moves the lagged critics: nothing reads them again in this update); then the updated critics and the actor
        // (discrete_sac.py:176-184) in one three-network forward
        const float* dh2[2] = {dheads[0], dheads[1]};
        const float* xs2[2] = {x, x};
        if (int rc = mlp_forward_twin(s, ws, m, crit, x, acts, splits)) return rc;
        critic_loss(s, 0, 2);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward_twin(s, ws, m, crit, x, acts, dh2, nullptr, 0, 0, scs)) return rc;
        if (int rc = mlp_weight_grads(s, ws, 2, m, xs2, acts, dh2, gk2, scs)) return rc;
        if (hp->critic_lr >= 0.0) {
            float* lag[2] = {st->critic1_old, st->critic2_old};
            if (int rc = ts::adam_step_multi(s, 2, crit, crit_m, crit_v, gk2, lag, P, adam_step, hp->critic_lr, hp->beta1,
                                             hp->beta2, hp->adam_eps, hp->tau))
                return rc;
        }
        const float* p3[3] = {st->critic1, st->critic2, st->actor};
        const Act a3[3] = {acts[0], acts[1], aa};
        if (int rc = mlp_forward_multi(s, ws, m, 3, p3, x, a3, splits)) return rc;
    } else {
        // critic 1 on the caller's stream, critic 2 on the side stream (independent chains, as ts_sac_update)
        if (int rc = ts::side_stream(ws, s, &side)) return rc;
        hipStream_t stq[2] = {s, side};
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
        for (int k = 0; k < 2; ++k) {
            hipStream_t sk = stq[k];
            if (int rc = mlp_forward(sk, ws, m, crit[k], x, acts[k], splits[k])) return rc;
            critic_loss(sk, k, 1);
            TS_LAUNCH_CHECK();
            if (int rc = mlp_backward(sk, ws, m, crit[k], x, acts[k], dheads[k], gk2[k], nullptr, 0, 0, scs[k])) return rc;
            if (hp->critic_lr >= 0.0)
                if (int rc = ts::adam_step(sk, crit[k], crit_m[k], crit_v[k], gk2[k], P, adam_step, hp->critic_lr, hp->beta1,
                                           hp->beta2, hp->adam_eps, 0.0, norm_part + 512 * k))
                    return rc;
        }
        // actor with the UPDATED critics (discrete_sac.py:176-184)
        if (int rc = mlp_forward(side, ws, m, st->critic2, x, acts[1], splits[1])) return rc;
        if (int rc = mlp_forward(s, ws, m, st->critic1, x, acts[0], splits[0])) return rc;
        if (int rc = mlp_forward(s, ws, m, st->actor, x, aa, splits[0])) return rc;
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    }
    if (d.hw == 32)
        hipLaunchKernelGGL(dsac_actor32_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, acts[0].out,
                           acts[1].out, log_alpha, (float)hp->alpha, B, d.act, dheads[2], neg_ent, fval);
    else
        hipLaunchKernelGGL(dsac_actor_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, aa.out, acts[0].out,
                           acts[1].out, log_alpha, (float)hp->alpha, B, d.act, d.hw, dheads[2], neg_ent, fval);
    hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(1024), 0, s, fval, B, stats_out5);
    TS_LAUNCH_CHECK();
    float* ga = g_out[2] ? g_out[2] : gbuf[0];
    if (int rc = mlp_backward(s, ws, m, st->actor, x, aa, dheads[2], ga, nullptr, 0, 0, scs[0])) return rc;
    if (hp->actor_lr >= 0.0)
        if (int rc = ts::adam_step(s, st->actor, st->actor_m, st->actor_v, ga, P, adam_step, hp->actor_lr, hp->beta1,
                                   hp->beta2, hp->adam_eps, 0.0, norm_part))
            return rc;
    // alpha (sac.py:203-209 with entropy = H), batch.weight (discrete_sac.py:174), Polyak
    AlphaArgs al2{};
    al2.logp = neg_ent; al2.B = B; al2.target_entropy = (float)hp->target_entropy;
    al2.log_alpha = hp->auto_alpha ? st->log_alpha : nullptr; al2.m = st->log_alpha_m; al2.v = st->log_alpha_v;
    const double bc1 = 1.0 - pow(hp->beta1, (double)adam_step), bc2 = 1.0 - pow(hp->beta2, (double)adam_step);
    al2.lr_step = (float)(hp->alpha_lr / bc1); al2.beta1 = (float)hp->beta1; al2.beta2 = (float)hp->beta2;
    al2.omb1 = (float)(1.0 - hp->beta1); al2.omb2 = (float)(1.0 - hp->beta2);
    al2.bc2_sqrt = (float)sqrt(bc2); al2.eps = (float)hp->adam_eps;
    al2.alpha_loss = stats_out5 + 4; al2.alpha_out = stats_out5 + 3; al2.fixed_alpha = (float)hp->alpha;
    al2.td1 = tds[0]; al2.td2 = tds[1]; al2.weight_out = weight_out;
    hipLaunchKernelGGL(sac_alpha_kernel, dim3(1), dim3(1024), 0, s, al2);
    if (hp->tau > 0.0 && !polyak_with_adam)
        hipLaunchKernelGGL(polyak2_kernel, dim3((unsigned)ts::ceil_div(P, 256)), dim3(256), 0, s, st->critic1_old,
                           st->critic1, st->critic2_old, st->critic2, P, (float)hp->tau, (float)(1.0 - hp->tau));
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// ---- REDQ ----------------------------------------------------------------------------------------------------------
int ts_redq_target_q(ts_workspace* ws, const float* actor, const float* critics_old, int64_t E, const int32_t* h_subset,
                     int64_t S, int mean_mode, const float* log_alpha, double fixed_alpha, const float* obs_next,
                     const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim, float* out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_redq_target_q: workspace is NULL");
    TS_REQUIRE(B >= 1 && E >= 1 && S >= 1 && S <= E && actor && critics_old && h_subset && obs_next && noise && out,
               TS_ERR_INVALID_ARG, "ts_redq_target_q: bad argument");
    for (int64_t k = 0; k < S; ++k)
        TS_REQUIRE(h_subset[k] >= 0 && h_subset[k] < E, TS_ERR_INVALID_ARG, "ts_redq_target_q: subset index out of range");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream), side;
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    const size_t spl = std::max(split_floats(ma), split_floats(mc));
    const int64_t pc = mc.total();
    if (int rc = ts::ws_reserve(ws, al(4 * B * d.ka) + al(4 * B * d.kc) + 6 * hbytes(B, d) + al(4 * B * 64) +
                                        al(4 * (size_t)S * B * 32) + al(4 * B) + 2 * al(4 * spl) + 4096))
        return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    float* x_c = c.take<float>(B * d.kc);
    const Act aa = take_act(c, B, 64, d.hid, d.depth);
    Act hh[2] = {take_act(c, B, 0, d.hid, d.depth), take_act(c, B, 0, d.hid, d.depth)};      // hidden buffers of the two streams
    float* qs = c.take<float>((size_t)S * B * 32);
    float* logp = c.take<float>(B);
    float* splits[2] = {c.take<float>(spl), c.take<float>(spl)};
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    hipStream_t st2[2] = {s, side};
    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs_next,
                       (const float*)nullptr, B, d.obs, d.act, d.ka, d.kc, x_a, x_c, (float*)nullptr);
    if (int rc = mlp_forward(s, ws, ma, actor, x_a, aa, splits[0])) return rc;
    hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B, d.act,
                       64, d.obs, d.kc, d.bound, x_c, (float*)nullptr, logp, (float*)nullptr);
    TS_LAUNCH_CHECK();
    if (S <= MULTI_MAX && mc.three() && ts::mlp3_supported(mc.l[0].IC, mc.l[0].OC, mc.l[2].OC)) {
        // the subset's members in one launch (no activations kept: inference)
        const float* pp[MULTI_MAX];
        Act as[MULTI_MAX];
        for (int64_t k = 0; k < S; ++k) { pp[k] = critics_old + (int64_t)h_subset[k] * pc; as[k] = act_of(nullptr, nullptr, qs + k * B * 32); }
        if (S == 1) { as[0] = hh[0]; as[0].out = qs; }
        if (int rc = mlp_forward_multi(s, ws, mc, (int)S, pp, x_c, as, splits)) return rc;
    } else {
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
        for (int64_t k = 0; k < S; ++k) {             // the subset's members alternate between the two streams
            const int w = (int)(k & 1);
            Act a = hh[w];
            a.out = qs + k * B * 32;
            if (int rc = mlp_forward(st2[w], ws, mc, critics_old + (int64_t)h_subset[k] * pc, x_c, a, splits[w])) return rc;
        }
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    }
    hipLaunchKernelGGL(redq_target_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, qs, (int)S, B * 32, logp,
                       log_alpha, (float)fixed_alpha, mean_mode, B, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_redq_update(ts_workspace* ws, const ts_redq_state* st, int64_t E, int64_t critic_step, int64_t actor_step,
                   int do_actor, const float* obs, const float* act, const float* returns, const float* weight,
                   const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp,
                   float* stats_out4, float* weight_out, float* grads_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_redq_update: workspace is NULL");
    TS_REQUIRE(st && hp && obs && act && returns && stats_out4 && B >= 1 && E >= 1 && E <= 64 && critic_step >= 1 &&
                   actor_step >= 1, TS_ERR_INVALID_ARG, "ts_redq_update: bad argument");
    TS_REQUIRE(st->actor && st->actor_m && st->actor_v && st->critics && st->critics_m && st->critics_v && st->critics_old,
               TS_ERR_INVALID_ARG, "ts_redq_update: NULL state pointer");
    TS_REQUIRE(!do_actor || noise, TS_ERR_INVALID_ARG, "ts_redq_update: the actor step needs the rsample() noise");
    TS_REQUIRE(!hp->auto_alpha || (st->log_alpha && st->log_alpha_m && st->log_alpha_v), TS_ERR_INVALID_ARG,
               "ts_redq_update: auto alpha needs log_alpha and its Adam moments");
    Dims d;
    if (int rc = make_dims(ws, obs_dim, act_dim, &d)) return rc;
    hipStream_t s = ts::as_stream(stream), side;
    const Mlp ma = make_mlp((int)B, d.ka, 64, d.hid, d.depth, d.fn), mc = make_mlp((int)B, d.kc, 32, d.hid, d.depth, d.fn);
    const size_t slab = std::max(slab_floats(ma), slab_floats(mc)), spl = std::max(split_floats(ma), split_floats(mc));
    const int64_t pa = ma.total(), pc = mc.total();
    // the critics' chains run CH members per launch on the fused path (ts_mlp.hip): CH scratch sets instead of two
    const bool batched = fused_backward(mc, false, 0, 0) && ts::mlp3_supported(mc.l[0].IC, mc.l[0].OC, mc.l[2].OC);
    static const int ch_env = getenv("TS_REDQ_CHUNK") ? atoi(getenv("TS_REDQ_CHUNK")) : 0;      // experiments
    const int ch_want = ch_env >= 1 && ch_env <= WGRADS_MAX_NETS ? ch_env : WGRADS_MAX_NETS;
    const int CH = batched ? (int)std::min<int64_t>(ch_want, E) : 2;
    const size_t bytes = al(4 * B * d.ka) + 5 * al(4 * B * d.kc) + (size_t)(2 * E + 2 + 2 * CH) * hbytes(B, d) +
                         (size_t)E * al(4 * B * 32) + 4 * al(4 * B * 64) + (size_t)CH * al(4 * slab) + al(4 * (size_t)E * pc) +
                         al(4 * pa) + 2 * al(4 * spl) + al(4 * (size_t)E * B) + 2 * al(4 * B) + al(4 * B * 3 * d.act) +
                         al(4 * (size_t)CH * B * 32) + al(256) + al(4 * 1024) + 8192;
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    float* x_a = c.take<float>(B * d.ka);
    float* x_c = c.take<float>(B * d.kc);
    float* x_p = c.take<float>(B * d.kc);
    float* dx = c.take<float>(B * d.kc);
    float* dx_sum = c.take<float>(B * d.kc);
    float* zeros = c.take<float>(B * d.kc);
    Act acts[64];
    for (int e = 0; e < E; ++e) acts[e] = take_act(c, B, 32, d.hid, d.depth);
    const Act aa = take_act(c, B, 64, d.hid, d.depth);
    float* d_head = c.take<float>(B * 64);
    float* dheads[2] = {c.take<float>(B * 64), c.take<float>(B * 64)};
    float* d_q = dheads[0];                          // actor phase: the critic-phase gradients are consumed by then
    BwdScratch scs[WGRADS_MAX_NETS];
    for (int k = 0; k < CH; ++k) scs[k] = take_scratch(c, B, d.hid, d.depth, slab);
    float* dheads_ch = c.take<float>((size_t)CH * B * 32);          // batched path: one head gradient per member of a chunk
    float* gcrit = c.take<float>((size_t)E * pc);
    float* gact = c.take<float>(pa);
    float* splits[2] = {c.take<float>(spl), c.take<float>(spl)};
    float* tds = c.take<float>((size_t)E * B);
    float* logp = c.take<float>(B);
    float* keep = c.take<float>(B * 3 * d.act);
    float* loss_parts = c.take<float>(64);
    float* norm_part = c.take<float>(1024);
    if (grads_out) { gcrit = grads_out; gact = grads_out + (size_t)E * pc; }
    const float* log_alpha = hp->auto_alpha ? st->log_alpha : nullptr;
    const float inv_eb = 1.f / ((float)E * (float)B);

    hipLaunchKernelGGL(sac_pack_kernel, dim3((unsigned)ts::ceil_div(B * (d.ka + d.kc) / 4, 256)), dim3(256), 0, s, obs, act, B,
                       d.obs, d.act, d.ka, d.kc, x_a, x_c, (float*)nullptr);
    TS_HIP_CHECK(hipMemsetAsync(dheads[0], 0, sizeof(float) * B * 64, s));
    TS_HIP_CHECK(hipMemsetAsync(dheads[1], 0, sizeof(float) * B * 64, s));
    // ensemble loss (redq.py:266-271): the members are independent chains, alternating between two streams
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    hipStream_t st2[2] = {s, side};
    if (batched) {
        // CH members per launch: forward chains, losses, input-gradient chains, 3 CH weight-gradient GEMMs, 3 CH slab sums
        // -- five launches per chunk instead of five per member
        const int64_t q_stride = E > 1 ? acts[1].out - acts[0].out : 0;     // (Act blocks are carved back to back)
        for (int e0 = 0; e0 < E; e0 += CH) {
            const int n = (int)std::min<int64_t>(CH, E - e0);
            const float* pp[WGRADS_MAX_NETS];
            const float* xs[WGRADS_MAX_NETS];
            const float* dh[WGRADS_MAX_NETS];
            float* gk[WGRADS_MAX_NETS];
            for (int k = 0; k < n; ++k) {
                pp[k] = st->critics + (int64_t)(e0 + k) * pc; xs[k] = x_c; dh[k] = dheads_ch + (int64_t)k * B * 32;
                gk[k] = gcrit + (int64_t)(e0 + k) * pc;
            }
            if (int rc = mlp_forward_multi(s, ws, mc, n, pp, x_c, acts + e0, splits)) return rc;
            hipLaunchKernelGGL(redq_critic_loss_kernel, dim3((unsigned)n, 1 + (unsigned)ts::ceil_div(B, 1024)), dim3(1024), 0, s, acts[e0].out, q_stride, returns, weight,
                               B, inv_eb, tds + (int64_t)e0 * B, dheads_ch, B * 32, loss_parts + e0);
            TS_LAUNCH_CHECK();
            if (int rc = mlp_backward_multi(s, ws, mc, n, pp, x_c, acts + e0, dh, nullptr, 0, 0, scs)) return rc;
            if (int rc = mlp_weight_grads(s, ws, n, mc, xs, acts + e0, dh, gk, scs)) return rc;
        }
    } else {
        if (int rc = ts::stream_wait(ws, s, side, 0)) return rc;
        for (int e = 0; e < E; ++e) {
            const int w = e & 1;
            const float* pe = st->critics + (int64_t)e * pc;
            if (int rc = mlp_forward(st2[w], ws, mc, pe, x_c, acts[e], splits[w])) return rc;
            hipLaunchKernelGGL(redq_critic_loss_kernel, dim3(1, 1 + (unsigned)ts::ceil_div(B, 1024)), dim3(1024), 0, st2[w], acts[e].out, (int64_t)0, returns, weight, B,
                               inv_eb, tds + (int64_t)e * B, dheads[w], (int64_t)0, loss_parts + e);
            TS_LAUNCH_CHECK();
            if (int rc = mlp_backward(st2[w], ws, mc, pe, x_c, acts[e], dheads[w], gcrit + (int64_t)e * pc, nullptr, 0, 0, scs[w]))
                return rc;
        }
        if (int rc = ts::stream_wait(ws, side, s, 1)) return rc;
    }
    hipLaunchKernelGGL(redq_finish_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, tds, loss_parts, (int)E, B,
                       stats_out4 + 1, weight_out);
    TS_LAUNCH_CHECK();
    if (hp->critic_lr >= 0.0)                        // one optimizer over the whole ensemble (test_redq.py:108)
        if (int rc = ts::adam_step(s, st->critics, st->critics_m, st->critics_v, gcrit, (int64_t)E * pc, critic_step,
                                   hp->critic_lr, hp->beta1, hp->beta2, hp->adam_eps, 0.0, norm_part))
            return rc;

    if (do_actor) {                                  // redq.py:277-289, every actor_delay-th update: one stream
        TS_HIP_CHECK(hipMemcpyAsync(x_p, x_c, sizeof(float) * B * d.kc, hipMemcpyDeviceToDevice, s));
        TS_HIP_CHECK(hipMemsetAsync(zeros, 0, sizeof(float) * B * d.kc, s));
        TS_HIP_CHECK(hipMemsetAsync(d_q, 0, sizeof(float) * B * 32, s));
        TS_HIP_CHECK(hipMemsetAsync(d_head, 0, sizeof(float) * B * 64, s));
        if (int rc = mlp_forward(s, ws, ma, st->actor, x_a, aa, splits[0])) return rc;
        hipLaunchKernelGGL(sac_policy_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise, B, d.act,
                           64, d.obs, d.kc, d.bound, x_p, (float*)nullptr, logp, keep);
        TS_LAUNCH_CHECK();
        for (int e = 0; e < E; ++e)
            if (int rc = mlp_forward(s, ws, mc, st->critics + (int64_t)e * pc, x_p, acts[e], splits[0])) return rc;
        // the member outputs sit at a fixed stride only if carved back to back: gather their column-0 base pointers
        // through the stride between consecutive Act blocks
        const int64_t stride = acts[E > 1 ? 1 : 0].out - acts[0].out;
        hipLaunchKernelGGL(redq_actor_loss_kernel, dim3(1), dim3(1024), 0, s, acts[0].out, (int)E, E > 1 ? stride : 0, logp,
                           log_alpha, (float)hp->alpha, B, d_q, stats_out4);
        TS_LAUNCH_CHECK();
        for (int e = 0; e < E; ++e) {
            if (int rc = mlp_backward(s, ws, mc, st->critics + (int64_t)e * pc, x_p, acts[e], d_q, nullptr, dx, d.obs,
                                      d.obs + d.act, scs[0]))
                return rc;
            hipLaunchKernelGGL(acc_cols_kernel, dim3((unsigned)ts::ceil_div(B * d.act, 256)), dim3(256), 0, s, dx_sum, dx, B,
                               d.kc, d.obs, d.act, e == 0 ? 1 : 0);
        }
        hipLaunchKernelGGL(sac_policy_bwd_kernel, dim3((unsigned)ts::ceil_div(B * 32, 256)), dim3(256), 0, s, aa.out, noise,
                           keep, dx_sum, zeros, log_alpha, (float)hp->alpha, B, d.act, 64, d.obs, d.kc, d.bound, d_head);
        TS_LAUNCH_CHECK();
        if (int rc = mlp_backward(s, ws, ma, st->actor, x_a, aa, d_head, gact, nullptr, 0, 0, scs[0])) return rc;
        if (hp->actor_lr >= 0.0)
            if (int rc = ts::adam_step(s, st->actor, st->actor_m, st->actor_v, gact, pa, actor_step, hp->actor_lr, hp->beta1,
                                       hp->beta2, hp->adam_eps, 0.0, norm_part))
                return rc;
        AlphaArgs al2{};                              // AutoAlpha.update(-log_prob) (redq.py:286-288)
        al2.logp = logp; al2.B = B; al2.target_entropy = (float)hp->target_entropy;
        al2.log_alpha = hp->auto_alpha ? st->log_alpha : nullptr; al2.m = st->log_alpha_m; al2.v = st->log_alpha_v;
        const double bc1 = 1.0 - pow(hp->beta1, (double)actor_step), bc2 = 1.0 - pow(hp->beta2, (double)actor_step);
        al2.lr_step = (float)(hp->alpha_lr / bc1); al2.beta1 = (float)hp->beta1; al2.beta2 = (float)hp->beta2;
        al2.omb1 = (float)(1.0 - hp->beta1); al2.omb2 = (float)(1.0 - hp->beta2);
        al2.bc2_sqrt = (float)sqrt(bc2); al2.eps = (float)hp->adam_eps;
        al2.alpha_loss = stats_out4 + 3; al2.alpha_out = stats_out4 + 2; al2.fixed_alpha = (float)hp->alpha;
        al2.td1 = nullptr; al2.td2 = nullptr; al2.weight_out = nullptr;
        hipLaunchKernelGGL(sac_alpha_kernel, dim3(1), dim3(1024), 0, s, al2);
    }
    if (hp->tau > 0.0)
        hipLaunchKernelGGL(polyak1_kernel, dim3((unsigned)ts::ceil_div((int64_t)E * pc, 256)), dim3(256), 0, s, st->critics_old,
                           st->critics, (int64_t)E * pc, (float)hp->tau, (float)(1.0 - hp->tau));
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
