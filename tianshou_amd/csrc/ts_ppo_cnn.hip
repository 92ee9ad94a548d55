// ts_ppo_cnn.hip -- PPO on the Atari actor-critic (shared NatureCNN trunk, categorical policy) for gfx950.
//
// Replaces, for the networks of examples/atari/atari_ppo.py:106-118 (DQNet(features_only=True,
// output_dim_added_layer=512) shared by DiscreteActor(softmax_output=False) and DiscreteCritic):
//   no-grad V(s) / log pi(a|s)            a2c.py:122-129, ppo.py:157-161
//   one minibatch of PPO._update_with_batch   ppo.py:179-216  (Categorical log_prob / entropy,
//       clipped surrogate with optional dual clip, clipped value loss, loss = clip + vf_coef vf - ent_coef ent)
//   Optimizer.step                        algorithm_base.py:484-500 (clip_grad_norm_ + Adam)
// The reference runs the shared trunk twice per minibatch (policy forward, critic forward) and lets autograd
// add the two paths; here the trunk runs once and the two heads are one 32-column GEMM (columns [0, A) = logits,
// column A = V), whose upstream gradient carries both paths.  Conv / linear layers: ts_conv.hip (fp32 MFMA).
#include <algorithm>

#include "ts_common.h"
#include "ts_conv.h"

#pragma clang fp contract(off)

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
}

namespace {

constexpr int HIDDEN = 512, HEAD = 32;

struct Net {
    int nl;                     // layers: 5 = conv1, conv2, conv3, fc, head (512 -> 32); 3 = l1, l2, head (MLP trunk)
    ts::ConvGeom l[5];
    int64_t off[6];             // off[nl] = parameter count
    int n_act;
    int obs_dim;                // MLP trunk: unpadded observation width (l[0].IC = obs_dim rounded up to 32); else 0
};

int make_net(int B, int c, int h, int w, int n_act, Net* n) {
    TS_REQUIRE(c >= 1 && h >= 1 && w >= 1 && n_act >= 1 && n_act < HEAD, TS_ERR_INVALID_ARG,
               "cnn actor-critic: bad dimensions (n_act <= 31)");
    static const int oc[3] = {32, 64, 64}, ks[3] = {8, 4, 3}, st[3] = {4, 2, 1};
    int ic = c, ih = h, iw = w;
    for (int i = 0; i < 3; ++i) {
        TS_REQUIRE(ih >= ks[i] && iw >= ks[i], TS_ERR_INVALID_ARG, "cnn actor-critic: observation too small");
        n->l[i] = ts::ConvGeom{B, ih, iw, ic, ks[i], ks[i], st[i], (ih - ks[i]) / st[i] + 1, (iw - ks[i]) / st[i] + 1, oc[i]};
        ic = oc[i]; ih = n->l[i].OH; iw = n->l[i].OW;
    }
    n->l[3] = ts::ConvGeom{B, 1, 1, ic * ih * iw, 1, 1, 1, 1, 1, HIDDEN};
    n->l[4] = ts::ConvGeom{B, 1, 1, HIDDEN, 1, 1, 1, 1, 1, HEAD};
    n->nl = 5;
    n->n_act = n_act;
    n->obs_dim = 0;
    int64_t o = 0;
    for (int i = 0; i < 5; ++i) { n->off[i] = o; o += n->l[i].param_elems(); }
    n->off[5] = o;
    return TS_OK;
}

// Net(obs, [hidden, hidden]) ReLU trunk (utils/net/common.py:343-369) shared by DiscreteActor / DiscreteCritic
// (test/discrete/test_ppo_discrete.py:88-98): Linear layers are the 1x1 case of the conv kernels.
int make_mlp_net(int B, int64_t obs_dim, int64_t hidden, int64_t n_act, Net* n) {
    TS_REQUIRE(obs_dim >= 1 && obs_dim <= 65536 && hidden >= 32 && hidden <= 2048 && hidden % 32 == 0 && n_act >= 1 &&
                   n_act < HEAD, TS_ERR_INVALID_ARG,
               "mlp actor-critic: obs_dim >= 1, hidden a multiple of 32 in [32, 2048], n_act <= 31");
    const int k0 = ((int)obs_dim + 31) / 32 * 32, hid = (int)hidden;
    n->l[0] = ts::ConvGeom{B, 1, 1, k0, 1, 1, 1, 1, 1, hid};
    n->l[1] = ts::ConvGeom{B, 1, 1, hid, 1, 1, 1, 1, 1, hid};
    n->l[2] = ts::ConvGeom{B, 1, 1, hid, 1, 1, 1, 1, 1, HEAD};
    n->nl = 3;
    n->n_act = (int)n_act;
    n->obs_dim = (int)obs_dim;
    int64_t o = 0;
    for (int i = 0; i < 3; ++i) { n->off[i] = o; o += n->l[i].param_elems(); }
    n->off[3] = o;
    for (int i = 4; i <= 5; ++i) n->off[i] = o;
    return TS_OK;
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Acts { float* h[5]; float* split; };

size_t split_floats(const Net& n) {
    size_t s = 4;
    for (int i = 0; i < n.nl; ++i) {
        const int ns = ts::conv_fwd_splits(n.l[i]);
        if (ns > 1) s = std::max(s, (size_t)ns * n.l[i].out_elems());
    }
    return s;
}

size_t acts_bytes(const Net& n) {
    size_t s = al(4 * split_floats(n));
    for (int i = 0; i < n.nl; ++i) s += al(4 * (size_t)n.l[i].out_elems());
    return s;
}

size_t front_bytes(const Net& n) {      // MLP trunk: the zero-padded copy of the observations
    return n.obs_dim ? al(4 * (size_t)n.l[0].in_elems()) : 0;
}

char* carve_acts(const Net& n, char* p, Acts* a) {
    for (int i = 0; i < n.nl; ++i) { a->h[i] = reinterpret_cast<float*>(p); p += al(4 * (size_t)n.l[i].out_elems()); }
    a->split = reinterpret_cast<float*>(p);
    return p + al(4 * split_floats(n));
}

int net_forward(hipStream_t s, ts_workspace* ws, const Net& n, const float* params, const void* obs, bool obs_u8,
                const Acts& a) {
    const float* x = static_cast<const float*>(obs);
    for (int i = 0; i < n.nl; ++i) {
        if (int rc = ts::conv_forward(s, n.l[i], x, params + n.off[i], a.h[i], i < n.nl - 1, a.split, ws, i == 0 && obs_u8))
            return rc;
        x = a.h[i];
    }
    return TS_OK;
}

// x[b] = [obs[b] | 0]: the K dimension of the first Linear layer padded to a multiple of 32
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ obs, int64_t B, int obs_dim, int k0,
                                                       float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * k0) return;
    const int64_t b = i / k0;
    const int j = (int)(i - b * k0);
    x[i] = j < obs_dim ? obs[b * obs_dim + j] : 0.f;
}

// log-softmax pieces of one sample's logits (A <= 31): returns log-sum-exp
__device__ __forceinline__ float logsumexp(const float* l, int A) {
    float m = l[0];
    for (int j = 1; j < A; ++j) m = fmaxf(m, l[j]);
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(l[j] - m);
    return m + logf(s);
}

// v_out[b] = V, logp_out[b] = Categorical(logits).log_prob(act[b])
__global__ __launch_bounds__(256) void cnn_infer_kernel(const float* __restrict__ head, const int64_t* __restrict__ act,
                                                        int64_t B, int A, float* __restrict__ v_out,
                                                        float* __restrict__ logp_out, float* __restrict__ logits_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* hb = head + b * HEAD;
    if (v_out) v_out[b] = hb[A];
    if (logp_out) logp_out[b] = hb[act[b]] - logsumexp(hb, A);
    if (logits_out) for (int j = 0; j < A; ++j) logits_out[b * A + j] = hb[j];
}

struct LossArgs {
    const float* head; const int64_t* act; const float* adv; const float* ret; const float* logp_old; const float* v_old;
    const float* adv_stats;      // {mean, std} of this minibatch's advantages (nullable)
    int64_t B; int A;
    float eps_clip, dual_clip, vf_coef, ent_coef; int value_clip;
    int algo;                    // 0: PPO (ppo.py:184-211), 1: A2C (a2c.py:262-273)
    float* d_head;               // [B, 32]
    float* partials;             // [blocks, 3]: sums of clip term, value term, entropy
};

// ppo.py:184-211 for one sample per thread; exact torch tie semantics as in ts_ppo.hip
__global__ __launch_bounds__(256) void cnn_ppo_loss_kernel(LossArgs g) {
    __shared__ float red[3][4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float w = 1.f / (float)g.B;
    float t_clip = 0.f, t_vf = 0.f, t_ent = 0.f;
    if (b < g.B) {
        const float* hb = g.head + b * HEAD;
        float* db = g.d_head + b * HEAD;
        const int A = g.A, a = (int)g.act[b];
        const float lse = logsumexp(hb, A);
        float H = 0.f;
        for (int j = 0; j < A; ++j) { const float lp = hb[j] - lse; H -= expf(lp) * lp; }   // Categorical.entropy
        const float logp = hb[a] - lse;
        float Ad = g.adv[b];
        if (g.adv_stats) Ad = (Ad - g.adv_stats[0]) / (g.adv_stats[1] + 1e-8f);            // ppo.py:184-186
        float term, dlogp;
        if (g.algo == 1) {                                        // actor_loss = -(log_prob * adv).mean()
            term = -(logp * Ad);
            dlogp = -Ad * w;
        } else {
            const float ratio = expf(logp - g.logp_old[b]);
            const float surr1 = ratio * Ad;
            const float surr2 = fminf(fmaxf(ratio, 1.f - g.eps_clip), 1.f + g.eps_clip) * Ad;
            const float clip1 = fminf(surr1, surr2);
            float basek = (surr1 <= surr2) ? Ad : 0.f;            // torch.min backward (ties: both branches equal)
            if (g.dual_clip > 0.f) {
                const float clip2 = fmaxf(clip1, g.dual_clip * Ad);
                if (Ad < 0.f) { term = -clip2; if (!(clip1 >= g.dual_clip * Ad)) basek = 0.f; }
                else term = -clip1;
            } else {
                term = -clip1;
            }
            dlogp = -basek * ratio * w;
        }
        for (int j = 0; j < HEAD; ++j) {
            float d = 0.f;
            if (j < A) {
                const float lp = hb[j] - lse, p = expf(lp);
                d = dlogp * ((j == a ? 1.f : 0.f) - p) + g.ent_coef * w * p * (lp + H);   // - ent_coef * dH/dl
            }
            db[j] = d;
        }
        const float value = hb[A], ret = g.ret[b];
        const float vf1 = (ret - value) * (ret - value);
        float vterm, dv;
        if (g.value_clip && g.algo == 0) {                                    // ppo.py:199-206
            const float vo = g.v_old[b], dvo = value - vo;
            const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
            const float vf2 = (ret - vclip) * (ret - vclip);
            vterm = fmaxf(vf1, vf2);
            const float g1 = -2.f * (ret - value);
            const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
            dv = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        } else {
            vterm = vf1;
            dv = -2.f * (ret - value);
        }
        db[A] = dv * g.vf_coef * w;
        t_clip = term; t_vf = vterm; t_ent = H;
    }
    float v3[3] = {t_clip, t_vf, t_ent};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = v3[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) g.partials[blockIdx.x * 3 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// losses[4] = {loss, clip, vf, ent} from the per-block partial sums (fixed order)
__global__ __launch_bounds__(256) void cnn_loss_finish_kernel(const float* __restrict__ partials, int n_blocks, int64_t B,
                                                              float vf_coef, float ent_coef, float* __restrict__ losses) {
    __shared__ float red[3][256];
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += 256) s += partials[i * 3 + k];
        red[k][threadIdx.x] = s;
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float inv = 1.f / (float)B;
        const float clip = red[0][0] * inv, vf = red[1][0] * inv, ent = red[2][0] * inv;
        losses[0] = clip + vf_coef * vf - ent_coef * ent;                    // ppo.py:211
        losses[1] = clip; losses[2] = vf; losses[3] = ent;
    }
}

// Input of the first layer: the observations themselves (CNN trunk, NHWC float32 / uint8) or their zero-padded copy at
// the front of the workspace (MLP trunk).
const void* first_input(hipStream_t s, const Net& n, ts_workspace* ws, const void* obs, int64_t B) {
    if (!n.obs_dim) return obs;
    float* x = static_cast<float*>(ws->base);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ts::ceil_div(B * n.l[0].IC, 256)), dim3(256), 0, s,
                       static_cast<const float*>(obs), B, n.obs_dim, n.l[0].IC, x);
    return x;
}

int ac_infer(ts_workspace* ws, const Net& n, const float* params, const void* obs, bool obs_u8, const int64_t* act,
             int64_t B, float* v_out, float* logp_out, float* logits_out, hipStream_t s) {
    if (int rc = ts::ws_reserve(ws, front_bytes(n) + acts_bytes(n))) return rc;
    Acts a;
    carve_acts(n, static_cast<char*>(ws->base) + front_bytes(n), &a);
    const void* x = first_input(s, n, ws, obs, B);
    if (int rc = net_forward(s, ws, n, params, x, obs_u8, a)) return rc;
    hipLaunchKernelGGL(cnn_infer_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, a.h[n.nl - 1], act, B,
                       n.n_act, v_out, logp_out, logits_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ac_ppo_step(ts_workspace* ws, const Net& n, float* params, float* adam_m, float* adam_v, int64_t adam_step,
                const void* obs, bool obs_u8, const int64_t* act, const float* adv, const float* returns,
                const float* logp_old, const float* v_old, int64_t B, const float* adv_stats, const ts_ppo_hparams* hp,
                float* losses_out4, float* grad_out, hipStream_t s, const char* who) {
    TS_REQUIRE(hp->algo == 0 || hp->algo == 1, TS_ERR_UNSUPPORTED, "%s: algo must be 0 (PPO) or 1 (A2C)", who);
    TS_REQUIRE(hp->algo == 1 || (logp_old && v_old), TS_ERR_INVALID_ARG, "%s: PPO needs logp_old and v_old", who);
    TS_REQUIRE(hp->algo == 1 || !hp->adv_norm || adv_stats, TS_ERR_INVALID_ARG, "%s: adv_norm needs adv_stats", who);
    const int nl = n.nl;
    const int64_t P = n.off[nl];
    const int n_blocks = (int)ts::ceil_div(B, 256);
    size_t slab = 0;
    for (int i = 0; i < nl; ++i)
        slab = std::max(slab, 4 * (size_t)ts::conv_wgrad_splits(n.l[i]) * n.l[i].param_elems());
    size_t bytes = front_bytes(n) + acts_bytes(n) + al(slab) + al(4 * (size_t)P) + al(12 * (size_t)n_blocks) + 4096;
    for (int i = 0; i < nl; ++i) bytes += al(4 * (size_t)n.l[i].out_elems());
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Acts a;
    char* p = carve_acts(n, static_cast<char*>(ws->base) + front_bytes(n), &a);
    float* dy[5] = {};
    for (int i = 0; i < nl; ++i) { dy[i] = reinterpret_cast<float*>(p); p += al(4 * (size_t)n.l[i].out_elems()); }
    float* slabs = reinterpret_cast<float*>(p); p += al(slab);
    float* grad = reinterpret_cast<float*>(p); p += al(4 * (size_t)P);
    float* partials = reinterpret_cast<float*>(p); p += al(12 * (size_t)n_blocks);
    float* norm_part = reinterpret_cast<float*>(p);
    if (grad_out) grad = grad_out;

    const void* x0 = first_input(s, n, ws, obs, B);
    if (int rc = net_forward(s, ws, n, params, x0, obs_u8, a)) return rc;
    LossArgs la{};
    la.head = a.h[nl - 1]; la.act = act; la.adv = adv; la.ret = returns; la.logp_old = logp_old; la.v_old = v_old;
    la.adv_stats = (hp->adv_norm && hp->algo == 0) ? adv_stats : nullptr;      // A2C does not normalise
    la.B = B; la.A = n.n_act;
    la.eps_clip = (float)hp->eps_clip; la.dual_clip = (float)hp->dual_clip; la.vf_coef = (float)hp->vf_coef;
    la.ent_coef = (float)hp->ent_coef; la.value_clip = hp->value_clip; la.algo = hp->algo;
    la.d_head = dy[nl - 1]; la.partials = partials;
    hipLaunchKernelGGL(cnn_ppo_loss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, la);
    hipLaunchKernelGGL(cnn_loss_finish_kernel, dim3(1), dim3(256), 0, s, partials, n_blocks, B, (float)hp->vf_coef,
                       (float)hp->ent_coef, losses_out4);
    TS_LAUNCH_CHECK();
    for (int i = nl - 1; i >= 0; --i) {
        const float* x = i == 0 ? static_cast<const float*>(x0) : a.h[i - 1];
        if (int rc = ts::conv_wgrad(s, n.l[i], x, dy[i], slabs, ws, i == 0 && obs_u8)) return rc;
        if (int rc = ts::slab_sum(s, slabs, ts::conv_wgrad_splits(n.l[i]), n.l[i].param_elems(), grad + n.off[i]))
            return rc;
        if (i > 0)
            if (int rc = ts::conv_dgrad(s, n.l[i], dy[i], params + n.off[i], a.h[i - 1], dy[i - 1], ws)) return rc;
    }
    if (hp->lr < 0.0) return TS_OK;
    return ts::optim_step(s, ts::optim_from(hp), params, adam_m, adam_v, grad, P, adam_step, hp->lr, hp->beta1, hp->beta2, hp->adam_eps,
                         hp->max_grad_norm, norm_part);
}

}  // namespace

extern "C" {

int ts_cnn_ac_layer_offsets(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t* h_offsets7, int64_t* h_geom) {
    Net n;
    if (int rc = make_net(1, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    TS_REQUIRE(h_offsets7, TS_ERR_INVALID_ARG, "ts_cnn_ac_layer_offsets: NULL output");
    for (int i = 0; i < 6; ++i) h_offsets7[i] = n.off[i];
    h_offsets7[6] = HEAD;
    if (h_geom)
        for (int i = 0; i < 5; ++i) {
            const ts::ConvGeom& g = n.l[i];
            const int64_t v[10] = {g.B, g.IH, g.IW, g.IC, g.KH, g.KW, g.S, g.OH, g.OW, g.OC};
            for (int j = 0; j < 10; ++j) h_geom[i * 10 + j] = v[j];
        }
    return TS_OK;
}

int ts_cnn_ac_infer(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                    const void* obs_nhwc, int obs_u8, const int64_t* act, int64_t B, float* v_out, float* logp_out,
                    float* logits_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_cnn_ac_infer: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_cnn_ac_infer: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs_nhwc && (act || !logp_out), TS_ERR_INVALID_ARG, "ts_cnn_ac_infer: NULL argument");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    return ac_infer(ws, n, params, obs_nhwc, obs_u8 != 0, act, B, v_out, logp_out, logits_out, ts::as_stream(stream));
}

int ts_mlp_ac_layout(int64_t obs_dim, int64_t hidden, int64_t n_act, int64_t* h_out3) {
    Net n;
    if (int rc = make_mlp_net(1, obs_dim, hidden, n_act, &n)) return rc;
    TS_REQUIRE(h_out3, TS_ERR_INVALID_ARG, "ts_mlp_ac_layout: NULL output");
    h_out3[0] = n.l[0].IC; h_out3[1] = HEAD; h_out3[2] = n.off[3];
    return TS_OK;
}

int ts_mlp_ac_infer(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t n_act,
                    const float* obs, const int64_t* act, int64_t B, float* v_out, float* logp_out, float* logits_out,
                    ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_mlp_ac_infer: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_mlp_ac_infer: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs && (act || !logp_out), TS_ERR_INVALID_ARG, "ts_mlp_ac_infer: NULL argument");
    Net n;
    if (int rc = make_mlp_net((int)B, obs_dim, hidden, n_act, &n)) return rc;
    return ac_infer(ws, n, params, obs, false, act, B, v_out, logp_out, logits_out, ts::as_stream(stream));
}

int ts_mlp_ppo_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                    int64_t hidden, int64_t n_act, const float* obs, const int64_t* act, const float* adv,
                    const float* returns, const float* logp_old, const float* v_old, int64_t B, const float* adv_stats,
                    const ts_ppo_hparams* hp, float* losses_out4, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_mlp_ppo_step: workspace is NULL");
    TS_REQUIRE(B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG, "ts_mlp_ppo_step: bad batch size / step");
    TS_REQUIRE(params && adam_m && adam_v && obs && act && adv && returns && hp && losses_out4, TS_ERR_INVALID_ARG,
               "ts_mlp_ppo_step: NULL argument");
    Net n;
    if (int rc = make_mlp_net((int)B, obs_dim, hidden, n_act, &n)) return rc;
    return ac_ppo_step(ws, n, params, adam_m, adam_v, adam_step, obs, false, act, adv, returns, logp_old, v_old, B,
                       adv_stats, hp, losses_out4, grad_out, ts::as_stream(stream), "ts_mlp_ppo_step");
}

int ts_cnn_ppo_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                    int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act, const float* adv,
                    const float* returns, const float* logp_old, const float* v_old, int64_t B,
                    const float* adv_stats, const ts_ppo_hparams* hp, float* losses_out4, float* grad_out,
                    ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_cnn_ppo_step: workspace is NULL");
    TS_REQUIRE(B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG, "ts_cnn_ppo_step: bad batch size / step");
    TS_REQUIRE(params && adam_m && adam_v && obs_nhwc && act && adv && returns && hp && losses_out4, TS_ERR_INVALID_ARG,
               "ts_cnn_ppo_step: NULL argument");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    return ac_ppo_step(ws, n, params, adam_m, adam_v, adam_step, obs_nhwc, obs_u8 != 0, act, adv, returns, logp_old, v_old,
                       B, adv_stats, hp, losses_out4, grad_out, ts::as_stream(stream), "ts_cnn_ppo_step");
}

}  // extern "C"
