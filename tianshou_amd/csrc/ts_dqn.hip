// ts_dqn.hip -- the DQN learn() step on NatureCNN (DQNet) for gfx950.
//
// Replaces, on device-resident NHWC float32 observations:
//   DQNet.forward                       tianshou/env/atari/atari_network.py:79-98, 111-122
//   DiscreteQLearningPolicy.forward     tianshou/algorithm/modelfree/dqn.py:101-143 (act = argmax_a Q)
//   DQN._target_q                       dqn.py:365-379
//   DQN._update_with_batch              dqn.py:381-404 (+ Optimizer.step algorithm_base.py:484-500)
// The conv / linear layers run on the fp32-MFMA implicit-GEMM kernels of ts_conv.hip; this file adds
// the small head (512 -> n_act), the TD loss and the orchestration.  Flat parameter layout: see
// include/tsengine.h (ts_dqn_param_count).
#include <cstring>
#include "ts_common.h"
#include "ts_conv.h"

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
}

namespace {

constexpr int HIDDEN = 512;
constexpr int MAX_ACT = 64;

struct Net {
    ts::ConvGeom l[4];          // conv1, conv2, conv3, fc1
    int n_act;
    int64_t off[5];             // parameter offsets of the five layers
    int64_t total;
};

int make_net(int B, int c, int h, int w, int n_act, Net* n) {
    TS_REQUIRE(c >= 1 && h >= 1 && w >= 1 && n_act >= 1 && n_act <= MAX_ACT, TS_ERR_INVALID_ARG,
               "dqn: bad network dimensions");
    static const int oc[3] = {32, 64, 64}, ks[3] = {8, 4, 3}, st[3] = {4, 2, 1};
    int ic = c, ih = h, iw = w;
    for (int i = 0; i < 3; ++i) {
        TS_REQUIRE(ih >= ks[i] && iw >= ks[i], TS_ERR_INVALID_ARG, "dqn: observation too small for DQNet");
        ts::ConvGeom& g = n->l[i];
        g = ts::ConvGeom{B, ih, iw, ic, ks[i], ks[i], st[i], (ih - ks[i]) / st[i] + 1, (iw - ks[i]) / st[i] + 1, oc[i]};
        ic = oc[i]; ih = g.OH; iw = g.OW;
    }
    n->l[3] = ts::ConvGeom{B, 1, 1, ic * ih * iw, 1, 1, 1, 1, 1, HIDDEN};
    n->n_act = n_act;
    int64_t o = 0;
    for (int i = 0; i < 4; ++i) { n->off[i] = o; o += n->l[i].param_elems(); }
    n->off[4] = o;
    o += (int64_t)(HIDDEN + 1) * n_act;
    n->total = o;
    return TS_OK;
}

// ---- head: Q = H4 . W5 + b5 (one wave per sample), optional greedy action -----------------------
__global__ __launch_bounds__(256) void head_forward_kernel(const float* __restrict__ h4, const float* __restrict__ wb,
                                                           int64_t B, int A, float* __restrict__ q,
                                                           int64_t* __restrict__ act) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float hv[HIDDEN / 64];
#pragma unroll
    for (int j = 0; j < HIDDEN / 64; ++j) hv[j] = h4[b * HIDDEN + lane + 64 * j];
    float best = 0.f;
    int best_a = 0;
    for (int a = 0; a < A; ++a) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < HIDDEN / 64; ++j) s += hv[j] * wb[(int64_t)(lane + 64 * j) * A + a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        s += wb[(int64_t)HIDDEN * A + a];
        if (lane == 0) q[b * A + a] = s;
        if (a == 0 || s > best) { best = s; best_a = a; }     // first maximum, like torch.argmax
    }
    if (act && lane == 0) act[b] = best_a;
}

// ---- DQN._target_q (dqn.py:365-379) ----------------------------------------------------------------
// With the n-step coefficients of ts_nstep_coefficients: out = the n-step return float(double(tq * mask) * gpow + mc)
// (algorithm_base.py:798-811) instead of tq.
__global__ __launch_bounds__(256) void target_q_kernel(const float* __restrict__ q_online,
                                                       const float* __restrict__ q_target, int64_t B, int A,
                                                       int is_double, float* __restrict__ out,
                                                       const float* __restrict__ ns_mask = nullptr,
                                                       const double* __restrict__ ns_gpow = nullptr,
                                                       const double* __restrict__ ns_mc = nullptr) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* sel = is_double ? q_online : q_target;
    float best = sel[b * A];
    int best_a = 0;
    for (int a = 1; a < A; ++a) {
        const float v = sel[b * A + a];
        if (v > best) { best = v; best_a = a; }
    }
    const float tq = q_target[b * A + best_a];
    if (ns_mask == nullptr) { out[b] = tq; return; }
    const float tqm = tq * ns_mask[b];
    const double qd = (double)tqm * ns_gpow[b];
    out[b] = (float)(qd + ns_mc[b]);
}

// ---- TD error, loss, d loss / d Q[b, act_b] and the head layer's backward pass in ONE launch (dqn.py:388-401) ----------
// d loss_b / d q[b, act_b] (times 1 / B for the mean) and the loss term: a function of one sample -- every workgroup that
// needs it recomputes it from q / returns / weight instead of reading it from a kernel launched before (three dependent
// 4 us launches with queue bubbles between them were 35 us of the C3 update's critical path).
__device__ __forceinline__ float td_terms(const float* __restrict__ q, const int64_t* __restrict__ act,
                                          const float* __restrict__ ret, const float* __restrict__ weight, int64_t b, int A,
                                          float huber_delta, float inv_b, float* t_out, float* l_out) {
    const float t = ret[b] - q[b * A + act[b]];
    float l, g;                                  // g = d loss_b / d q
    if (huber_delta > 0.f) {                     // torch.nn.functional.huber_loss(q, returns)
        const float ad = fabsf(t);
        if (ad < huber_delta) { l = 0.5f * t * t; g = -t; }
        else { l = huber_delta * (ad - 0.5f * huber_delta); g = t > 0.f ? -huber_delta : huber_delta; }
    } else {                                     // (td_error.pow(2) * weight).mean()
        const float w = weight ? weight[b] : 1.f;
        l = t * t * w;
        g = -2.f * t * w;
    }
    *t_out = t;
    *l_out = l;
    return g * inv_b;
}

// workgroup 0: td[b], loss = mean_b l_b.  Workgroups 1 .. HIDDEN + 1: dWb5[k, a] = sum_{b: act_b = a} H4[b, k] dq[b] (row
// HIDDEN: bias, H = 1); thread t owns the samples b = t (mod 256) in batch order, per action a wave shuffle tree, then the
// 4 wave sums in order.  The rest: dH4[b, k] = dq[b] W5[k, act_b] (H4[b, k] > 0).
__global__ __launch_bounds__(256) void head_backward_kernel(const float* __restrict__ q, const int64_t* __restrict__ act,
                                                            const float* __restrict__ ret, const float* __restrict__ weight,
                                                            const float* __restrict__ h4, const float* __restrict__ wb,
                                                            int64_t B, int A, float huber_delta, float* __restrict__ td,
                                                            float* __restrict__ loss, float* __restrict__ dwb,
                                                            float* __restrict__ dh4) {
    __shared__ float red[4][MAX_ACT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_b = 1.f / (float)B;
    float t, l;
    if (blockIdx.x == 0) {
        float lsum = 0.f;
        for (int64_t b = threadIdx.x; b < B; b += 256) {
            td_terms(q, act, ret, weight, b, A, huber_delta, inv_b, &t, &l);
            td[b] = t;
            lsum += l;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
        if (lane == 0) red[wave][0] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) *loss = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) * inv_b;
        return;
    }
    if (blockIdx.x <= HIDDEN + 1) {
        const int k = blockIdx.x - 1;
        for (int a0 = 0; a0 < A; a0 += 8) {                     // 8 actions per pass keeps the accumulators in registers
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int64_t b = threadIdx.x; b < B; b += 256) {
                const float v = (k < HIDDEN ? h4[b * HIDDEN + k] : 1.f) * td_terms(q, act, ret, weight, b, A, huber_delta, inv_b, &t, &l);
                const int ab = (int)act[b] - a0;
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] += a == ab ? v : 0.f;
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                float s = acc[a];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
                if (lane == 0) red[wave][a0 + a] = s;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < A)
            dwb[k * A + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        return;
    }
    const int64_t i = (int64_t)(blockIdx.x - HIDDEN - 2) * 256 + threadIdx.x;
    if (i >= B * HIDDEN) return;
    const int64_t b = i / HIDDEN;
    const int k = (int)(i - b * HIDDEN);
    const float g = td_terms(q, act, ret, weight, b, A, huber_delta, inv_b, &t, &l);
    dh4[i] = h4[i] > 0.f ? g * wb[(int64_t)k * A + act[b]] : 0.f;
}

struct Scratch {           // carve of the workspace for one network pass over B samples
    float* h[4];           // activations of conv1..fc1
    float* q;              // [B, A]
    float* split;          // forward split buffer (fc1)
    size_t bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

size_t fwd_scratch_bytes(const Net& n, int64_t B) {
    size_t s = 0;
    for (int i = 0; i < 4; ++i) s += align_up(sizeof(float) * n.l[i].out_elems());
    s += align_up(sizeof(float) * B * n.n_act);
    size_t sp = 0;
    for (int i = 0; i < 4; ++i) {
        const int ns = ts::conv_fwd_splits(n.l[i]);
        if (ns > 1) sp = std::max(sp, sizeof(float) * (size_t)ns * n.l[i].out_elems());
    }
    return s + align_up(sp);
}

char* carve_fwd(const Net& n, int64_t B, char* p, Scratch* sc) {
    for (int i = 0; i < 4; ++i) { sc->h[i] = reinterpret_cast<float*>(p); p += align_up(sizeof(float) * n.l[i].out_elems()); }
    sc->q = reinterpret_cast<float*>(p); p += align_up(sizeof(float) * B * n.n_act);
    sc->split = reinterpret_cast<float*>(p);
    size_t sp = 0;
    for (int i = 0; i < 4; ++i) {
        const int ns = ts::conv_fwd_splits(n.l[i]);
        if (ns > 1) sp = std::max(sp, sizeof(float) * (size_t)ns * n.l[i].out_elems());
    }
    return p + align_up(sp);
}

int net_forward(hipStream_t s, ts_workspace* ws, const Net& n, const float* params, const void* obs, bool obs_u8,
                int64_t B, const Scratch& sc, float* q_out, int64_t* act_out) {
    const float* x = static_cast<const float*>(obs);
    for (int i = 0; i < 4; ++i) {
        if (int rc = ts::conv_forward(s, n.l[i], x, params + n.off[i], sc.h[i], true, sc.split, ws, i == 0 && obs_u8))
            return rc;
        x = sc.h[i];
    }
    hipLaunchKernelGGL(head_forward_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, s, sc.h[3],
                       params + n.off[4], B, n.n_act, q_out, act_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace

extern "C" {

int64_t ts_dqn_param_count(int64_t c, int64_t h, int64_t w, int64_t n_act) {
    Net n;
    if (make_net(1, (int)c, (int)h, (int)w, (int)n_act, &n) != TS_OK) return -1;
    return n.total;
}

int ts_dqn_layer_offsets(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t* h_offsets6, int64_t* h_geom) {
    Net n;
    if (int rc = make_net(1, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    TS_REQUIRE(h_offsets6, TS_ERR_INVALID_ARG, "ts_dqn_layer_offsets: NULL output");
    for (int i = 0; i < 5; ++i) h_offsets6[i] = n.off[i];
    h_offsets6[5] = n.total;
    if (h_geom)
        for (int i = 0; i < 4; ++i) {
            const ts::ConvGeom& g = n.l[i];
            const int64_t v[10] = {g.B, g.IH, g.IW, g.IC, g.KH, g.KW, g.S, g.OH, g.OW, g.OC};
            for (int j = 0; j < 10; ++j) h_geom[i * 10 + j] = v[j];
        }
    return TS_OK;
}

int ts_dqn_forward(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                   const void* obs_nhwc, int obs_u8, int64_t B, float* q_out, int64_t* act_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dqn_forward: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_dqn_forward: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs_nhwc && q_out, TS_ERR_INVALID_ARG, "ts_dqn_forward: NULL argument");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    if (int rc = ts::ws_reserve(ws, fwd_scratch_bytes(n, B))) return rc;
    Scratch sc;
    carve_fwd(n, B, static_cast<char*>(ws->base), &sc);
    return net_forward(ts::as_stream(stream), ws, n, params, obs_nhwc, obs_u8 != 0, B, sc, q_out, act_out);
}

static int target_q_impl(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                         int64_t w, int64_t n_act, const void* obs_next_nhwc, int obs_u8, int64_t B, int is_double,
                         float* out, const float* ns_mask, const double* ns_gpow, const double* ns_mc, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dqn_target_q_fused: workspace is NULL");
    TS_REQUIRE(B >= 0, TS_ERR_INVALID_ARG, "ts_dqn_target_q_fused: negative batch");
    if (B == 0) return TS_OK;
    TS_REQUIRE(params && obs_next_nhwc && out, TS_ERR_INVALID_ARG, "ts_dqn_target_q_fused: NULL argument");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    const size_t one = fwd_scratch_bytes(n, B);
    if (int rc = ts::ws_reserve(ws, 2 * one)) return rc;
    Scratch sa, sb;
    carve_fwd(n, B, static_cast<char*>(ws->base), &sa);
    carve_fwd(n, B, static_cast<char*>(ws->base) + one, &sb);
    hipStream_t s = ts::as_stream(stream), side;
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    const bool two = params_old != nullptr;
    if (two) {      // Q_target(s') on the side stream, Q_online(s') on the caller's stream (only needed for double-Q)
        if (int rc = ts::stream_wait(ws, s, side, 9)) return rc;
        if (int rc = net_forward(side, ws, n, params_old, obs_next_nhwc, obs_u8 != 0, B, sb, sb.q, nullptr)) return rc;
    }
    if (!two || is_double)
        if (int rc = net_forward(s, ws, n, params, obs_next_nhwc, obs_u8 != 0, B, sa, sa.q, nullptr)) return rc;
    if (two)
        if (int rc = ts::stream_wait(ws, side, s, 10)) return rc;
    hipLaunchKernelGGL(target_q_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, sa.q, two ? sb.q : sa.q, B,
                       (int)n_act, is_double, out, ns_mask, ns_gpow, ns_mc);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_dqn_target_q_fused(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                          int64_t w, int64_t n_act, const void* obs_next_nhwc, int obs_u8, int64_t B, int is_double,
                          float* out, ts_stream_t stream) {
    return target_q_impl(ws, params, params_old, c, h, w, n_act, obs_next_nhwc, obs_u8, B, is_double, out, nullptr, nullptr,
                         nullptr, stream);
}

int ts_dqn_target_returns(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                          int64_t w, int64_t n_act, const void* obs_next_nhwc, int obs_u8, int64_t B, int is_double,
                          const float* nstep_mask, const double* nstep_gpow, const double* nstep_mc, float* returns_out,
                          ts_stream_t stream) {
    TS_REQUIRE(nstep_mask && nstep_gpow && nstep_mc, TS_ERR_INVALID_ARG, "ts_dqn_target_returns: NULL coefficient array");
    return target_q_impl(ws, params, params_old, c, h, w, n_act, obs_next_nhwc, obs_u8, B, is_double, returns_out, nstep_mask,
                         nstep_gpow, nstep_mc, stream);
}

int ts_dqn_target_q(const float* q_online, const float* q_target, int64_t B, int64_t n_act, int is_double,
                    float* out, ts_stream_t stream) {
    TS_REQUIRE(B >= 0 && n_act >= 1, TS_ERR_INVALID_ARG, "ts_dqn_target_q: bad sizes");
    if (B == 0) return TS_OK;
    TS_REQUIRE(q_target && out && (q_online || !is_double), TS_ERR_INVALID_ARG, "ts_dqn_target_q: NULL argument");
    hipLaunchKernelGGL(target_q_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, ts::as_stream(stream),
                       q_online, q_target, B, (int)n_act, is_double, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

static int dqn_update_impl(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                           int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act,
                           const float* returns, const float* weight, int64_t B, const ts_dqn_hparams* hp, float* td_out,
                           float* loss_out, float* grad_out, ts_stream_t stream, void* cache, const char* who,
                           const float* adam_dev = nullptr) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "%s: workspace is NULL", who);
    TS_REQUIRE(B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG, "%s: bad batch size / step", who);
    TS_REQUIRE(params && adam_m && adam_v && obs_nhwc && act && returns && hp && td_out && loss_out,
               TS_ERR_INVALID_ARG, "%s: NULL argument", who);
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);

    // workspace: forward scratch (unless the caller's cache holds the activations) | dq | dY of every layer | wgrad slabs |
    // flat gradient | norm partials
    size_t bytes = cache ? 0 : fwd_scratch_bytes(n, B);
    for (int i = 0; i < 4; ++i) bytes += align_up(sizeof(float) * n.l[i].out_elems());
    size_t slab[4];
    for (int i = 0; i < 4; ++i) {          // one slab set per layer: the weight gradients run side by side (ts::chain_backward)
        slab[i] = align_up(sizeof(float) * (size_t)ts::conv_wgrad_splits(n.l[i]) * n.l[i].param_elems());
        bytes += slab[i];
    }
    bytes += align_up(sizeof(float) * n.total) + 4096;
    if (int rc = ts::ws_reserve(ws, bytes)) return rc;
    Scratch sc;
    char* p = static_cast<char*>(ws->base);
    if (cache) carve_fwd(n, B, static_cast<char*>(cache), &sc);
    else p = carve_fwd(n, B, p, &sc);
    float* dy[4];
    for (int i = 0; i < 4; ++i) { dy[i] = reinterpret_cast<float*>(p); p += align_up(sizeof(float) * n.l[i].out_elems()); }
    float* slabs[4];
    for (int i = 0; i < 4; ++i) { slabs[i] = reinterpret_cast<float*>(p); p += slab[i]; }
    float* grad = reinterpret_cast<float*>(p); p += align_up(sizeof(float) * n.total);
    float* norm_part = reinterpret_cast<float*>(p);
    if (grad_out) grad = grad_out;

    // forward (keeps the activations) -- or the activations ts_dqn_forward_cache left in `cache` --, loss
    if (!cache)
        if (int rc = net_forward(s, ws, n, params, obs_nhwc, obs_u8 != 0, B, sc, sc.q, nullptr)) return rc;
    // loss, TD errors and the head layer's backward pass: one launch
    hipLaunchKernelGGL(head_backward_kernel, dim3((unsigned)(HIDDEN + 2 + ts::ceil_div(B * HIDDEN, 256))), dim3(256), 0, s, sc.q, act,
                       returns, weight, sc.h[3], params + n.off[4], B, n.n_act, (float)hp->huber_delta, td_out, loss_out,
                       grad + n.off[4], dy[3]);
    TS_LAUNCH_CHECK();
    if (int rc = ts::record_td(ws, s)) return rc;        // td_out / loss_out are written: ts_dqn_wait_td
    // fc1, conv3, conv2, conv1: input gradients down the caller's stream, the weight gradients beside them on the workspace's
    // side streams (ts::chain_backward)
    {
        const float* x[4]; const float* wb[4]; float* g[4];
        for (int i = 0; i < 4; ++i) {
            x[i] = i == 0 ? static_cast<const float*>(obs_nhwc) : sc.h[i - 1];
            wb[i] = params + n.off[i];
            g[i] = grad + n.off[i];
        }
        if (int rc = ts::chain_backward(s, ws, 4, n.l, x, dy, wb, slabs, g, obs_u8 != 0)) return rc;
    }
    if (hp->lr < 0.0) return TS_OK;      // gradient-only mode (tests, data-parallel all-reduce)
    if (adam_dev)       // a captured update: the step-dependent scalars live on the device (ts_dqn_learn_step)
        return ts::adam_step_dev(s, params, adam_m, adam_v, grad, n.total, adam_dev, hp->beta1, hp->beta2, hp->adam_eps,
                                 hp->max_grad_norm, norm_part);
    return ts::adam_step(s, params, adam_m, adam_v, grad, n.total, adam_step, hp->lr, hp->beta1, hp->beta2,
                         hp->adam_eps, hp->max_grad_norm, norm_part);
}


int ts_dqn_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                  int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act,
                  const float* returns, const float* weight, int64_t B, const ts_dqn_hparams* hp, float* td_out,
                  float* loss_out, float* grad_out, ts_stream_t stream) {
    return dqn_update_impl(ws, params, adam_m, adam_v, adam_step, c, h, w, n_act, obs_nhwc, obs_u8, act, returns, weight, B, hp,
                           td_out, loss_out, grad_out, stream, nullptr, "ts_dqn_update");
}

int ts_dqn_wait_td(ts_workspace* ws, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dqn_wait_td: workspace is NULL");
    TS_REQUIRE(ws->td_ev_ready, TS_ERR_INVALID_ARG, "ts_dqn_wait_td: no update call has run on this workspace");
    TS_HIP_CHECK(hipStreamWaitEvent(ts::as_stream(stream), ws->td_ev, 0));
    return TS_OK;
}

int64_t ts_dqn_cache_bytes(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t B) {
    Net n;
    if (B < 1 || make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n) != TS_OK) return -1;
    return (int64_t)fwd_scratch_bytes(n, B);
}

int ts_dqn_forward_cache(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                         const void* obs_nhwc, int obs_u8, int64_t B, void* cache, int64_t cache_bytes, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_dqn_forward_cache: workspace is NULL");
    TS_REQUIRE(B >= 1 && params && obs_nhwc && cache, TS_ERR_INVALID_ARG, "ts_dqn_forward_cache: bad argument");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(cache) & 255u) == 0, TS_ERR_INVALID_ARG, "ts_dqn_forward_cache: cache must be 256-byte aligned");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    TS_REQUIRE(cache_bytes >= (int64_t)fwd_scratch_bytes(n, B), TS_ERR_SHAPE, "ts_dqn_forward_cache: cache holds %lld bytes, %lld needed",
               (long long)cache_bytes, (long long)fwd_scratch_bytes(n, B));
    Scratch sc;
    carve_fwd(n, B, static_cast<char*>(cache), &sc);
    return net_forward(ts::as_stream(stream), ws, n, params, obs_nhwc, obs_u8 != 0, B, sc, sc.q, nullptr);
}

int ts_dqn_update_cached(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                         int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act,
                         const float* returns, const float* weight, int64_t B, const ts_dqn_hparams* hp, void* cache,
                         float* td_out, float* loss_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(cache != nullptr, TS_ERR_INVALID_ARG, "ts_dqn_update_cached: cache is NULL");
    return dqn_update_impl(ws, params, adam_m, adam_v, adam_step, c, h, w, n_act, obs_nhwc, obs_u8, act, returns, weight, B, hp,
                           td_out, loss_out, grad_out, stream, cache, "ts_dqn_update_cached");
}

// ---- one call per update on a device-resident frame replay (uniform or prioritized) -----------------------------------------
namespace {
struct LearnBatch { int64_t* idx; int64_t* act; uint8_t* obs; uint8_t* obs_next; float* mask; double* gpow; double* mc; double* u;
                    double* w64; float* w32; };
// the per-update scalars of a captured update, on the device: the Philox counter of the batch prepared ahead, Adam's two
// step-dependent scalars (ts::adam_step_scalars), and the loss before it is copied to the caller
struct LearnCtl { uint64_t counter_next; float adam[2]; float loss; };
struct LearnScratch { LearnBatch b[2]; float* returns; float* td; int* err; LearnCtl* ctl; void* cache; size_t cache_bytes; };

static size_t dqn_learn_carve(char* base, const Net& n, int64_t B, int64_t obs_elems, LearnScratch* out) {
    char* p = base;
    auto bytes = [&](size_t nbytes) { char* q = p; p += align_up(nbytes); return q; };
    LearnScratch sc{};
    for (int k = 0; k < 2; ++k) {
        sc.b[k].idx = reinterpret_cast<int64_t*>(bytes(8 * (size_t)B));
        sc.b[k].act = reinterpret_cast<int64_t*>(bytes(8 * (size_t)B));
        sc.b[k].obs = reinterpret_cast<uint8_t*>(bytes((size_t)(B * obs_elems)));
        sc.b[k].obs_next = reinterpret_cast<uint8_t*>(bytes((size_t)(B * obs_elems)));
        sc.b[k].mask = reinterpret_cast<float*>(bytes(4 * (size_t)B));
        sc.b[k].gpow = reinterpret_cast<double*>(bytes(8 * (size_t)B));
        sc.b[k].mc = reinterpret_cast<double*>(bytes(8 * (size_t)B));
        sc.b[k].u = reinterpret_cast<double*>(bytes(8 * (size_t)B));
        sc.b[k].w64 = reinterpret_cast<double*>(bytes(8 * (size_t)B));
        sc.b[k].w32 = reinterpret_cast<float*>(bytes(4 * (size_t)B));
    }
    sc.returns = reinterpret_cast<float*>(bytes(4 * (size_t)B));
    sc.td = reinterpret_cast<float*>(bytes(4 * (size_t)B));
    sc.err = reinterpret_cast<int*>(bytes(256));
    sc.ctl = reinterpret_cast<LearnCtl*>(bytes(256));
    sc.cache_bytes = fwd_scratch_bytes(n, B);
    sc.cache = bytes(sc.cache_bytes);
    if (out) *out = sc;
    return (size_t)(p - base);
}

// batch.weight as `_update_with_batch` sees it: to_torch_as(weight float64, q float32) (dqn.py:392-394)
static __global__ __launch_bounds__(256) void dqn_weight_f32_kernel(const double* __restrict__ w64, int64_t n, float* __restrict__ w32) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w32[i] = (float)w64[i];
}

static __global__ void dqn_learn_ctl_kernel(LearnCtl* ctl, uint64_t counter_next, float lr_step, float bc2_sqrt) {
    ctl->counter_next = counter_next;
    ctl->adam[0] = lr_step;
    ctl->adam[1] = bc2_sqrt;
}

// Everything a captured update has baked in: a call whose key differs drops the graphs and captures afresh.
struct LearnKey {
    ts_workspace* ws; ts_workspace* ws_aux; float* params; float* params_old; float* adam_m; float* adam_v;
    int64_t c, h, w, n_act, B, n_step; double gamma; int64_t is_double; ts_frame_replay rb; ts_dqn_hparams hp; uint64_t seed;
    void* scratch; hipStream_t stream;
};
struct LearnGraphs {
    LearnKey key;
    int warm;                       // calls run through the streams with this key (allocations and attributes are settled after two)
    bool disabled;                  // a capture failed once: the streams from now on
    hipGraphExec_t exec[2][2];      // [counter parity][periodic target sync]
    int64_t launches;               // updates replayed from a graph so far (ts_dqn_learn_graph_launches)
    hipStream_t origin;             // the stream the enqueue is captured from
};
static void learn_graphs_drop(LearnGraphs* g) {
    for (auto& row : g->exec)
        for (auto& e : row)
            if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
}
static void learn_graphs_free(void* p) {
    auto* g = static_cast<LearnGraphs*>(p);
    learn_graphs_drop(g);
    if (g->origin) (void)hipStreamDestroy(g->origin);
    delete g;
}
}  // namespace

int64_t ts_dqn_learn_scratch_bytes(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t B) {
    Net n;
    if (B < 1 || make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n) != TS_OK) return -1;
    return (int64_t)dqn_learn_carve(reinterpret_cast<char*>((uintptr_t)256), n, B, c * h * w, nullptr);
}

int ts_dqn_learn_step(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                      float* adam_v, int64_t adam_step, int64_t c, int64_t h, int64_t w, int64_t n_act,
                      const ts_frame_replay* rb, int64_t B, int64_t n_step, double gamma, int is_double, const ts_dqn_hparams* hp,
                      uint64_t seed, uint64_t counter, int prepared, void* scratch, int64_t scratch_bytes, float* td_out,
                      float* loss_out, int64_t* idx_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr && ws_aux != nullptr && ws != ws_aux, TS_ERR_WORKSPACE,
               "ts_dqn_learn_step: two distinct workspaces (the update's, and the one of the ahead-of-time forward pass and the replay "
               "stream)");
    TS_REQUIRE(params && adam_m && adam_v && rb && hp && scratch && loss_out && B >= 1 && adam_step >= 1, TS_ERR_INVALID_ARG,
               "ts_dqn_learn_step: bad argument");
    TS_REQUIRE(rb->offset && rb->lengths && rb->last_index && rb->done && rb->terminated && rb->rew && rb->frames && rb->act_col &&
                   rb->E >= 1 && rb->slots >= 1, TS_ERR_INVALID_ARG, "ts_dqn_learn_step: incomplete replay view");
    TS_REQUIRE(rb->tree == nullptr || (rb->prio_minmax && rb->bound >= 1), TS_ERR_INVALID_ARG,
               "ts_dqn_learn_step: a sum tree needs its bound and its {max, min} priority pair");
    TS_REQUIRE(rb->plane_elems == h * w, TS_ERR_SHAPE, "ts_dqn_learn_step: frames of %lld elements for a %lld x %lld network input",
               (long long)rb->plane_elems, (long long)h, (long long)w);
    TS_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255u) == 0, TS_ERR_INVALID_ARG,
               "ts_dqn_learn_step: scratch must be 256-byte aligned");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    LearnScratch sc;
    const size_t need = dqn_learn_carve(static_cast<char*>(scratch), n, B, c * h * w, &sc);
    TS_REQUIRE(scratch_bytes >= (int64_t)need, TS_ERR_SHAPE, "ts_dqn_learn_step: scratch holds %lld bytes, %lld needed",
               (long long)scratch_bytes, (long long)need);
    hipStream_t s = ts::as_stream(stream), side, side2, replay;
    if (int rc = ts::side_streams(ws, s, &side, &side2)) return rc;
    // the replay stream: the aux workspace's own first side stream (== s while ts_profile_begin is active on `ws`)
    if (ws->profiling) replay = s;
    else if (int rc = ts::side_stream(ws_aux, s, &replay)) return rc;

    // One update, enqueued on the streams.  captured: the same enqueue under stream capture -- the per-update scalars come from
    // sc.ctl, the loss stays in sc.ctl, and the replay stream joins `s` at the end (a graph has one end).
    auto body = [&](bool captured, hipStream_t s) -> int {
        // buffer.sample_indices (prio.py:63-67 / buffer_base.py:505-533) + get_weight (prio.py:69-79) -> batch.act / obs / obs_next
        // -> the network-free half of compute_nstep_return, for update `ctr`
        auto prepare = [&](hipStream_t st, const LearnBatch& b, uint64_t ctr, bool ctr_from_ctl) -> int {
            if (rb->tree) {
                if (int rc = ts::uniform_fill_f64(b.u, B, seed, ctr, ctr_from_ctl ? &sc.ctl->counter_next : nullptr, st)) return rc;
                if (int rc = ts_per_sample(ws_aux, rb->tree, rb->bound, b.u, B, rb->prio_minmax, rb->beta, rb->weight_norm, b.idx,
                                           b.w64, st))
                    return rc;
                hipLaunchKernelGGL(dqn_weight_f32_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, st, b.w64, B, b.w32);
                TS_LAUNCH_CHECK();
            } else if (int rc = ts_sample_indices_seeded(rb->offset, rb->E, rb->lengths, seed, ctr, B, b.idx, sc.err, st)) {
                return rc;
            }
            if (int rc = ts_gather_rows(rb->act_col, rb->slots, 8, b.idx, B, b.act, st)) return rc;
            if (int rc = ts_dqn_gather_pair(rb->frames, rb->slots, rb->plane_elems, b.idx, B, n_step, c, rb->offset, rb->E, rb->done,
                                            rb->last_index, rb->lengths, b.obs, b.obs_next, st))
                return rc;
            return ts_nstep_coefficients(b.idx, B, n_step, rb->offset, rb->E, rb->done, rb->terminated, rb->last_index, rb->lengths,
                                         rb->rew, gamma, b.mask, b.gpow, b.mc, st);
        };
        const LearnBatch& cur = sc.b[counter & 1];
        const LearnBatch& nxt = sc.b[(counter + 1) & 1];
        if (!prepared)
            if (int rc = prepare(s, cur, counter, false)) return rc;
        // Q_online(batch.obs) of the update on the second side stream, beside the two obs_next passes of _target_q
        if (int rc = ts::stream_wait(ws, s, side2, 6)) return rc;
        if (int rc = ts_dqn_forward_cache(ws->profiling ? ws : ws_aux, params, c, h, w, n_act, cur.obs, 1, B, sc.cache,
                                          (int64_t)sc.cache_bytes, side2))
            return rc;
        if (int rc = ts_dqn_target_returns(ws, params, params_old, c, h, w, n_act, cur.obs_next, 1, B, is_double, cur.mask, cur.gpow,
                                           cur.mc, sc.returns, s))
            return rc;
        if (sync_target && params_old)       // the periodic hard sync sits between _preprocess_batch and the update (dqn.py:283-285)
            TS_HIP_CHECK(hipMemcpyAsync(params_old, params, sizeof(float) * (size_t)n.total, hipMemcpyDeviceToDevice, s));
        if (int rc = ts::stream_wait(ws, side2, s, 7)) return rc;
        if (int rc = dqn_update_impl(ws, params, adam_m, adam_v, adam_step, c, h, w, n_act, cur.obs, 1, cur.act, sc.returns,
                                     rb->tree ? cur.w32 : nullptr, B, hp, sc.td, captured ? &sc.ctl->loss : loss_out, nullptr, reinterpret_cast<ts_stream_t>(s),
                                     sc.cache, "ts_dqn_learn_step", captured ? sc.ctl->adam : nullptr))
            return rc;
        // _postprocess_batch (prio.py:81-100: update_weight with the TD errors) and the next update's batch on the replay stream,
        // behind the loss kernel (ts::record_td) and beside the backward pass and the optimizer step
        if (replay != s) TS_HIP_CHECK(hipStreamWaitEvent(replay, ws->td_ev, 0));
        if (rb->tree)
            if (int rc = ts_per_update_weight(ws_aux, rb->tree, rb->bound, cur.idx, sc.td, B, rb->alpha, rb->prio_minmax, replay))
                return rc;
        if (int rc = prepare(replay, nxt, counter + 1, captured)) return rc;
        if (captured) return ts::stream_wait(ws_aux, replay, s, 9);
        return TS_OK;
    };
    // (the TD errors stay in `scratch` for the replay stream: the caller's copy may be released before that stream is done)
    auto copies = [&](bool captured) -> int {
        const LearnBatch& cur = sc.b[counter & 1];
        if (captured) TS_HIP_CHECK(hipMemcpyAsync(loss_out, &sc.ctl->loss, 4, hipMemcpyDeviceToDevice, s));
        if (td_out) TS_HIP_CHECK(hipMemcpyAsync(td_out, sc.td, 4 * (size_t)B, hipMemcpyDeviceToDevice, s));
        if (idx_out) TS_HIP_CHECK(hipMemcpyAsync(idx_out, cur.idx, 8 * (size_t)B, hipMemcpyDeviceToDevice, s));
        return TS_OK;
    };

    // behind everything the previous call left on the replay stream: its priority update and this update's batch
    if (int rc = ts::stream_wait(ws_aux, replay, s, 8)) return rc;

    // Steady state (same arguments as the calls before, batch prepared ahead): the update can be captured once per (counter
    // parity, periodic sync) and replayed -- one graph launch instead of ~45 kernel launches and ~20 event operations per update.
    // Measured on ROCm 7.2 (profiles/r06_dqn_learn_step_ab.txt): hipGraphLaunch spends as long on the host as the launches it
    // replaces (0.60 vs 0.56 ms) and the replay runs slower than the hand-placed streams (1,520 vs 1,615 updates/s), so the
    // graphs are opt-in: TS_DQN_GRAPH=1.
    const char* graph_env = getenv("TS_DQN_GRAPH");
    const bool graphs_on = graph_env && graph_env[0] == '1';
    if (!ws->learn_graphs) {
        ws->learn_graphs = new LearnGraphs();
        ws->learn_graphs_free = learn_graphs_free;
    }
    auto* lg = static_cast<LearnGraphs*>(ws->learn_graphs);
    LearnKey key;
    memset(&key, 0, sizeof(key));
    key.ws = ws; key.ws_aux = ws_aux; key.params = params; key.params_old = params_old; key.adam_m = adam_m; key.adam_v = adam_v;
    key.c = c; key.h = h; key.w = w; key.n_act = n_act; key.B = B; key.n_step = n_step; key.gamma = gamma; key.is_double = is_double;
    key.rb = *rb; key.rb.reserved = 0; key.hp = *hp; key.seed = seed; key.scratch = scratch; key.stream = s;
    if (memcmp(&key, &lg->key, sizeof(key)) != 0) {
        learn_graphs_drop(lg);
        lg->key = key;
        lg->warm = 0;
    }
    const bool steady = graphs_on && !lg->disabled && !ws->profiling && prepared && rb->tree && replay != s && side != s &&
                        side2 != s && hp->lr >= 0.0;
    if (!steady || lg->warm < 2) {
        if (int rc = body(false, s)) return rc;
        ++lg->warm;
        return copies(false);
    }
    hipGraphExec_t& exec = lg->exec[counter & 1][sync_target && params_old ? 1 : 0];
    if (!exec) {
        hipGraph_t graph = nullptr;
        const bool verbose = getenv("TS_DQN_GRAPH_VERBOSE") != nullptr;
        // (the caller's stream may be the null stream, which cannot capture: the enqueue is recorded from a stream of our own --
        // a graph does not remember its origin stream)
        if (!lg->origin) TS_HIP_CHECK(hipStreamCreateWithFlags(&lg->origin, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(lg->origin, hipStreamCaptureModeThreadLocal);
        bool ok = e == hipSuccess;
        if (!ok && verbose) fprintf(stderr, "ts_dqn_learn_step: hipStreamBeginCapture: %s\n", hipGetErrorString(e));
        if (ok) {
            const int rc = body(true, lg->origin);
            e = hipStreamEndCapture(lg->origin, &graph);
            ok = rc == TS_OK && e == hipSuccess && graph != nullptr;
            if (!ok && verbose)
                fprintf(stderr, "ts_dqn_learn_step: capture body rc %d (%s), hipStreamEndCapture: %s\n", rc, ts_last_error(),
                        hipGetErrorString(e));
        }
        if (ok) {
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            ok = e == hipSuccess;
            if (!ok && verbose) fprintf(stderr, "ts_dqn_learn_step: hipGraphInstantiate: %s\n", hipGetErrorString(e));
        }
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {            // nothing of the capture ran: the streams, from now on
            (void)hipGetLastError();
            exec = nullptr;
            lg->disabled = true;
            if (getenv("TS_DQN_GRAPH_VERBOSE")) fprintf(stderr, "ts_dqn_learn_step: stream capture failed, staying on the streams\n");
            if (int rc = body(false, s)) return rc;
            return copies(false);
        }
    }
    float sc2[2];
    ts::adam_step_scalars(adam_step, hp->lr, hp->beta1, hp->beta2, sc2);
    hipLaunchKernelGGL(dqn_learn_ctl_kernel, dim3(1), dim3(1), 0, s, sc.ctl, counter + 1, sc2[0], sc2[1]);
    TS_LAUNCH_CHECK();
    TS_HIP_CHECK(hipGraphLaunch(exec, s));
    ++lg->launches;
    return copies(true);
}

// The same update for a batch the CALLER drew (a host PrioritizedVectorReplayBuffer's sample_indices + get_weight: HipDQN.update()
// at hook level): batch.act / obs / obs_next gathered by index, n-step returns, the update -- what ts_dqn_learn_step does between
// its sampling and its priority update, as one call.  `rb->tree` is not looked at; TD errors go to td_out (the caller's
// _postprocess_batch hands them to buffer.update_weight).
int ts_dqn_learn_rows(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                      float* adam_v, int64_t adam_step, int64_t c, int64_t h, int64_t w, int64_t n_act,
                      const ts_frame_replay* rb, const int64_t* indices, const float* weight, int64_t B, int64_t n_step,
                      double gamma, int is_double, const ts_dqn_hparams* hp, void* scratch, int64_t scratch_bytes,
                      float* returns_out, float* td_out, float* loss_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr && ws_aux != nullptr && ws != ws_aux, TS_ERR_WORKSPACE,
               "ts_dqn_learn_rows: two distinct workspaces (the update's, and the one of the forward pass beside the target passes)");
    TS_REQUIRE(params && adam_m && adam_v && rb && hp && scratch && loss_out && indices && B >= 1 && adam_step >= 1,
               TS_ERR_INVALID_ARG, "ts_dqn_learn_rows: bad argument");
    TS_REQUIRE(rb->offset && rb->lengths && rb->last_index && rb->done && rb->terminated && rb->rew && rb->frames && rb->act_col &&
                   rb->E >= 1 && rb->slots >= 1, TS_ERR_INVALID_ARG, "ts_dqn_learn_rows: incomplete replay view");
    TS_REQUIRE(rb->plane_elems == h * w, TS_ERR_SHAPE, "ts_dqn_learn_rows: frames of %lld elements for a %lld x %lld network input",
               (long long)rb->plane_elems, (long long)h, (long long)w);
    TS_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255u) == 0, TS_ERR_INVALID_ARG,
               "ts_dqn_learn_rows: scratch must be 256-byte aligned");
    Net n;
    if (int rc = make_net((int)B, (int)c, (int)h, (int)w, (int)n_act, &n)) return rc;
    LearnScratch sc;
    const size_t need = dqn_learn_carve(static_cast<char*>(scratch), n, B, c * h * w, &sc);
    TS_REQUIRE(scratch_bytes >= (int64_t)need, TS_ERR_SHAPE, "ts_dqn_learn_rows: scratch holds %lld bytes, %lld needed",
               (long long)scratch_bytes, (long long)need);
    hipStream_t s = ts::as_stream(stream), side, side2;
    if (int rc = ts::side_streams(ws, s, &side, &side2)) return rc;
    const LearnBatch& cur = sc.b[0];
    if (int rc = ts_gather_rows(rb->act_col, rb->slots, 8, indices, B, cur.act, s)) return rc;
    if (int rc = ts_dqn_gather_pair(rb->frames, rb->slots, rb->plane_elems, indices, B, n_step, c, rb->offset, rb->E, rb->done,
                                    rb->last_index, rb->lengths, cur.obs, cur.obs_next, s))
        return rc;
    if (int rc = ts_nstep_coefficients(indices, B, n_step, rb->offset, rb->E, rb->done, rb->terminated, rb->last_index, rb->lengths,
                                       rb->rew, gamma, cur.mask, cur.gpow, cur.mc, s))
        return rc;
    // Q_online(batch.obs) of the update on the second side stream, beside the two obs_next passes of _target_q
    if (int rc = ts::stream_wait(ws, s, side2, 6)) return rc;
    if (int rc = ts_dqn_forward_cache(ws->profiling ? ws : ws_aux, params, c, h, w, n_act, cur.obs, 1, B, sc.cache,
                                      (int64_t)sc.cache_bytes, side2))
        return rc;
    if (int rc = ts_dqn_target_returns(ws, params, params_old, c, h, w, n_act, cur.obs_next, 1, B, is_double, cur.mask, cur.gpow,
                                       cur.mc, sc.returns, s))
        return rc;
    if (sync_target && params_old)       // the periodic hard sync sits between _preprocess_batch and the update (dqn.py:283-285)
        TS_HIP_CHECK(hipMemcpyAsync(params_old, params, sizeof(float) * (size_t)n.total, hipMemcpyDeviceToDevice, s));
    if (int rc = ts::stream_wait(ws, side2, s, 7)) return rc;
    if (int rc = dqn_update_impl(ws, params, adam_m, adam_v, adam_step, c, h, w, n_act, cur.obs, 1, cur.act, sc.returns, weight, B, hp,
                                 sc.td, loss_out, nullptr, reinterpret_cast<ts_stream_t>(s), sc.cache, "ts_dqn_learn_rows", nullptr))
        return rc;
    if (returns_out) TS_HIP_CHECK(hipMemcpyAsync(returns_out, sc.returns, 4 * (size_t)B, hipMemcpyDeviceToDevice, s));
    if (td_out) TS_HIP_CHECK(hipMemcpyAsync(td_out, sc.td, 4 * (size_t)B, hipMemcpyDeviceToDevice, s));
    return TS_OK;
}

int64_t ts_dqn_learn_graph_launches(ts_workspace* ws) {
    if (!ws || !ws->learn_graphs) return 0;
    auto* lg = static_cast<LearnGraphs*>(ws->learn_graphs);
    return lg->disabled ? -1 : lg->launches;
}

}  // extern "C"
