// ts_ppo.hip -- fused PPO/A2C actor-critic MLP step for gfx950 (fp32 MFMA).
//
// Replaces, for the MLP actor-critic of examples/mujoco/mujoco_ppo.py (obs -> 64 -> 64 -> {mu, V},
// tanh, state-independent log-sigma):
//   * the no-grad passes of _add_returns_and_advantages / PPO._preprocess_batch
//     (tianshou/algorithm/modelfree/a2c.py:122-129, ppo.py:157-161)          -> ppo_infer_kernel
//   * one minibatch iteration of PPO._update_with_batch (ppo.py:179-216): forward, clipped
//     surrogate / value / entropy loss, backward                                -> ppo_step_kernel
//   * Optimizer.step (algorithm_base.py:484-500): clip_grad_norm_ + Adam     -> ppo_reduce_slabs_kernel
//                                                                                  + ppo_adam_kernel
//
// Roofline: fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TF/s peak).  Algorithmic work of one update
// step = 60,544 flop / sample; the kernel issues 484 MFMAs (4096 flop each) per
// 32-sample tile.
//
// Data layout in a wave: activations are kept TRANSPOSED, H^T[feature, sample] = W . X^T, so that
// the MFMA's C/D layout (lane = sample column, 16 registers = 16 feature rows) of one layer is
// directly the B operand of the next layer: K is a summation index, so the k-step (t, r) is simply
// *defined* to carry feature 32 t + F(r, half) with F(r, h) = (r & 3) + 8 (r >> 2) + 4 h, and the
// weights (A operand, from LDS) are read in that order.  No cross-lane traffic between layers.
// Weight gradients contract over samples, so dZ and H are transposed once through wave-private
// LDS tiles ([feature][sample], pitch 36 floats: conflict-free ds_write_b32 and ds_read_b128).
#include "ts_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HID = 64;
constexpr int W2_PITCH = 68;                 // conflict-free ds_read_b128 of 4 consecutive k
constexpr int W2_SIZE = HID * W2_PITCH;      // floats per net
constexpr int ACT_PAD = 8;
constexpr int TILE_PITCH = 36;               // transposed activation tile pitch (floats)
constexpr int TILE_SIZE = 32 * TILE_PITCH;   // one 32-feature x 32-sample tile
constexpr int STEP_THREADS = 512;            // 8 waves, 2 per SIMD
constexpr int STEP_WAVES = STEP_THREADS / 64;
constexpr int N_EXTRA = 2;                   // slab tail: clip-loss sum, vf-loss sum

__device__ __forceinline__ int featF(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct Dims {
    int obs, act;
    // flat parameter offsets (floats), see include/tsengine.h
    int a_w1, a_b1, a_w2, a_b2, a_wmu, a_bmu, a_sig, p_actor;
    int c_w1, c_b1, c_w2, c_b2, c_wv, c_bv, p_total;
};

__host__ __device__ inline Dims make_dims(int obs, int act) {
    Dims d;
    d.obs = obs; d.act = act;
    d.a_w1 = 0;
    d.a_b1 = d.a_w1 + HID * obs;
    d.a_w2 = d.a_b1 + HID;
    d.a_b2 = d.a_w2 + HID * HID;
    d.a_wmu = d.a_b2 + HID;
    d.a_bmu = d.a_wmu + act * HID;
    d.a_sig = d.a_bmu + act;
    d.p_actor = d.a_sig + act;
    d.c_w1 = d.p_actor;
    d.c_b1 = d.c_w1 + HID * obs;
    d.c_w2 = d.c_b1 + HID;
    d.c_b2 = d.c_w2 + HID * HID;
    d.c_wv = d.c_b2 + HID;
    d.c_bv = d.c_wv + HID;
    d.p_total = d.c_bv + 1;
    return d;
}

// LDS carve (floats) shared by the step and inference kernels
template <int KS1>
struct Lds {
    static constexpr int W2 = 0;                                  // [2][64*68]
    static constexpr int W1 = W2 + 2 * W2_SIZE;                   // [2][2*KS1][64]  (k-major)
    static constexpr int W1_NET = 2 * KS1 * HID;
    static constexpr int B2 = W1 + 2 * W1_NET;                    // [2][64]
    static constexpr int WH = B2 + 2 * HID;                       // [2][h][t][r][8]
    static constexpr int WH_NET = 2 * 2 * 16 * ACT_PAD;
    static constexpr int SMALL = WH + 2 * WH_NET;                 // bmu[8] sig[8] bv[1] pad -> 32
    static constexpr int INFER_END = SMALL + 32;
    static constexpr int ACC = INFER_END;                         // net accumulator (flat layout)
};

// ---------------------------------------------------------------------------------------------
// weight staging: global flat params -> LDS images
template <int KS1>
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ params,
                                              const Dims& d, int nthreads) {
    using L = Lds<KS1>;
    const int tid = threadIdx.x;
    for (int net = 0; net < 2; ++net) {
        const int w1 = net ? d.c_w1 : d.a_w1, b1 = net ? d.c_b1 : d.a_b1;
        const int w2 = net ? d.c_w2 : d.a_w2, b2 = net ? d.c_b2 : d.a_b2;
        for (int i = tid; i < HID * HID; i += nthreads) {
            const int r = i >> 6, c = i & 63;
            lds[L::W2 + net * W2_SIZE + r * W2_PITCH + c] = params[w2 + i];
        }
        for (int i = tid; i < 2 * KS1 * HID; i += nthreads) {
            const int k = i >> 6, row = i & 63;
            float v = 0.f;
            if (k < d.obs) v = params[w1 + row * d.obs + k];
            else if (k == d.obs) v = params[b1 + row];
            lds[L::W1 + net * L::W1_NET + i] = v;
        }
        for (int i = tid; i < HID; i += nthreads) lds[L::B2 + net * HID + i] = params[b2 + i];
        for (int i = tid; i < L::WH_NET; i += nthreads) {
            const int a = i & 7, r = (i >> 3) & 15, t = (i >> 7) & 1, h = (i >> 8) & 1;
            const int f = 32 * t + featF(r, h);
            float v = 0.f;
            if (net == 0) { if (a < d.act) v = params[d.a_wmu + a * HID + f]; }
            else { if (a == 0) v = params[d.c_wv + f]; }
            lds[L::WH + net * L::WH_NET + i] = v;
        }
    }
    for (int i = tid; i < 32; i += nthreads) {
        float v = 0.f;
        if (i < 8) { if (i < d.act) v = params[d.a_bmu + i]; }
        else if (i < 16) { if (i - 8 < d.act) v = params[d.a_sig + (i - 8)]; }
        else if (i == 16) v = params[d.c_bv];
        lds[L::SMALL + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// forward trunk of one net for one 32-sample tile: x -> h1 -> h2 (transposed, in registers)
template <int KS1>
__device__ __forceinline__ void trunk_forward(const float* lds, int net, const float (&x)[KS1],
                                              int i, int h, f32x16 (&h1)[2], f32x16 (&h2)[2]) {
    using L = Lds<KS1>;
    const float* w1 = lds + L::W1 + net * L::W1_NET;
    const float* w2 = lds + L::W2 + net * W2_SIZE;
    const float* b2 = lds + L::B2 + net * HID;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS1; ++s) acc = mfma32(w1[(KS1 * h + s) * HID + 32 * t + i], x[s], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = tanhf(acc[r]);
        h1[t] = acc;
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + 32 * t2 + 8 * g + 4 * h);
            acc[4 * g + 0] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(w2 + (32 * t2 + i) * W2_PITCH + 32 * t + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma32(a[q], h1[t][4 * g + q], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = tanhf(acc[r]);
        h2[t2] = acc;
    }
}

// head on the VALU: out[a] = sum_f h2[f] * WH[a][f] over the lane's 32 features, halves combined
template <int KS1, int NA>
__device__ __forceinline__ void head_forward(const float* lds, int net, int h, const f32x16 (&h2)[2],
                                             float (&out)[NA]) {
    using L = Lds<KS1>;
    const float* wh = lds + L::WH + net * L::WH_NET + h * (2 * 16 * ACT_PAD);
#pragma unroll
    for (int a = 0; a < NA; ++a) out[a] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* p = wh + (t * 16 + r) * ACT_PAD;
            if constexpr (NA == 1) {
                out[0] += h2[t][r] * p[0];
            } else {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(p);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    out[a] += h2[t][r] * w0[a];
                    if (a + 4 < NA) out[a + 4] += h2[t][r] * w1[a];
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) out[a] += __shfl_xor(out[a], 32, 64);
}

template <int KS1>
__device__ __forceinline__ void load_x(const float* __restrict__ obs, int64_t row, int obs_dim, int h,
                                       float (&x)[KS1]) {
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int k = KS1 * h + s;
        float v = 0.f;
        if (k < obs_dim) v = obs[row * obs_dim + k];
        else if (k == obs_dim) v = 1.f;
        x[s] = v;
    }
}

constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;  // math.log(math.sqrt(2*pi))

// Normal(mu, sigma).log_prob(act).sum(-1)  (torch.distributions.Normal.log_prob, Independent)
__device__ __forceinline__ float gaussian_logp(const float (&mu)[ACT_PAD], const float (&act)[ACT_PAD],
                                               const float* sig, int act_dim) {
    float lp = 0.f;
#pragma unroll
    for (int a = 0; a < ACT_PAD; ++a) {
        if (a < act_dim) {
            const float sigma = expf(sig[a]);
            const float var = sigma * sigma;
            const float dlt = act[a] - mu[a];
            lp += -(dlt * dlt) / (2.f * var) - logf(sigma) - LOG_SQRT_2PI;
        }
    }
    return lp;
}

// ---------------------------------------------------------------------------------------------
// inference: v_out = V(obs), logp_out = log pi(act | obs)
template <int KS1>
__global__ __launch_bounds__(512) void ppo_infer_kernel(const float* __restrict__ params, Dims d,
                                                        const float* __restrict__ obs,
                                                        const float* __restrict__ act, int64_t n,
                                                        float* __restrict__ v_out,
                                                        float* __restrict__ logp_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<KS1>;
    stage_weights<KS1>(lds, params, d, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int64_t n_tiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * waves) {
        const int64_t srow = tile * 32 + i;
        const bool valid = srow < n;
        const int64_t row = valid ? srow : n - 1;
        float x[KS1];
        load_x<KS1>(obs, row, d.obs, h, x);
        f32x16 h1[2], h2[2];
        if (v_out) {
            trunk_forward<KS1>(lds, 1, x, i, h, h1, h2);
            float v[1];
            head_forward<KS1, 1>(lds, 1, h, h2, v);
            if (valid && h == 0) v_out[srow] = v[0] + lds[L::SMALL + 16];
        }
        if (logp_out) {
            trunk_forward<KS1>(lds, 0, x, i, h, h1, h2);
            float mu[ACT_PAD], a[ACT_PAD];
            head_forward<KS1, ACT_PAD>(lds, 0, h, h2, mu);
#pragma unroll
            for (int k = 0; k < ACT_PAD; ++k) {
                mu[k] += lds[L::SMALL + k];
                a[k] = (k < d.act) ? act[row * d.act + k] : 0.f;
            }
            const float lp = gaussian_logp(mu, a, lds + L::SMALL + 8, d.act);
            if (valid && h == 0) logp_out[srow] = lp;
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct StepArgs {
    const float* params;
    const float* obs;
    const float* act;
    const float* adv;
    const float* ret;
    const float* logp_old;
    const float* v_old;
    const int64_t* rows;      // minibatch row ids (perm slice) or NULL = identity
    int64_t n_rows;           // rows in this minibatch (local)
    float inv_batch;          // 1 / global minibatch size
    const float* adv_stats;   // {mean, std} of this minibatch (device) or NULL
    float eps_clip, dual_clip, vf_coef, ent_coef;
    int value_clip, adv_norm;
    float* slabs;             // [gridDim.x][slab_w]
    int slab_w;
};

// transposed tile write: lane (j, h) register r -> T[F(r,h)][j]
__device__ __forceinline__ void tile_write(float* tile, const f32x16& v, int j, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[featF(r, h) * TILE_PITCH + j] = v[r];
}

// 16 k-step operands of lane (i, h): T[i][16h .. 16h+15]
__device__ __forceinline__ void tile_read16(const float* tile, int i, int h, float (&o)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + i * TILE_PITCH + 16 * h + 4 * q);
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    // wave-private LDS hand-off between lanes of one wave: LDS executes a wave's instructions in
    // order; this only stops the compiler from moving accesses across the hand-off.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float half_sum32(float v) {  // sum over the 32 lanes of a half
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One net, one 32-sample tile: forward, loss, backward, weight gradients -> LDS accumulator.
template <int KS1, bool ACTOR>
__device__ __forceinline__ void net_tile(float* lds, float* acc, float* scratch, const StepArgs& g,
                                         const Dims& d, int64_t tile, int lane) {
    using L = Lds<KS1>;
    constexpr int net = ACTOR ? 0 : 1;
    const int i = lane & 31, h = lane >> 5;
    const int64_t srow = tile * 32 + i;
    const bool valid = srow < g.n_rows;
    const int64_t pos = valid ? srow : g.n_rows - 1;
    const int64_t row = g.rows ? g.rows[pos] : pos;

    float x[KS1];
    load_x<KS1>(g.obs, row, d.obs, h, x);
    f32x16 h1[2], h2[2];
    trunk_forward<KS1>(lds, net, x, i, h, h1, h2);

    // flat-layout offsets of this net inside the accumulator
    const int o_w1 = 0, o_b1 = HID * d.obs, o_w2 = o_b1 + HID, o_b2 = o_w2 + HID * HID;
    const int o_head = o_b2 + HID;                      // Wmu [act][64] | Wv [64]
    const int o_hb = o_head + (ACTOR ? d.act * HID : HID);  // bmu [act] | bv [1]
    const int o_sig = o_hb + d.act;                     // actor only
    const int p_net = ACTOR ? d.p_actor : (d.p_total - d.p_actor);

    constexpr int NA = ACTOR ? ACT_PAD : 1;
    float dout[NA];                                      // dL/d(head output) per sample
    if constexpr (ACTOR) {
        float mu[ACT_PAD], a[ACT_PAD];
        head_forward<KS1, ACT_PAD>(lds, 0, h, h2, mu);
        const float* sig = lds + L::SMALL + 8;
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            mu[k] += lds[L::SMALL + k];
            a[k] = (k < d.act) ? g.act[row * d.act + k] : 0.f;
        }
        const float logp = gaussian_logp(mu, a, sig, d.act);
        float A = g.adv[row];
        if (g.adv_norm) A = (A - g.adv_stats[0]) / (g.adv_stats[1] + 1e-8f);   // ppo.py:184-186
        const float ratio = expf(logp - g.logp_old[row]);                     // :187
        const float surr1 = ratio * A;
        const float lo = 1.f - g.eps_clip, hi = 1.f + g.eps_clip;
        const float surr2 = fminf(fmaxf(ratio, lo), hi) * A;                  // :190
        float term, dterm;  // term = -objective; dterm = d term / d ratio
        const float clip1 = fminf(surr1, surr2);
        float base = (surr1 <= surr2) ? A : 0.f;
        if (g.dual_clip > 0.f) {                                              // :191-194
            const float clip2 = fmaxf(clip1, g.dual_clip * A);
            if (A < 0.f) { term = -clip2; if (!(clip1 >= g.dual_clip * A)) base = 0.f; }
            else term = -clip1;
        } else {
            term = -clip1;                                                    // :196
        }
        dterm = -base;
        const float w = valid ? g.inv_batch : 0.f;
        const float dlogp = dterm * ratio * w;
        float dsig[ACT_PAD];
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            if (k < d.act) {
                const float sigma = expf(sig[k]);
                const float var = sigma * sigma;
                const float dlt = a[k] - mu[k];
                dout[k] = dlogp * dlt / var;
                dsig[k] = dlogp * (dlt * dlt / var - 1.f) - g.ent_coef * w;   // entropy: d/ds = 1
            } else {
                dout[k] = 0.f;
                dsig[k] = 0.f;
            }
        }
        // per-wave sums of the clip loss, d sigma, d bmu (count each sample once: half 0)
        float lsum = (h == 0) ? term * w : 0.f;
        lsum = half_sum32(lsum) ;
        if (lane == 0) atomicAdd(&acc[p_net], lsum);
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            if (k < d.act) {
                const float s1 = half_sum32(dsig[k]);
                const float s2 = half_sum32(dout[k]);
                if (lane == 0) {
                    atomicAdd(&acc[o_sig + k], s1);
                    atomicAdd(&acc[o_hb + k], s2);
                }
            }
        }
    } else {
        float v[1];
        head_forward<KS1, 1>(lds, 1, h, h2, v);
        const float value = v[0] + lds[L::SMALL + 16];
        const float ret = g.ret[row];
        float term, dv;
        const float vf1 = (ret - value) * (ret - value);
        if (g.value_clip) {                                                   // ppo.py:199-206
            const float vo = g.v_old[row];
            const float dvo = value - vo;
            const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
            const float vf2 = (ret - vclip) * (ret - vclip);
            term = fmaxf(vf1, vf2);
            // torch.max backward: the larger branch takes the gradient, ties split it; clamp
            // passes the gradient inside [-eps, eps].  Inside the range v_clip = vo + (v - vo)
            // differs from v by rounding, so either branch may win there.
            const float g1 = -2.f * (ret - value);
            const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
            dv = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        } else {
            term = vf1;                                                       // :208
            dv = -2.f * (ret - value);
        }
        const float w = valid ? g.inv_batch : 0.f;
        dout[0] = dv * g.vf_coef * w;
        float lsum = (h == 0) ? term * w : 0.f;
        lsum = half_sum32(lsum);
        const float bsum = half_sum32(dout[0]);
        if (lane == 0) {
            atomicAdd(&acc[p_net + 1], lsum);
            atomicAdd(&acc[o_hb], bsum);
        }
    }

    float* SA = scratch;               // A-side tile
    float* SB = scratch + TILE_SIZE;   // B-side tile

    // ---- head weight gradient: gW[a][f] = sum_s dout[s][a] * H2[s][f]  (lane = feature f)
    tile_write(SA, h2[0], i, h);
    tile_write(SB, h2[1], i, h);
    wave_lds_sync();
    {
        const float* rowp = (lane < 32 ? SA : SB) + (lane & 31) * TILE_PITCH;
        float hv[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 4 * q);
            hv[4 * q] = v[0]; hv[4 * q + 1] = v[1]; hv[4 * q + 2] = v[2]; hv[4 * q + 3] = v[3];
        }
        float gw[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) gw[a] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const float ds = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dout[a]), s));
                gw[a] += ds * hv[s];
            }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
            if (a < (ACTOR ? d.act : 1)) atomicAdd(&acc[o_head + a * HID + lane], gw[a]);
    }
    wave_lds_sync();

    // ---- dZ2 = (dout . Whead) * (1 - h2^2)   (in place in h2)
    {
        const float* wh = lds + L::WH + net * L::WH_NET + h * (2 * 16 * ACT_PAD);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* p = wh + (t * 16 + r) * ACT_PAD;
                float dh;
                if constexpr (ACTOR) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(p);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(p + 4);
                    dh = dout[0] * w0[0] + dout[1] * w0[1] + dout[2] * w0[2] + dout[3] * w0[3] +
                         dout[4] * w1[0] + dout[5] * w1[1] + dout[6] * w1[2] + dout[7] * w1[3];
                } else {
                    dh = dout[0] * p[0];
                }
                const float hv = h2[t][r];
                h2[t][r] = dh * (1.f - hv * hv);
            }
        }
    }

    // ---- dH1^T = W2^T . dZ2^T, then dZ1 = dH1 * (1 - h1^2)
    f32x16 dz1[2];
    {
        const float* w2 = lds + L::W2 + net * W2_SIZE;
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
            f32x16 accd = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accd = mfma32(w2[(32 * t + featF(r, h)) * W2_PITCH + 32 * t1 + i], h2[t][r], accd);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = h1[t1][r];
                accd[r] = accd[r] * (1.f - hv * hv);
            }
            dz1[t1] = accd;
        }
    }

    // ---- dW2[f2][f1] = sum_s dZ2[s][f2] H1[s][f1];  db2[f2] = sum_s dZ2[s][f2]
    auto dw2_pair = [&](int tM, int tN) {
        float av[16], bv[16];
        tile_read16(SA, i, h, av);
        tile_read16(SB, i, h, bv);
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) c = mfma32(av[s], bv[s], c);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            atomicAdd(&acc[o_w2 + (32 * tM + featF(r, h)) * HID + 32 * tN + i], c[r]);
    };
    auto db2_rows = [&](int tM) {
        if (lane < 32) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(SA + lane * TILE_PITCH + 4 * q);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            atomicAdd(&acc[o_b2 + 32 * tM + lane], s);
        }
    };
    tile_write(SA, h2[0], i, h);      // dZ2 rows 0..31
    tile_write(SB, h1[0], i, h);
    wave_lds_sync();
    dw2_pair(0, 0);
    db2_rows(0);
    wave_lds_sync();
    tile_write(SB, h1[1], i, h);
    wave_lds_sync();
    dw2_pair(0, 1);
    wave_lds_sync();
    tile_write(SA, h2[1], i, h);      // dZ2 rows 32..63
    wave_lds_sync();
    dw2_pair(1, 1);
    db2_rows(1);
    wave_lds_sync();
    tile_write(SB, h1[0], i, h);
    wave_lds_sync();
    dw2_pair(1, 0);
    wave_lds_sync();

    // ---- dW1aug[f1][k] = sum_s dZ1[s][f1] Xaug[s][k]   (k == obs is the bias column)
#pragma unroll
    for (int s = 0; s < KS1; ++s) SB[(KS1 * h + s) * TILE_PITCH + i] = x[s];
    if (2 * KS1 < 32) {  // rows never written by x: keep them finite
        for (int r = 2 * KS1 + h; r < 32; r += 2) SB[r * TILE_PITCH + i] = 0.f;
    }
#pragma unroll
    for (int tM = 0; tM < 2; ++tM) {
        tile_write(SA, dz1[tM], i, h);
        wave_lds_sync();
        float av[16], bv[16];
        tile_read16(SA, i, h, av);
        tile_read16(SB, i, h, bv);
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) c = mfma32(av[s], bv[s], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f1 = 32 * tM + featF(r, h);
            if (i < d.obs) atomicAdd(&acc[o_w1 + f1 * d.obs + i], c[r]);
            else if (i == d.obs) atomicAdd(&acc[o_b1 + f1], c[r]);
        }
        wave_lds_sync();
    }
}

template <int KS1>
__global__ __launch_bounds__(STEP_THREADS, 2) void ppo_step_kernel(StepArgs g, Dims d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<KS1>;
    const int acc_len = d.p_actor + N_EXTRA;           // actor is the larger net
    const int acc_pad = (acc_len + 3) & ~3;
    float* acc = lds + L::ACC;
    float* scratch = lds + L::ACC + acc_pad + (threadIdx.x >> 6) * (2 * TILE_SIZE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    stage_weights<KS1>(lds, g.params, d, STEP_THREADS);
    for (int k = threadIdx.x; k < acc_pad; k += STEP_THREADS) acc[k] = 0.f;
    __syncthreads();

    const int64_t n_tiles = (g.n_rows + 31) / 32;
    float* slab = g.slabs + (int64_t)blockIdx.x * g.slab_w;

    // ---- actor pass
    for (int64_t tile = (int64_t)blockIdx.x * STEP_WAVES + wave; tile < n_tiles;
         tile += (int64_t)gridDim.x * STEP_WAVES)
        net_tile<KS1, true>(lds, acc, scratch, g, d, tile, lane);
    __syncthreads();
    for (int k = threadIdx.x; k < d.p_actor; k += STEP_THREADS) slab[k] = acc[k];
    if (threadIdx.x == 0) slab[d.p_total] = acc[d.p_actor];           // clip-loss sum
    __syncthreads();
    for (int k = threadIdx.x; k < acc_pad; k += STEP_THREADS) acc[k] = 0.f;
    __syncthreads();

    // ---- critic pass
    const int p_critic = d.p_total - d.p_actor;
    for (int64_t tile = (int64_t)blockIdx.x * STEP_WAVES + wave; tile < n_tiles;
         tile += (int64_t)gridDim.x * STEP_WAVES)
        net_tile<KS1, false>(lds, acc, scratch, g, d, tile, lane);
    __syncthreads();
    for (int k = threadIdx.x; k < p_critic; k += STEP_THREADS) slab[d.p_actor + k] = acc[k];
    if (threadIdx.x == 0) slab[d.p_total + 1] = acc[p_critic + 1];    // vf-loss sum
}

// ---------------------------------------------------------------------------------------------
// slab reduction, stage 1: out[q][col] = sum over the q-th quarter of the slabs
__global__ __launch_bounds__(256) void ppo_reduce_slabs_kernel(const float* __restrict__ slabs,
                                                               int n_slabs, int slab_w, int n_cols,
                                                               float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int quarter = (n_slabs + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * quarter;
    const int s1 = min(s0 + quarter, n_slabs);
    float s = 0.f;
    if (col < n_cols) {
#pragma unroll 16
        for (int k = s0 + wave; k < s1; k += 4) s += slabs[(int64_t)k * slab_w + col];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < n_cols)
        out[blockIdx.y * slab_w + col] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

struct AdamArgs {
    float* params;
    float* m;
    float* v;
    const float* stage;      // [n_stage][slab_w] partial sums (or NULL when grad_in is given)
    int n_stage, slab_w, n_params;
    const float* grad_in;    // externally reduced gradient (data-parallel path) or NULL
    float* grad_out;         // final (unclipped) gradient, always written
    float max_grad_norm, lr_step, bc2_sqrt, beta1, beta2, eps;
    float vf_coef, ent_coef;
    int sig_off, act;        // for the entropy term
    float* losses;           // [4] loss, clip, vf, ent (or NULL)
    int apply;               // 0: only produce grad_out (+ loss parts)
};

__global__ __launch_bounds__(1024) void ppo_adam_kernel(AdamArgs a) {
    __shared__ float red[16];
    __shared__ float scale_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sq = 0.f;
    for (int p = tid; p < a.n_params; p += 1024) {
        float g;
        if (a.grad_in) g = a.grad_in[p];
        else {
            g = 0.f;
            for (int q = 0; q < a.n_stage; ++q) g += a.stage[q * a.slab_w + p];
        }
        a.grad_out[p] = g;
        sq += g * g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float norm = sqrtf(t);
        float scale = 1.f;
        if (a.max_grad_norm > 0.f) {
            // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
            scale = fminf(a.max_grad_norm / (norm + 1e-6f), 1.f);
        }
        scale_s = scale;
        if (a.losses) {
            float clip = 0.f, vf = 0.f;
            if (a.stage) {
                for (int q = 0; q < a.n_stage; ++q) {
                    clip += a.stage[q * a.slab_w + a.n_params];
                    vf += a.stage[q * a.slab_w + a.n_params + 1];
                }
            }
            float ent = 0.f;  // Normal.entropy() = 0.5 + 0.5 log(2 pi) + log(sigma), summed over actions
            for (int k = 0; k < a.act; ++k)
                ent += 0.5f + 0.5f * 1.8378770664093453f + logf(expf(a.params[a.sig_off + k]));
            a.losses[0] = clip + a.vf_coef * vf - a.ent_coef * ent;   // ppo.py:211
            a.losses[1] = clip;
            a.losses[2] = vf;
            a.losses[3] = ent;
        }
    }
    __syncthreads();
    if (!a.apply) return;
    const float scale = scale_s;
    for (int p = tid; p < a.n_params; p += 1024) {
        const float g = a.grad_out[p] * scale;
        float m = a.m[p], v = a.v[p];
        m = m + (g - m) * (1.f - a.beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        v = v * a.beta2 + (1.f - a.beta2) * g * g;         // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.params[p] = a.params[p] + (-a.lr_step * m) / denom;   // addcdiv_(m, denom, -step_size)
        a.m[p] = m;
        a.v[p] = v;
    }
}

// per-minibatch advantage mean / unbiased std for all steps of one update()
__global__ __launch_bounds__(1024) void ppo_adv_stats_kernel(const float* __restrict__ adv,
                                                             const int64_t* __restrict__ perm,
                                                             const int64_t* __restrict__ mb_offset,
                                                             float* __restrict__ out) {
    __shared__ double r1[16], r2[16];
    const int64_t lo = mb_offset[blockIdx.x], hi = mb_offset[blockIdx.x + 1];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t k = lo + threadIdx.x; k < hi; k += 1024) {
        const double v = (double)adv[perm ? perm[k] : k];
        s1 += v;
        s2 += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < 16; ++w) { t1 += r1[w]; t2 += r2[w]; }
        const double n = (double)(hi - lo);
        const double mean = t1 / n;
        const double var = (t2 - n * mean * mean) / (n - 1.0);   // torch.std(): unbiased
        out[2 * blockIdx.x] = (float)mean;
        out[2 * blockIdx.x + 1] = (float)sqrt(var > 0.0 ? var : 0.0);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
template <int KS1>
size_t step_lds_bytes(const Dims& d) {
    const int acc_pad = (d.p_actor + N_EXTRA + 3) & ~3;
    return sizeof(float) * (size_t)(Lds<KS1>::ACC + acc_pad + STEP_WAVES * 2 * TILE_SIZE);
}

template <int KS1>
size_t infer_lds_bytes() { return sizeof(float) * (size_t)Lds<KS1>::INFER_END; }

inline int ks1_for(int obs) { return (obs + 2) / 2; }  // ceil((obs + 1) / 2)

#define TS_KS1_DISPATCH(ks, CALL)                                                      \
    switch (ks) {                                                                      \
        case 1: { constexpr int K = 1; CALL; } break;                                  \
        case 2: { constexpr int K = 2; CALL; } break;                                  \
        case 3: { constexpr int K = 3; CALL; } break;                                  \
        case 4: { constexpr int K = 4; CALL; } break;                                  \
        case 6: { constexpr int K = 6; CALL; } break;                                  \
        case 9: { constexpr int K = 9; CALL; } break;                                  \
        case 12: { constexpr int K = 12; CALL; } break;                                \
        case 14: { constexpr int K = 14; CALL; } break;                                \
        case 16: { constexpr int K = 16; CALL; } break;                                \
        default: return ts::fail(TS_ERR_UNSUPPORTED, "obs_dim %d not supported by the fused MLP kernels", obs_dim); \
    }

inline int supported_ks(int ks) {
    switch (ks) { case 1: case 2: case 3: case 4: case 6: case 9: case 12: case 14: case 16: return ks; }
    // round up to the next instantiated width (extra k-steps multiply zero weights)
    const int avail[] = {1, 2, 3, 4, 6, 9, 12, 14, 16};
    for (int a : avail) if (a >= ks) return a;
    return -1;
}

struct WsLayout {
    size_t slabs, stage, grad, advstats, total;
};

inline WsLayout ws_layout(int n_wg, int slab_w, int n_params, int64_t n_steps) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WsLayout w;
    w.slabs = 0;
    w.stage = al(w.slabs + sizeof(float) * (size_t)n_wg * slab_w);
    w.grad = al(w.stage + sizeof(float) * 4 * (size_t)slab_w);
    w.advstats = al(w.grad + sizeof(float) * (size_t)n_params);
    w.total = al(w.advstats + sizeof(float) * 2 * (size_t)(n_steps > 0 ? n_steps : 1));
    return w;
}

int check_dims(int64_t obs_dim, int64_t act_dim) {
    TS_REQUIRE(obs_dim >= 1 && act_dim >= 1, TS_ERR_INVALID_ARG, "obs_dim / act_dim must be >= 1");
    TS_REQUIRE(act_dim <= ACT_PAD, TS_ERR_UNSUPPORTED, "act_dim %lld > %d not supported",
               (long long)act_dim, ACT_PAD);
    TS_REQUIRE(obs_dim <= 31, TS_ERR_UNSUPPORTED, "obs_dim %lld > 31 not supported by the fused MLP kernels",
               (long long)obs_dim);
    return TS_OK;
}

int n_compute_units() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// launches forward/backward of one minibatch into the slabs, returns number of workgroups
template <int KS1>
int launch_step(const StepArgs& g, const Dims& d, int n_wg, hipStream_t s) {
    const size_t lds = step_lds_bytes<KS1>(d);
    static bool attr_done = false;
    if (!attr_done) {
        TS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_step_kernel<KS1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((ppo_step_kernel<KS1>), dim3(n_wg), dim3(STEP_THREADS), lds, s, g, d);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

inline int step_grid(int64_t n_rows) {
    const int64_t tiles = (n_rows + 31) / 32;
    int64_t wg = (tiles + STEP_WAVES - 1) / STEP_WAVES;
    const int cus = n_compute_units();
    if (wg > cus) wg = cus;
    if (wg < 1) wg = 1;
    return (int)wg;
}

inline void fill_hparams(StepArgs& g, const ts_ppo_hparams* hp) {
    g.eps_clip = (float)hp->eps_clip;
    g.dual_clip = (float)(hp->dual_clip > 0.0 ? hp->dual_clip : 0.0);
    g.vf_coef = (float)hp->vf_coef;
    g.ent_coef = (float)hp->ent_coef;
    g.value_clip = hp->value_clip;
    g.adv_norm = hp->adv_norm;
}

inline AdamArgs adam_args(float* params, float* m, float* v, int64_t step, const Dims& d,
                          const ts_ppo_hparams* hp) {
    AdamArgs a{};
    a.params = params; a.m = m; a.v = v;
    a.n_params = d.p_total;
    a.max_grad_norm = (float)(hp->max_grad_norm > 0.0 ? hp->max_grad_norm : 0.0);
    const double bc1 = 1.0 - pow(hp->beta1, (double)step);
    const double bc2 = 1.0 - pow(hp->beta2, (double)step);
    a.lr_step = (float)(hp->lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.beta1 = (float)hp->beta1; a.beta2 = (float)hp->beta2; a.eps = (float)hp->adam_eps;
    a.vf_coef = (float)hp->vf_coef; a.ent_coef = (float)hp->ent_coef;
    a.sig_off = d.a_sig; a.act = d.act;
    return a;
}

}  // namespace

extern "C" {

int64_t ts_ppo_param_count(int64_t obs_dim, int64_t act_dim) {
    if (obs_dim < 1 || act_dim < 1) return -1;
    return make_dims((int)obs_dim, (int)act_dim).p_total;
}

int ts_ppo_infer(const float* params, int64_t obs_dim, int64_t act_dim, const float* obs,
                 const float* act, int64_t n, float* v_out, float* logp_out, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_ppo_infer: negative n");
    if (n == 0 || (!v_out && !logp_out)) return TS_OK;
    TS_REQUIRE(params && obs, TS_ERR_INVALID_ARG, "ts_ppo_infer: NULL params / obs");
    TS_REQUIRE(!logp_out || act, TS_ERR_INVALID_ARG, "ts_ppo_infer: logp_out needs act");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    hipStream_t s = ts::as_stream(stream);
    const int64_t tiles = (n + 31) / 32;
    int64_t wg = (tiles + 7) / 8;
    const int64_t cap = (int64_t)n_compute_units() * 2;
    if (wg > cap) wg = cap;
    TS_KS1_DISPATCH(ks, {
        hipLaunchKernelGGL((ppo_infer_kernel<K>), dim3((unsigned)wg), dim3(512), infer_lds_bytes<K>(), s,
                           params, d, obs, act, n, v_out, logp_out);
    });
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_ppo_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v,
                  int64_t adam_step0, int64_t obs_dim, int64_t act_dim, const float* obs,
                  const float* act, const float* adv, const float* returns,
                  const float* logp_old, const float* v_s, int64_t n, const int64_t* perm,
                  const int64_t* h_mb_offset, int64_t n_steps, const ts_ppo_hparams* hp,
                  float* losses_out, float* grads_out, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n >= 0 && n_steps >= 0 && adam_step0 >= 0, TS_ERR_INVALID_ARG, "ts_ppo_update: negative size");
    if (n_steps == 0) return TS_OK;
    TS_REQUIRE(ws && params && adam_m && adam_v && obs && act && adv && returns && logp_old && v_s &&
                   h_mb_offset && hp,
               TS_ERR_INVALID_ARG, "ts_ppo_update: NULL argument");
    int64_t max_rows = 0;
    for (int64_t k = 0; k < n_steps; ++k) {
        const int64_t rows = h_mb_offset[k + 1] - h_mb_offset[k];
        TS_REQUIRE(rows >= 1, TS_ERR_SHAPE, "ts_ppo_update: minibatch %lld is empty", (long long)k);
        TS_REQUIRE(perm || h_mb_offset[k + 1] <= n, TS_ERR_SHAPE, "ts_ppo_update: minibatch beyond n");
        if (rows > max_rows) max_rows = rows;
    }
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    const int slab_w = (d.p_total + N_EXTRA + 3) & ~3;
    const int max_wg = step_grid(max_rows);
    const WsLayout wl = ws_layout(max_wg, slab_w, d.p_total, n_steps);
    // device copy of the minibatch offsets lives behind the stats
    const size_t off_bytes = sizeof(int64_t) * (size_t)(n_steps + 1);
    rc = ts::ws_reserve(ws, wl.total + off_bytes + 256);
    if (rc != TS_OK) return rc;
    char* base = reinterpret_cast<char*>(ws->base);
    float* slabs = reinterpret_cast<float*>(base + wl.slabs);
    float* stage = reinterpret_cast<float*>(base + wl.stage);
    float* grad = grads_out ? grads_out : reinterpret_cast<float*>(base + wl.grad);
    float* advstats = reinterpret_cast<float*>(base + wl.advstats);
    int64_t* d_off = reinterpret_cast<int64_t*>(base + wl.total);
    hipStream_t s = ts::as_stream(stream);

    if (hp->adv_norm) {
        TS_HIP_CHECK(hipMemcpyAsync(d_off, h_mb_offset, off_bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ppo_adv_stats_kernel, dim3((unsigned)n_steps), dim3(1024), 0, s, adv, perm,
                           d_off, advstats);
        TS_LAUNCH_CHECK();
    }
    for (int64_t k = 0; k < n_steps; ++k) {
        StepArgs g{};
        g.params = params; g.obs = obs; g.act = act; g.adv = adv; g.ret = returns;
        g.logp_old = logp_old; g.v_old = v_s;
        g.rows = perm ? perm + h_mb_offset[k] : nullptr;
        g.n_rows = h_mb_offset[k + 1] - h_mb_offset[k];
        if (!perm) {  // identity rows: shift the base pointers instead
            const int64_t o = h_mb_offset[k];
            g.obs = obs + o * obs_dim; g.act = act + o * act_dim; g.adv = adv + o; g.ret = returns + o;
            g.logp_old = logp_old + o; g.v_old = v_s + o;
        }
        g.inv_batch = 1.0f / (float)g.n_rows;
        g.adv_stats = hp->adv_norm ? advstats + 2 * k : nullptr;
        fill_hparams(g, hp);
        g.slabs = slabs; g.slab_w = slab_w;
        const int n_wg = step_grid(g.n_rows);
        TS_KS1_DISPATCH(ks, { rc = launch_step<K>(g, d, n_wg, s); });
        if (rc != TS_OK) return rc;
        const int n_stage = n_wg >= 16 ? 4 : 1;
        hipLaunchKernelGGL(ppo_reduce_slabs_kernel, dim3((slab_w + 63) / 64, n_stage), dim3(256), 0, s,
                           slabs, n_wg, slab_w, d.p_total + N_EXTRA, stage);
        AdamArgs a = adam_args(params, adam_m, adam_v, adam_step0 + k + 1, d, hp);
        a.stage = stage; a.n_stage = n_stage; a.slab_w = slab_w;
        a.grad_in = nullptr; a.grad_out = grad;
        a.losses = losses_out ? losses_out + 4 * k : nullptr;
        a.apply = 1;
        hipLaunchKernelGGL(ppo_adam_kernel, dim3(1), dim3(1024), 0, s, a);
        TS_LAUNCH_CHECK();
    }
    return TS_OK;
}

int ts_ppo_grad(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim,
                const float* obs, const float* act, const float* adv, const float* returns,
                const float* logp_old, const float* v_s, int64_t n, const int64_t* perm_rows,
                int64_t n_rows, int64_t global_batch, const float* adv_stats,
                const ts_ppo_hparams* hp, float* grad_out, float* loss_parts_out,
                ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n_rows >= 1 && global_batch >= n_rows, TS_ERR_SHAPE, "ts_ppo_grad: bad batch sizes");
    TS_REQUIRE(ws && params && obs && act && adv && returns && logp_old && v_s && hp && grad_out,
               TS_ERR_INVALID_ARG, "ts_ppo_grad: NULL argument");
    TS_REQUIRE(perm_rows || n_rows <= n, TS_ERR_SHAPE, "ts_ppo_grad: n_rows > n");
    TS_REQUIRE(!hp->adv_norm || adv_stats, TS_ERR_INVALID_ARG, "ts_ppo_grad: adv_norm needs adv_stats");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    const int slab_w = (d.p_total + N_EXTRA + 3) & ~3;
    const int n_wg = step_grid(n_rows);
    const WsLayout wl = ws_layout(n_wg, slab_w, d.p_total, 1);
    rc = ts::ws_reserve(ws, wl.total);
    if (rc != TS_OK) return rc;
    char* base = reinterpret_cast<char*>(ws->base);
    float* slabs = reinterpret_cast<float*>(base + wl.slabs);
    float* stage = reinterpret_cast<float*>(base + wl.stage);
    hipStream_t s = ts::as_stream(stream);
    StepArgs g{};
    g.params = params; g.obs = obs; g.act = act; g.adv = adv; g.ret = returns;
    g.logp_old = logp_old; g.v_old = v_s; g.rows = perm_rows; g.n_rows = n_rows;
    g.inv_batch = 1.0f / (float)global_batch;
    g.adv_stats = hp->adv_norm ? adv_stats : nullptr;
    fill_hparams(g, hp);
    g.slabs = slabs; g.slab_w = slab_w;
    TS_KS1_DISPATCH(ks, { rc = launch_step<K>(g, d, n_wg, s); });
    if (rc != TS_OK) return rc;
    const int n_stage = n_wg >= 16 ? 4 : 1;
    hipLaunchKernelGGL(ppo_reduce_slabs_kernel, dim3((slab_w + 63) / 64, n_stage), dim3(256), 0, s, slabs,
                       n_wg, slab_w, d.p_total + N_EXTRA, stage);
    AdamArgs a = adam_args(nullptr, nullptr, nullptr, 1, d, hp);
    a.params = const_cast<float*>(params);
    a.stage = stage; a.n_stage = n_stage; a.slab_w = slab_w;
    a.grad_in = nullptr; a.grad_out = grad_out; a.losses = loss_parts_out; a.apply = 0;
    a.max_grad_norm = 0.f;
    hipLaunchKernelGGL(ppo_adam_kernel, dim3(1), dim3(1024), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_ppo_apply(float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                 int64_t act_dim, const float* grad, float* grad_scratch, const ts_ppo_hparams* hp,
                 ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(params && adam_m && adam_v && grad && grad_scratch && hp, TS_ERR_INVALID_ARG,
               "ts_ppo_apply: NULL argument");
    TS_REQUIRE(adam_step >= 1, TS_ERR_INVALID_ARG, "ts_ppo_apply: adam_step counts from 1");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    AdamArgs a = adam_args(params, adam_m, adam_v, adam_step, d, hp);
    a.stage = nullptr; a.n_stage = 0; a.slab_w = 0;
    a.grad_in = grad; a.grad_out = grad_scratch; a.losses = nullptr; a.apply = 1;
    hipLaunchKernelGGL(ppo_adam_kernel, dim3(1), dim3(1024), 0, ts::as_stream(stream), a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
