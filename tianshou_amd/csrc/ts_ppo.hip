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
#include <cstdlib>

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
constexpr int STEP_THREADS = 256;            // 4 waves (one per SIMD); two workgroups per CU
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

// LDS carve (floats).  NN = number of nets resident at once: the inference kernel keeps both,
// the step kernel stages one net at a time so that two 256-thread workgroups fit one CU (<= 80 KB).
template <int KS1, int NN>
struct Lds {
    static constexpr int W2 = 0;                                  // [NN][64*68]
    static constexpr int W1 = W2 + NN * W2_SIZE;                  // [NN][2*KS1][64]  (k-major)
    static constexpr int W1_NET = 2 * KS1 * HID;
    static constexpr int B2 = W1 + NN * W1_NET;                   // [NN][64]
    static constexpr int WH = B2 + NN * HID;                      // [NN][h][t][r][8]
    static constexpr int WH_NET = 2 * 2 * 16 * ACT_PAD;
    // SMALL: [0..7] bmu, [8..15] 1/(2 sigma^2), [16..23] log(sigma), [24] bv
    static constexpr int SMALL = WH + NN * WH_NET;
    static constexpr int END = SMALL + 32;                        // step kernel appends R + scratch
};

// ---------------------------------------------------------------------------------------------
// weight staging: global flat params -> LDS images
// Staging is written as "issue every global load of a batch, then store": the loads of a batch
// are independent, so one memory round trip covers the whole batch (a plain load->store loop makes
// hipcc wait for each load).
template <int NT, int COUNT, typename SrcFn, typename DstFn>
__device__ __forceinline__ void stage_batch(const float* __restrict__ params, float* lds, SrcFn src, DstFn dst) {
    constexpr int PER = (COUNT + NT - 1) / NT;
    float v[PER];
    int ok[PER];
    // unconditional loads from a clamped index + select: a branch around a load makes hipcc wait
    // for every load separately (s_waitcnt vmcnt(0) per element)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = threadIdx.x + NT * k;
        const int sidx = src(idx < COUNT ? idx : COUNT - 1);
        ok[k] = (idx < COUNT) & (sidx >= 0);
        v[k] = params[sidx >= 0 ? sidx : 0];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = ok[k] ? v[k] : 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = threadIdx.x + NT * k;
        if (idx < COUNT) lds[dst(idx)] = v[k];
    }
}

// One batch for everything a net needs (W2, W1 image, b2, head image, small block): every global
// load is in flight before the first LDS store.  The caller must __syncthreads() and then call
// finish_small() (sigma -> 1/(2 sigma^2), log sigma) followed by another __syncthreads().
template <int KS1, int NN, int NT>
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ params,
                                              const Dims& d, int first_net) {
    using L = Lds<KS1, NN>;
    constexpr int N_W2 = HID * HID, N_W1 = 2 * KS1 * HID, N_B2 = HID, N_WH = L::WH_NET;
    constexpr int PER_NET = N_W2 + N_W1 + N_B2 + N_WH;
    constexpr int COUNT = NN * PER_NET + 32;
    const int obs = d.obs, act = d.act;
    auto src = [&](int idx) -> int {
        if (idx >= NN * PER_NET) {                       // small block (raw values)
            const int i = idx - NN * PER_NET;
            if (i < 8) return i < act ? d.a_bmu + i : -1;
            if (i < 24) return (i & 7) < act ? d.a_sig + (i & 7) : -1;
            return i == 24 ? d.c_bv : -1;
        }
        const int slot = idx / PER_NET;
        int i = idx - slot * PER_NET;
        const int net = first_net + slot;
        if (i < N_W2) return (net ? d.c_w2 : d.a_w2) + i;
        i -= N_W2;
        if (i < N_W1) {
            const int k = i >> 6, row = i & 63;
            const int w1 = net ? d.c_w1 : d.a_w1, b1 = net ? d.c_b1 : d.a_b1;
            return k < obs ? w1 + row * obs + k : (k == obs ? b1 + row : -1);
        }
        i -= N_W1;
        if (i < N_B2) return (net ? d.c_b2 : d.a_b2) + i;
        i -= N_B2;
        const int a = i & 7, r = (i >> 3) & 15, t = (i >> 7) & 1, h = (i >> 8) & 1;
        const int f = 32 * t + featF(r, h);
        if (net == 0) return a < act ? d.a_wmu + a * HID + f : -1;
        return a == 0 ? d.c_wv + f : -1;
    };
    auto dst = [&](int idx) -> int {
        if (idx >= NN * PER_NET) return L::SMALL + (idx - NN * PER_NET);
        const int slot = idx / PER_NET;
        int i = idx - slot * PER_NET;
        if (i < N_W2) return L::W2 + slot * W2_SIZE + (i >> 6) * W2_PITCH + (i & 63);
        i -= N_W2;
        if (i < N_W1) return L::W1 + slot * L::W1_NET + i;
        i -= N_W1;
        if (i < N_B2) return L::B2 + slot * HID + i;
        i -= N_B2;
        return L::WH + slot * L::WH_NET + i;
    };
    stage_batch<NT, COUNT>(params, lds, src, dst);
}

// sigma = exp(sigma_param) (continuous.py:238); Normal.log_prob uses var = sigma^2 and log(sigma)
template <int KS1, int NN>
__device__ __forceinline__ void finish_small(float* lds) {
    using L = Lds<KS1, NN>;
    const int i = threadIdx.x;
    if (i >= 8 && i < 24) {
        const float sigma = expf(lds[L::SMALL + i]);
        lds[L::SMALL + i] = (i < 16) ? 1.f / (2.f * (sigma * sigma)) : logf(sigma);
    }
}

// Straight float4 copy of a ready-made LDS image (kept current by ppo_adam_kernel): replaces the per-element
// index arithmetic of stage_weights in the step kernel's two prologues.
template <int KS1, int NT>
__device__ __forceinline__ void stage_image(float* lds, const float* __restrict__ img, int tid = threadIdx.x) {
    constexpr int N4 = Lds<KS1, 1>::END / 4, PER = (N4 + NT - 1) / NT;
    static_assert(Lds<KS1, 1>::END % 4 == 0, "image size");
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + NT * k;
        v[k] = reinterpret_cast<const f32x4*>(img)[q < N4 ? q : N4 - 1];      // unconditional clamped loads
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + NT * k;
        if (q < N4) reinterpret_cast<f32x4*>(lds)[q] = v[k];
    }
}

// Builds both images (blockIdx.x = net) and the inverse table inv[param] = net * END + slot that lets the Adam
// kernel refresh the image in place.  sigma_param maps to two derived slots and is handled there explicitly.
template <int KS1>
__global__ __launch_bounds__(256) void ppo_build_image_kernel(const float* __restrict__ params, Dims d,
                                                              float* __restrict__ image, int* __restrict__ inv) {
    using L = Lds<KS1, 1>;
    const int net = blockIdx.x;
    float* img = image + net * L::END;
    for (int i = threadIdx.x; i < L::END; i += 256) img[i] = 0.f;
    __syncthreads();
    stage_weights<KS1, 1, 256>(img, params, d, net);
    __syncthreads();
    finish_small<KS1, 1>(img);
    if (!inv) return;
    // inverse table: the same (src, dst) enumeration as stage_weights
    constexpr int N_W2 = HID * HID, N_W1 = 2 * KS1 * HID, N_B2 = HID, N_WH = L::WH_NET;
    constexpr int PER_NET = N_W2 + N_W1 + N_B2 + N_WH;
    for (int idx = threadIdx.x; idx < PER_NET + 32; idx += 256) {
        int src = -1, dst = -1;
        if (idx >= PER_NET) {
            const int i = idx - PER_NET;
            if (net == 0 && i < 8 && i < d.act) { src = d.a_bmu + i; dst = L::SMALL + i; }
            if (net == 1 && i == 24) { src = d.c_bv; dst = L::SMALL + 24; }
        } else {
            int i = idx;
            if (i < N_W2) { src = (net ? d.c_w2 : d.a_w2) + i; dst = L::W2 + (i >> 6) * W2_PITCH + (i & 63); }
            else if ((i -= N_W2) < N_W1) {
                const int k = i >> 6, row = i & 63;
                if (k < d.obs) src = (net ? d.c_w1 : d.a_w1) + row * d.obs + k;
                else if (k == d.obs) src = (net ? d.c_b1 : d.a_b1) + row;
                dst = L::W1 + i;
            } else if ((i -= N_W1) < N_B2) { src = (net ? d.c_b2 : d.a_b2) + i; dst = L::B2 + i; }
            else {
                i -= N_B2;
                const int a = i & 7, r = (i >> 3) & 15, t = (i >> 7) & 1, h = (i >> 8) & 1;
                const int f = 32 * t + featF(r, h);
                if (net == 0) { if (a < d.act) src = d.a_wmu + a * HID + f; }
                else if (a == 0) src = d.c_wv + f;
                dst = L::WH + i;
            }
        }
        if (src >= 0) inv[src] = net * L::END + dst;
    }
}

// tanh(x) = 1 - 2 / (e^{2x} + 1): one v_exp_f32, one v_rcp_f32, three plain VALU.  Absolute error <= 1.2e-7 for x >= 0
// and <= 2.4e-7 for x < 0 (2 / (e + 1) is rounded in [1, 2) there); saturates to +-1 (e -> inf / 0).  The activations enter
// the next layer as sums of O(1) terms, so the absolute error is what matters.  History: round 1's odd polynomial below
// |x| = 0.5 (2 ulp RELATIVE accuracy) cost 11 more instructions per value; rounds 2-3 evaluated the formula on |x| and
// copied the sign back (6 instructions, symmetric error) -- dropping that pair shortens the step kernel by 0.8 us
// (profiles/r04_fused_tail_and_tanh_ab.txt) and holds every parity bar of tests/test_gpu_ppo.py / test_gpu_hooks.py.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// The same on the 16 accumulator registers of an MFMA tile, two values per VALU instruction where the ISA allows it
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on register pairs; the two transcendentals stay scalar): the fp32 MFMA shares
// the vector lanes with the VALU (no co-execution), so every VALU instruction saved is matrix issue time returned.
// Bit-identical to fast_tanh per element.
#ifndef TS_PK
#define TS_PK 2          // experiments (profiles/r04_packed_valu_ab.txt): 0 scalar everywhere, 1 packed tanh, 2 + packed tanh', 3 + packed head-gradient dots
#endif
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void tanh16(f32x16& a) {
#if TS_PK < 1
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fast_tanh(a[r]);
    return;
#endif
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 x = {a[r], a[r + 1]};
        const f32x2 y = x * 2.885390081777927f;
        f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
        e = e + 1.f;
        const f32x2 q = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
        const f32x2 t = 1.f - 2.f * q;
        a[r] = t[0];
        a[r + 1] = t[1];
    }
}

// v[r] = v[r] * (1 - h[r]^2) on register pairs (tanh'), bit-identical to the scalar expression
__device__ __forceinline__ void dtanh16(f32x16& v, const f32x16& hh) {
#if TS_PK < 2
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] * (1.f - hh[r] * hh[r]);
    return;
#endif
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 hv = {hh[r], hh[r + 1]};
        const f32x2 g = {v[r], v[r + 1]};
        const f32x2 o = g * (1.f - hv * hv);
        v[r] = o[0];
        v[r + 1] = o[1];
    }
}

// ---------------------------------------------------------------------------------------------
// forward trunk of one net for one 32-sample tile: x -> h1 -> h2 (transposed, in registers)
template <int KS1, int NN>
__device__ __forceinline__ void trunk_forward(const float* lds, int net, const float (&x)[KS1],
                                              int i, int h, f32x16 (&h1)[2], f32x16 (&h2)[2]) {
    using L = Lds<KS1, NN>;
    const float* w1 = lds + L::W1 + net * L::W1_NET;
    const float* w2 = lds + L::W2 + net * W2_SIZE;
    const float* b2 = lds + L::B2 + net * HID;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS1; ++s) acc = mfma32(w1[(KS1 * h + s) * HID + 32 * t + i], x[s], acc);
        tanh16(acc);
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
            __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every operand load
        }
        tanh16(acc);
        h2[t2] = acc;
    }
}

// head on the VALU: out[a] = sum_f h2[f] * WH[a][f] over the lane's 32 features, halves combined
template <int KS1, int NN, int NA>
__device__ __forceinline__ void head_forward(const float* lds, int net, int h, const f32x16 (&h2)[2],
                                             float (&out)[NA]) {
    using L = Lds<KS1, NN>;
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
            if (NA > 1 && (r & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bounds the operand-load hoisting
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

// Normal(mu, sigma).log_prob(act).sum(-1)  (torch.distributions.Normal.log_prob, Independent):
//   -(x - mu)^2 / (2 var) - log(sigma) - log(sqrt(2 pi));  sm = SMALL block (1/(2var) at 8.., log sigma at 16..)
__device__ __forceinline__ float gaussian_logp(const float (&mu)[ACT_PAD], const float (&act)[ACT_PAD],
                                               const float* sm, int act_dim) {
    float lp = 0.f;
#pragma unroll
    for (int a = 0; a < ACT_PAD; ++a) {
        if (a < act_dim) {
            const float dlt = act[a] - mu[a];
            lp += -(dlt * dlt) * sm[8 + a] - sm[16 + a] - LOG_SQRT_2PI;
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
                                                        float* __restrict__ logp_out,
                                                        float* __restrict__ mu_out, float mu_bound) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<KS1, 2>;
    stage_weights<KS1, 2, 512>(lds, params, d, 0);
    __syncthreads();
    finish_small<KS1, 2>(lds);
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
            trunk_forward<KS1, 2>(lds, 1, x, i, h, h1, h2);
            float v[1];
            head_forward<KS1, 2, 1>(lds, 1, h, h2, v);
            if (valid && h == 0) v_out[srow] = v[0] + lds[L::SMALL + 24];
        }
        if (logp_out || mu_out) {
            trunk_forward<KS1, 2>(lds, 0, x, i, h, h1, h2);
            float mu[ACT_PAD], a[ACT_PAD];
            head_forward<KS1, 2, ACT_PAD>(lds, 0, h, h2, mu);
#pragma unroll
            for (int k = 0; k < ACT_PAD; ++k) {
                mu[k] += lds[L::SMALL + k];
                if (mu_bound > 0.f) mu[k] = mu_bound * fast_tanh(mu[k]);          // continuous.py:230-231
                a[k] = (logp_out && k < d.act) ? act[row * d.act + k] : 0.f;
            }
            if (mu_out && valid && h == 0) {
#pragma unroll
                for (int k = 0; k < ACT_PAD; ++k)
                    if (k < d.act) mu_out[srow * d.act + k] = mu[k];
            }
            if (logp_out) {
                const float lp = gaussian_logp(mu, a, lds + L::SMALL, d.act);
                if (valid && h == 0) logp_out[srow] = lp;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct StepArgs {
    const float* params;
    const float* rec;         // packed per-sample records [n][rec_w]: obs | act | adv ret logp_old v_old | pad
    int rec_w;                // floats per record, multiple of 4 (16-byte rows)
    const int64_t* rows;      // minibatch row ids (perm slice) or NULL = identity
    int64_t n_rows;           // rows in this minibatch (local)
    float inv_batch;          // 1 / global minibatch size
    const float* adv_stats;   // {mean, std} of this minibatch (device) or NULL
    float eps_clip, dual_clip, vf_coef, ent_coef;
    int value_clip, adv_norm, a2c;
    int nets;                 // 0 / 3: both networks (ppo_step2_kernel); 1: actor only, 2: critic only (ppo_step1_kernel)
    float mu_bound;           // > 0: mu = mu_bound * tanh(head) (ContinuousActorProbabilistic(unbounded=False), continuous.py:230-231)
    const float* image;       // [2][Lds<KS1,1>::END] ready-made LDS images of the two nets (ppo_build_image_kernel)
    float* slabs;             // [gridDim.x][slab_w]
    int slab_w;
    long long* dbg;           // optional phase timestamps of workgroup 0 / wave 0 (diagnostics)
};

// Phase timestamps are a build option (-DTS_PHASE_MARKS: scripts/gpu_step_phases.py builds its own copy of the library
// with it); the shipped kernels carry no s_memtime and the shipped library no diagnostic entry point.
#ifdef TS_PHASE_MARKS
#define TS_MARK(g, k)                                                                  \
    do {                                                                               \
        if ((g).dbg && blockIdx.x == 0 && threadIdx.x == 0) (g).dbg[(k)] = (long long)__builtin_readcyclecounter(); \
        if ((g).dbg && threadIdx.x == 0 && ((k) == 0 || (k) == 17))                    \
            (g).dbg[64 + 2 * blockIdx.x + ((k) == 17)] = (long long)__builtin_amdgcn_s_memrealtime();   /* 100 MHz, chip-wide */ \
        if ((g).dbg && threadIdx.x == 0 && (k) == 0)      /* where the workgroup runs: HW_ID (CU / SH / SE) and XCC_ID */ \
            (g).dbg[64 + 1024 + blockIdx.x] = ((long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) | \
                                              (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);          \
    } while (0)
#else
#define TS_MARK(g, k) do { } while (0)
#endif

// transposed tile write: lane (j, h) register r -> T[F(r,h)][j]
__device__ __forceinline__ void tile_write(float* tile, const f32x16& v, int j, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[featF(r, h) * TILE_PITCH + j] = v[r];
}

__device__ __forceinline__ void wave_lds_sync() {
    // wave-private LDS hand-off between lanes of one wave: LDS executes a wave's instructions in
    // order; this only stops the compiler from moving accesses across the hand-off.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// per-sample inputs of one 32-sample tile.  The wave fetches the 32 packed records with 16-byte
// loads (consecutive lanes -> consecutive 16-byte pieces of a record: every 128-byte line that is
// touched is used almost completely), parks them in its LDS tile area and each lane then picks the
// columns its MFMA operands need.
template <int KS1>
struct TileIn {
    float x[KS1];
    float act[ACT_PAD];
    float adv, logp_old, ret, v_old;
    float w;          // inv_batch for valid rows, 0 for padding rows
};

// float4 loads per lane for one tile: obs <= 2 KS1 - 1, act <= 8, 4 aux -> parts <= (2 KS1 + 14) / 4
template <int KS1>
struct RecFetch {
    static constexpr int N = (32 * ((2 * KS1 + 14) / 4) + 63) / 64;
    f32x4 v[N];
    float w;
};

// row id of the lane's sample (lanes i and i+32 hold the same sample); issued first so that its
// latency overlaps the weight staging
struct RowId { int64_t id; float w; };

__device__ __forceinline__ RowId row_fetch(const StepArgs& g, int64_t tile, int lane) {
    const int i = lane & 31;
    const int64_t srow = tile * 32 + i;
    const bool valid = srow < g.n_rows;
    const int64_t pos = valid ? srow : g.n_rows - 1;
    RowId r;
    r.id = g.rows ? g.rows[pos] : pos;
    r.w = valid ? g.inv_batch : 0.f;
    return r;
}

// issue the global loads of a tile's records (no waiting here)
template <int KS1>
__device__ __forceinline__ RecFetch<KS1> rec_fetch(const StepArgs& g, const RowId& row, int lane) {
    RecFetch<KS1> f;
    constexpr int REC_FETCH = RecFetch<KS1>::N;
    f.w = row.w;
    const int parts = g.rec_w >> 2;
    const int total = 32 * parts;
    const int lo = (int)(row.id & 0xffffffffLL), hi = (int)(row.id >> 32);
#pragma unroll
    for (int k = 0; k < REC_FETCH; ++k) {
        int q = lane + 64 * k;
        q = q < total ? q : total - 1;           // unconditional (clamped) load, see stage_batch
        const int rec = q / parts, part = q - rec * parts;
        const int64_t rid = ((int64_t)__shfl(hi, rec, 64) << 32) | (uint32_t)__shfl(lo, rec, 64);
        f.v[k] = *reinterpret_cast<const f32x4*>(g.rec + rid * g.rec_w + part * 4);
    }
    return f;
}

// park the fetched records in the wave's LDS area (rb, >= 32 * rec_w floats) and extract the
// lane's operands
template <int KS1>
__device__ __forceinline__ TileIn<KS1> rec_commit(const RecFetch<KS1>& f, const StepArgs& g, const Dims& d,
                                                   float* rb, int lane) {
    constexpr int REC_FETCH = RecFetch<KS1>::N;
    const int i = lane & 31, h = lane >> 5;
    const int parts = g.rec_w >> 2;
    const int total = 32 * parts;
#pragma unroll
    for (int k = 0; k < REC_FETCH; ++k) {
        const int q = lane + 64 * k;
        if (q < total) *reinterpret_cast<f32x4*>(rb + q * 4) = f.v[k];   // rec * rec_w + part * 4 == q * 4
    }
    wave_lds_sync();
    TileIn<KS1> t;
    const float* r = rb + i * g.rec_w;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int k = KS1 * h + s;
        float v = 0.f;
        if (k < d.obs) v = r[k];
        else if (k == d.obs) v = 1.f;
        t.x[s] = v;
    }
#pragma unroll
    for (int k = 0; k < ACT_PAD; ++k) t.act[k] = (k < d.act) ? r[d.obs + k] : 0.f;
    const float* aux = r + d.obs + d.act;
    t.adv = aux[0];
    t.ret = aux[1];
    t.logp_old = aux[2];
    t.v_old = aux[3];
    t.w = f.w;
    wave_lds_sync();
    return t;
}

constexpr int P128 = 132;                      // pitch (floats) of the 128-sample tiles
constexpr int T2_FLOATS = 128 * P128;          // phase A: dZ2^T rows 0..63 | H1^T rows 64..127
constexpr int T2_XROW = 64;                    // phase B: dZ1^T rows 0..63 | X^T rows 64..95 | misc slots behind row 96
constexpr int MISC_ROWS = 9;                   // rows 0..7 head-weight gradients (lane = feature), row 8 = `misc`
constexpr int MISC_SLOT = MISC_ROWS * 64;
constexpr int T2_MISC = 96 * P128;
static_assert(T2_MISC + STEP_WAVES * MISC_SLOT <= T2_FLOATS, "misc slots must fit behind the phase-B tiles");

struct Slab2 {     // column offsets (floats) inside a workgroup slab, see slab2_layout()
    int w1[2], w2[2], b2[2], head[2], hb[2], sig, loss, width;
};

__host__ __device__ inline Slab2 slab2_layout(int act, int kp) {
    Slab2 L;
    int o = 0;
    for (int n = 0; n < 2; ++n) {
        const int n_head = n ? 1 : act;
        L.w1[n] = o; o += HID * kp;
        L.w2[n] = o; o += HID * HID;
        L.b2[n] = o; o += HID;
        L.head[n] = o; o += n_head * HID;
        L.hb[n] = o; o += n_head;
        if (n == 0) { L.sig = o; o += act; }
    }
    L.loss = o; o += N_EXTRA;
    L.width = (o + 3) & ~3;
    return L;
}

// slab column -> flat parameter index (>= p_total: the two loss sums; -1: padding)
__device__ __forceinline__ int slab_col_to_param(int col, const Dims& d, int kp) {
    const Slab2 L = slab2_layout(d.act, kp);
    if (col >= L.loss) return col < L.loss + N_EXTRA ? d.p_total + (col - L.loss) : -1;
    const int n = col >= L.w1[1];
    const int c = col - L.w1[n];
    if (c < HID * kp) {
        const int f = c / kp, k = c - f * kp;
        const int w1 = n ? d.c_w1 : d.a_w1, b1 = n ? d.c_b1 : d.a_b1;
        return k < d.obs ? w1 + f * d.obs + k : (k == d.obs ? b1 + f : -1);
    }
    return (n ? d.c_w2 : d.a_w2) + (c - HID * kp);       // W2, b2, head W, head b(, sigma) are contiguous in both
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// transposed write of one 32-feature block into a 128-sample tile: lane (j, h) register r -> T[row0 + F(r,h)][col]
__device__ __forceinline__ void tile128_write(float* T, int row0, const f32x16& v, int col, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(row0 + featF(r, h)) * P128 + col] = v[r];
}

// Slab stores are write-through (relaxed agent-scope atomic store = `global_store_dword ... sc1`): the 23 MB of slabs
// would otherwise sit dirty in the eight L2s until the kernel boundary and be written back there, in front of the
// reduction kernel that reads them (MI355X_MICROARCH.md, "boundary" / "publish-large" rows).
// The slab pointer is laundered through an empty asm in net_wgrad (hoisting, see there), which also hides from the compiler
// that it points to global memory: without the explicit address-space cast below the stores are FLAT stores, which count on
// the LDS counter (lgkmcnt) as well and stall the operand reads of the next gradient tile behind them.
using gfloat_ptr = __attribute__((address_space(1))) float*;
#ifndef TS_SLAB_PLAIN_STORES
__device__ __forceinline__ void slab_st(float* p, float v) {
    __hip_atomic_store((gfloat_ptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ __forceinline__ void slab_st(float* p, float v) { *(gfloat_ptr)p = v; }
#endif

__device__ __forceinline__ void store_acc(float* p, float v, bool first) {
    if (!first) v += *(gfloat_ptr)p;
    slab_st(p, v);
}

// One 32x32 output tile contracted over the workgroup's 128 samples: C[m][n] = sum_s A[rowA + m][s] B[rowB + n][s].
// `rsum` (optional) returns, in lanes 0..31, the row sums of A (sum over the 128 samples of row rowA + lane).
template <bool WANT_RSUM>
__device__ __forceinline__ f32x16 tile128_mma(const float* T, int rowA, int rowB, int i, int h, float* rsum) {
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* pa = T + (rowA + i) * P128 + 16 * h;
    const float* pb = T + (rowB + i) * P128 + 16 * h;
    float rs = 0.f;
    f32x4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = ld4(pa + 4 * q); b[q] = ld4(pb + 4 * q); }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        f32x4 an[4], bn[4];
        if (ch < 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { an[q] = ld4(pa + 32 * (ch + 1) + 4 * q); bn[q] = ld4(pb + 32 * (ch + 1) + 4 * q); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c = mfma32(a[q][e], b[q][e], c);
            if (WANT_RSUM) rs += (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
        }
        if (ch < 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = an[q]; b[q] = bn[q]; }
        }
    }
    if (WANT_RSUM) *rsum = rs + __shfl_xor(rs, 32, 64);
    return c;
}

// forward, loss and backward of one net for the wave's 32 samples, up to dZ2 (returned in h2), dZ1, the head-weight
// gradients gw (lane = feature) and the `misc` row (lanes 0..7 head-bias gradients, 8..15 sigma gradients, 16 loss sum)
template <int KS1, bool ACTOR, bool BOUNDED = false>
__device__ __forceinline__ void net_fwd_bwd(float* lds, float* scratch, const StepArgs& g, const Dims& d,
                                            const TileIn<KS1>& in, int lane_in, f32x16 (&h1)[2], f32x16 (&h2)[2],
                                            f32x16 (&dz1)[2], float (&gw)[ACTOR ? ACT_PAD : 1], float& misc) {
    using L = Lds<KS1, 1>;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));          // see net_tile: keeps lane-dependent addresses out of the prologue
    [[maybe_unused]] constexpr int MK = ACTOR ? 2 : 10;
    constexpr int NA = ACTOR ? ACT_PAD : 1;
    const int i = lane & 31, h = lane >> 5;
    trunk_forward<KS1, 1>(lds, 0, in.x, i, h, h1, h2);
    TS_MARK(g, MK + 0);

    float dout[NA];
    const float w = in.w;
    const float* sm = lds + L::SMALL;
    float* Qt = scratch;                     // [17][TILE_PITCH]: per-sample quantities, transposed, for the row sums
    if constexpr (ACTOR) {
        float mu[ACT_PAD];
        head_forward<KS1, 1, ACT_PAD>(lds, 0, h, h2, mu);
        const f32x4 b0 = ld4(sm), b1 = ld4(sm + 4), v0 = ld4(sm + 8), v1 = ld4(sm + 12), s0 = ld4(sm + 16), s1 = ld4(sm + 20);
        float dlt[ACT_PAD], inv_var[ACT_PAD];
        [[maybe_unused]] float bgrad[ACT_PAD];              // bounded actor: 1 - tanh^2 of the head output
        // ContinuousActorProbabilistic(unbounded=False): a template parameter, not a kernel argument -- the uniform branch
        // alone cost the unbounded (headline) kernel 1.4 us per launch through register allocation (profiles/r06_bounded_branch_ab.txt)
        constexpr bool bounded = BOUNDED;
        float logp = 0.f;
        int n_act = d.act;
        asm volatile("" : "+s"(n_act));      // otherwise the 8 per-action constants are hoisted into (spilled) VGPRs
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            // padding actions (k >= act): zero weights, bias 0, sigma_param 0 -> mu = 0, act = 0, log sigma = 0: the
            // term is an exact -0; only the constant has to be switched off
            const float bm = k < 4 ? b0[k & 3] : b1[k & 3], iv = k < 4 ? v0[k & 3] : v1[k & 3], ls = k < 4 ? s0[k & 3] : s1[k & 3];
            float m = mu[k] + bm;
            if (bounded) {                                   // uniform (kernel argument): no cost on the unbounded path
                const float t = fast_tanh(m);
                m = g.mu_bound * t;
                bgrad[k] = 1.f - t * t;
            }
            dlt[k] = in.act[k] - m;
            inv_var[k] = 2.f * iv;
            logp += -(dlt[k] * dlt[k]) * iv - ls - (k < n_act ? LOG_SQRT_2PI : 0.f);
        }
        float mean = 0.f, den = 1.f;
        if (g.adv_norm) { mean = g.adv_stats[0]; den = g.adv_stats[1] + 1e-8f; }      // uniform scalar loads
        const float A = (in.adv - mean) / den;                                       // ppo.py:184-186 (x - 0) / 1 is exact
        const bool a2c = g.a2c != 0;
        // A2C (a2c.py:266-267): term = -logp * adv, d term / d logp = -adv  == "ratio" fixed at 1
        const float ratio = a2c ? 1.f : expf(logp - in.logp_old);                    // :187
        const float surr1 = ratio * A;
        const float lo = 1.f - g.eps_clip, hi = 1.f + g.eps_clip;
        const float surr2 = fminf(fmaxf(ratio, lo), hi) * A;                         // :190
        const float clip1 = fminf(surr1, surr2);
        // torch.min backward: the smaller branch takes the gradient; inside the clip range both branches are the
        // same value and together pass the full gradient.
        float basek = (surr1 <= surr2) ? A : 0.f;
        const float dA = g.dual_clip * A;
        const bool dual = (g.dual_clip > 0.f) && (A < 0.f);                          // :191-194
        float term = dual ? -fmaxf(clip1, dA) : -clip1;                              // :196
        basek = (dual && !(clip1 >= dA)) ? 0.f : basek;
        term = a2c ? -logp * A : term;
        basek = a2c ? A : basek;
        const float dlogp = -basek * ratio * w;
        const float ent_w = g.ent_coef * w;
        float dsig[ACT_PAD];
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            dout[k] = dlogp * dlt[k] * inv_var[k];                                   // 0 for padding actions (dlt = 0)
            if (bounded) dout[k] = (dout[k] * g.mu_bound) * bgrad[k];               // MulBackward, then TanhBackward
            const float ds = dlogp * (dlt[k] * dlt[k] * inv_var[k] - 1.f) - ent_w;   // entropy: d/ds = 1
            dsig[k] = k < n_act ? ds : 0.f;
        }
        // rows 0..7 dout, 8..15 dsig, 16 loss term: half 0 writes the dout rows and the loss row, half 1 the dsig rows
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) Qt[(8 * h + k) * TILE_PITCH + i] = h ? dsig[k] : dout[k];
        if (h == 0) Qt[16 * TILE_PITCH + i] = term * w;
    } else {
        float v[1];
        head_forward<KS1, 1, 1>(lds, 0, h, h2, v);
        const float value = v[0] + sm[24];
        const float ret = in.ret;
        const float vf1 = (ret - value) * (ret - value);
        // ppo.py:199-206.  torch.max backward: the larger branch takes the gradient, ties split it; clamp passes the
        // gradient inside [-eps, eps].  Inside the range v_clip = vo + (v - vo) differs from v by rounding, so either
        // branch may win there.
        const float vo = in.v_old;
        const float dvo = value - vo;
        const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
        const float vf2 = (ret - vclip) * (ret - vclip);
        const float g1 = -2.f * (ret - value);
        const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
        const float dv_clip = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        const bool vc = g.value_clip != 0;
        const float term = vc ? fmaxf(vf1, vf2) : vf1;                               // :208
        const float dv = vc ? dv_clip : g1;
        dout[0] = dv * g.vf_coef * w;
        Qt[(16 * h) * TILE_PITCH + i] = h ? term * w : dout[0];                      // row 0: dout, row 16: loss term
    }
    wave_lds_sync();
    {
        const int row = lane < 16 ? lane : 16;
        const float* rp = Qt + row * TILE_PITCH;
        f32x4 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = ld4(rp + 4 * k);
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += (q[k][0] + q[k][1]) + (q[k][2] + q[k][3]);
        const bool mine = ACTOR ? (lane <= 16) : (lane == 0 || lane == 16);
        misc = mine ? sacc : 0.f;
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 1);
    __builtin_amdgcn_s_setprio(ACTOR ? 2 : 0);          // see ppo_step2_kernel: priority falls with progress

    // ---- head weight gradient: gw[a][f] = sum_s dout[s][a] * H2[s][f]  (lane = feature f).
    // dout of sample s is broadcast through the 4 padding columns of row s of the two tiles.
    float* SA = scratch;
    float* SB = scratch + TILE_SIZE;
    tile_write(SA, h2[0], i, h);
    tile_write(SB, h2[1], i, h);
    if (lane < 32) {
        if constexpr (ACTOR) {
            const f32x4 d0 = {dout[0], dout[1], dout[2], dout[3]};
            const f32x4 d1 = {dout[4], dout[5], dout[6], dout[7]};
            *reinterpret_cast<f32x4*>(SA + lane * TILE_PITCH + 32) = d0;
            *reinterpret_cast<f32x4*>(SB + lane * TILE_PITCH + 32) = d1;
        } else {
            SA[lane * TILE_PITCH + 32] = dout[0];
        }
    }
    wave_lds_sync();
    {
        const float* rowp = (lane < 32 ? SA : SB) + (lane & 31) * TILE_PITCH;
#pragma unroll
        for (int a = 0; a < NA; ++a) gw[a] = 0.f;
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const f32x4 hv = ld4(rowp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int smp = 4 * q + e;
                if constexpr (ACTOR) {
                    const f32x4 d0 = ld4(SA + smp * TILE_PITCH + 32);        // uniform address
                    const f32x4 d1 = ld4(SB + smp * TILE_PITCH + 32);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        gw[a] += d0[a] * hv[e];
                        gw[a + 4] += d1[a] * hv[e];
                    }
                } else {
                    gw[0] += SA[smp * TILE_PITCH + 32] * hv[e];
                }
            }
        }
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 2);

    // ---- dZ2 = (dout . Whead) * (1 - h2^2)   (in place in h2).  Actor: the eight products of a value run as four packed
    // multiply-adds over action pairs (even / odd partial sums, added at the end); tanh' on register pairs.
    {
        const float* wh = lds + L::WH + h * (2 * 16 * ACT_PAD);
        [[maybe_unused]] f32x2 d01, d23, d45, d67;
        if constexpr (ACTOR) {
            d01 = f32x2{dout[0], dout[1]}; d23 = f32x2{dout[2], dout[3]};
            d45 = f32x2{dout[4], dout[5]}; d67 = f32x2{dout[6], dout[7]};
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 dhv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* p = wh + (t * 16 + r) * ACT_PAD;
                if constexpr (ACTOR) {
                    const f32x4 w0 = ld4(p);
                    const f32x4 w1 = ld4(p + 4);
#if TS_PK < 3
                    dhv[r] = dout[0] * w0[0] + dout[1] * w0[1] + dout[2] * w0[2] + dout[3] * w0[3] +
                             dout[4] * w1[0] + dout[5] * w1[1] + dout[6] * w1[2] + dout[7] * w1[3];
                    continue;
#endif
                    f32x2 sacc = d01 * f32x2{w0[0], w0[1]};
                    sacc = d23 * f32x2{w0[2], w0[3]} + sacc;
                    sacc = d45 * f32x2{w1[0], w1[1]} + sacc;
                    sacc = d67 * f32x2{w1[2], w1[3]} + sacc;
                    dhv[r] = sacc[0] + sacc[1];
                } else {
                    dhv[r] = dout[0] * p[0];
                }
                if (ACTOR && (r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            dtanh16(dhv, h2[t]);
            h2[t] = dhv;
        }
    }

    // ---- dH1^T = W2^T . dZ2^T, then dZ1 = dH1 * (1 - h1^2)
    {
        const float* w2 = lds + L::W2;
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
            f32x16 accd = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accd = mfma32(w2[(32 * t + featF(r, h)) * W2_PITCH + 32 * t1 + i], h2[t][r], accd);
                __builtin_amdgcn_sched_barrier(0);
            }
            dtanh16(accd, h1[t1]);
            dz1[t1] = accd;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 3);
}

// the weight-gradient half of one net: shared 128-sample tiles, whole output tiles per wave, direct slab stores
template <int KS1, bool ACTOR>
__device__ __forceinline__ void net_wgrad(float* lds, const StepArgs& g, const Dims& d, const TileIn<KS1>& in,
                                          const f32x16 (&h1)[2], const f32x16 (&dz2)[2], const f32x16 (&dz1)[2],
                                          const float (&gw)[ACTOR ? ACT_PAD : 1], float misc, int wave, int lane_in,
                                          float* slab, const Slab2& SL, bool first) {
    [[maybe_unused]] constexpr int MK = ACTOR ? 2 : 10;
    constexpr int NA = ACTOR ? ACT_PAD : 1;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int KP = 2 * KS1;
    // lane, wave and the slab pointer are made opaque here: otherwise LLVM hoists the (loop-invariant) store and tile
    // addresses of the fully unrolled body out of the tile loop into the kernel prologue, where they are spilled
    int lane = lane_in;
    asm volatile("" : "+v"(lane), "+v"(wave), "+s"(slab));
    const int i = lane & 31, h = lane >> 5;
    const int col = 32 * wave + i;
    float* T = lds;
    __syncthreads();                       // B1: every wave is done with the weight image and its scratch area
    tile128_write(T, 0, dz2[0], col, h);
    tile128_write(T, 32, dz2[1], col, h);
    tile128_write(T, 64, h1[0], col, h);
    tile128_write(T, 96, h1[1], col, h);
    __syncthreads();                       // B2
    TS_MARK(g, MK + 4);
    {
        // dW2[f2][f1] = sum_s dZ2[s][f2] H1[s][f1]: wave (tM, tN) owns rows 32 tM.., columns 32 tN..; db2 = row sums
        const int tM = wave >> 1, tN = wave & 1;
        float rs;
        const f32x16 c = tile128_mma<true>(T, 32 * tM, 64 + 32 * tN, i, h, &rs);
        float* p = slab + SL.w2[net] + (32 * tM + 4 * h) * HID + 32 * tN + i;
        if (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_st(p + ((r & 3) + 8 * (r >> 2)) * HID, c[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_st(p + ((r & 3) + 8 * (r >> 2)) * HID, ((gfloat_ptr)p)[((r & 3) + 8 * (r >> 2)) * HID] + c[r]);
        }
        if (tN == 0 && h == 0) store_acc(slab + SL.b2[net] + 32 * tM + i, rs, first);
    }
    __syncthreads();                       // B3: all reads of the phase-A tiles are done
    TS_MARK(g, MK + 5);
    tile128_write(T, 0, dz1[0], col, h);
    tile128_write(T, 32, dz1[1], col, h);
#pragma unroll
    for (int s = 0; s < KS1; ++s) T[(T2_XROW + KS1 * h + s) * P128 + col] = in.x[s];
    {
        float* M = T + T2_MISC + wave * MISC_SLOT;
#pragma unroll
        for (int a = 0; a < NA; ++a) M[a * 64 + lane] = gw[a];
        M[8 * 64 + lane] = misc;
    }
    __syncthreads();                       // B4
    // the two dW1 tiles go to one wave pair, the small rows to the other; the pairs swap roles between the nets so
    // that every wave issues the same number of MFMAs per launch
    const bool w1_wave = ACTOR ? (wave < 2) : (wave >= 2);
    if (w1_wave) {
        // dW1aug[f1][k] = sum_s dZ1[s][f1] Xaug[s][k]   (k == obs is the bias column; rows of X^T beyond 2 KS1 hold
        // stale but finite H1 values whose output columns are not stored)
        const int tM = wave & 1;
        const f32x16 c = tile128_mma<false>(T, 32 * tM, T2_XROW, i, h, nullptr);
        if (i < KP) {
            float* p = slab + SL.w1[net] + (32 * tM + 4 * h) * KP + i;
            if (first) {
#pragma unroll
                for (int r = 0; r < 16; ++r) slab_st(p + ((r & 3) + 8 * (r >> 2)) * KP, c[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) slab_st(p + ((r & 3) + 8 * (r >> 2)) * KP, ((gfloat_ptr)p)[((r & 3) + 8 * (r >> 2)) * KP] + c[r]);
            }
        }
    } else {
        const int part = wave & 1;
        const float* M = T + T2_MISC;
        const int n_head = ACTOR ? d.act : 1;
        if (ACTOR) {
            // part 0: rows 0..3 and the misc row, part 1: rows 4..7
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * part + q;
                float v = 0.f;
#pragma unroll
                for (int sl = 0; sl < STEP_WAVES; ++sl) v += M[sl * MISC_SLOT + row * 64 + lane];
                if (row < n_head) store_acc(slab + SL.head[net] + row * HID + lane, v, first);
            }
        } else if (part == 1) {
            float v = 0.f;
#pragma unroll
            for (int sl = 0; sl < STEP_WAVES; ++sl) v += M[sl * MISC_SLOT + lane];
            store_acc(slab + SL.head[net] + lane, v, first);
        }
        if (part == 0) {
            float v = 0.f;
#pragma unroll
            for (int sl = 0; sl < STEP_WAVES; ++sl) v += M[sl * MISC_SLOT + 8 * 64 + lane];
            if (lane < 8) { if (lane < n_head) store_acc(slab + SL.hb[net] + lane, v, first); }
            else if (lane < 16) { if (ACTOR && lane - 8 < d.act) store_acc(slab + SL.sig + lane - 8, v, first); }
            else if (lane == 16) store_acc(slab + SL.loss + net, v, first);
        }
    }
    TS_MARK(g, MK + 6);
}

template <int KS1, bool BOUNDED = false>
__global__ __launch_bounds__(STEP_THREADS, 2) void ppo_step2_kernel(StepArgs g, Dims d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<KS1, 1>;
    static_assert(L::END + STEP_WAVES * 2 * TILE_SIZE <= T2_FLOATS, "weight image + wave scratch must fit under the tiles");
    const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
    const Slab2 SL = slab2_layout(d.act, 2 * KS1);

    const int64_t n_tiles = (g.n_rows + 31) / 32;
    const int64_t per_iter = (int64_t)gridDim.x * STEP_WAVES;
    const int64_t n_iter = (n_tiles + per_iter - 1) / per_iter;   // same for every wave: barriers inside
    const int64_t tile0 = (int64_t)blockIdx.x * STEP_WAVES + wave0;
    float* slab = g.slabs + (int64_t)blockIdx.x * g.slab_w;

    TS_MARK(g, 0);
    for (int64_t it = 0; it < n_iter; ++it) {
        // Two workgroups share a CU and the SIMDs issue the OLDEST wave first: left alone, the older workgroup of every CU
        // finishes in 41-45 us and the younger in 53-55 us, alone on its SIMDs for the last 10 us (per-workgroup clocks,
        // scripts/gpu_step_phases.py).  The waves therefore lower their own priority as they advance (3: actor trunk and
        // loss, 2: actor gradients, 1: critic trunk and loss, 0: critic gradients) -- whichever workgroup is behind wins
        // the arbitration: 45-47 / 52-54 us, launch 54.7 -> 53.0 us.  (It does not close the gap: the workgroup that is
        // behind is mostly waiting on its own latencies, and the other one fills those slots whatever its priority.  A
        // short last level -- 3 / 2 / 1 and 0 only for the final weight gradients -- measured 53.7 us.)
        __builtin_amdgcn_s_setprio(3);
        // per-iteration opaque copies: nothing lane- / wave-dependent is worth hoisting out of a loop that usually runs
        // once, and what LLVM hoists here ends up spilled in the prologue
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane), "+v"(wave));
        float* scratch = lds + L::END + wave * (2 * TILE_SIZE);
        // order: row ids (one dependent load) -> record gathers -> weight image (independent of both)
        const RowId row0 = row_fetch(g, it * per_iter + tile0, lane);
        const RecFetch<KS1> f = rec_fetch<KS1>(g, row0, lane);
        if (it > 0) __syncthreads();          // the previous iteration's phase-B readers are done
        stage_image<KS1, STEP_THREADS>(lds, g.image, 64 * wave + lane);
        const TileIn<KS1> in = rec_commit<KS1>(f, g, d, scratch, lane);
        __syncthreads();
        TS_MARK(g, 1);
        const bool first = it == 0;
        f32x16 h1[2], h2[2], dz1[2];
        float misc;
        {
            float gw[ACT_PAD];
            net_fwd_bwd<KS1, true, BOUNDED>(lds, scratch, g, d, in, lane, h1, h2, dz1, gw, misc);
            net_wgrad<KS1, true>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        }
        __syncthreads();                      // phase-B readers of the actor are done: the image region is free
        TS_MARK(g, 18);
        {
            int tid = 64 * wave + lane;
            asm volatile("" : "+v"(tid));       // see above: keeps the six load addresses out of the prologue
            stage_image<KS1, STEP_THREADS>(lds, g.image + L::END, tid);
        }
        __syncthreads();
        TS_MARK(g, 9);
        __builtin_amdgcn_s_setprio(1);
        {
            float gw[1];
            net_fwd_bwd<KS1, false>(lds, scratch, g, d, in, lane, h1, h2, dz1, gw, misc);
            net_wgrad<KS1, false>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        }
    }
    TS_MARK(g, 17);
}

// One of the two networks only (ts_ppo_hparams.nets = 1: actor, 2: critic): the same phases on the same operands as the
// corresponding half of ppo_step2_kernel; the other network's gradient columns and loss sum are written as zeros, so the
// slab reduction, the global norm and Adam see a zero gradient for it.  For callers whose other network is a stand-in:
// Reinforce's minibatch steps (A2C's actor loss with adv := returns; no critic exists) and the critic iterations of
// NPG / TRPO (A2C steps with a zero advantage; the actor takes no gradient there).
template <int KS1, bool BOUNDED = false>
__global__ __launch_bounds__(STEP_THREADS, 2) void ppo_step1_kernel(StepArgs g, Dims d, int net) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<KS1, 1>;
    const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
    const Slab2 SL = slab2_layout(d.act, 2 * KS1);

    const int64_t n_tiles = (g.n_rows + 31) / 32;
    const int64_t per_iter = (int64_t)gridDim.x * STEP_WAVES;
    const int64_t n_iter = (n_tiles + per_iter - 1) / per_iter;   // same for every wave: barriers inside
    const int64_t tile0 = (int64_t)blockIdx.x * STEP_WAVES + wave0;
    float* slab = g.slabs + (int64_t)blockIdx.x * g.slab_w;
    TS_MARK(g, 0);
    {   // the absent network's columns: [0, w1[1]) = actor incl. sigma, [w1[1], loss) = critic; its loss sum
        const int z0 = net == 0 ? SL.w1[1] : 0, z1 = net == 0 ? SL.loss : SL.w1[1];
        for (int c = z0 + (int)threadIdx.x; c < z1; c += STEP_THREADS) slab_st(slab + c, 0.f);
        if (threadIdx.x == 0) slab_st(slab + SL.loss + (1 - net), 0.f);
    }
    for (int64_t it = 0; it < n_iter; ++it) {
        __builtin_amdgcn_s_setprio(3);
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane), "+v"(wave));
        float* scratch = lds + L::END + wave * (2 * TILE_SIZE);
        const RowId row0 = row_fetch(g, it * per_iter + tile0, lane);
        const RecFetch<KS1> f = rec_fetch<KS1>(g, row0, lane);
        if (it > 0) __syncthreads();          // the previous iteration's phase-B readers are done
        stage_image<KS1, STEP_THREADS>(lds, g.image + (net ? L::END : 0), 64 * wave + lane);
        const TileIn<KS1> in = rec_commit<KS1>(f, g, d, scratch, lane);
        __syncthreads();
        TS_MARK(g, 1);
        const bool first = it == 0;
        f32x16 h1[2], h2[2], dz1[2];
        float misc;
        if (net == 0) {
            float gw[ACT_PAD];
            net_fwd_bwd<KS1, true, BOUNDED>(lds, scratch, g, d, in, lane, h1, h2, dz1, gw, misc);
            net_wgrad<KS1, true>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        } else {
            float gw[1];
            net_fwd_bwd<KS1, false>(lds, scratch, g, d, in, lane, h1, h2, dz1, gw, misc);
            net_wgrad<KS1, false>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        }
    }
    TS_MARK(g, 17);     // (-DTS_PHASE_MARKS builds only; the phases in between are marked inside net_fwd_bwd / net_wgrad)
}

#include "ts_ppo_q.h"
#include "ts_npg_q.h"

// ---------------------------------------------------------------------------------------------
// slab reduction: grad[col] = sum over all workgroup slabs (fixed order), plus the block's
// partial sum of squares over the parameter columns (for the global gradient norm).
constexpr int RED_THREADS = 1024;

__global__ __launch_bounds__(RED_THREADS) void ppo_reduce_slabs_kernel(const float* __restrict__ slabs,
                                                                       int n_slabs, int slab_w, Dims d, int kp, int k1q,
                                                                       float* __restrict__ grad,
                                                                       float* __restrict__ sumsq_part,
                                                                       const float* __restrict__ params,
                                                                       float* __restrict__ losses,
                                                                       float* __restrict__ parts) {
    __shared__ float red[RED_THREADS / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n_params = d.p_total;
    float s = 0.f;
    if (col < slab_w) {
#pragma unroll 16
        for (int k = wave; k < n_slabs; k += RED_THREADS / 64) s += slabs[(int64_t)k * slab_w + col];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RED_THREADS / 64; ++k) t += red[k][lane];
        // slab column -> flat parameter index (second-generation slabs keep W1 | b1 as one augmented matrix)
        // (k1q > 0: the slabs come from ppo_stepq_kernel, transposed sections, see ts_ppo_q.h)
        const int pidx = col < slab_w ? (k1q > 0 ? q4::slab3_col_to_param(col, d, k1q) : slab_col_to_param(col, d, kp)) : -1;
        // data-parallel path (parts != NULL): grad holds only the n_params gradient columns, the two loss sums go
        // to parts[1] (clip) and parts[2] (vf); parts[3] = entropy below, parts[0] is composed by the caller
        if (pidx >= 0 && pidx < (parts ? n_params : n_params + N_EXTRA)) grad[pidx] = t;
        if (parts && pidx == n_params) parts[1] = t;
        if (parts && pidx == n_params + 1) parts[2] = t;
        float q = (pidx >= 0 && pidx < n_params) ? t * t : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        if (lane == 0) sumsq_part[blockIdx.x] = q;
        if (parts && blockIdx.x == 0 && lane == 0) parts[0] = 0.f;
        if ((losses || parts) && blockIdx.x == 0 && lane == 0) {
            // Normal.entropy() = 0.5 + 0.5 log(2 pi) + log(sigma), summed over actions; identical
            // for every sample, so its batch mean (ppo.py:210) is the value itself.  Computed here,
            // before the Adam kernel touches sigma_param.
            float ent = 0.f;
            for (int k = 0; k < d.act; ++k)
                ent += 0.5f + 0.5f * 1.8378770664093453f + logf(expf(params[d.a_sig + k]));
            if (losses) losses[3] = ent;
            if (parts) parts[3] = ent;
        }
    }
}

struct AdamArgs {
    float* params;
    float* m;
    float* v;
    const float* grad;        // [n_params (+2 loss sums)] reduced, unclipped
    const float* sumsq_part;  // [n_part] partial sums of squares, or NULL -> computed here (DP path)
    int n_part, n_params;
    float max_grad_norm, lr_step, bc2_sqrt, beta1, beta2, eps;
    float omb1, omb2;            // (1 - beta) rounded from double, as torch passes them
    float vf_coef, ent_coef;
    int opt_kind, centered;      // ts_ppo_hparams.optimizer / rms_centered (TS_OPT_*)
    float wd, lr, alpha, oma, momentum;   // weight decay (both optimizers); RMSprop: lr, alpha, 1 - alpha, momentum
    float* losses;            // [4] loss, clip, vf, ent (or NULL); grad[n_params], grad[n_params+1] hold clip / vf
    int apply;                // 0: only losses
    float* image;             // LDS images of the step kernel to refresh (or NULL)
    const int* inv;           // param -> image slot (-1: none)
    int sig_off, act, small0; // sigma_param range and the actor image's SMALL block
};

constexpr int ADAM_THREADS = 256;

__global__ __launch_bounds__(ADAM_THREADS) void ppo_adam_kernel(AdamArgs a) {
    __shared__ float red[ADAM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this thread's parameter: issue its (independent) loads before the norm's dependent chain, one memory round
    // trip instead of two
    const int p = blockIdx.x * ADAM_THREADS + tid;
    const bool mine = a.apply && p < a.n_params;
    const int pc = mine ? p : 0;
    float g_raw = a.grad[pc], m = a.m[pc], v = a.v[pc], par = a.params[pc];
    const int slot = (mine && a.image) ? a.inv[pc] : -1;
    // global gradient norm: every workgroup re-reduces the (few) partials in the same order
    float sq = 0.f;
    if (a.sumsq_part) {
        for (int k = tid; k < a.n_part; k += ADAM_THREADS) sq += a.sumsq_part[k];
    } else {
        for (int q = tid; q < a.n_params; q += ADAM_THREADS) { const float gq = a.grad[q]; sq += gq * gq; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < ADAM_THREADS / 64; ++k) total += red[k];
    const float norm = sqrtf(total);
    float scale = 1.f;
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    if (a.max_grad_norm > 0.f) scale = fminf(a.max_grad_norm / (norm + 1e-6f), 1.f);

    if (a.losses && blockIdx.x == 0 && tid == 0) {
        const float clip = a.grad[a.n_params], vf = a.grad[a.n_params + 1];
        const float ent = a.losses[3];                            // written by the reduce kernel
        a.losses[0] = clip + a.vf_coef * vf - a.ent_coef * ent;   // ppo.py:211
        a.losses[1] = clip;
        a.losses[2] = vf;
    }
    if (mine) {
        float gq = g_raw * scale;
        if (a.wd != 0.f) gq = gq + a.wd * par;             // grad.add(param, alpha=weight_decay), inside optimizer.step
        float np_;
        if (a.opt_kind == TS_OPT_RMSPROP) {
            // torch/optim/rmsprop.py _single_tensor_rmsprop (optim.py:113-140): v = square_avg, m = momentum buffer / grad_avg
            v = v * a.alpha + a.oma * gq * gq;             // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
            float avg;
            if (a.centered) {
                m = m + (gq - m) * a.oma;                  // grad_avg.lerp_(grad, 1 - alpha)
                avg = sqrtf(v + -1.f * m * m);             // square_avg.addcmul(grad_avg, grad_avg, value=-1).sqrt_()
            } else {
                avg = sqrtf(v);
            }
            avg = avg + a.eps;
            if (a.momentum > 0.f) {
                m = m * a.momentum + gq / avg;             // buf.mul_(momentum).addcdiv_(grad, avg)
                np_ = par + -a.lr * m;                     // param.add_(buf, alpha=-lr)
            } else {
                np_ = par + (-a.lr * gq) / avg;            // param.addcdiv_(grad, avg, value=-lr)
            }
        } else {
            m = m + (gq - m) * a.omb1;                     // exp_avg.lerp_(grad, 1 - beta1)
            v = v * a.beta2 + a.omb2 * gq * gq;            // mul_(beta2).addcmul_(g, g, 1 - beta2)
            const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
            np_ = par + (-a.lr_step * m) / denom;          // addcdiv_(m, denom, -step_size)
        }
        a.params[p] = np_;
        a.m[p] = m;
        a.v[p] = v;
        if (a.image) {
            if (slot >= 0) a.image[slot] = np_;
            const int k = p - a.sig_off;
            if (k >= 0 && k < a.act) {                        // finish_small: 1 / (2 sigma^2), log sigma
                const float sigma = expf(np_);
                a.image[a.small0 + 8 + k] = 1.f / (2.f * (sigma * sigma));
                a.image[a.small0 + 16 + k] = logf(sigma);
            }
        }
    }
}

// packs the batch into per-sample records [n][rec_w]: obs | act | adv ret logp_old v_old | 0-pad
__global__ __launch_bounds__(256) void ppo_pack_kernel(const float* __restrict__ obs,
                                                       const float* __restrict__ act,
                                                       const float* __restrict__ adv,
                                                       const float* __restrict__ ret,
                                                       const float* __restrict__ logp_old,
                                                       const float* __restrict__ v_old, int64_t n,
                                                       int obs_dim, int act_dim, int rec_w,
                                                       float* __restrict__ rec) {
    const int64_t total = n * rec_w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / rec_w;
        const int c = (int)(t - row * rec_w);
        float v = 0.f;
        if (c < obs_dim) v = obs[row * obs_dim + c];
        else if (c < obs_dim + act_dim) v = act[row * act_dim + (c - obs_dim)];
        else if (c == obs_dim + act_dim) v = adv[row];
        else if (c == obs_dim + act_dim + 1) v = ret[row];
        else if (c == obs_dim + act_dim + 2) v = logp_old[row];
        else if (c == obs_dim + act_dim + 3) v = v_old[row];
        rec[t] = v;
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
size_t infer_lds_bytes() { return sizeof(float) * (size_t)Lds<KS1, 2>::END; }

// floats per workgroup slab
inline int slab_width(const Dims& d, int ks) {
    return slab2_layout(d.act, 2 * ks).width;
}

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
        default: return ts::fail(TS_ERR_UNSUPPORTED, "obs_dim %d not supported by the fused MLP kernels", (int)obs_dim); \
    }

inline int supported_ks(int ks) {
    // round up to the next instantiated width (extra k-steps multiply zero weights)
    const int avail[] = {1, 2, 3, 4, 6, 9, 12, 14, 16};
    for (int a : avail) if (a >= ks) return a;
    return -1;
}

struct WsLayout {
    size_t slabs, grad, sumsq, advstats, total;
    int n_red_blocks;
};

inline WsLayout ws_layout(int n_wg, int slab_w, int64_t n_steps) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WsLayout w;
    w.n_red_blocks = (slab_w + 63) / 64;
    w.slabs = 0;
    w.grad = al(w.slabs + sizeof(float) * (size_t)n_wg * slab_w);
    w.sumsq = al(w.grad + sizeof(float) * (size_t)slab_w);
    w.advstats = al(w.sumsq + sizeof(float) * (size_t)w.n_red_blocks);
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

// launches forward/backward of one minibatch into the slabs
template <int KS1>
int launch_step(ts_workspace* ws, const StepArgs& g, const Dims& d, int n_wg, hipStream_t s) {
    const size_t lds = sizeof(float) * (size_t)T2_FLOATS;
    static ts::DynLds attr2, attr1, attr2b, attr1b;      // per device (ts_common.h)
    const bool bounded = g.mu_bound > 0.f;               // (its own instantiation: see net_fwd_bwd)
    if (g.nets == 1 || g.nets == 2) {            // one network only (ts_ppo_hparams.nets)
        if (bounded) {
            if (int rc = attr1b.allow(reinterpret_cast<const void*>(&ppo_step1_kernel<KS1, true>), lds)) return rc;
        } else {
            if (int rc = attr1.allow(reinterpret_cast<const void*>(&ppo_step1_kernel<KS1, false>), lds)) return rc;
        }
        ts::ProfScope prof(ws, TS_KIND_PPO_STEP, s);
        if (bounded) hipLaunchKernelGGL((ppo_step1_kernel<KS1, true>), dim3(n_wg), dim3(STEP_THREADS), lds, s, g, d, g.nets - 1);
        else hipLaunchKernelGGL((ppo_step1_kernel<KS1, false>), dim3(n_wg), dim3(STEP_THREADS), lds, s, g, d, g.nets - 1);
    } else {
        if (bounded) {
            if (int rc = attr2b.allow(reinterpret_cast<const void*>(&ppo_step2_kernel<KS1, true>), lds)) return rc;
        } else {
            if (int rc = attr2.allow(reinterpret_cast<const void*>(&ppo_step2_kernel<KS1, false>), lds)) return rc;
        }
        ts::ProfScope prof(ws, TS_KIND_PPO_STEP, s);
        if (bounded) hipLaunchKernelGGL((ppo_step2_kernel<KS1, true>), dim3(n_wg), dim3(STEP_THREADS), lds, s, g, d);
        else hipLaunchKernelGGL((ppo_step2_kernel<KS1, false>), dim3(n_wg), dim3(STEP_THREADS), lds, s, g, d);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

inline int step_grid(int64_t n_rows) {
    const int64_t tiles = (n_rows + 31) / 32;
    int64_t wg = (tiles + STEP_WAVES - 1) / STEP_WAVES;
    static const int per_cu = [] { const char* e = getenv("TS_PPO_WG_PER_CU"); return e ? atoi(e) : 2; }();
    const int cap = per_cu * n_compute_units();   // two 256-thread workgroups per CU
    if (wg > cap) wg = cap;
    if (wg < 1) wg = 1;
    return (int)wg;
}

// Which step kernel runs a minibatch of `n_rows` rows, and the slab geometry that goes with it.
//   variant 0: ppo_step2_kernel / ppo_step1_kernel (128-sample workgroups, LDS weight image, 2 workgroups per CU)
//   variant 1: ppo_stepq_kernel (ts_ppo_q.h: 32-sample tiles split by features over 4 waves, one network per
//              workgroup, persistent over tiles, 4 workgroups per CU); n_slabs = workgroup PAIRS
// TS_PPO_STEPQ=0 / 1 / 2 forces a variant (A/B runs); TS_PPO_STEPQ_PAIRS caps the pairs (slab count / tiles per workgroup).
struct StepPlan { int variant, grid, n_slabs, slab_w, k1s, big; };

inline StepPlan plan_step(const Dims& d, int ks, int64_t n_rows, int nets, bool bounded = false) {
    // (read per call, not cached: A/B scripts flip them inside one process)
    const char* e_force = getenv("TS_PPO_STEPQ");
    const char* e_pairs = getenv("TS_PPO_STEPQ_PAIRS");
    const int force = e_force ? atoi(e_force) : -1, pairs_cap = e_pairs ? atoi(e_pairs) : 0;
    StepPlan pl{};
    const int k1s = q4::k1s_for(d.obs);
    // Default: the feature-split kernel up to 3 tiles per workgroup (24,576 rows on 256 CUs), the 128-sample kernel above.
    // Measured on MI355X (profiles/r05_step_kernel_by_rows.txt): 8,192 rows 17.6 vs 32.3 us, 16,384 rows 23.5 vs 32.2 us,
    // 32,768 rows 35.2 vs 34.5 us, 65,536 rows 57.1 vs 52.6 us (two workgroups gather every record there).
    const int64_t tiles_all = (n_rows + 31) / 32;
    // (a bounded actor -- ts_ppo_hparams.max_action > 0 -- runs on the 128-sample kernel at every row count: the tanh bound on
    // mu and its derivative are built into ppo_step2_kernel / ppo_step1_kernel only)
    const bool q = !bounded && (force < 0 ? (k1s > 0 && tiles_all <= 3 * (int64_t)n_compute_units()) : (force != 0 && k1s > 0));
    if (!q) {
        pl.variant = 0;
        pl.grid = pl.n_slabs = step_grid(n_rows);
        pl.slab_w = slab_width(d, ks);
        return pl;
    }
    pl.variant = 1;
    pl.k1s = k1s;
    pl.slab_w = q4::slab3_layout(4 * k1s).width;
    const int64_t tiles = (n_rows + 31) / 32;
    // TS_PPO_STEPQ=1: the 128-register build, four workgroups per CU; 2 (default): the 256-register build, two per CU --
    // half of them per network
    pl.big = force != 1;
    int64_t pairs = (pl.big ? 1 : 2) * (int64_t)n_compute_units();
    // up to one tile per CU and network: ONE workgroup per CU with two tiles each beats two per CU with one (8,192 rows:
    // 15.9 vs 17.6 us, 4,096 rows: 12.7 vs 16.0 us at 64 pairs) -- prologue and epilogue are paid once per workgroup
    if (pl.big && tiles_all <= pairs) pairs = (pairs + 1) / 2;
    if (pairs_cap > 0) pairs = pairs_cap;
    if (pairs > tiles) pairs = tiles;
    if (pairs < 1) pairs = 1;
    pl.n_slabs = (int)pairs;
    pl.grid = (nets == 1 || nets == 2) ? (int)pairs : 2 * (int)pairs;
    return pl;
}

template <int K1S>
int launch_stepq(ts_workspace* ws, const StepArgs& g, const Dims& d, const StepPlan& pl, hipStream_t s) {
    const size_t lds = q4::stepq_lds_bytes(g.rec_w);
    static ts::DynLds attr_q, attr_q2;           // per device, the largest size granted so far (ts_common.h)
    if (int rc = attr_q.allow(reinterpret_cast<const void*>(&q4::ppo_stepq_kernel<K1S>), lds)) return rc;
    if (int rc = attr_q2.allow(reinterpret_cast<const void*>(&q4::ppo_stepq2_kernel<K1S>), lds)) return rc;
    ts::ProfScope prof(ws, TS_KIND_PPO_STEP, s);
    if (pl.big) hipLaunchKernelGGL((q4::ppo_stepq2_kernel<K1S>), dim3(pl.grid), dim3(q4::QT), lds, s, g, d, pl.n_slabs);
    else hipLaunchKernelGGL((q4::ppo_stepq_kernel<K1S>), dim3(pl.grid), dim3(q4::QT), lds, s, g, d, pl.n_slabs);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

inline void fill_hparams(StepArgs& g, const ts_ppo_hparams* hp) {
    g.eps_clip = (float)hp->eps_clip;
    g.dual_clip = (float)(hp->dual_clip > 0.0 ? hp->dual_clip : 0.0);
    g.vf_coef = (float)hp->vf_coef;
    g.ent_coef = (float)hp->ent_coef;
    g.a2c = hp->algo == 1;
    g.nets = hp->nets;
    g.mu_bound = (float)(hp->max_action > 0.0 ? hp->max_action : 0.0);
    g.value_clip = g.a2c ? 0 : hp->value_clip;
    g.adv_norm = g.a2c ? 0 : hp->adv_norm;
    if (g.a2c) g.dual_clip = 0.f;
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
    a.omb1 = (float)(1.0 - hp->beta1); a.omb2 = (float)(1.0 - hp->beta2);
    a.vf_coef = (float)hp->vf_coef;
    a.ent_coef = (float)hp->ent_coef;
    a.opt_kind = hp->optimizer; a.centered = hp->rms_centered;
    a.wd = (float)hp->weight_decay; a.lr = (float)hp->lr;
    a.alpha = (float)hp->rms_alpha; a.oma = (float)(1.0 - hp->rms_alpha); a.momentum = (float)hp->rms_momentum;
    return a;
}

struct ImageBuf { size_t img_bytes, inv_bytes; int img_end; };

inline ImageBuf image_buf(const Dims& d, int ks) {
    ImageBuf b;
    b.img_end = 4960 + 128 * ks;                           // Lds<ks, 1>::END
    b.img_bytes = (sizeof(float) * 2 * (size_t)b.img_end + 255) & ~(size_t)255;
    b.inv_bytes = (sizeof(int) * (size_t)d.p_total + 255) & ~(size_t)255;
    return b;
}

// LDS images of both nets + the param -> image-slot table (see ppo_build_image_kernel)
int build_image(hipStream_t s, const float* params, const Dims& d, int ks, float* image, int* inv) {
    const int64_t obs_dim = d.obs;
    if (inv) TS_HIP_CHECK(hipMemsetAsync(inv, 0xff, sizeof(int) * (size_t)d.p_total, s));   // -1: no image slot
    TS_KS1_DISPATCH(ks, {
        static_assert(Lds<K, 1>::END == 4960 + 128 * K, "image size formula");
        hipLaunchKernelGGL((ppo_build_image_kernel<K>), dim3(2), dim3(256), 0, s, params, d, image, inv);
    });
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// Persistent images for the data-parallel path (ts_ppo_grad / ts_ppo_apply): owned by the workspace, rebuilt when
// the parameter vector, the dimensions or the validity flag change, refreshed in place by ts_ppo_apply's Adam.
int dp_image(ts_workspace* ws, hipStream_t s, const float* params, const Dims& d, int ks, float** image, int** inv) {
    const ImageBuf ib = image_buf(d, ks);
    const size_t need = ib.img_bytes + ib.inv_bytes;
    const int key = (d.obs << 8) | d.act;
    if (ws->ppo_image_bytes < need) {
        TS_HIP_CHECK(hipSetDevice(ws->device));
        if (ws->ppo_image) { TS_HIP_CHECK(hipDeviceSynchronize()); TS_HIP_CHECK(hipFree(ws->ppo_image)); }
        TS_HIP_CHECK(hipMalloc(&ws->ppo_image, need));
        ws->ppo_image_bytes = need;
        ws->ppo_image_params = nullptr;
    }
    *image = static_cast<float*>(ws->ppo_image);
    *inv = reinterpret_cast<int*>(static_cast<char*>(ws->ppo_image) + ib.img_bytes);
    if (ws->ppo_image_params != params || ws->ppo_image_key != key) {
        if (int rc = build_image(s, params, d, ks, *image, *inv)) return rc;
        ws->ppo_image_params = params;
        ws->ppo_image_key = key;
    }
    return TS_OK;
}

// forward/backward + slab reduction of one minibatch: grad[0..P) unclipped gradient,
// grad[P], grad[P+1] clip / vf loss sums, sumsq partials, losses[3] = entropy (if losses)
int run_grad(ts_workspace* ws, StepArgs& g, const Dims& d, int ks, const StepPlan& pl, float* slabs, float* grad,
             float* sumsq, float* losses, hipStream_t s, float* parts = nullptr) {
    const int64_t obs_dim = d.obs;
    int rc = TS_OK;
    g.slabs = slabs; g.slab_w = pl.slab_w;
    if (pl.variant == 1) {
        switch (pl.k1s) {
            case 2: rc = launch_stepq<2>(ws, g, d, pl, s); break;
            case 3: rc = launch_stepq<3>(ws, g, d, pl, s); break;
            case 5: rc = launch_stepq<5>(ws, g, d, pl, s); break;
            case 8: rc = launch_stepq<8>(ws, g, d, pl, s); break;
            default: return ts::fail(TS_ERR_UNSUPPORTED, "obs_dim %d not supported by the feature-split step kernel", d.obs);
        }
    } else {
        TS_KS1_DISPATCH(ks, { rc = launch_step<K>(ws, g, d, pl.grid, s); });
    }
    if (rc != TS_OK) return rc;
    {
        ts::ProfScope prof(ws, TS_KIND_PPO_REDUCE, s);
        hipLaunchKernelGGL(ppo_reduce_slabs_kernel, dim3((pl.slab_w + 63) / 64), dim3(RED_THREADS), 0, s, slabs,
                           pl.n_slabs, pl.slab_w, d, 2 * ks, pl.variant == 1 ? 4 * pl.k1s : 0, grad, sumsq, g.params, losses,
                           parts);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace

// ---- the one-launch actor passes of ts_npg_q.h, for ts_npg.hip (same library, not part of the C ABI)
namespace ts {

bool npg_fused_supported(int64_t obs_dim, int64_t hidden, int64_t act_dim) {
    const char* e = getenv("TS_NPG_FVP");                       // 0: the per-layer GEMM path (A/B runs)
    if (e && atoi(e) == 0) return false;
    return hidden == HID && obs_dim >= 1 && obs_dim <= 32 && act_dim >= 1 && act_dim <= ACT_PAD;
}

// Workgroups of a gradient / Fisher-vector-product / critic launch: TWO per CU although three fit (<= 168 registers) -- the
// third of every SIMD's registers left over is where the OTHER chain's kernel runs when NPGEngine.update has the critic
// iterations on a second stream (profiles/r05_npg_two_streams_grid_sweep.txt: 768 / 768 1,103, 512 / 512 1,170 steps/s).
static int npg_grid(int64_t B, int per_cu, const char* env = "TS_NPG_FVP_WGS") {
    const char* e = getenv(env);
    const int64_t tiles = (B + 31) / 32;
    int64_t n = e && atoi(e) > 0 ? atoi(e) : per_cu * (int64_t)n_compute_units();
    if (n > tiles) n = tiles;
    return (int)(n < 1 ? 1 : n);
}
static int npg_eval_grid(int64_t B, int n_cand) {
    const int64_t tiles = (B + 31) / 32;
    int64_t n = (4 * (int64_t)n_compute_units() + n_cand - 1) / n_cand;
    if (n > tiles) n = tiles;
    return (int)(n < 1 ? 1 : n);
}

// floats of slab scratch for the GRAD / FVP launches on B rows; of partial sums for an EVAL launch
size_t npg_fused_slab_floats(int64_t obs_dim, int64_t B) {
    const int grid = std::max(npg_grid(B, 2), npg_grid(B, 2, "TS_NPG_CRITIC_WGS"));
    return (size_t)grid * (size_t)q4::actor_slab_width(q4::k1s_for((int)obs_dim));
}
size_t npg_fused_eval_floats(int64_t B, int n_cand) { return (size_t)n_cand * (size_t)npg_eval_grid(B, n_cand) * 2; }

#define TS_NPG_LAUNCH(KERNEL, MODE, K, GRID)                                                                        \
    case K: {                                                                                                       \
        static ts::DynLds attr;                                                                                     \
        const size_t lds = q4::actor_lds_bytes<MODE>(K);                                                            \
        if (int rc_a = attr.allow(reinterpret_cast<const void*>(&q4::KERNEL<K>), lds)) return rc_a;                \
        ts::ProfScope prof(ws, TS_KIND_PPO_STEP, s);                                                                \
        hipLaunchKernelGGL((q4::KERNEL<K>), GRID, dim3(q4::QT), lds, s, g);                                         \
    } break;
#define TS_NPG_DISPATCH(KERNEL, MODE, GRID)                                                                         \
    switch (k1s) {                                                                                                  \
        TS_NPG_LAUNCH(KERNEL, MODE, 2, GRID)                                                                        \
        TS_NPG_LAUNCH(KERNEL, MODE, 3, GRID)                                                                        \
        TS_NPG_LAUNCH(KERNEL, MODE, 5, GRID)                                                                        \
        TS_NPG_LAUNCH(KERNEL, MODE, 8, GRID)                                                                        \
        default: TS_REQUIRE(false, TS_ERR_UNSUPPORTED, "npg fused pass: unsupported obs_dim");                      \
    }                                                                                                               \
    TS_LAUNCH_CHECK();

static int npg_param_count(int k0) { return (k0 + 1) * HID + (HID + 1) * HID + (HID + 1) * 32 + 32; }

// out = F v + damping v on B rows of x ([B][k0] zero-padded observations); theta / v / out in ts_npg.hip's block layout
int npg_fvp_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* v, const float* x, int obs, int k0, int act,
                  int64_t B, float damping, float* slabs, float* out) {
    const int k1s = q4::k1s_for(obs);
    TS_REQUIRE(k1s > 0 && B >= 1, TS_ERR_UNSUPPORTED, "npg_fvp_fused: unsupported shape");
    q4::ActorArgs g{};
    g.theta = theta; g.dir = v; g.x = x; g.n_rows = B; g.inv_batch = 1.f / (float)B;
    g.slabs = slabs; g.slab_w = q4::actor_slab_width(k1s);
    g.obs = obs; g.act = act; g.k0 = k0;
    const int grid = npg_grid(B, 2);
    TS_NPG_DISPATCH(npg_fvp_kernel, q4::NPG_FVP, dim3(grid))
    const int P = npg_param_count(k0);
    hipLaunchKernelGGL(q4::npg_actor_reduce_kernel, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, s, slabs, grid, g.slab_w, obs,
                       act, k0, 4 * k1s, v, out, P, damping, (float*)nullptr, (float)B, 0.f);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// grad = d surrogate / d theta (logp_old == NULL: -mean(logp adv), else -mean(exp(logp - logp_old) adv)), loss_out[0] = the
// surrogate, mu[B][8] = the policy mean at theta
int npg_grad_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* x, const float* actions, const float* adv,
                   const float* logp_old, int obs, int k0, int act, int64_t B, float* slabs, float* grad, float* loss_out,
                   float* mu) {
    const int k1s = q4::k1s_for(obs);
    TS_REQUIRE(k1s > 0 && B >= 1, TS_ERR_UNSUPPORTED, "npg_grad_fused: unsupported shape");
    q4::ActorArgs g{};
    g.theta = theta; g.x = x; g.n_rows = B; g.inv_batch = 1.f / (float)B;
    g.slabs = slabs; g.slab_w = q4::actor_slab_width(k1s);
    g.obs = obs; g.act = act; g.k0 = k0;
    g.actions = actions; g.adv = adv; g.logp_old = logp_old; g.mu = mu;
    const int grid = npg_grid(B, 2);
    TS_NPG_DISPATCH(npg_grad_kernel, q4::NPG_GRAD, dim3(grid))
    const int P = npg_param_count(k0);
    hipLaunchKernelGGL(q4::npg_actor_reduce_kernel, dim3((unsigned)((P + 1 + 63) / 64)), dim3(1024), 0, s, slabs, grid, g.slab_w,
                       obs, act, k0, 4 * k1s, (const float*)nullptr, grad, P, 0.f, loss_out, (float)B, -1.f);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// grad = d mse_loss(returns, V) / d critic (block layout without a log_sigma block), loss_out[0] = the loss
int npg_critic_grad_fused(hipStream_t s, ts_workspace* ws, const float* critic, const float* x, const float* returns, int obs,
                          int k0, int64_t B, float* slabs, float* grad, float* loss_out) {
    const int k1s = q4::k1s_for(obs);
    TS_REQUIRE(k1s > 0 && B >= 1, TS_ERR_UNSUPPORTED, "npg_critic_grad_fused: unsupported shape");
    q4::ActorArgs g{};
    g.theta = critic; g.x = x; g.n_rows = B; g.inv_batch = 1.f / (float)B;
    g.slabs = slabs; g.slab_w = q4::actor_slab_width(k1s);
    g.obs = obs; g.act = 1; g.k0 = k0;
    g.adv = returns;
    const int grid = npg_grid(B, 2, "TS_NPG_CRITIC_WGS");
    TS_NPG_DISPATCH(npg_critic_kernel, q4::NPG_CRITIC, dim3(grid))
    const int P = npg_param_count(k0) - 32;
    hipLaunchKernelGGL(q4::npg_actor_reduce_kernel, dim3((unsigned)((P + 1 + 63) / 64)), dim3(1024), 0, s, slabs, grid, g.slab_w,
                       obs, 1, k0, 4 * k1s, (const float*)nullptr, grad, P, 0.f, loss_out, (float)B, 1.f);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// res[2 c + {0, 1}] = {mean kl(old || candidate c), -mean(ratio adv) at candidate c (logp_old != NULL)}; candidates at
// cands + c * cand_stride; mu = the old mean per sample (npg_grad_fused).  apply_theta (NPG, n_cand == 1): the finish kernel
// also copies the candidate over theta and writes stats3[1] = kl, stats3[2] = 0
int npg_eval_fused(hipStream_t s, ts_workspace* ws, const float* theta_old, const float* cands, int64_t cand_stride, int n_cand,
                   const float* x, const float* actions, const float* adv, const float* logp_old, const float* mu, int obs, int k0,
                   int act, int64_t B, float* partial, float* res, float* apply_theta, float* apply_stats3) {
    const int k1s = q4::k1s_for(obs);
    TS_REQUIRE(k1s > 0 && B >= 1 && n_cand >= 1 && n_cand <= 32, TS_ERR_UNSUPPORTED, "npg_eval_fused: unsupported shape");
    q4::ActorArgs g{};
    g.theta = cands; g.cand_stride = cand_stride; g.theta_old = theta_old; g.x = x; g.n_rows = B; g.inv_batch = 1.f / (float)B;
    g.slabs = partial;
    g.obs = obs; g.act = act; g.k0 = k0;
    g.actions = actions; g.adv = adv; g.logp_old = logp_old; g.mu = const_cast<float*>(mu);
    const int grid = npg_eval_grid(B, n_cand);
    TS_NPG_DISPATCH(npg_eval_kernel, q4::NPG_EVAL, dim3(grid, n_cand))
    TS_REQUIRE(!apply_theta || n_cand == 1, TS_ERR_INVALID_ARG, "npg_eval_fused: only a single candidate can be applied");
    hipLaunchKernelGGL(q4::npg_eval_finish_kernel, dim3(n_cand), dim3(256), 0, s, partial, grid, (float)B, res,
                       q4::NpgApply{apply_theta, cands, npg_param_count(k0), apply_stats3});
    TS_LAUNCH_CHECK();
    return TS_OK;
}
// forward pass on B rows: head_out[row * head_stride + a] = the head's output a < act (NULL: skipped), logp_out[row] =
// log pi(actions[row]) (NULL: skipped; needs actions and an actor's vector).  A critic: act = 1, head_stride = 1.
int npg_infer_fused(hipStream_t s, ts_workspace* ws, const float* theta, const float* x, const float* actions, int obs, int k0,
                    int act, int64_t B, float* head_out, int head_stride, float* logp_out) {
    const int k1s = q4::k1s_for(obs);
    TS_REQUIRE(k1s > 0 && B >= 1 && (!logp_out || actions), TS_ERR_UNSUPPORTED, "npg_infer_fused: unsupported shape");
    q4::ActorArgs g{};
    g.theta = theta; g.x = x; g.n_rows = B; g.inv_batch = 1.f / (float)B;
    g.obs = obs; g.act = act; g.k0 = k0;
    g.actions = logp_out ? actions : nullptr; g.mu = head_out; g.mu_stride = head_stride; g.logp_out = logp_out;
    const int grid = npg_eval_grid(B, 1);
    TS_NPG_DISPATCH(npg_infer_kernel, q4::NPG_INFER, dim3(grid))
    return TS_OK;
}
#undef TS_NPG_DISPATCH
#undef TS_NPG_LAUNCH

}  // namespace ts

extern "C" {

int64_t ts_ppo_param_count(int64_t obs_dim, int64_t act_dim) {
    if (obs_dim < 1 || act_dim < 1) return -1;
    return make_dims((int)obs_dim, (int)act_dim).p_total;
}

int ts_ppo_infer(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, const float* obs,
                 const float* act, int64_t n, float* v_out, float* logp_out, ts_stream_t stream) {
    return ts_ppo_infer_bounded(ws, params, obs_dim, act_dim, 0.0, obs, act, n, v_out, logp_out, nullptr, stream);
}

int ts_ppo_infer_bounded(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, double max_action,
                         const float* obs, const float* act, int64_t n, float* v_out, float* logp_out, float* mu_out,
                         ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n >= 0 && max_action >= 0.0, TS_ERR_INVALID_ARG, "ts_ppo_infer: negative n / max_action");
    if (n == 0 || (!v_out && !logp_out && !mu_out)) return TS_OK;
    TS_REQUIRE(params && obs, TS_ERR_INVALID_ARG, "ts_ppo_infer: NULL params / obs");
    TS_REQUIRE(!logp_out || act, TS_ERR_INVALID_ARG, "ts_ppo_infer: logp_out needs act");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    hipStream_t s = ts::as_stream(stream);
    const int64_t tiles = (n + 31) / 32;
    int64_t wg = (tiles + 7) / 8;
    const int64_t cap = (int64_t)n_compute_units() * 2;
    if (wg > cap) wg = cap;
    {
        ts::ProfScope prof(ws, TS_KIND_PPO_INFER, s);
        TS_KS1_DISPATCH(ks, {
            hipLaunchKernelGGL((ppo_infer_kernel<K>), dim3((unsigned)wg), dim3(512), infer_lds_bytes<K>(), s,
                               params, d, obs, act, n, v_out, logp_out, mu_out, (float)max_action);
        });
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// act = mu + sigma * eps (dist.sample(); eps NULL = dist.mode), then Algorithm.map_action
// (algorithm_base.py:274-287): optional clip / tanh bounding and scaling to [low, high].
__global__ __launch_bounds__(256) void ppo_sample_map_kernel(const float* __restrict__ mu, const float* __restrict__ noise,
                                                             const float* __restrict__ sigma_param, int64_t n, int act_dim,
                                                             int bound, const float* __restrict__ low,
                                                             const float* __restrict__ high,
                                                             float* __restrict__ act_out, float* __restrict__ mapped_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * act_dim) return;
    const int k = (int)(i % act_dim);
    float a = mu[i];
    if (noise) a = a + expf(sigma_param[k]) * noise[i];
    act_out[i] = a;
    if (mapped_out) {
        float m = a;
        if (bound == 1) m = fminf(fmaxf(m, -1.f), 1.f);            // "clip"
        else if (bound == 2) m = tanhf(m);                         // "tanh"
        if (low) m = low[k] + (high[k] - low[k]) * (m + 1.f) / 2.f;
        mapped_out[i] = m;
    }
}

int ts_ppo_policy_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, const float* obs,
                          const float* noise, int64_t n, int bound_method, const float* low, const float* high,
                          float* act_out, float* mapped_out, ts_stream_t stream) {
    return ts_ppo_policy_forward_bounded(ws, params, obs_dim, act_dim, 0.0, obs, noise, n, bound_method, low, high, act_out,
                                         mapped_out, nullptr, stream);
}

int ts_ppo_policy_forward_bounded(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, double max_action,
                                  const float* obs, const float* noise, int64_t n, int bound_method, const float* low,
                                  const float* high, float* act_out, float* mapped_out, float* mu_out, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n >= 0 && bound_method >= 0 && bound_method <= 2 && max_action >= 0.0, TS_ERR_INVALID_ARG,
               "ts_ppo_policy_forward: bad argument");
    if (n == 0) return TS_OK;
    TS_REQUIRE(ws && params && obs && act_out && ((low == nullptr) == (high == nullptr)), TS_ERR_INVALID_ARG,
               "ts_ppo_policy_forward: NULL argument");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    hipStream_t s = ts::as_stream(stream);
    float* mu = mu_out;                                   // the caller's `logits[0]`, or scratch
    if (!mu) {
        if (int rc2 = ts::ws_reserve(ws, sizeof(float) * (size_t)n * (size_t)act_dim)) return rc2;
        mu = static_cast<float*>(ws->base);
    }
    const int64_t tiles = (n + 31) / 32;
    int64_t wg = (tiles + 7) / 8;
    const int64_t cap = (int64_t)n_compute_units() * 2;
    if (wg > cap) wg = cap;
    {
        ts::ProfScope prof(ws, TS_KIND_PPO_INFER, s);
        TS_KS1_DISPATCH(ks, {
            hipLaunchKernelGGL((ppo_infer_kernel<K>), dim3((unsigned)wg), dim3(512), infer_lds_bytes<K>(), s,
                               params, d, obs, (const float*)nullptr, n, (float*)nullptr, (float*)nullptr, mu,
                               (float)max_action);
        });
    }
    hipLaunchKernelGGL(ppo_sample_map_kernel, dim3((unsigned)ts::ceil_div(n * act_dim, 256)), dim3(256), 0, s, mu, noise,
                       params + d.a_sig, n, (int)act_dim, bound_method, low, high, act_out, mapped_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_gauss_sample_map(const float* mu, const float* noise, const float* log_sigma, int64_t n, int64_t act_dim,
                        int bound_method, const float* low, const float* high, float* act_out, float* mapped_out,
                        ts_stream_t stream) {
    TS_REQUIRE(n >= 0 && act_dim >= 1 && bound_method >= 0 && bound_method <= 2, TS_ERR_INVALID_ARG, "ts_gauss_sample_map: bad argument");
    if (n == 0) return TS_OK;
    TS_REQUIRE(mu && act_out && (!noise || log_sigma) && ((low == nullptr) == (high == nullptr)), TS_ERR_INVALID_ARG,
               "ts_gauss_sample_map: NULL argument");
    hipLaunchKernelGGL(ppo_sample_map_kernel, dim3((unsigned)ts::ceil_div(n * act_dim, 256)), dim3(256), 0, ts::as_stream(stream),
                       mu, noise, log_sigma, n, (int)act_dim, bound_method, low, high, act_out, mapped_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

static int rec_width(int64_t obs_dim, int64_t act_dim) { return (int)((obs_dim + act_dim + 4 + 3) & ~(int64_t)3); }

int64_t ts_ppo_record_width(int64_t obs_dim, int64_t act_dim) {
    if (obs_dim < 1 || act_dim < 1) return -1;
    return rec_width(obs_dim, act_dim);
}

int ts_ppo_pack_batch(const float* obs, const float* act, const float* adv, const float* returns,
                      const float* logp_old, const float* v_s, int64_t n, int64_t obs_dim,
                      int64_t act_dim, float* rec_out, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_ppo_pack_batch: negative n");
    if (n == 0) return TS_OK;
    TS_REQUIRE(obs && act && adv && returns && logp_old && v_s && rec_out, TS_ERR_INVALID_ARG,
               "ts_ppo_pack_batch: NULL argument");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(rec_out) & 15u) == 0, TS_ERR_INVALID_ARG,
               "ts_ppo_pack_batch: rec_out must be 16-byte aligned");
    const int rw = rec_width(obs_dim, act_dim);
    int64_t blocks = (n * rw + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ppo_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, ts::as_stream(stream), obs, act,
                       adv, returns, logp_old, v_s, n, (int)obs_dim, (int)act_dim, rw, rec_out);
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
    TS_REQUIRE(n >= 1 && n_steps >= 0 && adam_step0 >= 0, TS_ERR_INVALID_ARG, "ts_ppo_update: bad size");
    if (n_steps == 0) return TS_OK;
    TS_REQUIRE(ws && params && adam_m && adam_v && obs && act && adv && returns && logp_old && v_s &&
                   h_mb_offset && hp,
               TS_ERR_INVALID_ARG, "ts_ppo_update: NULL argument");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    // The slab area, the gradient vector and the partial sums of squares are sized by the LARGEST need over the update's
    // minibatches -- not by the largest minibatch: Batch.split(merge_last=True) makes the last one the largest, and the
    // others may run the other step kernel, whose slabs are wider and whose grid is its own (e.g. 20,000-row minibatches on
    // the feature-split kernel: 256 slabs of 11,608 floats; the 28,000-row last one on the 128-sample kernel: 219 of 11,088).
    size_t slab_floats = 0;
    int slab_w = 0;
    int64_t last_rows = -1;
    for (int64_t k = 0; k < n_steps; ++k) {
        const int64_t rows = h_mb_offset[k + 1] - h_mb_offset[k];
        TS_REQUIRE(rows >= 1, TS_ERR_SHAPE, "ts_ppo_update: minibatch %lld is empty", (long long)k);
        TS_REQUIRE(perm || h_mb_offset[k + 1] <= n, TS_ERR_SHAPE, "ts_ppo_update: minibatch beyond n");
        if (rows == last_rows) continue;
        last_rows = rows;
        const StepPlan pl = plan_step(d, ks, rows, hp->nets, hp->max_action > 0.0);
        slab_floats = std::max(slab_floats, (size_t)pl.n_slabs * (size_t)pl.slab_w);
        slab_w = std::max(slab_w, pl.slab_w);
    }
    const int rw = rec_width(obs_dim, act_dim);
    const WsLayout wl = ws_layout((int)((slab_floats + slab_w - 1) / slab_w), slab_w, n_steps);
    // behind the fixed part: device copy of the minibatch offsets, then the packed records
    const size_t off_bytes = (sizeof(int64_t) * (size_t)(n_steps + 1) + 255) & ~(size_t)255;
    const size_t rec_bytes = (sizeof(float) * (size_t)n * rw + 255) & ~(size_t)255;
    const ImageBuf ib = image_buf(d, ks);
    const size_t img_bytes = ib.img_bytes, inv_bytes = ib.inv_bytes;
    const int img_end = ib.img_end;
    rc = ts::ws_reserve(ws, wl.total + off_bytes + rec_bytes + img_bytes + inv_bytes + 256);
    if (rc != TS_OK) return rc;
    char* base = reinterpret_cast<char*>(ws->base);
    float* image = reinterpret_cast<float*>(base + wl.total + off_bytes + rec_bytes);
    int* inv = reinterpret_cast<int*>(base + wl.total + off_bytes + rec_bytes + img_bytes);
    float* slabs = reinterpret_cast<float*>(base + wl.slabs);
    float* grad = reinterpret_cast<float*>(base + wl.grad);
    float* sumsq = reinterpret_cast<float*>(base + wl.sumsq);
    float* advstats = reinterpret_cast<float*>(base + wl.advstats);
    int64_t* d_off = reinterpret_cast<int64_t*>(base + wl.total);
    float* rec = reinterpret_cast<float*>(base + wl.total + off_bytes);
    hipStream_t s = ts::as_stream(stream);

    rc = ts_ppo_pack_batch(obs, act, adv, returns, logp_old, v_s, n, obs_dim, act_dim, rec, stream);
    if (rc != TS_OK) return rc;
    rc = build_image(s, params, d, ks, image, inv);
    if (rc != TS_OK) return rc;
    if (hp->adv_norm) {
        TS_HIP_CHECK(hipMemcpyAsync(d_off, h_mb_offset, sizeof(int64_t) * (size_t)(n_steps + 1),
                                    hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ppo_adv_stats_kernel, dim3((unsigned)n_steps), dim3(1024), 0, s, adv, perm,
                           d_off, advstats);
        TS_LAUNCH_CHECK();
    }
    for (int64_t k = 0; k < n_steps; ++k) {
        StepArgs g{};
        g.params = params;
        g.rec_w = rw;
        g.n_rows = h_mb_offset[k + 1] - h_mb_offset[k];
        if (perm) { g.rec = rec; g.rows = perm + h_mb_offset[k]; }
        else { g.rec = rec + h_mb_offset[k] * rw; g.rows = nullptr; }   // identity rows: shift the base
        g.inv_batch = 1.0f / (float)g.n_rows;
        g.adv_stats = hp->adv_norm ? advstats + 2 * k : nullptr;
        g.image = image;
        fill_hparams(g, hp);
        float* losses = losses_out ? losses_out + 4 * k : nullptr;
        const StepPlan pl = plan_step(d, ks, g.n_rows, hp->nets, hp->max_action > 0.0);
        rc = run_grad(ws, g, d, ks, pl, slabs, grad, sumsq, losses, s);
        if (rc != TS_OK) return rc;
        AdamArgs a = adam_args(params, adam_m, adam_v, adam_step0 + k + 1, d, hp);
        // the partial sums of squares THIS step's reduction wrote: a ragged last minibatch may run the other step kernel,
        // whose slabs are wider (182 vs 174 reduction workgroups at obs 17 / act 6)
        a.grad = grad; a.sumsq_part = sumsq; a.n_part = (pl.slab_w + 63) / 64;
        a.losses = losses; a.apply = 1;
        a.image = image; a.inv = inv; a.sig_off = d.a_sig; a.act = d.act; a.small0 = img_end - 32;
        if (grads_out && k == n_steps - 1)
            TS_HIP_CHECK(hipMemcpyAsync(grads_out, grad, sizeof(float) * (size_t)d.p_total,
                                        hipMemcpyDeviceToDevice, s));
        {
            ts::ProfScope prof(ws, TS_KIND_PPO_ADAM, s);
            hipLaunchKernelGGL(ppo_adam_kernel, dim3((d.p_total + ADAM_THREADS - 1) / ADAM_THREADS),
                               dim3(ADAM_THREADS), 0, s, a);
        }
        TS_LAUNCH_CHECK();
    }
    return TS_OK;
}

int ts_ppo_grad(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim,
                const float* rec, int64_t n, const int64_t* perm_rows, int64_t n_rows,
                int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp,
                float* grad_out, float* loss_parts_out, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n_rows >= 1 && global_batch >= n_rows, TS_ERR_SHAPE, "ts_ppo_grad: bad batch sizes");
    TS_REQUIRE(ws && params && rec && hp && grad_out, TS_ERR_INVALID_ARG, "ts_ppo_grad: NULL argument");
    TS_REQUIRE(perm_rows || n_rows <= n, TS_ERR_SHAPE, "ts_ppo_grad: n_rows > n");
    TS_REQUIRE(!hp->adv_norm || adv_stats, TS_ERR_INVALID_ARG, "ts_ppo_grad: adv_norm needs adv_stats");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(rec) & 15u) == 0, TS_ERR_INVALID_ARG,
               "ts_ppo_grad: rec must be 16-byte aligned");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    const StepPlan pl = plan_step(d, ks, n_rows, hp->nets, hp->max_action > 0.0);
    const WsLayout wl = ws_layout(pl.n_slabs, pl.slab_w, 1);
    rc = ts::ws_reserve(ws, wl.total);
    if (rc != TS_OK) return rc;
    char* base = reinterpret_cast<char*>(ws->base);
    hipStream_t s = ts::as_stream(stream);
    float* image; int* inv;
    rc = dp_image(ws, s, params, d, ks, &image, &inv);
    if (rc != TS_OK) return rc;
    StepArgs g{};
    g.params = params; g.rec = rec; g.rec_w = rec_width(obs_dim, act_dim);
    g.rows = perm_rows; g.n_rows = n_rows; g.image = image;
    g.inv_batch = 1.0f / (float)global_batch;
    g.adv_stats = hp->adv_norm ? adv_stats : nullptr;
    fill_hparams(g, hp);
    // the slab reduction writes the gradient straight into grad_out and the loss sums into loss_parts_out
    float* parts = loss_parts_out ? loss_parts_out : reinterpret_cast<float*>(base + wl.advstats);
    return run_grad(ws, g, d, ks, pl, reinterpret_cast<float*>(base + wl.slabs), grad_out,
                    reinterpret_cast<float*>(base + wl.sumsq), nullptr, s, parts);
}

#ifdef TS_PHASE_MARKS
int ts_debug_ppo_step_cycles(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim,
                             const float* rec, const int64_t* perm_rows, int64_t n_rows,
                             const ts_ppo_hparams* hp, int64_t* h_cycles, int64_t n_marks,
                             ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(ws && params && rec && hp && h_cycles, TS_ERR_INVALID_ARG,
               "ts_debug_ppo_step_cycles: NULL argument");
    TS_REQUIRE(n_marks >= 18 && n_rows >= 1, TS_ERR_INVALID_ARG, "ts_debug_ppo_step_cycles: need >= 18 marks");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const int ks = supported_ks(ks1_for((int)obs_dim));
    const int slab_w = slab_width(d, ks);
    const int n_wg = step_grid(n_rows);
    const WsLayout wl = ws_layout(n_wg > 512 ? n_wg : 512, slab_w > 12288 ? slab_w : 12288, 1);
    const ImageBuf ib = image_buf(d, ks);
    const size_t dbg_bytes = 16384;          // 64 phase marks of workgroup 0 + (start, end) of up to 992 workgroups
    rc = ts::ws_reserve(ws, wl.total + dbg_bytes + ib.img_bytes + ib.inv_bytes);
    if (rc != TS_OK) return rc;
    char* base = reinterpret_cast<char*>(ws->base);
    long long* dbg = reinterpret_cast<long long*>(base + wl.total);
    float* image = reinterpret_cast<float*>(base + wl.total + dbg_bytes);
    hipStream_t s = ts::as_stream(stream);
    TS_HIP_CHECK(hipMemsetAsync(dbg, 0, dbg_bytes, s));
    rc = build_image(s, params, d, ks, image, nullptr);
    if (rc != TS_OK) return rc;
    StepArgs g{};
    g.params = params; g.rec = rec; g.rec_w = rec_width(obs_dim, act_dim);
    g.rows = perm_rows; g.n_rows = n_rows; g.image = image;
    g.inv_batch = 1.0f / (float)n_rows;
    fill_hparams(g, hp);
    g.adv_norm = 0;
    g.slabs = reinterpret_cast<float*>(base + wl.slabs); g.slab_w = slab_w; g.dbg = dbg;
    const StepPlan pl = plan_step(d, ks, n_rows, hp->nets, hp->max_action > 0.0);
    if (pl.variant == 1) {
        g.slab_w = pl.slab_w;        // (the slab area above is sized for the 128-sample kernel: n_wg x slab_w >= pairs x slab3_w)
        switch (pl.k1s) {
            case 2: rc = launch_stepq<2>(ws, g, d, pl, s); break;
            case 3: rc = launch_stepq<3>(ws, g, d, pl, s); break;
            case 5: rc = launch_stepq<5>(ws, g, d, pl, s); break;
            default: rc = launch_stepq<8>(ws, g, d, pl, s); break;
        }
    } else {
        TS_KS1_DISPATCH(ks, { rc = launch_step<K>(ws, g, d, n_wg, s); });
    }
    if (rc != TS_OK) return rc;
    static long long host[2048];
    TS_HIP_CHECK(hipMemcpyAsync(host, dbg, sizeof(host), hipMemcpyDeviceToHost, s));
    TS_HIP_CHECK(hipStreamSynchronize(s));
    for (int64_t k = 0; k < n_marks && k < 2048; ++k) h_cycles[k] = host[k];
    return TS_OK;
}
#endif

// One data-parallel gradient step behind ONE call: local gradient -> sum all-reduce of [grad | loss parts] -> clip + Adam.
int ts_ppo_dp_step(ts_workspace* ws, ts_comm* comm, float* params, float* adam_m, float* adam_v, int64_t adam_step,
                   int64_t obs_dim, int64_t act_dim, const float* rec, int64_t n, const int64_t* perm_rows, int64_t n_rows,
                   int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp, float* step_buf,
                   ts_stream_t stream) {
    TS_REQUIRE(step_buf != nullptr, TS_ERR_INVALID_ARG, "ts_ppo_dp_step: step_buf is NULL");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(step_buf) & 15u) == 0, TS_ERR_INVALID_ARG, "ts_ppo_dp_step: step_buf must be 16-byte aligned");
    const int64_t P = ts_ppo_param_count(obs_dim, act_dim);
    if (P < 0) return ts::fail(TS_ERR_INVALID_ARG, "ts_ppo_dp_step: bad dimensions");
    int rc = ts_ppo_grad(ws, params, obs_dim, act_dim, rec, n, perm_rows, n_rows, global_batch, adv_stats, hp, step_buf,
                         step_buf + P, stream);
    if (rc != TS_OK) return rc;
    if (comm) {
        rc = ts_allreduce(comm, step_buf, P + 4, stream);          // gradient + the four loss parts in one exchange
        if (rc != TS_OK) return rc;
    }
    return ts_ppo_apply(ws, params, adam_m, adam_v, adam_step, obs_dim, act_dim, step_buf, hp, stream);
}

int ts_ppo_step_plan(int64_t obs_dim, int64_t act_dim, int64_t n_rows, int32_t nets, int32_t* variant_out,
                     int32_t* grid_out, int32_t* slabs_out) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(n_rows >= 1, TS_ERR_INVALID_ARG, "ts_ppo_step_plan: n_rows must be >= 1");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    const StepPlan pl = plan_step(d, supported_ks(ks1_for((int)obs_dim)), n_rows, nets);
    if (variant_out) *variant_out = pl.variant == 0 ? 0 : (pl.big ? 2 : 1);
    if (grid_out) *grid_out = pl.grid;
    if (slabs_out) *slabs_out = pl.n_slabs;
    return TS_OK;
}

int ts_ppo_invalidate_image(ts_workspace* ws) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_ppo_invalidate_image: workspace is NULL");
    ws->ppo_image_params = nullptr;
    return TS_OK;
}

int ts_ppo_apply(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                 int64_t act_dim, const float* grad, const ts_ppo_hparams* hp, ts_stream_t stream) {
    int rc = check_dims(obs_dim, act_dim);
    if (rc != TS_OK) return rc;
    TS_REQUIRE(params && adam_m && adam_v && grad && hp, TS_ERR_INVALID_ARG, "ts_ppo_apply: NULL argument");
    TS_REQUIRE(adam_step >= 1, TS_ERR_INVALID_ARG, "ts_ppo_apply: adam_step counts from 1");
    const Dims d = make_dims((int)obs_dim, (int)act_dim);
    AdamArgs a = adam_args(params, adam_m, adam_v, adam_step, d, hp);
    a.grad = grad; a.sumsq_part = nullptr; a.n_part = 0; a.losses = nullptr; a.apply = 1;
    if (ws && ws->ppo_image && ws->ppo_image_params == params &&
        ws->ppo_image_key == ((d.obs << 8) | d.act)) {
        // keep ts_ppo_grad's images current (same mechanism as ts_ppo_update)
        const ImageBuf ib = image_buf(d, supported_ks(ks1_for((int)obs_dim)));
        a.image = static_cast<float*>(ws->ppo_image);
        a.inv = reinterpret_cast<const int*>(static_cast<char*>(ws->ppo_image) + ib.img_bytes);
        a.sig_off = d.a_sig; a.act = d.act; a.small0 = ib.img_end - 32;
    }
    hipLaunchKernelGGL(ppo_adam_kernel, dim3((d.p_total + ADAM_THREADS - 1) / ADAM_THREADS),
                       dim3(ADAM_THREADS), 0, ts::as_stream(stream), a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
