// ts_mlp_small.hip -- the whole PPO / A2C `_update_with_batch` of a SMALL discrete actor-critic in ONE launch (gfx950).
//
// BASELINE.json configs[0] (test/discrete/test_ppo_discrete.py:88-127: CartPole, obs 4, Net[64, 64] ReLU shared by
// DiscreteActor and DiscreteCritic, 2 actions, 2000 transitions, minibatch 64, repeat 10) is 310 dependent gradient steps
// of ~1.4 MFLOP each: as separate launches (ts_mlp_ppo_step: ~16 kernels per step + the host's row gathers) the update is
// pure launch latency -- 92 us per step, roofline fraction 0.001.  Here one persistent 256-thread workgroup runs the loop
// of ppo.py:174-216 + Optimizer.step (algorithm_base.py:484-500) end to end:
//   * the three weight matrices (8,352 floats) live in LDS for the whole update, the Adam moments and the gradient
//     accumulators live in the REGISTERS of the lane that owns each weight (the MFMA C/D layout of the weight-gradient
//     tiles fixes that ownership once), so an optimizer step touches no memory beyond LDS;
//   * every GEMM of a 64-sample chunk (forward x3, input gradients x2, weight gradients x3) is a handful of
//     v_mfma_f32_32x32x2_f32 tiles whose operands are read straight from odd-pitch LDS arrays (pitch 65 / 33: both the
//     row- and the column-major walk of a matrix are bank-conflict free, so no transposed copies exist);
//   * the rows of the next chunk (observation, action, advantage, return, log pi_old, V_old, gathered through the
//     minibatch permutation) are fetched into registers while the current chunk computes; minibatches larger than 64
//     rows (Batch.split's merged last one, batch.py:1205-1215) are several chunks accumulated in the same registers.
// Arithmetic = the reference's fp32 (Categorical log-softmax / entropy, clipped surrogate, dual clip, clipped value loss,
// clip_grad_norm_, Adam with torch's single-tensor formulas); sums run in a different order than torch's, parity bars in
// tests/test_gpu_ppo_discrete.py.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ts_common.h"

#pragma clang fp contract(off)

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int HID = 64, K0 = 32, HEAD = 32, CH = 64, NT = 256;
constexpr int P65 = 65, P33 = 33;
// LDS carve (floats)
constexpr int L_W1 = 0;                          // [K0 + 1][65]   row k (row K0 = bias), column f
constexpr int L_W2 = L_W1 + (K0 + 1) * P65;      // [65][65]
constexpr int L_WH = L_W2 + (HID + 1) * P65;     // [65][33]       columns [0, A) logits, A = value, rest zero
constexpr int L_X = L_WH + (HID + 1) * P33;      // [64][33]       chunk observations, zero padded
constexpr int L_H1 = L_X + CH * P33;             // [64][65]       post-ReLU activations
constexpr int L_H2 = L_H1 + CH * P65;
constexpr int L_D2 = L_H2 + CH * P65;            // dL/d(pre-activation 2)
constexpr int L_D1 = L_D2 + CH * P65;
constexpr int L_O = L_D1 + CH * P65;             // [64][33]       head outputs
constexpr int L_DO = L_O + CH * P33;
constexpr int L_SC = L_DO + CH * P33;            // [4][64] adv, ret, logp_old, v_old; [64] act (int bits)
constexpr int L_RED = L_SC + 5 * CH;             // [4] block-reduction scratch
constexpr int L_END = L_RED + 16;
constexpr int OFF2 = (K0 + 1) * HID, OFF3 = OFF2 + (HID + 1) * HID, P_TOTAL = OFF3 + (HID + 1) * HEAD;

struct Chunk {
    long long row0;        // offset of the chunk's first row in the concatenated row list
    int count;             // rows in this chunk (1..64)
    int step;              // gradient step (minibatch) the chunk belongs to
    int batch;             // rows of that minibatch
    int last;              // 1: last chunk of its minibatch -> optimizer step
};

struct StepCoef { float lr_step, bc2_sqrt; };

struct SmallArgs {
    float* params; float* m; float* v;
    const float* obs; const int64_t* act; const float* adv; const float* ret; const float* logp_old; const float* v_old;
    const int64_t* rows;
    const Chunk* chunks; int n_chunks;
    const float* adv_stats;        // [n_steps][2] {mean, std} or NULL
    const StepCoef* coef;          // [n_steps]
    int obs_dim, n_act;
    float eps_clip, dual_clip, vf_coef, ent_coef; int value_clip, algo;
    float max_norm, beta2, adam_eps, omb1, omb2;
    float* losses;                 // [n_steps][4]
};

__device__ __forceinline__ int featF(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// acc[r] (lane (i, h)) += sum_k A(m = F(r, h), k) B(k, n = i): the lane supplies A(i, k + h) and B(k + h, i) per k pair
template <typename FA, typename FB>
__device__ __forceinline__ f32x16 mma(f32x16 acc, int K, FA a_at, FB b_at) {
#pragma unroll 4
    for (int k = 0; k < K; k += 2) acc = mfma32(a_at(k), b_at(k), acc);
    return acc;
}

__device__ __forceinline__ f32x16 splat(float v) { return f32x16{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v}; }

// flat parameter index -> LDS address of the three weight images
__device__ __forceinline__ int lds_of_param(int p) {
    if (p < OFF2) return L_W1 + (p >> 6) * P65 + (p & 63);
    if (p < OFF3) { const int q = p - OFF2; return L_W2 + (q >> 6) * P65 + (q & 63); }
    const int q = p - OFF3;
    return L_WH + (q >> 5) * P33 + (q & 31);
}

__device__ __forceinline__ float logsumexp(const float* l, int A) {
    float m = l[0];
    for (int j = 1; j < A; ++j) m = fmaxf(m, l[j]);
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(l[j] - m);
    return m + logf(s);
}

// One owned weight-gradient tile: 16 gradient accumulators + the Adam moments of the same 16 weights.
struct Tile {
    f32x16 g, m, v;
    int flat0;             // flat parameter index of (row F(0, h), column i) -- rows advance by `ld`
    int ld;
};

__device__ __forceinline__ int tile_flat(const Tile& t, int r, int h) { return t.flat0 + (featF(r, h) - featF(0, h)) * t.ld; }

__global__ __launch_bounds__(NT, 1) void mlp_ppo_update_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    const int tm = wave >> 1, tn = wave & 1;
    const int A = a.n_act;
    const int kx = (a.obs_dim + 1) & ~1;             // forward-1 reduction length (columns beyond obs_dim are zero)
    const int kc = (A + 2) & ~1;                     // head columns that can be non-zero: [0, A], rounded up to even

    // ---- parameters -> LDS, moments -> owner registers
    for (int p = tid; p < P_TOTAL; p += NT) lds[lds_of_param(p)] = a.params[p];
    Tile tw2, tx;                                    // tw2: the wave's 32x32 tile of W2; tx: W1 (waves 0, 1) or WH (waves 2, 3)
    tw2.ld = HID; tw2.flat0 = OFF2 + (32 * tm + featF(0, h)) * HID + 32 * tn + i;
    if (wave < 2) { tx.ld = HID; tx.flat0 = featF(0, h) * HID + 32 * wave + i; }
    else { tx.ld = HEAD; tx.flat0 = OFF3 + (32 * (wave - 2) + featF(0, h)) * HEAD + i; }
    // bias owner: wave 1 lanes -> b1[lane], wave 2 -> b2[lane], wave 3 lanes < 32 -> bh[lane]
    const int bflat = wave == 1 ? K0 * HID + lane : (wave == 2 ? OFF2 + HID * HID + lane : (wave == 3 && lane < 32 ? OFF3 + HID * HEAD + lane : -1));
    float gb = 0.f, mb = 0.f, vb = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        tw2.m[r] = a.m[tile_flat(tw2, r, h)]; tw2.v[r] = a.v[tile_flat(tw2, r, h)];
        tx.m[r] = a.m[tile_flat(tx, r, h)]; tx.v[r] = a.v[tile_flat(tx, r, h)];
    }
    if (bflat >= 0) { mb = a.m[bflat]; vb = a.v[bflat]; }
    tw2.g = splat(0.f); tx.g = splat(0.f);

    // ---- prefetch pipeline: thread (s = tid >> 2, q = tid & 3) owns observation columns [8q, 8q + 8) and one scalar of
    // sample s.  `rown`: row id of sample s in the NEXT chunk (-1: none).
    const int s = tid >> 2, q = tid & 3;
    float ox[8], sc = 0.f;
    int64_t an = 0;
    auto row_of = [&](int ci) -> int64_t {
        if (ci >= a.n_chunks) return -1;
        const Chunk c = a.chunks[ci];
        return s < c.count ? a.rows[c.row0 + s] : -1;
    };
    // unconditional loads from clamped addresses + selects: a branch around a load makes hipcc wait for it on the spot
    const float* sc_src = q == 0 ? a.adv : (q == 1 ? a.ret : (q == 2 ? a.logp_old : a.v_old));
    const bool sc_on = sc_src != nullptr;
    if (!sc_on) sc_src = a.adv;
    auto fetch = [&](int64_t row) {
        const int64_t rc = row >= 0 ? row : 0;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = 8 * q + j;
            t[j] = a.obs[rc * a.obs_dim + (col < a.obs_dim ? col : 0)];
        }
        const float ts_ = sc_src[rc];
        const int64_t ta = a.act[rc];
#pragma unroll
        for (int j = 0; j < 8; ++j) ox[j] = (row >= 0 && 8 * q + j < a.obs_dim) ? t[j] : 0.f;
        sc = (row >= 0 && sc_on) ? ts_ : 0.f;
        an = row >= 0 ? ta : 0;
    };
    fetch(row_of(0));
    int64_t rown = row_of(1);
    float l_clip = 0.f, l_vf = 0.f, l_ent = 0.f;     // wave 0: per-lane partial loss sums of the current minibatch

    for (int ci = 0; ci < a.n_chunks; ++ci) {
        const Chunk c = a.chunks[ci];
        // commit the prefetched chunk, then start fetching the next one
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[L_X + s * P33 + 8 * q + j] = ox[j];
        lds[L_SC + q * CH + s] = sc;
        if (q == 0) lds[L_SC + 4 * CH + s] = __int_as_float((int)an);
        fetch(rown);
        rown = row_of(ci + 2);
        __syncthreads();

        // ---- forward 1: H1[s][f] = relu(sum_k X[s][k] W1[k][f] + b1[f])
        {
            const float* X = lds + L_X + (32 * tm + i) * P33 + h;
            const float* W = lds + L_W1 + h * P65 + 32 * tn + i;
            f32x16 acc = splat(lds[L_W1 + K0 * P65 + 32 * tn + i]);
            acc = mma(acc, kx, [&](int k) { return X[k]; }, [&](int k) { return W[k * P65]; });
            float* H = lds + L_H1 + (32 * tm) * P65 + 32 * tn + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) H[featF(r, h) * P65] = fmaxf(acc[r], 0.f);
        }
        __syncthreads();
        // ---- forward 2
        {
            const float* X = lds + L_H1 + (32 * tm + i) * P65 + h;
            const float* W = lds + L_W2 + h * P65 + 32 * tn + i;
            f32x16 acc = splat(lds[L_W2 + HID * P65 + 32 * tn + i]);
            acc = mma(acc, HID, [&](int k) { return X[k]; }, [&](int k) { return W[k * P65]; });
            float* H = lds + L_H2 + (32 * tm) * P65 + 32 * tn + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) H[featF(r, h) * P65] = fmaxf(acc[r], 0.f);
        }
        __syncthreads();
        // ---- heads: O[s][c] = sum_f H2[s][f] WH[f][c] + bh[c]; the four waves split (sample half) x (K half), the K
        // halves meet in LDS (wave pairs (0, 2) and (1, 3))
        {
            const int sm = wave & 1, kh = wave >> 1;
            const float* X = lds + L_H2 + (32 * sm + i) * P65 + 32 * kh + h;
            const float* W = lds + L_WH + (32 * kh + h) * P33 + i;
            f32x16 acc = splat(kh == 0 ? lds[L_WH + HID * P33 + i] : 0.f);
            acc = mma(acc, 32, [&](int k) { return X[k]; }, [&](int k) { return W[k * P33]; });
            float* O = lds + (kh == 0 ? L_O : L_DO) + (32 * sm) * P33 + i;       // second halves park in the dO array
#pragma unroll
            for (int r = 0; r < 16; ++r) O[featF(r, h) * P33] = acc[r];
        }
        __syncthreads();
        // ---- loss (wave 0, lane = sample): ppo.py:184-211 / a2c.py:262-273, exact torch tie semantics as ts_ppo_cnn.hip
        if (wave == 0) {
            float* o = lds + L_O + lane * P33;
            float* dd = lds + L_DO + lane * P33;
            const bool valid = lane < c.count;
            const float w = 1.f / (float)c.batch;
            for (int j = 0; j <= A; ++j) o[j] = o[j] + dd[j];                    // the two K halves
            const float* hb = o;
            const int act = __float_as_int(lds[L_SC + 4 * CH + lane]);
            const float lse = logsumexp(hb, A);
            float H = 0.f;
            for (int j = 0; j < A; ++j) { const float lp = hb[j] - lse; H -= expf(lp) * lp; }
            const float logp = hb[act] - lse;
            float Ad = lds[L_SC + 0 * CH + lane];
            if (a.adv_stats) Ad = (Ad - a.adv_stats[2 * c.step]) / (a.adv_stats[2 * c.step + 1] + 1e-8f);
            float term, dlogp;
            if (a.algo == 1) {
                term = -(logp * Ad);
                dlogp = -Ad * w;
            } else {
                const float ratio = expf(logp - lds[L_SC + 2 * CH + lane]);
                const float surr1 = ratio * Ad;
                const float surr2 = fminf(fmaxf(ratio, 1.f - a.eps_clip), 1.f + a.eps_clip) * Ad;
                const float clip1 = fminf(surr1, surr2);
                float basek = (surr1 <= surr2) ? Ad : 0.f;
                if (a.dual_clip > 0.f) {
                    const float clip2 = fmaxf(clip1, a.dual_clip * Ad);
                    if (Ad < 0.f) { term = -clip2; if (!(clip1 >= a.dual_clip * Ad)) basek = 0.f; }
                    else term = -clip1;
                } else {
                    term = -clip1;
                }
                dlogp = -basek * ratio * w;
            }
            const float value = hb[A], ret = lds[L_SC + 1 * CH + lane];
            const float vf1 = (ret - value) * (ret - value);
            float vterm, dv;
            if (a.value_clip && a.algo == 0) {
                const float vo = lds[L_SC + 3 * CH + lane], dvo = value - vo;
                const float vclip = vo + fminf(fmaxf(dvo, -a.eps_clip), a.eps_clip);
                const float vf2 = (ret - vclip) * (ret - vclip);
                vterm = fmaxf(vf1, vf2);
                const float g1 = -2.f * (ret - value);
                const float g2 = (dvo >= -a.eps_clip && dvo <= a.eps_clip) ? -2.f * (ret - vclip) : 0.f;
                dv = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
            } else {
                vterm = vf1;
                dv = -2.f * (ret - value);
            }
            for (int j = 0; j < HEAD; ++j) {
                float d = 0.f;
                if (valid && j < A) {
                    const float lp = hb[j] - lse, p = expf(lp);
                    d = dlogp * ((j == act ? 1.f : 0.f) - p) + a.ent_coef * w * p * (lp + H);
                }
                if (valid && j == A) d = dv * a.vf_coef * w;
                dd[j] = d;
            }
            if (valid) { l_clip += term; l_vf += vterm; l_ent += H; }
        }
        __syncthreads();
        // ---- dH2 = (dO . WH^T) * relu'(H2) -> D2
        {
            const float* X = lds + L_DO + (32 * tm + i) * P33 + h;
            const float* W = lds + L_WH + (32 * tn + i) * P33 + h;
            f32x16 acc = mma(splat(0.f), kc, [&](int k) { return X[k]; }, [&](int k) { return W[k]; });
            const float* H = lds + L_H2 + (32 * tm) * P65 + 32 * tn + i;
            float* D = lds + L_D2 + (32 * tm) * P65 + 32 * tn + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) D[featF(r, h) * P65] = H[featF(r, h) * P65] > 0.f ? acc[r] : 0.f;
        }
        __syncthreads();
        // ---- dW2 += H1^T . D2 (accumulated over the chunks of a minibatch); dH1 = (D2 . W2^T) * relu'(H1) -> D1;
        // wave 2: db2 += column sums of D2
        {
            const float* Am = lds + L_H1 + h * P65 + 32 * tm + i;
            const float* Bm = lds + L_D2 + h * P65 + 32 * tn + i;
            tw2.g = mma(tw2.g, CH, [&](int k) { return Am[k * P65]; }, [&](int k) { return Bm[k * P65]; });
            const float* X = lds + L_D2 + (32 * tm + i) * P65 + h;
            const float* W = lds + L_W2 + (32 * tn + i) * P65 + h;
            f32x16 acc = mma(splat(0.f), HID, [&](int k) { return X[k]; }, [&](int k) { return W[k]; });
            const float* H = lds + L_H1 + (32 * tm) * P65 + 32 * tn + i;
            float* D = lds + L_D1 + (32 * tm) * P65 + 32 * tn + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) D[featF(r, h) * P65] = H[featF(r, h) * P65] > 0.f ? acc[r] : 0.f;
            if (wave == 2) {
                float t = 0.f;
#pragma unroll 8
                for (int r = 0; r < CH; ++r) t += lds[L_D2 + r * P65 + lane];
                gb += t;
            }
        }
        __syncthreads();
        // ---- waves 0, 1: dW1 += X^T . D1;  waves 2, 3: dWH += H2^T . dO;  wave 1: db1, wave 3: dbh
        if (wave < 2) {
            const float* Am = lds + L_X + h * P33 + i;
            const float* Bm = lds + L_D1 + h * P65 + 32 * wave + i;
            tx.g = mma(tx.g, CH, [&](int k) { return Am[k * P33]; }, [&](int k) { return Bm[k * P65]; });
        } else {
            const float* Am = lds + L_H2 + h * P65 + 32 * (wave - 2) + i;
            const float* Bm = lds + L_DO + h * P33 + i;
            tx.g = mma(tx.g, CH, [&](int k) { return Am[k * P65]; }, [&](int k) { return Bm[k * P33]; });
        }
        if (wave == 1) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < CH; ++r) t += lds[L_D1 + r * P65 + lane];
            gb += t;
        } else if (wave == 3 && lane < 32) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < CH; ++r) t += lds[L_DO + r * P33 + lane];
            gb += t;
        }

        if (c.last) {
            // ---- Optimizer.step: clip_grad_norm_ over all parameters, then Adam (torch single-tensor arithmetic)
            float ss = gb * gb;
#pragma unroll
            for (int r = 0; r < 16; ++r) ss += tw2.g[r] * tw2.g[r] + tx.g[r] * tx.g[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
            if (lane == 0) lds[L_RED + wave] = ss;
            __syncthreads();
            const float norm = sqrtf((lds[L_RED] + lds[L_RED + 1]) + (lds[L_RED + 2] + lds[L_RED + 3]));
            const float scale = a.max_norm > 0.f ? fminf(a.max_norm / (norm + 1e-6f), 1.f) : 1.f;
            const StepCoef cf = a.coef[c.step];
            auto adam = [&](float g, float m, float v, int addr, float* m_out, float* v_out) {
                const float gq = g * scale;
                m = m + (gq - m) * a.omb1;
                v = v * a.beta2 + a.omb2 * gq * gq;
                const float denom = sqrtf(v) / cf.bc2_sqrt + a.adam_eps;
                lds[addr] = lds[addr] + (-cf.lr_step * m) / denom;
                *m_out = m; *v_out = v;
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float mo, vo;
                adam(tw2.g[r], tw2.m[r], tw2.v[r], lds_of_param(tile_flat(tw2, r, h)), &mo, &vo);
                tw2.m[r] = mo; tw2.v[r] = vo;
                adam(tx.g[r], tx.m[r], tx.v[r], lds_of_param(tile_flat(tx, r, h)), &mo, &vo);
                tx.m[r] = mo; tx.v[r] = vo;
            }
            if (bflat >= 0) adam(gb, mb, vb, lds_of_param(bflat), &mb, &vb);
            tw2.g = splat(0.f); tx.g = splat(0.f); gb = 0.f;
            if (wave == 0) {
                float v3[3] = {l_clip, l_vf, l_ent};
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v3[k] += __shfl_down(v3[k], off, 64);
                if (lane == 0) {
                    const float inv = 1.f / (float)c.batch;
                    const float clip = v3[0] * inv, vf = v3[1] * inv, ent = v3[2] * inv;
                    float* out = a.losses + 4 * (int64_t)c.step;
                    out[0] = clip + a.vf_coef * vf - a.ent_coef * ent;                   // ppo.py:211
                    out[1] = clip; out[2] = vf; out[3] = ent;
                }
                l_clip = l_vf = l_ent = 0.f;
            }
        }
        __syncthreads();
    }

    // ---- state back to HBM
    for (int p = tid; p < P_TOTAL; p += NT) a.params[p] = lds[lds_of_param(p)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a.m[tile_flat(tw2, r, h)] = tw2.m[r]; a.v[tile_flat(tw2, r, h)] = tw2.v[r];
        a.m[tile_flat(tx, r, h)] = tx.m[r]; a.v[tile_flat(tx, r, h)] = tx.v[r];
    }
    if (bflat >= 0) { a.m[bflat] = mb; a.v[bflat] = vb; }
}

// {mean, unbiased std} of every minibatch's advantages in float64 (ppo.py:184-186 through the host wrappers' float64
// statistics), one workgroup per gradient step
__global__ __launch_bounds__(256) void small_adv_stats_kernel(const float* __restrict__ adv, const int64_t* __restrict__ rows,
                                                              const long long* __restrict__ off, float* __restrict__ out) {
    __shared__ double red[2][4];
    const long long lo = off[blockIdx.x], hi = off[blockIdx.x + 1];
    double s1 = 0.0;
    for (long long k = lo + threadIdx.x; k < hi; k += 256) s1 += (double)adv[rows[k]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_down(s1, o, 64);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1;
    __syncthreads();
    const double n = (double)(hi - lo);
    const double mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
    double s2 = 0.0;
    for (long long k = lo + threadIdx.x; k < hi; k += 256) { const double d = (double)adv[rows[k]] - mean; s2 += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_down(s2, o, 64);
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (n - 1.0);
        out[2 * blockIdx.x] = (float)mean;
        out[2 * blockIdx.x + 1] = (float)sqrt(var);
    }
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace

extern "C" {

int ts_mlp_ppo_update_supported(int64_t obs_dim, int64_t hidden, int64_t n_act) {
    return obs_dim >= 1 && obs_dim <= K0 && hidden == HID && n_act >= 1 && n_act < HEAD;
}

int ts_mlp_ppo_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step0, int64_t obs_dim,
                      int64_t hidden, int64_t n_act, const float* obs, const int64_t* act, const float* adv,
                      const float* returns, const float* logp_old, const float* v_old, int64_t n, const int64_t* rows,
                      const int64_t* h_mb_offset, int64_t n_steps, const ts_ppo_hparams* hp, float* losses_out,
                      ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_mlp_ppo_update: workspace is NULL");
    TS_REQUIRE(ts_mlp_ppo_update_supported(obs_dim, hidden, n_act), TS_ERR_UNSUPPORTED,
               "ts_mlp_ppo_update: the one-launch update takes obs_dim <= 32, hidden == 64, n_act <= 31");
    TS_REQUIRE(n >= 1 && n_steps >= 0 && adam_step0 >= 0, TS_ERR_INVALID_ARG, "ts_mlp_ppo_update: bad size");
    if (n_steps == 0) return TS_OK;
    TS_REQUIRE(params && adam_m && adam_v && obs && act && adv && returns && rows && h_mb_offset && hp && losses_out,
               TS_ERR_INVALID_ARG, "ts_mlp_ppo_update: NULL argument");
    TS_REQUIRE(hp->algo == 0 || hp->algo == 1, TS_ERR_UNSUPPORTED, "ts_mlp_ppo_update: algo must be 0 (PPO) or 1 (A2C)");
    TS_REQUIRE(hp->algo == 1 || (logp_old && v_old), TS_ERR_INVALID_ARG, "ts_mlp_ppo_update: PPO needs logp_old and v_old");
    TS_REQUIRE(hp->lr >= 0.0, TS_ERR_INVALID_ARG, "ts_mlp_ppo_update: negative learning rate");
    std::vector<Chunk> chunks;
    std::vector<StepCoef> coef((size_t)n_steps);
    std::vector<long long> off((size_t)n_steps + 1);
    for (int64_t k = 0; k < n_steps; ++k) {
        const int64_t lo = h_mb_offset[k], hi = h_mb_offset[k + 1];
        TS_REQUIRE(hi > lo && lo >= 0, TS_ERR_SHAPE, "ts_mlp_ppo_update: minibatch %lld is empty", (long long)k);
        TS_REQUIRE(hi - lo < (1ll << 30), TS_ERR_SHAPE, "ts_mlp_ppo_update: minibatch too large");
        for (int64_t r = lo; r < hi; r += CH)
            chunks.push_back(Chunk{(long long)r, (int)std::min<int64_t>(CH, hi - r), (int)k, (int)(hi - lo), r + CH >= hi ? 1 : 0});
        const double t = (double)(adam_step0 + k + 1);
        coef[(size_t)k] = StepCoef{(float)(hp->lr / (1.0 - pow(hp->beta1, t))), (float)sqrt(1.0 - pow(hp->beta2, t))};
        off[(size_t)k] = lo;
    }
    off[(size_t)n_steps] = h_mb_offset[n_steps];
    const bool norm = hp->adv_norm && hp->algo == 0;
    const size_t b_chunks = al(sizeof(Chunk) * chunks.size()), b_coef = al(sizeof(StepCoef) * coef.size()),
                 b_off = al(sizeof(long long) * off.size()), b_stats = al(sizeof(float) * 2 * (size_t)n_steps);
    if (int rc = ts::ws_reserve(ws, b_chunks + b_coef + b_off + b_stats)) return rc;
    char* base = static_cast<char*>(ws->base);
    hipStream_t s = ts::as_stream(stream);
    // pageable sources: an event behind the copies is waited for before returning (the vectors go out of scope)
    TS_HIP_CHECK(hipMemcpyAsync(base, chunks.data(), sizeof(Chunk) * chunks.size(), hipMemcpyHostToDevice, s));
    TS_HIP_CHECK(hipMemcpyAsync(base + b_chunks, coef.data(), sizeof(StepCoef) * coef.size(), hipMemcpyHostToDevice, s));
    float* stats = nullptr;
    if (norm)
        TS_HIP_CHECK(hipMemcpyAsync(base + b_chunks + b_coef, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice, s));
    hipEvent_t copied;
    TS_HIP_CHECK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
    TS_HIP_CHECK(hipEventRecord(copied, s));
    if (norm) {
        stats = reinterpret_cast<float*>(base + b_chunks + b_coef + b_off);
        hipLaunchKernelGGL(small_adv_stats_kernel, dim3((unsigned)n_steps), dim3(256), 0, s, adv, rows,
                           reinterpret_cast<const long long*>(base + b_chunks + b_coef), stats);
        TS_LAUNCH_CHECK();
    }
    SmallArgs a{};
    a.params = params; a.m = adam_m; a.v = adam_v;
    a.obs = obs; a.act = act; a.adv = adv; a.ret = returns; a.logp_old = logp_old; a.v_old = v_old;
    a.rows = rows; a.chunks = reinterpret_cast<const Chunk*>(base); a.n_chunks = (int)chunks.size();
    a.adv_stats = stats; a.coef = reinterpret_cast<const StepCoef*>(base + b_chunks);
    a.obs_dim = (int)obs_dim; a.n_act = (int)n_act;
    a.eps_clip = (float)hp->eps_clip; a.dual_clip = (float)hp->dual_clip; a.vf_coef = (float)hp->vf_coef;
    a.ent_coef = (float)hp->ent_coef; a.value_clip = hp->value_clip; a.algo = hp->algo;
    a.max_norm = (float)(hp->max_grad_norm > 0.0 ? hp->max_grad_norm : 0.0);
    a.beta2 = (float)hp->beta2; a.adam_eps = (float)hp->adam_eps;
    a.omb1 = (float)(1.0 - hp->beta1); a.omb2 = (float)(1.0 - hp->beta2);
    a.losses = losses_out;
    const size_t lds = sizeof(float) * (size_t)L_END;
    static ts::DynLds attr;                      // per device (ts_common.h)
    if (int rc = attr.allow(reinterpret_cast<const void*>(&mlp_ppo_update_small_kernel), lds)) return rc;
    hipLaunchKernelGGL(mlp_ppo_update_small_kernel, dim3(1), dim3(NT), lds, s, a);
    TS_LAUNCH_CHECK();
    // the host vectors above are pageable stack objects: wait until the three table copies have read them (the event
    // sits right behind the copies, in front of the kernels -- the update itself stays asynchronous)
    (void)n;
    TS_HIP_CHECK(hipEventSynchronize(copied));
    TS_HIP_CHECK(hipEventDestroy(copied));
    return TS_OK;
}

}  // extern "C"
