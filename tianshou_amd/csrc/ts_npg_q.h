// ts_npg_q.h -- one-launch passes over the obs -> 64 -> 64 -> mu tanh Gaussian actor for NPG / TRPO on gfx950 (included by
// ts_ppo.hip behind ts_ppo_q.h, inside its anonymous namespace; called from ts_npg.hip through ts::npg_*_fused).
//
// Passes of NPG / TRPO's learn() on one (mini)batch, each ONE kernel over 32-sample tiles
// plus one small sum, where the per-layer path takes 8 - 20 GEMM / elementwise launches:
//   GRAD  the vanilla gradient of the surrogate (npg.py:152-158: -mean(logp adv); trpo.py:135-141: -mean(ratio adv)):
//         forward, per-sample loss, reverse pass.  Also leaves mu(theta) per sample for the KL evaluations.
//   FVP   one Fisher-vector product (NPG._MVP, npg.py:195-200: the mean KL differentiated twice).  At the expansion point
//         the KL's Hessian is the Gauss-Newton product  F v = (1/B) sum_b J_b^T diag(1 / sigma^2) J_b v + 2 v_sigma
//         (ts_npg.hip's header):
//           forward  H1 = tanh(W1 x + b1), H2 = tanh(W2 H1 + b2)                              (activations at theta)
//           tangent  T1 = (V1 x + vb1)(1 - H1^2), T2 = (W2 T1 + V2 H1 + vb2)(1 - H2^2), dmu = Wmu T2 + Vmu H2 + vbmu
//           u = dmu / sigma^2 / B;   reverse pass with u as the head gradient                  (J^T u)
//   CRITIC one critic iteration's gradient (npg.py:179-183: mse_loss(returns, V)): GRAD on the critic's vector, whose head
//         has one column.
//   INFER forward only: the head's outputs (an actor's mu, a critic's V -- the same 64-64 trunk with one head column) and
//         log pi(a | s) per sample: the network passes of NPG._preprocess_batch (npg.py:123-138).
//   EVAL  kl(old || candidate) and the surrogate at up to 32 candidate parameter vectors (npg.py:170-177's kl,
//         trpo.py:167-191's line search: every backtracking candidate in the same launch, blockIdx.y = candidate):
//         forward only, per-workgroup partial sums.
// Decomposition, LDS tiles, slab layout and the epilogue are those of the feature-split PPO step kernel (ts_ppo_q.h,
// q4::stepq_run<ACTOR>): workgroup = 4 waves = one 32-sample tile, wave w owns features [16 w, 16 w + 16) of every layer;
// the weights (FVP: theta's and the direction's) are MFMA A operands in registers; the tangents share the B operands
// (H1, x) the forward pass reads anyway.  v_mfma_f32_16x16x4_f32 per wave and tile: FVP 224, GRAD 142, EVAL 50.
//
// Parameter vectors use ts_npg.hip's block layout: L1 [k0 + 1, 64] | L2 [65, 64] | head [65, 32] | log_sigma [32]
// (rows = inputs, last row = bias, columns = outputs; k0 = obs rounded up to 32).  x is the zero-padded observation
// matrix [B, k0] of that file.

namespace q4 {

enum { NPG_FVP = 0, NPG_GRAD = 1, NPG_EVAL = 2, NPG_INFER = 3, NPG_CRITIC = 4 };

struct ActorArgs {
    const float* theta;       // parameters (EVAL: candidate c at theta + c * cand_stride)
    const float* dir;         // FVP: direction v (same layout)
    const float* x;           // [n_rows][k0] zero-padded observations
    int64_t n_rows;
    float inv_batch;
    float* slabs;             // FVP / GRAD: [gridDim.x][slab_w] (Slab3 actor sections | d log_sigma [8] | loss sum [4])
    int slab_w;               // EVAL: partial sums [gridDim.y][gridDim.x][2] = {kl, ratio adv}
    int obs, act, k0;
    const float* actions;     // GRAD / EVAL: [n_rows][act]
    const float* adv;         // [n_rows]
    const float* logp_old;    // [n_rows]; NULL: GRAD takes NPG's surrogate logp adv, EVAL skips the surrogate
    float* mu;                // [n_rows][8]: GRAD writes mu(theta), EVAL reads it (the old mean)
    const float* theta_old;   // EVAL: the old parameters (log_sigma)
    int64_t cand_stride;
    float* logp_out;          // INFER: [n_rows] log pi(actions) or NULL; mu (or NULL) receives the head's first `act` outputs
    int mu_stride;            //        per row at this stride (a critic: act = 1, stride 1 -> V)
};

template <int MODE>
struct LdsA {
    static constexpr int R1 = 0;                                               // sample-major H1; later u / dout and dZ2
    static constexpr int R1T = R1 + 32 * PS;                                   // FVP: sample-major T1 (tangent of H1)
    static constexpr int R2 = R1T + (MODE == NPG_FVP ? 32 * PS : 0);           // feature-major, rows private to the wave
    static constexpr int R3 = R2 + (MODE == NPG_EVAL || MODE == NPG_INFER ? 0 : HID * PF);          // feature-major H1
    static constexpr int PP = R3 + (MODE == NPG_EVAL || MODE == NPG_INFER ? 0 : HID * PF);          // head partials
    static constexpr int SM = PP + P_FLOATS;                                   // per-action constants
    static constexpr int BI = SM + 64;                                         // biases [4][64]: b1, b2, (FVP) vb1, vb2
    static constexpr int REC = BI + 4 * HID;                                   // [2][32][4 K1S] observation tiles
};

template <int MODE>
inline size_t actor_lds_bytes(int k1s) { return sizeof(float) * (size_t)(LdsA<MODE>::REC + 2 * 32 * 4 * k1s); }
inline int actor_slab_width(int k1s) { return slab3_layout(4 * k1s).sig + 12; }

#define TS_Q_LANE()                                  \
    int lane = lane0;                                \
    asm volatile("" : "+v"(lane));                   \
    [[maybe_unused]] const int tid = 64 * w + lane;  \
    [[maybe_unused]] const int n = lane & 15;        \
    [[maybe_unused]] const int gq = lane >> 4

// observation tile of 32 rows: 32 K1S float4 (<= 256: one per thread)
template <int K1S>
__device__ __forceinline__ f32x4 actor_fetch(const ActorArgs& g, int64_t tile, int tid) {
    const int qi = tid < 32 * K1S ? tid : 32 * K1S - 1;
    const int rec = qi / K1S, part = qi - rec * K1S;
    int64_t row = tile * 32 + rec;
    row = row < g.n_rows ? row : g.n_rows - 1;
    return *reinterpret_cast<const f32x4*>(g.x + row * g.k0 + part * 4);
}

// per-sample inputs of the loss (GRAD / EVAL): lane (sample 16 b + n, actions gq and 4 + gq)
struct SampleIn { float act0[2], act1[2], adv[2], lpo[2], mu0[2], mu1[2]; };

template <int K1S, int MODE>
__device__ __forceinline__ void actor_run(const ActorArgs& g, float* lds) {
    using L = LdsA<MODE>;
    constexpr bool FVP = MODE == NPG_FVP, GRAD = MODE == NPG_GRAD, EVAL = MODE == NPG_EVAL, INFER = MODE == NPG_INFER;
    constexpr bool CRITIC = MODE == NPG_CRITIC;            // GRAD with the critic's loss: mse_loss(returns, V), one head column
    constexpr bool FWD = EVAL || INFER;                    // forward only: no reverse pass, no gradient tiles
    constexpr int NB1 = (4 * K1S + 15) / 16;
    constexpr int RW = 4 * K1S;                            // floats per observation record in LDS
    const Slab3 SL = slab3_layout(4 * K1S);
    const int lane0 = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fb = 16 * w;
    const int obs = g.obs, n_act = g.act;
    const int p = blockIdx.x, n_wg = gridDim.x;
    const int o2 = (g.k0 + 1) * HID, o3 = o2 + (HID + 1) * HID, o_b1 = g.k0 * HID, o_b2 = o2 + HID * HID;
    const int o_bmu = o3 + HID * 32, o_sig = o3 + (HID + 1) * 32;
    const cgfloat_ptr th = (cgfloat_ptr)(g.theta + (EVAL ? (int64_t)blockIdx.y * g.cand_stride : 0));
    [[maybe_unused]] const cgfloat_ptr dv = (cgfloat_ptr)g.dir;
    float* R1 = lds + L::R1;
    [[maybe_unused]] float* R1T = lds + L::R1T;
    [[maybe_unused]] float* R2 = lds + L::R2;
    [[maybe_unused]] float* R3 = lds + L::R3;
    float* PP = lds + L::PP;
    float* SM = lds + L::SM;
    float* BIA = lds + L::BI;
    float* REC = lds + L::REC;
    [[maybe_unused]] float* slab = g.slabs + (int64_t)p * g.slab_w;
    const int64_t n_tiles = (g.n_rows + 31) / 32;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool ratio_mode = g.logp_old != nullptr;

    // ---- resident MFMA A operands (lane = (row m = n, k = gq)): theta's and (FVP) the direction's
    float W1a[K1S];                               // W1[fb + n][4 j + gq]
    [[maybe_unused]] float V1a[K1S];
    float W2f[16];                                // W2[fb + n][16 jj + 4 gq + r]   (forward form)
    [[maybe_unused]] float V2f[16];
    [[maybe_unused]] float W2t[16];               // W2[16 jj + 4 gq + r][fb + n]   (backward form)
    float WH[4];                                  // Wmu[n][fb + 4 gq + r] (rows >= act: 0)
    [[maybe_unused]] float VH[4];
    [[maybe_unused]] float WHb[2];                // Wmu[4 r + gq][fb + n]
    // (the biases -- initial accumulators of the lane's rows -- come back from LDS every tile: 16 registers fewer)
    {
        TS_Q_LANE();
#pragma unroll
        for (int jr = 0; jr < 16; ++jr) {
            const int f = 16 * (jr >> 2) + 4 * gq + (jr & 3);
            W2f[jr] = th[o2 + f * HID + fb + n];
            if constexpr (FVP) V2f[jr] = dv[o2 + f * HID + fb + n];
            if constexpr (!FWD) W2t[jr] = th[o2 + (fb + n) * HID + f];
        }
#pragma unroll
        for (int j = 0; j < K1S; ++j) {
            const int k = 4 * j + gq, kc = k < obs ? k : 0;
            const float a = th[kc * HID + fb + n];
            W1a[j] = k < obs ? a : 0.f;
            if constexpr (FVP) {
                const float b = dv[kc * HID + fb + n];
                V1a[j] = k < obs ? b : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = fb + 4 * gq + r;
            const int a = n < n_act ? n : 0;
            const float x0 = th[o3 + f * 32 + a];
            WH[r] = n < n_act ? x0 : 0.f;
            if constexpr (FVP) {
                const float x1 = dv[o3 + f * 32 + a];
                VH[r] = n < n_act ? x1 : 0.f;
            }
        }
        if constexpr (!FWD) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int a = 4 * r + gq;
                const float x0 = th[o3 + (fb + n) * 32 + (a < n_act ? a : 0)];
                WHb[r] = a < n_act ? x0 : 0.f;
            }
        }
        if (tid < HID) {
            BIA[tid] = th[o_b1 + tid];
            BIA[HID + tid] = th[o_b2 + tid];
            if constexpr (FVP) {
                BIA[2 * HID + tid] = dv[o_b1 + tid];
                BIA[3 * HID + tid] = dv[o_b2 + tid];
            }
        }
        // per-action constants (torch: sigma = exp(log_sigma); Normal.log_prob uses var = sigma^2 and log(sigma))
        if (tid < 8) {
            const int a = tid;
            const bool live = a < n_act;
            const float ls = (live && !CRITIC && (!INFER || g.actions)) ? th[o_sig + a] : 0.f;    // (a critic's vector has no log_sigma block)
            const float sigma = expf(ls), var = sigma * sigma;
            if constexpr (FVP) {
                SM[a] = live ? dv[o_bmu + a] : 0.f;                    // vbmu
                SM[8 + a] = 1.f / var;
            } else {
                SM[a] = live ? th[o_bmu + a] : 0.f;                    // bmu
                SM[8 + a] = 1.f / (2.f * var);
                SM[16 + a] = logf(sigma);
                SM[24 + a] = 1.f / var;
            }
            if constexpr (EVAL) {                                      // kl.py _kl_normal_normal, the per-action constants
                const float so = expf(live ? g.theta_old[o_sig + a] : 0.f);
                const float q = so / sigma, var_ratio = q * q;
                SM[32 + a] = 1.f / sigma;
                SM[40 + a] = live ? var_ratio - 1.f - logf(var_ratio) : 0.f;
            }
        }
        const f32x4 f0 = actor_fetch<K1S>(g, p, tid);
        if (tid < 32 * K1S) st4(REC + 4 * tid, f0);
    }

    // ---- persistent accumulators (MFMA C layout: lane (col n, group gq) register r = row 4 gq + r)
    [[maybe_unused]] f32x4 gW2[4], gW1[NB1];
    if constexpr (!FWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) gW2[c] = zero4;
#pragma unroll
        for (int c = 0; c < NB1; ++c) gW1[c] = zero4;
    }
    [[maybe_unused]] f32x4 gH = zero4;            // dWmu[4 gq + r][fb + n]
    [[maybe_unused]] float rs = 0.f, rs1 = 0.f;   // lane-partials of db2[fb + n], db1[fb + n]
    float sD0 = 0.f, sD1 = 0.f;                   // head-bias partial sums   (EVAL: kl / surrogate partial sums)
    [[maybe_unused]] float sS0 = 0.f, sS1 = 0.f, sL = 0.f;    // GRAD: d log_sigma / loss partial sums
    __syncthreads();
    int cur = 0;

    for (int64_t t = p; t < n_tiles; t += n_wg) {
        const int64_t t_next = t + n_wg;
        const bool has_next = t_next < n_tiles;          // uniform
        const float* RC = REC + cur * 32 * RW;
        float* RN = REC + (cur ^ 1) * 32 * RW;
        [[maybe_unused]] f32x4 h1[2];
        f32x4 fnext;
        [[maybe_unused]] SampleIn in;

        // ================= phase 1: H1 = tanh(W1 x + b1) [, T1 = (V1 x + vb1)(1 - H1^2)]: own 16 features x 32 samples
        {
            TS_Q_LANE();
            fnext = actor_fetch<K1S>(g, has_next ? t_next : t, tid);     // next tile's observations: in flight all tile long
            if constexpr (!FVP) {                                        // this tile's loss inputs: in flight until phase 3
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int64_t row = t * 32 + 16 * b + n;
                    row = row < g.n_rows ? row : g.n_rows - 1;
                    const int a0 = gq < n_act ? gq : 0, a1 = 4 + gq < n_act ? 4 + gq : 0;
                    if constexpr (CRITIC) {
                        in.adv[b] = g.adv[row];                          // the return of the sample
                    } else if constexpr (INFER) {
                        in.act0[b] = g.actions ? g.actions[row * n_act + a0] : 0.f;
                        in.act1[b] = g.actions ? g.actions[row * n_act + a1] : 0.f;
                    } else {
                        in.act0[b] = g.actions[row * n_act + a0];
                        in.act1[b] = g.actions[row * n_act + a1];
                        in.adv[b] = g.adv[row];
                        in.lpo[b] = ratio_mode ? g.logp_old[row] : 0.f;
                    }
                    if constexpr (EVAL) {
                        in.mu0[b] = g.mu[row * ACT_PAD + gq];
                        in.mu1[b] = g.mu[row * ACT_PAD + 4 + gq];
                    }
                }
            }
            float xv[2][K1S];
#pragma unroll
            for (int j = 0; j < K1S; ++j) {
#pragma unroll
                for (int b = 0; b < 2; ++b) xv[b][j] = RC[(16 * b + n) * RW + 4 * j + gq];
            }
            f32x4 acc[2];
            acc[0] = acc[1] = ld4(BIA + fb + 4 * gq);
            [[maybe_unused]] f32x4 tac[2];
            if constexpr (FVP) tac[0] = tac[1] = ld4(BIA + 2 * HID + fb + 4 * gq);
#pragma unroll
            for (int j = 0; j < K1S; ++j) {
                acc[0] = mfma16(W1a[j], xv[0][j], acc[0]);
                acc[1] = mfma16(W1a[j], xv[1][j], acc[1]);
                if constexpr (FVP) {
                    tac[0] = mfma16(V1a[j], xv[0][j], tac[0]);
                    tac[1] = mfma16(V1a[j], xv[1][j], tac[1]);
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tanh4(acc[b]);
                st4(R1 + (16 * b + n) * PS + fb + 4 * gq, acc[b]);
                if constexpr (FVP) {
                    dtanh4(tac[b], acc[b]);
                    st4(R1T + (16 * b + n) * PS + fb + 4 * gq, tac[b]);
                }
                if constexpr (!FWD) {
                    h1[b] = acc[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) R3[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][r];
                }
            }
        }
        __syncthreads();                                 // B1: H1 / T1 tiles complete

        // ================= phase 2: H2 [, T2]; head partials
        [[maybe_unused]] f32x4 h2[2];
        {
            TS_Q_LANE();
            f32x4 acc[2];
            acc[0] = acc[1] = ld4(BIA + HID + fb + 4 * gq);
            [[maybe_unused]] f32x4 tac[2];
            if constexpr (FVP) tac[0] = tac[1] = ld4(BIA + 3 * HID + fb + 4 * gq);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b0 = ld4(R1 + n * PS + 16 * jj + 4 * gq);
                const f32x4 b1 = ld4(R1 + (16 + n) * PS + 16 * jj + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = mfma16(W2f[4 * jj + r], b0[r], acc[0]);
                    acc[1] = mfma16(W2f[4 * jj + r], b1[r], acc[1]);
                    if constexpr (FVP) {
                        tac[0] = mfma16(V2f[4 * jj + r], b0[r], tac[0]);
                        tac[1] = mfma16(V2f[4 * jj + r], b1[r], tac[1]);
                    }
                }
                if constexpr (FVP) {
                    const f32x4 c0 = ld4(R1T + n * PS + 16 * jj + 4 * gq);
                    const f32x4 c1 = ld4(R1T + (16 + n) * PS + 16 * jj + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tac[0] = mfma16(W2f[4 * jj + r], c0[r], tac[0]);
                        tac[1] = mfma16(W2f[4 * jj + r], c1[r], tac[1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // bounds the operand-read hoisting (register pressure)
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tanh4(acc[b]);
                if constexpr (!FWD) {
                    h2[b] = acc[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) R2[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][r];   // for the head gradient
                }
                f32x4 pm = zero4;                        // rows = actions 4 gq + r
                if constexpr (FVP) {
                    dtanh4(tac[b], acc[b]);
                    f32x4 pn = zero4;                    // two chains
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pm = mfma16(WH[r], tac[b][r], pm);
                        pn = mfma16(VH[r], acc[b][r], pn);
                    }
                    pm = pm + pn;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pm = mfma16(WH[r], acc[b][r], pm);
                }
                if (gq < 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) PP[((w * 2 + b) * 8 + 4 * gq + r) * 16 + n] = pm[r];
                }
            }
        }
        __syncthreads();                                 // B2: head partials complete; R1 (H1) and R1T are free

        // ================= phase 3: the head gradient u (FVP) / the loss and its head gradient (GRAD) / kl and surrogate (EVAL)
        {
            TS_Q_LANE();
            const int a0 = gq, a1 = 4 + gq;
            const float bm0 = SM[a0], bm1 = SM[a1], iv0 = SM[8 + a0], iv1 = SM[8 + a1];
            [[maybe_unused]] float u0[2], u1[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int s = 16 * b + n;
                float m0 = bm0, m1 = bm1;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    m0 += PP[((ww * 2 + b) * 8 + a0) * 16 + n];
                    m1 += PP[((ww * 2 + b) * 8 + a1) * 16 + n];
                }
                const bool valid = t * 32 + s < g.n_rows;
                const float wgt = valid ? g.inv_batch : 0.f;
                if constexpr (FVP) {
                    u0[b] = m0 * iv0 * wgt;              // padding actions: zero weights and bias -> dmu = 0
                    u1[b] = m1 * iv1 * wgt;
                    sD0 += u0[b];
                    sD1 += u1[b];
                } else if constexpr (CRITIC) {
                    // npg.py:180: mse_loss(returns, V) -- lane (sample, action 0) holds V
                    const float td = m0 - in.adv[b];
                    u0[b] = a0 == 0 ? 2.f * td * wgt : 0.f;
                    u1[b] = 0.f;
                    sD0 += u0[b];
                    sL += (valid && gq == 0) ? td * td : 0.f;
                } else {
                    const float ls0 = SM[16 + a0], ls1 = SM[16 + a1];
                    const float c0 = a0 < n_act ? LOG_SQRT_2PI : 0.f, c1 = a1 < n_act ? LOG_SQRT_2PI : 0.f;
                    const float d0 = a0 < n_act ? in.act0[b] - m0 : 0.f, d1 = a1 < n_act ? in.act1[b] - m1 : 0.f;
                    // Normal.log_prob summed over the action dimension (padding actions contribute an exact 0)
                    float logp = (-(d0 * d0) * iv0 - ls0 - c0) + (-(d1 * d1) * iv1 - ls1 - c1);
                    logp = group4_sum(logp);
                    [[maybe_unused]] float ratio = 1.f, term = 0.f;
                    if constexpr (!INFER) {
                        ratio = ratio_mode ? expf(logp - in.lpo[b]) : 1.f;
                        term = ratio_mode ? ratio * in.adv[b] : logp * in.adv[b];
                    }
                    if constexpr (INFER) {
                        if (valid && w == 0) {
                            const int64_t row = t * 32 + s;
                            if (g.mu) {
                                if (a0 < n_act) g.mu[row * g.mu_stride + a0] = m0;
                                if (a1 < n_act) g.mu[row * g.mu_stride + a1] = m1;
                            }
                            if (g.logp_out && gq == 0) g.logp_out[row] = logp;
                        }
                    } else if constexpr (GRAD) {
                        const float dlogp = -in.adv[b] * ratio * wgt;
                        const float v0 = SM[24 + a0], v1 = SM[24 + a1];
                        u0[b] = dlogp * d0 * v0;
                        u1[b] = dlogp * d1 * v1;
                        const float ds0 = a0 < n_act ? dlogp * (d0 * d0 * v0 - 1.f) : 0.f;
                        const float ds1 = a1 < n_act ? dlogp * (d1 * d1 * v1 - 1.f) : 0.f;
                        sD0 += u0[b]; sD1 += u1[b]; sS0 += ds0; sS1 += ds1;
                        sL += (valid && gq == 0) ? term : 0.f;
                        if (valid && w == 0) {
                            g.mu[(t * 32 + s) * ACT_PAD + a0] = m0;
                            g.mu[(t * 32 + s) * ACT_PAD + a1] = m1;
                        }
                    } else {
                        // kl(old || new) of this sample's two actions (kl.py _kl_normal_normal): 0.5 (var_ratio + t1 - 1 - log var_ratio)
                        const float e0 = (in.mu0[b] - m0) * SM[32 + a0], e1 = (in.mu1[b] - m1) * SM[32 + a1];
                        const float k0v = a0 < n_act ? 0.5f * (SM[40 + a0] + e0 * e0) : 0.f;
                        const float k1v = a1 < n_act ? 0.5f * (SM[40 + a1] + e1 * e1) : 0.f;
                        sD0 += valid ? k0v + k1v : 0.f;                                   // every lane: its own two actions
                        sD1 += (valid && gq == 0 && ratio_mode) ? term : 0.f;             // one lane per sample
                    }
                }
                if constexpr (!FWD) {
                    R1[s * PS + fb + a0] = u0[b];        // sample-major, own columns (A operand of the head gradient)
                    R1[s * PS + fb + a1] = u1[b];
                }
            }
            if constexpr (!FWD) {
                f32x4 dz2[2];
                wave_lds_sync();
                {
                    const f32x4 bv0 = ld4(R2 + (fb + n) * PF + 4 * gq), bv1 = ld4(R2 + (fb + n) * PF + 16 + 4 * gq);
                    float av0[4], av1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        av0[r] = R1[(4 * gq + r) * PS + fb + (n & 7)];
                        av1[r] = R1[(16 + 4 * gq + r) * PS + fb + (n & 7)];
                    }
                    f32x4 g1 = zero4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gH = mfma16(av0[r], bv0[r], gH);
                        g1 = mfma16(av1[r], bv1[r], g1);
                    }
                    gH = gH + g1;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 dh = mfma16(WHb[0], u0[b], zero4);
                    dh = mfma16(WHb[1], u1[b], dh);
                    dtanh4(dh, h2[b]);
                    dz2[b] = dh;
                }
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    st4(R1 + (16 * b + n) * PS + fb + 4 * gq, dz2[b]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) R2[(fb + 4 * gq + r) * PF + 16 * b + n] = dz2[b][r];
                }
            } else {
                if (has_next && tid < 32 * K1S) st4(RN + 4 * tid, fnext);
            }
        }
        __syncthreads();                                 // B3: dZ2 (sample-major) complete   (EVAL: B0 of the next tile)

        // ================= phase 4: dZ1, weight gradients
        if constexpr (!FWD) {
            TS_Q_LANE();
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b0 = ld4(R1 + n * PS + 16 * jj + 4 * gq);
                const f32x4 b1 = ld4(R1 + (16 + n) * PS + 16 * jj + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = mfma16(W2t[4 * jj + r], b0[r], acc[0]);
                    acc[1] = mfma16(W2t[4 * jj + r], b1[r], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) dtanh4(acc[b], h1[b]);
            // dW2[f2 own][f1] += sum_s dZ2[s][f2] H1[s][f1];  db2[f2] += sum_s dZ2[s][f2]
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const f32x4 av = ld4(R2 + (fb + n) * PF + 16 * J + 4 * gq);
                rs += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 bv = ld4(R3 + (16 * c + n) * PF + 16 * J + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) gW2[c] = mfma16(av[r], bv[r], gW2[c]);
                    if (c & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // dZ1, feature-major, over the wave's own rows of R2 (its dZ2 rows have just been consumed)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) R2[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][r];
            }
            wave_lds_sync();
            // dW1[f1 own][k] += sum_s dZ1[s][f1] x[s][k];  db1[f1] += sum_s dZ1[s][f1]
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const f32x4 av = ld4(R2 + (fb + n) * PF + 16 * J + 4 * gq);
                rs1 += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int c = 0; c < NB1; ++c) {
                    const int k = 16 * c + n < RW ? 16 * c + n : RW - 1;
                    const float* xp = RC + (16 * J + 4 * gq) * RW + k;
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = xp[r * RW];
#pragma unroll
                    for (int r = 0; r < 4; ++r) gW1[c] = mfma16(av[r], bv[r], gW1[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_next && tid < 32 * K1S) st4(RN + 4 * tid, fnext);
            __syncthreads();                             // B0 of the next tile
        }
        cur ^= 1;
    }

    // ---- epilogue: the workgroup's sums leave once
    TS_Q_LANE();
    if constexpr (INFER) {
    } else if constexpr (EVAL) {
        // every wave sees every sample: wave 0's sums are the workgroup's (fixed order: 16 lanes of a row on DPP, then the
        // four rows)
        sD0 = group4_sum(row16_sum(sD0));
        sD1 = group4_sum(row16_sum(sD1));
        if (w == 0 && lane == 0) {
            float* out = g.slabs + ((int64_t)blockIdx.y * n_wg + p) * 2;
            out[0] = sD0;
            out[1] = sD1;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) slab_st4(slab + SL.w2t[0] + (16 * c + n) * HID + fb + 4 * gq, gW2[c]);
#pragma unroll
        for (int c = 0; c < NB1; ++c)
            if (16 * c + n < 4 * K1S) slab_st4(slab + SL.w1t[0] + (16 * c + n) * HID + fb + 4 * gq, gW1[c]);
        rs = group4_sum(rs);
        rs1 = group4_sum(rs1);
        if (gq == 0) {
            slab_st(slab + SL.b2[0] + fb + n, rs);
            slab_st(slab + SL.b1[0] + fb + n, rs1);
        }
        if (gq < 2) slab_st4(slab + SL.head[0] + (fb + n) * ACT_PAD + 4 * gq, gH);
        sD0 = row16_sum(sD0);
        sD1 = row16_sum(sD1);
        if (w == 0 && n == 0) {
            slab_st(slab + SL.hb[0] + gq, sD0);
            slab_st(slab + SL.hb[0] + 4 + gq, sD1);
        }
        if constexpr (GRAD || CRITIC) {
            // every wave sees every sample: the log_sigma and loss sums come from wave 0 alone
            sS0 = row16_sum(sS0);
            sS1 = row16_sum(sS1);
            sL = row16_sum(sL);
            if (w == 0 && n == 0) {
                slab_st(slab + SL.sig + gq, sS0);
                slab_st(slab + SL.sig + 4 + gq, sS1);
                if (gq == 0) slab_st(slab + SL.sig + 8, sL);
            }
        }
    }
}
#undef TS_Q_LANE

template <int K1S>
__global__ __launch_bounds__(QT, 3) void npg_fvp_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    actor_run<K1S, NPG_FVP>(g, lds);
}

template <int K1S>
__global__ __launch_bounds__(QT, 3) void npg_grad_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    actor_run<K1S, NPG_GRAD>(g, lds);
}

template <int K1S>
__global__ __launch_bounds__(QT, 3) void npg_critic_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    actor_run<K1S, NPG_CRITIC>(g, lds);
}

template <int K1S>
__global__ __launch_bounds__(QT, 4) void npg_eval_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    actor_run<K1S, NPG_EVAL>(g, lds);
}

template <int K1S>
__global__ __launch_bounds__(QT, 4) void npg_infer_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    actor_run<K1S, NPG_INFER>(g, lds);
}

// out[i] = sum over slabs of the column that holds parameter i (fixed order).  One workgroup = 64 parameters x 16 slab
// groups.  v != NULL (FVP): + damping v[i], and the log-sigma block gets the exact 2 v_s of the KL's Hessian (ts_npg.hip:
// fvp_finish_kernel).  v == NULL (GRAD / CRITIC): the log-sigma block (if P holds one) takes its slab columns and
// loss_out[0] = loss_sign (loss sum) / B.
__global__ __launch_bounds__(1024) void npg_actor_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int slab_w, int obs,
                                                                int act, int k0, int k1, const float* __restrict__ v,
                                                                float* __restrict__ out, int P, float damping,
                                                                float* __restrict__ loss_out, float n_rows, float loss_sign) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const Slab3 SL = slab3_layout(k1);
    const int o2 = (k0 + 1) * HID, o3 = o2 + (HID + 1) * HID, o_sig = o3 + (HID + 1) * 32;
    int col = -1;
    if (i < o2) {
        const int k = i >> 6, f = i & 63;
        if (k < obs) col = SL.w1t[0] + k * HID + f;
        else if (k == k0) col = SL.b1[0] + f;
    } else if (i < o3) {
        const int c = i - o2, k = c >> 6, f = c & 63;
        col = k < HID ? SL.w2t[0] + k * HID + f : SL.b2[0] + f;
    } else if (i < o_sig) {
        const int c = i - o3, f = c >> 5, a = c & 31;
        if (a < act) col = f < HID ? SL.head[0] + f * ACT_PAD + a : SL.hb[0] + a;
    } else if (!v) {
        if (i < P) { if (i - o_sig < act) col = SL.sig + (i - o_sig); }
        else if (i == P) col = SL.sig + 8;               // the loss sum rides behind the parameters
    }
    float s = 0.f;
    if (col >= 0) {
#pragma unroll 8
        for (int k = wave; k < n_slabs; k += 16) s += slabs[(int64_t)k * slab_w + col];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        if (v) {
            if (i < P) {
                const float vi = v[i];
                float r = t + vi * damping;
                if (i >= o_sig) r = (i - o_sig < act ? 2.f * vi : 0.f) + vi * damping;
                out[i] = r;
            }
        } else {
            if (i < P) out[i] = t;
            else if (i == P) loss_out[0] = loss_sign * (t / n_rows);
        }
    }
}

// res[2 c + {0, 1}] = {mean kl, -mean(ratio adv)} of candidate c from the EVAL partial sums [n_cand][n_wg][2]
// NPG (one candidate, always taken -- npg.py:170-177): theta <- the candidate, stats[1] = its kl, stats[2] = 0 in the same launch
struct NpgApply { float* theta; const float* cand; int P; float* stats3; };
__global__ __launch_bounds__(256) void npg_eval_finish_kernel(const float* __restrict__ partial, int n_wg, float n_rows,
                                                              float* __restrict__ res, NpgApply ap) {
    __shared__ float red[4][2];
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < n_wg; i += 256) {
        s0 += partial[((int64_t)c * n_wg + i) * 2];
        s1 += partial[((int64_t)c * n_wg + i) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0; red[threadIdx.x >> 6][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const float v = threadIdx.x == 0 ? t / n_rows : -(t / n_rows);
        res[2 * c + threadIdx.x] = v;
        if (ap.theta && threadIdx.x == 0) { ap.stats3[1] = v; ap.stats3[2] = 0.f; }
    }
    if (ap.theta)
        for (int i = threadIdx.x; i < ap.P; i += 256) ap.theta[i] = ap.cand[i];
}

}  // namespace q4
