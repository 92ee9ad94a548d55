// ts_ppo_step4.h -- fourth-generation PPO step kernel: the split-bf16 arithmetic of ts_ppo_step3.h (every fp32 operand
// as three bf16 pieces, six products, fp32 accumulate) on 16-sample tiles with FOUR waves per SIMD.
// Included by ts_ppo.hip inside its anonymous namespace, after ts_ppo_step3.h.
//
// Why a different shape: with the GEMMs on the bf16 matrix cores the step kernel is no longer bound by a pipe but by the
// dependent chain of one wave (forward -> loss -> backward -> weight gradients); two waves per SIMD (256 VGPRs each)
// cannot hide it.  v_mfma_f32_16x16x32_bf16 keeps a 64-feature activation of 16 samples in 16 registers
// instead of 32 for 32 samples, so a wave fits 128 VGPRs and a 1,024-thread workgroup (16 waves = 4 per SIMD, one
// workgroup per CU, 256 samples) shares ONE weight image -- both orientations of W2 resident, no gathers.
//
// Lane roles: n = lane & 15 is the wave's sample, gq = lane >> 4 the k-slot group.  C/D tile t of a 64-feature array
// (f32x4): register r of lane (n, gq) is feature 16 t + 4 gq + r.  The k-slot (gq, j) of K = 32 chunk c is DEFINED as
// feature 32 c + 16 (j >> 2) + 4 gq + (j & 3): tiles 2c, 2c + 1 of the producing layer, pairwise converted in the lane.
// Layer 1: k-slot (gq, j) = input column 8 gq + j (obs | 1 | 0, at most 32 columns).

namespace s4 {

using s3::P3;
using s3::Pk3;
using s3::bf16x8;
using s3::cvt_pk;
using s3::lo_f32;
using s3::split_pair;
using s3::u16;
using s3::u32x4;

constexpr int THREADS = 1024, WAVES = 16, SPW = 16;
constexpr int CHUNK = 1024;                         // one operand chunk: 64 lanes x 16 B
constexpr int W2_PIECE = 8 * CHUNK;                 // [tile 4][chunk 2]
constexpr int W2_BYTES = 3 * W2_PIECE;
// image (offsets inside one net's image; in LDS it sits BEHIND the wave scratch areas, with the backward copy of W2 last,
// so that the phase-A gradient tiles -- which overlay scratch + the forward part -- leave it intact for dH1)
constexpr int W2F_OFF = 0;                          // forward: rows = f2, slots = f1
constexpr int W1_OFF = W2_BYTES;
constexpr int W1_PIECE = 4 * CHUNK;                 // [tile 4], one K = 32 chunk
constexpr int W1_BYTES = 3 * W1_PIECE;
constexpr int F32_OFF = W1_OFF + W1_BYTES;          // fp32 part: b2[64] | head image [gq][t][r][8] | SMALL[32]
constexpr int B2_F = 0, WH_F = 64, SMALL_F = 64 + 512;
constexpr int F32_FLOATS = 64 + 512 + 32;
constexpr int W2B_OFF = F32_OFF + 4 * F32_FLOATS;   // backward: rows = f1, slots = f2
constexpr int IMG_BYTES = W2B_OFF + W2_BYTES;
static_assert(IMG_BYTES % 16 == 0 && W2B_OFF % 16 == 0, "image alignment");
constexpr int QP = 20;                              // pitch (floats) of the per-wave [row][16 samples] scratch tiles
constexpr int SCR_FLOATS = 64 * QP + SPW * ACT_PAD; // H2^T tile + the dout broadcast rows (also: parked records, Qt)
constexpr int IMG_LDS = WAVES * SCR_FLOATS * 4;     // LDS offset of the image
constexpr int LDS_BYTES = IMG_LDS + IMG_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
// weight-gradient tiles (overlay image + scratch): [feature][128 samples] bf16, three pieces, two rounds of 8 waves
constexpr int TP = 272;                             // 128 samples x 2 B + 16: conflict-free ds_read_b128
constexpr int TA_ROWS = 128, TA_PIECE = TA_ROWS * TP;
constexpr int TB_ROWS = 96, TB_PIECE = TB_ROWS * TP;
constexpr int TB_MISC = 3 * TB_PIECE;
static_assert(3 * TA_PIECE <= IMG_LDS + W2B_OFF && TB_MISC + WAVES * MISC_SLOT * 4 <= IMG_LDS + W2B_OFF && TB_MISC % 16 == 0,
              "the gradient tiles must end below the backward image");

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mma6(const P3& a, const P3& b, f32x4 c) {
    c = mfma16(a.p[2], b.p[0], c);
    c = mfma16(a.p[1], b.p[1], c);
    c = mfma16(a.p[0], b.p[2], c);
    c = mfma16(a.p[1], b.p[0], c);
    c = mfma16(a.p[0], b.p[1], c);
    c = mfma16(a.p[0], b.p[0], c);
    return c;
}

// tiles 2c, 2c + 1 of a 64-feature array -> the B / A operand pieces of chunk c
__device__ __forceinline__ P3 split_chunk(const f32x4& lo, const f32x4& hi) {
    P3 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4& v = q < 2 ? lo : hi;
        const Pk3 s = split_pair(v[2 * (q & 1)], v[2 * (q & 1) + 1]);
        o.p[0][q] = s.p0; o.p[1][q] = s.p1; o.p[2][q] = s.p2;
    }
    return o;
}

// ---- image: parameter -> slot(s).  code = kind << 24 | byte offset in [2][IMG_BYTES]; kind 1: bf16 piece 0 of a W2
// element (pieces W2_PIECE apart), 2: W1aug element (W1_PIECE apart), 3: fp32 slot, 0: none
__device__ __forceinline__ int hid_slot(int f, int& chunk) {       // feature f as a k-slot: chunk, byte offset (gq, j)
    const int fi = f & 31;
    chunk = f >> 5;
    const int gq = (fi & 15) >> 2, j = ((fi >> 4) << 2) | (fi & 3);
    return gq * 256 + j * 2;
}

__device__ __forceinline__ int param_code(int p, const Dims& d, int* code_t) {
    *code_t = 0;
    const int net = p >= d.p_actor;
    const int base = net * IMG_BYTES;
    const int w1 = net ? d.c_w1 : d.a_w1, b1 = net ? d.c_b1 : d.a_b1, w2 = net ? d.c_w2 : d.a_w2, b2 = net ? d.c_b2 : d.a_b2;
    if (p < w2) {                                   // W1[row][k] | b1[row] as column k = obs
        int row, k;
        if (p < b1) { const int q = p - w1; row = q / d.obs; k = q - row * d.obs; } else { row = p - b1; k = d.obs; }
        const int off = W1_OFF + (row >> 4) * CHUNK + ((row & 15) + 16 * (k >> 3)) * 16 + (k & 7) * 2;
        return (2 << 24) | (base + off);
    }
    if (p < b2) {                                   // W2[f2][f1]: forward and backward images
        const int q = p - w2, f2 = q >> 6, f1 = q & 63;
        int c1, c2;
        const int in1 = hid_slot(f1, c1), in2 = hid_slot(f2, c2);
        *code_t = (1 << 24) | (base + W2B_OFF + ((f1 >> 4) * 2 + c2) * CHUNK + (f1 & 15) * 16 + in2);
        return (1 << 24) | (base + W2F_OFF + ((f2 >> 4) * 2 + c1) * CHUNK + (f2 & 15) * 16 + in1);
    }
    if (p < b2 + HID) return (3 << 24) | (base + F32_OFF + 4 * (B2_F + (p - b2)));
    const int q = p - (b2 + HID);                   // head W | head b | sigma
    const int n_head = net ? 1 : d.act;
    if (q < n_head * HID) {
        const int a = q / HID, f = q - a * HID;
        const int t = f >> 4, gq = (f & 15) >> 2, r = f & 3;
        return (3 << 24) | (base + F32_OFF + 4 * (WH_F + ((gq * 4 + t) * 4 + r) * ACT_PAD + a));
    }
    const int e = q - n_head * HID;
    if (e < n_head) return (3 << 24) | (base + F32_OFF + 4 * (SMALL_F + (net ? 24 : e)));
    return 0;
}

__device__ __forceinline__ void image_put(char* image, int code, float v) {
    const int kind = code >> 24, off = code & 0xffffff;
    if (kind == 3) {
        *reinterpret_cast<float*>(image + off) = v;
    } else if (kind != 0) {
        const int stride = kind == 1 ? W2_PIECE : W1_PIECE;
        const unsigned p0 = cvt_pk(v, 0.f);
        const float r1 = v - lo_f32(p0);
        const unsigned p1 = cvt_pk(r1, 0.f);
        const float r2 = r1 - lo_f32(p1);
        const unsigned p2 = cvt_pk(r2, 0.f);
        *reinterpret_cast<u16*>(image + off) = (u16)p0;
        *reinterpret_cast<u16*>(image + off + stride) = (u16)p1;
        *reinterpret_cast<u16*>(image + off + 2 * stride) = (u16)p2;
    }
}

__device__ __forceinline__ void image_put_sigma(char* image, int k, float sigma_param) {
    const float sigma = expf(sigma_param);
    float* sm = reinterpret_cast<float*>(image + F32_OFF) + SMALL_F;
    sm[8 + k] = 1.f / (2.f * (sigma * sigma));
    sm[16 + k] = logf(sigma);
}

// one workgroup: zero both images, then every parameter writes its slot(s); inv[p] = code, inv[p_total + p] = W2 backward slot
__global__ __launch_bounds__(1024) void ppo_build_image4_kernel(const float* __restrict__ params, Dims d,
                                                                char* __restrict__ image, int* __restrict__ inv) {
    for (int i = threadIdx.x; i < 2 * IMG_BYTES / 4; i += 1024) reinterpret_cast<int*>(image)[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < d.p_total; p += 1024) {
        int code_t;
        const int code = param_code(p, d, &code_t);
        if (inv) { inv[p] = code; inv[d.p_total + p] = code_t; }
        const float v = params[p];
        image_put(image, code, v);
        image_put(image, code_t, v);
        const int k = p - d.a_sig;
        if (k >= 0 && k < d.act) image_put_sigma(image, k, v);
    }
}

__device__ __forceinline__ void stage_image4(char* lds, const char* __restrict__ img, int tid) {
    constexpr int N4 = IMG_BYTES / 16, PER = (N4 + THREADS - 1) / THREADS;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + THREADS * k;
        v[k] = reinterpret_cast<const f32x4*>(img)[q < N4 ? q : N4 - 1];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + THREADS * k;
        if (q < N4) reinterpret_cast<f32x4*>(lds)[q] = v[k];
    }
}

struct In4 {
    P3 xp;                       // layer-1 B operand: input columns 8 gq + j
    float act[ACT_PAD];
    float adv, logp_old, ret, v_old;
    float w;
};

// forward, loss and backward of one net for the wave's 16 samples
template <bool ACTOR>
__device__ __forceinline__ void net_fwd_bwd4(const char* L, float* scratch, const StepArgs& g, const Dims& d, const In4& in,
                                             int lane_in, f32x4 (&h1)[4], P3 (&dz2p)[2],
                                             float (&gw)[ACTOR ? ACT_PAD : 1], float& misc) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    constexpr int NA = ACTOR ? ACT_PAD : 1;
    constexpr int MK = ACTOR ? 2 : 10;
    const int n = lane & 15, gq = lane >> 4;
    const float* f32t = reinterpret_cast<const float*>(L + F32_OFF);
    f32x4 h2[4];
    {
        const char* w1 = L + W1_OFF + lane * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            P3 a;
#pragma unroll
            for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(w1 + p * W1_PIECE + t * CHUNK);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma6(a, in.xp, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fast_tanh(acc[r]);
            h1[t] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
        P3 h1p[2];
        h1p[0] = split_chunk(h1[0], h1[1]);
        h1p[1] = split_chunk(h1[2], h1[3]);
        const char* w2 = L + W2F_OFF + lane * 16;
        const float* b2 = f32t + B2_F;
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(b2 + 16 * t2 + 4 * gq);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                P3 a;
#pragma unroll
                for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(w2 + p * W2_PIECE + (t2 * 2 + c) * CHUNK);
                acc = mma6(a, h1p[c], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fast_tanh(acc[r]);
            h2[t2] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    TS_MARK(g, MK + 0);
    // ---- heads on the VALU: the lane's 16 features, then the four k-slot groups are summed
    const float* wh = f32t + WH_F + gq * (4 * 4 * ACT_PAD);
    float hout[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) hout[a] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* p = wh + (t * 4 + r) * ACT_PAD;
            if constexpr (ACTOR) {
                const f32x4 w0 = ld4(p), w1v = ld4(p + 4);
#pragma unroll
                for (int a = 0; a < 4; ++a) { hout[a] += h2[t][r] * w0[a]; hout[a + 4] += h2[t][r] * w1v[a]; }
            } else {
                hout[0] += h2[t][r] * p[0];
            }
            if (ACTOR && (r & 1)) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        hout[a] += __shfl_xor(hout[a], 16, 64);
        hout[a] += __shfl_xor(hout[a], 32, 64);
    }

    float dout[NA];
    const float w = in.w;
    const float* sm = f32t + SMALL_F;
    float* Qt = scratch;                         // [17][QP]: per-sample quantities, transposed, for the row sums
    if constexpr (ACTOR) {
        float dlt[ACT_PAD];
        float logp = 0.f;
        int n_act = d.act;
        asm volatile("" : "+s"(n_act));
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            // padding actions (k >= act): zero weights, bias 0, sigma_param 0 -> the term is an exact -0
            const float bm = sm[k], iv = sm[8 + k], ls = sm[16 + k];
            const float m = hout[k] + bm;
            dlt[k] = in.act[k] - m;
            logp += -(dlt[k] * dlt[k]) * iv - ls - (k < n_act ? LOG_SQRT_2PI : 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        float mean = 0.f, den = 1.f;
        if (g.adv_norm) { mean = g.adv_stats[0]; den = g.adv_stats[1] + 1e-8f; }
        const float A = (in.adv - mean) / den;
        const bool a2c = g.a2c != 0;
        const float ratio = a2c ? 1.f : expf(logp - in.logp_old);
        const float surr1 = ratio * A;
        const float lo = 1.f - g.eps_clip, hi = 1.f + g.eps_clip;
        const float surr2 = fminf(fmaxf(ratio, lo), hi) * A;
        const float clip1 = fminf(surr1, surr2);
        float basek = (surr1 <= surr2) ? A : 0.f;
        const float dA = g.dual_clip * A;
        const bool dual = (g.dual_clip > 0.f) && (A < 0.f);
        float term = dual ? -fmaxf(clip1, dA) : -clip1;
        basek = (dual && !(clip1 >= dA)) ? 0.f : basek;
        term = a2c ? -logp * A : term;
        basek = a2c ? A : basek;
        const float dlogp = -basek * ratio * w;
        const float ent_w = g.ent_coef * w;
        // rows 0..7 dout (written by group 0), 8..15 dsig (group 1), 16 loss term (group 2)
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            const float inv_var = 2.f * sm[8 + k];
            dout[k] = dlogp * dlt[k] * inv_var;
            const float ds = dlogp * (dlt[k] * dlt[k] * inv_var - 1.f) - ent_w;
            const float dsig = k < n_act ? ds : 0.f;
            if (gq < 2) Qt[(8 * gq + k) * QP + n] = gq ? dsig : dout[k];
        }
        if (gq == 2) Qt[16 * QP + n] = term * w;
        __builtin_amdgcn_sched_barrier(0);
    } else {
        const float value = hout[0] + sm[24];
        const float ret = in.ret;
        const float vf1 = (ret - value) * (ret - value);
        const float vo = in.v_old;
        const float dvo = value - vo;
        const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
        const float vf2 = (ret - vclip) * (ret - vclip);
        const float g1 = -2.f * (ret - value);
        const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
        const float dv_clip = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        const bool vc = g.value_clip != 0;
        const float term = vc ? fmaxf(vf1, vf2) : vf1;
        const float dv = vc ? dv_clip : g1;
        dout[0] = dv * g.vf_coef * w;
        if (gq == 0) Qt[n] = dout[0];
        else if (gq == 2) Qt[16 * QP + n] = term * w;
    }
    wave_lds_sync();
    {
        const int row = lane < 16 ? lane : 16;
        const float* rp = Qt + row * QP;
        const f32x4 q0 = ld4(rp), q1 = ld4(rp + 4), q2 = ld4(rp + 8), q3 = ld4(rp + 12);
        const float sacc = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3])) +
                           ((q2[0] + q2[1]) + (q2[2] + q2[3])) + ((q3[0] + q3[1]) + (q3[2] + q3[3]));
        const bool mine = ACTOR ? (lane <= 16) : (lane == 0 || lane == 16);
        misc = mine ? sacc : 0.f;
    }
    wave_lds_sync();

    TS_MARK(g, MK + 1);
    // ---- head weight gradient gw[a][f] = sum_s dout[s][a] H2[s][f]   (lane = feature f)
    float* HT = scratch;                         // [64][QP]
    float* DO = scratch + 64 * QP;               // [16][8]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) HT[(16 * t + 4 * gq + r) * QP + n] = h2[t][r];
    if (gq == 0) {
        if constexpr (ACTOR) {
            *reinterpret_cast<f32x4*>(DO + n * ACT_PAD) = f32x4{dout[0], dout[1], dout[2], dout[3]};
            *reinterpret_cast<f32x4*>(DO + n * ACT_PAD + 4) = f32x4{dout[4], dout[5], dout[6], dout[7]};
        } else {
            DO[n * ACT_PAD] = dout[0];
        }
    }
    wave_lds_sync();
    {
        const float* rowp = HT + lane * QP;
#pragma unroll
        for (int a = 0; a < NA; ++a) gw[a] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 hv = ld4(rowp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int smp = 4 * q + e;
                if constexpr (ACTOR) {
                    const f32x4 d0 = ld4(DO + smp * ACT_PAD), d1 = ld4(DO + smp * ACT_PAD + 4);     // uniform addresses
#pragma unroll
                    for (int a = 0; a < 4; ++a) { gw[a] += d0[a] * hv[e]; gw[a + 4] += d1[a] * hv[e]; }
                } else {
                    gw[0] += DO[smp * ACT_PAD] * hv[e];
                }
                if (ACTOR && (e & 1)) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wave_lds_sync();

    TS_MARK(g, MK + 2);
    // ---- dZ2 = (dout . Whead) (1 - h2^2), in place
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* p = wh + (t * 4 + r) * ACT_PAD;
            float dh;
            if constexpr (ACTOR) {
                const f32x4 w0 = ld4(p), w1v = ld4(p + 4);
                dh = dout[0] * w0[0] + dout[1] * w0[1] + dout[2] * w0[2] + dout[3] * w0[3] +
                     dout[4] * w1v[0] + dout[5] * w1v[1] + dout[6] * w1v[2] + dout[7] * w1v[3];
            } else {
                dh = dout[0] * p[0];
            }
            const float hv = h2[t][r];
            h2[t][r] = dh * (1.f - hv * hv);
            if ((r & 1) || !ACTOR) __builtin_amdgcn_sched_barrier(0);
        }
    dz2p[0] = split_chunk(h2[0], h2[1]);
    dz2p[1] = split_chunk(h2[2], h2[3]);

    TS_MARK(g, MK + 3);
}

// dH1^T = W2^T dZ2^T (backward image: still intact after the phase-A tiles), dZ1 = dH1 (1 - h1^2)
__device__ __forceinline__ void dh1_backward4(const char* img, const P3 (&dz2p)[2], const f32x4 (&h1)[4], int lane,
                                              f32x4 (&dz1)[4]) {
    const char* w2b = img + W2B_OFF + lane * 16;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            P3 a;
#pragma unroll
            for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(w2b + p * W2_PIECE + (t1 * 2 + c) * CHUNK);
            acc = mma6(a, dz2p[c], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float hv = h1[t1][r]; acc[r] = acc[r] * (1.f - hv * hv); }
        dz1[t1] = acc;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one chunk of the lane's column -> rows row0 + feature (+1) of every piece
__device__ __forceinline__ void tile_put4(char* T, int piece_stride, int row0, int c, const P3& v, int col, int gq) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 32 * c + 16 * (q >> 1) + 4 * gq + 2 * (q & 1);
            char* a = T + p * piece_stride + (row0 + f) * TP + col * 2;
            const unsigned w = v.p[p][q];
            *reinterpret_cast<u16*>(a) = (u16)w;
            *reinterpret_cast<u16*>(a + TP) = (u16)(w >> 16);
        }
}

__device__ __forceinline__ void tile_get4(const char* T, int piece_stride, int row, int c, int gq, P3& out) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
        out.p[p] = *reinterpret_cast<const u32x4*>(T + p * piece_stride + row * TP + (32 * c + 8 * gq) * 2);
}

template <int KS1, bool ACTOR>
__device__ __forceinline__ void net_wgrad4(char* L, const StepArgs& g, const Dims& d, const In4& in, const f32x4 (&h1)[4],
                                           const P3 (&dz2p)[2], const float (&gw)[ACTOR ? ACT_PAD : 1],
                                           float misc, int wave, int lane_in, float* slab, const Slab2& SL, bool first) {
    constexpr int NA = ACTOR ? ACT_PAD : 1;
    constexpr int MK = ACTOR ? 2 : 10;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int KP = 2 * KS1;
    int lane = lane_in;
    asm volatile("" : "+v"(lane), "+v"(wave), "+s"(slab));
    const int n = lane & 15, gq = lane >> 4;
    const int my_round = wave >> 3, col = 16 * (wave & 7) + n;
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // ---- phase A: dW2[f2][f1] = sum_s dZ2[s][f2] H1[s][f1]: wave (tM, tN) owns a 16 x 16 tile; db2 through an all-ones B
    const int tM = wave >> 2, tN = wave & 3;
    f32x4 c = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                       // B1: every wave is done with the weight image and its scratch area
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (my_round == round) {
            tile_put4(L, TA_PIECE, 0, 0, dz2p[0], col, gq);
            tile_put4(L, TA_PIECE, 0, 1, dz2p[1], col, gq);
            tile_put4(L, TA_PIECE, 64, 0, split_chunk(h1[0], h1[1]), col, gq);
            tile_put4(L, TA_PIECE, 64, 1, split_chunk(h1[2], h1[3]), col, gq);
        }
        if (ACTOR) TS_MARK(g, 20 + 3 * round);
        __syncthreads();
        if (ACTOR) TS_MARK(g, 21 + 3 * round);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            P3 a, b;
            tile_get4(L, TA_PIECE, 16 * tM + n, cc, gq, a);
            tile_get4(L, TA_PIECE, 64 + 16 * tN + n, cc, gq, b);
            c = mma6(a, b, c);
            if (tN == 0) {
                cb = mfma16(a.p[2], ones, cb);
                cb = mfma16(a.p[1], ones, cb);
                cb = mfma16(a.p[0], ones, cb);
            }
        }
        if (ACTOR) TS_MARK(g, 22 + 3 * round);
        __syncthreads();
    }
    TS_MARK(g, MK + 4);
    {
        float* p = slab + SL.w2[net] + (16 * tM + 4 * gq) * HID + 16 * tN + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = first ? c[r] : p[r * HID] + c[r];
            slab_st(p + r * HID, v);
        }
        if (tN == 0 && n == 0) {
            float* pb = slab + SL.b2[net] + 16 * tM + 4 * gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) store_acc(pb + r, cb[r], first);
        }
    }

    f32x4 dz1[4];
    dh1_backward4(L + IMG_LDS, dz2p, h1, lane, dz1);
    TS_MARK(g, MK + 5);
    // ---- phase B: dW1aug[f1][k] = sum_s dZ1[s][f1] Xaug[s][k] on eight waves (16 x 16 tiles), the small rows on the others
    float* M = reinterpret_cast<float*>(L + TB_MISC);
    {
        float* Mw = M + wave * MISC_SLOT;
#pragma unroll
        for (int a = 0; a < NA; ++a) Mw[a * 64 + lane] = gw[a];
        Mw[8 * 64 + lane] = misc;
    }
    const bool w1_wave = ACTOR ? (wave < 8) : (wave >= 8);
    const int v8 = wave & 7, tM1 = v8 >> 1, tK = v8 & 1;
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (my_round == round) {
            tile_put4(L, TB_PIECE, 0, 0, split_chunk(dz1[0], dz1[1]), col, gq);
            tile_put4(L, TB_PIECE, 0, 1, split_chunk(dz1[2], dz1[3]), col, gq);
            // X^T: k-slot (gq, j) is input column 8 gq + j -> row 64 + that
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    char* a = L + p * TB_PIECE + (64 + 8 * gq + 2 * q) * TP + col * 2;
                    const unsigned w = in.xp.p[p][q];
                    *reinterpret_cast<u16*>(a) = (u16)w;
                    *reinterpret_cast<u16*>(a + TP) = (u16)(w >> 16);
                }
        }
        __syncthreads();
        if (w1_wave) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                P3 a, b;
                tile_get4(L, TB_PIECE, 16 * tM1 + n, cc, gq, a);
                tile_get4(L, TB_PIECE, 64 + 16 * tK + n, cc, gq, b);
                c1 = mma6(a, b, c1);
            }
        } else if (round == 0) {
            // the eight other waves sum the small rows over the 16 wave slots: wave v8 takes head row v8, wave 0 also the misc row
            const int n_head = ACTOR ? d.act : 1;
            if (v8 < n_head) {
                float v = 0.f;
#pragma unroll
                for (int sl = 0; sl < WAVES; ++sl) v += M[sl * MISC_SLOT + v8 * 64 + lane];
                store_acc(slab + SL.head[net] + v8 * HID + lane, v, first);
            }
            if (v8 == 7) {
                float v = 0.f;
#pragma unroll
                for (int sl = 0; sl < WAVES; ++sl) v += M[sl * MISC_SLOT + 8 * 64 + lane];
                if (lane < 8) { if (lane < n_head) store_acc(slab + SL.hb[net] + lane, v, first); }
                else if (lane < 16) { if (ACTOR && lane - 8 < d.act) store_acc(slab + SL.sig + lane - 8, v, first); }
                else if (lane == 16) store_acc(slab + SL.loss + net, v, first);
            }
        }
        if (round == 0) __syncthreads();
    }
    if (w1_wave) {
        const int k = 16 * tK + n;
        if (k < KP) {
            float* p = slab + SL.w1[net] + (16 * tM1 + 4 * gq) * KP + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = first ? c1[r] : p[r * KP] + c1[r];
                slab_st(p + r * KP, v);
            }
        }
    }
    TS_MARK(g, MK + 6);
}

template <int KS1>
__global__ __launch_bounds__(THREADS) void ppo_step4_kernel(StepArgs g, Dims d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* L = reinterpret_cast<char*>(lds);
    const char* image = reinterpret_cast<const char*>(g.image);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, gq = lane >> 4;
    const Slab2 SL = slab2_layout(d.act, 2 * KS1);
    float* slab = g.slabs + (int64_t)blockIdx.x * g.slab_w;
    float* scratch = reinterpret_cast<float*>(L) + wave * SCR_FLOATS;
    const bool first = g.accumulate == 0;

    TS_MARK(g, 0);
    // row id of the lane's sample -> record gather (16 records per wave, 16-byte pieces over the 64 lanes)
    const int64_t srow = g.row0 + ((int64_t)blockIdx.x * WAVES + wave) * SPW + n;
    const bool valid = srow < g.n_rows;
    const int64_t pos = valid ? srow : g.n_rows - 1;
    const int64_t rid = g.rows ? g.rows[pos] : pos;
    const int parts = g.rec_w >> 2;
    const int total = SPW * parts;
    constexpr int REC_FETCH = (SPW * ((2 * KS1 + 14) / 4) + 63) / 64;
    f32x4 rv[REC_FETCH];
    {
        const int lo = (int)(rid & 0xffffffffLL), hi = (int)(rid >> 32);
#pragma unroll
        for (int k = 0; k < REC_FETCH; ++k) {
            int q = lane + 64 * k;
            q = q < total ? q : total - 1;
            const int rec = q / parts, part = q - rec * parts;
            const int64_t r = ((int64_t)__shfl(hi, rec, 64) << 32) | (uint32_t)__shfl(lo, rec, 64);
            rv[k] = *reinterpret_cast<const f32x4*>(g.rec + r * g.rec_w + part * 4);
        }
    }
    stage_image4(L + IMG_LDS, image, threadIdx.x);
    In4 in;
    {
#pragma unroll
        for (int k = 0; k < REC_FETCH; ++k) {
            const int q = lane + 64 * k;
            if (q < total) *reinterpret_cast<f32x4*>(scratch + q * 4) = rv[k];
        }
        wave_lds_sync();
        const float* r = scratch + n * g.rec_w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 8 * gq + 2 * q + e;
                const float x = r[k < d.obs ? k : 0];
                v[e] = k < d.obs ? x : (k == d.obs ? 1.f : 0.f);
            }
            const Pk3 s = split_pair(v[0], v[1]);
            in.xp.p[0][q] = s.p0; in.xp.p[1][q] = s.p1; in.xp.p[2][q] = s.p2;
        }
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) in.act[k] = (k < d.act) ? r[d.obs + k] : 0.f;
        const float* aux = r + d.obs + d.act;
        in.adv = aux[0];
        in.ret = aux[1];
        in.logp_old = aux[2];
        in.v_old = aux[3];
        in.w = valid ? g.inv_batch : 0.f;
        wave_lds_sync();
    }
    __syncthreads();
    TS_MARK(g, 1);
    f32x4 h1[4];
    P3 dz2p[2];
    float misc;
    {
        float gw[ACT_PAD];
        net_fwd_bwd4<true>(L + IMG_LDS, scratch, g, d, in, lane, h1, dz2p, gw, misc);
        net_wgrad4<KS1, true>(L, g, d, in, h1, dz2p, gw, misc, wave, lane, slab, SL, first);
    }
    __syncthreads();                      // phase-B readers of the actor are done: everything below the backward image is free
    TS_MARK(g, 18);
    stage_image4(L + IMG_LDS, image + IMG_BYTES, threadIdx.x);
    __syncthreads();
    TS_MARK(g, 9);
    {
        float gw[1];
        net_fwd_bwd4<false>(L + IMG_LDS, scratch, g, d, in, lane, h1, dz2p, gw, misc);
        net_wgrad4<KS1, false>(L, g, d, in, h1, dz2p, gw, misc, wave, lane, slab, SL, first);
    }
    TS_MARK(g, 17);
}

}  // namespace s4
