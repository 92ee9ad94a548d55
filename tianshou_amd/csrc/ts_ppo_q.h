// ts_ppo_q.h -- feature-split PPO/A2C step kernel for gfx950 (included by ts_ppo.hip, inside its anonymous namespace).
//
// Same mathematics as ppo_step2_kernel (one minibatch iteration of PPO._update_with_batch, ppo.py:179-216, for the
// obs -> 64 -> 64 -> {mu, V} tanh actor-critic), different decomposition -- built for the strong-scaling regime
// (BASELINE configs[3]: 65,536 / N rows per rank) and for more waves per SIMD at N = 1:
//
//   * one workgroup = 4 waves = ONE 32-sample tile of ONE network.  Wave w owns output features [16 w, 16 w + 16) of
//     every layer of that network: 16x16x4 fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles), two sample blocks of 16
//     = two independent accumulators per layer (the 40-cycle dependent latency is covered).  A wave's dependent
//     matrix work per tile is 142 (actor) / 122 (critic) MFMAs of 32 cycles instead of 242 of 64.
//   * the wave's weights (its rows of W1 | b1, W2, its columns of W2 for the backward pass, its slice of the head) are
//     MFMA A operands held in REGISTERS for the life of the workgroup, loaded straight from the flat parameter vector
//     (L2-resident): no LDS weight image, no staging prologue.
//   * activations cross waves through LDS tiles in two layouts: sample-major [32][68] (B operand of the next layer:
//     one ds_read_b128 = four k-steps) and feature-major [64][36] (both operands of the weight gradients, which
//     contract over samples).  Four workgroup barriers per tile.
//   * workgroups are persistent over tiles (tile = p, p + P, ...): weight gradients accumulate in MFMA accumulators
//     across tiles and leave once, as 16-byte write-through stores into the pair's slab (slab p: actor columns from the
//     actor workgroup, critic columns from the critic workgroup).  The next tile's records are fetched during the
//     current tile's last phase.
//   * actor head (<= 8 outputs) on MFMA rows 0..7; the loss runs on one (sample, 2 actions) pair per lane;
//     critic head on the VALU.
//   * <= 128 VGPRs, <= 40 KB LDS: four workgroups (16 waves) per CU.
//
// Slab layout ("Slab3", floats): gradients are stored TRANSPOSED where that makes the accumulator's four consecutive
// rows contiguous in memory (dW2^T [f1][f2], dW1aug^T [k][f1], actor head [f][8]); ppo_reduce_slabs_kernel maps slab
// columns back to flat parameter indices with slab3_col_to_param.

namespace q4 {

constexpr int QT = 256;          // threads per workgroup (4 waves)
constexpr int PS = 68;           // sample-major tile pitch  [32][PS]   (ds_read_b128 of 4 consecutive features)
#ifndef TS_Q_PF
#define TS_Q_PF 40
#endif
constexpr int PF = TS_Q_PF;      // feature-major tile pitch [64][PF].  36 / 40 / 44 / 52 measured alike (profiles/r05_lds_pitch_sweep.txt):
                                 // the bank-conflict cycles the counters show (0.3 of the LDS cycles) are not this pitch's
constexpr int P_FLOATS = 4 * 2 * 8 * 16;   // head partials [wave][block][action][16]

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct Slab3 {
    int w2t[2], w1t[2], b1[2], b2[2], head[2], hb[2], sig, loss, width;
};

// k1 = 4 * K1S (layer-1 contraction length: obs rounded up, zero weights beyond obs)
__host__ __device__ inline Slab3 slab3_layout(int k1) {
    Slab3 L;
    int o = 0;
    for (int n = 0; n < 2; ++n) {
        L.w2t[n] = o; o += HID * HID;           // [f1][f2]
        L.w1t[n] = o; o += k1 * HID;            // [k][f1]  (columns k >= obs: products with the record's other fields, unused)
        L.b1[n] = o; o += HID;
        L.b2[n] = o; o += HID;
        L.head[n] = o; o += n ? HID : HID * ACT_PAD;   // actor [f][8], critic [f]
        L.hb[n] = o; o += n ? 4 : ACT_PAD;
        if (n == 0) { L.sig = o; o += ACT_PAD; }
    }
    L.loss = o; o += 4;                          // [0] clip-loss sum, [1] vf-loss sum
    L.width = o;
    return L;
}

// slab column -> flat parameter index (>= p_total: the two loss sums; -1: padding)
__device__ __forceinline__ int slab3_col_to_param(int col, const Dims& d, int k1) {
    const Slab3 L = slab3_layout(k1);
    if (col >= L.loss) return col < L.loss + N_EXTRA ? d.p_total + (col - L.loss) : -1;
    const int n = col >= L.w2t[1];
    int c = col - L.w2t[n];
    if (c < HID * HID) return (n ? d.c_w2 : d.a_w2) + (c & 63) * HID + (c >> 6);       // [f1][f2] -> W2[f2][f1]
    c -= HID * HID;
    if (c < k1 * HID) {
        const int k = c >> 6, f = c & 63;
        return k < d.obs ? (n ? d.c_w1 : d.a_w1) + f * d.obs + k : -1;
    }
    c -= k1 * HID;
    if (c < HID) return (n ? d.c_b1 : d.a_b1) + c;
    c -= HID;
    if (c < HID) return (n ? d.c_b2 : d.a_b2) + c;
    c -= HID;
    if (n == 0) {
        if (c < HID * ACT_PAD) { const int f = c >> 3, a = c & 7; return a < d.act ? d.a_wmu + a * HID + f : -1; }
        c -= HID * ACT_PAD;
        if (c < ACT_PAD) return c < d.act ? d.a_bmu + c : -1;
        c -= ACT_PAD;
        return c < d.act ? d.a_sig + c : -1;
    }
    if (c < HID) return d.c_wv + c;
    c -= HID;
    return c == 0 ? d.c_bv : -1;
}

// LDS carve (floats).  REC (the tile's packed records, [2][32][rec_w], double-buffered) comes last: its size is a run-time
// value.
struct LdsQ {
    static constexpr int R1 = 0;                       // sample-major: H1, later dout (own columns) and dZ2
    static constexpr int R2 = R1 + 32 * PS;            // feature-major, rows private to the owning wave: H2, dZ2, dZ1
    static constexpr int R3 = R2 + HID * PF;           // feature-major H1 (read by every wave in the weight gradients)
    static constexpr int PP = R3 + HID * PF;           // head partials
    static constexpr int SM = PP + P_FLOATS;           // [0..7] bmu, [8..15] 1/(2 sigma^2), [16..23] log sigma, [24] bv
    static constexpr int REC = SM + 32;
};

inline size_t stepq_lds_bytes(int rec_w) { return sizeof(float) * (size_t)(LdsQ::REC + 2 * 32 * rec_w); }

__device__ __forceinline__ void tanh4(f32x4& a) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 x = {a[r], a[r + 1]};
        const f32x2 y = x * 2.885390081777927f;
        f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
        e = e + 1.f;
        const f32x2 qq = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
        const f32x2 t = 1.f - 2.f * qq;
        a[r] = t[0];
        a[r + 1] = t[1];
    }
}

// v = v * (1 - h^2)
__device__ __forceinline__ void dtanh4(f32x4& v, const f32x4& hh) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 hv = {hh[r], hh[r + 1]};
        const f32x2 gg = {v[r], v[r + 1]};
        const f32x2 o = gg * (1.f - hv * hv);
        v[r] = o[0];
        v[r + 1] = o[1];
    }
}

__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// 16-byte write-through store (global_store_dwordx4 ... sc1): the slabs are read by the reduction kernel right behind
// the boundary; left dirty in the eight L2s they would be written back there
__device__ __forceinline__ void slab_st4(float* p, const f32x4& v) {
    // s_nop 1: gfx940+ needs two wait states between a store of more than 8 bytes and a VALU write of its data registers;
    // the hazard recognizer does not look inside inline asm, and the accumulators stored here are dead behind the store --
    // LLVM reuses them at once (observed: the next store's address arithmetic landed in the previous store's data)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((gfloat_ptr)p), "v"(v) : "memory");
}

// sum over the 16 lanes of a row (lanes that differ in bits 0..3) on DPP: xor 1, xor 2 as quad permutes; row_half_mirror and
// row_mirror act as xor 4 / xor 8 once the value is uniform within quads / half rows.  (__shfl_xor is ds_bpermute_b32: an LDS
// round trip per step.)
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int y;
    switch (ctrl_sel) {
        case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true); break;      // quad_perm [1,0,3,2]
        case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true); break;      // quad_perm [2,3,0,1]
        case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true); break;     // row_half_mirror
        default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true); break;    // row_mirror
    }
    return v + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add(v, 0);
    v = dpp_add(v, 1);
    v = dpp_add(v, 2);
    v = dpp_add(v, 3);
    return v;
}
// sum over the four lane groups (lanes that differ in bits 4..5): gfx950's v_permlane16_swap / v_permlane32_swap exchange
// rows (16 lanes) / halves between two registers -- with both holding v, the sum of the two results is the xor-16 / xor-32
// butterfly.  Inline asm: the builtins' second result is mis-selected by this compiler (both extracts read the first
// register); s_nop 1 covers the VALU-write -> permlane-swap read hazard on either side.
__device__ __forceinline__ float group4_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = a + b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int K1S>
struct RecQ {
    static constexpr int N = K1S > 5 ? 2 : 1;          // float4 loads per thread: 32 * rec_w / 4 <= 256 for obs <= 19
    f32x4 v[N];
};

template <int K1S>
__device__ __forceinline__ RecQ<K1S> recq_fetch(const StepArgs& g, int64_t row_id, int tid, int lane) {
    RecQ<K1S> f;
    const int parts = g.rec_w >> 2, total = 32 * parts;
    const int lo = (int)(row_id & 0xffffffffLL), hi = (int)(row_id >> 32);
#pragma unroll
    for (int k = 0; k < RecQ<K1S>::N; ++k) {
        int qi = tid + QT * k;
        qi = qi < total ? qi : total - 1;                 // unconditional (clamped) load
        const int rec = qi / parts, part = qi - rec * parts;
        const int64_t rid = ((int64_t)__shfl(hi, rec, 64) << 32) | (uint32_t)__shfl(lo, rec, 64);
        f.v[k] = *reinterpret_cast<const f32x4*>(g.rec + rid * g.rec_w + part * 4);
    }
    return f;
}

template <int K1S>
__device__ __forceinline__ void recq_commit(const RecQ<K1S>& f, const StepArgs& g, float* rec, int tid) {
    const int total = 8 * g.rec_w;
#pragma unroll
    for (int k = 0; k < RecQ<K1S>::N; ++k) {
        const int qi = tid + QT * k;
        if (qi < total) st4(rec + 4 * qi, f.v[k]);
    }
}

// row id of sample (lane & 31) of a tile (every wave holds all 32 ids, lanes i and i + 32 the same one)
__device__ __forceinline__ int64_t rowq_fetch(const StepArgs& g, int64_t tile, int lane) {
    const int64_t srow = tile * 32 + (lane & 31);
    const int64_t pos = srow < g.n_rows ? srow : g.n_rows - 1;
    return g.rows ? g.rows[pos] : pos;
}

using cgfloat_ptr = const __attribute__((address_space(1))) float*;

// Lane-derived indices are re-derived from an opaque copy of the lane id at the top of every phase: LLVM otherwise hoists
// some forty loop-invariant LDS / global addresses out of the tile loop and keeps them in VGPRs across it, which pushes
// the kernel from <= 128 registers to 160+ (3 waves per SIMD) or into scratch.
#define TS_Q_LANE()                                  \
    int lane = lane0;                                \
    asm volatile("" : "+v"(lane));                   \
    [[maybe_unused]] const int tid = 64 * w + lane;  \
    [[maybe_unused]] const int n = lane & 15;        \
    [[maybe_unused]] const int gq = lane >> 4

// BIG: the 256-register build (two workgroups per CU): both operand forms of W2 and H1 stay in registers and the next
// tile's records are fetched two phases earlier; !BIG: the 128-register build (four workgroups per CU).
// -DTS_PHASE_MARKS builds (scripts/gpu_stepq_phases.py): shader-clock stamps of wave 0 of pair 0's two workgroups
// (dbg[64 net + k]: k = 0 entry, 1 first barrier, then 2 + 4 tile + phase behind each phase's barrier) and the 100 MHz
// start / end stamps of every workgroup (dbg[128 + 2 b], dbg[129 + 2 b]).
#ifdef TS_PHASE_MARKS
#define TS_QMARK(k)                                                                                          \
    do {                                                                                                     \
        if (g.dbg && p == 0 && threadIdx.x == 0 && (k) < 64) g.dbg[64 * net + (k)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#define TS_QREAL(which)                                                                                      \
    do {                                                                                                     \
        if (g.dbg && threadIdx.x == 0 && blockIdx.x < 960)                                                   \
            g.dbg[128 + 2 * blockIdx.x + (which)] = (long long)__builtin_amdgcn_s_memrealtime();             \
    } while (0)
#else
#define TS_QMARK(k) do { } while (0)
#define TS_QREAL(which) do { } while (0)
#endif

template <int K1S, bool ACTOR, bool BIG>
__device__ __forceinline__ void stepq_run(const StepArgs& g, const Dims& d, float* lds, int p, int n_pairs, float* slab,
                                          const Slab3& SL, bool zero_other) {
    using L = LdsQ;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int NB1 = (4 * K1S + 15) / 16;               // 16-column blocks of dW1^T
    const int lane0 = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fb = 16 * w;     // wave index: scalar
    const int obs = d.obs, n_act = d.act, rec_w = g.rec_w;
    const cgfloat_ptr prm = (cgfloat_ptr)g.params;
    const int o_w1 = ACTOR ? d.a_w1 : d.c_w1, o_b1 = ACTOR ? d.a_b1 : d.c_b1;
    const int o_w2 = ACTOR ? d.a_w2 : d.c_w2, o_b2 = ACTOR ? d.a_b2 : d.c_b2;
    float* R1 = lds + L::R1;
    float* R2 = lds + L::R2;
    float* R3 = lds + L::R3;
    float* PP = lds + L::PP;
    float* SM = lds + L::SM;
    float* REC = lds + L::REC;
    const int64_t n_tiles = (g.n_rows + 31) / 32;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    TS_QREAL(0);
    TS_QMARK(0);

    // ---- the wave's resident weights: MFMA A operands (lane = (row m = n, k = gq))
    float W1a[K1S];                 // W1[fb + n][4 j + gq]  (0 beyond obs: those k-steps multiply the record's other fields)
    f32x4 B1, B2;                   // b1 / b2[fb + 4 gq + r]: the accumulator rows of this lane (bias = initial accumulator)
    float WH[4];                    // actor: Wmu[n][fb + 4 gq + r] (rows >= act: 0);   critic: wv[fb + 4 gq + r]
    [[maybe_unused]] float WHb[2];  // actor: Wmu[4 r + gq][fb + n] (head backward, k = action)
    // W2 is NOT resident (its two operand forms are 32 registers: 3 waves per SIMD): every tile re-reads the forward form
    // W2[fb + n][.] during phase 1 and the backward form W2[.][fb + n] during phase 3, one phase ahead of their use (L2).
    int64_t rid;
    [[maybe_unused]] float W2fr[16], W2tr[16];      // BIG: resident copies
    {
        TS_Q_LANE();
        rid = rowq_fetch(g, p, lane);           // first tile's row ids before anything else: the record gather depends on them
        if constexpr (BIG) {
#pragma unroll
            for (int jr = 0; jr < 16; ++jr) {
                const int f = 16 * (jr >> 2) + 4 * gq + (jr & 3);
                W2fr[jr] = prm[o_w2 + (fb + n) * HID + f];
                W2tr[jr] = prm[o_w2 + f * HID + fb + n];
            }
        }
#pragma unroll
        for (int j = 0; j < K1S; ++j) {
            const int k = 4 * j + gq;
            const float v = prm[o_w1 + (fb + n) * obs + (k < obs ? k : 0)];
            W1a[j] = k < obs ? v : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            B1[r] = prm[o_b1 + fb + 4 * gq + r];
            B2[r] = prm[o_b2 + fb + 4 * gq + r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (ACTOR) {
                const float v = prm[d.a_wmu + (n < n_act ? n : 0) * HID + fb + 4 * gq + r];
                WH[r] = n < n_act ? v : 0.f;
            } else {
                WH[r] = prm[d.c_wv + fb + 4 * gq + r];
            }
        }
        if constexpr (ACTOR) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int a = 4 * r + gq;
                const float v = prm[d.a_wmu + (a < n_act ? a : 0) * HID + fb + n];
                WHb[r] = a < n_act ? v : 0.f;
            }
        }
        // small constants (torch: sigma = exp(sigma_param); Normal.log_prob uses var = sigma^2 and log(sigma))
        if (tid < 32) {
            float v = 0.f;
            if constexpr (ACTOR) {
                const int a = tid & 7;
                if (tid < 8) v = a < n_act ? prm[d.a_bmu + a] : 0.f;
                else if (tid < 24) {
                    const float sigma = expf(a < n_act ? prm[d.a_sig + a] : 0.f);
                    v = tid < 16 ? 1.f / (2.f * (sigma * sigma)) : logf(sigma);
                }
            } else if (tid == 24) v = prm[d.c_bv];
            SM[tid] = v;
        }
        // the other network's columns of the pair's slab (one-network launches only)
        if (zero_other) {
            const int z0 = SL.w2t[1 - net], z1 = net == 0 ? SL.loss : SL.w2t[1];
            for (int c = z0 + 4 * tid; c < z1; c += 4 * QT) slab_st4(slab + c, zero4);
            if (tid == 0) slab_st(slab + SL.loss + (1 - net), 0.f);
        }
        const RecQ<K1S> f0 = recq_fetch<K1S>(g, rid, tid, lane);
        recq_commit<K1S>(f0, g, REC, tid);              // tile 0 -> buffer 0
    }

    // ---- persistent accumulators (MFMA C layout: lane (col n, group gq) register r = row 4 gq + r)
    f32x4 gW2[4], gW1[NB1];
#pragma unroll
    for (int c = 0; c < 4; ++c) gW2[c] = zero4;
#pragma unroll
    for (int c = 0; c < NB1; ++c) gW1[c] = zero4;
    f32x4 gH = zero4;               // actor: dWmu[4 gq + r][fb + n];  critic: lane-partial of dwv[fb + 4 gq + r]
    float rs = 0.f, rs1 = 0.f;      // lane-partials of db2[fb + n], db1[fb + n]
    float sD0 = 0.f, sD1 = 0.f, sS0 = 0.f, sS1 = 0.f, sL = 0.f;    // head-bias / sigma / loss partial sums
    __syncthreads();                                     // B0 of the first tile
    TS_QMARK(1);
    int cur = 0;
    [[maybe_unused]] int mk = 2;

    for (int64_t t = p; t < n_tiles; t += n_pairs) {
        const int64_t t_next = t + n_pairs;
        const bool has_next = t_next < n_tiles;          // uniform
        const float* RC = REC + cur * 32 * rec_w;        // this tile's records; the other buffer receives the next tile's
        float* RN = REC + (cur ^ 1) * 32 * rec_w;
        float W2f[16];                                   // W2[fb + n][16 jj + 4 gq + r]       (phase 2)
        [[maybe_unused]] f32x4 h1[2];                    // BIG: H1 stays in registers for tanh' in phase 4
        [[maybe_unused]] RecQ<K1S> fnext_big;

        // ================= phase 1: H1 = tanh(W1aug Xaug^T), own 16 features x 32 samples
        {
            TS_Q_LANE();
            if constexpr (BIG) rid = rowq_fetch(g, has_next ? t_next : t, lane);     // next tile's row ids
            // B operands straight from the records: lane (sample 16 b + n, k = gq) reads field 4 j + gq (every read issued
            // before the first MFMA; fields >= obs meet zero weights)
            float xv[2][K1S];
#pragma unroll
            for (int j = 0; j < K1S; ++j) {
                const int k = 4 * j + gq;
                const int kc = (4 * j + 3 < K1S * 4 - 4) ? k : (k < rec_w ? k : rec_w - 1);   // only the last k-step can leave the row
#pragma unroll
                for (int b = 0; b < 2; ++b) xv[b][j] = RC[(16 * b + n) * rec_w + kc];
            }
            f32x4 acc[2] = {B1, B1};
#pragma unroll
            for (int j = 0; j < K1S; ++j) {
                acc[0] = mfma16(W1a[j], xv[0][j], acc[0]);
                acc[1] = mfma16(W1a[j], xv[1][j], acc[1]);
            }
            if constexpr (BIG) {
#pragma unroll
                for (int jr = 0; jr < 16; ++jr) W2f[jr] = W2fr[jr];
            } else {
                const cgfloat_ptr pw = prm + o_w2 + (fb + n) * HID + 4 * gq;
#pragma unroll
                for (int jr = 0; jr < 16; ++jr) W2f[jr] = pw[16 * (jr >> 2) + (jr & 3)];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tanh4(acc[b]);
                if constexpr (BIG) h1[b] = acc[b];
                st4(R1 + (16 * b + n) * PS + fb + 4 * gq, acc[b]);
#pragma unroll
                for (int r = 0; r < 4; ++r) R3[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][r];
            }
        }
        __syncthreads();                                 // B1: H1 tiles complete
        TS_QMARK(mk + 0);

        // ================= phase 2: H2 = tanh(W2 H1 + b2); head forward partials
        f32x4 h2[2];
        {
            TS_Q_LANE();
            f32x4 acc[2] = {B2, B2};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b0 = ld4(R1 + n * PS + 16 * jj + 4 * gq);
                const f32x4 b1 = ld4(R1 + (16 + n) * PS + 16 * jj + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = mfma16(W2f[4 * jj + r], b0[r], acc[0]);
                    acc[1] = mfma16(W2f[4 * jj + r], b1[r], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);       // bounds the operand-read hoisting (register pressure)
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tanh4(acc[b]);
                h2[b] = acc[b];
            }
            if constexpr (ACTOR) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) R2[(fb + 4 * gq + r) * PF + 16 * b + n] = h2[b][r];   // for the head gradient
                    f32x4 pm = zero4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pm = mfma16(WH[r], h2[b][r], pm);     // rows = actions 4 gq + r
                    if (gq < 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) PP[((w * 2 + b) * 8 + 4 * gq + r) * 16 + n] = pm[r];
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float pv = (h2[b][0] * WH[0] + h2[b][1] * WH[1]) + (h2[b][2] * WH[2] + h2[b][3] * WH[3]);
                    pv = group4_sum(pv);
                    if (gq == 0) PP[(w * 2 + b) * 16 + n] = pv;
                }
            }
        }
        __syncthreads();                                 // B2: head partials complete; R1 (H1) is free
        TS_QMARK(mk + 1);

        // ================= phase 3: loss, dout, head gradients, dZ2
        float W2t[16];                                   // W2[16 jj + 4 gq + r][fb + n]       (phase 4)
        {
            TS_Q_LANE();
            if constexpr (!BIG) rid = rowq_fetch(g, has_next ? t_next : t, lane);    // next tile's row ids: in flight during phase 3
            else fnext_big = recq_fetch<K1S>(g, rid, tid, lane);       // row ids have had phases 1-2; records: phases 3-4
            f32x4 dz2[2];
            if constexpr (ACTOR) {
                const int a0 = gq, a1 = 4 + gq;
                const float bm0 = SM[a0], bm1 = SM[a1], iv0 = SM[8 + a0], iv1 = SM[8 + a1], ls0 = SM[16 + a0], ls1 = SM[16 + a1];
                const float c0 = a0 < n_act ? LOG_SQRT_2PI : 0.f, c1 = a1 < n_act ? LOG_SQRT_2PI : 0.f;
                float mean = 0.f, den = 1.f;
                if (g.adv_norm) { mean = g.adv_stats[0]; den = g.adv_stats[1] + 1e-8f; }
                const bool a2c = g.a2c != 0;
                const float lo = 1.f - g.eps_clip, hi = 1.f + g.eps_clip;
                float dout0[2], dout1[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int s = 16 * b + n;
                    const float* rp = RC + s * rec_w;
                    float mu0 = bm0, mu1 = bm1;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        mu0 += PP[((ww * 2 + b) * 8 + a0) * 16 + n];
                        mu1 += PP[((ww * 2 + b) * 8 + a1) * 16 + n];
                    }
                    const int i0 = obs + a0 < rec_w ? obs + a0 : rec_w - 1, i1 = obs + a1 < rec_w ? obs + a1 : rec_w - 1;
                    const float x0 = rp[i0], x1 = rp[i1];
                    const float act0 = a0 < n_act ? x0 : 0.f, act1 = a1 < n_act ? x1 : 0.f;
                    const float* aux = rp + obs + n_act;
                    const float adv = aux[0], logp_old = aux[2];
                    const float wgt = (t * 32 + s < g.n_rows) ? g.inv_batch : 0.f;
                    const float d0 = act0 - mu0, d1 = act1 - mu1;
                    // Normal.log_prob summed over the action dimension (padding actions contribute an exact 0)
                    float logp = (-(d0 * d0) * iv0 - ls0 - c0) + (-(d1 * d1) * iv1 - ls1 - c1);
                    logp = group4_sum(logp);
                    const float A = g.adv_norm ? (adv - mean) / den : adv;                   // ppo.py:184-186 ((x - 0) / 1 == x)
                    const float ratio = a2c ? 1.f : expf(logp - logp_old);                   // :187
                    const float surr1 = ratio * A;
                    const float surr2 = fminf(fmaxf(ratio, lo), hi) * A;                     // :190
                    const float clip1 = fminf(surr1, surr2);
                    float basek = (surr1 <= surr2) ? A : 0.f;                                // torch.min backward
                    const float dA = g.dual_clip * A;
                    const bool dual = (g.dual_clip > 0.f) && (A < 0.f);                      // :191-194
                    float term = dual ? -fmaxf(clip1, dA) : -clip1;                          // :196
                    basek = (dual && !(clip1 >= dA)) ? 0.f : basek;
                    term = a2c ? -logp * A : term;                                           // a2c.py:266-267
                    basek = a2c ? A : basek;
                    const float dlogp = -basek * ratio * wgt;
                    const float ent_w = g.ent_coef * wgt;
                    const float v0 = 2.f * iv0, v1 = 2.f * iv1;
                    dout0[b] = dlogp * d0 * v0;                                              // 0 for padding actions (d = 0)
                    dout1[b] = dlogp * d1 * v1;
                    const float ds0 = a0 < n_act ? dlogp * (d0 * d0 * v0 - 1.f) - ent_w : 0.f;   // entropy: d/d sigma_param = 1
                    const float ds1 = a1 < n_act ? dlogp * (d1 * d1 * v1 - 1.f) - ent_w : 0.f;
                    sD0 += dout0[b]; sD1 += dout1[b]; sS0 += ds0; sS1 += ds1; sL += term * wgt;
                    // dout, sample-major, into this wave's own columns of R1 (A operand of the head gradient)
                    R1[s * PS + fb + a0] = dout0[b];
                    R1[s * PS + fb + a1] = dout1[b];
                    if constexpr (!BIG) __builtin_amdgcn_sched_barrier(0);       // one block's loss at a time (register pressure)
                }
                wave_lds_sync();
                // head weight gradient: gH[a][f] += sum_s dout[s][a] H2[s][f]   (rows a = n & 7; rows 8..15 repeat them, unused)
                {
                    const f32x4 bv0 = ld4(R2 + (fb + n) * PF + 4 * gq), bv1 = ld4(R2 + (fb + n) * PF + 16 + 4 * gq);
                    float av0[4], av1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        av0[r] = R1[(4 * gq + r) * PS + fb + (n & 7)];
                        av1[r] = R1[(16 + 4 * gq + r) * PS + fb + (n & 7)];
                    }
                    f32x4 g1 = zero4;                    // second chain: the 40-cycle dependent latency of a lone chain
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gH = mfma16(av0[r], bv0[r], gH);
                        g1 = mfma16(av1[r], bv1[r], g1);
                    }
                    gH = gH + g1;
                }
                // dH2 = Wmu^T dout (k = action 4 r + gq), dZ2 = dH2 * (1 - H2^2)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 dh = mfma16(WHb[0], dout0[b], zero4);
                    dh = mfma16(WHb[1], dout1[b], dh);
                    dtanh4(dh, h2[b]);
                    dz2[b] = dh;
                }
            } else {
                const float bv = SM[24];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int s = 16 * b + n;
                    const float* aux = RC + s * rec_w + obs + n_act;
                    const float ret = aux[1], vo = aux[3];
                    const float wgt = (t * 32 + s < g.n_rows) ? g.inv_batch : 0.f;
                    float value = bv;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) value += PP[(ww * 2 + b) * 16 + n];
                    const float vf1 = (ret - value) * (ret - value);
                    // ppo.py:199-206 (torch.max backward: the larger branch takes the gradient, ties split it)
                    const float dvo = value - vo;
                    const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
                    const float vf2 = (ret - vclip) * (ret - vclip);
                    const float g1 = -2.f * (ret - value);
                    const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
                    const float dv_clip = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
                    const bool vc = g.value_clip != 0;
                    const float term = vc ? fmaxf(vf1, vf2) : vf1;                           // :208
                    const float dv = vc ? dv_clip : g1;
                    const float dout = dv * g.vf_coef * wgt;
                    sD0 += dout;
                    sL += term * wgt;
                    f32x4 dh;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gH[r] += dout * h2[b][r];
                        dh[r] = dout * WH[r];
                    }
                    dtanh4(dh, h2[b]);
                    dz2[b] = dh;
                    if constexpr (!BIG) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (BIG) {
#pragma unroll
                for (int jr = 0; jr < 16; ++jr) W2t[jr] = W2tr[jr];
            } else {
                const cgfloat_ptr pw = prm + o_w2 + fb + n + 4 * gq * HID;
#pragma unroll
                for (int jr = 0; jr < 16; ++jr) W2t[jr] = pw[(16 * (jr >> 2) + (jr & 3)) * HID];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                st4(R1 + (16 * b + n) * PS + fb + 4 * gq, dz2[b]);
#pragma unroll
                for (int r = 0; r < 4; ++r) R2[(fb + 4 * gq + r) * PF + 16 * b + n] = dz2[b][r];
            }
        }
        __syncthreads();                                 // B3: dZ2 (sample-major) complete
        TS_QMARK(mk + 2);

        // ================= phase 4: dZ1, weight gradients
        {
            TS_Q_LANE();
            // next tile's records: in flight during phase 4 (the longest one: 80 MFMAs), committed at its end -- nobody
            // reads REC behind B3
            RecQ<K1S> fnext;
            if constexpr (BIG) fnext = fnext_big;
            else fnext = recq_fetch<K1S>(g, rid, tid, lane);
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
            for (int b = 0; b < 2; ++b) {                // !BIG: H1 back from the wave's own rows of R3
                f32x4 hv;
                if constexpr (BIG) hv = h1[b];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = R3[(fb + 4 * gq + r) * PF + 16 * b + n];
                }
                dtanh4(acc[b], hv);
            }
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
            // dW1[f1 own][k] += sum_s dZ1[s][f1] X[s][k];  db1[f1] += sum_s dZ1[s][f1].  B operand from the records:
            // lane (column k = 16 c + n, gq) reads field k of samples 16 J + 4 gq + r (columns >= obs are not stored)
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const f32x4 av = ld4(R2 + (fb + n) * PF + 16 * J + 4 * gq);
                rs1 += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int c = 0; c < NB1; ++c) {
                    const int k = 16 * c + n < rec_w ? 16 * c + n : rec_w - 1;
                    const float* xp = RC + (16 * J + 4 * gq) * rec_w + k;
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = xp[r * rec_w];
#pragma unroll
                    for (int r = 0; r < 4; ++r) gW1[c] = mfma16(av[r], bv[r], gW1[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_next) recq_commit<K1S>(fnext, g, RN, tid);
        }
        cur ^= 1;
        __syncthreads();                                 // B0 of the next tile
        TS_QMARK(mk + 3);
#ifdef TS_PHASE_MARKS
        mk += 4;
#endif
    }

    // ---- epilogue: the workgroup's gradient sums leave once, 16 bytes per store
    TS_Q_LANE();
#pragma unroll
    for (int c = 0; c < 4; ++c) slab_st4(slab + SL.w2t[net] + (16 * c + n) * HID + fb + 4 * gq, gW2[c]);
#pragma unroll
    for (int c = 0; c < NB1; ++c)
        if (16 * c + n < 4 * K1S) slab_st4(slab + SL.w1t[net] + (16 * c + n) * HID + fb + 4 * gq, gW1[c]);
    rs = group4_sum(rs);
    rs1 = group4_sum(rs1);
    if (gq == 0) {
        slab_st(slab + SL.b2[net] + fb + n, rs);
        slab_st(slab + SL.b1[net] + fb + n, rs1);
    }
    if constexpr (ACTOR) {
        if (gq < 2) slab_st4(slab + SL.head[0] + (fb + n) * ACT_PAD + 4 * gq, gH);
        sD0 = row16_sum(sD0); sD1 = row16_sum(sD1); sS0 = row16_sum(sS0); sS1 = row16_sum(sS1); sL = row16_sum(sL);
        if (w == 0 && n == 0) {
            slab_st(slab + SL.hb[0] + gq, sD0);
            slab_st(slab + SL.hb[0] + 4 + gq, sD1);
            slab_st(slab + SL.sig + gq, sS0);
            slab_st(slab + SL.sig + 4 + gq, sS1);
            if (gq == 0) slab_st(slab + SL.loss, sL);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) gH[r] = row16_sum(gH[r]);
        if (n == 0) slab_st4(slab + SL.head[1] + fb + 4 * gq, gH);
        sD0 = row16_sum(sD0); sL = row16_sum(sL);
        if (w == 0 && lane == 0) {
            slab_st(slab + SL.hb[1], sD0);
            slab_st(slab + SL.loss + 1, sL);
        }
    }
    TS_QMARK(mk);
    TS_QREAL(1);
}
#undef TS_Q_LANE

// grid: nets == 0 / 3: 2 P workgroups (b < P: actor of pair b, b >= P: critic of pair b - P); nets == 1 / 2: P workgroups
template <int K1S, bool BIG>
__device__ __forceinline__ void stepq_body(const StepArgs& g, const Dims& d, int n_pairs, float* lds) {
    const Slab3 SL = slab3_layout(4 * K1S);
    const bool one = g.nets == 1 || g.nets == 2;
    const int b = blockIdx.x;
    const int net = one ? g.nets - 1 : (b >= n_pairs);
    const int p = (!one && b >= n_pairs) ? b - n_pairs : b;
    float* slab = g.slabs + (int64_t)p * g.slab_w;
    if (net == 0) stepq_run<K1S, true, BIG>(g, d, lds, p, n_pairs, slab, SL, one);
    else stepq_run<K1S, false, BIG>(g, d, lds, p, n_pairs, slab, SL, one);
}

template <int K1S>
__global__ __launch_bounds__(QT, 4) void ppo_stepq_kernel(StepArgs g, Dims d, int n_pairs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stepq_body<K1S, false>(g, d, n_pairs, lds);
}

template <int K1S>
__global__ __launch_bounds__(QT, 3) void ppo_stepq2_kernel(StepArgs g, Dims d, int n_pairs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stepq_body<K1S, true>(g, d, n_pairs, lds);
}

inline int k1s_for(int obs) {      // instantiated layer-1 depths (k-steps of 4 observation fields)
    const int need = (obs + 3) / 4;
    const int avail[] = {2, 3, 5, 8};
    for (int a : avail) if (a >= need) return a;
    return -1;
}

}  // namespace q4
