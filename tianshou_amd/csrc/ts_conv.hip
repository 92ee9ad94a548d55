// ts_conv.hip -- fp32-MFMA implicit-GEMM convolution / linear layers for gfx950 (NHWC activations).
//
// Replaces the torch conv2d / linear forward + autograd backward the reference runs for DQNet
// (tianshou/env/atari/atari_network.py:79-98, 111-122; loss.backward() at algorithm_base.py:495).
// Three GEMM shapes per layer, all on v_mfma_f32_32x32x2_f32 (bf16 is not admissible at 1e-5 rtol):
//   forward  Y[m, oc]  = sum_k  Xcol[m, k] W[k, oc] + b[oc]        m = (b, oh, ow), k = (kh, kw, ic)
//   wgrad    dW[k, oc] = sum_m  Xcol[m, k] dY[m, oc]               (+ db[oc] = sum_m dY[m, oc])
//   dgrad    dX[p, ic] = sum_{taps, oc} dY[q(p, tap), oc] W[(tap, ic), oc]   gather form, one launch
//            slice per stride-parity class so that every row of a tile uses the same taps
// Xcol is never materialised: with NHWC activations the (kw, ic) part of k is contiguous in memory, so
// a 32-wide k chunk of one output pixel is one (or a few) 16-byte aligned runs, loaded with clamped
// unconditional float4 loads and staged through LDS.  Roofline: fp32 MFMA; see DESIGN.md.
#include <cstdlib>

#include "ts_common.h"
#include "ts_conv.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 32;          // reduction chunk staged per iteration
constexpr int THREADS = 256;    // 4 waves

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct GemmArgs {
    const float* A;       // forward / wgrad: layer input X;  dgrad: dY
    const float* Bm;      // forward / dgrad: Wb;             wgrad: dY
    float* C;             // forward: Y (or split slabs);      wgrad: slabs;  dgrad: dX
    const float* bias;
    const float* mask;
    ts::ConvGeom g;
    int M;                // forward / wgrad: B*OH*OW;  dgrad: rows per parity class = B*AH*AW
    int K;                // KH*KW*IC
    int relu;
    int chunks;           // reduction chunks per split
    int total_chunks;
    int AH, AW, JH, JW;   // dgrad: per-class output grid and taps per class
    int n_tile0;          // dgrad: first column tile of this launch
    int a_u8;             // forward / wgrad: the layer input is uint8 NHWC (raw frames), converted on load
    int xcd;              // wgrad: 1 = XCD-aware block -> (tile, split) map (xcd_item)
    int64_t slab_stride;
};

// Workgroup b of a launch runs on XCD b % 8 (observed dispatch rule, MI355X_MICROARCH.md "Workgroup dispatch": used for speed
// only), and every XCD has its own L2.  The weight-gradient grids order their work items tile-fastest: the (k tile, column
// tile) items of ONE reduction split -- which all read the same rows of X and dY -- are neighbours, i.e. land on eight
// DIFFERENT XCDs, and every XCD fetches those rows from HBM / the Infinity Cache for itself (C5 critics: 70 % L2 misses,
// 189 MB fetched for 12 MB of operands, profiles/r06_pmc_sac_tcc.txt).  xcd_item gives XCD x the x-th CONTIGUOUS range of the
// item sequence instead: block b takes item start(x) + b / 8 with x = b % 8, so a split's tiles share an L2 and run at about
// the same time.  A bijection on [0, n): results are bit-identical, any other block -> XCD placement is merely no faster.
__device__ __forceinline__ int xcd_item(int b, int n) {
    const int q = n >> 3, rem = n & 7, x = b & 7;
    return x * q + (x < rem ? x : rem) + (b >> 3);
}

// ------------------------------------------------------------------------------------------------
// rows = pixels: forward (DG = false) and dgrad (DG = true)
// ------------------------------------------------------------------------------------------------
template <bool DG, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(THREADS) void conv_rows_kernel(GemmArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LDA = BK + 1, LDB = DG ? BN + 2 : BN;
    constexpr int AJ = BM / 32, BJ = BN / 32;
    static_assert(WM * WN == 4, "four waves");
    __shared__ float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    __shared__ int s_in[BM];
    __shared__ int s_ac[DG ? BM : 1];
    __shared__ int s_out[DG ? BM : 1];
    const ts::ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = (blockIdx.y + (DG ? a.n_tile0 : 0)) * BN;
    int ph = 0, pw = 0, split = 0;
    if (DG) { ph = blockIdx.z / g.S; pw = blockIdx.z % g.S; } else { split = blockIdx.z; }

    for (int i = tid; i < BM; i += THREADS) {
        const int m = m0 + i;
        const bool ok = m < a.M;
        const int mc = ok ? m : a.M - 1;
        if (!DG) {
            const int b = mc / (g.OH * g.OW), rem = mc - b * g.OH * g.OW;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            s_in[i] = ((b * g.IH + oh * g.S) * g.IW + ow * g.S) * g.IC;
        } else {
            const int b = mc / (a.AH * a.AW), rem = mc - b * a.AH * a.AW;
            const int aa = rem / a.AW, cc = rem - aa * a.AW;
            s_in[i] = ((b * g.OH + aa) * g.OW + cc) * g.OC;
            s_ac[i] = aa | (cc << 16);
            const int ih = aa * g.S + ph, iw = cc * g.S + pw;
            s_out[i] = (ok && ih < g.IH && iw < g.IW) ? ((b * g.IH + ih) * g.IW + iw) * g.IC : -1;
        }
    }
    __syncthreads();

    const int q = tid & 7, rowi = tid >> 3;          // A staging: float4 q of rows rowi + 32 j
    const int run = g.KW * g.IC, pitch = g.IW * g.IC;
    f32x4 ar[AJ], br[BJ];
    // loop invariants of the staging loads in registers (they sit behind barriers, the compiler re-reads LDS otherwise), and
    // the (kh, k - kh run) split of this thread's k advanced by BK per chunk instead of a division per chunk: the B = 512
    // launches issued 4.3 VALU + 4.3 SALU instructions per MFMA (profiles/r04_pmc_dqn_kernels.txt)
    int sin_r[AJ], sac_r[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) { sin_r[j] = s_in[rowi + 32 * j]; sac_r[j] = DG ? s_ac[rowi + 32 * j] : 0; }
    const int c_first = (DG ? 0 : split) * a.chunks;
    int kh_r = 0, krem_r = 0;
    if (!DG) { const int k = c_first * BK + 4 * q; kh_r = k / run; krem_r = k - kh_r * run; }
    const float* bptr[BJ];
    if (!DG) {
#pragma unroll
        for (int i = 0; i < BJ; ++i) {
            const int e = tid + THREADS * i, kk = e / (BN / 4), n4 = e % (BN / 4);
            bptr[i] = a.Bm + (int64_t)(c_first * BK + kk) * g.OC + n0 + 4 * n4;
        }
    }
    const int64_t bstep = (int64_t)BK * g.OC;
    int dg_oc0 = 0, dg_jh = 0, dg_jw = 0;            // dgrad: (tap, first output channel) of the next chunk (chunks start at 0)

    auto gload = [&](int) {                          // chunks are requested in order, once each
        if (!DG) {
            const int koff = kh_r * pitch + krem_r;
            krem_r += BK;                            // next chunk (chunks are loaded in order, once each)
            while (krem_r >= run) { krem_r -= run; ++kh_r; }
            if (a.a_u8) {          // 4 consecutive uint8 pixels/channels per lane -> 4 floats (exact)
                const uint8_t* a8 = reinterpret_cast<const uint8_t*>(a.A);
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(a8 + sin_r[j] + koff);
                    ar[j] = f32x4{(float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24)};
                }
            } else {
#pragma unroll
                for (int j = 0; j < AJ; ++j)
                    ar[j] = *reinterpret_cast<const f32x4*>(a.A + sin_r[j] + koff);
            }
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                br[i] = *reinterpret_cast<const f32x4*>(bptr[i]);
                bptr[i] += bstep;
            }
        } else {
            const int oc0 = dg_oc0, jh = dg_jh, jw = dg_jw;          // k0 = (jh JW + jw) OC + oc0, advanced per chunk
            dg_oc0 += BK;
            if (dg_oc0 >= g.OC) { dg_oc0 -= g.OC; if (++dg_jw == a.JW) { dg_jw = 0; ++dg_jh; } }
            const int shift = (jh * g.OW + jw) * g.OC - oc0 - 4 * q;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int ac = sac_r[j];
                const int oh = (ac & 0xffff) - jh, ow = (ac >> 16) - jw;
                const bool ok = oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW;
                const f32x4 v = *reinterpret_cast<const f32x4*>(a.A + (ok ? sin_r[j] - shift : 4 * q));
                ar[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int tapk = ((ph + g.S * jh) * g.KW + pw + g.S * jw) * g.IC;
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                const int e = tid + THREADS * i, n = e >> 3, kk4 = e & 7;
                br[i] = *reinterpret_cast<const f32x4*>(a.Bm + (int64_t)(tapk + n0 + n) * g.OC + oc0 + 4 * kk4);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            float* d = &As[(rowi + 32 * j) * LDA + 4 * q];
            d[0] = ar[j][0]; d[1] = ar[j][1]; d[2] = ar[j][2]; d[3] = ar[j][3];
        }
#pragma unroll
        for (int i = 0; i < BJ; ++i) {
            const int e = tid + THREADS * i;
            if (!DG) {
                const int kk = e / (BN / 4), n4 = e % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[kk * LDB + 4 * n4]) = br[i];
            } else {
                const int n = e >> 3, kk4 = e & 7;
#pragma unroll
                for (int x = 0; x < 4; ++x) Bs[(4 * kk4 + x) * LDB + n] = br[i][x];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[tm][tn][x] = 0.f;

    const int c_begin = split * a.chunks;            // (== c_first for the forward pass; dgrad launches are not split)
    const int c_end = min(a.total_chunks, c_begin + a.chunks);
    if (c_begin < c_end) gload(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (c + 1 < c_end) gload(c + 1);
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 2; ++kk2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[tm] = As[(wm * TM * 32 + tm * 32 + r) * LDA + 2 * kk2 + h];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[tn] = Bs[(2 * kk2 + h) * LDB + wn * TN * 32 + tn * 32 + r];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(av[tm], bv[tn], acc[tm][tn]);
        }
    }

    const bool to_slab = !DG && a.slab_stride > 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + tn * 32 + r;
            const float bias = (!DG && !to_slab) ? a.bias[col] : 0.f;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int row = wm * TM * 32 + tm * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                float v = acc[tm][tn][x];
                if (!DG) {
                    const int m = m0 + row;
                    if (m < a.M) {
                        if (to_slab) {
                            a.C[split * a.slab_stride + (int64_t)m * g.OC + col] = v;
                        } else {
                            v += bias;
                            if (a.relu) v = fmaxf(v, 0.f);
                            a.C[(int64_t)m * g.OC + col] = v;
                        }
                    }
                } else {
                    const int o = s_out[row];
                    if (o >= 0) {
                        if (a.mask) v = a.mask[o + col] > 0.f ? v : 0.f;
                        a.C[o + col] = v;
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// wgrad: rows = k, columns = oc, reduction over output pixels; split over pixel ranges into slabs
// ------------------------------------------------------------------------------------------------
// LDS of one weight-gradient workgroup: As | Bs | s_red | s_row.  The caller owns the buffer, so that a kernel that inlines
// several tile variants (conv_wgrad_group_kernel) overlays them instead of adding them up.
template <int TM, int TN, int WM, int WN>
constexpr int wgrad_smem_floats() { return BK * (WM * TM * 32) + BK * (WN * TN * 32) + THREADS + 2 * BK; }

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_wgrad_body(const GemmArgs& a, const int kt, const int ny, const int split, float* smem) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int AQ = BM / 4;                 // float4 per staged row of A
    constexpr int ARS = THREADS / AQ;          // row stride between a thread's A loads
    constexpr int AI = BK / ARS, BJ = BN / 32;
    static_assert(WM * WN == 4, "four waves");
    float* As = smem;                          // [BK * BM], 16-byte aligned
    float* Bs = As + BK * BM;                  // [BK * BN]
    float* s_red = Bs + BK * BN;               // [THREADS]
    int (*s_row)[BK] = reinterpret_cast<int (*)[BK]>(s_red + THREADS);      // [2][BK]
    const ts::ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const int n0 = ny * BN;
    const int run = g.KW * g.IC, pitch = g.IW * g.IC;

    const int k4 = tid % AQ, arow = tid / AQ;
    const int k = kt * BM + 4 * k4;
    const int kc = k < a.K ? k : 0;
    const int kh = kc / run;
    const int koff = kh * pitch + (kc - kh * run);
    f32x4 ar[AI], br[BJ];

    auto rowinfo = [&](int c) {
        if (tid < BK) {
            const int m = min(c * BK + tid, a.M - 1);
            const int b = m / (g.OH * g.OW), rem = m - b * g.OH * g.OW;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            s_row[c & 1][tid] = ((b * g.IH + oh * g.S) * g.IW + ow * g.S) * g.IC;
        }
    };
    auto gload = [&](int c) {
        if (a.a_u8) {
            const uint8_t* a8 = reinterpret_cast<const uint8_t*>(a.A);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(a8 + s_row[c & 1][arow + ARS * i] + koff);
                ar[i] = f32x4{(float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24)};
            }
        } else {
#pragma unroll
            for (int i = 0; i < AI; ++i)
                ar[i] = *reinterpret_cast<const f32x4*>(a.A + s_row[c & 1][arow + ARS * i] + koff);
        }
#pragma unroll
        for (int i = 0; i < BJ; ++i) {
            const int e = tid + THREADS * i, mm = e / (BN / 4), n4 = e % (BN / 4);
            const int m = c * BK + mm;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.Bm + (int64_t)min(m, a.M - 1) * g.OC + n0 + 4 * n4);
            br[i] = m < a.M ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4*>(&As[(arow + ARS * i) * BM + 4 * k4]) = ar[i];
#pragma unroll
        for (int i = 0; i < BJ; ++i) {
            const int e = tid + THREADS * i, mm = e / (BN / 4), n4 = e % (BN / 4);
            *reinterpret_cast<f32x4*>(&Bs[mm * BN + 4 * n4]) = br[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[tm][tn][x] = 0.f;
    float bsum = 0.f;
    const int bn = tid % BN, bp = tid / BN;

    const int c_begin = split * a.chunks;
    const int c_end = min(a.total_chunks, c_begin + a.chunks);
    if (c_begin < c_end) rowinfo(c_begin);
    __syncthreads();
    if (c_begin < c_end) gload(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();
        lstore();
        if (c + 1 < c_end) rowinfo(c + 1);
        __syncthreads();
        if (c + 1 < c_end) gload(c + 1);
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 2; ++kk2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[tm] = As[(2 * kk2 + h) * BM + wm * TM * 32 + tm * 32 + r];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[tn] = Bs[(2 * kk2 + h) * BN + wn * TN * 32 + tn * 32 + r];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(av[tm], bv[tn], acc[tm][tn]);
        }
        if (kt == 0) {
#pragma unroll
            for (int mm = bp; mm < BK; mm += THREADS / BN) bsum += Bs[mm * BN + bn];
        }
    }

    float* out = a.C + split * a.slab_stride;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + tn * 32 + r;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int kr = kt * BM + wm * TM * 32 + tm * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                if (kr < a.K) out[(int64_t)kr * g.OC + col] = acc[tm][tn][x];
            }
        }
    if (kt == 0) {
        s_red[tid] = bsum;
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < THREADS / BN; ++p) s += s_red[p * BN + tid];
            out[(int64_t)a.K * g.OC + n0 + tid] = s;
        }
    }
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(THREADS) void conv_wgrad_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[wgrad_smem_floats<TM, TN, WM, WN>()];
    int kt = blockIdx.x, ny = blockIdx.y, split = blockIdx.z;
    if (a.xcd) {          // (dispatch order of a 3-D grid: x fastest)
        const int per = gridDim.x * gridDim.y;
        const int item = xcd_item((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), per * (int)gridDim.z);
        split = item / per;
        const int rem = item - split * per;
        ny = rem / (int)gridDim.x;
        kt = rem - ny * (int)gridDim.x;
    }
    conv_wgrad_body<TM, TN, WM, WN>(a, kt, ny, split, smem);
}

// The weight gradients of several layers in ONE launch (the three Linear layers of a SAC-family network, or of both twin
// critics: 0.07 - 0.9 GFLOP each, 9 - 15 us as launches of their own, most of it launch and tail).  blockIdx.x runs over
// the layers' (k tile, column tile, split) grids back to back; a layer is the 128 x 64 or the 128 x 32 variant by its
// output width, exactly as in conv_wgrad, so that the result is bit-identical to separate launches.
constexpr int WGRAD_GROUP_MAX = 16;
struct WgradGroup {
    GemmArgs a[WGRAD_GROUP_MAX];
    int first[WGRAD_GROUP_MAX + 1];        // first linear block of layer i
    int gx[WGRAD_GROUP_MAX], gy[WGRAD_GROUP_MAX];
    int wide[WGRAD_GROUP_MAX];             // 1: 64 columns (<2, 1, 2, 2>, or <1, 1, 2, 2> in the SMALL kernel), 0: <1, 1, 4, 1> (32 columns)
    int n;
    int xcd;                               // 1: XCD-aware block -> item map (xcd_item)
};

static_assert(sizeof(WgradGroup) <= 4000, "kernel arguments are limited to 4 KB");

template <bool SMALL>
__global__ __launch_bounds__(THREADS) void conv_wgrad_group_kernel(WgradGroup gr) {
    const int item = gr.xcd ? xcd_item((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int i = 0;
    while (i + 1 < gr.n && item >= gr.first[i + 1]) ++i;
    const int local = item - gr.first[i];
    const int per = gr.gx[i] * gr.gy[i];
    const int split = local / per, rem = local - split * per;
    const int ny = rem / gr.gx[i], kt = rem - ny * gr.gx[i];
    constexpr int S1 = wgrad_smem_floats<2, 1, 2, 2>() > wgrad_smem_floats<1, 1, 4, 1>() ? wgrad_smem_floats<2, 1, 2, 2>()
                                                                                      : wgrad_smem_floats<1, 1, 4, 1>();
    // SMALL: only the 64 x 64 and 128 x 32 variants are inlined -- 65 registers instead of 98 (the kernel's allocation is its
    // largest variant's): 7 instead of 4 waves per SIMD
    constexpr int S2 = wgrad_smem_floats<1, 1, 2, 2>() > wgrad_smem_floats<1, 1, 4, 1>() ? wgrad_smem_floats<1, 1, 2, 2>()
                                                                                      : wgrad_smem_floats<1, 1, 4, 1>();
    constexpr int SMEM = SMALL ? S2 : S1;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    if constexpr (SMALL) {
        if (gr.wide[i]) conv_wgrad_body<1, 1, 2, 2>(gr.a[i], kt, ny, split, smem);
        else conv_wgrad_body<1, 1, 4, 1>(gr.a[i], kt, ny, split, smem);
    } else {
        if (gr.wide[i]) conv_wgrad_body<2, 1, 2, 2>(gr.a[i], kt, ny, split, smem);
        else conv_wgrad_body<1, 1, 4, 1>(gr.a[i], kt, ny, split, smem);
    }
}

// out[i] = sum_s slabs[s][i]: a workgroup owns 64 consecutive floats (16 float4 columns) and splits the
// slabs over its 16 thread rows; fixed-order tree over the rows (deterministic).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int nslab, int64_t n,
                                                       float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int64_t i = ((int64_t)blockIdx.x * 16 + col) * 4;            // n % 4 == 0
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n)
        for (int k = part; k < nslab; k += 16) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * n + i);
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int st = 8; st > 0; st >>= 1) {
        if (part < st) red[threadIdx.x] += red[threadIdx.x + 16 * st];
        __syncthreads();
    }
    if (part == 0 && i < n) *reinterpret_cast<f32x4*>(out + i) = red[col];
}

// Few slabs (<= 16: every thread row of slab_sum_kernel holds at most one): a thread owns one float4 column, loads its NS values
// and adds them in slab_sum_kernel's tree order (rows beyond the slab count hold zero there; x + 0 is exact), so the result
// is bit-identical -- without leaving 16 - NS of every 16 threads idle (fc1 of the NatureCNN: 1.6 M gradients in 4 slabs,
// 25,096 workgroups of 4 KB each -> 1,569 workgroups)
template <int NS>
__global__ __launch_bounds__(256) void slab_sum_few_kernel(const float* __restrict__ slabs, int nslab, int64_t n,
                                                           float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 r[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r[k] = z;
        if (k < nslab) r[k] += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * n + i);
    }
#pragma unroll
    for (int st = NS / 2; st > 0; st >>= 1)
#pragma unroll
        for (int q = 0; q < st; ++q) r[q] += r[q + st];
    *reinterpret_cast<f32x4*>(out + i) = r[0];
}

// The same sum for up to eight independent slab sets in one launch (the layers of one or two networks' backward passes)
constexpr int SLAB_SEGS_MAX = 16;
struct SlabSegs {
    const float* slabs[SLAB_SEGS_MAX]; float* out[SLAB_SEGS_MAX]; int64_t n[SLAB_SEGS_MAX]; int nslab[SLAB_SEGS_MAX];
    unsigned first_block[SLAB_SEGS_MAX + 1];
};
__global__ __launch_bounds__(256) void slab_sum_multi_kernel(SlabSegs a) {
    __shared__ f32x4 red[256];
    int sg = 0;
#pragma unroll
    for (int k = 1; k < SLAB_SEGS_MAX; ++k) sg += blockIdx.x >= a.first_block[k] ? 1 : 0;
    const float* __restrict__ slabs = a.slabs[sg];
    const int64_t n = a.n[sg];
    const int nslab = a.nslab[sg];
    const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int64_t i = ((int64_t)(blockIdx.x - a.first_block[sg]) * 16 + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n)
        for (int k = part; k < nslab; k += 16) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * n + i);
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int st = 8; st > 0; st >>= 1) {
        if (part < st) red[threadIdx.x] += red[threadIdx.x + 16 * st];
        __syncthreads();
    }
    if (part == 0 && i < n) *reinterpret_cast<f32x4*>(a.out[sg] + i) = red[col];
}

// Y[m, n] = act(sum_s buf[s][m, n] + bias[n])   (finish of a split forward)
__global__ __launch_bounds__(256) void split_finish_kernel(const float* __restrict__ buf, int nsplit, int64_t mn,
                                                           int n, const float* __restrict__ bias, int relu,
                                                           float* __restrict__ y) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;                                  // n % 4 == 0, so a float4 never straddles rows
    f32x4 s = *reinterpret_cast<const f32x4*>(buf + i);
    for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(buf + (int64_t)k * mn + i);
    s += *reinterpret_cast<const f32x4*>(bias + (i % n));
    if (relu) { s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f); }
    *reinterpret_cast<f32x4*>(y + i) = s;
}

int check_geom(const ts::ConvGeom& g) {
    TS_REQUIRE(g.B > 0 && g.OH > 0 && g.OW > 0, TS_ERR_INVALID_ARG, "conv: empty geometry");
    TS_REQUIRE(g.OH == (g.IH - g.KH) / g.S + 1 && g.OW == (g.IW - g.KW) / g.S + 1, TS_ERR_INVALID_ARG,
               "conv: inconsistent output size");
    TS_REQUIRE(g.K() % BK == 0 && (g.KW * g.IC) % 4 == 0 && (g.IW * g.IC) % 4 == 0 && (g.S * g.IC) % 4 == 0,
               TS_ERR_INVALID_ARG, "conv: unsupported shape (K %% 32, 16-byte alignment of the im2col runs)");
    TS_REQUIRE(g.OC % 32 == 0, TS_ERR_INVALID_ARG, "conv: output channels must be a multiple of 32");
    TS_REQUIRE(g.in_elems() < (1ll << 31) && g.out_elems() < (1ll << 31), TS_ERR_INVALID_ARG,
               "conv: tensor too large for 32-bit offsets (split the batch)");
    return TS_OK;
}

GemmArgs base_args(const ts::ConvGeom& g) {
    GemmArgs a{};
    a.g = g;
    a.M = g.B * g.OH * g.OW;
    a.K = g.K();
    return a;
}

constexpr int TARGET_WGS = 768;    // 256 CUs x ~3 workgroups

// Row-tile choice.  Measured at the C3 / C5 shapes (scripts/gpu_conv_micro.py): the half-size tiles (128 x 32,
// 64 x 64) win everywhere (conv1 53.5 -> 46.6 us, conv2 47.7 -> 39.3, fc1 35.5 -> 29.7): with twice as many,
// shorter workgroups the prologue loads / epilogue stores of one overlap the MFMA phase of another instead of
// every workgroup of a single resident round hitting them in lock step.  The full-size tiles stay for very large
// grids, where they halve the weight-tile traffic.
bool small_rows(int M, int bn, int other_tiles) {
    static const char* force = getenv("TS_CONV_SMALL");      // experiments: "1" / "0" force the choice
    if (force) return force[0] == '1';
    const int bm = bn == 64 ? 128 : 256;
    return ts::ceil_div(M, bm) * other_tiles < 8192;
}

}  // namespace

namespace ts {

int conv_fwd_splits(const ConvGeom& g) {
    if (conv2_use_forward(g, false)) return 1;       // second generation never splits the reduction
    const int bn = g.OC % 64 == 0 ? 64 : 32, bm = bn == 64 ? 128 : 256;
    const int64_t tiles = ceil_div((int64_t)g.B * g.OH * g.OW, bm) * (g.OC / bn);
    const int chunks = g.K() / BK;
    if (tiles >= 192 || chunks < 16) return 1;
    int want = (int)ceil_div(TARGET_WGS / 2, tiles);
    int per = (int)ceil_div(chunks, want);
    if (per < 4) per = 4;
    return (int)ceil_div(chunks, per);
}

// 64 x 64 tiles (k rows x columns) instead of 128 x 64 for 64-column layers.  Round 6 (profiles/r06_wgrad_ab.txt): the
// weight-gradient kernels at these sizes are bound by how many workgroups hide each other's load -> LDS -> MFMA latencies (matrix
// pipe busy 37 % of the cycles, waves parked at waitcnt / barriers 43-50 %: profiles/r06_pmc_sac_sq.txt), not by operand traffic
// (the XCD-aware block order removed the 70 % L2-miss re-reads and bought 7 %) and not by MFMAs per barrier (128 x 128 tiles:
// 17 % slower).  The 64 x 64 variant needs 65 registers instead of 98: 7 instead of 4 waves per SIMD.  Default ON inside the
// grouped launch (the SAC family: -6 % on the launch, REDQ -9 %), OFF for single launches (DQN's weight gradients run beside
// the input-gradient chain on side streams: 151 -> 141 us of kernels, no change of the update).  TS_WGRAD_TILE64=0 / 1 forces
// it (A/B runs; read per call).  Every tiling sums the same chunks in the same order: results are bit-identical.
static bool wgrad_tile64(const ConvGeom& g, bool grouped) {
    const char* e = getenv("TS_WGRAD_TILE64");
    return g.OC % 64 == 0 && (e ? atoi(e) != 0 : grouped);
}

int conv_wgrad_splits(const ConvGeom& g) {
    if (conv2_use_wgrad(g, false)) return conv2_wgrad_splits(g);
    // (the split count follows the 128 x 64 tile count whatever the tile variant: every output element sums the same chunks in
    // the same order -- bit-identical results)
    const int bn = g.OC % 64 == 0 ? 64 : 32;
    const int64_t tiles = ceil_div(g.K(), 128) * (g.OC / bn);
    const int chunks = (int)ceil_div((int64_t)g.B * g.OH * g.OW, BK);
    int want = (int)ceil_div(TARGET_WGS, tiles);
    int per = (int)ceil_div(chunks, want);
    if (per < 4) per = 4;
    return (int)ceil_div(chunks, per);
}

int conv_forward(hipStream_t s, const ConvGeom& g, const float* X, const float* Wb, float* Y, bool relu,
                 float* split_buf, ts_workspace* prof, bool x_u8) {
    if (int rc = check_geom(g)) return rc;
    if (conv2_use_forward(g, x_u8)) return conv2_forward(s, g, X, Wb, Y, relu, prof, x_u8);
    GemmArgs a = base_args(g);
    a.a_u8 = x_u8;
    a.A = X; a.Bm = Wb; a.bias = Wb + (int64_t)a.K * g.OC; a.relu = relu;
    a.total_chunks = a.K / BK;
    const int nsplit = conv_fwd_splits(g);
    a.chunks = (int)ceil_div(a.total_chunks, nsplit);
    if (nsplit > 1) {
        TS_REQUIRE(split_buf, TS_ERR_INVALID_ARG, "conv_forward: split buffer missing");
        a.C = split_buf; a.slab_stride = g.out_elems();
    } else {
        a.C = Y; a.slab_stride = 0;
    }
    {
        ProfScope scope(prof, TS_KIND_CONV_FWD, s);
        const int bn = g.OC % 64 == 0 ? 64 : 32;
        const bool small = small_rows(a.M, bn, g.OC / bn * nsplit);
        const int bm = (bn == 64 ? 128 : 256) / (small ? 2 : 1);
        dim3 grid((unsigned)ceil_div(a.M, bm), g.OC / bn, nsplit);
        if (bn == 64 && !small) hipLaunchKernelGGL((conv_rows_kernel<false, 2, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
        else if (bn == 64) hipLaunchKernelGGL((conv_rows_kernel<false, 1, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
        else if (!small) hipLaunchKernelGGL((conv_rows_kernel<false, 2, 1, 4, 1>), grid, dim3(THREADS), 0, s, a);
        else hipLaunchKernelGGL((conv_rows_kernel<false, 1, 1, 4, 1>), grid, dim3(THREADS), 0, s, a);
    }
    TS_LAUNCH_CHECK();
    if (nsplit > 1) {
        const int64_t mn = g.out_elems();
        hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)ceil_div(mn, 1024)), dim3(256), 0, s, split_buf,
                           nsplit, mn, g.OC, a.bias, (int)relu, Y);
        TS_LAUNCH_CHECK();
    }
    return TS_OK;
}

// TS_WGRAD_XCD=0: the plain block order (A/B runs; read per call)
static int wgrad_xcd() {
    const char* e = getenv("TS_WGRAD_XCD");
    return e ? atoi(e) != 0 : 1;
}

int conv_wgrad(hipStream_t s, const ConvGeom& g, const float* X, const float* dY, float* slabs,
               ts_workspace* prof, bool x_u8) {
    if (int rc = check_geom(g)) return rc;
    if (conv2_use_wgrad(g, false)) {
        // (the split count is a function of the geometry alone: uint8 first layers have 32 channels, see conv2_use_wgrad)
        if (conv2_use_wgrad(g, x_u8)) return conv2_wgrad(s, g, X, dY, slabs, prof, x_u8);
        return ts::fail(TS_ERR_UNSUPPORTED, "conv_wgrad: uint8 input with %d output channels", g.OC);
    }
    GemmArgs a = base_args(g);
    a.a_u8 = x_u8;
    a.A = X; a.Bm = dY; a.C = slabs;
    a.total_chunks = (int)ceil_div(a.M, BK);
    const int nsplit = conv_wgrad_splits(g);
    a.chunks = (int)ceil_div(a.total_chunks, nsplit);
    a.slab_stride = g.param_elems();
    a.xcd = wgrad_xcd();
    ProfScope scope(prof, TS_KIND_CONV_WGRAD, s);
    if (wgrad_tile64(g, false)) {
        dim3 grid((unsigned)ceil_div(a.K, 64), g.OC / 64, nsplit);
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
    } else if (g.OC % 64 == 0) {
        dim3 grid((unsigned)ceil_div(a.K, 128), g.OC / 64, nsplit);
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
    } else {
        dim3 grid((unsigned)ceil_div(a.K, 128), g.OC / 32, nsplit);
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 4, 1>), grid, dim3(THREADS), 0, s, a);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int conv_wgrad_group(hipStream_t s, int n, const ConvGeom* g, const float* const* X, const float* const* dY,
                     float* const* slabs, ts_workspace* prof) {
    TS_REQUIRE(n >= 1 && n <= WGRAD_GROUP_MAX, TS_ERR_INVALID_ARG, "conv_wgrad_group: 1 .. %d layers", WGRAD_GROUP_MAX);
    bool one_launch = n > 1;
    for (int i = 0; i < n; ++i) {
        if (int rc = check_geom(g[i])) return rc;
        if (conv2_use_wgrad(g[i], false)) one_launch = false;        // large layers have their own kernels (ts_conv2.hip)
    }
    static const bool off = getenv("TS_WGRAD_NO_GROUP") != nullptr;     // experiments
    if (!one_launch || off) {
        for (int i = 0; i < n; ++i)
            if (int rc = conv_wgrad(s, g[i], X[i], dY[i], slabs[i], prof, false)) return rc;
        return TS_OK;
    }
    WgradGroup gr;
    gr.n = n;
    int blocks = 0;
    bool small = false;
    for (int i = 0; i < n; ++i) small = small || wgrad_tile64(g[i], true);
    for (int i = 0; i < n; ++i) {
        GemmArgs a = base_args(g[i]);
        a.a_u8 = 0;
        a.A = X[i]; a.Bm = dY[i]; a.C = slabs[i];
        a.total_chunks = (int)ceil_div(a.M, BK);
        const int nsplit = conv_wgrad_splits(g[i]);
        a.chunks = (int)ceil_div(a.total_chunks, nsplit);
        a.slab_stride = g[i].param_elems();
        gr.a[i] = a;
        gr.wide[i] = g[i].OC % 64 == 0 ? 1 : 0;
        gr.gx[i] = (int)ceil_div(a.K, (small && gr.wide[i]) ? 64 : 128);
        gr.gy[i] = g[i].OC / (gr.wide[i] ? 64 : 32);
        gr.first[i] = blocks;
        blocks += gr.gx[i] * gr.gy[i] * nsplit;
    }
    for (int i = n; i <= WGRAD_GROUP_MAX; ++i) gr.first[i] = blocks;
    gr.xcd = wgrad_xcd();
    ProfScope scope(prof, TS_KIND_CONV_WGRAD, s);
    if (small) hipLaunchKernelGGL(conv_wgrad_group_kernel<true>, dim3((unsigned)blocks), dim3(THREADS), 0, s, gr);
    else hipLaunchKernelGGL(conv_wgrad_group_kernel<false>, dim3((unsigned)blocks), dim3(THREADS), 0, s, gr);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int conv_dgrad(hipStream_t s, const ConvGeom& g, const float* dY, const float* Wb, const float* mask,
               float* dX, ts_workspace* prof, int col_begin, int col_end) {
    if (int rc = check_geom(g)) return rc;
    TS_REQUIRE(g.KH % g.S == 0 && g.KW % g.S == 0, TS_ERR_INVALID_ARG,
               "conv_dgrad: kernel size must be a multiple of the stride");
    TS_REQUIRE(g.IC % 32 == 0, TS_ERR_INVALID_ARG, "conv_dgrad: input channels must be a multiple of 32");
    if (conv2_use_dgrad(g, prof != nullptr, col_begin, col_end)) return conv2_dgrad(s, g, dY, Wb, mask, dX, prof);
    GemmArgs a = base_args(g);
    a.A = dY; a.Bm = Wb; a.C = dX; a.mask = mask;
    a.AH = (int)ceil_div(g.IH, g.S); a.AW = (int)ceil_div(g.IW, g.S);
    a.JH = g.KH / g.S; a.JW = g.KW / g.S;
    a.M = g.B * a.AH * a.AW;
    a.total_chunks = a.JH * a.JW * g.OC / BK;
    a.chunks = a.total_chunks;
    if (col_end < 0) col_end = g.IC;
    TS_REQUIRE(0 <= col_begin && col_begin < col_end && col_end <= g.IC, TS_ERR_INVALID_ARG,
               "conv_dgrad: bad column range");
    ProfScope scope(prof, TS_KIND_CONV_DGRAD, s);
    const int bn = g.IC % 64 == 0 ? 64 : 32;
    a.n_tile0 = col_begin / bn;
    const int n_tiles = (int)ceil_div(col_end, bn) - a.n_tile0;
    const bool small = small_rows(a.M, bn, n_tiles * g.S * g.S);
    const int bm = (bn == 64 ? 128 : 256) / (small ? 2 : 1);
    dim3 grid((unsigned)ceil_div(a.M, bm), n_tiles, g.S * g.S);
    if (bn == 64 && !small) hipLaunchKernelGGL((conv_rows_kernel<true, 2, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
    else if (bn == 64) hipLaunchKernelGGL((conv_rows_kernel<true, 1, 1, 2, 2>), grid, dim3(THREADS), 0, s, a);
    else if (!small) hipLaunchKernelGGL((conv_rows_kernel<true, 2, 1, 4, 1>), grid, dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((conv_rows_kernel<true, 1, 1, 4, 1>), grid, dim3(THREADS), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int slab_sum(hipStream_t s, const float* slabs, int nslab, int64_t n, float* out) {
    if (n <= 0) return TS_OK;
    TS_REQUIRE(n % 4 == 0, TS_ERR_INVALID_ARG, "slab_sum: length must be a multiple of 4");
    static const bool few = getenv("TS_SLAB_SUM_FEW") == nullptr || getenv("TS_SLAB_SUM_FEW")[0] != '0';     // A/B switch
    const unsigned g = (unsigned)ceil_div(n, 1024);
    if (few && nslab <= 4 && n >= 65536)
        hipLaunchKernelGGL(slab_sum_few_kernel<4>, dim3(g), dim3(256), 0, s, slabs, nslab, n, out);
    else if (few && nslab <= 8 && n >= 65536)
        hipLaunchKernelGGL(slab_sum_few_kernel<8>, dim3(g), dim3(256), 0, s, slabs, nslab, n, out);
    else if (few && nslab <= 16 && n >= 65536)
        hipLaunchKernelGGL(slab_sum_few_kernel<16>, dim3(g), dim3(256), 0, s, slabs, nslab, n, out);
    else
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, s, slabs, nslab, n, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int chain_backward(hipStream_t s, ts_workspace* ws, int n, const ConvGeom* l, const float* const* x, float* const* dy,
                   const float* const* wb, float* const* slabs, float* const* grad, bool x0_u8) {
    TS_REQUIRE(n >= 1 && n <= 8, TS_ERR_INVALID_ARG, "chain_backward: 1 .. 8 layers");
    hipStream_t side[2];
    if (int rc = side_streams(ws, s, &side[0], &side[1])) return rc;
    for (int i = n - 1; i >= 0; --i) {
        hipStream_t w = i == 0 ? s : side[(n - 1 - i) & 1];
        if (int rc = stream_wait(ws, s, w, i)) return rc;                 // dY_i (and everything before it) is ready
        if (int rc = conv_wgrad(w, l[i], x[i], dy[i], slabs[i], ws, i == 0 && x0_u8)) return rc;
        if (int rc = slab_sum(w, slabs[i], conv_wgrad_splits(l[i]), l[i].param_elems(), grad[i])) return rc;
        if (i > 0)
            if (int rc = conv_dgrad(s, l[i], dy[i], wb[i], x[i], dy[i - 1], ws)) return rc;
    }
    if (int rc = stream_wait(ws, side[0], s, 8)) return rc;
    return stream_wait(ws, side[1], s, 9);
}

int slab_sum_multi(hipStream_t s, const SlabSeg* segs, int nseg) {
    TS_REQUIRE(nseg >= 1 && nseg <= SLAB_SEGS_MAX, TS_ERR_INVALID_ARG, "slab_sum_multi: 1..%d segments", SLAB_SEGS_MAX);
    SlabSegs a{};
    unsigned blocks = 0;
    for (int k = 0; k < SLAB_SEGS_MAX; ++k) {
        a.first_block[k] = blocks;
        if (k < nseg) {
            TS_REQUIRE(segs[k].n % 4 == 0 && segs[k].n > 0, TS_ERR_INVALID_ARG, "slab_sum_multi: lengths must be multiples of 4");
            a.slabs[k] = segs[k].slabs; a.out[k] = segs[k].out; a.n[k] = segs[k].n; a.nslab[k] = segs[k].nslab;
            blocks += (unsigned)ceil_div(segs[k].n, 64);
        } else {
            a.first_block[k] = 0xffffffffu;
        }
    }
    a.first_block[SLAB_SEGS_MAX] = blocks;
    hipLaunchKernelGGL(slab_sum_multi_kernel, dim3(blocks), dim3(256), 0, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace ts

// ---- C ABI: single layers (parity tests, other network shapes) -----------------------------------
namespace {
ts::ConvGeom geom_of(const int64_t* d) {
    ts::ConvGeom g{(int)d[0], (int)d[1], (int)d[2], (int)d[3], (int)d[4], (int)d[5], (int)d[6], 0, 0, (int)d[7]};
    g.OH = (g.IH - g.KH) / g.S + 1;
    g.OW = (g.IW - g.KW) / g.S + 1;
    return g;
}
int check_dims(const int64_t* d) {
    TS_REQUIRE(d != nullptr, TS_ERR_INVALID_ARG, "conv: NULL dims");
    for (int i = 0; i < 8; ++i) TS_REQUIRE(d[i] >= 1 && d[i] < (1 << 24), TS_ERR_INVALID_ARG, "conv: bad dims");
    TS_REQUIRE(d[1] >= d[4] && d[2] >= d[5], TS_ERR_INVALID_ARG, "conv: kernel larger than the input");
    return TS_OK;
}
}  // namespace

extern "C" {

int ts_conv_forward(ts_workspace* ws, const void* x, int x_u8, const float* wb, float* y, const int64_t* h_dims,
                    int relu, ts_stream_t stream) {
    if (int rc = check_dims(h_dims)) return rc;
    TS_REQUIRE(x && wb && y, TS_ERR_INVALID_ARG, "ts_conv_forward: NULL argument");
    const ts::ConvGeom g = geom_of(h_dims);
    float* split = nullptr;
    const int ns = ts::conv_fwd_splits(g);
    if (ns > 1) {
        TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_conv_forward: workspace is NULL");
        if (int rc = ts::ws_reserve(ws, sizeof(float) * (size_t)ns * g.out_elems())) return rc;
        split = static_cast<float*>(ws->base);
    }
    return ts::conv_forward(ts::as_stream(stream), g, static_cast<const float*>(x), wb, y, relu != 0, split, nullptr,
                            x_u8 != 0);
}

int ts_conv_backward(ts_workspace* ws, const void* x, int x_u8, const float* wb, const float* dy, const float* mask,
                     float* d_wb, float* dx, const int64_t* h_dims, ts_stream_t stream) {
    if (int rc = check_dims(h_dims)) return rc;
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_conv_backward: workspace is NULL");
    TS_REQUIRE(x && wb && dy && d_wb, TS_ERR_INVALID_ARG, "ts_conv_backward: NULL argument");
    const ts::ConvGeom g = geom_of(h_dims);
    const int ns = ts::conv_wgrad_splits(g);
    if (int rc = ts::ws_reserve(ws, sizeof(float) * (size_t)ns * g.param_elems())) return rc;
    float* slabs = static_cast<float*>(ws->base);
    hipStream_t s = ts::as_stream(stream);
    if (int rc = ts::conv_wgrad(s, g, static_cast<const float*>(x), dy, slabs, nullptr, x_u8 != 0)) return rc;
    if (int rc = ts::slab_sum(s, slabs, ns, g.param_elems(), d_wb)) return rc;
    if (dx) {
        // conv_dgrad uses ws for event pairs and its own scratch allocation only (never ws->base, which holds the slabs)
        return ts::conv_dgrad(s, g, dy, wb, mask, dx, ws);
    }
    return TS_OK;
}

}  // extern "C"
