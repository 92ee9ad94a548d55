// ts_conv2.hip -- second generation of the fp32-MFMA implicit-GEMM layers for gfx950: the large-M regime
// (minibatch 65,536 of the Atari-shape PPO update, examples/atari/atari_ppo.py:106-118 on
// tianshou/env/atari/atari_network.py:60-122; autograd backward at algorithm_base.py:495).
//
// Same three GEMMs as ts_conv.hip, same operand order, same k-sequential summation (forward / dgrad are
// bit-identical to the first-generation kernels), different data movement:
//
//   rows2 (forward, dgrad)   The weight operand of a column block lives in LDS for the whole lifetime of a
//       persistent workgroup (conv1 32 KB, conv2 128 KB, conv3 144 KB; wider layers stream 32-deep slices through a
//       double buffer).  The activation operand never touches LDS: lane (row r, half h) fetches 16 consecutive
//       reduction elements of its row with 16-byte global loads (a 32-deep chunk of a row is one 128-byte line shared
//       by the two halves) and eight v_permlane32_swap turn "lane half h holds k = 16h .. 16h+15" into the MFMA's
//       "k-step j reads k = 2j + h", so the reduction stays in sequential k order.  No barrier in the K loop of the
//       resident form; eight waves per workgroup drift freely over their own row tiles.
//
//   wgrad2   Both operands have the reduction index (output pixel) as their slow axis, so natural row-major copies
//       of 32-pixel chunks are staged through a double-buffered LDS ring (one barrier per chunk, loads for chunk c+2
//       in flight behind the MFMAs of chunk c); 2x2 / 3x1 register tiles per wave, four-wave workgroups, two per CU.
//
// Roofline: fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TF/s); DESIGN.md section 4.4.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ts_common.h"
#include "ts_conv.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int CK = 32;      // reduction elements per chunk

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// n / d for n < 2^31 by multiplication: p = 32 + ceil(log2 d), magic = floor(2^p / d) + 1 (33 bits), q = (n * magic) >> p.
// Exact because n * (magic * d - 2^p) <= n * d < 2^p.
struct Divisor { unsigned long long magic; int shift; int d; };
__device__ __forceinline__ unsigned fastdiv(unsigned n, const Divisor& v) {
    return (unsigned)(((unsigned long long)n * v.magic) >> v.shift);
}
Divisor divisor_of(int d) {
    int l = 0;
    while ((1ll << l) < d) ++l;
    Divisor v;
    v.shift = 32 + l;
    v.magic = (unsigned long long)(((unsigned __int128)1 << v.shift) / (unsigned)d) + 1;
    v.d = d;
    return v;
}

// lanes 0-31 hold k = 0..15 of their row in x[0..15], lanes 32-63 hold k = 16..31:
// afterwards x[2j] (j < 8) is the operand of k-step j and x[2j+1] of k-step 8 + j (lane half h = k parity).
__device__ __forceinline__ void kseq_swap(float (&x)[16]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[2 * j]), __float_as_uint(x[2 * j + 1]), false, false);
        x[2 * j] = __uint_as_float(r[0]);
        x[2 * j + 1] = __uint_as_float(r[1]);
    }
}
__device__ __forceinline__ constexpr int kstep_reg(int j) { return j < 8 ? 2 * j : 2 * (j - 8) + 1; }

// One input-gradient class: input pixels (ih, iw) = (S (a0 + y) + ph, S (c0 + x) + pw), y < na, x < nc, which all receive
// exactly the taps jh in [jh0, jh0 + njh), jw in [jw0, jw0 + njw) (kh = ph + S jh, kw = pw + S jw): a stride-parity
// class cut further by how far the pixel is from the border, so that no row of a class multiplies a tap that falls
// outside the output grid (the plain gather form spends 19 % (conv2) / 40 % (conv3) of its MFMAs on such zeros).
struct DgClass {
    int ph, pw, a0, c0, na, nc, jh0, jw0, njh, njw;
    int M, tiles, first_wg, wgs;
    Divisor plane, wdt;
};
constexpr int MAX_DG_CLASSES = 64;
constexpr int MAX_CHUNKS = 512;          // reduction chunks of 32 per row (K <= 16,384)

struct Rows2Args {
    const void* A;            // forward: layer input (float32 / uint8 NHWC); dgrad: dY
    const float* W;           // forward: Wb[K + 1][OC] (or a transposed copy for linear dgrad); dgrad: Wb
    float* C;
    const float* bias;        // forward: bias row (NULL: none)
    const float* mask;        // dgrad: ReLU mask source (layer input) or NULL
    ts::ConvGeom g;
    int M;                    // rows: forward B*OH*OW, dgrad B*AH*AW (per stride-parity class)
    int K;                    // reduction length: forward KH*KW*IC, dgrad JH*JW*OC
    int N;                    // columns: forward OC, dgrad IC
    int ldw;                  // row pitch of W in floats
    int ldc;                  // forward: row pitch of C
    int relu;
    int tiles;                // forward: workgroup row tiles
    Divisor plane, wdt;       // forward: / (OH*OW), / OW
    const DgClass* classes;   // dgrad: class table (device memory), indexed through blockIdx.x
    int n_classes;
    int ps;                   // dgrad, "pixel-shuffle" form: the S x S stride parities of an input super-pixel (a, c) read the SAME
                              // dY pixels (a - jh, c - jw), each through its own taps -- so they are the COLUMN blocks of one GEMM
                              // (N = S S IC, column (ph S + pw) IC + ic) instead of S S GEMMs of IC columns: every activation
                              // operand fetched from L2 feeds S S times the MFMAs (conv2 of the Nature CNN: 128 columns)
};

// ------------------------------------------------------------------------------------------------
// rows = pixels.  DG: dgrad (gather form, one blockIdx.z per stride-parity class).  U8: uint8 layer input.
// RES: the whole [K][BN] weight block is resident in LDS, otherwise 32-deep slices are double-buffered.
// Eight waves as (8 / WN) x WN; a wave owns TM x TN tiles of 32 x 32.
// ------------------------------------------------------------------------------------------------
template <bool DG, bool U8, bool RES, int TM, int TN, int WN, int WAVES, bool KSEQ>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, (TM * TN >= 4 && WAVES == 8) ? 2 : 4)))
void conv_rows2_kernel(Rows2Args a) {
    constexpr int THREADS = WAVES * 64, WM = WAVES / WN, BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(!(DG && U8), "dgrad reads float32 gradients");
    extern __shared__ __attribute__((aligned(16))) float Bs[];      // RES: [K][BN]; else [2][CK][BN]
    const ts::ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    int nchunks = a.K / CK;
    DgClass cls{};
    int wg = blockIdx.x, n_wg = gridDim.x, M = a.M, tiles = a.tiles, nchunks_ = nchunks;
    Divisor plane = a.plane, wdt = a.wdt;
    if (DG) {
        int z = 0;
        while (z + 1 < a.n_classes && (int)blockIdx.x >= a.classes[z + 1].first_wg) ++z;
        cls = a.classes[z];
        wg = blockIdx.x - cls.first_wg; n_wg = cls.wgs; M = cls.M; tiles = cls.tiles;
        plane = cls.plane; wdt = cls.wdt;
        nchunks_ = cls.njh * cls.njw * g.OC / CK;
    }
    const int ph = cls.ph, pw = cls.pw;
    nchunks = nchunks_;

    // ---- weight block -> LDS (resident form) -------------------------------------------------------
    if (RES) {
        constexpr int UN = 8;                  // loads in flight per thread
        if (!DG) {
            const int row4 = BN / 4, total = a.K * row4;
            for (int e0 = tid; e0 < total; e0 += THREADS * UN) {
                f32x4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = min(e0 + THREADS * u, total - 1);
                    const int k = e / row4, n4 = e - k * row4;
                    const int col = min(n0 + 4 * n4, a.N - 4);
                    v[u] = *reinterpret_cast<const f32x4*>(a.W + (int64_t)k * a.ldw + col);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = e0 + THREADS * u;
                    if (e < total) *reinterpret_cast<f32x4*>(&Bs[4 * e]) = v[u];        // k * BN + 4 * n4 == 4 * e
                }
            }
        } else {
            // Bs[(j, oc)][ic] = Wb[(tap(j), n0 + ic)][oc]; consecutive threads take consecutive ic (conflict-free stores)
            const int oc4n = g.OC / 4;
            const int per_tap = BN * oc4n, total = cls.njh * cls.njw * per_tap;
            for (int e0 = tid; e0 < total; e0 += THREADS * UN) {
                f32x4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = min(e0 + THREADS * u, total - 1);
                    const int j = e / per_tap, rem = e - j * per_tap;
                    const int oc4 = rem / BN, icl = rem - oc4 * BN;
                    const int tjh = j / cls.njw, jh = cls.jh0 + tjh, jw = cls.jw0 + j - tjh * cls.njw;
                    int pph = ph, ppw = pw, ic = n0 + icl;
                    if (a.ps) {                // column -> (parity, input channel)
                        const int par = ic / g.IC;
                        ic -= par * g.IC; pph = par / g.S; ppw = par - pph * g.S;
                    }
                    const int tapk = ((pph + g.S * jh) * g.KW + ppw + g.S * jw) * g.IC;
                    v[u] = *reinterpret_cast<const f32x4*>(a.W + (int64_t)(tapk + ic) * a.ldw + 4 * oc4);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = e0 + THREADS * u;
                    if (e < total) {
                        const int j = e / per_tap, rem = e - j * per_tap;
                        const int oc4 = rem / BN, icl = rem - oc4 * BN;
#pragma unroll
                        for (int x = 0; x < 4; ++x) Bs[(j * g.OC + 4 * oc4 + x) * BN + icl] = v[u][x];
                    }
                }
            }
        }
        __syncthreads();
    }

    const int t_begin = (int)((int64_t)wg * tiles / n_wg);
    const int t_end = (int)((int64_t)(wg + 1) * tiles / n_wg);

    // Byte offset of reduction chunk c inside a row of the activation operand, tabulated once per workgroup (the K loop
    // then needs no division): forward kh * pitch + position inside the (kw, ic) run; dgrad oc0 - (jh OW + jw) OC.
    __shared__ int s_koff[MAX_CHUNKS];
    for (int c = tid; c < nchunks; c += THREADS) {
        const int k0 = c * CK;
        int off;
        if (!DG) {
            const int run = g.KW * g.IC, kh = k0 / run;
            off = kh * g.IW * g.IC + (k0 - kh * run);
        } else {
            const int tp = k0 / g.OC, oc0 = k0 - tp * g.OC;
            const int tjh = tp / cls.njw, jh = cls.jh0 + tjh, jw = cls.jw0 + tp - tjh * cls.njw;
            off = oc0 - (jh * g.OW + jw) * g.OC;
        }
        s_koff[c] = off;
    }
    __syncthreads();
    using ael = typename std::conditional<U8, uint8_t, float>::type;
    const ael* Ap = static_cast<const ael*>(a.A);

    // streamed form: a thread's share of one 32-deep weight slice
    constexpr int BJ = RES ? 1 : (CK * BN / 4) / THREADS;
    f32x4 breg[BJ];
    auto gload_b = [&](int c) {
        if (!RES) {
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                const int e = tid + THREADS * i, kk = e / (BN / 4), n4 = e % (BN / 4);
                const int col = min(n0 + 4 * n4, a.N - 4);
                breg[i] = *reinterpret_cast<const f32x4*>(a.W + (int64_t)(c * CK + kk) * a.ldw + col);
            }
        }
    };
    auto lstore_b = [&](int slot) {
        if (!RES) {
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                const int e = tid + THREADS * i, kk = e / (BN / 4), n4 = e % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[(slot * CK + kk) * BN + 4 * n4]) = breg[i];
            }
        }
    };

    for (int t = t_begin; t < t_end; ++t) {
        const int mrow0 = t * BM + wm * TM * 32;
        // ---- this lane's rows: unsigned 32-bit element offsets from the operand base (tensors hold < 2^31 elements)
        unsigned abase[TM];
        int obase[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = mrow0 + tm * 32 + r;
            const bool ok = m < M;
            const unsigned mc = ok ? m : M - 1;
            const unsigned b = fastdiv(mc, plane), rem = mc - b * plane.d;
            const unsigned y = fastdiv(rem, wdt), x = rem - y * wdt.d;
            if (!DG) {
                abase[tm] = ((b * g.IH + y * g.S) * g.IW + x * g.S) * g.IC + 16 * h;
                obase[tm] = 0;
            } else {
                const unsigned aa = cls.a0 + y, cc = cls.c0 + x;
                abase[tm] = ((b * g.OH + aa) * g.OW + cc) * g.OC + 16 * h;
                obase[tm] = ok ? (int)(((b * g.IH + aa * g.S + ph) * g.IW + cc * g.S + pw) * g.IC) : -1;
            }
        }

        float areg[2][TM][U8 ? 1 : 16];      // float32 input: the operands themselves
        u32x4 araw[2][U8 ? TM : 1];          // uint8 input: 16 packed pixels per row, converted when consumed
        auto load_a = [&](auto bufc, int c) {
            constexpr int buf = decltype(bufc)::value;
            const int koff = s_koff[c];          // wave-uniform
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const ael* p = Ap + (unsigned)(abase[tm] + koff);
                if constexpr (U8) {
                    araw[buf][tm] = *reinterpret_cast<const u32x4*>(p);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
                        areg[buf][tm][4 * i + 0] = v[0]; areg[buf][tm][4 * i + 1] = v[1];
                        areg[buf][tm][4 * i + 2] = v[2]; areg[buf][tm][4 * i + 3] = v[3];
                    }
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

        auto compute = [&](auto bufc, int c) {
            constexpr int buf = decltype(bufc)::value;
            float x[TM][16];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if constexpr (U8) {
                    const u32x4 v = araw[buf][tm];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        x[tm][4 * i + 0] = (float)(v[i] & 0xffu);
                        x[tm][4 * i + 1] = (float)((v[i] >> 8) & 0xffu);
                        x[tm][4 * i + 2] = (float)((v[i] >> 16) & 0xffu);
                        x[tm][4 * i + 3] = (float)(v[i] >> 24);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[tm][i] = areg[buf][tm][i];
                }
                if (KSEQ) kseq_swap(x[tm]);
            }
            // KSEQ: k-step j multiplies k = 2j + h (sequential order); otherwise k = 16h + j (no cross-lane swaps)
            const float* bp = Bs + (RES ? c * CK : (c & 1) * CK) * BN + (KSEQ ? h : 16 * h) * BN + (wn * TN) * 32 + r;
            constexpr int KS = KSEQ ? 2 : 1;
            float bnext[TN];                       // operands of step j + 1 are fetched before the MFMAs of step j issue
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bnext[tn] = bp[tn * 32];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float bv[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[tn] = bnext[tn];
                if (j < 15) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bnext[tn] = bp[KS * (j + 1) * BN + tn * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(x[tm][KSEQ ? kstep_reg(j) : j], bv[tn], acc[tm][tn]);
            }
        };

        constexpr std::integral_constant<int, 0> B0{};
        constexpr std::integral_constant<int, 1> B1{};
        // Two chunks per iteration with no branch inside (a conditional use would let the compiler sink the prefetch
        // loads into it); an odd last chunk is peeled.  sched_barrier: the loads of the next chunk stay ahead of the MFMAs.
        const int npair = nchunks & ~1, clast = max(nchunks - 1, 0);
        if (RES) {
            if (nchunks > 0) load_a(B0, 0);
            for (int c = 0; c < npair; c += 2) {
                load_a(B1, c + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(B0, c);
                load_a(B0, min(c + 2, clast));
                __builtin_amdgcn_sched_barrier(0);
                compute(B1, c + 1);
            }
            if (nchunks & 1) compute(B0, nchunks - 1);
        } else {
            gload_b(0);
            load_a(B0, 0);
            __syncthreads();                       // every wave is done with the previous tile's slices
            lstore_b(0);
            gload_b(min(1, clast));
            for (int c = 0; c < npair; c += 2) {
                __syncthreads();                   // slice c visible; slot 1 free
                lstore_b(1);
                gload_b(min(c + 2, clast));
                load_a(B1, c + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(B0, c);
                __syncthreads();                   // slice c + 1 visible; slot 0 free
                lstore_b(0);                       // (slice min(c + 2, last): harmless when there is no such slice)
                gload_b(min(c + 3, clast));
                load_a(B0, min(c + 2, clast));
                __builtin_amdgcn_sched_barrier(0);
                compute(B1, c + 1);
            }
            if (nchunks & 1) {
                __syncthreads();
                compute(B0, nchunks - 1);
            }
        }

        // ---- epilogue.  Wave-uniform branches only (mask present? tile completely inside the matrix?): a branch per
        // element would serialise every mask load behind its own wait.  Mask values of a 32 x 32 tile are fetched as one
        // batch of 16 loads before they are used.
        auto epilogue = [&](auto maskc, auto fullc) {
            constexpr bool MASK = decltype(maskc)::value, FULL = decltype(fullc)::value;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = n0 + (wn * TN + tn) * 32 + r;
                    const bool col_ok = col < a.N;
                    int ocol = col;            // dgrad: element offset of this column from the row's pixel
                    if (DG && a.ps) {
                        const int par = col / g.IC, pph = par / g.S;
                        ocol = (pph * g.IW + (par - pph * g.S)) * g.IC + (col - par * g.IC);
                    }
                    float bias = 0.f;
                    if (!DG && a.bias) bias = a.bias[col_ok ? col : 0];
                    // mask loads of a tile go out as batches (one dependent round trip each): 16 where the register
                    // budget allows (32-column variants), 8 for the 16-wave 64-column variants
                    constexpr int EB = (TM == 1 && TN == 1) ? 16 : 8;
#pragma unroll
                    for (int x0 = 0; x0 < 16; x0 += EB) {
                        unsigned off[EB];
                        bool ok[EB];
#pragma unroll
                        for (int i = 0; i < EB; ++i) {
                            const int x = x0 + i, row = (x & 3) + 8 * (x >> 2) + 4 * h;
                            if (!DG) {
                                const int m = mrow0 + tm * 32 + row;
                                ok[i] = (FULL || m < M) && col_ok;
                                off[i] = ok[i] ? (unsigned)m * (unsigned)a.ldc + col : 0u;
                            } else {
                                const int ob = __shfl(obase[tm], row, 64);
                                ok[i] = (FULL || ob >= 0) && col_ok;
                                off[i] = ok[i] ? (unsigned)(ob + ocol) : 0u;
                            }
                        }
                        float mk[MASK ? EB : 1];
                        if constexpr (MASK) {
#pragma unroll
                            for (int i = 0; i < EB; ++i) mk[i] = a.mask[off[i]];
                        }
#pragma unroll
                        for (int i = 0; i < EB; ++i) {
                            float v = acc[tm][tn][x0 + i];
                            if (!DG) {
                                v += bias;
                                if (a.relu) v = fmaxf(v, 0.f);
                            }
                            if constexpr (MASK) v = mk[i] > 0.f ? v : 0.f;
                            if (ok[i]) a.C[off[i]] = v;
                        }
                    }
                }
        };
        const bool full = t * BM + BM <= M && n0 + BN <= a.N;
        if (a.mask) {
            if (full) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::true_type{}, std::false_type{});
        } else {
            if (full) epilogue(std::false_type{}, std::true_type{}); else epilogue(std::false_type{}, std::false_type{});
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad2: rows = k, columns = oc, reduction over output pixels m in chunks of 32 through a double-buffered LDS ring.
// ------------------------------------------------------------------------------------------------
struct Wgrad2Args {
    const void* X;            // layer input (float32 / uint8 NHWC)
    const float* dY;
    float* slabs;
    ts::ConvGeom g;
    int M, K;
    int chunks, total_chunks;
    int64_t slab_stride;
    Divisor plane, wdt;
};

template <bool U8, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad2_kernel(Wgrad2Args a) {
    constexpr int THREADS = WM * WN * 64, BKT = WM * TM * 32, BN = WN * TN * 32;
    using xel = typename std::conditional<U8, uint8_t, float>::type;
    constexpr int XPIECE = U8 ? 16 : 4;                       // elements per 16-byte piece
    constexpr int XROW = BKT / XPIECE;                        // pieces per staged row
    constexpr int XI = (CK * XROW + THREADS - 1) / THREADS;   // pieces per thread and chunk
    constexpr int DROW = BN / 4, DI = (CK * DROW + THREADS - 1) / THREADS;
    __shared__ __attribute__((aligned(16))) xel Xs[2][CK * BKT];
    __shared__ __attribute__((aligned(16))) float Ds[2][CK * BN];
    __shared__ int s_row[4][CK];             // im2col row offsets of chunks c .. c + 3 (slot = chunk & 3)
    __shared__ float s_red[THREADS];
    const ts::ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const int kt = blockIdx.x, n0 = blockIdx.y * BN, split = blockIdx.z;
    const int run = g.KW * g.IC, pitch = g.IW * g.IC;

    // this thread's pieces of a staged chunk: (row mm, piece q) -> element offset inside an im2col row.  All global
    // addresses are unsigned 32-bit element offsets from the tensor bases (tensors hold < 2^31 elements).
    unsigned xoff[XI];
    int xmm[XI], xq[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int e = tid + THREADS * i;
        xmm[i] = e / XROW; xq[i] = e - xmm[i] * XROW;
        const int k = kt * BKT + XPIECE * xq[i];
        const int kc = k < a.K ? k : 0;
        const int kh = kc / run;
        xoff[i] = (unsigned)(kh * pitch + (kc - kh * run));
    }
    u32x4 xr[XI];
    f32x4 dr[DI];
    bool dok[DI];
    const xel* Xp = static_cast<const xel*>(a.X);

    auto rowinfo = [&](int c) {
        if (tid < CK) {
            const unsigned m = min(c * CK + tid, a.M - 1);
            const unsigned b = fastdiv(m, a.plane), rem = m - b * a.plane.d;
            const unsigned oh = fastdiv(rem, a.wdt), ow = rem - oh * a.wdt.d;
            s_row[c & 3][tid] = (int)(((b * g.IH + oh * g.S) * g.IW + ow * g.S) * g.IC);
        }
    };
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            if (CK * XROW % THREADS == 0 || xmm[i] < CK)
                xr[i] = *reinterpret_cast<const u32x4*>(Xp + ((unsigned)s_row[c & 3][min(xmm[i], CK - 1)] + xoff[i]));
        }
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const int e = tid + THREADS * i, mm = e / DROW, n4 = e - mm * DROW;
            if (CK * DROW % THREADS == 0 || mm < CK) {
                const int m = c * CK + mm;
                dok[i] = m < a.M;                       // applied when the piece is written to LDS (keeps the load in flight)
                dr[i] = *reinterpret_cast<const f32x4*>(a.dY + ((unsigned)min(m, a.M - 1) * (unsigned)g.OC + n0 + 4 * n4));
            }
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (CK * XROW % THREADS == 0 || xmm[i] < CK)
                *reinterpret_cast<u32x4*>(&Xs[slot][U8 ? xmm[i] * BKT + XPIECE * xq[i]
                                                           : ((XPIECE * xq[i] / 32) * CK + xmm[i]) * 32 + (XPIECE * xq[i]) % 32]) = xr[i];
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const int e = tid + THREADS * i, mm = e / DROW, n4 = e - mm * DROW;
            if (CK * DROW % THREADS == 0 || mm < CK)
                *reinterpret_cast<f32x4*>(&Ds[slot][((4 * n4 / 32) * CK + mm) * 32 + (4 * n4) % 32]) =
                    dok[i] ? dr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
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
    const int c_last = c_end - 1;
    if (c_begin < c_end) {
        rowinfo(c_begin);
        rowinfo(min(c_begin + 1, c_last));
        rowinfo(min(c_begin + 2, c_last));
        __syncthreads();
        gload(c_begin);
        lstore(c_begin & 1);
        gload(min(c_begin + 1, c_last));
    }
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();                           // slot c & 1 complete; slot (c + 1) & 1 no longer read by anyone;
                                                   // row offsets of chunk c + 2 (written one iteration ago) visible
        if (c + 1 < c_end) lstore((c + 1) & 1);
        gload(min(c + 2, c_last));
        if (c + 3 < c_end) rowinfo(c + 3);         // slot (c + 3) & 3 was last read for chunk c - 1, three barriers ago
        // LDS tiles are [32-column tile][pixel][32]: the operands of consecutive k-steps of one wave tile lie 256 bytes
        // apart, so every ds_read carries its address as an immediate offset (row-major [pixel][BKT] put them 2-4 KB
        // apart and cost a v_add per read -- VALU slots that the fp32 MFMA cannot overlap)
        // (uint8 frames keep row-major [pixel][BKT] rows: 16-byte pieces of a tiled layout would collide in the banks)
        constexpr int XT = U8 ? 32 : CK * 32, XS = U8 ? BKT : 32;           // element strides: 32-column tile, pixel row
        const xel* xs = Xs[c & 1] + wm * TM * XT + h * XS + r;
        const float* ds = Ds[c & 1] + (wn * TN * CK + h) * 32 + r;
        xel anext[TM];                             // operands of step j + 1 are fetched before the MFMAs of step j issue
        float bnext[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) anext[tm] = xs[tm * XT];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bnext[tn] = ds[tn * CK * 32];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[tm] = (float)anext[tm];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[tn] = bnext[tn];
            if (j < 15) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) anext[tm] = xs[tm * XT + 2 * (j + 1) * XS];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bnext[tn] = ds[(tn * CK + 2 * (j + 1)) * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(av[tm], bv[tn], acc[tm][tn]);
        }
        if (kt == 0) {
#pragma unroll
            for (int mm = bp; mm < CK; mm += THREADS / BN) bsum += Ds[c & 1][((bn / 32) * CK + mm) * 32 + bn % 32];
        }
    }

    float* out = a.slabs + split * a.slab_stride;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + (wn * TN + tn) * 32 + r;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int kr = kt * BKT + (wm * TM + tm) * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
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

// out[c][r] = in[r][c]  (linear layers: dgrad is a forward pass with the transposed weight matrix)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

int num_cus() {
    static int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess) v = p.multiProcessorCount;
        }
        return v;
    }();
    return n;
}

// mode: -1 never, 0 automatic, 1 whenever the shape allows it (ts_conv_set_generation / TS_CONV_V2)
int& v2_mode_ref() {
    static int m = [] {
        const char* e = getenv("TS_CONV_V2");
        if (!e) return 0;
        return e[0] == '0' ? -1 : (e[0] == '1' ? 1 : 0);
    }();
    return m;
}
int v2_mode() { return v2_mode_ref(); }

constexpr size_t LDS_MAX = 160 * 1024;
// Row counts from which generation 2 wins (scripts/gpu_conv2_check.py bench, profiles/r03_conv2_thresholds.txt): its
// persistent workgroups need a few row tiles each.  Convolutions: 2^18 rows (at the C3 batch of 512 only the first layer
// qualifies, from 2^17 rows; the shapes of the small-batch parity tests all stay on generation 1); linear layers (rows = samples): 16,384 rows of a layer with at
// least 2^16 weights (narrow heads stay on generation 1 at every size).
constexpr int64_t CONV2_MIN_ROWS = 1 << 18;
constexpr int64_t CONV2_FIRST_MIN_ROWS = 1 << 17;
constexpr int64_t LINEAR2_MIN_ROWS = 1 << 14;
constexpr int64_t LINEAR2_MIN_WEIGHTS = 1 << 16;

bool big_enough(const ts::ConvGeom& g, int64_t rows) {
    const bool linear = g.KH == 1 && g.KW == 1 && g.IH == 1 && g.IW == 1;
    if (linear) return rows >= LINEAR2_MIN_ROWS && (int64_t)g.IC * g.OC >= LINEAR2_MIN_WEIGHTS;
    // 32-channel first layers (resident 32 KB weight block, two workgroups per CU) win from 2^17 rows on -- the C3 batch:
    // 512 x 20 x 20 = 204,800 rows, forward 47.5 -> 41.1 us, weight gradient 59.7 -> 50.7 us, DQN 1,218 -> 1,250 updates/s
    static const int64_t first_min = [] { const char* e = getenv("TS_CONV2_FIRST_MIN_ROWS"); return e ? atoll(e) : CONV2_FIRST_MIN_ROWS; }();
    if (g.OC == 32 && g.K() <= 256) return rows >= first_min;
    return rows >= CONV2_MIN_ROWS;
}

// 1 for launches on the workspace's side stream, 0 for everything else (the caller's stream)
int stream_slot(const ts_workspace* ws, hipStream_t s) { return (ws->side_ready && s == ws->side) ? 1 : 0; }

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

struct DgPlan {                      // host side of an input-gradient launch
    std::vector<DgClass> classes;
    ts_workspace* ws;
    const ts::ConvGeom* g;
};

// Fills tiles / workgroup shares of the classes for row tiles of `bm`, places the table in a device slot of the workspace
// (re-used while geometry, tile size and workgroup budget repeat) and returns the total workgroup count.
int place_classes(DgPlan& plan, int bm, int budget, hipStream_t s, const DgClass** dev, int* total_wgs) {
    ts_workspace* ws = plan.ws;
    const ts::ConvGeom& g = *plan.g;
    // cost of a class = row tiles x (reduction chunks + a fixed per-tile share: row decode, first loads, epilogue)
    const double overhead = env_int("TS_DG_OVERHEAD", 0);
    const int n = (int)plan.classes.size();
    std::vector<double> cost(n);
    double work = 0.0;
    for (int i = 0; i < n; ++i) {
        DgClass& c = plan.classes[i];
        c.tiles = (int)ts::ceil_div(c.M, bm);
        cost[i] = (double)c.tiles * (c.njh * c.njw * g.OC / CK + overhead);
        work += cost[i];
    }
    // one workgroup per class, the rest of the budget by work (largest remainder first); the total never exceeds the
    // budget: a workgroup beyond what the chip holds at once would run as a second round after the first drains
    const int spare = std::max(0, budget - n);
    std::vector<double> want(n);
    int given = 0;
    for (int i = 0; i < n; ++i) {
        DgClass& c = plan.classes[i];
        want[i] = cost[i] / work * spare;
        c.wgs = 1 + (int)want[i];
        given += (int)want[i];
        want[i] -= (int)want[i];
    }
    for (int left = spare - given; left > 0; --left) {
        int best = 0;
        for (int i = 1; i < n; ++i) if (want[i] > want[best]) best = i;
        ++plan.classes[best].wgs;
        want[best] = -1.0;
    }
    int first = 0;
    for (auto& c : plan.classes) {
        c.wgs = (int)std::max<int64_t>(1, std::min<int64_t>(c.wgs, c.tiles));
        c.first_wg = first;
        first += c.wgs;
    }
    *total_wgs = first;
    const long long key[16] = {g.B, g.IH, g.IW, g.IC, g.KH, g.KW, g.S, g.OH, g.OW, g.OC, bm, budget, (long long)plan.classes.size(), 1};
    if (!ws->dg_tables) {
        TS_HIP_CHECK(hipMalloc(&ws->dg_tables, sizeof(DgClass) * MAX_DG_CLASSES * 16));
    }
    DgClass* base = static_cast<DgClass*>(ws->dg_tables);
    // the slots are split between the workspace's two launch streams: a slot is read and rewritten on one stream only,
    // so the copy below is ordered behind every kernel that still reads the table it replaces
    const int lane = stream_slot(ws, s);
    for (int i = 8 * lane; i < 8 * lane + 8; ++i)
        if (memcmp(ws->dg_key[i], key, sizeof(key)) == 0) { *dev = base + (size_t)i * MAX_DG_CLASSES; return TS_OK; }
    const int slot = 8 * lane + ws->dg_next[lane];
    ws->dg_next[lane] = (ws->dg_next[lane] + 1) % 8;
    memcpy(ws->dg_key[slot], key, sizeof(key));
    DgClass* d = base + (size_t)slot * MAX_DG_CLASSES;
    TS_HIP_CHECK(hipMemcpyAsync(d, plan.classes.data(), sizeof(DgClass) * plan.classes.size(), hipMemcpyHostToDevice, s));
    *dev = d;
    return TS_OK;
}

template <bool DG, bool U8, bool RES, int TM, int TN, int WN, int WAVES, bool KSEQ>
int launch_one(dim3 grid, size_t lds, hipStream_t s, Rows2Args& a, int wgs_per_col, DgPlan* plan) {
    constexpr int BM = (WAVES / WN) * TM * 32;
    if (DG) {
        int total = 0;
        if (int rc = place_classes(*plan, BM, wgs_per_col, s, &a.classes, &total)) return rc;
        a.n_classes = (int)plan->classes.size();
        grid.x = (unsigned)total;
    } else {
        a.tiles = (int)ts::ceil_div(a.M, BM);
        grid.x = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.tiles, wgs_per_col));
    }
    auto kernel = conv_rows2_kernel<DG, U8, RES, TM, TN, WN, WAVES, KSEQ>;
    if (lds > 64 * 1024) {
        TS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(WAVES * 64), lds, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

// Variant choice.  shape = (TN, WN) is fixed by the layer; (waves, TM) and the k order are tuning knobs
// (TS_R2_WAVES / TS_R2_TM / TS_R2_KSEQ override the defaults for experiments).
template <bool DG, bool U8, bool RES, int TN, int WN>
int launch_rows2(dim3 grid, size_t lds, hipStream_t s, Rows2Args& a, int wgs_per_col, int waves, int tm, bool kseq,
                 DgPlan* plan = nullptr) {
    waves = env_int("TS_R2_WAVES", waves); tm = env_int("TS_R2_TM", tm); kseq = env_int("TS_R2_KSEQ", kseq) != 0;
#define TS_R2_CASE(W, T, Q) \
    if (waves == W && tm == T && kseq == Q) return launch_one<DG, U8, RES, T, TN, WN, W, Q>(grid, lds, s, a, wgs_per_col, plan);
    TS_R2_CASE(8, 2, true) TS_R2_CASE(8, 2, false) TS_R2_CASE(16, 1, true) TS_R2_CASE(16, 1, false)
    if constexpr (TN == 1) { TS_R2_CASE(16, 2, true) TS_R2_CASE(16, 2, false) }
#undef TS_R2_CASE
    return ts::fail(TS_ERR_UNSUPPORTED, "conv rows2: variant (waves %d, TM %d) is not instantiated", waves, tm);
}

bool shape_ok_rows(const ts::ConvGeom& g) {
    // 32-bit element offsets inside the kernels (check_geom: < 2^31 elements per tensor); chunk-offset table: K <= 16,384
    return (g.KW * g.IC) % CK == 0 && g.K() % CK == 0 && g.OC % 32 == 0 && g.OH * g.OW < 65536 && g.OW < 65536 &&
           g.IH * g.IW < 65536 && g.in_elems() < (1ll << 31) && g.out_elems() < (1ll << 31) &&
           g.K() <= MAX_CHUNKS * CK && g.OC * (g.KH / std::max(1, g.S)) * (g.KW / std::max(1, g.S)) <= MAX_CHUNKS * CK;
}

}  // namespace

namespace ts {

// Runs of input positions (index a within one stride-parity class) that receive the same tap interval.
struct TapSeg { int lo, n, j0, nj; };
static void tap_segments(int IH, int OH, int KH, int S, int p, std::vector<TapSeg>* out) {
    out->clear();
    const int n = IH - p <= 0 ? 0 : (int)ceil_div(IH - p, S), JH = KH / S;
    for (int a = 0; a < n; ++a) {
        const int lo = std::max(0, a - (OH - 1)), hi = std::min(JH - 1, a);
        const int nj = std::max(0, hi - lo + 1), j0 = nj ? lo : 0;
        if (!out->empty() && out->back().j0 == j0 && out->back().nj == nj) ++out->back().n;
        else out->push_back(TapSeg{a, 1, j0, nj});
    }
}

static int dgrad_classes(const ConvGeom& g, std::vector<DgClass>* out) {
    out->clear();
    std::vector<TapSeg> rows, cols;
    for (int ph = 0; ph < g.S; ++ph) {
        tap_segments(g.IH, g.OH, g.KH, g.S, ph, &rows);
        for (int pw = 0; pw < g.S; ++pw) {
            tap_segments(g.IW, g.OW, g.KW, g.S, pw, &cols);
            for (const TapSeg& r : rows)
                for (const TapSeg& c : cols) {
                    DgClass d{};
                    d.ph = ph; d.pw = pw; d.a0 = r.lo; d.c0 = c.lo; d.na = r.n; d.nc = c.n;
                    d.jh0 = r.j0; d.jw0 = c.j0; d.njh = r.nj; d.njw = c.nj;
                    if (d.njh == 0 || d.njw == 0) d.njh = d.njw = 0;          // no window covers these pixels: dX = 0
                    d.M = g.B * r.n * c.n;
                    d.plane = divisor_of(r.n * c.n); d.wdt = divisor_of(c.n);
                    out->push_back(d);
                }
        }
    }
    TS_REQUIRE(!out->empty() && (int)out->size() <= MAX_DG_CLASSES, TS_ERR_UNSUPPORTED,
               "conv2_dgrad: %d border classes (limit %d)", (int)out->size(), MAX_DG_CLASSES);
    return TS_OK;
}

static bool is_linear(const ConvGeom& g) { return g.KH == 1 && g.KW == 1 && g.IH == 1 && g.IW == 1; }

bool conv2_use_forward(const ConvGeom& g, bool x_u8) {
    const int mode = v2_mode();
    if (mode < 0 || !shape_ok_rows(g)) return false;
    if (x_u8 && g.OC != 32) return false;                               // uint8 frames: first layer only (32 channels)
    return mode > 0 || big_enough(g, (int64_t)g.B * g.OH * g.OW);
}

// Pixel-shuffle form of a strided layer's input gradient (Rows2Args.ps): S S IC = 128 columns, the parities' tap runs coincide
// (IH, IW multiples of S: every parity of a super-pixel row sees the same output rows), the [K / S^2][128] weight block fits LDS.
// conv2 of the Nature CNN.  TS_DGRAD_PS=0 keeps one GEMM per parity.
static bool dgrad_ps_ok(const ConvGeom& g) {
    if (!(g.S >= 2 && g.IC == 32 && g.S * g.S * g.IC == 128 && g.IH % g.S == 0 && g.IW % g.S == 0 && g.KH % g.S == 0 &&
          g.KW % g.S == 0 && (size_t)(g.KH / g.S) * (g.KW / g.S) * g.OC * 128 * 4 <= LDS_MAX && env_int("TS_DGRAD_PS", 1)))
        return false;
    std::vector<TapSeg> r0, c0, rp, cp;
    tap_segments(g.IH, g.OH, g.KH, g.S, 0, &r0);
    tap_segments(g.IW, g.OW, g.KW, g.S, 0, &c0);
    auto eq = [](const std::vector<TapSeg>& x, const std::vector<TapSeg>& y) {
        if (x.size() != y.size()) return false;
        for (size_t i = 0; i < x.size(); ++i)
            if (x[i].lo != y[i].lo || x[i].n != y[i].n || x[i].j0 != y[i].j0 || x[i].nj != y[i].nj) return false;
        return true;
    };
    for (int p = 1; p < g.S; ++p) {
        tap_segments(g.IH, g.OH, g.KH, g.S, p, &rp);
        tap_segments(g.IW, g.OW, g.KW, g.S, p, &cp);
        if (!eq(r0, rp) || !eq(c0, cp)) return false;
    }
    return true;
}

bool conv2_use_dgrad(const ConvGeom& g, bool have_ws, int col_begin, int col_end) {
    const int mode = v2_mode();
    if (mode < 0 || !shape_ok_rows(g) || g.IC % 32 != 0) return false;
    if (g.KH % g.S != 0 || g.KW % g.S != 0) return false;
    if (col_begin != 0 || (col_end >= 0 && col_end != g.IC)) return false;      // column ranges: first generation
    if (!have_ws) return false;                                         // transposed weights / class tables live there
    if (!is_linear(g)) {
        if (g.IC > 64) return false;
        const size_t lds = (size_t)(g.KH / g.S) * (g.KW / g.S) * g.OC * (g.IC % 64 == 0 ? 64 : 32) * 4;
        if (lds > LDS_MAX) return false;
        // at most MAX_DG_CLASSES border classes: S^2 parities x (<= 2 KH/S + 1 row runs) x (<= 2 KW/S + 1 column runs)
        if ((int64_t)g.S * g.S * (2 * (g.KH / g.S) + 1) * (2 * (g.KW / g.S) + 1) > MAX_DG_CLASSES) {
            std::vector<DgClass> probe;
            if (dgrad_classes(g, &probe) != TS_OK) return false;
        }
    }
    if (mode > 0 || big_enough(g, (int64_t)g.B * g.IH * g.IW)) return true;
    // the pixel-shuffle form pays from fewer rows on (the C3 batch: 512 x 20 x 20 input pixels; TS_DGRAD_PS_MIN_ROWS)
    static const int64_t ps_min = [] { const char* e = getenv("TS_DGRAD_PS_MIN_ROWS"); return e ? atoll(e) : (int64_t)1 << 62; }();
    return !is_linear(g) && (int64_t)g.B * g.IH * g.IW >= ps_min && dgrad_ps_ok(g);
}

bool conv2_use_wgrad(const ConvGeom& g, bool x_u8) {
    const int mode = v2_mode();
    if (mode < 0 || !shape_ok_rows(g)) return false;
    if (x_u8 && g.OC != 32) return false;
    return mode > 0 || big_enough(g, (int64_t)g.B * g.OH * g.OW);
}

// tile plan of wgrad2 (rows of dW x columns per workgroup):
//   0: 256 x 32, 4 waves (2x1 tiles)      1: 512 x 64, 8 waves (2x2)      2: 192 x 64, 4 waves (3x1, 2x2 waves)
//   3: 256 x 64, 8 waves (2x1, 4x2 waves)   4: 256 x 128, 8 waves (2x2, 4x2 waves; OC % 128 == 0) -- 1.5 x the flops per staged byte of 1
static int wgrad2_plan(const ConvGeom& g, int* bkt, int* bn, int* per_cu) {
    if (g.OC % 64 != 0) { *bkt = 256; *bn = 32; *per_cu = 2; return 0; }
    static const char* force = getenv("TS_WGRAD2_PLAN");
    const int k = g.K();
    const int64_t w512 = ceil_div(k, 512) * 512, w192 = ceil_div(k, 192) * 192, w256 = ceil_div(k, 256) * 256;
    // measured at minibatch 65,536: 512-row tiles (plan 1) win whenever they waste < 15 % (fc1: 3136 -> 3584), the
    // 192-row plan wins for conv3 (576 = 3 x 192 against 1024), plan 3 never
    int plan = (w512 * 100 <= (int64_t)k * 115) ? 1 : (w192 <= w256 ? 2 : 3);
    // round 6: 256 x 128 tiles wherever the layer has the columns and 256-row tiles waste < 15 % (fc1 at 65,536 rows: 2.02 -> 1.87 ms,
    // 104 -> 113 TF/s, profiles/r06_wgrad2_plan4.txt): 1.5 x the flops per byte staged through LDS
    if (g.OC % 128 == 0 && w256 * 100 <= (int64_t)k * 115) plan = 4;
    if (force && force[0] >= '1' && force[0] <= '3') plan = force[0] - '0';
    if (force && force[0] == '4' && g.OC % 128 == 0) plan = 4;
    *bn = 64;
    if (plan == 4) { *bkt = 256; *bn = 128; *per_cu = 1; }
    else if (plan == 1) { *bkt = 512; *per_cu = 1; }
    else if (plan == 2) { *bkt = 192; *per_cu = 2; }
    else { *bkt = 256; *per_cu = 1; }
    return plan;
}

int conv2_wgrad_splits(const ConvGeom& g) {
    int bkt, bn, per_cu;
    wgrad2_plan(g, &bkt, &bn, &per_cu);
    const int64_t tiles = ceil_div(g.K(), bkt) * (g.OC / bn);
    const int chunks = (int)ceil_div((int64_t)g.B * g.OH * g.OW, CK);
    // Split count: every workgroup does the same work, so the launch runs in rounds of (resident workgroups); choose the
    // count whose last round is the fullest (fc1: 56 tiles x 9 splits = 504 of 2 x 256 slots, against 224 of 256 for the
    // obvious 4), among counts that leave each split at least 8 chunks and at most ~4 rounds of slabs to sum.
    const int64_t slots = (int64_t)per_cu * 256;
    const int max_splits = (int)std::max<int64_t>(1, std::min<int64_t>(chunks / 8, ceil_div(4 * slots, tiles)));
    auto eff_of = [&](int sp) {
        const int per = (int)ceil_div(chunks, sp);
        if ((int)ceil_div(chunks, per) != sp) return 0.0;               // not reachable with equal shares
        const int64_t wgs = tiles * sp;
        return (double)wgs / (double)(ceil_div(wgs, slots) * slots);
    };
    double top = 0.0;
    for (int sp = 1; sp <= max_splits; ++sp) top = std::max(top, eff_of(sp));
    int best = 1;
    for (int sp = 1; sp <= max_splits; ++sp)
        if (eff_of(sp) >= top - 0.02) { best = sp; break; }              // the fewest splits within 2 % of the fullest
    return best;
}

// Forward through the rows2 kernel.  `W` / `ldw` / `N` describe the weight matrix actually multiplied (the layer's own
// Wb, or a transposed copy when a linear dgrad is expressed as a forward pass; `mask` is that pass's ReLU mask).
static int rows2_forward(hipStream_t s, const ConvGeom& g, const void* X, bool x_u8, const float* W, int ldw, int N,
                         const float* bias, const float* mask, float* Y, bool relu, ts_workspace* prof, int kind) {
    Rows2Args a{};
    a.A = X; a.W = W; a.C = Y; a.bias = bias; a.mask = mask; a.g = g;
    a.M = g.B * g.OH * g.OW; a.K = g.K(); a.N = N; a.ldw = ldw; a.ldc = N; a.relu = relu;
    a.plane = divisor_of(g.OH * g.OW); a.wdt = divisor_of(g.OW);
    ProfScope scope(prof, kind, s);
    const int cus = num_cus();
    const int res_max_nb = env_int("TS_CONV2_RES_MAX_NB", 8);
    if (N == 32 && (size_t)a.K * 32 * 4 <= LDS_MAX) {
        const size_t lds = (size_t)a.K * 32 * 4;
        const int per_cu = lds <= 40 * 1024 ? 2 : 1;
        if (x_u8) return launch_rows2<false, true, true, 1, 1>(dim3(1, 1, 1), lds, s, a, per_cu * cus, 8, 2, false);
        return launch_rows2<false, false, true, 1, 1>(dim3(1, 1, 1), lds, s, a, per_cu * cus, 8, 2, false);
    }
    TS_REQUIRE(!x_u8, TS_ERR_UNSUPPORTED, "conv rows2: uint8 input is instantiated for 32 output channels only");
    if (N % 128 == 0 && (size_t)a.K * 128 * 4 <= LDS_MAX && N / 128 <= res_max_nb) {
        const int nb = N / 128;
        return launch_rows2<false, false, true, 2, 2>(dim3(1, nb, 1), (size_t)a.K * 128 * 4, s, a, std::max(1, cus / nb), 16, 1, false);
    }
    if (N % 64 == 0 && (size_t)a.K * 64 * 4 <= LDS_MAX && N / 64 <= res_max_nb) {
        const int nb = N / 64;
        return launch_rows2<false, false, true, 2, 1>(dim3(1, nb, 1), (size_t)a.K * 64 * 4, s, a, std::max(1, cus / nb), 16, 1, false);
    }
    // streamed weight slices, 256 x 128 workgroup tiles (ragged last column block allowed).  The column blocks of one
    // row range run on different workgroups at the same pace, so the activation rows they share are fetched together.
    TS_REQUIRE(N % 4 == 0 && N >= 4, TS_ERR_UNSUPPORTED, "conv rows2: unsupported streamed shape");
    const int nb = (int)ceil_div(N, 128);
    return launch_rows2<false, false, false, 2, 2>(dim3(1, nb, 1), (size_t)2 * CK * 128 * 4, s, a, std::max(1, cus / nb), 16, 1, false);
}

int conv2_forward(hipStream_t s, const ConvGeom& g, const float* X, const float* Wb, float* Y, bool relu,
                  ts_workspace* prof, bool x_u8) {
    return rows2_forward(s, g, X, x_u8, Wb, g.OC, g.OC, Wb + (int64_t)g.K() * g.OC, nullptr, Y, relu, prof, TS_KIND_CONV_FWD);
}

int conv2_dgrad(hipStream_t s, const ConvGeom& g, const float* dY, const float* Wb, const float* mask, float* dX,
                ts_workspace* ws) {
    if (is_linear(g)) {
        // dX[m, ic] = sum_oc dY[m, oc] Wt[oc, ic]: a forward pass over the transposed weights (no bias)
        TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "conv2_dgrad: workspace (transposed weights) missing");
        const size_t need = sizeof(float) * (size_t)g.IC * g.OC;
        const int k = stream_slot(ws, s);          // per launch stream: the twin critics run concurrently (ts_common.h)
        if (ws->conv_scratch_bytes[k] < need) {
            if (ws->conv_scratch[k]) TS_HIP_CHECK(hipFree(ws->conv_scratch[k]));
            ws->conv_scratch[k] = nullptr; ws->conv_scratch_bytes[k] = 0;
            TS_HIP_CHECK(hipMalloc(&ws->conv_scratch[k], need));
            ws->conv_scratch_bytes[k] = need;
        }
        float* wt = static_cast<float*>(ws->conv_scratch[k]);
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(g.OC, 32), (unsigned)ceil_div(g.IC, 32)), dim3(256), 0, s,
                           Wb, g.IC, g.OC, wt);
        TS_LAUNCH_CHECK();
        const ConvGeom t{g.B, 1, 1, g.OC, 1, 1, 1, 1, 1, g.IC};
        return rows2_forward(s, t, dY, false, wt, g.IC, g.IC, nullptr, mask, dX, false, ws, TS_KIND_CONV_DGRAD);
    }
    Rows2Args a{};
    a.A = dY; a.W = Wb; a.C = dX; a.mask = mask; a.g = g;
    a.N = g.IC; a.ldw = g.OC;
    DgPlan plan;
    plan.ws = ws; plan.g = &g;
    if (int rc = dgrad_classes(g, &plan.classes)) return rc;
    int kmax = 0;
    for (const auto& c : plan.classes) kmax = std::max(kmax, c.njh * c.njw * g.OC);
    a.K = kmax;
    ProfScope scope(ws, TS_KIND_CONV_DGRAD, s);
    const int cus = num_cus();
    if (dgrad_ps_ok(g)) {          // pixel-shuffle form: parity (0, 0)'s classes stand for all S S parities
        std::vector<DgClass> merged;
        for (const auto& c : plan.classes)
            if (c.ph == 0 && c.pw == 0) merged.push_back(c);
        plan.classes.swap(merged);
        a.N = g.S * g.S * g.IC; a.ps = 1;
        return launch_rows2<true, false, true, 2, 2>(dim3(1, 1, 1), (size_t)kmax * 128 * 4, s, a, cus, 16, 1, false, &plan);
    }
    if (g.IC % 64 == 0) {
        const size_t lds = (size_t)kmax * 64 * 4;
        const int nb = g.IC / 64;
        return launch_rows2<true, false, true, 2, 1>(dim3(1, nb, 1), lds, s, a, std::max(1, cus / nb), 16, 1, false, &plan);
    }
    const size_t lds = (size_t)kmax * 32 * 4;
    const int nb = g.IC / 32;
    return launch_rows2<true, false, true, 1, 1>(dim3(1, nb, 1), lds, s, a, std::max(1, cus / nb), 16, 2, false, &plan);
}

int conv2_wgrad(hipStream_t s, const ConvGeom& g, const float* X, const float* dY, float* slabs, ts_workspace* prof,
                bool x_u8) {
    Wgrad2Args a{};
    a.X = X; a.dY = dY; a.slabs = slabs; a.g = g;
    a.M = g.B * g.OH * g.OW; a.K = g.K();
    a.total_chunks = (int)ceil_div(a.M, CK);
    const int nsplit = conv2_wgrad_splits(g);
    a.chunks = (int)ceil_div(a.total_chunks, nsplit);
    a.slab_stride = g.param_elems();
    a.plane = divisor_of(g.OH * g.OW); a.wdt = divisor_of(g.OW);
    int bkt, bn, per_cu;
    const int plan = wgrad2_plan(g, &bkt, &bn, &per_cu);
    dim3 grid((unsigned)ceil_div(a.K, bkt), g.OC / bn, nsplit);
    ProfScope scope(prof, TS_KIND_CONV_WGRAD, s);
    if (plan == 0) {
        if (x_u8) hipLaunchKernelGGL((conv_wgrad2_kernel<true, 4, 1, 2, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_wgrad2_kernel<false, 4, 1, 2, 1>), grid, dim3(256), 0, s, a);
    } else {
        TS_REQUIRE(!x_u8, TS_ERR_UNSUPPORTED, "conv wgrad2: uint8 input is instantiated for 32 output channels only");
        if (plan == 1 && env_int("TS_WGRAD2_W16", 0)) hipLaunchKernelGGL((conv_wgrad2_kernel<false, 8, 2, 2, 1>), grid, dim3(1024), 0, s, a);
        else if (plan == 1) hipLaunchKernelGGL((conv_wgrad2_kernel<false, 8, 1, 2, 2>), grid, dim3(512), 0, s, a);
        else if (plan == 4) hipLaunchKernelGGL((conv_wgrad2_kernel<false, 4, 2, 2, 2>), grid, dim3(512), 0, s, a);
        else if (plan == 2) hipLaunchKernelGGL((conv_wgrad2_kernel<false, 2, 2, 3, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_wgrad2_kernel<false, 4, 2, 2, 1>), grid, dim3(512), 0, s, a);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // namespace ts

extern "C" int ts_conv_set_generation(int mode) {
    const int prev = v2_mode_ref();
    v2_mode_ref() = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
    return prev;
}
