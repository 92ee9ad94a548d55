// ts_mlp.hip -- the SAC-family MLPs (Net[256, 256] + head, ReLU; tianshou/utils/net/common.py:90-178,
// continuous.py:144-169,220-238): all three layers of a forward pass, or all input gradients of a backward pass, as ONE
// launch instead of three GEMM launches.
//
// Why: at the C5 batch (4096 rows) a 256 x 256 Linear layer is 0.54 GFLOP -- 4.6 us of MFMA issue on the whole chip --
// but costs 10-15 us as its own launch (launch, im2col tables, first operand round trip, an epilogue in which all 256
// workgroups store at once: DESIGN.md 4.4).  A chain of three dependent layers pays that three times and round-trips
// the 4 MB activations through HBM in between.
//
// How: a workgroup owns 16 rows of the batch for all layers (256 workgroups at B = 4096: one per CU, eight waves = two
// per SIMD; launches that carry several networks use four-wave workgroups, two per CU: see mlp3_fwd_kernel).  Activations stay in LDS ([row][k], pitch K + 4).  What a workgroup has to move is the weights: 720 KB out
// of L2 for 12 MFLOP -- the launch is bound by how fast ONE CU streams them, and the access pattern decides that rate
// (scripts/ubench/l2_stream.hip, one 512-thread workgroup per CU, matrix resident in L2): 16-byte loads in which sixteen
// lanes cover 256 contiguous bytes, or 4-byte loads over 64-byte pieces, reach 57-66 B/clk/CU; the 8-byte loads of the
// round-2 kernel (dwordx2 column pairs) 33-35 B/clk, whatever the number in flight.  So:
//   * a wave owns 64 output columns of a layer (four 16 x 16 tiles of v_mfma_f32_16x16x4_f32) and HALF of the reduction
//     (waves 0-3: the four column blocks over the first half of k, waves 4-7 over the second; narrower layers split k
//     further); lane (n, kq) fetches, for the 16-deep block it is working on, the four weight rows k = 16 b + 4 kq + t
//     with one dwordx4 each at columns 4 n .. 4 n + 3 of the wave's block -- a 16-lane group reads 256 contiguous bytes --
//     and register e of that load is the B operand of tile e, whose columns are {4 n + e}.  (Transposed layers of the
//     backward pass read 16 bytes along k from sixteen weight rows: 64-byte pieces.)  No LDS staging: no two waves of a
//     workgroup share a weight.
//   * the loads run seven blocks (28 KB per wave) ahead of the MFMAs in a ring of eight register stages, and the ring
//     does not drain between layers: while a layer's last blocks are consumed, the freed stages take the first blocks of
//     the NEXT layer's weights (they do not depend on activations), and the first layer's are requested together with
//     the input rows.  The activation operand is read from LDS one block ahead.
//   * the partial sums of the k splits meet in LDS: every wave writes its 16 x 64 block, one barrier, then all 512 threads
//     sum the splits in fixed order, add the bias / apply the ReLU mask and write the layer's output row-major into LDS
//     (next layer's operand) and HBM (16 bytes per lane, whole rows).
// Summation order: per output element the k splits are summed separately, inside a split ascending in 16-blocks, inside
// a block k = 16 b + 4 kq + t in step t (lane quarter kq: a permutation applied to both operands), then split 0 + split 1
// (+ ...) + bias.
#include "ts_mlp.h"

#include <algorithm>
#include <cstdlib>

#include "ts_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int HID = 256;
constexpr int HP = HID + 4;          // LDS pitch of a hidden activation row
constexpr int NST = 8;               // register stages of the weight ring (blocks in flight + 1)
// A workgroup owns 16 RB batch rows (RB = 1 or 2 row blocks: MFMA M tiles that share every weight operand).
constexpr int part_floats(int nw, int rows) { return rows * (nw * 64 + 4 * nw); }      // k-split partial sums: splits x rows x (columns + 4)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifdef TS_MLP_MARKS
__device__ unsigned long long g_mlp_marks[8];
__device__ unsigned long long g_mlp_trace[8][40];        // workgroup 0: per wave, the clock at the start of every block
#define MMARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mlp_marks[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define MTRACE(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < 40) g_mlp_trace[threadIdx.x >> 6][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MMARK(k) do {} while (0)
#define MTRACE(slot) do {} while (0)
#endif

// blockIdx.y = network: networks of one shape on the same input (twin critics, the members of a REDQ ensemble) share a launch
constexpr int MAXN = ts::MLP3_MAX_NETS;
struct MlpArgs {
    const float* x[MAXN];           // input rows per network (the same pointer for networks that share their input)
    const float* wb1[MAXN]; const float* wb2[MAXN]; const float* wb3[MAXN];
    float* h1[MAXN]; float* h2[MAXN]; float* out[MAXN];
    int M, K1;
};

// Epilogue of a layer
enum : int { EP_BIAS_RELU = 0, EP_BIAS = 1, EP_MASK = 2, EP_PLAIN = 3 };

// One 16-deep block of a wave's weights: [k step t][tile e] (forward) or [tile e][k step t] (transposed)
using Stage = f32x4[4];

// The weight stream of one layer for this wave.
//   TRANS = false: out[:, c] = act(sum_k A[:, k] W[k, c] + b[c])       W[k, c] = wb[k * PITCH + c]   (forward);
//                  TPW = 4: tile e = columns c0 + 64 cw + 4 n + e (dwordx4), TPW = 2: c0 + 32 cw + 2 n + e (dwordx2, the
//                  32-column heads)
//   TRANS = true : out[:, c] = sum_k A[:, k] W[c0 + c, k] (* mask)      W[r, k] = wb[r * PITCH + k]   (input gradient);
//                  tile e = columns 64 cw + 16 e + n, tiles past `ncols` repeat the last one and are not read back
// `ncols` columns (a multiple of 16) are spread over ncw = ceil(ncols / (16 TPW)) column blocks (a power of two); the
// remaining factor nks = 8 / ncw splits the reduction (K / 16 blocks, divisible by nks for every shape mlp3_supported
// admits).
template <int PITCH, bool TRANS, int TPW>
struct WStream {
    const float* wb;
    int c0, ncols, ncw, nks, cw, ks, nbw, b0, n, kq;
    int trow[4];                 // TRANS: row of W (= output column) of tile e for this lane (clamped)
    __device__ __forceinline__ WStream(const float* w, int K, int ncols_, int c0_, int wave, int lane, int nw)
        : wb(w), c0(c0_), ncols(ncols_), n(lane & 15), kq(lane >> 4) {
        const int per = 16 * TPW;
        const int want = (ncols + per - 1) / per;                    // 1, 2 or 4 (0: no output at all)
        ncw = want <= 1 ? 1 : (want == 2 ? 2 : 4);
        nks = nw / ncw;
        cw = wave % ncw;
        ks = wave / ncw;
        nbw = ncols > 0 ? K / 16 / nks : 0;
        b0 = ks * nbw;
#pragma unroll
        for (int e = 0; e < 4; ++e) trow[e] = c0 + min(64 * cw + 16 * e, max(ncols - 16, 0)) + n;
    }
    __device__ __forceinline__ void load(int j, Stage& s) const {
        // blocks past the end (the ring runs NST - 1 blocks ahead; a layer without work for this wave) fetch the first rows of
        // the matrix with every lane on the same address: L1 hits that cost no L2 bandwidth.  (An offset select, not a
        // branch: a branch around loads makes every later s_waitcnt conservative.)
        const unsigned keep = j < nbw ? 0xffffffffu : 0u;
        const int k = 16 * (b0 + j) + 4 * kq;
        if (TRANS) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s[e] = *reinterpret_cast<const f32x4*>(wb + ((unsigned)(trow[e] * PITCH + k) & keep));
        } else if (TPW == 4) {
            const unsigned off = (unsigned)(k * PITCH + c0 + 64 * cw + 4 * n) & keep;
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = *reinterpret_cast<const f32x4*>(wb + off + t * PITCH);
        } else {
            const unsigned off = (unsigned)(k * PITCH + c0 + 32 * cw + 2 * n) & keep;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(wb + off + t * PITCH);
                s[t][0] = v[0];
                s[t][1] = v[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // blocks 0 .. NST - 2 into stages 0 .. NST - 2: what a layer expects to find when it starts
    // (in two parts around the kernel's first barrier: the SIMDs issue oldest wave first, and with all seven blocks of the
    // four older waves queued ahead of them the younger waves' first block arrives later still)
    template <int FROM, int TO>
    __device__ __forceinline__ void prime(Stage (&st)[NST]) const {
#pragma unroll
        for (int i = FROM; i < TO; ++i) load(i, st[i]);
    }
    __device__ __forceinline__ void prime_one(int i, Stage (&st)[NST]) const { load(i, st[i]); }
};

struct NoNext {
    __device__ __forceinline__ void prime_one(int, Stage (&)[NST]) const {}
};

// One layer: the wave's share of the products, the k splits summed through `part`, the result in o_lds ([16][HP],
// nullable) and o_g (row pitch o_ld, nullable; rows >= M are not stored).  A = a_lds [16][a_pitch].  The stream `w` must
// have been primed into `st`; while the last blocks are consumed the freed stages take the first blocks of `next`.
// Two barriers: partial sums visible / the output visible (and `part` free again).
template <int PITCH, bool TRANS, int TPW, int EP, int TH, int RB, typename Next>
__device__ __forceinline__ void mlp_layer(const WStream<PITCH, TRANS, TPW>& w, Stage (&st)[NST], const Next& next,
                                          const float* a_lds, int a_pitch, float* part, float* o_lds,
                                          float* __restrict__ o_g, int o_ld, const float* __restrict__ mask,
                                          const float* __restrict__ bias, int m0, int M, int tid, int trace0 = 0) {
    const int n = w.n, kq = w.kq, nbw = w.nbw;
    const int ncl = 16 * TPW * w.ncw;                 // columns this layer computes (ncols rounded up to whole column blocks)
    const int pp = ncl + 4;                           // pitch of a partial-sum row
    // what the finish reads from memory (bias / ReLU mask) is requested now: a load issued after the next layer's
    // prefetches would wait for all of them (loads return in order)
    const int q4 = w.ncols / 4;                       // float4 per output row
    constexpr int ROWS = 16 * RB;
    const int nf4 = ROWS * q4;                        // float4 outputs of the layer: at most RB * 1024 / TH per thread
    constexpr int NQ = RB * 1024 / TH;
    f32x4 epi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = min(tid + q * TH, max(nf4 - 1, 0));
        const int row = q4 > 0 ? i / q4 : 0, c4 = i - row * q4;
        if (EP == EP_MASK) epi[q] = *reinterpret_cast<const f32x4*>(mask + (size_t)min(m0 + row, M - 1) * o_ld + w.c0 + 4 * c4);
        else if (EP == EP_BIAS_RELU || EP == EP_BIAS) epi[q] = *reinterpret_cast<const f32x4*>(bias + w.c0 + 4 * c4);
        else epi[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[RB][4];
#pragma unroll
    for (int mt = 0; mt < RB; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* al = a_lds + n * a_pitch + 16 * w.b0 + 4 * kq;                   // + 16 j   (row = lane & 15 of row block mt)
    // the activation operand of a block is read from LDS one block ahead of the MFMAs that use it (a wave that waits for
    // an LDS round trip before every sixteen MFMAs leaves the matrix pipe idle, and so does its partner on the SIMD)
    f32x4 a_cur[RB];
#pragma unroll
    for (int mt = 0; mt < RB; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(al + 16 * mt * a_pitch);
    auto compute = [&](int j, const Stage& s) {
        MTRACE(trace0 + j);
        f32x4 a_nxt[RB];
#pragma unroll
        for (int mt = 0; mt < RB; ++mt)
            a_nxt[mt] = *reinterpret_cast<const f32x4*>(al + 16 * mt * a_pitch + 16 * min(j + 1, max(nbw - 1, 0)));
        __builtin_amdgcn_sched_barrier(0);
        // every weight operand meets the RB row blocks back to back: one stream of weights per 16 RB rows
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < TPW; ++e)
#pragma unroll
                for (int mt = 0; mt < RB; ++mt) acc[mt][e] = mfma16(a_cur[mt][t], TRANS ? s[e][t] : s[t][e], acc[mt][e]);
#pragma unroll
        for (int mt = 0; mt < RB; ++mt) a_cur[mt] = a_nxt[mt];
        __builtin_amdgcn_sched_barrier(0);
    };
    // ring of NST register stages: block j lives in stage j % NST and is loaded NST - 1 blocks ahead
    int j = 0;
    for (; j + NST <= nbw; j += NST) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            w.load(j + i + NST - 1, st[(i + NST - 1) % NST]);
            compute(j + i, st[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) {         // the last nbw % NST blocks are already in their stages
        if (j + i < nbw) compute(j + i, st[i]);
        next.prime_one(i, st);
    }

    // partial sums of this wave: part[ks][row][column of the layer]
    if (nbw > 0) {
        float* pw = part + (size_t)(w.ks * ROWS) * pp;
#pragma unroll
        for (int mt = 0; mt < RB; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float* pr = pw + (16 * mt + 4 * kq + v) * pp;
                if (TRANS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr[64 * w.cw + 16 * e + n] = acc[mt][e][v];
                } else if (TPW == 4) {
                    *reinterpret_cast<f32x4*>(pr + 64 * w.cw + 4 * n) = f32x4{acc[mt][0][v], acc[mt][1][v], acc[mt][2][v], acc[mt][3][v]};
                } else {
                    *reinterpret_cast<f32x2*>(pr + 32 * w.cw + 2 * n) = f32x2{acc[mt][0][v], acc[mt][1][v]};
                }
            }
    }
    __syncthreads();
    // finish: 16 x ncols outputs, 16 bytes per thread and step, whole rows
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + q * TH;
        if (i < nf4) {
            const int row = i / q4, c4 = i - row * q4;
            f32x4 val = *reinterpret_cast<const f32x4*>(part + row * pp + 4 * c4);
            for (int k = 1; k < w.nks; ++k) val += *reinterpret_cast<const f32x4*>(part + (k * ROWS + row) * pp + 4 * c4);
            const bool live = m0 + row < M;
            if (EP == EP_BIAS_RELU || EP == EP_BIAS) val += epi[q];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (EP == EP_BIAS_RELU) val[t] = fmaxf(val[t], 0.f);
                if (EP == EP_MASK) val[t] = (live && epi[q][t] > 0.f) ? val[t] : 0.f;
            }
            if (o_lds) *reinterpret_cast<f32x4*>(o_lds + row * HP + 4 * c4) = val;
            if (o_g && live) *reinterpret_cast<f32x4*>(o_g + (size_t)(m0 + row) * o_ld + w.c0 + 4 * c4) = val;
        }
    }
    __syncthreads();
}

// x tile / upstream-gradient tile of this workgroup: [16][K] floats -> LDS (rows past M repeat the last row).  Two halves:
// the global loads are issued BEFORE the first weights are requested and committed to LDS after (loads return in order:
// behind 28 weight loads the input rows would arrive last although the barrier waits for them first).
template <int TH, int RB> constexpr int xr_count() { return 16 * RB * 1024 / 4 / TH; }     // rows x 1024 floats / threads / 4
template <int TH, int RB>
__device__ __forceinline__ void load_rows_issue(const float* __restrict__ src, int K, int m0, int M, int tid,
                                                f32x4 (&xr)[xr_count<TH, RB>()]) {
    constexpr int XR = xr_count<TH, RB>();
    const int q4 = K / 4, total = 16 * RB * q4;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int i = min(tid + it * TH, total - 1);
        const int row = i / q4, c = i - row * q4;
        const int m = min(m0 + row, M - 1);
        xr[it] = *reinterpret_cast<const f32x4*>(src + (size_t)m * K + 4 * c);
        if ((it + 1) * TH >= total) break;          // uniform
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int TH, int RB>
__device__ __forceinline__ void load_rows_commit(int K, float* dst, int pitch, int tid, const f32x4 (&xr)[xr_count<TH, RB>()]) {
    constexpr int XR = xr_count<TH, RB>();
    const int q4 = K / 4, total = 16 * RB * q4;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int i = tid + it * TH;
        const int row = i / q4, c = i - row * q4;
        if (i < total) *reinterpret_cast<f32x4*>(dst + row * pitch + 4 * c) = xr[it];
        if ((it + 1) * TH >= total) break;
    }
}

// NW = waves per workgroup.  8: one workgroup per CU, the reduction of a hidden layer split over two wave groups -- single
// launches (256 workgroups at B = 4096).  4: no k split in the hidden layers, 78 KB of LDS -- TWO workgroups per CU, which in a
// multi-network launch (blockIdx.y = network) are the same rows of two different networks: their weight streams and MFMA
// phases interleave on the CU instead of running as two generations.
// RB = 2 (multi-network launches that still fill the chip with 32-row workgroups: the twin critics at B = 4096): ONE
// eight-wave workgroup per CU owns 32 rows of one network -- a CU streams one network's weights for 32 rows instead of two
// networks' weights for 16 rows each, half the bytes through its vector memory path for the same matrix work.  LDS: the
// second hidden activation overlays the input rows (dead after layer 1's products).
template <int N3, int NW, int RB>
__global__ __launch_bounds__(64 * NW) void mlp3_fwd_kernel(MlpArgs a) {
    constexpr int TH = 64 * NW;
    constexpr int ROWS = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * ROWS;
    const int net = blockIdx.y;
    const float* wb1 = a.wb1[net];
    const float* wb2 = a.wb2[net];
    const float* wb3 = a.wb3[net];
    const int xp = a.K1 + 4;
    float* xs = lds;
    const int x_floats = RB == 1 ? ROWS * xp : max(ROWS * xp, ROWS * HP);
    float* h1s = xs + x_floats;
    float* h2s = RB == 1 ? h1s + ROWS * HP : xs;
    float* part = RB == 1 ? h2s + ROWS * HP : h1s + ROWS * HP;
    Stage st[NST];
    constexpr int TP3 = N3 == 32 ? 2 : 4;
    const WStream<HID, false, 4> w1(wb1, a.K1, HID, 0, wave, lane, NW);
    const WStream<HID, false, 4> w2(wb2, HID, HID, 0, wave, lane, NW);
    const WStream<N3, false, TP3> w3(wb3, HID, N3, 0, wave, lane, NW);
    MMARK(0);
    {
        f32x4 xr[xr_count<TH, RB>()];
        load_rows_issue<TH, RB>(a.x[net], a.K1, m0, a.M, tid, xr);
        w1.template prime<0, 2>(st);               // the first weights travel while the input rows do
        load_rows_commit<TH, RB>(a.K1, xs, xp, tid, xr);
    }
    __syncthreads();
    w1.template prime<2, NST - 1>(st);
    MMARK(1);
    mlp_layer<HID, false, 4, EP_BIAS_RELU, TH, RB>(w1, st, w2, xs, xp, part, h1s, a.h1[net], HID, nullptr, wb1 + (size_t)a.K1 * HID,
                                               m0, a.M, tid, 0);
    MMARK(2);
    mlp_layer<HID, false, 4, EP_BIAS_RELU, TH, RB>(w2, st, w3, h1s, HP, part, h2s, a.h2[net], HID, nullptr, wb2 + (size_t)HID * HID,
                                               m0, a.M, tid, 16);
    MMARK(3);
    mlp_layer<N3, false, TP3, EP_BIAS, TH, RB>(w3, st, NoNext{}, h2s, HP, part, nullptr, a.out[net], N3, nullptr,
                                           wb3 + (size_t)HID * N3, m0, a.M, tid, 32);
    MMARK(4);
}

// Input gradients of the same chain: dh2 = (d_out W3^T) * (h2 > 0), dh1 = (dh2 W2^T) * (h1 > 0) and, when asked for,
// dx[:, dx_c0 : dx_c0 + 16 dx_nt] = dh1 W1^T restricted to those input columns (SAC / TD3: the action columns of the
// critic input, ddpg.py / sac.py actor losses).  dh1 / dh2 go to HBM for the weight-gradient GEMMs.
struct BwdArgs {
    const float* d_out[MAXN]; const float* wb1[MAXN]; const float* wb2[MAXN]; const float* wb3[MAXN];
    const float* h1[MAXN]; const float* h2[MAXN];
    float* dh1[MAXN]; float* dh2[MAXN]; float* dx[MAXN];
    int M, K1, dx_c0, dx_nt;
};

template <int N3, int NW, int RB>
__global__ __launch_bounds__(64 * NW) void mlp3_bwd_kernel(BwdArgs a) {
    constexpr int TH = 64 * NW;
    constexpr int ROWS = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * ROWS;
    float* ds = lds;                          // [16][N3 + 4]
    float* g2s = ds + ROWS * (N3 + 4);        // [16][HP]
    float* g1s = g2s + ROWS * HP;
    float* part = g1s + ROWS * HP;
    Stage st[NST];
    const int net = blockIdx.y;
    float* dx = a.dx[net];
    const WStream<N3, true, 4> w3(a.wb3[net], N3, HID, 0, wave, lane, NW);
    const WStream<HID, true, 4> w2(a.wb2[net], HID, HID, 0, wave, lane, NW);
    const WStream<HID, true, 4> w1(a.wb1[net], HID, dx ? 16 * a.dx_nt : 0, a.dx_c0, wave, lane, NW);
    MMARK(0);
    {
        f32x4 xr[xr_count<TH, RB>()];
        load_rows_issue<TH, RB>(a.d_out[net], N3, m0, a.M, tid, xr);
        w3.template prime<0, 2>(st);
        load_rows_commit<TH, RB>(N3, ds, N3 + 4, tid, xr);
    }
    __syncthreads();
    w3.template prime<2, NST - 1>(st);
    MMARK(1);
    mlp_layer<N3, true, 4, EP_MASK, TH, RB>(w3, st, w2, ds, N3 + 4, part, g2s, a.dh2[net], HID, a.h2[net], nullptr, m0, a.M, tid, 0);
    MMARK(2);
    mlp_layer<HID, true, 4, EP_MASK, TH, RB>(w2, st, w1, g2s, HP, part, g1s, a.dh1[net], HID, a.h1[net], nullptr, m0, a.M, tid, 16);
    MMARK(3);
    if (dx != nullptr) {
        mlp_layer<HID, true, 4, EP_PLAIN, TH, RB>(w1, st, NoNext{}, g1s, HP, part, nullptr, dx, a.K1, nullptr, nullptr, m0, a.M, tid, 32);
        MMARK(4);
    }
}

}  // namespace

#ifdef TS_MLP_MARKS
extern "C" int ts_debug_mlp_marks(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_marks), sizeof(unsigned long long) * 8);
}
extern "C" int ts_debug_mlp_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_trace), sizeof(unsigned long long) * 8 * 40);
}
#endif

namespace ts {

bool mlp3_supported(int K1, int hidden, int head_cols) {
    static const bool off = getenv("TS_MLP_PER_LAYER") != nullptr;      // experiments: force the per-layer GEMM path
    return !off && hidden == HID && (head_cols == 32 || head_cols == 64) && K1 % 32 == 0 && K1 >= 32 && K1 <= 1024;
}

namespace {
int n_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
template <typename K>
int allow_lds(K kernel) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
}  // namespace

int mlp3_forward_n(hipStream_t s, int nets, const float* x, int M, int K1, const float* const* wb1, const float* const* wb2,
                   const float* const* wb3, int head_cols, float* const* h1, float* const* h2, float* const* out,
                   ts_workspace* prof) {
    const float* xs[MLP3_MAX_NETS];
    for (int k = 0; k < MLP3_MAX_NETS; ++k) xs[k] = x;
    return mlp3_forward_nx(s, nets, xs, M, K1, wb1, wb2, wb3, head_cols, h1, h2, out, prof);
}

int mlp3_forward_nx(hipStream_t s, int nets, const float* const* xs, int M, int K1, const float* const* wb1, const float* const* wb2,
                    const float* const* wb3, int head_cols, float* const* h1, float* const* h2, float* const* out,
                    ts_workspace* prof) {
    TS_REQUIRE(mlp3_supported(K1, HID, head_cols), TS_ERR_UNSUPPORTED, "mlp3_forward: unsupported shape");
    TS_REQUIRE(nets >= 1 && nets <= MLP3_MAX_NETS, TS_ERR_INVALID_ARG, "mlp3_forward: 1 .. %d networks", MLP3_MAX_NETS);
    TS_REQUIRE(M >= 1 && xs, TS_ERR_INVALID_ARG, "mlp3_forward: bad argument");
    MlpArgs a{};
    a.M = M; a.K1 = K1;
    for (int k = 0; k < nets; ++k) {
        TS_REQUIRE(xs[k] && wb1[k] && wb2[k] && wb3[k] && out[k], TS_ERR_INVALID_ARG, "mlp3_forward: bad argument");
        a.x[k] = xs[k];
        a.wb1[k] = wb1[k]; a.wb2[k] = wb2[k]; a.wb3[k] = wb3[k];
        a.h1[k] = h1 ? h1[k] : nullptr; a.h2[k] = h2 ? h2[k] : nullptr; a.out[k] = out[k];
    }
    // four-wave workgroups (two per CU) when several networks share the launch and two such workgroups fit a CU's LDS
    static const int force_nw = getenv("TS_MLP_NW") ? atoi(getenv("TS_MLP_NW")) : 0;
    constexpr int ROWS = 16;
    // four-wave 16-row workgroups (two per CU) for several networks when two of them fit a CU's LDS, eight-wave 16-row
    // workgroups for one network; TS_MLP_RB=2: 32-row workgroups (eight waves, one per CU) when several networks share the
    // launch and 32-row workgroups still fill the chip
    static const int force_rb = getenv("TS_MLP_RB") ? atoi(getenv("TS_MLP_RB")) : 0;
    const size_t lds32 = sizeof(float) * (size_t)(std::max(32 * (K1 + 4), 32 * HP) + 32 * HP + part_floats(8, 32));
    // Measured (profiles/r05_mlp3_32_row_workgroups.txt): SAC C5 2,619 -> 2,660, TD3 / DDPG +0.6 / +1.1 %, DiscreteSAC
    // 3,724 -> 3,589 updates/s -- the twin launches are at 62 % of the fp32-MFMA issue rate already (2.95 GFLOP in 34 us), not
    // bound by the weight stream; off unless TS_MLP_RB=2.
    const bool two = force_rb == 2 && nets > 1 && (int64_t)nets * ceil_div(M, 32) >= 3 * n_cus() / 4 &&
                     lds32 <= 160 * 1024 && !force_nw;
    const size_t lds4 = sizeof(float) * (size_t)(ROWS * (K1 + 4) + 2 * ROWS * HP + part_floats(4, ROWS));
    const bool four = force_nw ? force_nw == 4 : (nets > 1 && lds4 <= 80 * 1024);
    const size_t lds = two ? lds32 : four ? lds4 : sizeof(float) * (size_t)(ROWS * (K1 + 4) + 2 * ROWS * HP + part_floats(8, ROWS));
    const dim3 grid((unsigned)ceil_div(M, two ? 32 : ROWS), (unsigned)nets);
    ProfScope scope(prof, TS_KIND_CONV_FWD, s);
#define TS_MLP_FWD(N3_, NW_, RB_)                                                                     \
    do {                                                                                              \
        static const int once = allow_lds(&mlp3_fwd_kernel<N3_, NW_, RB_>);                           \
        (void)once;                                                                                   \
        hipLaunchKernelGGL((mlp3_fwd_kernel<N3_, NW_, RB_>), grid, dim3(64 * NW_), lds, s, a);        \
    } while (0)
    if (head_cols == 32) { if (two) TS_MLP_FWD(32, 8, 2); else if (four) TS_MLP_FWD(32, 4, 1); else TS_MLP_FWD(32, 8, 1); }
    else { if (two) TS_MLP_FWD(64, 8, 2); else if (four) TS_MLP_FWD(64, 4, 1); else TS_MLP_FWD(64, 8, 1); }
#undef TS_MLP_FWD
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int mlp3_forward(hipStream_t s, const float* x, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                 int head_cols, float* h1, float* h2, float* out, ts_workspace* prof) {
    return mlp3_forward_n(s, 1, x, M, K1, &wb1, &wb2, &wb3, head_cols, &h1, &h2, &out, prof);
}

int mlp3_backward_n(hipStream_t s, int nets, const float* const* d_out, int M, int K1, const float* const* wb1,
                    const float* const* wb2, const float* const* wb3, int head_cols, const float* const* h1,
                    const float* const* h2, float* const* dh1, float* const* dh2, float* const* dx, int col0, int col1,
                    ts_workspace* prof) {
    TS_REQUIRE(mlp3_supported(K1, HID, head_cols), TS_ERR_UNSUPPORTED, "mlp3_backward: unsupported shape");
    TS_REQUIRE(nets >= 1 && nets <= MLP3_MAX_NETS, TS_ERR_INVALID_ARG, "mlp3_backward: 1 .. %d networks", MLP3_MAX_NETS);
    TS_REQUIRE(M >= 1, TS_ERR_INVALID_ARG, "mlp3_backward: bad argument");
    BwdArgs a{};
    a.M = M; a.K1 = K1;
    const bool want_dx = dx && dx[0];
    for (int k = 0; k < nets; ++k) {
        TS_REQUIRE(d_out[k] && wb1[k] && wb2[k] && wb3[k] && h1[k] && h2[k] && dh1[k] && dh2[k], TS_ERR_INVALID_ARG,
                   "mlp3_backward: bad argument");
        TS_REQUIRE(!want_dx || dx[k], TS_ERR_INVALID_ARG, "mlp3_backward: input gradients for all networks or none");
        a.d_out[k] = d_out[k]; a.wb1[k] = wb1[k]; a.wb2[k] = wb2[k]; a.wb3[k] = wb3[k];
        a.h1[k] = h1[k]; a.h2[k] = h2[k]; a.dh1[k] = dh1[k]; a.dh2[k] = dh2[k]; a.dx[k] = want_dx ? dx[k] : nullptr;
    }
    if (want_dx) {
        TS_REQUIRE(0 <= col0 && col0 < col1 && col1 <= K1, TS_ERR_INVALID_ARG, "mlp3_backward: bad column range");
        a.dx_c0 = col0 / 16 * 16;
        a.dx_nt = (int)ceil_div(col1 - a.dx_c0, 16);
        TS_REQUIRE(a.dx_nt <= 8, TS_ERR_UNSUPPORTED, "mlp3_backward: input-gradient range wider than 128 columns");
    }
    static const int force_nw = getenv("TS_MLP_NW") ? atoi(getenv("TS_MLP_NW")) : 0;
    static const int force_rb = getenv("TS_MLP_RB") ? atoi(getenv("TS_MLP_RB")) : 0;
    constexpr int ROWS = 16;
    const bool two = force_rb == 2 && nets > 1 && (int64_t)nets * ceil_div(M, 32) >= 3 * n_cus() / 4 && !force_nw;
    const bool four = force_nw ? force_nw == 4 : nets > 1;
    const int rows = two ? 32 : ROWS;
    const size_t lds = sizeof(float) * (size_t)(rows * (head_cols + 4) + 2 * rows * HP + part_floats(two ? 8 : four ? 4 : 8, rows));
    const dim3 grid((unsigned)ceil_div(M, rows), (unsigned)nets);
    ProfScope scope(prof, TS_KIND_CONV_DGRAD, s);
#define TS_MLP_BWD(N3_, NW_, RB_)                                                                     \
    do {                                                                                              \
        static const int once = allow_lds(&mlp3_bwd_kernel<N3_, NW_, RB_>);                           \
        (void)once;                                                                                   \
        hipLaunchKernelGGL((mlp3_bwd_kernel<N3_, NW_, RB_>), grid, dim3(64 * NW_), lds, s, a);        \
    } while (0)
    if (head_cols == 32) { if (two) TS_MLP_BWD(32, 8, 2); else if (four) TS_MLP_BWD(32, 4, 1); else TS_MLP_BWD(32, 8, 1); }
    else { if (two) TS_MLP_BWD(64, 8, 2); else if (four) TS_MLP_BWD(64, 4, 1); else TS_MLP_BWD(64, 8, 1); }
#undef TS_MLP_BWD
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int mlp3_backward(hipStream_t s, const float* d_out, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                  int head_cols, const float* h1, const float* h2, float* dh1, float* dh2, float* dx, int col0, int col1,
                  ts_workspace* prof) {
    return mlp3_backward_n(s, 1, &d_out, M, K1, &wb1, &wb2, &wb3, head_cols, &h1, &h2, &dh1, &dh2, &dx, col0, col1, prof);
}

bool mlp3_backward_supported(int K1, int hidden, int head_cols, bool want_dx, int col0, int col1) {
    if (!mlp3_supported(K1, hidden, head_cols)) return false;
    return !want_dx || ceil_div(col1 - col0 / 16 * 16, 16) <= 8;
}

}  // namespace ts
