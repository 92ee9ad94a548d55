// ts_mlp.hip -- the forward pass of the SAC-family MLPs (Net[256, 256] + head, ReLU; tianshou/utils/net/common.py:90-178,
// continuous.py:144-169,220-238) as ONE launch instead of three GEMM launches.
//
// Why: at the C5 batch (4096 rows) a 256 x 256 Linear layer is 0.54 GFLOP -- 4.6 us of MFMA issue on the whole chip --
// but costs 10-15 us as its own launch (launch, im2col tables, first operand round trip, an epilogue in which all 256
// workgroups store at once: DESIGN.md 4.4).  A chain of three dependent layers pays that three times and round-trips
// the 4 MB activations through HBM in between.
//
// How: a workgroup owns 16 rows of the batch for all three layers (256 workgroups at B = 4096: one per CU, eight waves
// = two per SIMD).  Activations stay in LDS ([row][k], pitch K + 4); every wave owns 32 output columns of a hidden
// layer (two 16 x 16 tiles of v_mfma_f32_16x16x4_f32) and streams exactly its share of the weight matrix from L2
// straight into the MFMA B-operand layout: lane (n, kq) loads W[k][col0 + n] for the four k of its quarter -- 64-byte
// segments of four weight rows per load instruction, no LDS staging because no two waves of a workgroup share a
// weight.  Weight traffic is 256 KB per workgroup per hidden layer out of L2 (64 B/clk/CU: half of the MFMA time); the
// loads run three 32-deep k groups (48 registers) ahead of the MFMAs that consume them.
// Summation order: within every 16-wide k block lane quarter kq contributes k = 16 blk + 4 kq + t at step t (a
// permutation of the block applied to both operands).
#include "ts_mlp.h"

#include <cstdlib>

#include "ts_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int ROWS = 16;             // batch rows per workgroup
constexpr int THREADS = 512;         // 8 waves
constexpr int HID = 256;
constexpr int HP = HID + 4;          // LDS pitch of a hidden activation row

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifdef TS_MLP_MARKS
__device__ unsigned long long g_mlp_marks[8];
__device__ unsigned long long g_mlp_trace[8][40];        // workgroup 0: per wave, the clock at the start of every 32-deep group
#define MMARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mlp_marks[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define MTRACE(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < 40) g_mlp_trace[threadIdx.x >> 6][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MMARK(k) do {} while (0)
#define MTRACE(slot) do {} while (0)
#endif

struct MlpArgs {
    const float* x; const float* wb1; const float* wb2; const float* wb3;
    float* h1; float* h2; float* out;
    int M, K1;
};

// Epilogue of a layer
enum : int { EP_BIAS_RELU = 0, EP_BIAS = 1, EP_MASK = 2, EP_PLAIN = 3 };

// One layer for this wave's TPW column tiles (16 columns each, tiles tile0 .. tile0 + TPW - 1 of `nt`):
//   TRANS = false: out[:, c] = act(sum_k A[:, k] W[k, c] + b[c])        W[k, c] = wb[k * PITCH + c]   (forward)
//   TRANS = true : out[:, c] = sum_k A[:, k] W[c0 + c, k] (* mask)       W[r, k] = wb[r * PITCH + k]   (input gradient:
//                  the rows of the layer matrix are contiguous along the contraction, one dwordx4 per four MFMAs)
// A = a_lds [16][a_pitch]; K = contraction length (multiple of 32).  The result goes to o_lds ([16][HP], nullable) and
// to o_g (row pitch o_ld, nullable; rows >= M are not stored).
//   Forward layers with two tiles per wave load COLUMN PAIRS: the wave's 32 columns are split by parity (tile j =
//   columns c0 + 32 wave + 2 n + j), so that lane n fetches both tiles' weights of one k with a single dwordx2 and a
//   16-lane group reads one whole 128-byte line.  Half as many loads in flight per byte matters because a wave can
//   have at most 63 outstanding (vmcnt): with dword loads three 32-deep k groups, with dwordx2 six.
// The weight stream of one layer for this wave: group g = the 32 reduction rows 32 g .. 32 g + 31 of the wave's columns,
// loaded into one register stage ([2 sixteen-row blocks][tile], a float4 = the lane's four k of the block).
//   Forward layers with two tiles per wave load COLUMN PAIRS: the wave's 32 columns are split by parity (tile j =
//   columns c0 + 32 wave + 2 n + j), so that lane n fetches both tiles' weights of one k with a single dwordx2 and a
//   16-lane group reads one whole 128-byte line.  Half as many loads in flight per byte matters because a wave can
//   have at most 63 outstanding (vmcnt): with dword loads three 32-deep k groups, with dwordx2 six.
using Stage = f32x4[2][2];
constexpr int MAX_NST = 6;

template <int PITCH, int TPW, bool TRANS>
struct WStream {
    static constexpr bool PAIR = !TRANS && TPW == 2;
    static constexpr int NST = PAIR ? 6 : 4;                              // register stages = groups in flight + 1
    const float* wb;
    int ng, nt, c0, tile0, n, kq;
    int tj[TPW];
    __device__ __forceinline__ WStream(const float* w, int K, int nt_, int c0_, int wave, int lane)
        : wb(w), ng(K / 32), nt(nt_), c0(c0_), tile0(wave * TPW), n(lane & 15), kq(lane >> 4) {
        // tiles past `nt` (ragged last wave) recompute the last valid tile and are not stored
#pragma unroll
        for (int j = 0; j < TPW; ++j) tj[j] = max(min(tile0 + j, nt - 1), 0);
    }
    __device__ __forceinline__ bool active() const { return tile0 < nt; }      // any columns of this layer for this wave?
    __device__ __forceinline__ void load(int g, Stage& s) const {
        // groups past the end (the ring runs NST - 1 groups ahead; a wave without columns in this layer) fetch the first
        // rows of the matrix with every lane on the same address: L1 hits that cost no L2 bandwidth -- these launches are
        // bound by the weight stream out of L2.  (An offset select, not a branch: a branch around loads makes every later
        // s_waitcnt conservative.)
        const bool live = g < ng && active();
        const unsigned keep = live ? 0xffffffffu : 0u;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            if (PAIR) {
                using f32x2 = __attribute__((ext_vector_type(2))) float;
                const unsigned off = (unsigned)((32 * g + 16 * blk + 4 * kq) * PITCH + c0 + tile0 * 16 + 2 * n) & keep;
                const float* w = wb + off;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(w + t * PITCH);
                    s[blk][0][t] = v[0];
                    s[blk][TPW - 1][t] = v[1];
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (TRANS) {
                    const unsigned off = (unsigned)((c0 + tj[j] * 16 + n) * PITCH + 32 * g + 16 * blk + 4 * kq) & keep;
                    s[blk][j] = *reinterpret_cast<const f32x4*>(wb + off);
                } else {
                    const unsigned off = (unsigned)((32 * g + 16 * blk + 4 * kq) * PITCH + c0 + tj[j] * 16 + n) & keep;
                    const float* w = wb + off;
#pragma unroll
                    for (int t = 0; t < 4; ++t) s[blk][j][t] = w[t * PITCH];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // groups 0 .. NST - 2 into stages 0 .. NST - 2: what a layer expects to find when it starts
    __device__ __forceinline__ void prime(Stage (&st)[MAX_NST]) const {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) load(i, st[i]);
    }
    __device__ __forceinline__ void prime_one(int i, Stage (&st)[MAX_NST]) const {
        if (i < NST - 1) load(i, st[i]);
    }
};

struct NoNext {
    __device__ __forceinline__ void prime_one(int, Stage (&)[MAX_NST]) const {}
};

// One layer for this wave's TPW column tiles (16 columns each, tiles tile0 .. tile0 + TPW - 1 of `nt`):
//   TRANS = false: out[:, c] = act(sum_k A[:, k] W[k, c] + b[c])        W[k, c] = wb[k * PITCH + c]   (forward)
//   TRANS = true : out[:, c] = sum_k A[:, k] W[c0 + c, k] (* mask)       W[r, k] = wb[r * PITCH + k]   (input gradient:
//                  the rows of the layer matrix are contiguous along the contraction, one dwordx4 per four MFMAs)
// A = a_lds [16][a_pitch]; K = contraction length (multiple of 32).  The result goes to o_lds ([16][HP], nullable) and
// to o_g (row pitch o_ld, nullable; rows >= M are not stored).
// The stream `w` must have been primed into `st` (the kernels do that before the barrier the layer's input waits for:
// weights do not depend on activations).  While the last groups are consumed the freed stages take the first groups of
// `next`, the following layer's stream, so that no layer starts with an empty pipeline.
template <int PITCH, int TPW, bool TRANS, int EP, typename Next>
__device__ __forceinline__ void mlp_layer(const WStream<PITCH, TPW, TRANS>& w, Stage (&st)[MAX_NST], const Next& next,
                                          const float* a_lds, int a_pitch, float* o_lds, float* __restrict__ o_g,
                                          int o_ld, const float* __restrict__ mask, int m0, int M, int trace0 = 0) {
    constexpr int NST = WStream<PITCH, TPW, TRANS>::NST;
    constexpr bool PAIR = WStream<PITCH, TPW, TRANS>::PAIR;
    if (!w.active()) {                                           // no columns of this layer for this wave
#pragma unroll
        for (int i = 0; i < MAX_NST - 1; ++i) next.prime_one(i, st);
        return;
    }
    const int n = w.n, kq = w.kq, tile0 = w.tile0, nt = w.nt, c0 = w.c0, ng = w.ng;
    const float* wb = w.wb;
    const int K = 32 * ng;
    f32x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* al = a_lds + n * a_pitch + 4 * kq;                       // + 32 g + 16 blk   (row = lane & 15)
    // what the epilogue reads from memory (bias / ReLU mask) is fetched now: a load issued after the next layer's
    // prefetches would make the epilogue wait for all of them (loads return in order)
    float epi[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int col = PAIR ? c0 + tile0 * 16 + 2 * n + j : c0 + w.tj[j] * 16 + n;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (EP == EP_MASK) epi[j][v] = mask[(size_t)min(m0 + 4 * kq + v, M - 1) * o_ld + col];
            else if ((EP == EP_BIAS_RELU || EP == EP_BIAS) && v == 0) epi[j][0] = wb[(size_t)K * PITCH + col];
            else epi[j][v] = 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the activation operand of a sixteen-deep block is read from LDS one block ahead of the MFMAs that use it (a wave
    // that waits for an LDS round trip before every eight MFMAs issues at 40 % of the matrix rate, and so does its
    // partner on the SIMD: measured, one workgroup alone on the chip takes as long as 256)
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(al);
    auto compute = [&](int g, const Stage& s) {
        MTRACE(trace0 + g);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4 a_nxt = *reinterpret_cast<const f32x4*>(al + min(32 * g + 16 * blk + 16, K - 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < TPW; ++j) acc[j] = mfma16(a_cur[t], s[blk][j][t], acc[j]);
            a_cur = a_nxt;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ring of NST register stages: group g lives in stage g % NST and is loaded NST - 1 groups ahead
    int g = 0;
    for (; g + NST <= ng; g += NST) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            w.load(g + i + NST - 1, st[(i + NST - 1) % NST]);
            compute(g + i, st[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) {         // the last ng % NST groups are already in their stages
        if (g + i < ng) compute(g + i, st[i]);
        next.prime_one(i, st);
    }
#pragma unroll
    for (int i = NST - 1; i < MAX_NST - 1; ++i) next.prime_one(i, st);

#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const bool tile_ok = tile0 + j < nt;                      // (no branch around the epilogue: its loads stay up front)
        const int col = PAIR ? c0 + tile0 * 16 + 2 * n + j : c0 + (tile0 + j) * 16 + n;
        const float bias = (EP == EP_BIAS_RELU || EP == EP_BIAS) ? epi[j][0] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 4 * kq + v;
            const bool live = m0 + row < M;
            float val = acc[j][v] + bias;
            if (EP == EP_BIAS_RELU) val = fmaxf(val, 0.f);
            if (EP == EP_MASK) val = (live && epi[j][v] > 0.f) ? val : 0.f;
            if (o_lds && tile_ok) o_lds[row * HP + col - c0] = val;
            if (o_g && live && tile_ok) o_g[(size_t)(m0 + row) * o_ld + col] = val;
        }
    }
}

// x tile / upstream-gradient tile of this workgroup: [16][K] floats -> LDS (rows past M repeat the last row).  Two halves:
// the global loads are issued BEFORE the first weights are requested and committed to LDS after (loads return in order:
// behind 40 weight loads the input rows would arrive last although the barrier waits for them first).
constexpr int XR = 8;                // 16 rows x 1024 floats / 512 threads / 4
__device__ __forceinline__ void load_rows_issue(const float* __restrict__ src, int K, int m0, int M, int tid, f32x4 (&xr)[XR]) {
    const int q4 = K / 4, total = ROWS * q4;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int i = min(tid + it * THREADS, total - 1);
        const int row = i / q4, c = i - row * q4;
        const int m = min(m0 + row, M - 1);
        xr[it] = *reinterpret_cast<const f32x4*>(src + (size_t)m * K + 4 * c);
        if ((it + 1) * THREADS >= total) break;          // uniform
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void load_rows_commit(int K, float* dst, int pitch, int tid, const f32x4 (&xr)[XR]) {
    const int q4 = K / 4, total = ROWS * q4;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int i = tid + it * THREADS;
        const int row = i / q4, c = i - row * q4;
        if (i < total) *reinterpret_cast<f32x4*>(dst + row * pitch + 4 * c) = xr[it];
        if ((it + 1) * THREADS >= total) break;
    }
}

// [16][HID] tile in LDS (pitch HP) -> rows m0 .. of a [M, HID] matrix, 16 bytes per lane
__device__ __forceinline__ void store_rows(const float* src, float* __restrict__ dst, int m0, int M, int tid) {
    if (!dst) return;
    for (int i = tid; i < ROWS * (HID / 4); i += THREADS) {
        const int row = i / (HID / 4), c = i % (HID / 4);
        if (m0 + row < M)
            *reinterpret_cast<f32x4*>(dst + (size_t)(m0 + row) * HID + 4 * c) = *reinterpret_cast<const f32x4*>(src + row * HP + 4 * c);
    }
}

template <int N3>
__global__ __launch_bounds__(THREADS) void mlp3_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * ROWS;
    const int xp = a.K1 + 4;
    float* xs = lds;
    float* h1s = xs + ROWS * xp;
    float* h2s = h1s + ROWS * HP;
    Stage st[MAX_NST];
    const WStream<HID, 2, false> w1(a.wb1, a.K1, HID / 16, 0, wave, lane);
    const WStream<HID, 2, false> w2(a.wb2, HID, HID / 16, 0, wave, lane);
    const WStream<N3, 1, false> w3(a.wb3, HID, N3 / 16, 0, wave, lane);
    MMARK(0);
    {
        f32x4 xr[XR];
        load_rows_issue(a.x, a.K1, m0, a.M, tid, xr);
        w1.prime(st);                              // the first weights travel while the input rows do
        load_rows_commit(a.K1, xs, xp, tid, xr);
    }
    __syncthreads();
    MMARK(1);
    mlp_layer<HID, 2, false, EP_BIAS_RELU>(w1, st, w2, xs, xp, h1s, nullptr, HID, nullptr, m0, a.M, 0);
    MTRACE(a.K1 / 32);
    MMARK(2);
    __syncthreads();
    MMARK(3);
    store_rows(h1s, a.h1, m0, a.M, tid);
    mlp_layer<HID, 2, false, EP_BIAS_RELU>(w2, st, w3, h1s, HP, h2s, nullptr, HID, nullptr, m0, a.M, a.K1 / 32 + 2);
    MTRACE(a.K1 / 32 + 10);
    __syncthreads();
    MMARK(4);
    store_rows(h2s, a.h2, m0, a.M, tid);
    mlp_layer<N3, 1, false, EP_BIAS>(w3, st, NoNext{}, h2s, HP, nullptr, a.out, N3, nullptr, m0, a.M);
    MMARK(5);
}

// Input gradients of the same chain: dh2 = (d_out W3^T) * (h2 > 0), dh1 = (dh2 W2^T) * (h1 > 0) and, when asked for,
// dx[:, dx_c0 : dx_c0 + 16 dx_nt] = dh1 W1^T restricted to those input columns (SAC / TD3: the action columns of the
// critic input, ddpg.py / sac.py actor losses).  dh1 / dh2 go to HBM for the weight-gradient GEMMs.
struct BwdArgs {
    const float* d_out; const float* wb1; const float* wb2; const float* wb3;
    const float* h1; const float* h2;
    float* dh1; float* dh2; float* dx;
    int M, K1, dx_c0, dx_nt;
};

template <int N3>
__global__ __launch_bounds__(THREADS) void mlp3_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * ROWS;
    float* ds = lds;                          // [16][N3 + 4]
    float* g2s = ds + ROWS * (N3 + 4);        // [16][HP]
    float* g1s = g2s + ROWS * HP;
    Stage st[MAX_NST];
    const WStream<N3, 2, true> w3(a.wb3, N3, HID / 16, 0, wave, lane);
    const WStream<HID, 2, true> w2(a.wb2, HID, HID / 16, 0, wave, lane);
    const WStream<HID, 1, true> w1(a.wb1, HID, a.dx ? a.dx_nt : 0, a.dx_c0, wave, lane);
    {
        f32x4 xr[XR];
        load_rows_issue(a.d_out, N3, m0, a.M, tid, xr);
        w3.prime(st);
        load_rows_commit(N3, ds, N3 + 4, tid, xr);
    }
    __syncthreads();
    mlp_layer<N3, 2, true, EP_MASK>(w3, st, w2, ds, N3 + 4, g2s, nullptr, HID, a.h2, m0, a.M);
    __syncthreads();
    store_rows(g2s, a.dh2, m0, a.M, tid);
    mlp_layer<HID, 2, true, EP_MASK>(w2, st, w1, g2s, HP, g1s, nullptr, HID, a.h1, m0, a.M);
    __syncthreads();
    store_rows(g1s, a.dh1, m0, a.M, tid);
    if (a.dx == nullptr) return;
    mlp_layer<HID, 1, true, EP_PLAIN>(w1, st, NoNext{}, g1s, HP, nullptr, a.dx, a.K1, nullptr, m0, a.M);
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

int mlp3_forward(hipStream_t s, const float* x, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                 int head_cols, float* h1, float* h2, float* out, ts_workspace* prof) {
    TS_REQUIRE(mlp3_supported(K1, HID, head_cols), TS_ERR_UNSUPPORTED, "mlp3_forward: unsupported shape");
    TS_REQUIRE(M >= 1 && x && wb1 && wb2 && wb3 && out, TS_ERR_INVALID_ARG, "mlp3_forward: bad argument");
    MlpArgs a{x, wb1, wb2, wb3, h1, h2, out, M, K1};
    const size_t lds = sizeof(float) * (size_t)(ROWS * (K1 + 4) + 2 * ROWS * HP);
    const dim3 grid((unsigned)ceil_div(M, ROWS));
    ProfScope scope(prof, TS_KIND_CONV_FWD, s);
    if (head_cols == 32) {
        if (lds > 64 * 1024) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp3_fwd_kernel<32>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            (void)attr;
        }
        hipLaunchKernelGGL((mlp3_fwd_kernel<32>), grid, dim3(THREADS), lds, s, a);
    } else {
        if (lds > 64 * 1024) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp3_fwd_kernel<64>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            (void)attr;
        }
        hipLaunchKernelGGL((mlp3_fwd_kernel<64>), grid, dim3(THREADS), lds, s, a);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int mlp3_backward(hipStream_t s, const float* d_out, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                  int head_cols, const float* h1, const float* h2, float* dh1, float* dh2, float* dx, int col0, int col1,
                  ts_workspace* prof) {
    TS_REQUIRE(mlp3_supported(K1, HID, head_cols), TS_ERR_UNSUPPORTED, "mlp3_backward: unsupported shape");
    TS_REQUIRE(M >= 1 && d_out && wb1 && wb2 && wb3 && h1 && h2 && dh1 && dh2, TS_ERR_INVALID_ARG,
               "mlp3_backward: bad argument");
    BwdArgs a{d_out, wb1, wb2, wb3, h1, h2, dh1, dh2, dx, M, K1, 0, 0};
    if (dx) {
        TS_REQUIRE(0 <= col0 && col0 < col1 && col1 <= K1, TS_ERR_INVALID_ARG, "mlp3_backward: bad column range");
        a.dx_c0 = col0 / 16 * 16;
        a.dx_nt = (int)ceil_div(col1 - a.dx_c0, 16);
        TS_REQUIRE(a.dx_nt <= 8, TS_ERR_UNSUPPORTED, "mlp3_backward: input-gradient range wider than 128 columns");
    }
    const size_t lds = sizeof(float) * (size_t)(ROWS * (head_cols + 4) + 2 * ROWS * HP);
    const dim3 grid((unsigned)ceil_div(M, ROWS));
    ProfScope scope(prof, TS_KIND_CONV_DGRAD, s);
    if (head_cols == 32) hipLaunchKernelGGL((mlp3_bwd_kernel<32>), grid, dim3(THREADS), lds, s, a);
    else hipLaunchKernelGGL((mlp3_bwd_kernel<64>), grid, dim3(THREADS), lds, s, a);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

bool mlp3_backward_supported(int K1, int hidden, int head_cols, bool want_dx, int col0, int col1) {
    if (!mlp3_supported(K1, hidden, head_cols)) return false;
    return !want_dx || ceil_div(col1 - col0 / 16 * 16, 16) <= 8;
}

}  // namespace ts
