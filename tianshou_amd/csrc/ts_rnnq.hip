// ts_rnnq.hip -- recurrent Q network for DQN on stacked observations (DRQN) for gfx950.
//
// Replaces, on device-resident float32 batches:
//   Recurrent.forward                       tianshou/utils/net/common.py:400-452: fc1 -> nn.LSTM(L layers, batch_first) -> fc2 on
//                                           the last step; training obs [B, T, dim] (T = the buffer's stack_num), no initial
//                                           state; evaluation obs [B, dim] (T = 1) with the carried (hidden, cell) state
//   DiscreteQLearningPolicy.forward         modelfree/dqn.py:101-143 (argmax)
//   DQN._update_with_batch                  dqn.py:381-404 on that network: TD error, Huber / weighted-MSE loss, backward
//                                           through time, clip + Adam (setup: test/discrete/test_drqn.py:79-101)
// torch's LSTM cell (aten/src/ATen/native/RNN.cpp, gate order i, f, g, o): gates = x W_ih^T + b_ih + h W_hh^T + b_hh,
// c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c').
//
// Every matrix product runs on the fp32-MFMA GEMM kernels of ts_conv.hip (a Linear layer is their 1x1 case).  Activations are
// TIME-MAJOR ([T][B][.]) so that each step is one contiguous row block: the input projections of a layer (x_t W_ih for all t)
// are ONE GEMM over T B rows, as are the weight gradients of W_ih and W_hh (the hidden states are stored with a leading h_{-1}
// block, so "h_{t-1} for all t" is contiguous too); only the recurrent products h_{t-1} W_hh and their input gradients are
// per-step GEMMs.  This file adds the cell kernels (forward / backward), the head / loss kernels and the orchestration.
//
// Flat layout (last row of every block = bias): fc1 [k0 + 1, H] | per layer: W_ih [H + 1, 4H] | W_hh [H + 1, 4H] | fc2 [H + 1, 32]
// (k0 = obs_dim rounded up to 32, the padding rows are and stay zero; head columns [0, n_act) = Q, the rest zero).
#include <algorithm>
#include <cstdlib>

#include "ts_common.h"
#include "ts_conv.h"

#pragma clang fp contract(off)

namespace ts {
int adam_step(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, int64_t step,
              double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
}

namespace {

constexpr int HEAD = 32;
constexpr int MAX_LAYERS = 8;

struct RNet {
    ts::ConvGeom fc1, ih0, ih_all, hh_step, head;     // rows: T B, T B, T B, B, B.  ih0 = input projection of layer 0
    int64_t off_fc1, off_ih[MAX_LAYERS], off_hh[MAX_LAYERS], off_head, count;
    int obs, k0, H, L, A, T;
    int has_fc1;          // Recurrent (common.py:372): fc1 in front of the LSTM; RecurrentActorProb / RecurrentCritic
                          // (continuous.py:241, 325): the LSTM reads the observation directly
    int extra, head_in;   // RecurrentCritic: the head reads [h_T | act] (extra = act_dim), padded to a multiple of 32
    int64_t B;
    const ts::ConvGeom& ih(int l) const { return l == 0 ? ih0 : ih_all; }
};

int make_rnet(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t B, int64_t T, RNet* n,
              int has_fc1 = 1, int64_t extra = 0) {
    TS_REQUIRE(obs_dim >= 1 && obs_dim <= 65536 && hidden >= 32 && hidden <= 1024 && hidden % 32 == 0 && layers >= 1 &&
                   layers <= MAX_LAYERS && n_act >= 1 && n_act <= HEAD,
               TS_ERR_INVALID_ARG, "rnnq: obs_dim >= 1, hidden a multiple of 32 in [32, 1024], 1..8 layers, n_act <= 32");
    TS_REQUIRE(B >= 1 && T >= 1 && T <= 4096 && B * T <= (int64_t)1 << 24, TS_ERR_INVALID_ARG, "rnnq: B >= 1, 1 <= T <= 4096, B T <= 2^24");
    n->obs = (int)obs_dim; n->k0 = (n->obs + 31) / 32 * 32; n->H = (int)hidden; n->L = (int)layers; n->A = (int)n_act;
    n->T = (int)T; n->B = B;
    TS_REQUIRE(extra >= 0 && extra <= 4096, TS_ERR_INVALID_ARG, "rnn: 0 <= extra head inputs <= 4096");
    n->has_fc1 = has_fc1 != 0; n->extra = (int)extra;
    const int rows = (int)(B * T), H = n->H;
    n->head_in = n->extra ? (H + n->extra + 31) / 32 * 32 : H;
    n->fc1 = ts::ConvGeom{rows, 1, 1, n->k0, 1, 1, 1, 1, 1, H};
    n->ih_all = ts::ConvGeom{rows, 1, 1, H, 1, 1, 1, 1, 1, 4 * H};
    n->ih0 = n->has_fc1 ? n->ih_all : ts::ConvGeom{rows, 1, 1, n->k0, 1, 1, 1, 1, 1, 4 * H};
    n->hh_step = ts::ConvGeom{(int)B, 1, 1, H, 1, 1, 1, 1, 1, 4 * H};
    n->head = ts::ConvGeom{(int)B, 1, 1, n->head_in, 1, 1, 1, 1, 1, HEAD};
    int64_t o = 0;
    n->off_fc1 = o; if (n->has_fc1) o += n->fc1.param_elems();
    for (int l = 0; l < n->L; ++l) {
        n->off_ih[l] = o; o += n->ih(l).param_elems();
        n->off_hh[l] = o; o += n->ih_all.param_elems();
    }
    n->off_head = o; o += n->head.param_elems();
    n->count = o;
    return TS_OK;
}

size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Carve {
    char* p;
    float* f(size_t n) { float* r = reinterpret_cast<float*>(p); p += al(4 * n); return r; }
};

size_t split_floats(const RNet& n) {
    size_t s = 4;
    for (const ts::ConvGeom* g : {&n.fc1, &n.ih0, &n.ih_all, &n.hh_step, &n.head}) {
        const int ns = ts::conv_fwd_splits(*g);
        if (ns > 1) s = std::max(s, (size_t)ns * g->out_elems());
    }
    return s;
}

size_t slab_floats(const RNet& n) {
    size_t s = 0;
    for (const ts::ConvGeom* g : {&n.fc1, &n.ih0, &n.ih_all, &n.head})
        s = std::max(s, (size_t)ts::conv_wgrad_splits(*g) * g->param_elems());
    return s;
}

// forward activations kept for the backward pass
struct Acts {
    float* x;                      // [T B, k0] time-major padded observations
    float* x1;                     // [T B, H]  fc1 output = input of layer 0
    float* hbuf[MAX_LAYERS];       // [(T + 1) B, H]: block 0 = h_{-1}, block t + 1 = h_t
    float* cbuf[MAX_LAYERS];       // [(T + 1) B, H]: block 0 = c_{-1}
    float* gact[MAX_LAYERS];       // [T B, 4H] gate activations (i, f, g, o)
    float* gih;                    // [T B, 4H] input projections of the current layer
    float* ghh;                    // [B, 4H]   recurrent projection of the current step
    float* out;                    // [B, 32]   head output
    float* hcat;                   // [B, head_in] = [h_T | extra | 0] when the head has extra inputs (RecurrentCritic)
    float* split;
};

size_t acts_bytes(const RNet& n) {
    const size_t rows = (size_t)n.B * n.T, H = n.H, B = n.B;
    return al(4 * rows * n.k0) + al(4 * rows * H) + n.L * (2 * al(4 * (rows + B) * H) + al(4 * rows * 4 * H)) + al(4 * rows * 4 * H) +
           al(4 * B * 4 * H) + al(4 * B * HEAD) + al(4 * B * n.head_in) + al(4 * split_floats(n));
}

Acts take_acts(Carve& c, const RNet& n) {
    const size_t rows = (size_t)n.B * n.T, H = n.H, B = n.B;
    Acts a{};
    a.x = c.f(rows * n.k0);
    a.x1 = c.f(rows * H);
    for (int l = 0; l < n.L; ++l) { a.hbuf[l] = c.f((rows + B) * H); a.cbuf[l] = c.f((rows + B) * H); a.gact[l] = c.f(rows * 4 * H); }
    a.gih = c.f(rows * 4 * H);
    a.ghh = c.f(B * 4 * H);
    a.out = c.f(B * HEAD);
    a.hcat = c.f(B * n.head_in);
    a.split = c.f(split_floats(n));
    return a;
}

// ---- elementwise kernels ---------------------------------------------------------------------------------------------
// x[t][b][j] = obs[b][t][j] (j < obs_dim), 0 for the padding columns
__global__ __launch_bounds__(256) void pad_time_major_kernel(const float* __restrict__ obs, int64_t B, int T, int obs_dim, int k0,
                                                             float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T * k0) return;
    const int j = (int)(i % k0);
    const int64_t row = i / k0, t = row / B, b = row - t * B;
    x[i] = j < obs_dim ? obs[(b * T + t) * obs_dim + j] : 0.f;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// one LSTM step for [B, H] units: gates (pre-activations gih + ghh) -> gact = (i, f, g, o), c = f c_prev + i g, h = o tanh(c)
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gih, const float* __restrict__ ghh,
                                                        const float* __restrict__ c_prev, int64_t B, int H,
                                                        float* __restrict__ gact, float* __restrict__ c, float* __restrict__ h) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int64_t b = idx / H;
    const int j = (int)(idx - b * H);
    const int64_t g0 = b * 4 * H + j;
    const float i = sigmoidf_(gih[g0] + ghh[g0]);
    const float f = sigmoidf_(gih[g0 + H] + ghh[g0 + H]);
    const float g = tanhf(gih[g0 + 2 * H] + ghh[g0 + 2 * H]);
    const float o = sigmoidf_(gih[g0 + 3 * H] + ghh[g0 + 3 * H]);
    gact[g0] = i; gact[g0 + H] = f; gact[g0 + 2 * H] = g; gact[g0 + 3 * H] = o;
    const float cn = f * c_prev[idx] + i * g;
    c[idx] = cn;
    h[idx] = o * tanhf(cn);
}

// backward of one step: dh = dh_out (from the layer above / the head) + dh_rec (from step t + 1), dc carried in `dc`
// (in: d loss / d c_t from step t + 1, out: d loss / d c_{t-1}); writes the pre-activation gate gradients dg [B, 4H].
// last != 0: step T - 1 (no contribution from a later step).
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh_out, const float* __restrict__ dh_rec,
                                                            float* __restrict__ dc, const float* __restrict__ gact,
                                                            const float* __restrict__ c, const float* __restrict__ c_prev,
                                                            int64_t B, int H, int last, float* __restrict__ dg) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int64_t b = idx / H;
    const int j = (int)(idx - b * H);
    const int64_t g0 = b * 4 * H + j;
    const float i = gact[g0], f = gact[g0 + H], g = gact[g0 + 2 * H], o = gact[g0 + 3 * H];
    const float dh = last ? dh_out[idx] : dh_out[idx] + dh_rec[idx];
    const float tc = tanhf(c[idx]);
    const float d_o = dh * tc;
    float dct = dh * o * (1.f - tc * tc);
    if (!last) dct += dc[idx];
    dg[g0] = dct * g * (i * (1.f - i));
    dg[g0 + H] = dct * c_prev[idx] * (f * (1.f - f));
    dg[g0 + 2 * H] = dct * i * (1.f - g * g);
    dg[g0 + 3 * H] = d_o * (o * (1.f - o));
    dc[idx] = dct * f;
}

// ---- a whole LSTM layer in one launch (H = 32, 64, 128) -----------------------------------------------------------------------
// Per step the recurrent product h_{t-1} W_hh is a [B, H] x [H, 4H] GEMM -- 33 MFLOP at the DRQN shape (B = 128, H = 128) --
// and was one GEMM launch + one cell launch per step: 8 launches of ~5 us per layer and pass, three passes per update,
// 51 GEMM launches in all for 1.3 GFLOP.  The recurrence only couples the units of ONE batch row, so a workgroup that owns
// 16 batch rows and ALL 4H gate columns can run the T steps of a layer by itself:
//   * wave w owns the hidden units [16 w, 16 w + 16): its four 16 x 16 accumulator tiles (v_mfma_f32_16x16x4_f32) are the four
//     gates (i, f, g, o) of those units, so the cell arithmetic happens on the accumulator registers of the lane that owns
//     (rows 4 q .. 4 q + 3, unit n) -- c_t never leaves them;
//   * the wave's share of W_hh (H x 64 floats = H VGPRs per lane) is loaded ONCE and stays in registers for all T steps;
//   * h_t goes to a [16][H + 4] LDS tile (the A operand of the next step: one ds_read_b128 per sixteen k) and to HBM together
//     with c_t and the gate activations (the backward pass and the layer above read them).
// k order inside a product: MFMA (j, t) sums k = 16 j + 4 q + t over the lane quarters q; j ascending, t ascending.
using f32x4 = __attribute__((ext_vector_type(4))) float;

// gate functions of the layer kernels on the hardware exp / rcp (v_exp_f32, v_rcp_f32: absolute error <= 2.5e-7, the same
// forms as the PPO step kernel's tanh); the per-step kernels keep expf / tanhf.  128 units x 16 rows x 5 functions per step on
// one CU: the libm forms cost as many issue cycles as the step's 256 MFMAs.
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh_(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f); }

// zero_init: no initial state -- h_{-1} = c_{-1} = 0 are written to block 0 of hbuf / cbuf here (the backward pass reads them)
template <int H>
__global__ __launch_bounds__(H * 4) void lstm_layer_fwd_kernel(const float* __restrict__ gih, const float* __restrict__ whh,
                                                              float* __restrict__ hbuf, float* __restrict__ cbuf,
                                                              float* __restrict__ gact, int B, int T, int zero_init) {
    constexpr int KB = H / 16, HP = H + 4;
    __shared__ __attribute__((aligned(16))) float hs[16 * HP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * 16, unit = 16 * w + n;
    const size_t blk = (size_t)B * H;
    float wreg[KB][4][4];
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) wreg[j][t][e] = whh[(size_t)(16 * j + 4 * q + t) * 4 * H + e * H + unit];
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = whh[(size_t)H * 4 * H + e * H + unit];
    int row[4];
    bool live[4];
    float c[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        live[v] = b0 + 4 * q + v < B;
        row[v] = min(b0 + 4 * q + v, B - 1);
        if (zero_init) {
            c[v] = 0.f;
            hs[(4 * q + v) * HP + unit] = 0.f;
            if (live[v]) { cbuf[(size_t)row[v] * H + unit] = 0.f; hbuf[(size_t)row[v] * H + unit] = 0.f; }
        } else {
            c[v] = cbuf[(size_t)row[v] * H + unit];
            hs[(4 * q + v) * HP + unit] = hbuf[(size_t)row[v] * H + unit];
        }
    }
    for (int t = 0; t < T; ++t) {
        __syncthreads();                                   // h_{t-1} is in hs
        f32x4 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the input projections of this step travel while the products run
        float pre[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[e][v] = gih[((size_t)t * B + row[v]) * 4 * H + e * H + unit];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hs + n * HP + 16 * j + 4 * q);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[tt], wreg[j][tt][e], acc[e], 0, 0, 0);
        }
        __syncthreads();                                   // every wave has read hs
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float gi = fast_sigmoid(pre[0][v] + (acc[0][v] + bias[0]));
            const float gf = fast_sigmoid(pre[1][v] + (acc[1][v] + bias[1]));
            const float gg = fast_tanh_(pre[2][v] + (acc[2][v] + bias[2]));
            const float go = fast_sigmoid(pre[3][v] + (acc[3][v] + bias[3]));
            const float cn = gf * c[v] + gi * gg;
            const float hn = go * fast_tanh_(cn);
            c[v] = cn;
            hs[(4 * q + v) * HP + unit] = hn;
            if (live[v]) {
                float* ga = gact + ((size_t)t * B + row[v]) * 4 * H + unit;
                ga[0] = gi; ga[H] = gf; ga[2 * H] = gg; ga[3 * H] = go;
                cbuf[(size_t)(t + 1) * blk + (size_t)row[v] * H + unit] = cn;
                hbuf[(size_t)(t + 1) * blk + (size_t)row[v] * H + unit] = hn;
            }
        }
    }
}

// Backward through time of the same layer in one launch: per step the cell's backward arithmetic (lstm_cell_bwd_kernel) on
// the registers of the lane that owns (rows 4 q .. 4 q + 3, unit n), the pre-activation gate gradients to HBM (the weight-
// gradient GEMMs and the layer below read them) and to a [16][4H + 4] LDS tile, then dh_{t-1} = dg_t W_hh^T with the wave's 16
// rows of W_hh (4H floats per lane) resident in registers; dc and the recurrent dh never leave the registers.
template <int H>
__global__ __launch_bounds__(H * 4) void lstm_layer_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ whh,
                                                              const float* __restrict__ gact, const float* __restrict__ cbuf,
                                                              float* __restrict__ dg, int B, int T) {
    constexpr int KB = 4 * H / 16, GP = 4 * H + 4;
    __shared__ __attribute__((aligned(16))) float gs[16 * GP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * 16, unit = 16 * w + n;
    const size_t blk = (size_t)B * H;
    // B operand of dh = dg W_hh^T: B[k][n] = W_hh[unit n of this wave][gate column k], k = 16 j + 4 q + t
    f32x4 wreg[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) wreg[j] = *reinterpret_cast<const f32x4*>(whh + (size_t)unit * 4 * H + 16 * j + 4 * q);
    int row[4];
    bool live[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { live[v] = b0 + 4 * q + v < B; row[v] = min(b0 + 4 * q + v, B - 1); }
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 dh_rec = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const size_t i1 = (size_t)row[v] * H + unit;
            const float* ga = gact + ((size_t)t * B + row[v]) * 4 * H + unit;
            const float i = ga[0], f = ga[H], g = ga[2 * H], o = ga[3 * H];
            const float dho = dout[(size_t)t * blk + i1];
            const float dh = last ? dho : dho + dh_rec[v];
            const float tc = fast_tanh_(cbuf[(size_t)(t + 1) * blk + i1]);
            const float d_o = dh * tc;
            float dct = dh * o * (1.f - tc * tc);
            if (!last) dct += dc[v];
            const float d0 = dct * g * (i * (1.f - i));
            const float d1 = dct * cbuf[(size_t)t * blk + i1] * (f * (1.f - f));
            const float d2 = dct * i * (1.f - g * g);
            const float d3 = d_o * (o * (1.f - o));
            dc[v] = dct * f;
            float* gl = gs + (4 * q + v) * GP + unit;
            gl[0] = d0; gl[H] = d1; gl[2 * H] = d2; gl[3 * H] = d3;
            if (live[v]) {
                float* go = dg + ((size_t)t * B + row[v]) * 4 * H + unit;
                go[0] = d0; go[H] = d1; go[2 * H] = d2; go[3 * H] = d3;
            }
        }
        if (t == 0) break;
        __syncthreads();                                   // dg_t is in gs
        dh_rec = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(gs + n * GP + 16 * j + 4 * q);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) dh_rec = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[tt], wreg[j][tt], dh_rec, 0, 0, 0);
        }
        __syncthreads();                                   // every wave has read gs
    }
}

bool lstm_fused_ok(int H) {
    const bool off = getenv("TS_RNN_PER_STEP") != nullptr;      // A/B and tests: one GEMM + one cell launch per step (read per call)
    return !off && (H == 32 || H == 64 || H == 128);
}

template <int H>
void launch_layer_fwd(hipStream_t s, const float* gih, const float* whh, float* hbuf, float* cbuf, float* gact, int B, int T,
                      int zero_init) {
    hipLaunchKernelGGL(lstm_layer_fwd_kernel<H>, dim3((unsigned)ts::ceil_div(B, 16)), dim3(H * 4), 0, s, gih, whh, hbuf, cbuf, gact, B, T,
                       zero_init);
}
template <int H>
void launch_layer_bwd(hipStream_t s, const float* dout, const float* whh, const float* gact, const float* cbuf, float* dg, int B, int T) {
    hipLaunchKernelGGL(lstm_layer_bwd_kernel<H>, dim3((unsigned)ts::ceil_div(B, 16)), dim3(H * 4), 0, s, dout, whh, gact, cbuf, dg, B, T);
}

// q_out[b, a] = head[b, a]; act_out[b] = argmax_a (first maximum, torch.max)
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ head, int64_t B, int A, float* __restrict__ q_out,
                                                       int64_t* __restrict__ act_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    int best = 0;
    float bv = head[b * HEAD];
    for (int a = 0; a < A; ++a) {
        const float v = head[b * HEAD + a];
        if (q_out) q_out[b * A + a] = v;
        if (v > bv) { bv = v; best = a; }
    }
    if (act_out) act_out[b] = best;
}

// DQN._target_q (dqn.py:365-379) from the two head outputs [B, 32]: q_target[b, argmax_a q_online[b, a]] (double Q) or
// max_a q_target[b, a]
__global__ __launch_bounds__(256) void rnn_target_q_kernel(const float* __restrict__ head_online, const float* __restrict__ head_target,
                                                           int64_t B, int A, int is_double, float* __restrict__ out,
                                                           const float* __restrict__ ns_mask, const double* __restrict__ ns_gpow,
                                                           const double* __restrict__ ns_mc) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* sel = is_double ? head_online : head_target;
    float best = sel[b * HEAD];
    int best_a = 0;
    for (int a = 1; a < A; ++a) {
        const float v = sel[b * HEAD + a];
        if (v > best) { best = v; best_a = a; }
    }
    const float tq = head_target[b * HEAD + best_a];
    if (ns_mask == nullptr) { out[b] = tq; return; }
    // the arithmetic half of compute_nstep_return on ts_nstep_coefficients' outputs (as ts_dqn.hip target_q_kernel)
    const float tqm = tq * ns_mask[b];
    const double qd = (double)tqm * ns_gpow[b];
    out[b] = (float)(qd + ns_mc[b]);
}

// TD error, loss and d loss / d head (dqn.py:388-401)
__global__ __launch_bounds__(1024) void td_loss_kernel(const float* __restrict__ head, const int64_t* __restrict__ act,
                                                       const float* __restrict__ ret, const float* __restrict__ weight, int64_t B,
                                                       float huber_delta, float* __restrict__ td, float* __restrict__ d_head,
                                                       float* __restrict__ loss) {
    __shared__ float red[1024];
    const float inv_b = 1.f / (float)B;
    float lsum = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const int a = (int)act[b];
        const float t = ret[b] - head[b * HEAD + a];
        td[b] = t;
        float l, g;
        if (huber_delta > 0.f) {                     // torch.nn.functional.huber_loss(q, returns), mean
            const float ad = fabsf(t);
            if (ad < huber_delta) { l = 0.5f * t * t; g = -t; }
            else { l = huber_delta * (ad - 0.5f * huber_delta); g = t > 0.f ? -huber_delta : huber_delta; }
        } else {                                     // (td_error.pow(2) * weight).mean()
            const float w = weight ? weight[b] : 1.f;
            l = t * t * w;
            g = -2.f * t * w;
        }
        for (int j = 0; j < HEAD; ++j) d_head[b * HEAD + j] = j == a ? g * inv_b : 0.f;
        lsum += l;
    }
    red[threadIdx.x] = lsum;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_b;
}

// hcat[b] = [h[b, 0..H) | extra[b, 0..E) | 0-pad]
__global__ __launch_bounds__(256) void head_concat_kernel(const float* __restrict__ h, const float* __restrict__ extra, int64_t B,
                                                          int H, int E, int W, float* __restrict__ hcat) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * W) return;
    const int64_t b = i / W;
    const int j = (int)(i - b * W);
    hcat[i] = j < H ? h[b * H + j] : (j < H + E ? extra[b * E + (j - H)] : 0.f);
}

// dst[b, 0..H) = src[b, 0..H) of a [B, W] matrix
__global__ __launch_bounds__(256) void take_cols_kernel(const float* __restrict__ src, int64_t B, int H, int W,
                                                        float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    dst[i] = src[b * W + (i - b * H)];
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ---- network passes --------------------------------------------------------------------------------------------------
// h0 / c0 (nullable): initial state [L][B][H]
int forward(hipStream_t s, ts_workspace* ws, const RNet& n, const float* p, const float* obs, const float* h0, const float* c0,
            const Acts& a, const float* extra = nullptr) {
    const int64_t B = n.B, rows = B * n.T;
    const int H = n.H;
    const size_t blk = (size_t)B * H;
    hipLaunchKernelGGL(pad_time_major_kernel, dim3((unsigned)ts::ceil_div(rows * n.k0, 256)), dim3(256), 0, s, obs, B, n.T, n.obs,
                       n.k0, a.x);
    TS_LAUNCH_CHECK();
    if (n.has_fc1)
        if (int rc = ts::conv_forward(s, n.fc1, a.x, p + n.off_fc1, a.x1, false, a.split, ws)) return rc;
    const bool fused = lstm_fused_ok(H);
    for (int l = 0; l < n.L; ++l) {
        const int zero_init = fused && !h0 && !c0;          // the layer kernel writes the zero state itself
        if (h0) TS_HIP_CHECK(hipMemcpyAsync(a.hbuf[l], h0 + (size_t)l * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
        else if (!zero_init) TS_HIP_CHECK(hipMemsetAsync(a.hbuf[l], 0, 4 * blk, s));
        if (c0) TS_HIP_CHECK(hipMemcpyAsync(a.cbuf[l], c0 + (size_t)l * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
        else if (!zero_init) TS_HIP_CHECK(hipMemsetAsync(a.cbuf[l], 0, 4 * blk, s));
        const float* in = l == 0 ? (n.has_fc1 ? a.x1 : a.x) : a.hbuf[l - 1] + blk;
        if (int rc = ts::conv_forward(s, n.ih(l), in, p + n.off_ih[l], a.gih, false, a.split, ws)) return rc;
        if (fused) {                       // the T steps of the layer in one launch
            if (H == 128) launch_layer_fwd<128>(s, a.gih, p + n.off_hh[l], a.hbuf[l], a.cbuf[l], a.gact[l], (int)B, n.T, zero_init);
            else if (H == 64) launch_layer_fwd<64>(s, a.gih, p + n.off_hh[l], a.hbuf[l], a.cbuf[l], a.gact[l], (int)B, n.T, zero_init);
            else launch_layer_fwd<32>(s, a.gih, p + n.off_hh[l], a.hbuf[l], a.cbuf[l], a.gact[l], (int)B, n.T, zero_init);
            TS_LAUNCH_CHECK();
            continue;
        }
        for (int t = 0; t < n.T; ++t) {
            if (int rc = ts::conv_forward(s, n.hh_step, a.hbuf[l] + t * blk, p + n.off_hh[l], a.ghh, false, a.split, ws)) return rc;
            hipLaunchKernelGGL(lstm_cell_kernel, dim3((unsigned)ts::ceil_div((int64_t)blk, 256)), dim3(256), 0, s,
                               a.gih + (size_t)t * 4 * blk, a.ghh, a.cbuf[l] + t * blk, B, H, a.gact[l] + (size_t)t * 4 * blk,
                               a.cbuf[l] + (t + 1) * blk, a.hbuf[l] + (t + 1) * blk);
            TS_LAUNCH_CHECK();
        }
    }
    const float* h_last = a.hbuf[n.L - 1] + (size_t)n.T * blk;
    if (n.extra) {
        TS_REQUIRE(extra != nullptr, TS_ERR_INVALID_ARG, "rnn: this head needs its extra inputs");
        hipLaunchKernelGGL(head_concat_kernel, dim3((unsigned)ts::ceil_div(B * n.head_in, 256)), dim3(256), 0, s, h_last, extra,
                           B, H, n.extra, n.head_in, a.hcat);
        TS_LAUNCH_CHECK();
        h_last = a.hcat;
    }
    return ts::conv_forward(s, n.head, h_last, p + n.off_head, a.out, false, a.split, ws);
}

struct Bwd {
    float* dg;        // [T B, 4H] gate gradients of the current layer (layer l: dg + l * T B 4H -- the weight gradients of a layer
                      // run on a side stream while the layer below already writes its own)
    float* dout;      // [T B, H]  d loss / d (layer output) at every step
    float* din;       // [T B, H]  d loss / d (layer input)
    float* dh_rec;    // [B, H]
    float* dc;        // [B, H]
    float* slabs[3];  // split-K partials of the weight gradients: one buffer per stream that runs them ([0] first side stream,
                      // [1] second side stream, [2] the caller's)
};

size_t bwd_bytes(const RNet& n) {
    const size_t rows = (size_t)n.B * n.T, H = n.H, B = n.B;
    return al(4 * n.L * rows * 4 * H) + 2 * al(4 * rows * H) + 2 * al(4 * B * H) + 3 * al(4 * slab_floats(n));
}

Bwd take_bwd(Carve& c, const RNet& n) {
    const size_t rows = (size_t)n.B * n.T, H = n.H, B = n.B;
    Bwd b;
    b.dg = c.f(n.L * rows * 4 * H);
    b.dout = c.f(rows * H);
    b.din = c.f(rows * H);
    b.dh_rec = c.f(B * H);
    b.dc = c.f(B * H);
    for (int k = 0; k < 3; ++k) b.slabs[k] = c.f(slab_floats(n));
    return b;
}

int wgrad_to(hipStream_t s, ts_workspace* ws, const ts::ConvGeom& g, const float* x, const float* dy, float* slabs, float* out) {
    if (int rc = ts::conv_wgrad(s, g, x, dy, slabs, ws)) return rc;
    return ts::slab_sum(s, slabs, ts::conv_wgrad_splits(g), g.param_elems(), out);
}

// grad[0 .. count) = d loss / d params given d loss / d head output (d_head [B, 32]).
// The chain head -> (backward through time, input gradient) per layer runs down the caller's stream; the weight-gradient
// GEMMs (+ slab sums) of the head and of every layer need only that layer's gate gradients and run beside the chain on the
// workspace's two side streams -- W_ih's on the first, W_hh's on the second, each with its own partial-sum buffer -- (they were
// 40 % of the serial launches of a DRQN update); fc1's, the last one, follows the chain on the caller's stream.
int backward(hipStream_t s, ts_workspace* ws, const RNet& n, const float* p, const Acts& a, const float* d_head, float* grad,
             Bwd bw) {
    const int64_t B = n.B;
    const int H = n.H, T = n.T;
    const size_t blk = (size_t)B * H;
    const unsigned gcell = (unsigned)ts::ceil_div((int64_t)blk, 256);
    hipStream_t w, w2;
    if (int rc = ts::side_streams(ws, s, &w, &w2)) return rc;
    // head: only the last step of the top layer receives a gradient
    const float* h_last = n.extra ? a.hcat : a.hbuf[n.L - 1] + (size_t)T * blk;
    if (T > 1) TS_HIP_CHECK(hipMemsetAsync(bw.dout, 0, 4 * (size_t)(T - 1) * blk, s));
    if (n.extra) {
        // d loss / d [h_T | extra | pad] needs the (then dead) concat buffer: its weight gradient first, on this stream
        if (int rc = wgrad_to(s, ws, n.head, h_last, d_head, bw.slabs[2], grad + n.off_head)) return rc;
        if (int rc = ts::conv_dgrad(s, n.head, d_head, p + n.off_head, nullptr, a.hcat, ws, 0, H)) return rc;
        hipLaunchKernelGGL(take_cols_kernel, dim3(gcell), dim3(256), 0, s, a.hcat, B, H, n.head_in, bw.dout + (size_t)(T - 1) * blk);
        TS_LAUNCH_CHECK();
    } else {
        if (int rc = ts::stream_wait(ws, s, w, 0)) return rc;                 // d_head is ready
        if (int rc = wgrad_to(w, ws, n.head, h_last, d_head, bw.slabs[0], grad + n.off_head)) return rc;
        if (int rc = ts::conv_dgrad(s, n.head, d_head, p + n.off_head, nullptr, bw.dout + (size_t)(T - 1) * blk, ws)) return rc;
    }
    for (int l = n.L - 1; l >= 0; --l) {
        float* dg = bw.dg + (size_t)l * T * 4 * blk;
        if (lstm_fused_ok(H)) {            // backward through time of the layer in one launch
            if (H == 128) launch_layer_bwd<128>(s, bw.dout, p + n.off_hh[l], a.gact[l], a.cbuf[l], dg, (int)B, T);
            else if (H == 64) launch_layer_bwd<64>(s, bw.dout, p + n.off_hh[l], a.gact[l], a.cbuf[l], dg, (int)B, T);
            else launch_layer_bwd<32>(s, bw.dout, p + n.off_hh[l], a.gact[l], a.cbuf[l], dg, (int)B, T);
            TS_LAUNCH_CHECK();
        } else
        for (int t = T - 1; t >= 0; --t) {
            hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(gcell), dim3(256), 0, s, bw.dout + (size_t)t * blk, bw.dh_rec, bw.dc,
                               a.gact[l] + (size_t)t * 4 * blk, a.cbuf[l] + (size_t)(t + 1) * blk, a.cbuf[l] + (size_t)t * blk, B, H,
                               t == T - 1 ? 1 : 0, dg + (size_t)t * 4 * blk);
            TS_LAUNCH_CHECK();
            if (t > 0)
                if (int rc = ts::conv_dgrad(s, n.hh_step, dg + (size_t)t * 4 * blk, p + n.off_hh[l], nullptr, bw.dh_rec, ws)) return rc;
        }
        if (int rc = ts::stream_wait(ws, s, w, 1 + l)) return rc;             // the layer's gate gradients are complete
        if (w2 != s) TS_HIP_CHECK(hipStreamWaitEvent(w2, ws->side_ev[1 + l], 0));   // the same event: one record on `s`
        const float* in = l == 0 ? (n.has_fc1 ? a.x1 : a.x) : a.hbuf[l - 1] + blk;
        if (int rc = wgrad_to(w, ws, n.ih(l), in, dg, bw.slabs[0], grad + n.off_ih[l])) return rc;
        if (int rc = wgrad_to(w2, ws, n.ih_all, a.hbuf[l], dg, bw.slabs[1], grad + n.off_hh[l])) return rc;   // rows t: h_{t-1}
        if (l == 0 && !n.has_fc1) break;                                      // nothing below the first LSTM layer takes a gradient
        if (int rc = ts::conv_dgrad(s, n.ih_all, dg, p + n.off_ih[l], nullptr, bw.din, ws)) return rc;
        std::swap(bw.dout, bw.din);
    }
    if (n.has_fc1)
        if (int rc = wgrad_to(s, ws, n.fc1, a.x, bw.dout, bw.slabs[2], grad + n.off_fc1)) return rc;
    if (int rc = ts::stream_wait(ws, w, s, 15)) return rc;                   // every gradient block is written
    return ts::stream_wait(ws, w2, s, 7);
}

// bounded head outputs: mu = max_action tanh(head) (continuous.py:230-231, 313-314), in place on the first A columns
__global__ __launch_bounds__(256) void head_tanh_kernel(float* __restrict__ head, int64_t B, int A, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * A) return;
    const int64_t b = i / A;
    float* p = head + b * HEAD + (i - b * A);
    *p = scale * tanhf(*p);
}

// d_head[b, 0..32) from d_out[b, 0..A): through the tanh bound when `bounded` (y = scale tanh(z): dz = dy scale (1 - (y/scale)^2))
__global__ __launch_bounds__(256) void head_dout_kernel(const float* __restrict__ d_out, const float* __restrict__ head_y, int64_t B,
                                                        int A, int bounded, float scale, float* __restrict__ d_head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * HEAD) return;
    const int64_t b = i / HEAD;
    const int j = (int)(i - b * HEAD);
    float g = 0.f;
    if (j < A) {
        g = d_out[b * A + j];
        if (bounded) { const float t = head_y[i] / scale; g = g * scale * (1.f - t * t); }
    }
    d_head[i] = g;
}

__global__ __launch_bounds__(256) void head_copy_kernel(const float* __restrict__ head, int64_t B, int A, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * A) return;
    const int64_t b = i / A;
    out[i] = head[b * HEAD + (i - b * A)];
}

}  // namespace

extern "C" {

int ts_rnnq_layout(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t* h_out) {
    TS_REQUIRE(h_out != nullptr, TS_ERR_INVALID_ARG, "ts_rnnq_layout: h_out is NULL");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, 1, 1, &n)) return rc;
    h_out[0] = n.k0; h_out[1] = n.count; h_out[2] = n.off_fc1;
    for (int l = 0; l < n.L; ++l) { h_out[3 + 2 * l] = n.off_ih[l]; h_out[4 + 2 * l] = n.off_hh[l]; }
    h_out[3 + 2 * n.L] = n.off_head;
    return TS_OK;
}

int ts_rnnq_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                    const float* obs, int64_t B, int64_t T, const float* h_in, const float* c_in, float* q_out, int64_t* act_out,
                    float* h_out, float* c_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rnnq_forward: workspace is NULL");
    TS_REQUIRE(params && obs && (q_out || act_out) && (h_in == nullptr) == (c_in == nullptr), TS_ERR_INVALID_ARG,
               "ts_rnnq_forward: bad argument (hidden and cell state come together)");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, B, T, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    if (int rc = ts::ws_reserve(ws, acts_bytes(n) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    const Acts a = take_acts(c, n);
    if (int rc = forward(s, ws, n, params, obs, h_in, c_in, a)) return rc;
    hipLaunchKernelGGL(head_out_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, a.out, B, n.A, q_out, act_out);
    TS_LAUNCH_CHECK();
    const size_t blk = (size_t)B * n.H;
    for (int l = 0; l < n.L; ++l) {
        if (h_out) TS_HIP_CHECK(hipMemcpyAsync(h_out + l * blk, a.hbuf[l] + (size_t)n.T * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
        if (c_out) TS_HIP_CHECK(hipMemcpyAsync(c_out + l * blk, a.cbuf[l] + (size_t)n.T * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
    }
    return TS_OK;
}

static int rnnq_target_impl(ts_workspace* ws, const float* params, const float* params_old, int64_t obs_dim, int64_t hidden,
                            int64_t layers, int64_t n_act, const float* obs_next, int64_t B, int64_t T, int is_double, float* out,
                            const float* ns_mask, const double* ns_gpow, const double* ns_mc, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rnnq_target_q_fused: workspace is NULL");
    TS_REQUIRE(params && obs_next && out, TS_ERR_INVALID_ARG, "ts_rnnq_target_q_fused: NULL argument");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, B, T, &n)) return rc;
    hipStream_t s = ts::as_stream(stream), side;
    if (int rc = ts::ws_reserve(ws, 2 * acts_bytes(n) + 8192)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    const Acts a1 = take_acts(c, n), a2 = take_acts(c, n);
    if (int rc = ts::side_stream(ws, s, &side)) return rc;
    const bool two = params_old != nullptr;
    if (two) {      // Q_target(s') on the side stream beside Q_online(s') (a pass at B = 128 occupies eight CUs)
        if (int rc = ts::stream_wait(ws, s, side, 9)) return rc;
        if (int rc = forward(side, ws, n, params_old, obs_next, nullptr, nullptr, a2)) return rc;
    }
    if (!two || is_double)
        if (int rc = forward(s, ws, n, params, obs_next, nullptr, nullptr, a1)) return rc;
    if (two)
        if (int rc = ts::stream_wait(ws, side, s, 10)) return rc;
    hipLaunchKernelGGL(rnn_target_q_kernel, dim3((unsigned)ts::ceil_div(B, 256)), dim3(256), 0, s, a1.out, two ? a2.out : a1.out, B,
                       n.A, is_double, out, ns_mask, ns_gpow, ns_mc);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_rnnq_target_q_fused(ts_workspace* ws, const float* params, const float* params_old, int64_t obs_dim, int64_t hidden,
                           int64_t layers, int64_t n_act, const float* obs_next, int64_t B, int64_t T, int is_double, float* out,
                           ts_stream_t stream) {
    return rnnq_target_impl(ws, params, params_old, obs_dim, hidden, layers, n_act, obs_next, B, T, is_double, out, nullptr,
                            nullptr, nullptr, stream);
}

int ts_rnnq_target_returns(ts_workspace* ws, const float* params, const float* params_old, int64_t obs_dim, int64_t hidden,
                           int64_t layers, int64_t n_act, const float* obs_next, int64_t B, int64_t T, int is_double,
                           const float* ns_mask, const double* ns_gpow, const double* ns_mc, float* returns_out,
                           ts_stream_t stream) {
    TS_REQUIRE(ns_mask && ns_gpow && ns_mc, TS_ERR_INVALID_ARG, "ts_rnnq_target_returns: NULL coefficient array");
    return rnnq_target_impl(ws, params, params_old, obs_dim, hidden, layers, n_act, obs_next, B, T, is_double, returns_out,
                            ns_mask, ns_gpow, ns_mc, stream);
}

static int rnnq_update_impl(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                   int64_t hidden, int64_t layers, int64_t n_act, const float* obs, const int64_t* act, const float* returns,
                   const float* weight, int64_t B, int64_t T, const ts_dqn_hparams* hp, float* td_out, float* loss_out,
                   float* grad_out, ts_stream_t stream, void* cache) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rnnq_update: workspace is NULL");
    TS_REQUIRE(params && obs && act && returns && hp && td_out && loss_out, TS_ERR_INVALID_ARG, "ts_rnnq_update: NULL argument");
    TS_REQUIRE(hp->lr < 0.0 || (adam_m && adam_v && adam_step >= 1), TS_ERR_INVALID_ARG, "ts_rnnq_update: Adam state missing");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, B, T, &n)) return rc;
    hipStream_t s = ts::as_stream(stream);
    if (int rc = ts::ws_reserve(ws, acts_bytes(n) + bwd_bytes(n) + al(4 * B * HEAD) + al(4 * n.count) + 8192)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    Acts a = take_acts(c, n);
    const Bwd bw = take_bwd(c, n);
    float* d_head = c.f(B * HEAD);
    float* grad = c.f(n.count);
    float* norm_part = c.f(1024);
    if (grad_out) grad = grad_out;
    if (cache) {            // the activations ts_rnnq_forward_cache left there
        Carve cc{static_cast<char*>(cache)};
        a = take_acts(cc, n);
    } else if (int rc = forward(s, ws, n, params, obs, nullptr, nullptr, a)) {
        return rc;
    }
    hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(1024), 0, s, a.out, act, returns, weight, B, (float)hp->huber_delta, td_out,
                       d_head, loss_out);
    TS_LAUNCH_CHECK();
    if (int rc = backward(s, ws, n, params, a, d_head, grad, bw)) return rc;
    if (hp->lr < 0.0) return TS_OK;
    return ts::adam_step(s, params, adam_m, adam_v, grad, n.count, adam_step, hp->lr, hp->beta1, hp->beta2, hp->adam_eps,
                         hp->max_grad_norm, norm_part);
}

int ts_rnnq_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                   int64_t hidden, int64_t layers, int64_t n_act, const float* obs, const int64_t* act, const float* returns,
                   const float* weight, int64_t B, int64_t T, const ts_dqn_hparams* hp, float* td_out, float* loss_out,
                   float* grad_out, ts_stream_t stream) {
    return rnnq_update_impl(ws, params, adam_m, adam_v, adam_step, obs_dim, hidden, layers, n_act, obs, act, returns, weight, B, T,
                            hp, td_out, loss_out, grad_out, stream, nullptr);
}

int64_t ts_rnnq_cache_bytes(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t B, int64_t T) {
    RNet n;
    if (make_rnet(obs_dim, hidden, layers, n_act, B, T, &n) != TS_OK) return -1;
    return (int64_t)acts_bytes(n);
}

int ts_rnnq_forward_cache(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                          const float* obs, int64_t B, int64_t T, void* cache, int64_t cache_bytes, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_rnnq_forward_cache: workspace is NULL");
    TS_REQUIRE(params && obs && cache, TS_ERR_INVALID_ARG, "ts_rnnq_forward_cache: NULL argument");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(cache) & 255u) == 0, TS_ERR_INVALID_ARG, "ts_rnnq_forward_cache: cache must be 256-byte aligned");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, B, T, &n)) return rc;
    TS_REQUIRE(cache_bytes >= (int64_t)acts_bytes(n), TS_ERR_SHAPE, "ts_rnnq_forward_cache: cache holds %lld bytes, %lld needed",
               (long long)cache_bytes, (long long)acts_bytes(n));
    Carve cc{static_cast<char*>(cache)};
    const Acts a = take_acts(cc, n);
    return forward(ts::as_stream(stream), ws, n, params, obs, nullptr, nullptr, a);
}

int ts_rnnq_update_cached(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                          int64_t hidden, int64_t layers, int64_t n_act, const float* obs, const int64_t* act, const float* returns,
                          const float* weight, int64_t B, int64_t T, const ts_dqn_hparams* hp, void* cache, float* td_out,
                          float* loss_out, float* grad_out, ts_stream_t stream) {
    TS_REQUIRE(cache != nullptr, TS_ERR_INVALID_ARG, "ts_rnnq_update_cached: cache is NULL");
    return rnnq_update_impl(ws, params, adam_m, adam_v, adam_step, obs_dim, hidden, layers, n_act, obs, act, returns, weight, B, T,
                            hp, td_out, loss_out, grad_out, stream, cache);
}

// ---- LSTM trunk + linear head, generic (RecurrentActorProb / RecurrentCritic, continuous.py:241-380) ------------------------
int ts_lstm_net_layout(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim, int64_t has_fc1, int64_t extra_dim,
                       int64_t* h_out) {
    TS_REQUIRE(h_out != nullptr, TS_ERR_INVALID_ARG, "ts_lstm_net_layout: h_out is NULL");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, out_dim, 1, 1, &n, (int)has_fc1, extra_dim)) return rc;
    h_out[0] = n.k0; h_out[1] = n.count; h_out[2] = n.has_fc1 ? n.off_fc1 : -1;
    for (int l = 0; l < n.L; ++l) { h_out[3 + 2 * l] = n.off_ih[l]; h_out[4 + 2 * l] = n.off_hh[l]; }
    h_out[3 + 2 * n.L] = n.off_head;
    h_out[4 + 2 * n.L] = n.head_in;
    return TS_OK;
}

int ts_lstm_net_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim,
                        int64_t has_fc1, int64_t extra_dim, const float* obs, const float* extra, int64_t B, int64_t T,
                        const float* h_in, const float* c_in, double tanh_scale, float* out, float* h_out, float* c_out,
                        ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_lstm_net_forward: workspace is NULL");
    TS_REQUIRE(params && obs && out && (h_in == nullptr) == (c_in == nullptr), TS_ERR_INVALID_ARG,
               "ts_lstm_net_forward: bad argument (hidden and cell state come together)");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, out_dim, B, T, &n, (int)has_fc1, extra_dim)) return rc;
    hipStream_t s = ts::as_stream(stream);
    if (int rc = ts::ws_reserve(ws, acts_bytes(n) + 4096)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    const Acts a = take_acts(c, n);
    if (int rc = forward(s, ws, n, params, obs, h_in, c_in, a, extra)) return rc;
    const unsigned g = (unsigned)ts::ceil_div(B * n.A, 256);
    if (tanh_scale > 0.0) hipLaunchKernelGGL(head_tanh_kernel, dim3(g), dim3(256), 0, s, a.out, B, n.A, (float)tanh_scale);
    hipLaunchKernelGGL(head_copy_kernel, dim3(g), dim3(256), 0, s, a.out, B, n.A, out);
    TS_LAUNCH_CHECK();
    const size_t blk = (size_t)B * n.H;
    for (int l = 0; l < n.L; ++l) {
        if (h_out) TS_HIP_CHECK(hipMemcpyAsync(h_out + l * blk, a.hbuf[l] + (size_t)n.T * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
        if (c_out) TS_HIP_CHECK(hipMemcpyAsync(c_out + l * blk, a.cbuf[l] + (size_t)n.T * blk, 4 * blk, hipMemcpyDeviceToDevice, s));
    }
    return TS_OK;
}

int ts_lstm_net_backward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim,
                         int64_t has_fc1, int64_t extra_dim, const float* obs, const float* extra, int64_t B, int64_t T,
                         const float* h_in, const float* c_in, double tanh_scale, const float* d_out, float* out, float* grad_out,
                         ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_lstm_net_backward: workspace is NULL");
    TS_REQUIRE(params && obs && d_out && grad_out && (h_in == nullptr) == (c_in == nullptr), TS_ERR_INVALID_ARG,
               "ts_lstm_net_backward: bad argument");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, out_dim, B, T, &n, (int)has_fc1, extra_dim)) return rc;
    hipStream_t s = ts::as_stream(stream);
    if (int rc = ts::ws_reserve(ws, acts_bytes(n) + bwd_bytes(n) + al(4 * B * HEAD) + 8192)) return rc;
    Carve c{static_cast<char*>(ws->base)};
    const Acts a = take_acts(c, n);
    const Bwd bw = take_bwd(c, n);
    float* d_head = c.f(B * HEAD);
    if (int rc = forward(s, ws, n, params, obs, h_in, c_in, a, extra)) return rc;
    const unsigned g = (unsigned)ts::ceil_div(B * n.A, 256);
    if (tanh_scale > 0.0) hipLaunchKernelGGL(head_tanh_kernel, dim3(g), dim3(256), 0, s, a.out, B, n.A, (float)tanh_scale);
    if (out) hipLaunchKernelGGL(head_copy_kernel, dim3(g), dim3(256), 0, s, a.out, B, n.A, out);
    hipLaunchKernelGGL(head_dout_kernel, dim3((unsigned)ts::ceil_div(B * HEAD, 256)), dim3(256), 0, s, d_out, a.out, B, n.A,
                       tanh_scale > 0.0 ? 1 : 0, (float)(tanh_scale > 0.0 ? tanh_scale : 1.0), d_head);
    TS_LAUNCH_CHECK();
    return backward(s, ws, n, params, a, d_head, grad_out, bw);
}

// ---- one call per update on a uniform device-resident replay buffer -------------------------------------------------------
namespace {
struct LearnBatch { int64_t* idx; int64_t* act; float* obs; float* obs_next; float* mask; double* gpow; double* mc; };
struct LearnScratch { LearnBatch b[2]; float* returns; int* err; void* cache; size_t cache_bytes; };

size_t learn_carve(char* base, const RNet& n, int64_t B, int64_t T, int64_t D, LearnScratch* out) {
    Carve c{base};
    auto bytes = [&](size_t nbytes) { char* p = c.p; c.p += al(nbytes); return p; };
    LearnScratch sc{};
    for (int k = 0; k < 2; ++k) {
        sc.b[k].idx = reinterpret_cast<int64_t*>(bytes(8 * (size_t)B));
        sc.b[k].act = reinterpret_cast<int64_t*>(bytes(8 * (size_t)B));
        sc.b[k].obs = reinterpret_cast<float*>(bytes(4 * (size_t)(B * T * D)));
        sc.b[k].obs_next = reinterpret_cast<float*>(bytes(4 * (size_t)(B * T * D)));
        sc.b[k].mask = reinterpret_cast<float*>(bytes(4 * (size_t)B));
        sc.b[k].gpow = reinterpret_cast<double*>(bytes(8 * (size_t)B));
        sc.b[k].mc = reinterpret_cast<double*>(bytes(8 * (size_t)B));
    }
    sc.returns = reinterpret_cast<float*>(bytes(4 * (size_t)B));
    sc.err = reinterpret_cast<int*>(bytes(256));
    sc.cache_bytes = acts_bytes(n);
    sc.cache = bytes(sc.cache_bytes);
    if (out) *out = sc;
    return (size_t)(c.p - base);
}
}  // namespace

int64_t ts_rnnq_learn_scratch_bytes(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t B, int64_t T) {
    RNet n;
    if (make_rnet(obs_dim, hidden, layers, n_act, B, T, &n) != TS_OK) return -1;
    return (int64_t)learn_carve(reinterpret_cast<char*>((uintptr_t)256), n, B, T, obs_dim, nullptr);
}

int ts_rnnq_learn_step(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                       float* adam_v, int64_t adam_step, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                       const ts_rows_replay* rb, int64_t B, int64_t T, int64_t n_step, double gamma, int is_double,
                       const ts_dqn_hparams* hp, uint64_t seed, uint64_t counter, int prepared, void* scratch,
                       int64_t scratch_bytes, float* td_out, float* loss_out, int64_t* idx_out, ts_stream_t stream) {
    TS_REQUIRE(ws != nullptr && ws_aux != nullptr && ws != ws_aux, TS_ERR_WORKSPACE,
               "ts_rnnq_learn_step: two distinct workspaces (the update's and the ahead-of-time forward pass's)");
    TS_REQUIRE(params && adam_m && adam_v && rb && hp && scratch && td_out && loss_out && B >= 1 && adam_step >= 1,
               TS_ERR_INVALID_ARG, "ts_rnnq_learn_step: bad argument");
    TS_REQUIRE(rb->offset && rb->lengths && rb->last_index && rb->done && rb->terminated && rb->rew && rb->obs_rows &&
                   rb->act_col && rb->E >= 1 && rb->slots >= 1, TS_ERR_INVALID_ARG, "ts_rnnq_learn_step: incomplete replay view");
    TS_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255u) == 0, TS_ERR_INVALID_ARG,
               "ts_rnnq_learn_step: scratch must be 256-byte aligned");
    RNet n;
    if (int rc = make_rnet(obs_dim, hidden, layers, n_act, B, T, &n)) return rc;
    LearnScratch sc;
    const size_t need = learn_carve(static_cast<char*>(scratch), n, B, T, obs_dim, &sc);
    TS_REQUIRE(scratch_bytes >= (int64_t)need, TS_ERR_SHAPE, "ts_rnnq_learn_step: scratch holds %lld bytes, %lld needed",
               (long long)scratch_bytes, (long long)need);
    hipStream_t s = ts::as_stream(stream), side, side2;
    if (int rc = ts::side_streams(ws, s, &side, &side2)) return rc;
    // buffer.sample_indices -> batch.obs / obs_next / act -> the network-free half of compute_nstep_return, for update `ctr`
    auto prepare = [&](hipStream_t st, const LearnBatch& b, uint64_t ctr) -> int {
        if (int rc = ts_sample_indices_seeded(rb->offset, rb->E, rb->lengths, seed, ctr, B, b.idx, sc.err, st)) return rc;
        if (int rc = ts_stacked_rows_pair(rb->obs_rows, rb->obs_next_rows, rb->slots, obs_dim, b.idx, B, n_step, T, rb->offset,
                                          rb->E, rb->done, rb->last_index, rb->lengths, rb->act_col, b.obs, b.obs_next, b.act, st))
            return rc;
        return ts_nstep_coefficients(b.idx, B, n_step, rb->offset, rb->E, rb->done, rb->terminated, rb->last_index, rb->lengths,
                                     rb->rew, gamma, b.mask, b.gpow, b.mc, st);
    };
    const LearnBatch& cur = sc.b[counter & 1];
    const LearnBatch& nxt = sc.b[(counter + 1) & 1];
    if (!prepared)
        if (int rc = prepare(s, cur, counter)) return rc;
    // Q_online(batch.obs) of the update on the second side stream, beside the two obs_next passes of _target_q
    if (int rc = ts::stream_wait(ws, s, side2, 6)) return rc;
    // (while ts_profile_begin is active everything above runs on `s` in sequence: the pass then uses -- and is timed by -- `ws`)
    if (int rc = ts_rnnq_forward_cache(ws->profiling ? ws : ws_aux, params, obs_dim, hidden, layers, n_act, cur.obs, B, T, sc.cache,
                                       (int64_t)sc.cache_bytes, side2))
        return rc;
    if (!params_old)        // no lagged pass whose wait on `s` would order the side stream behind the previous update
        if (int rc = ts::stream_wait(ws, s, side, 5)) return rc;
    if (int rc = ts_rnnq_target_returns(ws, params, params_old, obs_dim, hidden, layers, n_act, cur.obs_next, B, T, is_double,
                                        cur.mask, cur.gpow, cur.mc, sc.returns, s))
        return rc;
    // the next update's batch behind the lagged network's pass on the first side stream (ordered behind the previous update by
    // that pass's wait on `s`; the update's closing join orders the next call behind it)
    if (int rc = prepare(side, nxt, counter + 1)) return rc;
    if (sync_target && params_old)       // the periodic hard sync sits between _preprocess_batch and the update (dqn.py:283-285)
        TS_HIP_CHECK(hipMemcpyAsync(params_old, params, 4 * (size_t)n.count, hipMemcpyDeviceToDevice, s));
    if (int rc = ts::stream_wait(ws, side2, s, 7)) return rc;
    if (int rc = rnnq_update_impl(ws, params, adam_m, adam_v, adam_step, obs_dim, hidden, layers, n_act, cur.obs, cur.act, sc.returns,
                                  nullptr, B, T, hp, td_out, loss_out, nullptr, stream, sc.cache))
        return rc;
    // (the backward pass's closing join of the side stream -- stream order behind `prepare` -- makes the next batch visible to
    // the next call on `s`)
    if (idx_out) TS_HIP_CHECK(hipMemcpyAsync(idx_out, cur.idx, 8 * (size_t)B, hipMemcpyDeviceToDevice, s));
    return TS_OK;
}

}  // extern "C"
