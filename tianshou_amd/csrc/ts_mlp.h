// ts_mlp.h -- three Linear layers of a [K1 -> 256 -> 256 -> N3] ReLU MLP in one launch (internal to libtsengine).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct ts_workspace;

namespace ts {

// Shapes mlp3_forward covers: hidden 256, head 32 or 64 columns, K1 a multiple of 32 up to 1024.
bool mlp3_supported(int K1, int hidden, int head_cols);

// h1 = relu(x W1 + b1), h2 = relu(h1 W2 + b2), out = h2 W3 + b3.  x [M, K1] (zero-padded columns), Wb_i in the
// layer-matrix layout of ts_conv.h ([K_i + 1, N_i], last row = bias).  h1 / h2 ([M, 256]) are written because the
// backward pass reads them; nullptr skips the store (inference-only chains).
int mlp3_forward(hipStream_t s, const float* x, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                 int head_cols, float* h1, float* h2, float* out, ts_workspace* prof = nullptr);

// The same for up to MLP3_MAX_NETS networks of equal shape on the same input in ONE launch (twin critics, ensemble members:
// blockIdx.y = network).
constexpr int MLP3_MAX_NETS = 8;
int mlp3_forward_n(hipStream_t s, int nets, const float* x, int M, int K1, const float* const* wb1, const float* const* wb2,
                   const float* const* wb3, int head_cols, float* const* h1, float* const* h2, float* const* out,
                   ts_workspace* prof = nullptr);
// ... each network on its OWN input rows xs[k] ([M, K1] each; ts_sac_learn_rows: the lagged critics on (s', a') beside the live
// critics on (s, a)).  Per-row results do not depend on which networks share a launch.
int mlp3_forward_nx(hipStream_t s, int nets, const float* const* xs, int M, int K1, const float* const* wb1, const float* const* wb2,
                    const float* const* wb3, int head_cols, float* const* h1, float* const* h2, float* const* out,
                    ts_workspace* prof = nullptr);

// The input gradients of the same chain in one launch: dh2 = (d_out W3^T) * (h2 > 0), dh1 = (dh2 W2^T) * (h1 > 0)
// ([M, 256] each, consumed by the weight-gradient GEMMs), and dx[:, col0:col1) = dh1 W1^T for up to 128 input columns
// (dx nullable; row pitch K1; the 16-column tiles covering the range are written, as ts::conv_dgrad does).
bool mlp3_backward_supported(int K1, int hidden, int head_cols, bool want_dx, int col0, int col1);
int mlp3_backward(hipStream_t s, const float* d_out, int M, int K1, const float* wb1, const float* wb2, const float* wb3,
                  int head_cols, const float* h1, const float* h2, float* dh1, float* dh2, float* dx, int col0, int col1,
                  ts_workspace* prof = nullptr);

// up to MLP3_MAX_NETS networks of equal shape (each with its own upstream gradient, activations and outputs) in one launch;
// input gradients (dx) for all of them or for none
int mlp3_backward_n(hipStream_t s, int nets, const float* const* d_out, int M, int K1, const float* const* wb1,
                    const float* const* wb2, const float* const* wb3, int head_cols, const float* const* h1,
                    const float* const* h2, float* const* dh1, float* const* dh2, float* const* dx, int col0, int col1,
                    ts_workspace* prof = nullptr);

}  // namespace ts
