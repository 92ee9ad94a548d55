// ts_conv.h -- fp32-MFMA implicit-GEMM convolution / linear layers (internal to libtsengine).
//
// Activations are NHWC ("pixel-major") float32: X[B][IH][IW][IC].  A layer's parameters are one
// row-major matrix Wb[(KH*KW*IC) + 1][OC]: row k = (kh, kw, ic) holds the weights of all output
// channels for that tap, the last row is the bias.  A Linear layer is the 1x1 case (IH = IW = 1).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct ts_workspace;

namespace ts {

struct ConvGeom {
    int B, IH, IW, IC, KH, KW, S, OH, OW, OC;
    int K() const { return KH * KW * IC; }
    int64_t in_elems() const { return (int64_t)B * IH * IW * IC; }
    int64_t out_elems() const { return (int64_t)B * OH * OW * OC; }
    int64_t param_elems() const { return (int64_t)(K() + 1) * OC; }
};

// Split plans (deterministic: partial results go to slabs that a second kernel sums in order).
int conv_fwd_splits(const ConvGeom& g);      // >1 only for layers whose output grid cannot fill the chip
int conv_wgrad_splits(const ConvGeom& g);

// Y = act(X (*) W + b).  `split_buf` (conv_fwd_splits(g) * out_elems floats) is needed when splits > 1.
int conv_forward(hipStream_t s, const ConvGeom& g, const float* X, const float* Wb, float* Y, bool relu,
                 float* split_buf, ts_workspace* prof = nullptr, bool x_u8 = false);
// x_u8: X points to uint8 NHWC data (raw frames) with the same geometry; values are converted on load.
// slabs[s][(K+1)*OC] = partial d(loss)/d(Wb) over the s-th share of the output pixels.
int conv_wgrad(hipStream_t s, const ConvGeom& g, const float* X, const float* dY, float* slabs,
               ts_workspace* prof = nullptr, bool x_u8 = false);
// dX = (dY (*)^T W) * (mask > 0); mask = the layer input as produced by the previous ReLU (or null).
int conv_dgrad(hipStream_t s, const ConvGeom& g, const float* dY, const float* Wb, const float* mask,
               float* dX, ts_workspace* prof = nullptr, int col_begin = 0, int col_end = -1);
// (col_begin, col_end): only the input-channel tiles covering [col_begin, col_end) are computed.
// out[i] = sum_s slabs[s][i]
int slab_sum(hipStream_t s, const float* slabs, int nslab, int64_t n, float* out);
// Backward pass of a chain of n layers (layer i reads x[i], x[i + 1] = its output after ReLU; dy[n - 1] = the upstream
// gradient, already masked): grad[i] = dL/d(weights, bias) of layer i, dy[i - 1] = (dy[i] W_i^T) * (x[i] > 0).
// The input gradients run down the caller's stream.  The weight gradient of layer i (+ its slab sum) needs dY_i only:
// layers n-1, n-2, ... alternate between the workspace's two side streams, so none waits for the one above it (with one
// side stream the four weight gradients of the NatureCNN formed the critical path of the backward pass); the first
// layer's -- nothing below depends on it -- stays on the caller's stream behind the last input gradient.  slabs[i]:
// conv_wgrad_splits(l[i]) * param_elems floats, one buffer PER LAYER.  On return the caller's stream has joined both side
// streams.
int chain_backward(hipStream_t s, ts_workspace* ws, int n, const ConvGeom* l, const float* const* x, float* const* dy,
                   const float* const* wb, float* const* slabs, float* const* grad, bool x0_u8);
// The weight gradients of up to sixteen small layers in one launch (bit-identical to conv_wgrad per layer; layers that
// qualify for the large-M kernels are launched on their own).  slabs[i]: conv_wgrad_splits(g[i]) * param_elems floats.
int conv_wgrad_group(hipStream_t s, int n, const ConvGeom* g, const float* const* X, const float* const* dY,
                     float* const* slabs, ts_workspace* prof = nullptr);
// the same for up to sixteen independent slab sets in one launch (same summation order per element as slab_sum)
struct SlabSeg { const float* slabs; int nslab; int64_t n; float* out; };
int slab_sum_multi(hipStream_t s, const SlabSeg* segs, int nseg);


// ---- second generation (ts_conv2.hip): large-M kernels, chosen inside conv_forward / conv_wgrad / conv_dgrad --------
// (weight block resident in LDS, activation operand straight from global memory into MFMA registers; same k-sequential
// summation as the first generation, so forward / dgrad results are bit-identical).  TS_CONV_V2=0 / 1 forces the choice.
bool conv2_use_forward(const ConvGeom& g, bool x_u8);
bool conv2_use_wgrad(const ConvGeom& g, bool x_u8);
bool conv2_use_dgrad(const ConvGeom& g, bool have_ws, int col_begin, int col_end);
int conv2_wgrad_splits(const ConvGeom& g);
int conv2_forward(hipStream_t s, const ConvGeom& g, const float* X, const float* Wb, float* Y, bool relu,
                  ts_workspace* prof, bool x_u8);
int conv2_wgrad(hipStream_t s, const ConvGeom& g, const float* X, const float* dY, float* slabs, ts_workspace* prof,
                bool x_u8);
int conv2_dgrad(hipStream_t s, const ConvGeom& g, const float* dY, const float* Wb, const float* mask, float* dX,
                ts_workspace* ws);

}  // namespace ts
