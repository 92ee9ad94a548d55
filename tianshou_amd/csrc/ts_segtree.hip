// ts_segtree.hip -- sum-tree and prioritized-replay kernels for gfx950.
//
// Replaces SegmentTree._setitem/_reduce/_get_prefix_sum_idx (tianshou/data/utils/segtree.py:95-134)
// and PrioritizedReplayBuffer.sample_indices/get_weight/update_weight
// (tianshou/data/buffer/prio.py:63-107).  Roofline: latency (log2(bound) dependent 8-byte
// accesses per query); the tree's upper levels stay L2 resident.
//
// float64 tree arithmetic is kept identical to the reference: one add per node, children
// summed left + right, no FMA contraction.
#include "ts_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int ST_THREADS = 1024;

// One workgroup repairs the tree level by level (segtree.py:98-101).  Duplicate leaves are
// resolved like NumPy fancy assignment: the entry with the largest position k wins.
template <typename ValT>
__global__ __launch_bounds__(ST_THREADS) void segtree_setitem_kernel(double* tree,
                                                                     const int64_t* index,
                                                                     const ValT* value, int64_t K,
                                                                     int64_t bound, int32_t* winner) {
    // leaf phase
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) atomicMax(&winner[index[k] - bound], (int32_t)k);
    __syncthreads();
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) {
        const int64_t leaf = index[k];
        if (winner[leaf - bound] == (int32_t)k) tree[leaf] = (double)value[k];
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) winner[index[k] - bound] = -1;  // restore
    // internal levels: every thread recomputes the parents of its entries; entries sharing a
    // parent write the same value.
    for (int64_t shift = 1; (bound >> shift) >= 1; ++shift) {
        __syncthreads();
        for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) {
            const int64_t node = index[k] >> shift;
            // children were written by this workgroup before the barrier (workgroup-scope
            // visibility is what __syncthreads() provides; one workgroup = one CU = one L1)
            tree[node] = tree[2 * node] + tree[2 * node + 1];
        }
    }
}

// Bulk path (K > SEGTREE_BULK_K leaves, e.g. initialising or re-prioritising a large share of a 2^20-leaf tree): the
// single-workgroup kernel above walks K entries x log2(bound) levels on one CU (11.5 ms for 2^20 leaves).  Here the leaf
// phase runs chip-wide and the internal levels are rebuilt one launch per level, bottom up, every node = left child +
// right child: bit-identical to the incremental repair (a node the update does not reach already equals that sum).
template <typename ValT>
__global__ __launch_bounds__(256) void segtree_bulk_claim_kernel(const int64_t* __restrict__ index, int64_t K, int64_t bound,
                                                                 int32_t* __restrict__ winner) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < K) atomicMax(&winner[index[k] - bound], (int32_t)k);
}

template <typename ValT>
__global__ __launch_bounds__(256) void segtree_bulk_leaf_kernel(double* __restrict__ tree, const int64_t* __restrict__ index,
                                                                const ValT* __restrict__ value, int64_t K, int64_t bound,
                                                                const int32_t* __restrict__ winner) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < K) {
        const int64_t leaf = index[k];
        if (winner[leaf - bound] == (int32_t)k) tree[leaf] = (double)value[k];     // NumPy: the last duplicate wins
    }
}

__global__ __launch_bounds__(256) void segtree_bulk_restore_kernel(const int64_t* __restrict__ index, int64_t K, int64_t bound,
                                                                   int32_t* __restrict__ winner) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < K) winner[index[k] - bound] = -1;
}

// nodes [first, 2 first) of one level
__global__ __launch_bounds__(256) void segtree_level_kernel(double* __restrict__ tree, int64_t first) {
    const int64_t node = first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (node < 2 * first) tree[node] = tree[2 * node] + tree[2 * node + 1];
}

constexpr int64_t SEGTREE_BULK_K = 8192;

__global__ void segtree_reduce_kernel(const double* tree, int64_t start, int64_t end, double* out) {
    // segtree.py:108-116, single lane
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double result = 0.0;
    while (end - start > 1) {
        if ((start & 1) == 0) result += tree[start + 1];
        start >>= 1;
        if ((end & 1) == 1) result += tree[end - 1];
        end >>= 1;
    }
    *out = result;
}

__device__ __forceinline__ int64_t prefix_descend(double& v, int64_t bound, const double* sums) {
    int64_t index = 1;
    while (index < bound) {                  // segtree.py:126-132
        index *= 2;
        const double lsons = sums[index];
        const bool direct = lsons < v;
        v = v - lsons * (direct ? 1.0 : 0.0);
        index += direct ? 1 : 0;
    }
    return index - bound;
}

__global__ void segtree_prefix_kernel(double* value, int64_t K, int64_t bound, const double* sums,
                                      int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride) {
        double v = value[k];
        out[k] = prefix_descend(v, bound, sums);
        value[k] = v;
    }
}

// prio.py:63-79 + :104-106 in one workgroup: u * total -> descent -> IS weight -> / max
__global__ __launch_bounds__(ST_THREADS) void per_sample_kernel(const double* tree, int64_t bound,
                                                                const double* u, int64_t K,
                                                                const double* prio_minmax, double beta,
                                                                int weight_norm, int64_t* idx_out,
                                                                double* weight_out) {
    __shared__ double wmax_s[ST_THREADS / 64];
    const double total = tree[1];            // SegmentTree.reduce() :55-56
    const double min_prio = prio_minmax[1];
    double local_max = -INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) {
        double v = u[k] * total;             // prio.py:65
        const int64_t idx = prefix_descend(v, bound, tree);
        idx_out[k] = idx;
        const double w = pow(tree[idx + bound] / min_prio, -beta);  // prio.py:79
        weight_out[k] = w;
        local_max = fmax(local_max, w);
    }
    if (!weight_norm) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmax(local_max, __shfl_down(local_max, off, 64));
    if ((threadIdx.x & 63) == 0) wmax_s[threadIdx.x >> 6] = local_max;
    __syncthreads();
    double m = wmax_s[0];
    for (int w = 1; w < ST_THREADS / 64; ++w) m = fmax(m, wmax_s[w]);
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) weight_out[k] = weight_out[k] / m;  // :106
}

// prio.py:87-90: weight = |td| + eps (float32), leaf value weight ** alpha (float32 pow),
// running max/min of the un-exponentiated weight.
__global__ __launch_bounds__(ST_THREADS) void per_prepare_update_kernel(
    const int64_t* index, const float* new_weight, int64_t K, int64_t bound, float alpha, float eps,
    int64_t* leaf_out, float* val_out, double* prio_minmax) {
    __shared__ float mx_s[ST_THREADS / 64], mn_s[ST_THREADS / 64];
    float mx = -INFINITY, mn = INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += ST_THREADS) {
        const float w = fabsf(new_weight[k]) + eps;
        leaf_out[k] = index[k] + bound;
        val_out[k] = powf(w, alpha);
        mx = fmaxf(mx, w);
        mn = fminf(mn, w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
        mn = fminf(mn, __shfl_down(mn, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        mx_s[threadIdx.x >> 6] = mx;
        mn_s[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ST_THREADS / 64; ++w) {
            mx = fmaxf(mx, mx_s[w]);
            mn = fminf(mn, mn_s[w]);
        }
        if (K > 0) {
            prio_minmax[0] = fmax(prio_minmax[0], (double)mx);
            prio_minmax[1] = fmin(prio_minmax[1], (double)mn);
        }
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t g = ts::ceil_div(n, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int ts_segtree_setitem(ts_workspace* ws, double* tree, int64_t bound, const int64_t* index,
                       const void* value, int value_dtype, int64_t K, ts_stream_t stream) {
    TS_REQUIRE(K >= 0 && bound >= 1, TS_ERR_INVALID_ARG, "ts_segtree_setitem: bad size");
    TS_REQUIRE((bound & (bound - 1)) == 0, TS_ERR_INVALID_ARG,
               "ts_segtree_setitem: bound must be a power of two");
    TS_REQUIRE(K < ((int64_t)1 << 31), TS_ERR_UNSUPPORTED, "ts_segtree_setitem: K too large");
    TS_REQUIRE(value_dtype == 0 || value_dtype == 1, TS_ERR_INVALID_ARG,
               "ts_segtree_setitem: value_dtype must be 0 (f32) or 1 (f64)");
    if (K == 0) return TS_OK;
    TS_REQUIRE(tree && index && value, TS_ERR_INVALID_ARG, "ts_segtree_setitem: NULL array argument");
    hipStream_t s = ts::as_stream(stream);
    int32_t* winner = nullptr;
    int rc = ts::ws_winner(ws, bound, s, &winner);
    if (rc != TS_OK) return rc;
    if (K > SEGTREE_BULK_K) {
        const unsigned gk = (unsigned)ts::ceil_div(K, (int64_t)256);          // one entry per thread: no cap
        hipLaunchKernelGGL(segtree_bulk_claim_kernel<float>, dim3(gk), dim3(256), 0, s, index, K, bound, winner);
        if (value_dtype == 1)
            hipLaunchKernelGGL(segtree_bulk_leaf_kernel<double>, dim3(gk), dim3(256), 0, s, tree, index, (const double*)value,
                               K, bound, winner);
        else
            hipLaunchKernelGGL(segtree_bulk_leaf_kernel<float>, dim3(gk), dim3(256), 0, s, tree, index, (const float*)value, K,
                               bound, winner);
        hipLaunchKernelGGL(segtree_bulk_restore_kernel, dim3(gk), dim3(256), 0, s, index, K, bound, winner);
        for (int64_t first = bound >> 1; first >= 1; first >>= 1)
            hipLaunchKernelGGL(segtree_level_kernel, dim3((unsigned)ts::ceil_div(first, (int64_t)256)), dim3(256), 0, s, tree,
                               first);
        TS_LAUNCH_CHECK();
        return TS_OK;
    }
    if (value_dtype == 1)
        hipLaunchKernelGGL(segtree_setitem_kernel<double>, dim3(1), dim3(ST_THREADS), 0, s, tree,
                           index, (const double*)value, K, bound, winner);
    else
        hipLaunchKernelGGL(segtree_setitem_kernel<float>, dim3(1), dim3(ST_THREADS), 0, s, tree,
                           index, (const float*)value, K, bound, winner);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_segtree_reduce(const double* tree, int64_t start, int64_t end, double* out,
                      ts_stream_t stream) {
    TS_REQUIRE(tree && out, TS_ERR_INVALID_ARG, "ts_segtree_reduce: NULL argument");
    TS_REQUIRE(start >= 0 && end >= 0, TS_ERR_INVALID_ARG, "ts_segtree_reduce: negative node");
    hipLaunchKernelGGL(segtree_reduce_kernel, dim3(1), dim3(64), 0, ts::as_stream(stream), tree,
                       start, end, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_segtree_prefix_sum_idx(double* value, int64_t K, int64_t bound, const double* sums,
                              int64_t* out, ts_stream_t stream) {
    TS_REQUIRE(K >= 0 && bound >= 1, TS_ERR_INVALID_ARG, "ts_segtree_prefix_sum_idx: bad size");
    if (K == 0) return TS_OK;
    TS_REQUIRE(value && sums && out, TS_ERR_INVALID_ARG,
               "ts_segtree_prefix_sum_idx: NULL array argument");
    hipLaunchKernelGGL(segtree_prefix_kernel, dim3(grid_for(K, 128)), dim3(128), 0,
                       ts::as_stream(stream), value, K, bound, sums, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_per_sample(ts_workspace* ws, const double* tree, int64_t bound, const double* u, int64_t K,
                  const double* prio_minmax, double beta, int weight_norm, int64_t* idx_out,
                  double* weight_out, ts_stream_t stream) {
    (void)ws;
    TS_REQUIRE(K >= 0 && bound >= 1, TS_ERR_INVALID_ARG, "ts_per_sample: bad size");
    if (K == 0) return TS_OK;
    TS_REQUIRE(tree && u && prio_minmax && idx_out && weight_out, TS_ERR_INVALID_ARG,
               "ts_per_sample: NULL array argument");
    hipLaunchKernelGGL(per_sample_kernel, dim3(1), dim3(ST_THREADS), 0, ts::as_stream(stream), tree,
                       bound, u, K, prio_minmax, beta, weight_norm, idx_out, weight_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_per_update_weight(ts_workspace* ws, double* tree, int64_t bound, const int64_t* index,
                         const float* new_weight, int64_t K, double alpha, double* prio_minmax,
                         ts_stream_t stream) {
    TS_REQUIRE(K >= 0 && bound >= 1, TS_ERR_INVALID_ARG, "ts_per_update_weight: bad size");
    if (K == 0) return TS_OK;
    TS_REQUIRE(tree && index && new_weight && prio_minmax, TS_ERR_INVALID_ARG,
               "ts_per_update_weight: NULL array argument");
    hipStream_t s = ts::as_stream(stream);
    // scratch: leaf ids (int64[K]) + float32 values[K]
    int rc = ts::ws_reserve(ws, (size_t)K * (sizeof(int64_t) + sizeof(float)) + 64);
    if (rc != TS_OK) return rc;
    int64_t* leaf = reinterpret_cast<int64_t*>(ws->base);
    float* val = reinterpret_cast<float*>(leaf + K);
    const float eps = 1.1920928955078125e-07f;  // np.finfo(np.float32).eps, prio.py:40
    hipLaunchKernelGGL(per_prepare_update_kernel, dim3(1), dim3(ST_THREADS), 0, s, index,
                       new_weight, K, bound, (float)alpha, eps, leaf, val, prio_minmax);
    TS_LAUNCH_CHECK();
    return ts_segtree_setitem(ws, tree, bound, leaf, val, 0, K, stream);
}

}  // extern "C"
