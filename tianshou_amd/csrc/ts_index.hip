// ts_index.hip -- replay-buffer index math and row gathers for gfx950 (integer work, bit-exact).
//
// Replaces _next_index/_prev_index (tianshou/data/buffer/manager.py:311-363),
// ReplayBufferManager.unfinished_index (:85-91), sample_indices(0) (:216-234) and the
// fancy-index gathers of ReplayBuffer.__getitem__ (tianshou/data/buffer/buffer_base.py:605-649).
// Roofline: HBM / latency (8 B index read + 8 B write + one 1-byte `done` gather per query).
#include <algorithm>

#include "ts_common.h"

namespace {

__device__ __forceinline__ int64_t pymod(int64_t a, int64_t m) {
    const int64_t r = a % m;
    return r < 0 ? r + m : r;
}

// largest e with offset[e] <= idx; offset ascending with offset[0] == 0 and idx < offset[E]
__device__ __forceinline__ int64_t find_sub(const int64_t* offset, int64_t E, int64_t idx) {
    int64_t lo = 0, hi = E;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offset[mid] <= idx) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool NEXT>
__global__ void step_index_kernel(const int64_t* index, int64_t I, const int64_t* offset,
                                  int64_t E, const uint8_t* done, const int64_t* last_index,
                                  const int64_t* lengths, int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = offset[E];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < I; i += stride) {
        const int64_t idx = pymod(index[i], total);         // manager.py:319 / :347
        const int64_t e = find_sub(offset, E, idx);
        const int64_t start = offset[e];
        const int64_t len = lengths[e];
        const int64_t cur_len = len > 1 ? len : 1;          // max(1, cur_len) :330 / :358
        if (NEXT) {
            const int64_t end_flag = (done[idx] != 0) | (idx == last_index[e]);     // :361
            out[i] = pymod(idx - start + 1 - end_flag, cur_len) + start;            // :362
        } else {
            const int64_t subind = pymod(idx - start - 1, cur_len);                 // :333
            const int64_t end_flag = (done[subind + start] != 0) | (subind + start == last_index[e]);
            out[i] = pymod(subind + end_flag, cur_len) + start;                     // :335
        }
    }
}

// ReplayBuffer.get with stack_num > 1 (buffer_base.py:586-596): column stack-1-j holds prev^j(index)
__global__ void stack_indices_kernel(const int64_t* index, int64_t I, int64_t stack, const int64_t* offset,
                                     int64_t E, const uint8_t* done, const int64_t* last_index,
                                     const int64_t* lengths, int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = offset[E];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < I; i += stride) {
        int64_t idx = index[i];
        out[i * stack + stack - 1] = idx;                   // val[indices] is taken before prev()
        for (int64_t j = 1; j < stack; ++j) {
            idx = pymod(idx, total);
            const int64_t e = find_sub(offset, E, idx);
            const int64_t start = offset[e];
            const int64_t len = lengths[e];
            const int64_t cur_len = len > 1 ? len : 1;
            const int64_t subind = pymod(idx - start - 1, cur_len);
            const int64_t end_flag = (done[subind + start] != 0) | (subind + start == last_index[e]);
            idx = pymod(subind + end_flag, cur_len) + start;
            out[i * stack + stack - 1 - j] = idx;
        }
    }
}

// frames u8 [n_planes][plane_elems] -> float32 NHWC [B][plane_elems][C]: out[b, p, c] = src[plane[b, c], p]
template <int C>
__global__ __launch_bounds__(256) void gather_planes_kernel(const uint8_t* __restrict__ src, int64_t plane_elems,
                                                            const int64_t* __restrict__ plane, int64_t B, int Cdyn,
                                                            float* __restrict__ out) {
    const int Cn = C > 0 ? C : Cdyn;
    const int64_t b = blockIdx.y;
    const int64_t* pl = plane + b * Cn;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < plane_elems; p += (int64_t)gridDim.x * 256) {
        if (C == 4) {
            float4 v;
            v.x = (float)src[pl[0] * plane_elems + p];
            v.y = (float)src[pl[1] * plane_elems + p];
            v.z = (float)src[pl[2] * plane_elems + p];
            v.w = (float)src[pl[3] * plane_elems + p];
            *reinterpret_cast<float4*>(out + (b * plane_elems + p) * 4) = v;
        } else {
            for (int c = 0; c < Cn; ++c) out[(b * plane_elems + p) * Cn + c] = (float)src[pl[c] * plane_elems + p];
        }
    }
}

// ReplayBufferManager.add bookkeeping (manager.py:131-198 + buffer_base.py:360-418), one thread per entry.
// Entries must address distinct sub-buffers (as every collector step does); then the per-entry updates are
// independent and the float64 episode-return accumulation is the reference's, bit for bit.
struct AddArgs {
    const int64_t* ids; int64_t K;
    const double* rew; const uint8_t* term; const uint8_t* trunc;
    const int64_t* offset;
    int64_t* insertion; int64_t* lengths; int64_t* last_index; double* ep_return; int64_t* ep_len; int64_t* ep_start;
    double* rew_B; uint8_t* term_B; uint8_t* trunc_B; uint8_t* done_B;
    int64_t* index_out; double* ep_return_out; int64_t* ep_len_out; int64_t* ep_start_out;
};

__global__ __launch_bounds__(256) void buffer_add_state_kernel(AddArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.K) return;
    const int64_t e = a.ids ? a.ids[k] : k;
    const int64_t start = a.offset[e], maxsize = a.offset[e + 1] - start;
    const bool done = a.term[k] || a.trunc[k];                        // manager.py:150
    const int64_t cur = a.insertion[e];                                // buffer_base.py:381
    const int64_t len = a.lengths[e];
    a.lengths[e] = len + 1 < maxsize ? len + 1 : maxsize;              // :382
    const int64_t nxt = (cur + 1) % maxsize;                           // :383
    a.insertion[e] = nxt;
    const double er = a.ep_return[e] + a.rew[k];                       // :385
    const int64_t el = a.ep_len[e] + 1;                                // :386
    const int64_t idx = cur + start;                                   // manager.py:170
    a.index_out[k] = idx;
    a.ep_return_out[k] = done ? er : 0.0;                              // buffer_base.py:397-410
    a.ep_len_out[k] = done ? el : 0;
    a.ep_start_out[k] = a.ep_start[e] + start;                         // manager.py:171
    a.ep_return[e] = done ? 0.0 : er;                                  // :414-418
    a.ep_len[e] = done ? 0 : el;
    if (done) a.ep_start[e] = nxt;
    a.last_index[e] = idx;                                             // manager.py:176
    a.rew_B[idx] = a.rew[k];
    a.term_B[idx] = a.term[k] != 0;
    a.trunc_B[idx] = a.trunc[k] != 0;
    a.done_B[idx] = done;
}

struct ScatterKeys { void* dst[8]; const void* src[8]; int64_t row_bytes[8]; int n; };

// dst_key[index[k]] = src_key[k] for up to 8 keys in one launch (blockIdx.y = key); 16-byte path when aligned
__global__ __launch_bounds__(256) void scatter_rows_kernel(ScatterKeys keys, const int64_t* __restrict__ index, int64_t K) {
    const int key = blockIdx.y;
    const int64_t rb = keys.row_bytes[key];
    char* dst = static_cast<char*>(keys.dst[key]);
    const char* src = static_cast<const char*>(keys.src[key]);
    const bool vec = (rb % 16 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0);
    const int64_t unit = vec ? 16 : 1, per_row = rb / unit, total = K * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / per_row, j = i - k * per_row;
        const int64_t row = index[k];
        if (vec) reinterpret_cast<uint4*>(dst + row * rb)[j] = reinterpret_cast<const uint4*>(src + k * rb)[j];
        else dst[row * rb + j] = src[k * rb + j];
    }
}

// same gather, uint8 out (NHWC uint8 [B][plane_elems][C]): the conv kernels convert on load, so the float32 copy of
// the observations (4x the bytes) never exists
template <int C>
__global__ __launch_bounds__(256) void gather_planes_u8_kernel(const uint8_t* __restrict__ src, int64_t plane_elems,
                                                               const int64_t* __restrict__ plane, int64_t B, int Cdyn,
                                                               uint8_t* __restrict__ out) {
    const int Cn = C > 0 ? C : Cdyn;
    const int64_t b = blockIdx.y;
    const int64_t* pl = plane + b * Cn;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < plane_elems; p += (int64_t)gridDim.x * 256) {
        if (C == 4) {
            const uint32_t v = (uint32_t)src[pl[0] * plane_elems + p] | ((uint32_t)src[pl[1] * plane_elems + p] << 8) |
                               ((uint32_t)src[pl[2] * plane_elems + p] << 16) | ((uint32_t)src[pl[3] * plane_elems + p] << 24);
            *reinterpret_cast<uint32_t*>(out + (b * plane_elems + p) * 4) = v;
        } else {
            for (int c = 0; c < Cn; ++c) out[(b * plane_elems + p) * Cn + c] = src[pl[c] * plane_elems + p];
        }
    }
}

// The frame-stack case the Atari workloads run (C = 4 planes of plane_elems % 16 == 0 bytes, e.g. 84 x 84): every thread moves
// 16 pixels of the four planes -- four 16-byte loads, a 4 x 4 byte transpose per dword group (v_perm_b32), four 16-byte stores --
// instead of four single-byte loads and one 4-byte store per pixel: 16 x fewer memory instructions for the same bytes
// (round 3 measured the byte-wise kernel at ~2 TB/s of read + write traffic, 8 % of the Atari-shape PPO update).
__global__ __launch_bounds__(256) void gather_planes_u8x16_kernel(const uint8_t* __restrict__ src, int64_t plane_elems,
                                                                  const int64_t* __restrict__ plane, int64_t B,
                                                                  uint8_t* __restrict__ out) {
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    const int gpp = (int)(plane_elems >> 4);                       // 16-pixel groups per plane
    const int64_t total = B * gpp;
    for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < total; gidx += (int64_t)gridDim.x * 256) {
        const int64_t b = gidx / gpp;
        const int q = (int)(gidx - b * gpp);
        const int64_t* pl = plane + b * 4;
        u32x4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const u32x4*>(src + pl[c] * plane_elems + 16 * q);
        u32x4* o = reinterpret_cast<u32x4*>(out + (b * plane_elems + 16 * q) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // dwords a, b, c, d hold pixels 4j .. 4j+3 of planes 0 .. 3; output dword p = {a.p, b.p, c.p, d.p}
            const uint32_t t0 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x05010400u), t1 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x05010400u), u1 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x07030602u);
            o[j] = u32x4{__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                         __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u)};
        }
    }
}

// Both frame-stack gathers of a DQN-family update on a `save_only_last_obs` buffer in ONE launch: the batch's own stacked
// observations (ReplayBuffer.get(index, "obs") with stack_num = 4, buffer_base.py:586-596) and the stacked observations at
// next(indices_after_n) that _target_q reads as obs_next (algorithm_base.py:772-791, buffer_base.py:624-626).  One workgroup
// per sample: lane 0 of wave 0 walks prev() three times from the index, lane 0 of wave 1 walks next() n_step times and then
// prev() three times (the same arithmetic as step_index_kernel / stack_indices_kernel / ts_returns.hip next_one -- the two
// chains run side by side), then all 256 threads copy the eight planes with the 16-pixel transposing copy of
// gather_planes_u8x16_kernel.  Replaces six launches (n-step indices, next(), two index stacks, two gathers).
__global__ __launch_bounds__(256) void dqn_gather_pair_kernel(const uint8_t* __restrict__ src, int64_t plane_elems,
                                                              const int64_t* __restrict__ index, int n_step,
                                                              const int64_t* __restrict__ offset, int64_t E,
                                                              const uint8_t* __restrict__ done,
                                                              const int64_t* __restrict__ last_index,
                                                              const int64_t* __restrict__ lengths,
                                                              uint8_t* __restrict__ out_s, uint8_t* __restrict__ out_n) {
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    __shared__ int64_t planes[2][4];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    if ((tid & 63) == 0 && tid < 128) {
        const int which = tid >> 6;
        const int64_t total = offset[E];
        int64_t idx = index[b];
        if (which == 1) {
            for (int n = 0; n < n_step; ++n) {              // n_step - 1 steps to indices_after_n, one more to obs_next's slot
                idx = pymod(idx, total);
                const int64_t e = find_sub(offset, E, idx);
                const int64_t start = offset[e], len = lengths[e], cur_len = len > 1 ? len : 1;
                const int64_t end_flag = (done[idx] != 0) | (idx == last_index[e]);
                idx = pymod(idx - start + 1 - end_flag, cur_len) + start;
            }
        }
        planes[which][3] = idx;                              // val[indices] is taken before prev()
        for (int j = 1; j < 4; ++j) {
            idx = pymod(idx, total);
            const int64_t e = find_sub(offset, E, idx);
            const int64_t start = offset[e], len = lengths[e], cur_len = len > 1 ? len : 1;
            const int64_t subind = pymod(idx - start - 1, cur_len);
            const int64_t end_flag = (done[subind + start] != 0) | (subind + start == last_index[e]);
            idx = pymod(subind + end_flag, cur_len) + start;
            planes[which][3 - j] = idx;
        }
    }
    __syncthreads();
    const int gpp = (int)(plane_elems >> 4);
    for (int gi = tid; gi < 2 * gpp; gi += 256) {
        const int which = gi >= gpp;
        const int q = gi - which * gpp;
        u32x4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const u32x4*>(src + planes[which][c] * plane_elems + 16 * q);
        u32x4* o = reinterpret_cast<u32x4*>((which ? out_n : out_s) + (b * plane_elems + 16 * q) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t0 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x05010400u), t1 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x05010400u), u1 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x07030602u);
            o[j] = u32x4{__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                         __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u)};
        }
    }
}

// The vector-observation counterpart of dqn_gather_pair_kernel (DRQN, test/discrete/test_drqn.py:79-101: float rows, stack_num
// steps per sample): ReplayBuffer.get(index, "obs") and the stacked observations `_target_q` reads n steps on -- from
// `rows_next` at indices_after_n when the buffer stores obs_next, else from `rows` at next(indices_after_n)
// (buffer_base.py:586-596, 624-626; algorithm_base.py:772-791) -- and, optionally, batch.act = act_col[index].  One wave per
// sample: lanes 0 and 32 walk the two index chains side by side, then the 64 lanes copy the 2 x T rows of D floats.
// Replaces seven launches (n-step indices, next(), two index stacks, two row gathers, the action gather).
constexpr int ROWS_T_MAX = 16;
__global__ __launch_bounds__(256) void stacked_rows_pair_kernel(const float* __restrict__ rows, const float* __restrict__ rows_next,
                                                                int D, const int64_t* __restrict__ index, int64_t B, int n_step,
                                                                int T, const int64_t* __restrict__ offset, int64_t E,
                                                                const uint8_t* __restrict__ done,
                                                                const int64_t* __restrict__ last_index,
                                                                const int64_t* __restrict__ lengths,
                                                                const int64_t* __restrict__ act_col, float* __restrict__ out_s,
                                                                float* __restrict__ out_n, int64_t* __restrict__ act_out) {
    __shared__ int64_t slot[4][2][ROWS_T_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b < B && (lane & 31) == 0) {
        const int which = lane >> 5;
        const int64_t total = offset[E];
        int64_t idx = index[b];
        if (which == 1) {
            const int walk = rows_next ? n_step - 1 : n_step;     // to indices_after_n, and one more to obs_next's slot
            for (int n = 0; n < walk; ++n) {
                idx = pymod(idx, total);
                const int64_t e = find_sub(offset, E, idx);
                const int64_t start = offset[e], len = lengths[e], cur_len = len > 1 ? len : 1;
                const int64_t end_flag = (done[idx] != 0) | (idx == last_index[e]);
                idx = pymod(idx - start + 1 - end_flag, cur_len) + start;
            }
        }
        slot[wave][which][T - 1] = idx;                      // val[indices] is taken before prev()
        for (int j = 1; j < T; ++j) {
            idx = pymod(idx, total);
            const int64_t e = find_sub(offset, E, idx);
            const int64_t start = offset[e], len = lengths[e], cur_len = len > 1 ? len : 1;
            const int64_t subind = pymod(idx - start - 1, cur_len);
            const int64_t end_flag = (done[subind + start] != 0) | (subind + start == last_index[e]);
            idx = pymod(subind + end_flag, cur_len) + start;
            slot[wave][which][T - 1 - j] = idx;
        }
    }
    if (b < B && lane == 1 && act_col) act_out[b] = act_col[pymod(index[b], offset[E])];    // act[index] with Python's negative indices
    __syncthreads();
    if (b >= B) return;
    const int per = T * D;
    for (int e = lane; e < 2 * per; e += 64) {
        const int which = e >= per;
        const int r = e - which * per;
        const int t = r / D, d = r - t * D;
        const float* src = (which && rows_next) ? rows_next : rows;
        (which ? out_n : out_s)[b * per + r] = src[slot[wave][which][t] * D + d];
    }
}

// single workgroup, order-preserving compaction over the E sub-buffers
__global__ void unfinished_kernel(int64_t E, const uint8_t* done, const int64_t* last_index,
                                  const int64_t* lengths, int64_t* out, int64_t* n_out) {
    __shared__ int wave_cnt[16];
    __shared__ int64_t base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t e0 = 0; e0 < E; e0 += blockDim.x) {
        const int64_t e = e0 + threadIdx.x;
        bool keep = false;
        int64_t last = 0;
        if (e < E) {
            last = last_index[e];
            keep = lengths[e] > 0 && done[last] == 0;
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ULL << lane) - 1ULL));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < n_waves; ++w) {
            if (w < wave) woff += wave_cnt[w];
            tot += wave_cnt[w];
        }
        const int64_t base = base_s;
        if (keep) out[base + woff + before] = last;
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base_s;
}

// single workgroup exclusive prefix sum of lengths -> prefix[E+1]
__global__ void lengths_prefix_kernel(const int64_t* lengths, int64_t E, int64_t* prefix) {
    __shared__ int64_t wave_sum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t e0 = 0; e0 < E; e0 += blockDim.x) {
        const int64_t e = e0 + threadIdx.x;
        const int64_t v = e < E ? lengths[e] : 0;
        int64_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < n_waves; ++w) {
            if (w < wave) woff += wave_sum[w];
            tot += wave_sum[w];
        }
        const int64_t carry = carry_s;
        if (e < E) prefix[e] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) prefix[E] = carry_s;
}

__global__ void sample_all_kernel(const int64_t* offset, int64_t E, const int64_t* lengths,
                                  const int64_t* insertion, const int64_t* prefix, int64_t total,
                                  int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        // sub-buffer e with prefix[e] <= p < prefix[e+1] (empty sub-buffers have equal bounds)
        int64_t lo = 0, hi = E;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t e = lo;
        const int64_t j = p - prefix[e];
        const int64_t len = lengths[e];
        // [insertion, len) ++ [0, insertion)   (buffer_base.py:519-524); insertion == len when
        // the sub-buffer has not wrapped yet, which makes the first range empty.
        int64_t slot = insertion[e] + j;
        if (slot >= len) slot -= len;
        out[p] = offset[e] + slot;
    }
}

// ReplayBufferManager.sample_indices(batch_size > 0) (manager.py:216-234) on the device, single workgroup:
//   buffer_idx = RandomState.choice(E, bs, p = lengths / lengths.sum())     == cdf.searchsorted(u, side="right") with
//                cdf = p.cumsum(); cdf /= cdf[-1]  (NumPy's legacy choice), u = the bs uniform draws it consumes;
//   sample_num = bincount(buffer_idx);  output = concat over sub-buffers e of offset[e] + child.choice(len_e, n_e),
//   i.e. position j of the output belongs to the sub-buffer e with start[e] <= j < start[e + 1] (start = exclusive
//   prefix of sample_num) and takes the j-th within-buffer draw.  The draws are inputs: `within_i` (the reference's own
//   randint values, for parity) or `within_u` (uniforms in [0, 1): slot = floor(u * len_e), the device-RNG fast path).
// The cdf is built sequentially in float64 by one thread, in NumPy's operation order (E is at most a few thousand).
constexpr int SAMPLE_MAX_E = 4096;

// Philox-4x32-10 (Salmon et al., SC'11): the counter-based generator of ts_normal_fill, here for the engine's own uniform
// draws.  uniform53: a double in [0, 1) from 53 random bits (NumPy's random_sample recipe) of counter (index, stream).
__device__ __forceinline__ void philox10(uint32_t (&c)[4], uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ double uniform53_of(uint32_t w0, uint32_t w1) {
    const uint32_t a = w0 >> 5, b = w1 >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double uniform53(uint64_t seed, uint64_t stream, int64_t index, int which) {
    uint32_t c[4] = {(uint32_t)index, (uint32_t)((uint64_t)index >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    philox10(c, seed);
    return uniform53_of(c[2 * which], c[2 * which + 1]);
}

constexpr int SAMPLE_STAGED_E = 512;      // lengths / offsets of up to this many sub-buffers are staged in LDS

__global__ __launch_bounds__(1024) void sample_random_kernel(const int64_t* __restrict__ offset, int64_t E,
                                                             const int64_t* __restrict__ lengths,
                                                             const double* __restrict__ u_buffer,
                                                             const int64_t* __restrict__ within_i,
                                                             const double* __restrict__ within_u, int64_t bs,
                                                             int64_t* __restrict__ out, int* __restrict__ err,
                                                             uint64_t seed = 0, uint64_t stream = 0) {
    __shared__ double cdf[SAMPLE_MAX_E];
    __shared__ int count[SAMPLE_MAX_E + 1];
    __shared__ int64_t len_s[SAMPLE_STAGED_E], off_s[SAMPLE_STAGED_E];
    __shared__ int64_t wsum[16];
    __shared__ int wave_count[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool staged = E <= SAMPLE_STAGED_E;           // the output loop reads lengths / offsets from LDS
    // total = lengths.sum(): exact in any order (integers) -- every thread its strided share, wave shuffles, 16 wave sums
    int64_t part = 0;
    for (int e = tid; e < E; e += 1024) {
        const int64_t l = lengths[e];
        part += l;
        if (staged) { len_s[e] = l; off_s[e] = offset[e]; }
    }
    for (int e = tid; e <= E; e += 1024) count[e] = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    int64_t total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) total += wsum[w];
    // p = lengths / total in parallel; p.cumsum() by ONE thread in NumPy's order (sequential float64 adds: the rounding of
    // a parallel scan would differ); cdf /= cdf[-1] in parallel.  (The serial pass used to load lengths[e] from global
    // memory inside its loop: 160 ns per sub-buffer, 80 us at 512 sub-buffers.)
    for (int e = tid; e < E; e += 1024) cdf[e] = (double)(staged ? len_s[e] : lengths[e]) / (double)total;
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        for (int64_t e = 0; e < E; ++e) { acc += cdf[e]; cdf[e] = acc; }
        if (total <= 0) *err = 1;
    }
    __syncthreads();
    const double last = cdf[E - 1];
    __syncthreads();
    for (int e = tid; e < E; e += 1024) cdf[e] /= last;
    __syncthreads();
    // Histogram of the draws over the sub-buffers.  Few sub-buffers (<= 64): lane e of every wave counts the wave's draws of
    // sub-buffer e in a register (one ballot per sub-buffer), the 16 waves' counts meet in LDS once at the end -- LDS
    // atomics on 16 counters serialise on their addresses (11 of this kernel's 19 us at the C5 batch); many sub-buffers:
    // one atomic per draw, contention is low there.
    const bool few = E <= 64;
    int mine = 0;
    // the engine's own draws: one Philox block per index holds both of its uniforms (words 0-1: the sub-buffer draw, words
    // 2-3: the within-buffer draw of output position k) -- the first four rounds' second halves are kept for the output loop
    uint32_t keep[4][2] = {};
    for (int64_t k0 = 0; k0 < bs; k0 += 4096) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // (compile-time q: `keep` stays in registers)
            if (k0 + 1024 * q >= bs) break;            // uniform
            const int64_t k = k0 + 1024 * q + tid;
            int lo = -1;
            if (k < bs) {
                double u;
                if (u_buffer) u = u_buffer[k];
                else {
                    uint32_t c[4] = {(uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
                    philox10(c, seed);
                    u = uniform53_of(c[0], c[1]);
                    if (k0 == 0) { keep[q][0] = c[2]; keep[q][1] = c[3]; }
                }
                int l = 0, hi = (int)E;                // first e with cdf[e] > u  (searchsorted side="right")
                while (l < hi) { const int mid = (l + hi) >> 1; if (cdf[mid] <= u) l = mid + 1; else hi = mid; }
                lo = l >= (int)E ? (int)E - 1 : l;
                if (!few) atomicAdd(&count[lo], 1);
            }
            if (few) {
                for (int e = 0; e < (int)E; ++e) {
                    const unsigned long long m = __ballot(lo == e);
                    mine += lane == e ? __popcll(m) : 0;
                }
            }
        }
    }
    if (few) {
        wave_count[wave][lane] = mine;
        __syncthreads();
        if (tid < E) {
            int c = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) c += wave_count[w][tid];
            count[tid] = c;
        }
    }
    __syncthreads();
    if (tid == 0) {                                    // exclusive prefix, in place
        int run = 0;
        for (int64_t e = 0; e <= E; ++e) { const int c = count[e]; count[e] = run; run += c; }
    }
    __syncthreads();
    // four outputs per thread and round: the sub-buffer searches first, then the four draws' loads side by side
    for (int64_t j0 = 0; j0 < bs; j0 += 4096) {
        int los[4];
        int64_t lens[4], offs[4], wi[4];
        double wu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 1024 * q + tid;
            int lo = 0, hi = (int)E;                   // last e with start[e] <= j
            if (j < bs)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (count[mid] <= (int)j) lo = mid; else hi = mid; }
            los[q] = lo;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 1024 * q + tid;
            const int64_t jc = j < bs ? j : bs - 1;
            lens[q] = staged ? len_s[los[q]] : lengths[los[q]];
            offs[q] = staged ? off_s[los[q]] : offset[los[q]];
            wi[q] = within_i ? within_i[jc] : 0;
            wu[q] = within_i ? 0.0 : (within_u ? within_u[jc] : (j0 == 0 && !u_buffer ? uniform53_of(keep[q][0], keep[q][1])
                                                                                      : uniform53(seed, stream, jc, 1)));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 1024 * q + tid;
            if (j >= bs) continue;
            const int64_t len = lens[q];
            int64_t slot;
            if (within_i) slot = wi[q];
            else { slot = (int64_t)(wu[q] * (double)len); if (slot >= len) slot = len - 1; }
            if (slot < 0 || slot >= len) *err = 2;
            out[j] = offs[q] + slot;
        }
    }
}

template <typename V>
__global__ void gather_rows_vec_kernel(const V* src, int64_t n_src, int64_t row_vecs,
                                       const int64_t* index, int64_t I, V* out) {
    const int64_t total = I * row_vecs;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / row_vecs, k = t - i * row_vecs;
        int64_t r = index[i];
        if (r < 0) r += n_src;  // NumPy negative indexing
        out[t] = src[r * row_vecs + k];
    }
}

// Batch.__getitem__ over several keys (data/batch.py:714-738: every key gathered at the same indices) in ONE launch:
// blockIdx.y = key, 4-byte words.
constexpr int GATHER_MULTI_MAX = 8;
struct GatherMulti { const uint32_t* src[GATHER_MULTI_MAX]; uint32_t* out[GATHER_MULTI_MAX]; int64_t row_words[GATHER_MULTI_MAX]; };
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherMulti g, int64_t n_src, const int64_t* __restrict__ index,
                                                                int64_t I) {
    const int a = blockIdx.y;
    const int64_t rw = g.row_words[a], total = I * rw;
    const uint32_t* __restrict__ src = g.src[a];
    uint32_t* __restrict__ out = g.out[a];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / rw, k = t - i * rw;
        int64_t r = index[i];
        if (r < 0) r += n_src;  // NumPy negative indexing
        out[t] = src[r * rw + k];
    }
}

// Keyed bijection of [0, n): 4-round Feistel network on the smallest even bit width covering n,
// cycle-walked back into range.  One thread per output slot, no sort, no scratch.
__device__ __forceinline__ uint32_t feistel_round(uint32_t x, uint32_t k) {
    x ^= k;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}

__global__ void random_permutation_kernel(int64_t* out, int64_t n, int half_bits, uint64_t seed) {
    const uint32_t mask = (1u << half_bits) - 1u;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), k2 = k0 * 0xC2B2AE3Du + 0x27D4EB2Fu,
                   k3 = k1 * 0x165667B1u + 0x9E3779B9u;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t v = (uint64_t)i;
        do {   // cycle walking: the domain is a power of 4 >= n, expected < 4 iterations
            uint32_t l = (uint32_t)(v >> half_bits) & mask, r = (uint32_t)v & mask;
            uint32_t t;
            t = l ^ (feistel_round(r, k0) & mask); l = r; r = t;
            t = l ^ (feistel_round(r, k1) & mask); l = r; r = t;
            t = l ^ (feistel_round(r, k2) & mask); l = r; r = t;
            t = l ^ (feistel_round(r, k3) & mask); l = r; r = t;
            v = ((uint64_t)l << half_bits) | r;
        } while ((int64_t)v >= n);
        out[i] = (int64_t)v;
    }
}

// Standard-normal noise for rsample() (torch.distributions.Normal.rsample: loc + eps * scale, eps ~ N(0, 1); SAC
// sac.py:228-236 draws one [B, act_dim] block per forward): counter-based Philox-4x32-10 (Salmon et al., SC'11) keyed by
// `seed`, counter = (element quad, stream offset), two Box-Muller pairs per counter -> four floats per thread.  The same
// (seed, offset) always gives the same numbers whatever the launch shape; not torch's generator stream.
// (ts::philox_round / ts::normal4 live in ts_common.h: ts_sac_learn_rows' packing launch draws the same numbers)
using ts::philox_round;

// np.random.rand(n) of the engine's own stream: u[i] = uniform53(seed, counter, i, 0) (PrioritizedReplayBuffer.sample_indices,
// prio.py:65, inside ts_dqn_learn_step)
__global__ __launch_bounds__(256) void uniform_fill_f64_kernel(double* __restrict__ out, int64_t n, uint64_t seed, uint64_t counter,
                                                               const uint64_t* __restrict__ counter_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (counter_dev) counter = *counter_dev;
    if (i < n) out[i] = uniform53(seed, counter, i, 0);
}

__global__ __launch_bounds__(256) void normal_fill_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= n) return;
    float z[4];
    ts::normal4(q, seed, offset, z);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * q + e < n) out[4 * q + e] = z[e];
}

inline int grid_for(int64_t n, int block) {
    int64_t g = ts::ceil_div(n, block);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

namespace ts {
int uniform_fill_f64(double* out, int64_t n, uint64_t seed, uint64_t counter, const uint64_t* counter_dev, hipStream_t s) {
    hipLaunchKernelGGL(uniform_fill_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, out, n, seed, counter, counter_dev);
    TS_LAUNCH_CHECK();
    return TS_OK;
}
}  // namespace ts

extern "C" {

static int step_index(bool next, const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                      const uint8_t* done, const int64_t* last_index, const int64_t* lengths,
                      int64_t* out, ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && E >= 1, TS_ERR_INVALID_ARG, "ts_%s_index: bad size", next ? "next" : "prev");
    if (I == 0) return TS_OK;
    TS_REQUIRE(index && offset && done && last_index && lengths && out, TS_ERR_INVALID_ARG,
               "ts_%s_index: NULL array argument", next ? "next" : "prev");
    hipStream_t s = ts::as_stream(stream);
    if (next)
        hipLaunchKernelGGL(step_index_kernel<true>, dim3(grid_for(I, 256)), dim3(256), 0, s, index,
                           I, offset, E, done, last_index, lengths, out);
    else
        hipLaunchKernelGGL(step_index_kernel<false>, dim3(grid_for(I, 256)), dim3(256), 0, s, index,
                           I, offset, E, done, last_index, lengths, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_next_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                  const uint8_t* done, const int64_t* last_index, const int64_t* lengths,
                  int64_t* out, ts_stream_t stream) {
    return step_index(true, index, I, offset, E, done, last_index, lengths, out, stream);
}

int ts_prev_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                  const uint8_t* done, const int64_t* last_index, const int64_t* lengths,
                  int64_t* out, ts_stream_t stream) {
    return step_index(false, index, I, offset, E, done, last_index, lengths, out, stream);
}

int ts_unfinished_index(const int64_t* offset, int64_t E, const uint8_t* done,
                        const int64_t* last_index, const int64_t* lengths, int64_t* out,
                        int64_t* n_out, ts_stream_t stream) {
    (void)offset;
    TS_REQUIRE(E >= 1, TS_ERR_INVALID_ARG, "ts_unfinished_index: E must be >= 1");
    TS_REQUIRE(done && last_index && lengths && out && n_out, TS_ERR_INVALID_ARG,
               "ts_unfinished_index: NULL array argument");
    hipLaunchKernelGGL(unfinished_kernel, dim3(1), dim3(1024), 0, ts::as_stream(stream), E, done,
                       last_index, lengths, out, n_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_sample_indices_all(ts_workspace* ws, const int64_t* offset, int64_t E,
                          const int64_t* lengths, const int64_t* insertion, int64_t total,
                          int64_t* out, ts_stream_t stream) {
    TS_REQUIRE(E >= 1 && total >= 0, TS_ERR_INVALID_ARG, "ts_sample_indices_all: bad size");
    if (total == 0) return TS_OK;
    TS_REQUIRE(offset && lengths && insertion && out, TS_ERR_INVALID_ARG,
               "ts_sample_indices_all: NULL array argument");
    int rc = ts::ws_reserve(ws, sizeof(int64_t) * (size_t)(E + 1));
    if (rc != TS_OK) return rc;
    int64_t* prefix = reinterpret_cast<int64_t*>(ws->base);
    hipStream_t s = ts::as_stream(stream);
    hipLaunchKernelGGL(lengths_prefix_kernel, dim3(1), dim3(1024), 0, s, lengths, E, prefix);
    hipLaunchKernelGGL(sample_all_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, offset, E,
                       lengths, insertion, prefix, total, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_sample_indices_random(const int64_t* offset, int64_t E, const int64_t* lengths, const double* u_buffer,
                             const int64_t* within_i, const double* within_u, int64_t batch_size, int64_t* out,
                             int* err_flag, ts_stream_t stream) {
    TS_REQUIRE(E >= 1 && E <= SAMPLE_MAX_E, TS_ERR_UNSUPPORTED, "ts_sample_indices_random: 1 <= E <= %d sub-buffers", SAMPLE_MAX_E);
    TS_REQUIRE(batch_size >= 0, TS_ERR_INVALID_ARG, "ts_sample_indices_random: negative batch_size");
    if (batch_size == 0) return TS_OK;
    TS_REQUIRE(offset && lengths && u_buffer && out && err_flag && ((within_i != nullptr) != (within_u != nullptr)),
               TS_ERR_INVALID_ARG, "ts_sample_indices_random: NULL argument, or not exactly one of within_i / within_u");
    hipLaunchKernelGGL(sample_random_kernel, dim3(1), dim3(1024), 0, ts::as_stream(stream), offset, E, lengths,
                       u_buffer, within_i, within_u, batch_size, out, err_flag);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_sample_indices_seeded(const int64_t* offset, int64_t E, const int64_t* lengths, uint64_t seed, uint64_t counter,
                             int64_t batch_size, int64_t* out, int* err_flag, ts_stream_t stream) {
    TS_REQUIRE(E >= 1 && E <= SAMPLE_MAX_E, TS_ERR_UNSUPPORTED, "ts_sample_indices_seeded: 1 <= E <= %d sub-buffers", SAMPLE_MAX_E);
    TS_REQUIRE(batch_size >= 0, TS_ERR_INVALID_ARG, "ts_sample_indices_seeded: negative batch_size");
    if (batch_size == 0) return TS_OK;
    TS_REQUIRE(offset && lengths && out && err_flag, TS_ERR_INVALID_ARG, "ts_sample_indices_seeded: NULL argument");
    hipLaunchKernelGGL(sample_random_kernel, dim3(1), dim3(1024), 0, ts::as_stream(stream), offset, E, lengths,
                       (const double*)nullptr, (const int64_t*)nullptr, (const double*)nullptr, batch_size, out, err_flag, seed,
                       counter);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_random_permutation(int64_t* out, int64_t n, uint64_t seed, ts_stream_t stream) {
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_random_permutation: negative n");
    if (n == 0) return TS_OK;
    TS_REQUIRE(out != nullptr, TS_ERR_INVALID_ARG, "ts_random_permutation: out is NULL");
    TS_REQUIRE(n <= ((int64_t)1 << 40), TS_ERR_UNSUPPORTED, "ts_random_permutation: n too large");
    int bits = 2;
    while (((int64_t)1 << bits) < n) bits += 2;        // even width so that both halves are equal
    hipLaunchKernelGGL(random_permutation_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                       ts::as_stream(stream), out, n, bits / 2, seed);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, ts_stream_t stream) {
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_normal_fill: negative n");
    if (n == 0) return TS_OK;
    TS_REQUIRE(out != nullptr, TS_ERR_INVALID_ARG, "ts_normal_fill: out is NULL");
    hipLaunchKernelGGL(normal_fill_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, ts::as_stream(stream), out, n,
                       seed, offset);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_uniform_fill_f64(double* out, int64_t n, uint64_t seed, uint64_t counter, ts_stream_t stream) {
    TS_REQUIRE(n >= 0, TS_ERR_INVALID_ARG, "ts_uniform_fill_f64: negative n");
    if (n == 0) return TS_OK;
    TS_REQUIRE(out != nullptr, TS_ERR_INVALID_ARG, "ts_uniform_fill_f64: out is NULL");
    return ts::uniform_fill_f64(out, n, seed, counter, nullptr, ts::as_stream(stream));
}

int ts_gather_rows(const void* src, int64_t n_rows_src, int64_t row_bytes, const int64_t* index,
                   int64_t I, void* out, ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && row_bytes >= 0 && n_rows_src >= 0, TS_ERR_INVALID_ARG,
               "ts_gather_rows: negative size");
    if (I == 0 || row_bytes == 0) return TS_OK;
    TS_REQUIRE(src && index && out, TS_ERR_INVALID_ARG, "ts_gather_rows: NULL array argument");
    hipStream_t s = ts::as_stream(stream);
    const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out) |
                        (uintptr_t)row_bytes;
    if ((a & 15u) == 0) {
        const int64_t rv = row_bytes / 16;
        hipLaunchKernelGGL(gather_rows_vec_kernel<uint4>, dim3(grid_for(I * rv, 256)), dim3(256), 0,
                           s, (const uint4*)src, n_rows_src, rv, index, I, (uint4*)out);
    } else if ((a & 3u) == 0) {
        const int64_t rv = row_bytes / 4;
        hipLaunchKernelGGL(gather_rows_vec_kernel<uint32_t>, dim3(grid_for(I * rv, 256)), dim3(256),
                           0, s, (const uint32_t*)src, n_rows_src, rv, index, I, (uint32_t*)out);
    } else {
        hipLaunchKernelGGL(gather_rows_vec_kernel<uint8_t>, dim3(grid_for(I * row_bytes, 256)),
                           dim3(256), 0, s, (const uint8_t*)src, n_rows_src, row_bytes, index, I,
                           (uint8_t*)out);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_gather_rows_multi(int64_t n_keys, const void* const* h_src, const int64_t* h_row_bytes, int64_t n_rows_src,
                         const int64_t* index, int64_t I, void* const* h_out, ts_stream_t stream) {
    TS_REQUIRE(n_keys >= 1 && n_keys <= GATHER_MULTI_MAX && I >= 0 && n_rows_src >= 0, TS_ERR_INVALID_ARG,
               "ts_gather_rows_multi: 1 .. %d keys, non-negative sizes", GATHER_MULTI_MAX);
    if (I == 0) return TS_OK;
    TS_REQUIRE(h_src && h_row_bytes && h_out && index, TS_ERR_INVALID_ARG, "ts_gather_rows_multi: NULL argument");
    GatherMulti g{};
    int64_t max_words = 0;
    for (int k = 0; k < (int)n_keys; ++k) {
        TS_REQUIRE(h_src[k] && h_out[k] && h_row_bytes[k] >= 4 && h_row_bytes[k] % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(h_src[k]) | reinterpret_cast<uintptr_t>(h_out[k])) & 3u) == 0,
                   TS_ERR_INVALID_ARG, "ts_gather_rows_multi: key %d: rows of whole, 4-byte aligned words only", k);
        g.src[k] = static_cast<const uint32_t*>(h_src[k]);
        g.out[k] = static_cast<uint32_t*>(h_out[k]);
        g.row_words[k] = h_row_bytes[k] / 4;
        max_words = std::max(max_words, g.row_words[k]);
    }
    hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(grid_for(I * max_words, 256), (unsigned)n_keys), dim3(256), 0,
                       ts::as_stream(stream), g, n_rows_src, index, I);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_stack_indices(const int64_t* index, int64_t I, int64_t stack_num, const int64_t* offset, int64_t E,
                     const uint8_t* done, const int64_t* last_index, const int64_t* lengths, int64_t* out,
                     ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && stack_num >= 1 && E >= 1, TS_ERR_INVALID_ARG, "ts_stack_indices: bad sizes");
    if (I == 0) return TS_OK;
    TS_REQUIRE(index && offset && done && last_index && lengths && out, TS_ERR_INVALID_ARG,
               "ts_stack_indices: NULL argument");
    int64_t blocks = ts::ceil_div(I, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(stack_indices_kernel, dim3((unsigned)blocks), dim3(256), 0, ts::as_stream(stream), index, I,
                       stack_num, offset, E, done, last_index, lengths, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_dqn_gather_pair(const uint8_t* frames, int64_t n_planes, int64_t plane_elems, const int64_t* index, int64_t B,
                       int64_t n_step, int64_t stack_num, const int64_t* offset, int64_t E, const uint8_t* done,
                       const int64_t* last_index, const int64_t* lengths, uint8_t* obs_out, uint8_t* obs_next_out,
                       ts_stream_t stream) {
    TS_REQUIRE(B >= 0 && n_step >= 1 && E >= 1 && plane_elems >= 1 && n_planes >= 1, TS_ERR_INVALID_ARG,
               "ts_dqn_gather_pair: bad sizes");
    TS_REQUIRE(stack_num == 4 && plane_elems % 16 == 0, TS_ERR_UNSUPPORTED,
               "ts_dqn_gather_pair: stack_num 4 and plane sizes that are multiples of 16 bytes (use ts_stack_indices + "
               "ts_gather_planes_nhwc_u8 otherwise)");
    if (B == 0) return TS_OK;
    TS_REQUIRE(frames && index && offset && done && last_index && lengths && obs_out && obs_next_out, TS_ERR_INVALID_ARG,
               "ts_dqn_gather_pair: NULL argument");
    TS_REQUIRE(((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(obs_out) |
                 reinterpret_cast<uintptr_t>(obs_next_out)) & 15u) == 0, TS_ERR_INVALID_ARG,
               "ts_dqn_gather_pair: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(dqn_gather_pair_kernel, dim3((unsigned)B), dim3(256), 0, ts::as_stream(stream), frames, plane_elems, index,
                       (int)n_step, offset, E, done, last_index, lengths, obs_out, obs_next_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_stacked_rows_pair(const float* rows, const float* rows_next, int64_t n_rows, int64_t row_elems, const int64_t* index,
                         int64_t B, int64_t n_step, int64_t stack_num, const int64_t* offset, int64_t E, const uint8_t* done,
                         const int64_t* last_index, const int64_t* lengths, const int64_t* act_col, float* obs_out,
                         float* obs_next_out, int64_t* act_out, ts_stream_t stream) {
    TS_REQUIRE(B >= 0 && n_step >= 1 && E >= 1 && row_elems >= 1 && n_rows >= 1, TS_ERR_INVALID_ARG,
               "ts_stacked_rows_pair: bad sizes");
    TS_REQUIRE(stack_num >= 1 && stack_num <= ROWS_T_MAX && row_elems * stack_num <= (1 << 20), TS_ERR_UNSUPPORTED,
               "ts_stacked_rows_pair: 1 <= stack_num <= %d (use ts_stack_indices + ts_gather_rows otherwise)", ROWS_T_MAX);
    if (B == 0) return TS_OK;
    TS_REQUIRE(rows && index && offset && done && last_index && lengths && obs_out && obs_next_out &&
                   (act_col == nullptr) == (act_out == nullptr), TS_ERR_INVALID_ARG, "ts_stacked_rows_pair: NULL argument");
    hipLaunchKernelGGL(stacked_rows_pair_kernel, dim3((unsigned)ts::ceil_div(B, 4)), dim3(256), 0, ts::as_stream(stream), rows,
                       rows_next, (int)row_elems, index, B, (int)n_step, (int)stack_num, offset, E, done, last_index, lengths,
                       act_col, obs_out, obs_next_out, act_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_gather_planes_nhwc_u8(const uint8_t* src, int64_t n_planes, int64_t plane_elems, const int64_t* plane_index,
                             int64_t B, int64_t C, uint8_t* out, ts_stream_t stream) {
    TS_REQUIRE(B >= 0 && C >= 1 && C <= 64 && plane_elems >= 1 && n_planes >= 1, TS_ERR_INVALID_ARG,
               "ts_gather_planes_nhwc_u8: bad sizes");
    if (B == 0) return TS_OK;
    TS_REQUIRE(src && plane_index && out, TS_ERR_INVALID_ARG, "ts_gather_planes_nhwc_u8: NULL argument");
    TS_REQUIRE(B <= 65535, TS_ERR_UNSUPPORTED, "ts_gather_planes_nhwc_u8: at most 65535 rows per call");
    if (C == 4 && plane_elems % 16 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        const int64_t groups = B * (plane_elems / 16);
        const unsigned blocks = (unsigned)std::min<int64_t>(ts::ceil_div(groups, 256), 256 * 32);
        hipLaunchKernelGGL(gather_planes_u8x16_kernel, dim3(blocks), dim3(256), 0, ts::as_stream(stream), src, plane_elems,
                           plane_index, B, out);
        TS_LAUNCH_CHECK();
        return TS_OK;
    }
    int64_t bx = ts::ceil_div(plane_elems, 256);
    if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)B);
    if (C == 4)
        hipLaunchKernelGGL(gather_planes_u8_kernel<4>, grid, dim3(256), 0, ts::as_stream(stream), src, plane_elems,
                           plane_index, B, 4, out);
    else
        hipLaunchKernelGGL(gather_planes_u8_kernel<0>, grid, dim3(256), 0, ts::as_stream(stream), src, plane_elems,
                           plane_index, B, (int)C, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_gather_planes_nhwc(const uint8_t* src, int64_t n_planes, int64_t plane_elems, const int64_t* plane_index,
                          int64_t B, int64_t C, float* out, ts_stream_t stream) {
    TS_REQUIRE(B >= 0 && C >= 1 && C <= 64 && plane_elems >= 1 && n_planes >= 1, TS_ERR_INVALID_ARG,
               "ts_gather_planes_nhwc: bad sizes");
    if (B == 0) return TS_OK;
    TS_REQUIRE(src && plane_index && out, TS_ERR_INVALID_ARG, "ts_gather_planes_nhwc: NULL argument");
    TS_REQUIRE(B <= 65535, TS_ERR_UNSUPPORTED, "ts_gather_planes_nhwc: at most 65535 rows per call");
    int64_t bx = ts::ceil_div(plane_elems, 256);
    if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)B);
    if (C == 4)
        hipLaunchKernelGGL(gather_planes_kernel<4>, grid, dim3(256), 0, ts::as_stream(stream), src, plane_elems,
                           plane_index, B, 4, out);
    else
        hipLaunchKernelGGL(gather_planes_kernel<0>, grid, dim3(256), 0, ts::as_stream(stream), src, plane_elems,
                           plane_index, B, (int)C, out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_buffer_add(const int64_t* buffer_ids, int64_t K, const double* rew, const uint8_t* terminated,
                  const uint8_t* truncated, const int64_t* offset, int64_t E, int64_t* insertion, int64_t* lengths,
                  int64_t* last_index, double* ep_return, int64_t* ep_len, int64_t* ep_start, double* rew_B,
                  uint8_t* terminated_B, uint8_t* truncated_B, uint8_t* done_B, const ts_scatter_key* h_keys,
                  int n_keys, int64_t* index_out, double* ep_return_out, int64_t* ep_len_out, int64_t* ep_start_out,
                  ts_stream_t stream) {
    TS_REQUIRE(K >= 0 && E >= 1 && n_keys >= 0 && n_keys <= 8, TS_ERR_INVALID_ARG, "ts_buffer_add: bad sizes");
    TS_REQUIRE(buffer_ids || K <= E, TS_ERR_SHAPE, "ts_buffer_add: more entries than sub-buffers");
    if (K == 0) return TS_OK;
    TS_REQUIRE(rew && terminated && truncated && offset && insertion && lengths && last_index && ep_return && ep_len &&
                   ep_start && rew_B && terminated_B && truncated_B && done_B && index_out && ep_return_out &&
                   ep_len_out && ep_start_out && (h_keys || n_keys == 0),
               TS_ERR_INVALID_ARG, "ts_buffer_add: NULL argument");
    hipStream_t s = ts::as_stream(stream);
    AddArgs a{buffer_ids, K, rew, terminated, truncated, offset, insertion, lengths, last_index, ep_return, ep_len,
              ep_start, rew_B, terminated_B, truncated_B, done_B, index_out, ep_return_out, ep_len_out, ep_start_out};
    hipLaunchKernelGGL(buffer_add_state_kernel, dim3((unsigned)ts::ceil_div(K, 256)), dim3(256), 0, s, a);
    if (n_keys > 0) {
        ScatterKeys sk{};
        sk.n = n_keys;
        int64_t max_bytes = 0;
        for (int i = 0; i < n_keys; ++i) {
            TS_REQUIRE(h_keys[i].dst && h_keys[i].src && h_keys[i].row_bytes >= 1, TS_ERR_INVALID_ARG,
                       "ts_buffer_add: bad scatter key");
            sk.dst[i] = h_keys[i].dst; sk.src[i] = h_keys[i].src; sk.row_bytes[i] = h_keys[i].row_bytes;
            max_bytes = std::max(max_bytes, h_keys[i].row_bytes);
        }
        int64_t bx = ts::ceil_div(K * ts::ceil_div(max_bytes, 16), 256);
        if (bx > 4096) bx = 4096;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)bx, (unsigned)n_keys), dim3(256), 0, s, sk, index_out, K);
    }
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
