// ts_returns.hip -- GAE (segmented reverse affine scan) and n-step return kernels for gfx950.
//
// Replaces Algorithm.compute_episodic_return / _gae and compute_nstep_return / _nstep_return
// (tianshou/algorithm/algorithm_base.py:653-719,1085-1140 and :721-817,1160-1222).
//
// Roofline: HBM.  Algorithmic traffic of the GAE scan = 22 B / transition
// (v_s 4 + v_s_ 4 + rew 4 + terminated 1 + truncated 1 read, adv 4 + returns 4 written;
// 26 B when rew is float64).  The arithmetic is float64 per element (as numba's) which is far
// below the 78 TF/s f64 vector rate at these byte counts.
//
// Scan formulation: A_i = d_i + c_i * A_{i+1} is the affine map x -> d_i + c_i x.  Maps compose
// associatively, (a,b) o (a',b') = (a a', b + a b'), so a tile of TILE consecutive transitions
// collapses to one map.  Pass 1 writes one map per tile; pass 2 folds the maps of all later
// tiles into the tile's carry-in (early exit once the product of c's is exactly 0, i.e. at the
// first episode end), re-scans the tile from registers and writes adv / returns.
//
// No FMA contraction in this file: the reference (numba, NumPy) rounds mul and add separately,
// and the n-step path is required to be bit-exact in float64.
#include <algorithm>
#include <cstdlib>

#include "ts_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int GAE_THREADS = 256;
constexpr int GAE_ITEMS = 8;
constexpr int GAE_TILE = GAE_THREADS * GAE_ITEMS;  // 2048 transitions per workgroup
constexpr int GAE_WAVES = GAE_THREADS / 64;

struct Aff {
    double a, b;  // x -> b + a * x
};

__device__ __forceinline__ Aff aff_identity() { return Aff{1.0, 0.0}; }

// value on the left of L's span given the value on the right of R's span
__device__ __forceinline__ Aff compose(const Aff& L, const Aff& R) {
    return Aff{L.a * R.a, L.b + L.a * R.b};
}

__device__ __forceinline__ Aff shfl_down_aff(const Aff& v, int off) {
    return Aff{__shfl_down(v.a, off, 64), __shfl_down(v.b, off, 64)};
}

// Ordered (non-commutative) reduction over the 64 lanes; valid in lane 0.
__device__ __forceinline__ Aff wave_reduce_aff(Aff v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Aff r = shfl_down_aff(v, off);
        if (lane + off < 64) v = compose(v, r);
    }
    return v;
}

// Inclusive suffix scan over the 64 lanes: lane l gets f_l o f_{l+1} o ... o f_63.
__device__ __forceinline__ Aff wave_suffix_scan_aff(Aff v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Aff r = shfl_down_aff(v, off);
        if (lane + off < 64) v = compose(v, r);
    }
    return v;
}

// Ordered reduction over the whole workgroup; result broadcast to every thread.
__device__ __forceinline__ Aff block_reduce_aff(Aff v, Aff* lds /*[GAE_WAVES]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Aff w = wave_reduce_aff(v, lane);
    __syncthreads();  // protect lds reuse between calls
    if (lane == 0) lds[wave] = w;
    __syncthreads();
    Aff t = lds[0];
#pragma unroll
    for (int k = 1; k < GAE_WAVES; ++k) t = compose(t, lds[k]);
    return t;
}

template <typename RewT>
struct GaeArgs {
    const float* v_s;
    const float* v_n;
    const RewT* rew;
    const uint8_t* term;
    const uint8_t* trunc;
    const int64_t* cut_pos;
    const int64_t* d_n_cut;
    int64_t n_cut;
    int64_t n;
    double gamma, gl, v_scale, ret_div;
    const uint32_t* cutbits;    // nullable: bit i set <=> position i is a cut (built by gae_cutbits_kernel when the
                                // cut list is too long for every tile to scan it)
};

// Raw inputs of the thread's GAE_ITEMS transitions (issued before anything that has to wait, so
// that their HBM latency overlaps the ticket / cut-list round trips).
struct GaeRaw {
    float fv[GAE_ITEMS], fn[GAE_ITEMS];
    double rw[GAE_ITEMS];
    uint8_t te[GAE_ITEMS], tr[GAE_ITEMS];
};

template <typename RewT, bool VEC>
__device__ __forceinline__ void gae_load_raw(const GaeArgs<RewT>& g, int64_t base, GaeRaw& r) {
    const bool full = base + GAE_ITEMS <= g.n;
    if (VEC && full) {
        const float4* pv = reinterpret_cast<const float4*>(g.v_s + base);
        const float4* pn = reinterpret_cast<const float4*>(g.v_n + base);
        float4 a0 = pv[0], a1 = pv[1], b0 = pn[0], b1 = pn[1];
        r.fv[0] = a0.x; r.fv[1] = a0.y; r.fv[2] = a0.z; r.fv[3] = a0.w;
        r.fv[4] = a1.x; r.fv[5] = a1.y; r.fv[6] = a1.z; r.fv[7] = a1.w;
        r.fn[0] = b0.x; r.fn[1] = b0.y; r.fn[2] = b0.z; r.fn[3] = b0.w;
        r.fn[4] = b1.x; r.fn[5] = b1.y; r.fn[6] = b1.z; r.fn[7] = b1.w;
        if constexpr (sizeof(RewT) == 4) {
            const float4* pr = reinterpret_cast<const float4*>(g.rew + base);
            float4 r0 = pr[0], r1 = pr[1];
            r.rw[0] = r0.x; r.rw[1] = r0.y; r.rw[2] = r0.z; r.rw[3] = r0.w;
            r.rw[4] = r1.x; r.rw[5] = r1.y; r.rw[6] = r1.z; r.rw[7] = r1.w;
        } else {
            const double2* pr = reinterpret_cast<const double2*>(g.rew + base);
#pragma unroll
            for (int k = 0; k < GAE_ITEMS / 2; ++k) {
                double2 q = pr[k];
                r.rw[2 * k] = q.x;
                r.rw[2 * k + 1] = q.y;
            }
        }
        const uint2 t8 = *reinterpret_cast<const uint2*>(g.term + base);
        const uint2 u8 = *reinterpret_cast<const uint2*>(g.trunc + base);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.te[k] = (t8.x >> (8 * k)) & 0xFF;
            r.te[4 + k] = (t8.y >> (8 * k)) & 0xFF;
            r.tr[k] = (u8.x >> (8 * k)) & 0xFF;
            r.tr[4 + k] = (u8.y >> (8 * k)) & 0xFF;
        }
    } else {
#pragma unroll
        for (int k = 0; k < GAE_ITEMS; ++k) {
            const int64_t i = base + k;
            const bool ok = i < g.n;
            r.fv[k] = ok ? g.v_s[i] : 0.f;
            r.fn[k] = ok ? g.v_n[i] : 0.f;
            r.rw[k] = ok ? (double)g.rew[i] : 0.0;
            r.te[k] = ok ? g.term[i] : (uint8_t)1;
            r.tr[k] = ok ? g.trunc[i] : (uint8_t)1;
        }
    }
}

// per-element (d, c) plus the scaled value vs needed for returns.  Elements past n behave like
// a finished episode with zero reward (c = 0, d = 0).
template <typename RewT>
__device__ __forceinline__ void gae_convert(const GaeArgs<RewT>& g, int64_t base, const GaeRaw& r,
                                            const uint32_t* cutmask, double (&vs)[GAE_ITEMS],
                                            double (&d)[GAE_ITEMS], double (&c)[GAE_ITEMS]) {
    const int local = (int)(base % GAE_TILE);
#pragma unroll
    for (int k = 0; k < GAE_ITEMS; ++k) {
        const int bit = local + k;
        const bool cut = (cutmask[bit >> 5] >> (bit & 31)) & 1u;
        const bool end = (r.te[k] != 0) | (r.tr[k] != 0) | cut;
        vs[k] = (double)r.fv[k] * g.v_scale;
        const double vn = ((double)r.fn[k] * g.v_scale) * (r.te[k] != 0 ? 0.0 : 1.0);
        d[k] = r.rw[k] + vn * g.gamma - vs[k];
        c[k] = end ? 0.0 : g.gl;
    }
}

template <typename RewT, bool VEC>
__device__ __forceinline__ void gae_load_items(const GaeArgs<RewT>& g, int64_t base,
                                               const uint32_t* cutmask, double (&vs)[GAE_ITEMS],
                                               double (&d)[GAE_ITEMS], double (&c)[GAE_ITEMS]) {
    GaeRaw r;
    gae_load_raw<RewT, VEC>(g, base, r);
    gae_convert(g, base, r, cutmask, vs, d, c);
}

constexpr int GAE_CUT_SCAN_MAX = 1024;     // longer cut lists go through the global bitmask

// Bitmask path: the tile's 64 cut words, one per lane (every wave loads the same 256 bytes).
struct GaeCutPre { uint32_t word; };

template <typename RewT>
__device__ __forceinline__ GaeCutPre gae_cut_prefetch(const GaeArgs<RewT>& g, int64_t tile_start) {
    const int64_t w = tile_start / 32 + (threadIdx.x & 63);
    const int64_t last = (g.n - 1) / 32;
    GaeCutPre c;
    c.word = g.cutbits[w < last ? w : last];
    if (w > last) c.word = 0u;
    return c;
}

template <typename RewT>
__device__ __forceinline__ void gae_cutmask_from_prefetch(const GaeArgs<RewT>&, int64_t, const GaeCutPre& c,
                                                          uint32_t* cutmask) {
    if (threadIdx.x < GAE_TILE / 32) cutmask[threadIdx.x] = c.word;
    __syncthreads();
}

// bits[i / 32] |= 1 << (i % 32) for every cut position i (the region is zeroed by the caller)
__global__ __launch_bounds__(256) void gae_cutbits_kernel(const int64_t* __restrict__ cut_pos, int64_t n_cut,
                                                          const int64_t* __restrict__ d_n_cut, int64_t n,
                                                          uint32_t* __restrict__ bits) {
    int64_t m = n_cut;
    if (d_n_cut) { const int64_t dn = *d_n_cut; m = dn < m ? dn : m; }
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < m; k += (int64_t)gridDim.x * 256) {
        const int64_t p = cut_pos[k];
        if (p >= 0 && p < n) atomicOr(&bits[p >> 5], 1u << (p & 31));
    }
}

template <typename RewT>
__device__ __forceinline__ void gae_build_cutmask(const GaeArgs<RewT>& g, int64_t tile_start,
                                                  uint32_t* cutmask) {
    if (g.cutbits) {
        const GaeCutPre c = gae_cut_prefetch(g, tile_start);
        gae_cutmask_from_prefetch(g, tile_start, c, cutmask);
        return;
    }
    for (int k = threadIdx.x; k < GAE_TILE / 32; k += GAE_THREADS) cutmask[k] = 0u;
    __syncthreads();
    int64_t n_cut = g.n_cut;
    if (g.d_n_cut) {
        const int64_t dn = *g.d_n_cut;
        n_cut = dn < n_cut ? dn : n_cut;
    }
    for (int64_t k = threadIdx.x; k < n_cut; k += GAE_THREADS) {
        const int64_t p = g.cut_pos[k] - tile_start;
        if (p >= 0 && p < GAE_TILE) atomicOr(&cutmask[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
}

__device__ __forceinline__ Aff items_to_aff(const double (&d)[GAE_ITEMS],
                                            const double (&c)[GAE_ITEMS]) {
    Aff f = aff_identity();
#pragma unroll
    for (int k = GAE_ITEMS - 1; k >= 0; --k) {
        f.b = d[k] + c[k] * f.b;
        f.a = c[k] * f.a;
    }
    return f;
}


template <bool VEC, typename RewT>
__device__ __forceinline__ void gae_store_outputs(const GaeArgs<RewT>& g, int64_t base,
                                                  const double (&adv)[GAE_ITEMS],
                                                  const double (&ret)[GAE_ITEMS], float* adv_out,
                                                  float* ret_out, double* adv64, double* ret64) {
    const bool full = base + GAE_ITEMS <= g.n;
    if (VEC && full) {
        float4 a0, a1, r0, r1;
        a0.x = (float)adv[0]; a0.y = (float)adv[1]; a0.z = (float)adv[2]; a0.w = (float)adv[3];
        a1.x = (float)adv[4]; a1.y = (float)adv[5]; a1.z = (float)adv[6]; a1.w = (float)adv[7];
        r0.x = (float)(ret[0] / g.ret_div); r0.y = (float)(ret[1] / g.ret_div);
        r0.z = (float)(ret[2] / g.ret_div); r0.w = (float)(ret[3] / g.ret_div);
        r1.x = (float)(ret[4] / g.ret_div); r1.y = (float)(ret[5] / g.ret_div);
        r1.z = (float)(ret[6] / g.ret_div); r1.w = (float)(ret[7] / g.ret_div);
        float4* pa = reinterpret_cast<float4*>(adv_out + base);
        float4* pr = reinterpret_cast<float4*>(ret_out + base);
        pa[0] = a0; pa[1] = a1;
        pr[0] = r0; pr[1] = r1;
    } else {
#pragma unroll
        for (int k = 0; k < GAE_ITEMS; ++k)
            if (base + k < g.n) {
                adv_out[base + k] = (float)adv[k];
                ret_out[base + k] = (float)(ret[k] / g.ret_div);
            }
    }
    if (adv64 || ret64) {
#pragma unroll
        for (int k = 0; k < GAE_ITEMS; ++k)
            if (base + k < g.n) {
                if (adv64) adv64[base + k] = adv[k];
                if (ret64) ret64[base + k] = ret[k];
            }
    }
}

__device__ __forceinline__ void gae_store_partials(double s1, double s2, int lane, int wave, int64_t tile,
                                                   double* red, double* ret_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if (lane == 0) {
        red[2 * wave] = s1;
        red[2 * wave + 1] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < GAE_WAVES; ++w) {
            t1 += red[2 * w];
            t2 += red[2 * w + 1];
        }
        ret_partials[2 * tile] = t1;
        ret_partials[2 * tile + 1] = t2;
    }
}

// ---------------------------------------------------------------------------------------------
// Single-pass scan: one launch, every input byte read once.  Workgroups claim tiles from the END
// of the array through a ticket (so a tile only ever waits for tiles claimed before it: no
// residency assumption, no deadlock), publish their tile map, and fold the maps of the following
// tiles as soon as those are published.  The maps do not depend on each other (unlike a prefix
// sum's running prefix), so there is no serial chain across tiles; the wait is normally one
// flag poll.  Hand-off per MI355X rules: 8-byte agent-scope (sc1, write-through) payload stores,
// drain, agent-scope flag store; consumer polls the flag relaxed at agent scope and reads the
// payload with agent-scope loads.  Flags carry the launch epoch, so nothing is reset between
// launches.
constexpr int GAE_SHARDS = 8;   // ticket counters (one word saturates at ~88 atomics/us)

struct GaeSyncHeader {
    struct alignas(128) { unsigned long long v; } ticket[GAE_SHARDS];   // shard k hands out sequence numbers 8 j + k;
                                                                        // one cache line each: atomics serialise per LINE
    unsigned int error;       // set when a spin ran out (results invalid, no hang)
    unsigned int pad[3];
};

struct GaeLaunch {
    unsigned long long base[GAE_SHARDS];     // running ticket base of every shard
    int direct;                              // 1: tile = f(blockIdx), grid small enough to be resident
};

struct GaeTileMsg {
    unsigned long long a_bits, b_bits;
    unsigned int flag;
    unsigned int pad;
};

constexpr unsigned GAE_SPIN_LIMIT = 1u << 22;


template <typename RewT, bool VEC>
__global__ __launch_bounds__(GAE_THREADS) void gae_single_pass(GaeArgs<RewT> g, GaeSyncHeader* hdr,
                                                               GaeTileMsg* msg, int64_t n_tiles,
                                                               GaeLaunch launch,
                                                               unsigned int epoch, float* adv_out,
                                                               float* ret_out, double* adv64,
                                                               double* ret64, double* ret_partials) {
    __shared__ uint32_t cutmask[GAE_TILE / 32];
    __shared__ Aff lds[GAE_WAVES];
    __shared__ double red[2 * GAE_WAVES];
    __shared__ int64_t tile_s;
    __shared__ int closed_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t tile;
    if (launch.direct) {
        // every workgroup of this launch is resident at once (small grid): no ticket, no broadcast
        tile = n_tiles - 1 - (int64_t)blockIdx.x;
    } else {
        if (threadIdx.x == 0) {
            const int shard = blockIdx.x & (GAE_SHARDS - 1);
            const unsigned long long t = __hip_atomic_fetch_add(&hdr->ticket[shard].v, 1ULL, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
            const int64_t seq = (int64_t)(t - launch.base[shard]) * GAE_SHARDS + shard;
            tile_s = n_tiles - 1 - seq;                        // sequence 0 -> last tile
        }
        __syncthreads();
        tile = tile_s;
    }
    const int64_t tile_start = tile * GAE_TILE;
    double vs[GAE_ITEMS], d[GAE_ITEMS], c[GAE_ITEMS];
    const int64_t base = tile_start + (int64_t)threadIdx.x * GAE_ITEMS;
    {
        // the tile's inputs are requested first; the cut information (its 64 words of the global bitmask, or a scan
        // of the short cut list -- measured faster than prefetching list entries per thread) overlaps their latency
        GaeRaw raw;
        if (g.cutbits) {
            const GaeCutPre cuts = gae_cut_prefetch(g, tile_start);
            gae_load_raw<RewT, VEC>(g, base, raw);
            gae_cutmask_from_prefetch(g, tile_start, cuts, cutmask);
        } else {
            gae_load_raw<RewT, VEC>(g, base, raw);
            gae_build_cutmask(g, tile_start, cutmask);
        }
        gae_convert(g, base, raw, cutmask, vs, d, c);
    }
    const Aff mine = items_to_aff(d, c);

    // in-tile suffix scan first (its wave maps also give the tile map)
    const Aff incl = wave_suffix_scan_aff(mine, lane);
    Aff excl = shfl_down_aff(incl, 1);
    if (lane == 63) excl = aff_identity();
    if (lane == 0) lds[wave] = incl;
    // A tile whose LAST transition ends an episode (or sits on a cut: the end of a sub-buffer, which is where power-of-two
    // tiles of a [envs x steps] rollout end) takes nothing from the tiles after it: no poll, no second round of barriers.
    if (threadIdx.x == GAE_THREADS - 1) closed_s = c[GAE_ITEMS - 1] == 0.0 ? 1 : 0;
    __syncthreads();
    const bool closed = closed_s != 0;
    Aff wmap[GAE_WAVES];
#pragma unroll
    for (int w = 0; w < GAE_WAVES; ++w) wmap[w] = lds[w];
    if (threadIdx.x == 0) {
        Aff t = wmap[0];
#pragma unroll
        for (int w = 1; w < GAE_WAVES; ++w) t = compose(t, wmap[w]);
        __hip_atomic_store(&msg[tile].a_bits, (unsigned long long)__double_as_longlong(t.a), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&msg[tile].b_bits, (unsigned long long)__double_as_longlong(t.b), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&msg[tile].flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // carry-in: fold the maps of the following tiles until an episode end.  Growing windows
    // (1, 15, 240, then 256 tiles per round): with episode ends in nearly every tile only the next
    // tile's flag is ever polled, while the worst case (no end at all) still costs O(n_tiles / 256)
    // rounds per tile.
    Aff acc = aff_identity();
    int64_t t0 = tile + 1;
    for (int round = 0; !closed && t0 < n_tiles; ++round) {
        const int width = round == 0 ? 1 : (round == 1 ? 15 : (round == 2 ? 240 : GAE_THREADS));
        Aff f = aff_identity();
        const int64_t t = t0 + threadIdx.x;
        if ((int)threadIdx.x < width && t < n_tiles) {
            unsigned spins = 0;
            while (__hip_atomic_load(&msg[t].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > GAE_SPIN_LIMIT) { hdr->error = 1u; break; }
            }
            const unsigned long long ab = __hip_atomic_load(&msg[t].a_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long bb = __hip_atomic_load(&msg[t].b_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f = Aff{__longlong_as_double((long long)ab), __longlong_as_double((long long)bb)};
        }
        acc = compose(acc, block_reduce_aff(f, lds));
        if (acc.a == 0.0) break;  // an episode ended: nothing further can leak in (uniform)
        t0 += width;
    }
    double wave_carry = acc.b;
#pragma unroll
    for (int w = GAE_WAVES - 1; w >= 0; --w)
        if (w > wave) wave_carry = wmap[w].b + wmap[w].a * wave_carry;
    double x = excl.b + excl.a * wave_carry;

    double adv[GAE_ITEMS], ret[GAE_ITEMS];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = GAE_ITEMS - 1; k >= 0; --k) {
        x = d[k] + c[k] * x;
        adv[k] = x;
        ret[k] = x + vs[k];
        if (base + k < g.n) {
            s1 += ret[k];
            s2 += ret[k] * ret[k];
        }
    }
    gae_store_outputs<VEC>(g, base, adv, ret, adv_out, ret_out, adv64, ret64);
    if (ret_partials) gae_store_partials(s1, s2, lane, wave, tile, red, ret_partials);
}

// ---------------------------------------------------------------------------------------------
__global__ void isin_positions_kernel(const int64_t* indices, int64_t n, const int64_t* unf,
                                      int64_t n_unf, int64_t* out, int64_t capacity,
                                      unsigned long long* count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t v = indices[i];
        int64_t lo = 0, hi = n_unf;  // binary search in the ascending list
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (unf[mid] < v) lo = mid + 1; else hi = mid;
        }
        if (lo < n_unf && unf[lo] == v) {
            const unsigned long long p = atomicAdd(count, 1ULL);
            if ((int64_t)p < capacity) out[p] = i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// n-step
__device__ __forceinline__ int64_t pymod(int64_t a, int64_t m) {
    const int64_t r = a % m;
    return r < 0 ? r + m : r;
}

__device__ __forceinline__ int64_t find_sub(const int64_t* offset, int64_t E, int64_t idx) {
    // largest e with offset[e] <= idx  (offset ascending, offset[0] == 0, idx < offset[E])
    int64_t lo = 0, hi = E;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offset[mid] <= idx) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int64_t next_one(int64_t idx, const int64_t* offset, int64_t E,
                                            const uint8_t* done, const int64_t* last_index,
                                            const int64_t* lengths, bool* is_end) {
    // manager.py:347-363
    idx = pymod(idx, offset[E]);
    const int64_t e = find_sub(offset, E, idx);
    const int64_t start = offset[e];
    const int64_t len = lengths[e];
    const int64_t cur_len = len > 1 ? len : 1;
    const int64_t end_flag = (done[idx] != 0) | (idx == last_index[e]);
    if (is_end) *is_end = (done[idx] != 0) | (len > 0 && idx == last_index[e]);
    return pymod(idx - start + 1 - end_flag, cur_len) + start;
}

__global__ void nstep_return_kernel(const double* rew, const uint8_t* end_flag, const float* tq,
                                    const int64_t* stacked, int64_t I, int64_t A, int64_t N,
                                    double gamma, float* out, double* out64) {
    const int64_t total = I * A;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / A;
        double mc = 0.0;
        int64_t gammas = N;
        for (int64_t n = N - 1; n >= 0; --n) {
            const int64_t now = stacked[n * I + i];
            if (end_flag[now]) {
                gammas = n + 1;
                mc = 0.0;
            }
            const double tmp = gamma * mc;
            mc = rew[now] + tmp;
        }
        double gpow = 1.0;
        for (int64_t k = 0; k < gammas; ++k) gpow = gpow * gamma;
        const double q = (double)tq[t] * gpow;
        const double r = q + mc;
        out[t] = (float)r;
        if (out64) out64[t] = r;
    }
}

__global__ void nstep_indices_kernel(const int64_t* indices, int64_t I, int64_t N,
                                     const int64_t* offset, int64_t E, const uint8_t* done,
                                     const int64_t* last_index, const int64_t* lengths,
                                     int64_t* after, int64_t* stacked) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < I; i += stride) {
        int64_t cur = indices[i];
        if (stacked) stacked[i] = cur;
        for (int64_t n = 1; n < N; ++n) {
            cur = next_one(cur, offset, E, done, last_index, lengths, nullptr);
            if (stacked) stacked[n * I + i] = cur;
        }
        after[i] = cur;
    }
}

constexpr int NSTEP_MAX = 32;

// One thread walks one index (rewards, end flags, value mask, gamma^n_eff, reward sum); the A columns of the block's SB
// indices are then written by all 128 threads together, consecutive threads on consecutive columns.  SB (<= 128, chosen by
// the host: fewer indices per block the more columns there are) keeps the column work spread over the chip: with one thread
// per index looping over its columns, QRDQN's 200 quantile columns were 200 strided stores per thread of four workgroups
// (55 us per update).
__global__ __launch_bounds__(128) void nstep_fused_kernel(const int64_t* indices, int64_t I, int64_t N,
                                   const int64_t* offset, int64_t E, const uint8_t* done,
                                   const uint8_t* terminated, const int64_t* last_index,
                                   const int64_t* lengths, const double* rew, const float* tq,
                                   int64_t A, double gamma, float* out, double* out64, int SB) {
    __shared__ float s_mask[128];
    __shared__ double s_gpow[128], s_mc[128];
    for (int64_t base = (int64_t)blockIdx.x * SB; base < I; base += (int64_t)gridDim.x * SB) {
        const int64_t i = base + threadIdx.x;
        if ((int)threadIdx.x < SB && i < I) {
            // forward walk: reward and end flag of each of the N stacked transitions.
            // indices[i] itself is used un-wrapped as in the reference (stacked_indices_NI[0] = indices).
            double r[NSTEP_MAX];
            bool e[NSTEP_MAX];
            int64_t cur = indices[i];
            for (int n = 0; n < (int)N; ++n) {
                bool is_end;
                const int64_t nxt = next_one(cur, offset, E, done, last_index, lengths, &is_end);
                r[n] = rew[cur];
                e[n] = is_end;
                if (n + 1 < (int)N) cur = nxt;
            }
            double mc = 0.0;
            int64_t gammas = N;
            for (int n = (int)N - 1; n >= 0; --n) {
                if (e[n]) {
                    gammas = n + 1;
                    mc = 0.0;
                }
                const double tmp = gamma * mc;
                mc = r[n] + tmp;
            }
            double gpow = 1.0;
            for (int64_t k = 0; k < gammas; ++k) gpow = gpow * gamma;
            s_mask[threadIdx.x] = terminated[cur] ? 0.f : 1.f;  // value_mask(idx_after_n), :798
            s_gpow[threadIdx.x] = gpow;
            s_mc[threadIdx.x] = mc;
        }
        __syncthreads();
        const int64_t rows = (I - base) < SB ? (I - base) : SB;
        for (int64_t el = threadIdx.x; el < rows * A; el += 128) {
            const int64_t sr = el / A;
            // target_q_IA *= mask happens in float32 in the reference (f32 array * bool)
            const float tqm = tq[base * A + el] * s_mask[sr];
            const double q = (double)tqm * s_gpow[sr];
            const double v = q + s_mc[sr];
            out[base * A + el] = (float)v;
            if (out64) out64[base * A + el] = v;
        }
        __syncthreads();
    }
}

// The part of nstep_fused_kernel that does not depend on target_q: per index the value mask, gamma^n_eff and the
// discounted reward sum, so that  returns = float(double(target_q * mask) * gpow + mc)  -- the same three operations on the
// same values -- can be finished by whoever produces target_q (ts_dqn_target_returns), and the walk itself can run ahead of
// the target network's passes.
__global__ void nstep_coef_kernel(const int64_t* indices, int64_t I, int64_t N, const int64_t* offset, int64_t E,
                                  const uint8_t* done, const uint8_t* terminated, const int64_t* last_index,
                                  const int64_t* lengths, const double* rew, double gamma, float* mask_out,
                                  double* gpow_out, double* mc_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < I; i += stride) {
        double r[NSTEP_MAX];
        bool e[NSTEP_MAX];
        int64_t cur = indices[i];
        for (int n = 0; n < (int)N; ++n) {
            bool is_end;
            const int64_t nxt = next_one(cur, offset, E, done, last_index, lengths, &is_end);
            r[n] = rew[cur];
            e[n] = is_end;
            if (n + 1 < (int)N) cur = nxt;
        }
        double mc = 0.0;
        int64_t gammas = N;
        for (int n = (int)N - 1; n >= 0; --n) {
            if (e[n]) {
                gammas = n + 1;
                mc = 0.0;
            }
            const double tmp = gamma * mc;
            mc = r[n] + tmp;
        }
        double gpow = 1.0;
        for (int64_t k = 0; k < gammas; ++k) gpow = gpow * gamma;
        mask_out[i] = terminated[cur] ? 0.f : 1.f;
        gpow_out[i] = gpow;
        mc_out[i] = mc;
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t g = ts::ceil_div(n, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int gae_sync_reserve(ts_workspace* ws, int64_t n_tiles, hipStream_t stream) {
    if (!ws) return ts::fail(TS_ERR_WORKSPACE, "workspace is NULL");
    if (ws->gae_sync && ws->gae_sync_tiles >= n_tiles && ws->gae_epoch < 0xfffffff0u) return TS_OK;
    TS_HIP_CHECK(hipSetDevice(ws->device));
    int64_t cap = ws->gae_sync_tiles > 0 ? ws->gae_sync_tiles : 1024;
    while (cap < n_tiles) cap *= 2;
    if (ws->gae_sync) {
        TS_HIP_CHECK(hipDeviceSynchronize());
        if (cap != ws->gae_sync_tiles) {
            TS_HIP_CHECK(hipFree(ws->gae_sync));
            ws->gae_sync = nullptr;
        }
    }
    const size_t bytes = sizeof(GaeSyncHeader) + sizeof(GaeTileMsg) * (size_t)cap;
    if (!ws->gae_sync) TS_HIP_CHECK(hipMalloc(&ws->gae_sync, bytes));
    TS_HIP_CHECK(hipMemsetAsync(ws->gae_sync, 0, bytes, stream));
    ws->gae_sync_tiles = cap;
    for (int k = 0; k < GAE_SHARDS; ++k) ws->gae_ticket_base[k] = 0;
    ws->gae_epoch = 0;
    return TS_OK;
}

template <typename RewT>
int launch_gae(ts_workspace* ws, const GaeArgs<RewT>& g_in, float* adv_out, float* ret_out,
               double* adv64, double* ret64, double* ret_partials, hipStream_t stream) {
    GaeArgs<RewT> g = g_in;
    const int64_t n_tiles = ts::ceil_div(g.n, GAE_TILE);
    // workspace: [cut bitmask]
    g.cutbits = nullptr;
    if (g.n_cut > GAE_CUT_SCAN_MAX) {
        const size_t words = (size_t)ts::ceil_div(g.n, 32);
        if (int rc = ts::ws_reserve(ws, 4 * words)) return rc;
        uint32_t* bits = reinterpret_cast<uint32_t*>(ws->base);
        TS_HIP_CHECK(hipMemsetAsync(bits, 0, 4 * words, stream));
        const unsigned blocks = (unsigned)std::min<int64_t>(ts::ceil_div(g.n_cut, 256), 1024);
        hipLaunchKernelGGL(gae_cutbits_kernel, dim3(blocks), dim3(256), 0, stream, g.cut_pos, g.n_cut, g.d_n_cut, g.n, bits);
        TS_LAUNCH_CHECK();
        g.cutbits = bits;
    }
    const bool vec = aligned16(g.v_s) && aligned16(g.v_n) && aligned16(g.rew) &&
                     (reinterpret_cast<uintptr_t>(g.term) & 7u) == 0 &&
                     (reinterpret_cast<uintptr_t>(g.trunc) & 7u) == 0 && aligned16(adv_out) &&
                     aligned16(ret_out);
    {
        int rc = gae_sync_reserve(ws, n_tiles, stream);
        if (rc != TS_OK) return rc;
        GaeSyncHeader* hdr = reinterpret_cast<GaeSyncHeader*>(ws->gae_sync);
        GaeTileMsg* msg = reinterpret_cast<GaeTileMsg*>(hdr + 1);
        const unsigned int epoch = ++ws->gae_epoch;
        GaeLaunch base;
        // <= 4 workgroups of 256 threads per CU are resident for certain (102 VGPRs -> 4 waves per
        // SIMD, 0.4 KB LDS): then no workgroup can wait for one that has not started.  Larger grids take tickets
        // (forward progress by construction).  TS_GAE_DIRECT_MAX (experiments): blockIdx order for larger grids too --
        // 5 % faster at 2^24, but correct only while every XCD dispatches its workgroups in index order.
        static const int64_t direct_max = getenv("TS_GAE_DIRECT_MAX") ? atoll(getenv("TS_GAE_DIRECT_MAX")) : 1024;
        base.direct = n_tiles <= direct_max ? 1 : 0;
        for (int k = 0; k < GAE_SHARDS; ++k) {
            base.base[k] = ws->gae_ticket_base[k];
            // blocks b with b % 8 == k claim from shard k: as many as sequence numbers = k (mod 8)
            if (!base.direct)
                ws->gae_ticket_base[k] += (unsigned long long)((n_tiles - k + GAE_SHARDS - 1) / GAE_SHARDS);
        }
        ts::ProfScope prof(ws, TS_KIND_GAE_APPLY, stream);
        if (vec)
            hipLaunchKernelGGL((gae_single_pass<RewT, true>), dim3((unsigned)n_tiles), dim3(GAE_THREADS), 0,
                               stream, g, hdr, msg, n_tiles, base, epoch, adv_out, ret_out, adv64, ret64,
                               ret_partials);
        else
            hipLaunchKernelGGL((gae_single_pass<RewT, false>), dim3((unsigned)n_tiles), dim3(GAE_THREADS), 0,
                               stream, g, hdr, msg, n_tiles, base, epoch, adv_out, ret_out, adv64, ret64,
                               ret_partials);
        TS_LAUNCH_CHECK();
        return TS_OK;
    }
}

}  // namespace

extern "C" {

int64_t ts_gae_num_tiles(int64_t n) { return n <= 0 ? 0 : ts::ceil_div(n, GAE_TILE); }

int ts_gae_scan(ts_workspace* ws, const float* v_s, const float* v_s_next, const void* rew,
                int rew_dtype, const uint8_t* terminated, const uint8_t* truncated,
                const int64_t* cut_pos, int64_t n_cut, const int64_t* d_n_cut, int64_t n,
                double gamma, double gae_lambda, double v_scale, double ret_div, float* adv_out,
                float* returns_out, double* adv64, double* ret64, double* ret_partials,
                ts_stream_t stream) {
    TS_REQUIRE(n >= 0 && n_cut >= 0, TS_ERR_INVALID_ARG, "ts_gae_scan: negative size");
    if (n == 0) return TS_OK;
    TS_REQUIRE(v_s && v_s_next && rew && terminated && truncated && adv_out && returns_out,
               TS_ERR_INVALID_ARG, "ts_gae_scan: NULL array argument");
    TS_REQUIRE(n_cut == 0 || cut_pos, TS_ERR_INVALID_ARG, "ts_gae_scan: cut_pos is NULL");
    TS_REQUIRE(rew_dtype == 0 || rew_dtype == 1, TS_ERR_INVALID_ARG,
               "ts_gae_scan: rew_dtype must be 0 (f32) or 1 (f64)");
    TS_REQUIRE(ret_div != 0.0, TS_ERR_INVALID_ARG, "ts_gae_scan: ret_div must be non-zero");
    hipStream_t s = ts::as_stream(stream);
    if (rew_dtype == 0) {
        GaeArgs<float> g{v_s, v_s_next, (const float*)rew, terminated, truncated, cut_pos, d_n_cut,
                         n_cut, n, gamma, gamma * gae_lambda, v_scale, ret_div};
        return launch_gae(ws, g, adv_out, returns_out, adv64, ret64, ret_partials, s);
    }
    GaeArgs<double> g{v_s, v_s_next, (const double*)rew, terminated, truncated, cut_pos, d_n_cut,
                      n_cut, n, gamma, gamma * gae_lambda, v_scale, ret_div};
    return launch_gae(ws, g, adv_out, returns_out, adv64, ret64, ret_partials, s);
}

int ts_gae_check(ts_workspace* ws, int* error_out, ts_stream_t stream) {
    TS_REQUIRE(ws && error_out, TS_ERR_INVALID_ARG, "ts_gae_check: NULL argument");
    *error_out = 0;
    if (!ws->gae_sync) return TS_OK;
    GaeSyncHeader h;
    hipStream_t s = ts::as_stream(stream);
    TS_HIP_CHECK(hipMemcpyAsync(&h, ws->gae_sync, sizeof(h), hipMemcpyDeviceToHost, s));
    TS_HIP_CHECK(hipStreamSynchronize(s));
    *error_out = (int)h.error;
    return TS_OK;
}

int ts_isin_positions(const int64_t* indices, int64_t n, const int64_t* unfinished,
                      int64_t n_unfinished, int64_t* cut_pos_out, int64_t capacity,
                      int64_t* n_cut_out, ts_stream_t stream) {
    TS_REQUIRE(n >= 0 && n_unfinished >= 0 && capacity >= 0, TS_ERR_INVALID_ARG,
               "ts_isin_positions: negative size");
    TS_REQUIRE(n_cut_out, TS_ERR_INVALID_ARG, "ts_isin_positions: n_cut_out is NULL");
    hipStream_t s = ts::as_stream(stream);
    TS_HIP_CHECK(hipMemsetAsync(n_cut_out, 0, sizeof(int64_t), s));
    if (n == 0 || n_unfinished == 0) return TS_OK;
    TS_REQUIRE(indices && unfinished && cut_pos_out, TS_ERR_INVALID_ARG,
               "ts_isin_positions: NULL array argument");
    hipLaunchKernelGGL(isin_positions_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, indices, n,
                       unfinished, n_unfinished, cut_pos_out, capacity,
                       reinterpret_cast<unsigned long long*>(n_cut_out));
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_nstep_return(const double* rew_B, const uint8_t* end_flag_B, const float* target_q_IA,
                    const int64_t* stacked_indices_NI, int64_t I, int64_t A, int64_t n_step,
                    int64_t B, double gamma, float* out, double* out64, ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && A >= 0 && B >= 0, TS_ERR_INVALID_ARG, "ts_nstep_return: negative size");
    TS_REQUIRE(n_step >= 1, TS_ERR_INVALID_ARG, "ts_nstep_return: n_step must be >= 1");
    if (I * A == 0) return TS_OK;
    TS_REQUIRE(rew_B && end_flag_B && target_q_IA && stacked_indices_NI && out, TS_ERR_INVALID_ARG,
               "ts_nstep_return: NULL array argument");
    hipLaunchKernelGGL(nstep_return_kernel, dim3(grid_for(I * A, 256)), dim3(256), 0,
                       ts::as_stream(stream), rew_B, end_flag_B, target_q_IA, stacked_indices_NI, I,
                       A, n_step, gamma, out, out64);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_nstep_indices(const int64_t* indices, int64_t I, int64_t n_step, const int64_t* offset,
                     int64_t E, const uint8_t* done, const int64_t* last_index,
                     const int64_t* lengths, int64_t* after_out, int64_t* stacked_out,
                     ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && E >= 1, TS_ERR_INVALID_ARG, "ts_nstep_indices: bad size");
    TS_REQUIRE(n_step >= 1, TS_ERR_INVALID_ARG, "ts_nstep_indices: n_step must be >= 1");
    if (I == 0) return TS_OK;
    TS_REQUIRE(indices && offset && done && last_index && lengths && after_out, TS_ERR_INVALID_ARG,
               "ts_nstep_indices: NULL array argument");
    hipLaunchKernelGGL(nstep_indices_kernel, dim3(grid_for(I, 256)), dim3(256), 0,
                       ts::as_stream(stream), indices, I, n_step, offset, E, done, last_index,
                       lengths, after_out, stacked_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_nstep_return_fused(const int64_t* indices, int64_t I, int64_t n_step,
                          const int64_t* offset, int64_t E, const uint8_t* done,
                          const uint8_t* terminated, const int64_t* last_index,
                          const int64_t* lengths, const double* rew_B, const float* target_q_IA,
                          int64_t A, double gamma, float* out, double* out64,
                          ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && A >= 0 && E >= 1, TS_ERR_INVALID_ARG, "ts_nstep_return_fused: bad size");
    TS_REQUIRE(n_step >= 1, TS_ERR_INVALID_ARG, "ts_nstep_return_fused: n_step must be >= 1");
    TS_REQUIRE(n_step <= NSTEP_MAX, TS_ERR_UNSUPPORTED,
               "ts_nstep_return_fused: n_step %lld > %d, use ts_nstep_indices + ts_nstep_return",
               (long long)n_step, NSTEP_MAX);
    if (I * A == 0) return TS_OK;
    TS_REQUIRE(indices && offset && done && terminated && last_index && lengths && rew_B &&
                   target_q_IA && out,
               TS_ERR_INVALID_ARG, "ts_nstep_return_fused: NULL array argument");
    const int sb = A >= 64 ? 4 : (A >= 8 ? 16 : 128);          // indices per workgroup: ~1,000 output elements each
    hipLaunchKernelGGL(nstep_fused_kernel, dim3(grid_for(I, sb)), dim3(128), 0,
                       ts::as_stream(stream), indices, I, n_step, offset, E, done, terminated,
                       last_index, lengths, rew_B, target_q_IA, A, gamma, out, out64, sb);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int ts_nstep_coefficients(const int64_t* indices, int64_t I, int64_t n_step, const int64_t* offset, int64_t E,
                          const uint8_t* done_B, const uint8_t* terminated_B, const int64_t* last_index,
                          const int64_t* lengths, const double* rew_B, double gamma, float* mask_out, double* gpow_out,
                          double* mc_out, ts_stream_t stream) {
    TS_REQUIRE(I >= 0 && E >= 1, TS_ERR_INVALID_ARG, "ts_nstep_coefficients: bad size");
    TS_REQUIRE(n_step >= 1 && n_step <= NSTEP_MAX, TS_ERR_INVALID_ARG, "ts_nstep_coefficients: 1 <= n_step <= %d", NSTEP_MAX);
    if (I == 0) return TS_OK;
    TS_REQUIRE(indices && offset && done_B && terminated_B && last_index && lengths && rew_B && mask_out && gpow_out && mc_out,
               TS_ERR_INVALID_ARG, "ts_nstep_coefficients: NULL array argument");
    hipLaunchKernelGGL(nstep_coef_kernel, dim3(grid_for(I, 128)), dim3(128), 0, ts::as_stream(stream), indices, I, n_step,
                       offset, E, done_B, terminated_B, last_index, lengths, rew_B, gamma, mask_out, gpow_out, mc_out);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

}  // extern "C"
