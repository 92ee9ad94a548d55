// ts_common.h -- shared host-side helpers of libtsengine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/tsengine.h"

namespace ts {

// thread-local error string returned by ts_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define TS_HIP_CHECK(expr)                                                                 \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ts::fail(TS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                           \
    } while (0)

#define TS_LAUNCH_CHECK()                                                                  \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess)                                                              \
            return ts::fail(TS_ERR_HIP, "kernel launch failed: %s (%s:%d)",               \
                            hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

#define TS_REQUIRE(cond, code, ...)                        \
    do {                                                   \
        if (!(cond)) return ts::fail((code), __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(ts_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ts

// Opaque workspace: one growable device allocation carved by the entry points.
struct ts_workspace {
    int device;
    size_t max_bytes;
    void* base;
    size_t bytes;
    // persistent sum-tree winner table (int32[bound], kept at -1 between calls)
    int32_t* winner;
    int64_t winner_len;
    // single-pass GAE scan: persistent cross-workgroup hand-off state (never reset between launches:
    // tags are launch epochs, tile tickets are offset by the running base)
    void* gae_sync;              // [ticket u64][error u32][pad] + per tile {a bits, b bits, flag}
    int64_t gae_sync_tiles;      // capacity in tiles
    unsigned long long gae_ticket_base[8];
    unsigned int gae_epoch;
    // optional per-kernel timing with HIP events on the launch stream (ts_profile_begin/end)
    int profiling;
    hipEvent_t* ev;      // [2 * ev_cap] start/stop pairs
    int* ev_kind;        // [ev_cap]
    int ev_cap, ev_n;
    // second stream for independent kernels of one entry point (e.g. weight- and input-gradient GEMMs of a layer):
    // forked from / joined into the caller's stream with events, created on first use
    // PPO data-parallel path: persistent LDS images + param -> slot table (see ts_ppo.hip dp_image)
    void* ppo_image;
    size_t ppo_image_bytes;
    const float* ppo_image_params;
    int ppo_image_key;
    // transposed weight matrices of linear input-gradient passes (ts_conv2.hip), grown on demand.  One buffer per
    // launch stream of the workspace ([0] the caller's stream, [1] ws->side): the twin critics of the SAC family run
    // their backward chains concurrently on the two streams with the same workspace.
    void* conv_scratch[2];
    size_t conv_scratch_bytes[2];
    // input-gradient class tables of ts_conv2.hip: 16 device-resident slots, keyed by geometry on the host; slots
    // 0..7 belong to the caller's stream, 8..15 to ws->side (a slot is only ever rewritten in stream order)
    void* dg_tables;
    long long dg_key[16][16];
    int dg_next[2];
    // hidden width of the Net[h, h] MLPs of the SAC / TD3 / DDPG / REDQ entry points called with this workspace
    // (ts_mlp_set_hidden; 0 = 256, the width of examples/mujoco/mujoco_sac.py)
    int mlp_hidden;
    // number of hidden layers of those MLPs (ts_mlp_set_trunk; 0 = 2, the depth of the examples' nets)
    int mlp_depth;
    // 1: nn.Tanh after every hidden layer of those MLPs instead of Net's default nn.ReLU (ts_mlp_set_activation)
    int mlp_act_tanh;
    // max_action of a BOUNDED Gaussian actor (mu = max_action * tanh(mu), continuous.py:230-231) of the SAC / REDQ entry
    // points (ts_sac_set_actor_bound; 0 = unbounded, the actors of the examples)
    float sac_actor_bound;
    hipStream_t side;
    hipStream_t side2;           // second side stream (ts::side_streams): created together with `side`
    hipEvent_t side_ev[16];
    int side_ready;
    // recorded by every ts_dqn_update* / ts_distq_update / ts_rainbow_update call right after its TD-error (priority) kernel
    // (ts::record_td, waited for by ts_dqn_wait_td): the priority update and the next batch's sampling need nothing else
    // from the update and can run beside its backward pass
    hipEvent_t td_ev;
    int td_ev_ready;
    // captured update graphs of ts_dqn_learn_step (host state owned by ts_dqn.hip; released through the hook)
    void* learn_graphs;
    void (*learn_graphs_free)(void*);
};

namespace ts {
// Ensures ws->base holds at least `bytes`; (re)allocation synchronises the device once.
int ws_reserve(ts_workspace* ws, size_t bytes);
int ws_winner(ts_workspace* ws, int64_t bound, hipStream_t stream, int32_t** out);

// Side stream of a workspace (created on first use).  Ordering is expressed with event slots:
//   ts::stream_wait(ws, from, to, slot): everything enqueued on `to` after this call runs after everything
//   enqueued on `from` before it.
int side_stream(ts_workspace* ws, hipStream_t main, hipStream_t* out);   // == main while profiling (clean per-kernel times)
int stream_wait(ts_workspace* ws, hipStream_t from, hipStream_t to, int slot);
int side_streams(ts_workspace* ws, hipStream_t main, hipStream_t* a, hipStream_t* b);     // both side streams
int record_td(ts_workspace* ws, hipStream_t s);       // the new priorities / the loss of an update are written on `s`
// Adam with its two per-step scalars {step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t)} (adam_step_scalars: the float32 values
// adam_step passes by value) read from device memory: a captured launch replayed with a new step number (ts_dqn_learn_step)
void adam_step_scalars(int64_t step, double lr, double beta1, double beta2, float out[2]);
int adam_step_dev(hipStream_t s, float* params, float* m, float* v, const float* grad, int64_t n, const float* step_dev,
                  double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);
// ts_uniform_fill_f64 with the Philox counter read from device memory when counter_dev != NULL (same use)
int uniform_fill_f64(double* out, int64_t n, uint64_t seed, uint64_t counter, const uint64_t* counter_dev, hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE function attribute: a process-wide "done" flag would skip it on
// the second GPU a process drives (tests iterating devices, threaded data parallelism).  One of these per kernel (a function-
// local static): remembers, per device, the largest size already granted; racing host threads at worst set it twice.
struct DynLds {
    std::atomic<int> granted[16] = {};
    int allow(const void* fn, size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
        std::atomic<int>& g = granted[dev & 15];
        if ((int)bytes <= g.load(std::memory_order_acquire)) return TS_OK;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(TS_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %zu) failed: %s", bytes, hipGetErrorString(e));
        int cur = g.load(std::memory_order_relaxed);
        while (cur < (int)bytes && !g.compare_exchange_weak(cur, (int)bytes, std::memory_order_release)) {}
        return TS_OK;
    }
};

// Optimizer.step over one flat parameter vector (ts_optim.hip): torch.optim.Adam (optim.py:89-110) or torch.optim.RMSprop
// (optim.py:113-140), both with the optional L2 term `grad += weight_decay * param` that torch applies inside step(),
// after clip_grad_norm_.  State vectors: Adam exp_avg -> m, exp_avg_sq -> v; RMSprop square_avg -> v and the momentum
// buffer (momentum > 0) or grad_avg (centered) -> m.
struct OptimDesc {
    int kind = TS_OPT_ADAM;
    int centered = 0;
    double weight_decay = 0.0, alpha = 0.99, momentum = 0.0;
};
inline OptimDesc optim_from(const ts_ppo_hparams* hp) {
    OptimDesc o;
    o.kind = hp->optimizer; o.centered = hp->rms_centered;
    o.weight_decay = hp->weight_decay; o.alpha = hp->rms_alpha; o.momentum = hp->rms_momentum;
    return o;
}
int optim_step(hipStream_t s, const OptimDesc& o, float* params, float* m, float* v, const float* grad, int64_t n,
               int64_t step, double lr, double beta1, double beta2, double eps, double max_grad_norm, float* norm_scratch);

// Counter-based normal noise (ts_normal_fill; rsample()'s eps): Philox-4x32-10 keyed by `seed`, counter = (element quad q,
// stream offset), two Box-Muller pairs per counter -> z[0..3] = elements 4q .. 4q + 3 of the stream.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void normal4(int64_t q, uint64_t seed, uint64_t offset, float (&z)[4]) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // u1 in (0, 1], u2 in [0, 1): 24 random bits each
        const float u1 = ((float)(c[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = (float)(c[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        z[2 * p] = rad * cs;
        z[2 * p + 1] = rad * sn;
    }
}

// Brackets one kernel launch with a start/stop event pair when profiling is enabled.
struct ProfScope {
    ts_workspace* ws;
    hipStream_t stream;
    int slot;
    ProfScope(ts_workspace* w, int kind, hipStream_t s);
    ~ProfScope();
};
}  // namespace ts
