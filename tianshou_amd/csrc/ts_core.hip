// ts_core.hip -- error reporting, version, workspace management.
#include <cstdlib>
#include <cstring>

#include "ts_common.h"

namespace ts {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int ws_reserve(ts_workspace* ws, size_t bytes) {
    if (!ws) return fail(TS_ERR_WORKSPACE, "workspace is NULL");
    if (bytes <= ws->bytes) return TS_OK;
    if (ws->max_bytes && bytes > ws->max_bytes)
        return fail(TS_ERR_WORKSPACE, "workspace needs %zu bytes, limit is %zu", bytes, ws->max_bytes);
    size_t want = ws->bytes ? ws->bytes : (size_t)1 << 20;
    while (want < bytes) want *= 2;
    if (ws->max_bytes && want > ws->max_bytes) want = ws->max_bytes;
    TS_HIP_CHECK(hipSetDevice(ws->device));
    if (ws->base) {
        // kernels of earlier calls may still read the old block
        TS_HIP_CHECK(hipDeviceSynchronize());
        TS_HIP_CHECK(hipFree(ws->base));
        ws->base = nullptr;
        ws->bytes = 0;
    }
    TS_HIP_CHECK(hipMalloc(&ws->base, want));
    ws->bytes = want;
    return TS_OK;
}

int ws_winner(ts_workspace* ws, int64_t bound, hipStream_t stream, int32_t** out) {
    if (!ws) return fail(TS_ERR_WORKSPACE, "workspace is NULL");
    if (ws->winner_len < bound) {
        TS_HIP_CHECK(hipSetDevice(ws->device));
        if (ws->winner) {
            TS_HIP_CHECK(hipDeviceSynchronize());
            TS_HIP_CHECK(hipFree(ws->winner));
            ws->winner = nullptr;
            ws->winner_len = 0;
        }
        TS_HIP_CHECK(hipMalloc((void**)&ws->winner, sizeof(int32_t) * (size_t)bound));
        TS_HIP_CHECK(hipMemsetAsync(ws->winner, 0xFF, sizeof(int32_t) * (size_t)bound, stream));
        ws->winner_len = bound;
    }
    *out = ws->winner;
    return TS_OK;
}

static int side_create(ts_workspace* ws) {
    if (!ws->side_ready) {
        TS_HIP_CHECK(hipSetDevice(ws->device));
        // TS_SIDE_PRIORITY=low|high (experiments): the workspace's side streams below / above the caller's stream in the
        // hardware scheduler's order (profiles/r06_side_priority_ab.txt)
        const char* pe = getenv("TS_SIDE_PRIORITY");
        if (pe && (pe[0] == 'l' || pe[0] == 'h')) {
            int least = 0, greatest = 0;
            TS_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int prio = pe[0] == 'l' ? least : greatest;
            TS_HIP_CHECK(hipStreamCreateWithPriority(&ws->side, hipStreamNonBlocking, prio));
            TS_HIP_CHECK(hipStreamCreateWithPriority(&ws->side2, hipStreamNonBlocking, prio));
        } else {
            TS_HIP_CHECK(hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
            TS_HIP_CHECK(hipStreamCreateWithFlags(&ws->side2, hipStreamNonBlocking));
        }
        for (int i = 0; i < 16; ++i) TS_HIP_CHECK(hipEventCreateWithFlags(&ws->side_ev[i], hipEventDisableTiming));
        ws->side_ready = 1;
    }
    return TS_OK;
}

int side_stream(ts_workspace* ws, hipStream_t main, hipStream_t* out) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "side_stream: workspace is NULL");
    static const bool single = getenv("TS_NO_SIDE_STREAM") != nullptr;      // experiments: everything on one stream
    if (ws->profiling || single) { *out = main; return TS_OK; }      // serial launches: per-kernel event pairs do not overlap
    if (int rc = side_create(ws)) return rc;
    *out = ws->side;
    return TS_OK;
}

int side_streams(ts_workspace* ws, hipStream_t main, hipStream_t* a, hipStream_t* b) {
    if (int rc = side_stream(ws, main, a)) return rc;
    *b = *a == main ? main : ws->side2;
    return TS_OK;
}

int stream_wait(ts_workspace* ws, hipStream_t from, hipStream_t to, int slot) {
    if (from == to) return TS_OK;
    TS_REQUIRE(ws && ws->side_ready && slot >= 0 && slot < 16, TS_ERR_WORKSPACE, "stream_wait: bad workspace / slot");
    TS_HIP_CHECK(hipEventRecord(ws->side_ev[slot], from));
    TS_HIP_CHECK(hipStreamWaitEvent(to, ws->side_ev[slot], 0));
    return TS_OK;
}

int record_td(ts_workspace* ws, hipStream_t s) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "record_td: workspace is NULL");
    if (!ws->td_ev_ready) {
        TS_HIP_CHECK(hipSetDevice(ws->device));
        TS_HIP_CHECK(hipEventCreateWithFlags(&ws->td_ev, hipEventDisableTiming));
        ws->td_ev_ready = 1;
    }
    TS_HIP_CHECK(hipEventRecord(ws->td_ev, s));
    return TS_OK;
}

ProfScope::ProfScope(ts_workspace* w, int kind, hipStream_t s) : ws(w), stream(s), slot(-1) {
    if (!ws || !ws->profiling || ws->ev_n >= ws->ev_cap) return;
    slot = ws->ev_n++;
    ws->ev_kind[slot] = kind;
    (void)hipEventRecord(ws->ev[2 * slot], stream);
}

ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(ws->ev[2 * slot + 1], stream);
}

}  // namespace ts

extern "C" {

int ts_profile_begin(ts_workspace* ws) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_profile_begin: workspace is NULL");
    if (!ws->ev) {
        const int cap = 4096;
        ws->ev = (hipEvent_t*)calloc(2 * (size_t)cap, sizeof(hipEvent_t));
        ws->ev_kind = (int*)calloc((size_t)cap, sizeof(int));
        TS_REQUIRE(ws->ev && ws->ev_kind, TS_ERR_WORKSPACE, "ts_profile_begin: host allocation failed");
        TS_HIP_CHECK(hipSetDevice(ws->device));
        for (int i = 0; i < 2 * cap; ++i) TS_HIP_CHECK(hipEventCreate(&ws->ev[i]));
        ws->ev_cap = cap;
    }
    ws->ev_n = 0;
    ws->profiling = 1;
    return TS_OK;
}

int ts_profile_end(ts_workspace* ws, double* ms_by_kind, int64_t* count_by_kind, int n_kinds) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_profile_end: workspace is NULL");
    TS_REQUIRE(ms_by_kind && count_by_kind && n_kinds >= 1, TS_ERR_INVALID_ARG,
               "ts_profile_end: bad output arguments");
    ws->profiling = 0;
    for (int k = 0; k < n_kinds; ++k) { ms_by_kind[k] = 0.0; count_by_kind[k] = 0; }
    for (int i = 0; i < ws->ev_n; ++i) {
        TS_HIP_CHECK(hipEventSynchronize(ws->ev[2 * i + 1]));
        float ms = 0.f;
        TS_HIP_CHECK(hipEventElapsedTime(&ms, ws->ev[2 * i], ws->ev[2 * i + 1]));
        const int k = ws->ev_kind[i];
        if (k >= 0 && k < n_kinds) { ms_by_kind[k] += (double)ms; count_by_kind[k] += 1; }
    }
    ws->ev_n = 0;
    return TS_OK;
}

const char* ts_version(void) { return "tsengine 0.1.0 (gfx950)"; }

const char* ts_last_error(void) { return ts::err_buf(); }

int ts_workspace_create(ts_workspace** out, int device, size_t max_bytes) {
    TS_REQUIRE(out != nullptr, TS_ERR_INVALID_ARG, "ts_workspace_create: out is NULL");
    ts_workspace* ws = (ts_workspace*)calloc(1, sizeof(ts_workspace));
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_workspace_create: host allocation failed");
    ws->device = device;
    ws->max_bytes = max_bytes;
    *out = ws;
    return TS_OK;
}

int ts_workspace_side_stream(ts_workspace* ws, int which, ts_stream_t* stream_out) {
    TS_REQUIRE(ws != nullptr && stream_out != nullptr, TS_ERR_WORKSPACE, "ts_workspace_side_stream: NULL argument");
    TS_REQUIRE(which == 0 || which == 1, TS_ERR_INVALID_ARG, "ts_workspace_side_stream: which = 0 or 1");
    if (int rc = ts::side_create(ws)) return rc;
    *stream_out = reinterpret_cast<ts_stream_t>(which == 0 ? ws->side : ws->side2);
    return TS_OK;
}

int ts_mlp_set_hidden(ts_workspace* ws, int64_t hidden) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_mlp_set_hidden: workspace is NULL");
    // (the SAC / TD3 / DDPG / REDQ entry points take widths up to 1024 and check that themselves; DiscreteSAC's, which
    // receive the width as an argument and read only the depth from the workspace, up to 2048)
    TS_REQUIRE(hidden == 0 || (hidden >= 32 && hidden <= 2048 && hidden % 32 == 0), TS_ERR_INVALID_ARG,
               "ts_mlp_set_hidden: 0 (default 256) or a multiple of 32 in [32, 2048], got %lld", (long long)hidden);
    ws->mlp_hidden = (int)hidden;
    ws->mlp_depth = 0;
    ws->mlp_act_tanh = 0;
    return TS_OK;
}

int ts_mlp_set_activation(ts_workspace* ws, int activation) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_mlp_set_activation: workspace is NULL");
    TS_REQUIRE(activation == TS_NET_ACT_RELU || activation == TS_NET_ACT_TANH, TS_ERR_UNSUPPORTED,
               "ts_mlp_set_activation: TS_NET_ACT_RELU or TS_NET_ACT_TANH");
    ws->mlp_act_tanh = activation == TS_NET_ACT_TANH ? 1 : 0;
    return TS_OK;
}

int ts_sac_set_actor_bound(ts_workspace* ws, double max_action) {
    TS_REQUIRE(ws != nullptr, TS_ERR_WORKSPACE, "ts_sac_set_actor_bound: workspace is NULL");
    TS_REQUIRE(max_action >= 0.0 && max_action < 1e30, TS_ERR_INVALID_ARG, "ts_sac_set_actor_bound: max_action >= 0 (0 = unbounded)");
    ws->sac_actor_bound = (float)max_action;
    return TS_OK;
}

int ts_mlp_set_trunk(ts_workspace* ws, int64_t hidden, int64_t depth) {
    TS_REQUIRE(depth == 0 || (depth >= 1 && depth <= TS_MLP_MAX_HIDDEN_LAYERS), TS_ERR_INVALID_ARG,
               "ts_mlp_set_trunk: 0 (default 2) or 1 .. %d hidden layers, got %lld", TS_MLP_MAX_HIDDEN_LAYERS, (long long)depth);
    if (int rc = ts_mlp_set_hidden(ws, hidden)) return rc;
    ws->mlp_depth = (int)depth;
    return TS_OK;
}

int ts_workspace_destroy(ts_workspace* ws) {
    if (!ws) return TS_OK;
    if (ws->learn_graphs && ws->learn_graphs_free) { (void)hipSetDevice(ws->device); ws->learn_graphs_free(ws->learn_graphs); }
    if (ws->side_ready) {
        (void)hipSetDevice(ws->device);
        (void)hipStreamSynchronize(ws->side);
        for (int i = 0; i < 16; ++i) (void)hipEventDestroy(ws->side_ev[i]);
        (void)hipStreamSynchronize(ws->side2);
        (void)hipStreamDestroy(ws->side);
        (void)hipStreamDestroy(ws->side2);
    }
    if (ws->td_ev_ready) { (void)hipSetDevice(ws->device); (void)hipEventDestroy(ws->td_ev); }
    if (ws->ppo_image) { (void)hipSetDevice(ws->device); (void)hipDeviceSynchronize(); (void)hipFree(ws->ppo_image); }
    for (int k = 0; k < 2; ++k)
        if (ws->conv_scratch[k]) { (void)hipSetDevice(ws->device); (void)hipDeviceSynchronize(); (void)hipFree(ws->conv_scratch[k]); }
    if (ws->dg_tables) { (void)hipSetDevice(ws->device); (void)hipDeviceSynchronize(); (void)hipFree(ws->dg_tables); }
    if (ws->base || ws->winner || ws->ev || ws->gae_sync) {
        (void)hipSetDevice(ws->device);
        (void)hipDeviceSynchronize();
        if (ws->gae_sync) (void)hipFree(ws->gae_sync);
        if (ws->base) (void)hipFree(ws->base);
        if (ws->winner) (void)hipFree(ws->winner);
        if (ws->ev) {
            for (int i = 0; i < 2 * ws->ev_cap; ++i) (void)hipEventDestroy(ws->ev[i]);
            free(ws->ev);
            free(ws->ev_kind);
        }
    }
    free(ws);
    return TS_OK;
}

}  // extern "C"
