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

}  // namespace ts

extern "C" {

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

int ts_workspace_destroy(ts_workspace* ws) {
    if (!ws) return TS_OK;
    if (ws->base || ws->winner) {
        (void)hipSetDevice(ws->device);
        (void)hipDeviceSynchronize();
        if (ws->base) (void)hipFree(ws->base);
        if (ws->winner) (void)hipFree(ws->winner);
    }
    free(ws);
    return TS_OK;
}

}  // extern "C"
