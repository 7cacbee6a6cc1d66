// csrc/mvo_api_ba.cpp -- the bundle-adjustment entry points of include/mvo_hip.h (argument checks + dispatch into
// ba_host.cpp).  A translation unit of its own so that tests/sim can compile exactly this file, ba_host.cpp and
// ba_kernels.hip against the CPU emulation of the HIP runtime (tests/sim/hip_emu) -- the product library is built from
// the same sources with hipcc.
#include <cstring>

#include "mvo_internal.h"

int ba_demand_replay(const double* times, int n, unsigned char* decisions);  // ba_host.cpp
extern int g_ba_use_mfma, g_ba_wgs, g_ba_same_l2, g_ba_profile, g_ba_block_solver, g_ba_cu_share, g_ba_xcd_reserve, g_ba_edge_rows, g_ba_service, g_ba_chunk_pieces, g_ba_uv_global;  // ba_host.cpp
// mvo_debug_set("ba_*", v): validation paths and planner overrides the tests compare
int ba_debug_set(const char* key, int value) {
    if (!std::strcmp(key, "ba_mfma")) g_ba_use_mfma = value;
    else if (!std::strcmp(key, "ba_profile")) g_ba_profile = value;
    else if (!std::strcmp(key, "ba_same_l2")) g_ba_same_l2 = value;
    else if (!std::strcmp(key, "ba_wgs")) g_ba_wgs = value;
    else if (!std::strcmp(key, "ba_block_solver")) g_ba_block_solver = value;
    else if (!std::strcmp(key, "ba_cu_share")) g_ba_cu_share = value;
    else if (!std::strcmp(key, "ba_xcd_reserve")) g_ba_xcd_reserve = value;
    else if (!std::strcmp(key, "ba_edge_rows")) g_ba_edge_rows = value;
    else if (!std::strcmp(key, "ba_service")) g_ba_service = value;
    else if (!std::strcmp(key, "ba_chunk_pieces")) g_ba_chunk_pieces = value;
    else if (!std::strcmp(key, "ba_uv_global")) g_ba_uv_global = value;
    else return MVO_ERR_INVALID;
    return MVO_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------------------- BA
static int ba_check(mvo_ctx* ctx, const mvo_ba_problem* p) {
    if (!ctx || !p || p->n_poses < 0 || p->n_points < 0 || p->n_edges < 0)
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if ((p->n_poses && !p->pose_T_w_c) || (p->n_points && !p->points) ||
        (p->n_edges && (!p->edge_pose || !p->edge_point || !p->edge_uv)))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "null array", hipSuccess);
    for (int e = 0; e < p->n_edges; ++e)
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 ||
            p->edge_point[e] >= p->n_points)
            return mvo_set_err(ctx, MVO_ERR_INVALID, "edge index out of range", hipSuccess);
    return MVO_OK;
}
int mvo_ba_set_mode(mvo_ctx* ctx, int mode) {
    if (!ctx || (mode != MVO_BA_MODE_LATENCY && mode != MVO_BA_MODE_THROUGHPUT && mode != MVO_BA_MODE_SHARED))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad BA mode", hipSuccess);
    ctx->ba_throughput_mode = mode != MVO_BA_MODE_LATENCY;
    ctx->ba_never_resident = mode == MVO_BA_MODE_SHARED;
    return MVO_OK;
}
int mvo_bundle_adjustment(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st) {
    int r = ba_check(ctx, p);
    if (r) return r;
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_solve_device(ctx, p, st);
}
int mvo_bundle_adjustment_begin(mvo_ctx* ctx, const mvo_ba_problem* p) {
    int r = ba_check(ctx, p);
    if (r) return r;
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_begin_device(ctx, p);
}
int mvo_bundle_adjustment_end(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st) {
    if (!ctx) return MVO_ERR_INVALID;
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_end_device(ctx, p, st);
}
int mvo_ba_solve_batch(mvo_ctx* ctx, mvo_ba_problem* problems, int n, mvo_ba_stats* stats) {
    if (!ctx || n < 0 || (n && !problems)) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    for (int i = 0; i < n; ++i) {
        int r = ba_check(ctx, &problems[i]);
        if (r) return r;
    }
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_solve_batch_device(ctx, problems, n, stats);
}

int mvo_ba_prepare(mvo_ctx* ctx, const mvo_ba_problem* p, mvo_ba_handle** handle) {
    if (!handle) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    int r = ba_check(ctx, p);
    if (r) return r;
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_prepare_device(ctx, p, handle);
}
int mvo_ba_solve_resident(mvo_ctx* ctx, mvo_ba_handle* handle) {
    if (!ctx || !handle) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_run_device(ctx, handle);
}
int mvo_ba_fetch(mvo_ctx* ctx, mvo_ba_handle* handle, double* poses, double* points, mvo_ba_stats* stats) {
    if (!ctx || !handle) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    return ba_fetch_device(ctx, handle, poses, points, stats);
}
void mvo_ba_release(mvo_ctx* ctx, mvo_ba_handle* handle) {
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    ba_release_device(ctx, handle);
}
int mvo_debug_ba_service_times(int device, double* out5) {
    if (!out5) return MVO_ERR_INVALID;
    ba_service_times(device, out5);
    return MVO_OK;
}

int mvo_debug_ba_resident_stats(int device, long long* windows, long long* grid_starts) {
    ba_resident_stats(device, windows, grid_starts);
    return MVO_OK;
}

int mvo_debug_ba_path_switches(int device, long long* n) {
    if (!n) return MVO_ERR_INVALID;
    ba_resident_stats(device, nullptr, nullptr, nullptr, n);
    return MVO_OK;
}

int mvo_debug_ba_resident_cycles(int device, double* shader_cycles) {
    if (!shader_cycles) return MVO_ERR_INVALID;
    ba_resident_stats(device, nullptr, nullptr, shader_cycles);
    return MVO_OK;
}

int mvo_ba_launch_stats(int device, long long* launches, long long* windows, double* ms, int reset) {
    if (device < 0 || device > 15) return MVO_ERR_INVALID;
    ba_launch_stats(device, launches, windows, ms, reset);
    return MVO_OK;
}
int mvo_debug_ba_trace_enable(mvo_ctx* ctx, int on) {
    if (!ctx) return MVO_ERR_INVALID;
    ba_set_trace(ctx, on);
    return MVO_OK;
}
int mvo_debug_get_ba_trace(mvo_ctx* ctx, mvo_ba_handle* handle, double* rows, int cap, int* n) {
    if (!ctx) return MVO_ERR_INVALID;
    return ba_get_trace(ctx, handle, rows, cap, n);
}
int mvo_debug_get_ba_plan(mvo_ctx* ctx, mvo_ba_handle* handle, int* wgs, int* nsplit, int32_t* wg_pt_start, int cap) {
    if (!ctx) return MVO_ERR_INVALID;
    return ba_get_plan(ctx, handle, wgs, nsplit, wg_pt_start, cap);
}

int mvo_debug_ba_demand_replay(const double* times, int n, uint8_t* decisions) {
    if (!times || !decisions || n < 0) return MVO_ERR_INVALID;
    return ba_demand_replay(times, n, decisions);
}

int mvo_debug_get_ba_phases(mvo_ctx* ctx, long long* cycles, int n, int* wgs) {
    if (!ctx || !cycles) return MVO_ERR_INVALID;
    for (int i = 0; i < n && i < 16; ++i) cycles[i] = ctx->ba_phase[i];
    if (wgs) *wgs = ctx->ba_wgs;
    return MVO_OK;
}

}  // extern "C"
