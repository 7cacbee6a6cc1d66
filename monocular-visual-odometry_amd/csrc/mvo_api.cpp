// csrc/mvo_api.cpp -- the C-ABI of libmvo_hip.so (include/mvo_hip.h).  Thin: argument checks, H2D/D2H staging,
// the reference's host-side filters (thresholds, Lowe ratio, de-dup) and calls into the kernel launchers.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>

#include "mvo_internal.h"

int orb_setup_geometry(mvo_ctx* ctx, int w, int h);
int orb_grid_select(mvo_ctx* ctx, std::vector<mvo_keypoint>& kps, int image_rows, int image_cols);
int orb_detect_device(mvo_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, int channels,
                      std::vector<mvo_keypoint>& out);
int orb_describe_device(mvo_ctx* ctx, std::vector<mvo_keypoint>& kps, int w, int h, uint8_t* desc_host);
void orb_border_filter(std::vector<mvo_keypoint>& kps, int w, int h);

int mvo_set_err(mvo_ctx* c, int code, const char* what, hipError_t e) {
    if (c) {
        c->err = what ? what : "";
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

// ---------------------------------------------------------------------------------------------- profiling
void mvo_prof_begin(mvo_ctx* c, const char* name) {
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!c->prof_pool.empty()) {
        ev = c->prof_pool.back();
        c->prof_pool.pop_back();
    } else {
        (void)hipEventCreate(&ev.first);
        (void)hipEventCreate(&ev.second);
    }
    (void)hipEventRecord(ev.first, c->stream);
    c->prof_pending.push_back({name, ev});
}
void mvo_prof_end(mvo_ctx* c) { (void)hipEventRecord(c->prof_pending.back().second.second, c->stream); }
void mvo_prof_collect(mvo_ctx* c) {
    if (c->prof_pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            ProfEntry& e = c->prof_acc[p.first];
            e.launches++;
            e.ms += ms;
        }
        c->prof_pool.push_back(p.second);
    }
    c->prof_pending.clear();
}

int ba_debug_set(const char* key, int value);  // mvo_api_ba.cpp: the "ba_*" knobs
// ---- admission gate (mvo_internal.h)
std::atomic<int> g_extract_concurrency{std::getenv("MVO_EXTRACT_CONCURRENCY") ? std::atoi(std::getenv("MVO_EXTRACT_CONCURRENCY")) : 8};  // (read by waiting threads, set by mvo_set_extract_concurrency)
namespace {
struct GateState {
    std::mutex m;
    std::condition_variable cv;
    int in_flight = 0;
};
GateState* g_gates = new GateState[16];  // (never destroyed: contexts may outlive static destruction order)
}  // namespace
ExtractGate::ExtractGate(const mvo_ctx* ctx) {
    const int cap = g_extract_concurrency.load(std::memory_order_relaxed);
    if (!ctx || !ctx->ba_throughput_mode || cap <= 0) return;
    device = ctx->device & 15;
    GateState& g = g_gates[device];
    std::unique_lock<std::mutex> lk(g.m);
    // (a limit raised meanwhile lets the section in; 0 = gate switched off while waiting)
    g.cv.wait(lk, [&] {
        const int lim = g_extract_concurrency.load(std::memory_order_relaxed);
        return lim <= 0 || g.in_flight < lim;
    });
    ++g.in_flight;
}
void ExtractGate::release() {
    if (device < 0) return;
    GateState& g = g_gates[device];
    {
        std::lock_guard<std::mutex> lk(g.m);
        --g.in_flight;
    }
    g.cv.notify_one();
    device = -1;
}

static int g_match_host_out = 1;  // measurement knob: 0 = k_knn2 delivers into HBM, a copy follows

extern "C" {

int mvo_create(mvo_ctx** out, int device) {
    if (!out) return MVO_ERR_INVALID;
    *out = nullptr;
    // One ctx = one HIP stream; streams beyond the runtime's hardware-queue limit (default 4) share a queue and
    // their kernels serialise.  Takes effect only if the HIP runtime has not been initialised yet.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MVO_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MVO_ERR_NO_DEVICE;
    mvo_ctx* ctx = new mvo_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return MVO_ERR_NO_DEVICE;
    }
    // config/config.yaml:65-69,94-95
    ctx->orb = mvo_orb_params{8000, 1.2f, 4, 20, 1500, 16, 8, 1};  // config.yaml:65-69,94-95; pyramid: INTER_LINEAR_EXACT
    ctx->orb_configured = true;
    *out = ctx;
    return MVO_OK;
}

unsigned long long mvo_ctx_uid(const mvo_ctx* ctx) { return ctx ? ctx->uid : 0ull; }

int mvo_set_wait_policy(int device, int policy) {
    if (policy < MVO_WAIT_AUTO || policy > MVO_WAIT_BLOCK) return MVO_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MVO_ERR_NO_DEVICE;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (hipSetDevice(device) != hipSuccess) return MVO_ERR_NO_DEVICE;
    static const unsigned flags[4] = {hipDeviceScheduleAuto, hipDeviceScheduleSpin, hipDeviceScheduleYield, hipDeviceScheduleBlockingSync};
    const hipError_t e = hipSetDeviceFlags(flags[policy]);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);  // (the flags belong to `device`; the caller's current device is not ours to change)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return MVO_ERR_HIP;
    }
    return MVO_OK;
}

int mvo_set_extract_concurrency(int n) {
    const int prev = g_extract_concurrency.exchange(n < 0 ? 0 : n);
    // a raised limit (or 0 = gate off) lets waiting sections in.  The notification is made under each gate's mutex: a waiter that
    // has evaluated its predicate with the old limit still holds that mutex until it blocks, so it cannot miss this wake-up
    for (int d = 0; d < 16; ++d) {
        std::lock_guard<std::mutex> lk(g_gates[d].m);
        g_gates[d].cv.notify_all();
    }
    return prev;
}

int mvo_create_sibling(mvo_ctx* parent, mvo_ctx** out) {
    if (!out) return MVO_ERR_INVALID;
    *out = nullptr;
    if (!parent) return MVO_ERR_INVALID;
    if (hipSetDevice(parent->device) != hipSuccess) return MVO_ERR_NO_DEVICE;
    mvo_ctx* ctx = new mvo_ctx();
    ctx->device = parent->device;
    ctx->stream = parent->stream;  // no stream (= no share of a hardware queue) of its own
    ctx->owns_stream = false;
    if (hipEventCreateWithFlags(&ctx->ev, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return MVO_ERR_NO_DEVICE;
    }
    ctx->orb = parent->orb;
    ctx->orb_configured = true;
    *out = ctx;
    return MVO_OK;
}

void mvo_destroy(mvo_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    track_release(ctx);
    ba_pool_release(ctx);
    void* dev[] = {ctx->d_img, ctx->d_raw,  ctx->d_blur, ctx->d_tabs, ctx->d_pyr_regs,
                   ctx->d_kp,   ctx->d_desc_buf, ctx->d_mq,    ctx->d_mt,   ctx->d_mqxy,      ctx->d_mtxy,
                   ctx->d_mout, ctx->d_fh_slots, ctx->d_fh_line, ctx->d_fh_arrive};
    for (void* p : dev)
        if (p) mvo_free_on_current_device(p);
    if (ctx->h_pin) ba_service_free(ctx->device, ctx->h_pin, true);
    for (auto& p : ctx->prof_pending) ctx->prof_pool.push_back(p.second);
    for (auto& e : ctx->prof_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    (void)hipEventDestroy(ctx->ev);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* mvo_last_error(const mvo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int mvo_synchronize(mvo_ctx* ctx) {
    if (!ctx) return MVO_ERR_INVALID;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    ba_service_park(ctx->device);
    return MVO_OK;
}

int mvo_orb_configure(mvo_ctx* ctx, const mvo_orb_params* p) {
    if (!ctx || !p) return MVO_ERR_INVALID;
    if (p->nlevels < 1 || p->nlevels > MVO_MAX_LEVELS || p->scale_factor <= 1.f || p->nfeatures < 0 ||
        p->fast_threshold < 1 || p->fast_threshold > 254 || p->grid_size < 1 || p->grid_max_per_cell < 0 ||
        p->max_keypoints < 0)
        return mvo_set_err(ctx, MVO_ERR_INVALID, "mvo_orb_configure: parameter out of range", hipSuccess);
    ctx->orb = *p;
    ctx->orb_configured = true;
    ctx->grid_rows = ctx->grid_cols = 0;
    ctx->img_w = ctx->img_h = 0;  // forces the geometry (level sizes, quotas, tables) to be rebuilt
    ctx->pyr_valid = ctx->blur_valid = false;
    return MVO_OK;
}

static int check_image(mvo_ctx* ctx, const void* img, int w, int h, int stride, int ch) {
    if (!ctx) return MVO_ERR_INVALID;
    if (!img || w < 1 || h < 1 || (ch != 1 && ch != 3 && ch != 4) || stride < w * ch)
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad image arguments", hipSuccess);
    return MVO_OK;
}

static int upload_image(mvo_ctx* ctx, const uint8_t* img, int h, int stride) {
    const size_t bytes = (size_t)h * stride;
    if (ctx->d_img_cap < bytes) {
        if (ctx->d_img) ba_service_free(ctx->device, ctx->d_img, false);
        ctx->d_img = nullptr;
        ctx->d_img_cap = 0;
        MVO_HIP(hipMalloc((void**)&ctx->d_img, bytes + 64));
        ctx->d_img_cap = bytes;
    }
    MVO_HIP(hipMemcpyAsync(ctx->d_img, img, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MVO_OK;
}

static int calc_keypoints_common(mvo_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, int ch,
                                 mvo_keypoint* kps, int cap, int* n) {
    std::vector<mvo_keypoint> v;
    int r = orb_detect_device(ctx, d_img, w, h, stride, ch, v);
    if (r) return r;
    if ((r = orb_grid_select(ctx, v, h, w))) return r;
    if (ctx->prof) mvo_prof_collect(ctx);
    *n = (int)v.size();
    if ((int)v.size() > cap) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "keypoint buffer too small", hipSuccess);
    std::copy(v.begin(), v.end(), kps);
    return MVO_OK;
}

int mvo_calc_keypoints(mvo_ctx* ctx, const uint8_t* image, int w, int h, int stride, int ch, mvo_keypoint* kps,
                       int cap, int* n) {
    int r = check_image(ctx, image, w, h, stride, ch);
    if (r) return r;
    if (!kps || !n) return mvo_set_err(ctx, MVO_ERR_INVALID, "null output", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    if ((r = upload_image(ctx, image, h, stride))) return r;
    return calc_keypoints_common(ctx, ctx->d_img, w, h, stride, ch, kps, cap, n);
}

int mvo_calc_keypoints_dev(mvo_ctx* ctx, const void* d_image, int w, int h, int stride, int ch, mvo_keypoint* kps,
                           int cap, int* n) {
    int r = check_image(ctx, d_image, w, h, stride, ch);
    if (r) return r;
    if (!kps || !n) return mvo_set_err(ctx, MVO_ERR_INVALID, "null output", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    return calc_keypoints_common(ctx, (const uint8_t*)d_image, w, h, stride, ch, kps, cap, n);
}

static int describe_common(mvo_ctx* ctx, int w, int h, mvo_keypoint* kps, int* n, uint8_t* desc,
                           std::vector<mvo_keypoint>& v) {
    v.assign(kps, kps + *n);
    orb_border_filter(v, w, h);
    for (const mvo_keypoint& k : v)
        if (k.octave < 0 || k.octave >= ctx->pyr_levels_built)
            return mvo_set_err(ctx, MVO_ERR_INVALID, "keypoint octave outside the built pyramid", hipSuccess);
    int r = orb_describe_device(ctx, v, w, h, desc);
    if (r) return r;
    if (ctx->prof) mvo_prof_collect(ctx);
    std::copy(v.begin(), v.end(), kps);
    *n = (int)v.size();
    return MVO_OK;
}

int mvo_calc_descriptors(mvo_ctx* ctx, const uint8_t* image, int w, int h, int stride, int ch, int reuse_pyramid,
                         mvo_keypoint* kps, int* n, uint8_t* desc, uint8_t* rgb) {
    if (!ctx || !kps || !n || !desc || *n < 0) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    int r;
    if (reuse_pyramid) {
        if (!ctx->pyr_valid || ctx->img_w != w || ctx->img_h != h)
            return mvo_set_err(ctx, MVO_ERR_STATE, "reuse_pyramid without a cached pyramid of this size", hipSuccess);
        if (rgb && (r = check_image(ctx, image, w, h, stride, ch))) return r;
    } else {
        if ((r = check_image(ctx, image, w, h, stride, ch))) return r;
        // cv::ORB::compute: nLevels = max octave + 1 (orb2 is created with level_pyramid, feature_match.cpp:45)
        int need = 1;
        for (int i = 0; i < *n; ++i) need = std::max(need, kps[i].octave + 1);
        if ((r = orb_setup_geometry(ctx, w, h))) return r;
        if (need > ctx->pyr.nlevels)
            return mvo_set_err(ctx, MVO_ERR_INVALID, "keypoint octave exceeds level_pyramid", hipSuccess);
        if ((r = upload_image(ctx, image, h, stride))) return r;
        ctx->pyr_valid = ctx->blur_valid = false;
        if ((r = orb_launch_pyramid(ctx, ctx->d_img, stride, ch, need))) return r;
        ctx->pyr_levels_built = need;
        ctx->pyr_valid = true;
    }
    std::vector<mvo_keypoint> v;
    if ((r = describe_common(ctx, w, h, kps, n, desc, v))) return r;
    if (rgb) {  // frame.h:80-85 + basics::getPixelAt (opencv_funcs.cpp:10-32): BGR -> r,g,b
        for (size_t j = 0; j < v.size(); ++j) {
            int x = (int)std::floor(v[j].x), y = (int)std::floor(v[j].y);
            const uint8_t* px = image + (size_t)y * stride + (size_t)x * ch;
            if (ch >= 3) {
                rgb[3 * j] = px[2];
                rgb[3 * j + 1] = px[1];
                rgb[3 * j + 2] = px[0];
            } else {
                rgb[3 * j] = rgb[3 * j + 1] = rgb[3 * j + 2] = px[0];
            }
        }
    }
    return MVO_OK;
}

int mvo_calc_descriptors_dev(mvo_ctx* ctx, mvo_keypoint* kps, int* n, uint8_t* desc, const void** d_desc_out) {
    if (!ctx || !kps || !n || *n < 0) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    if (!ctx->pyr_valid)
        return mvo_set_err(ctx, MVO_ERR_STATE, "no cached pyramid: call mvo_calc_keypoints[_dev] first", hipSuccess);
    std::vector<mvo_keypoint> v;
    int r = describe_common(ctx, ctx->img_w, ctx->img_h, kps, n, desc, v);
    if (r) return r;
    if (!desc) MVO_HIP(hipStreamSynchronize(ctx->stream));
    if (d_desc_out) *d_desc_out = ctx->d_desc;
    return MVO_OK;
}

int mvo_select_uniform_kpts_by_grid(mvo_ctx* ctx, mvo_keypoint* kps, int* n, int image_rows, int image_cols) {
    if (!ctx || !kps || !n || *n < 0) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    std::vector<mvo_keypoint> v(kps, kps + *n);
    int r = orb_grid_select(ctx, v, image_rows, image_cols);
    if (r) return r;
    std::copy(v.begin(), v.end(), kps);
    *n = (int)v.size();
    return MVO_OK;
}

// ---------------------------------------------------------------------------------------------- matching
static int ensure_match_bufs(mvo_ctx* ctx, int nq, int nt) {
    if (nq > ctx->m_cap_q) {
        if (ctx->d_mq) ba_service_free(ctx->device, ctx->d_mq, false);
        if (ctx->d_mqxy) ba_service_free(ctx->device, ctx->d_mqxy, false);
        if (ctx->d_mout) ba_service_free(ctx->device, ctx->d_mout, false);
        ctx->d_mq = nullptr;
        ctx->d_mqxy = nullptr;
        ctx->d_mout = nullptr;
        ctx->m_cap_q = 0;
        int cap = std::max(4096, nq + nq / 2);
        MVO_HIP(hipMalloc((void**)&ctx->d_mq, (size_t)cap * 32));
        MVO_HIP(hipMalloc((void**)&ctx->d_mqxy, (size_t)cap * 8));
        // results + up to 64 slices of partials + one arrival counter per group of 64 queries (self re-arming, zeroed once)
        const size_t body = (size_t)cap * (16 + 64 * 16), ctr = ((size_t)cap / 64 + 2) * 4;
        MVO_HIP(hipMalloc((void**)&ctx->d_mout, body + ctr));
        ctx->d_marrive = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ctx->d_mout) + body);
        MVO_HIP(hipMemsetAsync(ctx->d_marrive, 0, ctr, ctx->stream));
        ctx->m_cap_q = cap;
    }
    if (nt > ctx->m_cap_t) {
        if (ctx->d_mt) ba_service_free(ctx->device, ctx->d_mt, false);
        if (ctx->d_mtxy) ba_service_free(ctx->device, ctx->d_mtxy, false);
        ctx->d_mt = nullptr;
        ctx->d_mtxy = nullptr;
        ctx->m_cap_t = 0;
        int cap = std::max(4096, nt + nt / 2);
        MVO_HIP(hipMalloc((void**)&ctx->d_mt, (size_t)cap * 32));
        MVO_HIP(hipMalloc((void**)&ctx->d_mtxy, (size_t)cap * 8));
        ctx->m_cap_t = cap;
    }
    return MVO_OK;
}

static int knn2_common(mvo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* idx,
                       int32_t* dist) {
    // the merge kernel writes the nq x (idx[2], dist[2]) block straight into the pinned staging buffer
    int r = mvo_ensure_pinned(ctx, (size_t)nq * 16);
    if (r) return r;
    ExtractGate gate(ctx);
    if (g_match_host_out) {
        if ((r = match_launch_knn2(ctx, d_q, nq, d_t, nt, ctx->d_mout, reinterpret_cast<int32_t*>(ctx->h_pin)))) return r;
    } else {  // measurement knob: the kernel delivers into HBM and a copy follows (kernel time without the PCIe write tail)
        if ((r = match_launch_knn2(ctx, d_q, nq, d_t, nt, ctx->d_mout, nullptr))) return r;
        MVO_HIP(hipMemcpyAsync(ctx->h_pin, ctx->d_mout, (size_t)nq * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    gate.release();
    if (ctx->prof) mvo_prof_collect(ctx);
    std::memcpy(idx, ctx->h_pin, (size_t)nq * 8);
    std::memcpy(dist, ctx->h_pin + (size_t)nq * 8, (size_t)nq * 8);
    return MVO_OK;
}

int mvo_match_knn2(mvo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    if (!ctx || nq < 0 || nt < 0 || (nq && (!q || !idx || !dist)) || (nt && !t))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (nq == 0) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    int r = ensure_match_bufs(ctx, nq, nt);
    if (r) return r;
    MVO_HIP(hipMemcpyAsync(ctx->d_mq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    if (nt) MVO_HIP(hipMemcpyAsync(ctx->d_mt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    return knn2_common(ctx, ctx->d_mq, nq, ctx->d_mt, nt, idx, dist);
}

int mvo_match_knn2_dev(mvo_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, int32_t* idx, int32_t* dist) {
    if (!ctx || nq < 0 || nt < 0 || (nq && (!d_q || !idx || !dist)) || (nt && !d_t))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (nq == 0) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    int r = ensure_match_bufs(ctx, nq, 0);
    if (r) return r;
    return knn2_common(ctx, (const uint8_t*)d_q, nq, (const uint8_t*)d_t, nt, idx, dist);
}

int mvo_match_radius_l1(mvo_ctx* ctx, const uint8_t* q, const float* qxy, int nq, const uint8_t* t, const float* txy,
                        int nt, float max_px, int32_t* idx, int32_t* sum) {
    if (!ctx || nq < 0 || nt < 0 || (nq && (!q || !qxy || !idx || !sum)) || (nt && (!t || !txy)))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (nq == 0) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    int r = ensure_match_bufs(ctx, nq, nt);
    if (r) return r;
    MVO_HIP(hipMemcpyAsync(ctx->d_mq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(ctx->d_mqxy, qxy, (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nt) {
        MVO_HIP(hipMemcpyAsync(ctx->d_mt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
        MVO_HIP(hipMemcpyAsync(ctx->d_mtxy, txy, (size_t)nt * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((r = match_launch_radius_l1(ctx, ctx->d_mq, ctx->d_mqxy, nq, ctx->d_mt, ctx->d_mtxy, nt, max_px, ctx->d_mout)))
        return r;
    if ((r = mvo_ensure_pinned(ctx, (size_t)nq * 8))) return r;
    MVO_HIP(hipMemcpyAsync(ctx->h_pin, ctx->d_mout, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) mvo_prof_collect(ctx);
    std::memcpy(idx, ctx->h_pin, (size_t)nq * 4);
    std::memcpy(sum, ctx->h_pin + (size_t)nq * 4, (size_t)nq * 4);
    return MVO_OK;
}

int mvo_remove_duplicated_matches(mvo_dmatch* m, int* n) {
    if (!n || *n < 0 || (*n && !m)) return MVO_ERR_INVALID;
    // feature_match.cpp:241-260: std::sort by trainIdx (unstable; comparator ignores the distance), keep the
    // first of every run
    std::sort(m, m + *n, [](const mvo_dmatch& a, const mvo_dmatch& b) { return a.trainIdx < b.trainIdx; });
    int k = 0;
    for (int i = 0; i < *n; ++i)
        if (i == 0 || m[i].trainIdx != m[i - 1].trainIdx) m[k++] = m[i];
    // (in-place compaction is safe: k <= i, and m[i-1] is read before slot i-1 can be overwritten only by itself)
    *n = k;
    return MVO_OK;
}

int mvo_match_features(mvo_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int method,
                       double xiang_gao_ratio, double lowe_ratio, const float* xy1, const float* xy2, float max_px,
                       mvo_dmatch* out, int cap, int* n) {
    if (!ctx || !n || n1 < 0 || n2 < 0) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    *n = 0;
    std::vector<mvo_dmatch> matches;
    double min_dis = 9999999, max_dis = 0;
    int r;
    if (method == 1 || method == 3) {
        std::vector<mvo_dmatch> all;
        if (method == 3) {
            if (n1 && n2 && (!xy1 || !xy2))
                return mvo_set_err(ctx, MVO_ERR_INVALID, "method 3 needs keypoint positions", hipSuccess);
            std::vector<int32_t> idx(n1), sum(n1);
            if ((r = mvo_match_radius_l1(ctx, d1, xy1, n1, d2, xy2, n2, max_px, idx.data(), sum.data()))) return r;
            for (int i = 0; i < n1; ++i)
                if (idx[i] >= 0) all.push_back({i, idx[i], -1, (float)((double)sum[i] / 32)});
        } else {
            std::vector<int32_t> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
            if ((r = mvo_match_knn2(ctx, d1, n1, d2, n2, idx.data(), dist.data()))) return r;
            for (int i = 0; i < n1; ++i)
                if (idx[2 * i] >= 0) all.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
        for (const mvo_dmatch& m : all) {  // feature_match.cpp:179-186
            double dist = m.distance;
            if (dist < min_dis) min_dis = dist;
            if (dist > max_dis) max_dis = dist;
        }
        const double thr = std::max<float>(min_dis * xiang_gao_ratio, 30.0);  // :187
        for (const mvo_dmatch& m : all)
            if (m.distance < thr) matches.push_back(m);
    } else if (method == 2) {
        std::vector<int32_t> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
        if ((r = mvo_match_knn2(ctx, d1, n1, d2, n2, idx.data(), dist.data()))) return r;
        for (int i = 0; i < n1; ++i) {
            if (idx[2 * i + 1] < 0) continue;  // the reference would read knn_matches[i][1] out of bounds
            const double d = (float)dist[2 * i];
            if (d < lowe_ratio * (float)dist[2 * i + 1]) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
    } else {
        return mvo_set_err(ctx, MVO_ERR_INVALID, "feature_match.cpp::matchFeatures: wrong method index.", hipSuccess);
    }
    int cnt = (int)matches.size();
    mvo_remove_duplicated_matches(matches.data(), &cnt);
    *n = cnt;
    if (cnt > cap || (cnt && !out)) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "match buffer too small", hipSuccess);
    std::copy(matches.begin(), matches.begin() + cnt, out);
    return MVO_OK;
}


// matchFeatures with both descriptor sets already in HBM (e.g. the ping-pong buffers of mvo_calc_descriptors_dev)
int mvo_match_features_dev(mvo_ctx* ctx, const void* d_d1, int n1, const void* d_d2, int n2, int method,
                           double xiang_gao_ratio, double lowe_ratio, mvo_dmatch* out, int cap, int* n) {
    if (!ctx || !n || n1 < 0 || n2 < 0) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    *n = 0;
    if (method != 1 && method != 2)
        return mvo_set_err(ctx, MVO_ERR_INVALID, "feature_match.cpp::matchFeatures: wrong method index.", hipSuccess);
    std::vector<int32_t> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
    int r = mvo_match_knn2_dev(ctx, d_d1, n1, d_d2, n2, idx.data(), dist.data());
    if (r) return r;
    std::vector<mvo_dmatch> matches;
    if (method == 1) {
        double min_dis = 9999999;
        for (int i = 0; i < n1; ++i)
            if (idx[2 * i] >= 0 && dist[2 * i] < min_dis) min_dis = dist[2 * i];
        const double thr = std::max<float>(min_dis * xiang_gao_ratio, 30.0);
        for (int i = 0; i < n1; ++i)
            if (idx[2 * i] >= 0 && (float)dist[2 * i] < thr) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
    } else {
        for (int i = 0; i < n1; ++i) {
            if (idx[2 * i + 1] < 0) continue;
            const double d = (float)dist[2 * i];
            if (d < lowe_ratio * (float)dist[2 * i + 1]) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
    }
    int cnt = (int)matches.size();
    mvo_remove_duplicated_matches(matches.data(), &cnt);
    *n = cnt;
    if (cnt > cap || (cnt && !out)) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "match buffer too small", hipSuccess);
    std::copy(matches.begin(), matches.begin() + cnt, out);
    return MVO_OK;
}

// ---------------------------------------------------------------------------------------------- debug hooks
int mvo_debug_set(const char* key, int value) {
    if (key && !std::strncmp(key, "ba_", 3)) return ba_debug_set(key, value);
    if (key && !std::strcmp(key, "extract_concurrency")) {
        (void)mvo_set_extract_concurrency(value);  // (the one setter: it also wakes the sections waiting at the gate)
        return MVO_OK;
    }
    if (key && !std::strcmp(key, "match_host_out")) {
        g_match_host_out = value;
        return MVO_OK;
    }
    if (key && !std::strcmp(key, "match_mfma")) {
        g_match_mfma = value;
        return MVO_OK;
    }
    if (key && !std::strcmp(key, "pyr_force_chain")) {
        g_pyr_force_chain = value;
        return MVO_OK;
    }
    if (key && !std::strcmp(key, "pnp_replay_skew")) {
        g_pnp_replay_skew = value;
        return MVO_OK;
    }
    return MVO_ERR_INVALID;
}

int mvo_debug_get_level(mvo_ctx* ctx, int level, int blurred, uint8_t* out, int cap, int* w, int* h, int* stride) {
    if (!ctx || !ctx->pyr_valid || level < 0 || level >= ctx->pyr_levels_built)
        return mvo_set_err(ctx, MVO_ERR_STATE, "no such cached level", hipSuccess);
    const LevelInfo& L = ctx->pyr.lv[level];
    const size_t bytes = (size_t)L.stride * (L.h + 2 * MVO_BORDER);
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (stride) *stride = L.stride;
    if (!out) return MVO_OK;
    if ((size_t)cap < bytes) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "level buffer too small", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    if (blurred && !ctx->blur_valid) {  // the product path blurs per keypoint window; the whole level only on request
        int r = orb_launch_blur(ctx, ctx->pyr_levels_built);
        if (r) return r;
    }
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    MVO_HIP(hipMemcpy(out, (blurred ? ctx->d_blur : ctx->d_raw) + L.off, bytes, hipMemcpyDeviceToHost));
    return MVO_OK;
}

int mvo_debug_get_candidates(mvo_ctx* ctx, void* out, int cap, int* n) {
    if (!ctx || !ctx->pyr_valid || !n) return mvo_set_err(ctx, MVO_ERR_STATE, "no detection cached", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    *n = (int)ctx->last_cand.size();
    if (!out) return MVO_OK;
    if (*n > cap) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "candidate buffer too small", hipSuccess);
    std::copy(ctx->last_cand.begin(), ctx->last_cand.end(), static_cast<DevCandidate*>(out));
    return MVO_OK;
}

// ---------------------------------------------------------------------------------------------- measurement
int mvo_profile_enable(mvo_ctx* ctx, int on) {
    if (!ctx) return MVO_ERR_INVALID;
    mvo_prof_collect(ctx);
    ctx->prof = on != 0;
    return MVO_OK;
}
int mvo_profile_reset(mvo_ctx* ctx) {
    if (!ctx) return MVO_ERR_INVALID;
    mvo_prof_collect(ctx);
    ctx->prof_acc.clear();
    return MVO_OK;
}
int mvo_profile_get(mvo_ctx* ctx, mvo_kernel_time* out, int cap) {
    if (!ctx) return MVO_ERR_INVALID;
    mvo_prof_collect(ctx);
    int i = 0;
    for (auto& kv : ctx->prof_acc) {
        if (i < cap && out) {
            std::snprintf(out[i].name, sizeof(out[i].name), "%s", kv.first.c_str());
            out[i].launches = kv.second.launches;
            out[i].total_ms = kv.second.ms;
        }
        ++i;
    }
    return i;
}

}  // extern "C"
