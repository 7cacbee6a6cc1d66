// csrc/track_host.cpp -- host side of the tracking rows (include/mvo_hip.h "tracking"): device buffers, the
// sequential parts of cv::solvePnPRansac (subset drawing with cv::RNG, the adaptive iteration count) and the
// map residency used by getMappointsInCurrentView_.  Reference: src/vo/vo.cpp:16-49 and 270-357.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "mvo_internal.h"

struct mvo_map {
    float* d_pos = nullptr;
    uint8_t* d_desc = nullptr;
    int n = 0, cap = 0;
};

struct mvo_track_state {
    // PnP: pairs, subsets, per-hypothesis results, refinement scratch
    double *d_Mg = nullptr, *d_mg = nullptr;
    int cap_n = 0;
    uint8_t* d_in = nullptr;  // pairs + subsets of a solvePnPRansac call, contiguous like their pinned staging (one upload)
    size_t cap_in = 0;
    double* d_models = nullptr;
    int32_t* d_counts = nullptr;
    int cap_h = 0;
    uint8_t* d_masks = nullptr;
    int last_loop_len = 0;  // iterations the RANSAC loop of the previous mvo_solve_pnp_ransac on this ctx ran (0: none yet)
    size_t cap_masks = 0;
    // record of the last solve (mvo_debug_get_pnp)
    std::vector<double> models;
    std::vector<int32_t> counts;
    int32_t info[6] = {-1, 0, 0, 0, 0, 0};
    // triangulation
    float *d_tri_in = nullptr, *d_tri_out = nullptr;
    int cap_tri = 0;
    // essential-matrix RANSAC
    double* d_emq = nullptr;  // q1 (2n) then q2 (2n)
    uint8_t* d_em_mask = nullptr;
    int cap_em = 0;
    int32_t *d_em_subsets = nullptr, *d_em_nm = nullptr, *d_em_counts = nullptr;
    double* d_em_E = nullptr;
    std::vector<int32_t> em_counts;  // record of the last call: [iterations evaluated x 10]
    int32_t em_info[5] = {-1, -1, 0, 0, 0};
    // map points in view
    uint8_t* d_view_desc = nullptr;
    int32_t* d_view_n = nullptr;
    int cap_view = 0;
};

int g_pnp_replay_skew = 0;

namespace {

template <class T>
void free_dev(T*& p) {
    if (p) mvo_free_on_current_device(p);
    p = nullptr;
}

mvo_track_state* state(mvo_ctx* ctx) {
    if (!ctx->track) ctx->track = new mvo_track_state();
    return ctx->track;
}

int ensure_pnp(mvo_ctx* ctx, int n, int n_hyp) {
    mvo_track_state* s = state(ctx);
    if (n > s->cap_n) {
        free_dev(s->d_Mg);
        free_dev(s->d_mg);
        s->cap_n = 0;
        const int cap = std::max(4096, n + n / 2);
        MVO_HIP(hipMalloc((void**)&s->d_Mg, (size_t)cap * 3 * sizeof(double)));
        MVO_HIP(hipMalloc((void**)&s->d_mg, (size_t)cap * 2 * sizeof(double)));
        s->cap_n = cap;
    }
    if (n_hyp > s->cap_h) {
        free_dev(s->d_models);
        free_dev(s->d_counts);
        s->cap_h = 0;
        const int cap = std::max(128, n_hyp);
        MVO_HIP(hipMalloc((void**)&s->d_models, (size_t)cap * 12 * sizeof(double)));
        MVO_HIP(hipMalloc((void**)&s->d_counts, (size_t)cap * sizeof(int32_t)));
        s->cap_h = cap;
    }
    const size_t in_need = (size_t)n * 20 + (size_t)n_hyp * 20 + 64;
    if (in_need > s->cap_in) {
        free_dev(s->d_in);
        s->cap_in = 0;
        const size_t cap = in_need + in_need / 2 + 4096;
        MVO_HIP(hipMalloc((void**)&s->d_in, cap));
        s->cap_in = cap;
    }
    const size_t need = (size_t)n_hyp * (size_t)n;
    if (need > s->cap_masks) {
        free_dev(s->d_masks);
        s->cap_masks = 0;
        const size_t cap = std::max<size_t>(need + need / 2, (size_t)1 << 20);
        MVO_HIP(hipMalloc((void**)&s->d_masks, cap));
        s->cap_masks = cap;
    }
    return MVO_OK;
}

// cv::RNG (multiply-with-carry) as RANSACPointSetRegistrator::run seeds it: RNG rng((uint64)-1).
struct MwcRng {
    uint64_t state = 0xffffffffffffffffULL;
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690ULL + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int lo, int hi) { return lo == hi ? lo : (int)(next() % (uint32_t)(hi - lo) + lo); }
};

// RANSACPointSetRegistrator::getSubset: draw until the slot differs from the earlier ones (PnP's checkSubset
// accepts every sample).
void draw_subsets(int count, int n_iters, int32_t* out) {
    MwcRng rng;
    for (int it = 0; it < n_iters; ++it) {
        int32_t* s = out + 5 * it;
        for (int i = 0; i < 5; ++i) {
            bool fresh;
            do {
                s[i] = rng.uniform(0, count);
                fresh = true;
                for (int j = 0; j < i; ++j) fresh = fresh && s[j] != s[i];
            } while (!fresh);
        }
    }
}

// cv::RANSACUpdateNumIters
int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

// cv::Mat::inv() (DECOMP_LU) of a 4x4 matrix; `rows` = how many rows of the inverse the caller wants
bool invert_pose_lu(const double* T, double* out, int rows = 3) {
    double A[4][4], B[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            A[i][j] = T[4 * i + j];
            B[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int j = i + 1; j < 4; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[piv][i])) piv = j;
        if (std::fabs(A[piv][i]) < DBL_EPSILON * 100) return false;
        if (piv != i) {
            std::swap_ranges(A[i], A[i] + 4, A[piv]);
            std::swap_ranges(B[i], B[i] + 4, B[piv]);
        }
        const double d = -1 / A[i][i];
        for (int j = i + 1; j < 4; ++j) {
            const double alpha = A[j][i] * d;
            for (int c = i + 1; c < 4; ++c) A[j][c] += alpha * A[i][c];
            for (int c = 0; c < 4; ++c) B[j][c] += alpha * B[i][c];
        }
    }
    for (int i = 3; i >= 0; --i)
        for (int j = 0; j < 4; ++j) {
            double s = B[i][j];
            for (int c = i + 1; c < 4; ++c) s -= A[i][c] * B[c][j];
            B[i][j] = s / A[i][i];
        }
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = B[i][j];
    return true;
}

}  // namespace

void track_release(mvo_ctx* ctx) {
    mvo_track_state* s = ctx->track;
    if (!s) return;
    free_dev(s->d_Mg);
    free_dev(s->d_mg);
    free_dev(s->d_in);
    free_dev(s->d_models);
    free_dev(s->d_counts);
    free_dev(s->d_masks);
    free_dev(s->d_view_desc);
    free_dev(s->d_view_n);
    free_dev(s->d_tri_in);
    free_dev(s->d_tri_out);
    free_dev(s->d_emq);
    free_dev(s->d_em_mask);
    free_dev(s->d_em_subsets);
    free_dev(s->d_em_nm);
    free_dev(s->d_em_counts);
    free_dev(s->d_em_E);
    delete s;
    ctx->track = nullptr;
}

extern "C" {

// ---------------------------------------------------------------------------------------------- map residency
int mvo_map_create(mvo_ctx* ctx, mvo_map** map) {
    if (!ctx || !map) return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    *map = new mvo_map();
    return MVO_OK;
}

void mvo_map_release(mvo_ctx* ctx, mvo_map* map) {
    if (!map) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    free_dev(map->d_pos);
    free_dev(map->d_desc);
    delete map;
}

int mvo_map_upload(mvo_ctx* ctx, mvo_map* map, const float* pos, const uint8_t* desc, int n) {
    if (!ctx || !map || n < 0 || (n && (!pos || !desc)))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    if (n > map->cap) {
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        free_dev(map->d_pos);
        free_dev(map->d_desc);
        map->cap = 0;
        const int cap = std::max(4096, n + n / 2);
        MVO_HIP(hipMalloc((void**)&map->d_pos, (size_t)cap * 3 * sizeof(float)));
        MVO_HIP(hipMalloc((void**)&map->d_desc, (size_t)cap * 32));
        map->cap = cap;
    }
    map->n = n;
    if (n) {
        // staged through pinned memory so that the caller's arrays may be reused as soon as we return
        int r = mvo_ensure_pinned(ctx, (size_t)n * 44);
        if (r) return r;
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(ctx->h_pin, pos, (size_t)n * 12);
        std::memcpy(ctx->h_pin + (size_t)n * 12, desc, (size_t)n * 32);
        MVO_HIP(hipMemcpyAsync(map->d_pos, ctx->h_pin, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        MVO_HIP(hipMemcpyAsync(map->d_desc, ctx->h_pin + (size_t)n * 12, (size_t)n * 32, hipMemcpyHostToDevice,
                               ctx->stream));
        MVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MVO_OK;
}

int mvo_map_update_positions(mvo_ctx* ctx, mvo_map* map, const float* pos, int first, int n) {
    if (!ctx || !map || first < 0 || n < 0 || first + n > map->n || (n && !pos))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (!n) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    int r = mvo_ensure_pinned(ctx, (size_t)n * 12);
    if (r) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(ctx->h_pin, pos, (size_t)n * 12);
    MVO_HIP(hipMemcpyAsync(map->d_pos + 3 * (size_t)first, ctx->h_pin, (size_t)n * 12, hipMemcpyHostToDevice,
                           ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    return MVO_OK;
}

int mvo_map_points_in_view(mvo_ctx* ctx, mvo_map* map, const double* T_w_c, double fx, double fy, double cx, double cy,
                           int cols, int rows, int32_t* idx, float* px, int cap, int* n, const void** d_desc_out) {
    if (!ctx || !map || !T_w_c || !n || cap < 0 || (cap && (!idx || !px)))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    *n = 0;
    if (d_desc_out) *d_desc_out = nullptr;
    TrackViewArgs a;
    if (!invert_pose_lu(T_w_c, a.T)) return mvo_set_err(ctx, MVO_ERR_INVALID, "T_w_c is singular", hipSuccess);
    a.fx = fx;
    a.fy = fy;
    a.cx = cx;
    a.cy = cy;
    a.cols = cols;
    a.rows = rows;
    if (map->n == 0) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    mvo_track_state* s = state(ctx);
    if (map->n > s->cap_view) {
        free_dev(s->d_view_desc);
        s->cap_view = 0;
        const int c = std::max(4096, map->n + map->n / 2);
        MVO_HIP(hipMalloc((void**)&s->d_view_desc, (size_t)c * 32));
        s->cap_view = c;
    }
    if (!s->d_view_n) MVO_HIP(hipMalloc((void**)&s->d_view_n, 4));
    // the kernel writes the count, the indices and the pixels straight into the pinned buffer (the descriptors of the survivors
    // stay in HBM for the matcher): one synchronisation, no copy
    int r = mvo_ensure_pinned(ctx, 192 + (size_t)map->n * 12);
    if (r) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));  // (nothing of an earlier call may still be reading the staging buffer)
    int32_t* h_n = reinterpret_cast<int32_t*>(ctx->h_pin);
    int32_t* h_idx = reinterpret_cast<int32_t*>(ctx->h_pin + 64);
    float* h_px = reinterpret_cast<float*>(ctx->h_pin + 64 + ((size_t)map->n * 4 + 63) / 64 * 64);  // (float2 stores)
    if ((r = track_launch_map_in_view(ctx, map->d_pos, map->d_desc, map->n, a, h_idx, h_px, s->d_view_desc, h_n))) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    const int cnt = *h_n;
    *n = cnt;
    if (d_desc_out) *d_desc_out = s->d_view_desc;
    if (ctx->prof) mvo_prof_collect(ctx);
    if (cnt > cap) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "mvo_map_points_in_view: output buffers too small", hipSuccess);
    if (cnt) {
        std::memcpy(idx, h_idx, (size_t)cnt * 4);
        std::memcpy(px, h_px, (size_t)cnt * 8);
    }
    return MVO_OK;
}

// ---------------------------------------------------------------------------------------------- solvePnPRansac
int mvo_solve_pnp_ransac(mvo_ctx* ctx, const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx,
                         double cy, int iterations, float reprojection_error, double confidence, double* rvec,
                         double* tvec, int32_t* inliers, int cap, int* n_inliers, int* found) {
    if (!ctx || n < 0 || (n && (!pts3d || !pts2d)) || !rvec || !tvec || !n_inliers || !found || cap < 0 ||
        (cap && !inliers))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (!(confidence > 0 && confidence < 1))  // CV_Assert in RANSACPointSetRegistrator::run
        return mvo_set_err(ctx, MVO_ERR_INVALID, "mvo_solve_pnp_ransac: confidence must be in (0, 1)", hipSuccess);
    *n_inliers = 0;
    *found = 0;
    for (int k = 0; k < 3; ++k) rvec[k] = tvec[k] = 0;
    constexpr int kModel = 5;
    mvo_track_state* s = state(ctx);
    s->models.clear();
    s->counts.clear();
    s->info[0] = -1;
    for (int k = 1; k < 6; ++k) s->info[k] = 0;
    if (n < kModel) return MVO_OK;  // vo.cpp:321 (kMinPtsForPnP) never gets here; solvePnPRansac would reject it
    if (cap < n) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "mvo_solve_pnp_ransac: inlier buffer smaller than n", hipSuccess);
    const int n_hyp = n == kModel ? 1 : std::max(iterations, 1);
    MVO_HIP(hipSetDevice(ctx->device));
    int r = ensure_pnp(ctx, n, n_hyp);
    if (r) return r;
    // stage pairs + subsets: ONE upload; the kernels deliver their results into the pinned buffer themselves (models and
    // counts by the hypothesis kernel, the refined pose and the inlier mask by the refinement kernel): no copy comes back
    const size_t b3 = (size_t)n * 12, b2 = (size_t)n * 8, bs = (size_t)n_hyp * kModel * 4;
    const size_t o_out = (b3 + b2 + bs + 63) / 64 * 64, o_counts = o_out + 128, o_models = o_counts + ((size_t)n_hyp * 4 + 63) / 64 * 64,
                 o_mask = o_models + (size_t)n_hyp * 96;
    if ((r = mvo_ensure_pinned(ctx, o_mask + (size_t)n + 64))) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(ctx->h_pin, pts3d, b3);
    std::memcpy(ctx->h_pin + b3, pts2d, b2);
    int32_t* subsets = reinterpret_cast<int32_t*>(ctx->h_pin + b3 + b2);
    if (n == kModel)
        for (int i = 0; i < kModel; ++i) subsets[i] = i;
    else
        draw_subsets(n, n_hyp, subsets);
    MVO_HIP(hipMemcpyAsync(s->d_in, ctx->h_pin, b3 + b2 + bs, hipMemcpyHostToDevice, ctx->stream));
    const float* d_p3 = reinterpret_cast<const float*>(s->d_in);
    const float* d_p2 = reinterpret_cast<const float*>(s->d_in + b3);
    const int32_t* d_subsets = reinterpret_cast<const int32_t*>(s->d_in + b3 + b2);
    uint8_t* h_out = ctx->h_pin + o_out;
    uint8_t* h_counts = ctx->h_pin + o_counts;
    uint8_t* h_models = ctx->h_pin + o_models;
    uint8_t* h_mask = ctx->h_pin + o_mask;
    const TrackCamera cam{fx, fy, cx, cy};
    const float thr2 = (float)((double)reprojection_error * (double)reprojection_error);
    // How many hypotheses go first.  The sequential loop of RANSACPointSetRegistrator::run shortens itself as soon as a good model
    // turns up (three quarters of the pairs inliers: ~27 of the 100 iterations) -- a lone sequence evaluates all `iterations`
    // hypotheses at once anyway (they run side by side: the price of one), but where many sequences share the GPU a hypothesis is
    // 200 us of a CU: contexts in THROUGHPUT / SHARED mode evaluate the first 32, let the replay of the loop's bookkeeping say
    // whether the loop would have gone on, and only then launch the rest (one more round trip in that case).  The result is what
    // the sequential loop produces either way.  MVO_PNP_CHUNK: 0 = never, n > 0 = first n for every ctx (A/B).
    // A sequence whose loop ran long last time (few inliers among its pairs: the loop needs all its iterations) gets all hypotheses
    // at once again -- two chunks would only add a round trip and a second refinement; the inlier ratio of a sequence changes slowly
    // from frame to frame.  (Measured with 32 sequences: +8 % frames/s where the loop stops after ~27 iterations, -23 % where it
    // always needs 100 and the chunks were used blindly.)
    static const int env_chunk = std::getenv("MVO_PNP_CHUNK") ? std::atoi(std::getenv("MVO_PNP_CHUNK")) : -1;
    const bool chunked = n != kModel && n_hyp > 48 &&
                         (env_chunk >= 0 ? env_chunk > 0 : (ctx->ba_throughput_mode != 0 && s->last_loop_len > 0 && s->last_loop_len <= 28));
    const int first = chunked ? std::min(n_hyp, env_chunk > 0 ? env_chunk : 32) : n_hyp;
    if ((r = track_launch_pnp_hypotheses(ctx, d_p3, d_p2, n, d_subsets, first, cam, thr2, s->d_models, s->d_counts, s->d_masks,
                                         reinterpret_cast<double*>(h_models), reinterpret_cast<int32_t*>(h_counts))))
        return r;
    // The refinement kernel replays the sequential bookkeeping of RANSACPointSetRegistrator::run over the counts and
    // refines the model it selects; the host repeats the replay (its own libm) on the counts that come back with
    // the result and only launches again if it disagrees -- one host round trip per call.
    const int mode = n == kModel ? 1 : 0;
    const double dev_conf = g_pnp_replay_skew ? 0.5 : confidence;
    if ((r = track_launch_pnp_refine(ctx, d_p3, d_p2, s->d_masks, n, cam, s->d_models, s->d_counts, first, dev_conf,
                                     mode == 1 ? 0 : -1, mode, s->d_Mg, s->d_mg, h_mask, reinterpret_cast<double*>(h_out))))
        return r;
    auto fetch = [&]() -> int {
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        return MVO_OK;
    };
    if ((r = fetch())) return r;
    int evaluated = first;
    int best = -1;
    if (mode == 1) {  // "model_points == npoints": the kernel result is the answer and every pair an inlier
        best = 0;
        s->info[1] = 1;
    } else {
        // the loop's bookkeeping with its own bound (n_hyp), over the counts that exist so far
        const int32_t* cnts = reinterpret_cast<const int32_t*>(h_counts);
        int niters = n_hyp, max_good = 0, it = 0;
        for (;;) {
            for (; it < niters && it < evaluated; ++it) {
                const int good = cnts[it];
                if (good > std::max(max_good, kModel - 1)) {
                    max_good = good;
                    best = it;
                    niters = update_num_iters(confidence, (double)(n - good) / n, kModel, niters);
                }
            }
            if (it >= niters || evaluated >= n_hyp) break;
            // the sequential loop goes on behind the first chunk: the remaining hypotheses, then the refinement over all counts
            if ((r = track_launch_pnp_hypotheses(ctx, d_p3, d_p2, n, d_subsets + (size_t)kModel * evaluated, n_hyp - evaluated, cam, thr2,
                                                 s->d_models + 12 * (size_t)evaluated, s->d_counts + evaluated, s->d_masks + (size_t)evaluated * n,
                                                 reinterpret_cast<double*>(h_models) + 12 * (size_t)evaluated,
                                                 reinterpret_cast<int32_t*>(h_counts) + evaluated)))
                return r;
            if ((r = track_launch_pnp_refine(ctx, d_p3, d_p2, s->d_masks, n, cam, s->d_models, s->d_counts, n_hyp, dev_conf, -1, mode,
                                             s->d_Mg, s->d_mg, h_mask, reinterpret_cast<double*>(h_out))))
                return r;
            if ((r = fetch())) return r;
            evaluated = n_hyp;
        }
        s->info[1] = it;
        s->last_loop_len = it;
    }
    s->counts.assign(reinterpret_cast<int32_t*>(h_counts), reinterpret_cast<int32_t*>(h_counts) + evaluated);
    s->models.resize((size_t)evaluated * 12);
    std::memcpy(s->models.data(), h_models, (size_t)evaluated * 96);
    s->info[5] = evaluated;
    s->info[0] = best;
    double out[12];
    std::memcpy(out, h_out, sizeof(out));
    if (best >= 0 && (int)out[10] != best) {  // the device's replay chose differently: refine the right hypothesis
        s->info[5] = -evaluated;              // (visible to tests through mvo_debug_get_pnp)
        if ((r = track_launch_pnp_refine(ctx, d_p3, d_p2, s->d_masks, n, cam, s->d_models, s->d_counts, evaluated,
                                         confidence, best, mode, s->d_Mg, s->d_mg, h_mask, reinterpret_cast<double*>(h_out))))
            return r;
        if ((r = fetch())) return r;
        std::memcpy(out, h_out, sizeof(out));
    }
    if (ctx->prof) mvo_prof_collect(ctx);
    if (best < 0) return MVO_OK;  // no model with at least 5 inliers
    for (int k = 0; k < 3; ++k) {
        rvec[k] = out[k];
        tvec[k] = out[3 + k];
    }
    s->info[2] = (int)out[7];
    s->info[3] = (int)out[8];
    s->info[4] = (int)out[9];
    int cnt = 0;
    if (mode == 1) {
        for (int i = 0; i < n; ++i) inliers[cnt++] = i;
    } else {
        for (int i = 0; i < n; ++i)
            if (h_mask[i]) inliers[cnt++] = i;
    }
    *n_inliers = cnt;
    *found = 1;
    return MVO_OK;
}

// ---------------------------------------------------------------------------------------------- keyframe row
int mvo_triangulate_points(mvo_ctx* ctx, const float* kp_prev, const float* kp_curr, int n, double fx, double fy, double cx,
                           double cy, const double* R, const double* t, float* pts3d_in_prev, float* pts3d_in_curr) {
    if (!ctx || n < 0 || !R || !t || (n && (!kp_prev || !kp_curr || (!pts3d_in_prev && !pts3d_in_curr))))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (n == 0) return MVO_OK;
    MVO_HIP(hipSetDevice(ctx->device));
    mvo_track_state* s = state(ctx);
    if (n > s->cap_tri) {
        free_dev(s->d_tri_in);
        free_dev(s->d_tri_out);
        s->cap_tri = 0;
        const int cap = std::max(4096, n + n / 2);
        MVO_HIP(hipMalloc((void**)&s->d_tri_in, (size_t)cap * 4 * sizeof(float)));
        MVO_HIP(hipMalloc((void**)&s->d_tri_out, (size_t)cap * 6 * sizeof(float)));
        s->cap_tri = cap;
    }
    int r = mvo_ensure_pinned(ctx, (size_t)n * 24);
    if (r) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(ctx->h_pin, kp_prev, (size_t)n * 8);
    std::memcpy(ctx->h_pin + (size_t)n * 8, kp_curr, (size_t)n * 8);
    MVO_HIP(hipMemcpyAsync(s->d_tri_in, ctx->h_pin, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    const TrackCamera cam{fx, fy, cx, cy};
    if ((r = track_launch_triangulate(ctx, s->d_tri_in, s->d_tri_in + 2 * (size_t)n, n, cam, R, t, s->d_tri_out,
                                      s->d_tri_out + 3 * (size_t)n)))
        return r;
    MVO_HIP(hipMemcpyAsync(ctx->h_pin, s->d_tri_out, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) mvo_prof_collect(ctx);
    if (pts3d_in_prev) std::memcpy(pts3d_in_prev, ctx->h_pin, (size_t)n * 12);
    if (pts3d_in_curr) std::memcpy(pts3d_in_curr, ctx->h_pin + (size_t)n * 12, (size_t)n * 12);
    return MVO_OK;
}

// geometry::helperFindInlierMatchesByEpipolarCons = the inlier mask of cv::findEssentialMat(RANSAC)
int mvo_find_essential_inliers(mvo_ctx* ctx, const float* kp_prev, const float* kp_curr, int n, double fx, double fy,
                               double cx, double cy, double prob, double threshold, int32_t* inliers, int cap,
                               int* n_inliers) {
    if (!ctx || n < 0 || !n_inliers || cap < 0 || (n && (!kp_prev || !kp_curr)) || (cap && !inliers))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "bad arguments", hipSuccess);
    if (!(prob > 0 && prob < 1))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "mvo_find_essential_inliers: prob must be in (0, 1)", hipSuccess);
    *n_inliers = 0;
    mvo_track_state* s = state(ctx);
    s->em_counts.clear();
    s->em_info[0] = s->em_info[1] = -1;
    s->em_info[2] = s->em_info[3] = s->em_info[4] = 0;
    constexpr int kModel = 5, kMaxIters = 1000;  // createRANSACPointSetRegistrator(cb, 5, threshold, prob) -> maxIters 1000
    if (n < kModel) return MVO_OK;
    if (cap < n) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "mvo_find_essential_inliers: inlier buffer smaller than n", hipSuccess);
    MVO_HIP(hipSetDevice(ctx->device));
    if (n > s->cap_em) {
        free_dev(s->d_emq);
        free_dev(s->d_em_mask);
        s->cap_em = 0;
        const int c = std::max(4096, n + n / 2);
        MVO_HIP(hipMalloc((void**)&s->d_emq, (size_t)c * 4 * sizeof(double)));
        MVO_HIP(hipMalloc((void**)&s->d_em_mask, (size_t)c));
        s->cap_em = c;
    }
    if (!s->d_em_subsets) {
        MVO_HIP(hipMalloc((void**)&s->d_em_subsets, (size_t)kMaxIters * 5 * sizeof(int32_t)));
        MVO_HIP(hipMalloc((void**)&s->d_em_nm, (size_t)kMaxIters * sizeof(int32_t)));
        MVO_HIP(hipMalloc((void**)&s->d_em_counts, (size_t)kMaxIters * 10 * sizeof(int32_t)));
        MVO_HIP(hipMalloc((void**)&s->d_em_E, (size_t)kMaxIters * 90 * sizeof(double)));
    }
    // findEssentialMat(points1, points2, focal, pp, ...): K = [focal 0 pp.x; 0 focal pp.y], pp a cv::Point2f built from
    // K(0,2), K(1,2) (epipolar_geometry.cpp:27-28); points to double, (p - c) / f; threshold /= (fx + fy) / 2
    const double focal = (fx + fy) / 2;
    const double pcx = (double)(float)cx, pcy = (double)(float)cy;
    const size_t bq = (size_t)n * 4 * sizeof(double), bs = (size_t)kMaxIters * 5 * sizeof(int32_t);
    int r = mvo_ensure_pinned(ctx, std::max(bq + bs, (size_t)kMaxIters * 40 + (size_t)n + 64));
    if (r) return r;
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    double* q = reinterpret_cast<double*>(ctx->h_pin);
    for (int i = 0; i < n; ++i) {
        q[2 * i] = ((double)kp_prev[2 * i] - pcx) / focal;
        q[2 * i + 1] = ((double)kp_prev[2 * i + 1] - pcy) / focal;
        q[2 * (size_t)n + 2 * i] = ((double)kp_curr[2 * i] - pcx) / focal;
        q[2 * (size_t)n + 2 * i + 1] = ((double)kp_curr[2 * i + 1] - pcy) / focal;
    }
    int32_t* subsets = reinterpret_cast<int32_t*>(ctx->h_pin + bq);
    const int total = n == kModel ? 1 : kMaxIters;
    if (n == kModel)
        for (int i = 0; i < kModel; ++i) subsets[i] = i;
    else
        draw_subsets(n, kMaxIters, subsets);
    MVO_HIP(hipMemcpyAsync(s->d_emq, q, bq, hipMemcpyHostToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(s->d_em_subsets, subsets, (size_t)total * 5 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const double thr = threshold / ((focal + focal) / 2);
    const float thr2 = (float)(thr * thr);
    const double* d_q1 = s->d_emq;
    const double* d_q2 = s->d_emq + 2 * (size_t)n;
    // The hypotheses are evaluated in growing chunks; after each chunk the sequential bookkeeping of
    // RANSACPointSetRegistrator::run is advanced over the new counts until the loop's own stopping rule is met.
    const int chunk_end[2] = {256, kMaxIters};  // a chunk costs ~0.25 ms whatever its size (one wave per hypothesis)
    int evaluated = 0, niters = total, max_good = 0, it = 0, best_it = -1, best_m = -1;
    bool first_wait = true;
    for (int c = 0; c < 2 && it < niters; ++c) {
        const int end = std::min(chunk_end[c], total);
        if (end <= evaluated) continue;
        if ((r = track_launch_em_hypotheses(ctx, d_q1, d_q2, n, s->d_em_subsets + 5 * (size_t)evaluated, end - evaluated, thr2,
                                            s->d_em_E + 90 * (size_t)evaluated, s->d_em_nm + evaluated,
                                            s->d_em_counts + 10 * (size_t)evaluated)))
            return r;
        if (first_wait) {  // the staging area still holds the uploads of this call
            MVO_HIP(hipStreamSynchronize(ctx->stream));
            first_wait = false;
        }
        int32_t* h_counts = reinterpret_cast<int32_t*>(ctx->h_pin);
        MVO_HIP(hipMemcpyAsync(h_counts, s->d_em_counts + 10 * (size_t)evaluated, (size_t)(end - evaluated) * 40,
                               hipMemcpyDeviceToHost, ctx->stream));
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        s->em_counts.insert(s->em_counts.end(), h_counts, h_counts + (size_t)(end - evaluated) * 10);
        evaluated = end;
        for (; it < niters && it < evaluated; ++it)
            for (int m = 0; m < 10; ++m) {
                const int good = n == kModel && s->em_counts[(size_t)it * 10 + m] >= 0 ? n : s->em_counts[(size_t)it * 10 + m];
                if (good < 0) break;
                if (good > std::max(max_good, kModel - 1)) {
                    max_good = good;
                    best_it = it;
                    best_m = m;
                    niters = update_num_iters(prob, (double)(n - good) / n, kModel, niters);
                }
                if (n == kModel) break;  // count == modelPoints: the first model is the answer
            }
    }
    s->em_info[0] = best_it;
    s->em_info[1] = best_m;
    s->em_info[2] = it;
    s->em_info[3] = evaluated;
    if (best_it < 0) {
        if (ctx->prof) mvo_prof_collect(ctx);
        return MVO_OK;
    }
    int cnt = 0;
    if (n == kModel) {
        for (int i = 0; i < n; ++i) inliers[cnt++] = i;
    } else {
        if ((r = track_launch_em_mask(ctx, d_q1, d_q2, n, s->d_em_E + 90 * (size_t)best_it + 9 * (size_t)best_m, thr2,
                                      s->d_em_mask)))
            return r;
        MVO_HIP(hipMemcpyAsync(ctx->h_pin, s->d_em_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n; ++i)
            if (ctx->h_pin[i]) inliers[cnt++] = i;
    }
    if (ctx->prof) mvo_prof_collect(ctx);
    *n_inliers = cnt;
    return MVO_OK;
}

int mvo_debug_get_essential(mvo_ctx* ctx, int32_t* counts, int cap_iters, int32_t* info) {
    if (!ctx || !ctx->track) return mvo_set_err(ctx, MVO_ERR_STATE, "no essential-matrix call on this ctx yet", hipSuccess);
    const mvo_track_state* s = ctx->track;
    const int iters = (int)(s->em_counts.size() / 10);
    if (info) std::memcpy(info, s->em_info, sizeof(s->em_info));
    if (iters > cap_iters) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "mvo_debug_get_essential: buffer too small", hipSuccess);
    if (counts && iters) std::memcpy(counts, s->em_counts.data(), s->em_counts.size() * 4);
    return iters;
}

int mvo_retain_good_triangulation(const float* pts3d_in_curr, int n, const double* T_w_c_curr, const double* T_w_c_ref,
                                  double min_triang_angle, double max_ratio_to_median, int32_t* keep, int* n_keep,
                                  double* angles) {
    if (n < 0 || !n_keep || !T_w_c_curr || !T_w_c_ref || (n && (!pts3d_in_curr || !keep))) return MVO_ERR_INVALID;
    *n_keep = 0;
    if (n == 0) return MVO_OK;  // vo.cpp:198-199
    std::vector<double> ang(n);
    for (int i = 0; i < n; ++i) {
        const float* pc = pts3d_in_curr + 3 * (size_t)i;
        double to_curr[3], to_ref[3];
        for (int r = 0; r < 3; ++r) {  // preTranslatePoint3f(p_in_curr, T_w_c) -> float, then the two rays
            const double* row = T_w_c_curr + 4 * r;
            double acc = 0;
            acc += row[0] * (double)pc[0];
            acc += row[1] * (double)pc[1];
            acc += row[2] * (double)pc[2];
            acc += row[3] * 1.0;
            const double pw = (double)(float)acc;
            to_curr[r] = T_w_c_curr[4 * r + 3] - pw;
            to_ref[r] = T_w_c_ref[4 * r + 3] - pw;
        }
        double dot = 0, n1 = 0, n2 = 0;
        for (int r = 0; r < 3; ++r) dot += to_curr[r] * to_ref[r];
        for (int r = 0; r < 3; ++r) n1 = n1 + to_curr[r] * to_curr[r];
        for (int r = 0; r < 3; ++r) n2 = n2 + to_ref[r] * to_ref[r];
        ang[i] = std::acos(dot / (std::sqrt(n1) * std::sqrt(n2))) / 3.1415926 * 180.0;  // vo.cpp:210
    }
    std::vector<double> sorted(ang);
    std::sort(sorted.begin(), sorted.end());
    const double median = sorted[n / 2];
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        if (angles) angles[i] = ang[i];
        if (ang[i] < min_triang_angle || ang[i] / median > max_ratio_to_median) continue;  // vo.cpp:233-235
        keep[cnt++] = i;
    }
    *n_keep = cnt;
    return MVO_OK;
}

int mvo_rodrigues(const double* rvec, double* R) {
    if (!rvec || !R) return MVO_ERR_INVALID;
    const double x = rvec[0], y = rvec[1], z = rvec[2];
    const double theta = std::sqrt(x * x + y * y + z * z);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return MVO_OK;
    }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
    const double rx = x * it, ry = y * it, rz = z * it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * r_x[k];
    return MVO_OK;
}

int mvo_invert_pose(const double* T, double* T_inv) {
    if (!T || !T_inv) return MVO_ERR_INVALID;
    double tmp[16];
    if (!invert_pose_lu(T, tmp, 4)) return MVO_ERR_INVALID;
    std::memcpy(T_inv, tmp, sizeof(tmp));
    return MVO_OK;
}

int mvo_debug_get_pnp(mvo_ctx* ctx, double* models, int32_t* counts, int cap, int32_t* info) {
    if (!ctx || !ctx->track) return mvo_set_err(ctx, MVO_ERR_STATE, "no PnP solve on this ctx yet", hipSuccess);
    const mvo_track_state* s = ctx->track;
    const int h = (int)s->counts.size();
    if (info) std::memcpy(info, s->info, sizeof(s->info));
    if (h > cap) return mvo_set_err(ctx, MVO_ERR_CAPACITY, "mvo_debug_get_pnp: buffers too small", hipSuccess);
    if (models && h) std::memcpy(models, s->models.data(), (size_t)h * 96);
    if (counts && h) std::memcpy(counts, s->counts.data(), (size_t)h * 4);
    return h;
}
}
