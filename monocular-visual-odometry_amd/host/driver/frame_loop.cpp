// host/driver/frame_loop.cpp -- the per-sequence frame loop in native code: what run_vo.cpp:117-154 does around
// VisualOdometry::addFrame, reduced to the hot-path calls, one call of frame_loop_run per host thread.  bench.py
// drives one of these per sequence shard so that the measurement is not serialised by the Python interpreter lock
// (every per-frame C-ABI call made from Python pays ~20-40 us of interpreter time under the GIL).
// Built into host/driver/libmvo_frame_loop.so, links only libmvo_hip.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "mvo_hip.h"

extern "C" {

struct frame_loop_cfg {
    mvo_ctx* ctx;
    const void* const* d_frames;  // n_frames device pointers (BGR images resident in HBM)
    int32_t n_frames, width, height, stride, channels, max_kp;
    mvo_ba_handle* ba;            // resident BA window (mvo_ba_prepare)
    int32_t n_poses;
    // tracking rows (optional, track != 0)
    int32_t track, keyframe_every;
    mvo_map* map;
    int32_t n_map;
    const double* T_w_c;          // 16
    double K4[4];
    const float *pts3d, *pts2d;   // PnP pairs
    int32_t n_pairs;
    const float *kf_ref, *kf_cur; // keyframe matches
    int32_t kf_n;
    const double *kf_T_curr_to_prev, *kf_T_w_cur, *kf_T_w_ref;  // 16 each
};

struct frame_loop_state {  // persists between calls (warm-up, timed region)
    const void* prev_desc;
    int32_t prev_n;
    int32_t frame_no;
    int32_t n_kp, n_match, n_inliers, n_tri, ba_trials, ba_iterations;
};

// Advances the shard by `steps` frames; traj gets steps x 12 doubles (x y z, then R column-major, vo_io.cpp:58-75).
// Returns MVO_OK or the first failing status (mvo_last_error(ctx) has the text).
int frame_loop_run(const frame_loop_cfg* c, frame_loop_state* st, int steps, double* traj) {
    std::vector<mvo_keypoint> kps((size_t)c->max_kp + 16);
    std::vector<mvo_dmatch> matches((size_t)c->max_kp + 16);
    std::vector<double> poses((size_t)c->n_poses * 16);
    std::vector<int32_t> idx, inl;
    std::vector<float> px, tri;
    if (c->track) {
        idx.resize(c->n_map > 0 ? c->n_map : 1);
        px.resize(2 * idx.size());
        inl.resize((size_t)(c->n_pairs > c->kf_n ? c->n_pairs : c->kf_n) + 1);
        tri.resize(3 * (size_t)c->kf_n + 3);
    }
    for (int s = 0; s < steps; ++s) {
        const void* img = c->d_frames[st->frame_no % c->n_frames];
        int n = 0, r;
        if ((r = mvo_calc_keypoints_dev(c->ctx, img, c->width, c->height, c->stride, c->channels, kps.data(), (int)kps.size(), &n)))
            return r;
        const void* d_desc = nullptr;
        if ((r = mvo_calc_descriptors_dev(c->ctx, kps.data(), &n, nullptr, &d_desc))) return r;
        if (st->prev_desc && n && st->prev_n) {
            int nm = 0;
            if ((r = mvo_match_features_dev(c->ctx, st->prev_desc, st->prev_n, d_desc, n, 2, 2.0, 0.8, matches.data(),
                                            (int)matches.size(), &nm)))
                return r;
            st->n_match = nm;
        }
        st->prev_desc = d_desc;
        st->prev_n = n;
        st->n_kp = n;
        if (c->track) {
            int nv = 0, nm = 0, n_inl = 0, found = 0;
            const void* d_map_desc = nullptr;
            if ((r = mvo_map_points_in_view(c->ctx, c->map, c->T_w_c, c->K4[0], c->K4[1], c->K4[2], c->K4[3], c->width, c->height,
                                            idx.data(), px.data(), (int)idx.size(), &nv, &d_map_desc)))
                return r;
            if (nv && n) {
                if ((int)matches.size() < nv) matches.resize(nv);
                if ((r = mvo_match_features_dev(c->ctx, d_map_desc, nv, d_desc, n, 1, 2.0, 1.0, matches.data(), (int)matches.size(), &nm)))
                    return r;
            }
            double rvec[3], tvec[3];
            if ((r = mvo_solve_pnp_ransac(c->ctx, c->pts3d, c->pts2d, c->n_pairs, c->K4[0], c->K4[1], c->K4[2], c->K4[3], 100, 2.0f,
                                          0.999, rvec, tvec, inl.data(), (int)inl.size(), &n_inl, &found)))
                return r;
            st->n_inliers = n_inl;
        }
        mvo_ba_stats bs;
        if ((r = mvo_ba_solve_resident(c->ctx, c->ba))) return r;
        if ((r = mvo_ba_fetch(c->ctx, c->ba, poses.data(), nullptr, &bs))) return r;
        st->ba_trials += bs.trials;
        st->ba_iterations += bs.iterations;
        if (c->track && c->keyframe_every > 0 && st->frame_no % c->keyframe_every == 0) {
            int n_inl = 0, n_keep = 0;
            if ((r = mvo_find_essential_inliers(c->ctx, c->kf_ref, c->kf_cur, c->kf_n, c->K4[0], c->K4[1], c->K4[2], c->K4[3], 0.999, 1.0,
                                                inl.data(), (int)inl.size(), &n_inl)))
                return r;
            std::vector<float> a(2 * (size_t)n_inl), b(2 * (size_t)n_inl);
            for (int i = 0; i < n_inl; ++i) {
                std::memcpy(&a[2 * i], c->kf_ref + 2 * (size_t)inl[i], 8);
                std::memcpy(&b[2 * i], c->kf_cur + 2 * (size_t)inl[i], 8);
            }
            double R[9], t[3];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) R[3 * i + j] = c->kf_T_curr_to_prev[4 * i + j];
                t[i] = c->kf_T_curr_to_prev[4 * i + 3];
            }
            if ((r = mvo_triangulate_points(c->ctx, a.data(), b.data(), n_inl, c->K4[0], c->K4[1], c->K4[2], c->K4[3], R, t, nullptr,
                                            tri.data())))
                return r;
            std::vector<int32_t> keep(n_inl > 0 ? n_inl : 1);
            if ((r = mvo_retain_good_triangulation(tri.data(), n_inl, c->kf_T_w_cur, c->kf_T_w_ref, 1.0, 20.0, keep.data(), &n_keep, nullptr)))
                return r;
            st->n_tri = n_keep;
        }
        // trajectory row of the newest frame of the window: x y z, then R column by column
        double* row = traj + 12 * (size_t)s;
        const double* T = poses.data();
        row[0] = T[3];
        row[1] = T[7];
        row[2] = T[11];
        for (int col = 0; col < 3; ++col)
            for (int rr = 0; rr < 3; ++rr) row[3 + 3 * col + rr] = T[4 * rr + col];
        st->frame_no++;
    }
    return MVO_OK;
}
}
