// host/driver/frame_loop.cpp -- the per-sequence frame loop in native code: what run_vo.cpp:117-154 does around
// VisualOdometry::addFrame, reduced to the hot-path calls, one loop object per sequence shard, driven by one host
// thread each (bench.py).  Per frame:
//   extract (image resident in HBM) -> descriptors stay in HBM -> match against the previous frame's descriptors ->
//   [tracking rows] -> bundle adjustment of the sliding window.
// The bundle adjustment is done the way the reference does it on EVERY frame (src/vo/vo.cpp:384-478 ->
// src/optimization/g2o_ba.cpp:172-317): the window is marshalled from live Frame / MapPoint objects into pointer
// lists (vo::buildBundleAdjustmentWindow), flattened, planned, uploaded, solved and written back through the
// pointers.  A shard holds a pool of distinct windows (different scenes, observations and initial errors) and
// rotates through them, restoring a window's initial state before it is handed to the optimiser, so that no frame
// re-solves what the previous one left behind.
// With `pipeline` the loop overlaps the extraction + matching of frame i + 1 (ctx) with the bundle adjustment of
// frame i (ctx_ba): both are independent of each other, the results are the same as in the serial order.
// Built into host/driver/libmvo_frame_loop.so; links only libmvo_hip.so.
#include <cstdint>
#include <chrono>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "my_slam/vo/ba_window.h"
#include "mvo_hip.h"

using namespace my_slam;

extern "C" {

struct frame_loop_cfg {
    mvo_ctx* ctx;                 // extraction + matching (+ tracking rows)
    mvo_ctx* ctx_ba;              // bundle adjustment (may equal ctx; a second ctx is needed for `pipeline`)
    const void* const* d_frames;  // n_frames device pointers (BGR images resident in HBM)
    int32_t n_frames, width, height, stride, channels, max_kp;
    int32_t ba_mode;              // 0 none, 1 window rebuilt per frame (the metric), 2 one resident window re-solved
    int32_t pipeline;             // 1: extract frame i + 1 while the window of frame i is being solved
    int32_t fix_points;           // is_ba_fix_map_points (config.yaml:123)
    double K4[4];                 // fx fy cx cy
    // tracking rows (optional, track != 0)
    int32_t track, keyframe_every;
    mvo_map* map;
    int32_t n_map;
    const double* T_w_c;          // 16
    const float *pts3d, *pts2d;   // PnP pairs
    int32_t n_pairs;
    const float *kf_ref, *kf_cur; // keyframe matches
    int32_t kf_n;
    const double *kf_T_curr_to_prev, *kf_T_w_cur, *kf_T_w_ref;  // 16 each
    int32_t ba_throughput;        // 1: mvo_ba_set_mode(ctx_ba, MVO_BA_MODE_THROUGHPUT) (many sequences per GPU); 2: MVO_BA_MODE_SHARED
                                  // (many sequences, and stages other than the bundle adjustment dominate a frame)
    const void* const* h_frames;  // optional: the same frames in (pinned) host memory; when set, every frame is handed to
                                  // mvo_calc_keypoints as a HOST image like run_vo.cpp:114 does (H2D inside the loop)
    int32_t chain;                // 1: the newest frame of every window takes its pose and its map-point connections from
                                  // solvePnPRansac on its own 3D-2D pairs (vo.cpp:304-357) before the window is marshalled
};

struct frame_loop_state {  // progress counters, read back by the caller
    int32_t frame_no;
    int32_t n_kp, n_match, n_inliers, n_tri;
    int64_t ba_trials, ba_iterations, ba_solves, ba_edges;
    // host wall-clock per stage, nanoseconds, accumulated: extract+match call, window restore (bench scaffolding),
    // window marshalling (buildBundleAdjustmentWindow), job.begin (flatten + plan + upload + submit), job.end (wait for
    // the solve + scatter); in pipeline mode the extraction of frame i+1 sits between begin and end
    int64_t ns_extract, ns_restore, ns_build, ns_begin, ns_end;
    int64_t ba_failed_solves, ba_stale_steps;  // LM trials that ended at a failed factorisation / whose stale step was applied
};

}  // extern "C"

namespace {

using Clock = std::chrono::steady_clock;
inline int64_t ns_since(Clock::time_point t0) {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count();
}

// One BA window as the reference holds it: frames_buff_ (vo.h:64) with keypoints_, inliers_to_mappt_connections_ and
// T_w_c_, a Map with the MapPoints, plus the initial state to restore before every solve.
struct Window {
    std::deque<vo::Frame::Ptr> frames_buff;  // oldest first; frames_buff[0] is the extra frame vo.cpp:395 leaves out
    vo::Map::Ptr map;
    std::vector<double> poses0;              // F x 16
    std::vector<float> points0;              // L x 3
    std::vector<int> point_ids;              // map point id of landmark l
    int F = 0, L = 0, E = 0;
    mvo_ba_handle* resident = nullptr;       // ba_mode 2
    // chained mode: the 3D-2D pairs of the newest frame as matching would deliver them (a quarter of them wrong), the
    // keypoint index of every pair
    std::vector<float> pnp3d, pnp2d;
    std::vector<int> pnp_kpt;
    std::vector<int32_t> pnp_inl;
    std::vector<vo::PtConn> conn0;            // connection of keypoint k of the newest frame when every pair is kept
};

struct Loop {
    frame_loop_cfg c{};
    frame_loop_state st{};
    std::vector<std::unique_ptr<Window>> windows;
    cv::Mat K, info;
    const void* prev_desc = nullptr;
    int prev_n = 0;
    // extraction results of the frame that was processed ahead (pipeline mode)
    bool have_next = false;
    const void* next_desc = nullptr;
    int next_n = 0, next_match = 0;
    std::vector<mvo_keypoint> kps;
    std::vector<mvo_dmatch> matches;
    std::vector<int32_t> idx, inl;
    std::vector<float> px, tri;
    optimization::BundleAdjustmentJob job;
    vo::BaWindow pending;  // pointer lists of the solve in flight (must outlive it)
    Window* pending_w = nullptr;
    // parity capture (frame_loop_capture): keypoints, descriptors and matches of the last extracted frame, copied out of the
    // buffers the loop works on -- what bench.py hands to the oracle after the timed region
    bool capture = false;
    int cap_frame = -1, cap_n = 0, cap_nm = 0;
    std::vector<mvo_keypoint> cap_kps;
    std::vector<uint8_t> cap_desc;
    std::vector<mvo_dmatch> cap_matches;
};

void restore(Window& w) {
    // frames_buff[1..F] are optimised (newest = back); frames_buff[0] only keeps the buffer one longer than the window
    for (int i = 0; i < w.F; ++i) {
        cv::Mat& T = w.frames_buff[(size_t)i + 1]->T_w_c_;
        for (int k = 0; k < 16; ++k) T.at<double>(k / 4, k % 4) = w.poses0[16 * (size_t)(w.F - 1 - i) + k];
    }
    for (int l = 0; l < w.L; ++l) {
        cv::Point3f& p = w.map->map_points_[w.point_ids[l]]->pos_;
        p.x = w.points0[3 * (size_t)l];
        p.y = w.points0[3 * (size_t)l + 1];
        p.z = w.points0[3 * (size_t)l + 2];
    }
}

int extract_and_match(Loop& L, int frame_no, const void** d_desc, int* n_out, int* n_match) {
    const frame_loop_cfg& c = L.c;
    int n = 0, r;
    if (c.h_frames) {  // host image in, as the reference's loop hands it over (imread -> createFrame -> calcKeyPoints)
        if ((r = mvo_calc_keypoints(c.ctx, (const uint8_t*)c.h_frames[frame_no % c.n_frames], c.width, c.height, c.stride, c.channels,
                                    L.kps.data(), (int)L.kps.size(), &n)))
            return r;
    } else if ((r = mvo_calc_keypoints_dev(c.ctx, c.d_frames[frame_no % c.n_frames], c.width, c.height, c.stride, c.channels, L.kps.data(),
                                           (int)L.kps.size(), &n))) {
        return r;
    }
    if (L.capture) L.cap_desc.resize(32 * L.kps.size());
    if ((r = mvo_calc_descriptors_dev(c.ctx, L.kps.data(), &n, L.capture ? L.cap_desc.data() : nullptr, d_desc))) return r;
    *n_match = 0;
    if (L.prev_desc && n && L.prev_n) {
        int nm = 0;
        if ((r = mvo_match_features_dev(c.ctx, L.prev_desc, L.prev_n, *d_desc, n, 2, 2.0, 0.8, L.matches.data(), (int)L.matches.size(), &nm)))
            return r;
        *n_match = nm;
    }
    if (L.capture) {
        L.cap_frame = frame_no;
        L.cap_n = n;
        L.cap_nm = *n_match;
        L.cap_kps.assign(L.kps.begin(), L.kps.begin() + n);
        L.cap_matches.assign(L.matches.begin(), L.matches.begin() + *n_match);
    }
    L.prev_desc = *d_desc;
    L.prev_n = n;
    *n_out = n;
    return MVO_OK;
}

}  // namespace

extern "C" {

void* frame_loop_create(const frame_loop_cfg* cfg) {
    Loop* L = new Loop();
    L->c = *cfg;
    if (!L->c.ctx_ba) L->c.ctx_ba = L->c.ctx;
    // (both contexts of the sequence: the mode also tells the detection who restores the candidate order)
    const int ba_mode = cfg->ba_throughput == 2 ? MVO_BA_MODE_SHARED : (cfg->ba_throughput ? MVO_BA_MODE_THROUGHPUT : MVO_BA_MODE_LATENCY);
    (void)mvo_ba_set_mode(L->c.ctx_ba, ba_mode);
    (void)mvo_ba_set_mode(L->c.ctx, ba_mode);
    L->kps.resize((size_t)cfg->max_kp + 16);
    L->matches.resize((size_t)cfg->max_kp + 16);
    if (cfg->track) {
        L->idx.resize(cfg->n_map > 0 ? cfg->n_map : 1);
        L->px.resize(2 * L->idx.size());
        L->inl.resize((size_t)(cfg->n_pairs > cfg->kf_n ? cfg->n_pairs : cfg->kf_n) + 1);
        L->tri.resize(3 * (size_t)cfg->kf_n + 3);
    }
    L->K = cv::Mat::eye(3, 3, CV_64FC1);
    L->K.at<double>(0, 0) = cfg->K4[0];
    L->K.at<double>(1, 1) = cfg->K4[1];
    L->K.at<double>(0, 2) = cfg->K4[2];
    L->K.at<double>(1, 2) = cfg->K4[3];
    L->info = cv::Mat::eye(2, 2, CV_64FC1);  // information_matrix (config.yaml:122)
    return L;
}

// Adds one window to the pool: pose i of `poses0` is the i-th NEWEST frame (the order bundleAdjustment receives them,
// vo.cpp:414-421), edges (pose, landmark, measured pixel) as in mvo_ba_problem.  Builds the Frame / MapPoint objects.
int frame_loop_add_window(void* h, int F, int Lm, int E, const double* poses0, const double* points0, const int32_t* edge_pose,
                          const int32_t* edge_point, const double* edge_uv) {
    Loop* L = static_cast<Loop*>(h);
    std::unique_ptr<Window> w(new Window());
    w->F = F;
    w->L = Lm;
    w->E = E;
    w->poses0.assign(poses0, poses0 + 16 * (size_t)F);
    w->points0.resize(3 * (size_t)Lm);
    for (size_t i = 0; i < 3 * (size_t)Lm; ++i) w->points0[i] = (float)points0[i];
    w->map.reset(new vo::Map());
    cv::Mat nodesc;
    for (int l = 0; l < Lm; ++l) {
        vo::MapPoint::Ptr mp(new vo::MapPoint(cv::Point3f(w->points0[3 * (size_t)l], w->points0[3 * (size_t)l + 1], w->points0[3 * (size_t)l + 2]),
                                              nodesc, nodesc));
        w->point_ids.push_back(mp->id_);
        w->map->insertMapPoint(mp);
    }
    for (int i = 0; i <= F; ++i) w->frames_buff.push_back(vo::Frame::createFrame(cv::Mat()));
    // pose index i (newest first) = frames_buff[F - i]; every observation becomes a keypoint + a map-point connection
    for (int e = 0; e < E; ++e) {
        vo::Frame& f = *w->frames_buff[(size_t)(F - edge_pose[e])];
        const int kpt_idx = (int)f.keypoints_.size();
        f.keypoints_.push_back(cv::KeyPoint((float)edge_uv[2 * (size_t)e], (float)edge_uv[2 * (size_t)e + 1], 31));
        vo::PtConn conn;
        conn.pt_ref_idx = -1;
        conn.pt_map_idx = w->point_ids[edge_point[e]];
        f.inliers_to_mappt_connections_[kpt_idx] = conn;
    }
    if (L->c.chain) {
        // the newest frame (pose index 0 = frames_buff[F]): its observations become the pairs PnP sees; every fourth pair gets
        // a wrong pixel (a mismatch), deterministically
        vo::Frame& f = *w->frames_buff[(size_t)F];
        unsigned lcg = 12345u + (unsigned)Lm;
        for (int e = 0, k = 0; e < E; ++e) {
            if (edge_pose[e] != 0) continue;
            const int l = edge_point[e];
            float u = (float)edge_uv[2 * (size_t)e], v = (float)edge_uv[2 * (size_t)e + 1];
            if (k % 4 == 3) {
                lcg = lcg * 1664525u + 1013904223u;
                u += (float)((int)(lcg >> 16) % 61 - 30);
                lcg = lcg * 1664525u + 1013904223u;
                v += (float)((int)(lcg >> 16) % 61 - 30);
            }
            w->pnp3d.insert(w->pnp3d.end(), {w->points0[3 * (size_t)l], w->points0[3 * (size_t)l + 1], w->points0[3 * (size_t)l + 2]});
            w->pnp2d.insert(w->pnp2d.end(), {u, v});
            w->pnp_kpt.push_back(k);
            ++k;
        }
        w->pnp_inl.resize(w->pnp_kpt.size() + 1);
        w->conn0.resize(f.keypoints_.size());
        for (auto& kv : f.inliers_to_mappt_connections_) w->conn0[(size_t)kv.first] = kv.second;
    }
    restore(*w);
    if (L->c.ba_mode == 2 && L->windows.empty()) {  // the resident variant keeps the first window in HBM
        hot_path_ctx_binding() = L->c.ctx_ba;
        vo::BaWindow bw = vo::buildBundleAdjustmentWindow(w->frames_buff, w->map, F);
        optimization::FlatBundle fb;
        fb.flatten(bw.v_pts_2d, bw.v_pts_2d_to_3d_idx, L->K, bw.um_pts_3d_in_prev_frames, bw.v_camera_poses, L->info, L->c.fix_points != 0);
        int r = mvo_ba_prepare(L->c.ctx_ba, &fb.pr, &w->resident);
        hot_path_ctx_binding() = nullptr;
        if (r) return r;
    }
    L->windows.push_back(std::move(w));
    return MVO_OK;
}

// Advances the shard by `steps` frames; traj gets steps x 12 doubles (x y z, then R column-major, vo_io.cpp:58-75):
// the refined pose of the newest frame of the window solved in that step.  Returns MVO_OK or the first failing status.
int frame_loop_run(void* h, int steps, double* traj) {
    Loop& L = *static_cast<Loop*>(h);
    const frame_loop_cfg& c = L.c;
    frame_loop_state& st = L.st;
    hot_path_ctx_binding() = c.ctx_ba;  // the adapters (optimization::bundleAdjustment) run on the shard's BA ctx
    int status = MVO_OK;
    try {
        for (int s = 0; s < steps && status == MVO_OK; ++s) {
            const void* d_desc = nullptr;
            int n = 0, nm = 0, r = MVO_OK;
            // ---- features of this frame (already there when the previous step ran ahead)
            if (L.have_next) {
                d_desc = L.next_desc;
                n = L.next_n;
                nm = L.next_match;
                L.have_next = false;
            } else {
                const auto t0 = Clock::now();
                r = extract_and_match(L, st.frame_no, &d_desc, &n, &nm);
                st.ns_extract += ns_since(t0);
                if (r) {
                    status = r;
                    break;
                }
            }
            st.n_kp = n;
            st.n_match = nm;
            if (c.track) {
                int nv = 0, nm2 = 0, n_inl = 0, found = 0;
                const void* d_map_desc = nullptr;
                if ((r = mvo_map_points_in_view(c.ctx, c.map, c.T_w_c, c.K4[0], c.K4[1], c.K4[2], c.K4[3], c.width, c.height, L.idx.data(),
                                                L.px.data(), (int)L.idx.size(), &nv, &d_map_desc))) {
                    status = r;
                    break;
                }
                if (nv && n) {
                    if ((int)L.matches.size() < nv) L.matches.resize(nv);
                    if ((r = mvo_match_features_dev(c.ctx, d_map_desc, nv, d_desc, n, 1, 2.0, 1.0, L.matches.data(), (int)L.matches.size(), &nm2))) {
                        status = r;
                        break;
                    }
                }
                double rvec[3], tvec[3];
                if ((r = mvo_solve_pnp_ransac(c.ctx, c.pts3d, c.pts2d, c.n_pairs, c.K4[0], c.K4[1], c.K4[2], c.K4[3], 100, 2.0f, 0.999, rvec,
                                              tvec, L.inl.data(), (int)L.inl.size(), &n_inl, &found))) {
                    status = r;
                    break;
                }
                st.n_inliers = n_inl;
            }
            // ---- bundle adjustment of this frame's window
            double* row = traj + 12 * (size_t)s;
            std::memset(row, 0, 12 * sizeof(double));
            if (c.ba_mode && !L.windows.empty()) {
                Window& w = *L.windows[(size_t)st.frame_no % L.windows.size()];
                const cv::Mat* Tnew = nullptr;
                std::vector<double> poses;
                mvo_ba_stats bs{};
                if (c.ba_mode == 2) {
                    Window& w0 = *L.windows[0];
                    poses.resize(16 * (size_t)w0.F);
                    if ((r = mvo_ba_solve_resident(c.ctx_ba, w0.resident)) || (r = mvo_ba_fetch(c.ctx_ba, w0.resident, poses.data(), nullptr, &bs))) {
                        status = r;
                        break;
                    }
                    st.ba_edges += w0.E;
                } else {
                    auto t0 = Clock::now();
                    restore(w);  // the state this window had when it was "new"
                    st.ns_restore += ns_since(t0);
                    if (c.chain && !w.pnp_kpt.empty()) {
                        // VisualOdometry::poseEstimationPnP_ (vo.cpp:304-357): pose of the new frame from solvePnPRansac on its
                        // own pairs, then ONLY the inliers become map-point connections -- the edges BA gets for this frame
                        vo::Frame& f = *w.frames_buff.back();
                        double rvec[3], tvec[3], R[9], Tcw[16] = {0}, Twc[16];
                        int n_inl = 0, found = 0;
                        if ((r = mvo_solve_pnp_ransac(c.ctx, w.pnp3d.data(), w.pnp2d.data(), (int)w.pnp_kpt.size(), c.K4[0], c.K4[1], c.K4[2],
                                                      c.K4[3], 100, 2.0f, 0.999, rvec, tvec, w.pnp_inl.data(), (int)w.pnp_inl.size(), &n_inl,
                                                      &found))) {
                            status = r;
                            break;
                        }
                        st.n_inliers = n_inl;
                        if (found && n_inl >= 3) {
                            mvo_rodrigues(rvec, R);
                            for (int i = 0; i < 3; ++i) {
                                for (int j = 0; j < 3; ++j) Tcw[4 * i + j] = R[3 * i + j];
                                Tcw[4 * i + 3] = tvec[i];
                            }
                            Tcw[15] = 1;
                            mvo_invert_pose(Tcw, Twc);
                            for (int k = 0; k < 16; ++k) f.T_w_c_.at<double>(k / 4, k % 4) = Twc[k];
                            std::unordered_map<int, vo::PtConn> conns;
                            conns.reserve((size_t)n_inl * 2);
                            for (int i = 0; i < n_inl; ++i) {
                                const int kpt = w.pnp_kpt[(size_t)w.pnp_inl[i]];
                                conns[kpt] = w.conn0[(size_t)kpt];
                            }
                            f.inliers_to_mappt_connections_.swap(conns);
                        }
                    }
                    // VisualOdometry::callBundleAdjustment_ (vo.cpp:384-478) in two halves
                    t0 = Clock::now();
                    L.pending = vo::buildBundleAdjustmentWindow(w.frames_buff, w.map, w.F);
                    st.ns_build += ns_since(t0);
                    t0 = Clock::now();
                    L.job.begin(L.pending.v_pts_2d, L.pending.v_pts_2d_to_3d_idx, L.K, L.pending.um_pts_3d_in_prev_frames,
                                L.pending.v_camera_poses, L.info, c.fix_points != 0, c.fix_points == 0);
                    st.ns_begin += ns_since(t0);
                    if (c.pipeline && s + 1 < steps) {  // run ahead: features of the next frame while the window is solved
                        t0 = Clock::now();
                        r = extract_and_match(L, st.frame_no + 1, &L.next_desc, &L.next_n, &L.next_match);
                        st.ns_extract += ns_since(t0);
                        if (r) {
                            L.job.end();
                            status = r;
                            break;
                        }
                        L.have_next = true;
                    }
                    t0 = Clock::now();
                    L.job.end();
                    st.ns_end += ns_since(t0);
                    bs = L.job.last;
                    Tnew = &w.frames_buff.back()->T_w_c_;
                    st.ba_edges += w.E;
                }
                st.ba_trials += bs.trials;
                st.ba_failed_solves += bs.failed_solves;
                st.ba_stale_steps += bs.stale_steps;
                st.ba_iterations += bs.iterations;
                st.ba_solves += 1;
                // trajectory row of the newest frame of the window: x y z, then R column by column
                double T[16];
                for (int k = 0; k < 16; ++k) T[k] = Tnew ? Tnew->at<double>(k / 4, k % 4) : poses[k];
                row[0] = T[3];
                row[1] = T[7];
                row[2] = T[11];
                for (int col = 0; col < 3; ++col)
                    for (int rr = 0; rr < 3; ++rr) row[3 + 3 * col + rr] = T[4 * rr + col];
            }
            if (c.track && c.keyframe_every > 0 && st.frame_no % c.keyframe_every == 0) {
                int n_inl = 0, n_keep = 0;
                if ((r = mvo_find_essential_inliers(c.ctx, c.kf_ref, c.kf_cur, c.kf_n, c.K4[0], c.K4[1], c.K4[2], c.K4[3], 0.999, 1.0,
                                                    L.inl.data(), (int)L.inl.size(), &n_inl))) {
                    status = r;
                    break;
                }
                std::vector<float> a(2 * (size_t)n_inl), b(2 * (size_t)n_inl);
                for (int i = 0; i < n_inl; ++i) {
                    std::memcpy(&a[2 * i], c.kf_ref + 2 * (size_t)L.inl[i], 8);
                    std::memcpy(&b[2 * i], c.kf_cur + 2 * (size_t)L.inl[i], 8);
                }
                double R[9], t[3];
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) R[3 * i + j] = c.kf_T_curr_to_prev[4 * i + j];
                    t[i] = c.kf_T_curr_to_prev[4 * i + 3];
                }
                if ((r = mvo_triangulate_points(c.ctx, a.data(), b.data(), n_inl, c.K4[0], c.K4[1], c.K4[2], c.K4[3], R, t, nullptr, L.tri.data()))) {
                    status = r;
                    break;
                }
                std::vector<int32_t> keep(n_inl > 0 ? n_inl : 1);
                if ((r = mvo_retain_good_triangulation(L.tri.data(), n_inl, c.kf_T_w_cur, c.kf_T_w_ref, 1.0, 20.0, keep.data(), &n_keep, nullptr))) {
                    status = r;
                    break;
                }
                st.n_tri = n_keep;
            }
            st.frame_no++;
        }
    } catch (const std::exception&) {
        status = MVO_ERR_HIP;  // (the adapters throw; mvo_last_error(ctx_ba) has the text)
    }
    hot_path_ctx_binding() = nullptr;
    return status;
}

void frame_loop_get_state(void* h, frame_loop_state* out) { *out = static_cast<Loop*>(h)->st; }

// ---- parity hooks for bench.py (outside the timed region) ----------------------------------------------------------------
// Window k of the pool at its initial state, marshalled and flattened exactly as frame_loop_run hands it to the solver
// (restore -> buildBundleAdjustmentWindow -> FlatBundle::flatten: edge order = the iteration order of the frames'
// unordered_maps, f32 storage of pixels and landmarks widened to f64): what the oracle must be given to reproduce the
// trajectory row the timed loop wrote for a frame that used this window.  Arrays: poses F x 16, points L x 3, edges E.
// poses_now / points_now (optional): the state the objects are in when this is called -- for the window of the last frame
// of a run, what its solve wrote back (poses as f64, landmarks as f32: g2o_ba.cpp:298-316) -- in the same flat order.
int frame_loop_export_window(void* h, int k, int cap_edges, double* poses, double* points, int32_t* edge_pose, int32_t* edge_point,
                             double* edge_uv, int32_t* FLE, double* poses_now, double* points_now) {
    Loop& L = *static_cast<Loop*>(h);
    if (k < 0 || (size_t)k >= L.windows.size()) return MVO_ERR_INVALID;
    Window& w = *L.windows[(size_t)k];
    if (poses_now || points_now) {
        vo::BaWindow now = vo::buildBundleAdjustmentWindow(w.frames_buff, w.map, w.F);
        optimization::FlatBundle fn;
        fn.flatten(now.v_pts_2d, now.v_pts_2d_to_3d_idx, L.K, now.um_pts_3d_in_prev_frames, now.v_camera_poses, L.info, L.c.fix_points != 0);
        if (fn.pr.n_poses > w.F || fn.pr.n_points > w.L) return MVO_ERR_CAPACITY;
        if (poses_now) std::memcpy(poses_now, fn.poses.data(), fn.poses.size() * sizeof(double));
        if (points_now) std::memcpy(points_now, fn.pts.data(), fn.pts.size() * sizeof(double));
    }
    restore(w);
    vo::BaWindow bw = vo::buildBundleAdjustmentWindow(w.frames_buff, w.map, w.F);
    optimization::FlatBundle fb;
    fb.flatten(bw.v_pts_2d, bw.v_pts_2d_to_3d_idx, L.K, bw.um_pts_3d_in_prev_frames, bw.v_camera_poses, L.info, L.c.fix_points != 0);
    FLE[0] = fb.pr.n_poses;
    FLE[1] = fb.pr.n_points;
    FLE[2] = fb.pr.n_edges;
    if (fb.pr.n_edges > cap_edges || fb.pr.n_poses > w.F || fb.pr.n_points > w.L) return MVO_ERR_CAPACITY;
    std::memcpy(poses, fb.poses.data(), fb.poses.size() * sizeof(double));
    std::memcpy(points, fb.pts.data(), fb.pts.size() * sizeof(double));
    std::memcpy(edge_pose, fb.ep.data(), fb.ep.size() * sizeof(int32_t));
    std::memcpy(edge_point, fb.el.data(), fb.el.size() * sizeof(int32_t));
    std::memcpy(edge_uv, fb.uv.data(), fb.uv.size() * sizeof(double));
    return MVO_OK;
}

// on != 0: every extraction from now on also copies its keypoints, descriptors (through the host output of
// mvo_calc_descriptors_dev) and matches aside.
void frame_loop_capture(void* h, int on) { static_cast<Loop*>(h)->capture = on != 0; }

// The last captured frame: frame number (-1 = none), counts; kps / desc (n x 32) / matches are copied when non-null.
int frame_loop_get_capture(void* h, int32_t* frame_no, int32_t* n, int32_t* nm, mvo_keypoint* kps, uint8_t* desc, mvo_dmatch* matches) {
    Loop& L = *static_cast<Loop*>(h);
    *frame_no = L.cap_frame;
    *n = L.cap_n;
    *nm = L.cap_nm;
    if (kps) std::memcpy(kps, L.cap_kps.data(), (size_t)L.cap_n * sizeof(mvo_keypoint));
    if (desc) std::memcpy(desc, L.cap_desc.data(), (size_t)L.cap_n * 32);
    if (matches) std::memcpy(matches, L.cap_matches.data(), (size_t)L.cap_nm * sizeof(mvo_dmatch));
    return MVO_OK;
}

void frame_loop_destroy(void* h) {
    Loop* L = static_cast<Loop*>(h);
    if (!L) return;
    for (auto& w : L->windows)
        if (w->resident) mvo_ba_release(L->c.ctx_ba, w->resident);
    delete L;
}
}
