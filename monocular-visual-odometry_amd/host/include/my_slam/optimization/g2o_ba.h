// my_slam/optimization/g2o_ba.h -- drop-in for the reference's include/my_slam/optimization/g2o_ba.h:16-30
// (implementation src/optimization/g2o_ba.cpp:34-317).  Same signatures: raw pointers into live Frame / MapPoint
// storage come in, poses (and optionally points) are overwritten in place.  The graph the reference hands to g2o
// is flattened here and solved by libmvo_hip.so (one persistent LM launch on the MI355X).
#ifndef MY_SLAM_G2O_BA_H
#define MY_SLAM_G2O_BA_H
#include "my_slam/common_include.h"

namespace my_slam {
namespace optimization {

// The graph of g2o_ba.cpp:193-271 as flat arrays: pose vertices in the order of v_camera_g2o_poses, landmark vertices
// in the iteration order of the unordered_map (like the vertex ids of g2o_ba.cpp:225-243), one edge per observation.
struct FlatBundle {
    vector<double> poses, pts, uv;
    vector<int> ep, el, slot2id;
    mvo_ba_problem pr{};

    void flatten(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx, const cv::Mat& K,
                 std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses,
                 const cv::Mat& information_matrix, bool is_fix_map_pts) {
        const int num_frames = (int)v_camera_g2o_poses.size();
        poses.resize(16 * (size_t)num_frames);
        for (int i = 0; i < num_frames; ++i)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) poses[16 * i + 4 * r + c] = v_camera_g2o_poses[i]->at<double>(r, c);
        std::unordered_map<int, int> id2slot;
        id2slot.reserve(pts_3d.size());
        slot2id.clear();
        pts.clear();
        for (auto it = pts_3d.begin(); it != pts_3d.end(); ++it) {
            id2slot[it->first] = (int)slot2id.size();
            slot2id.push_back(it->first);
            pts.push_back(it->second->x);
            pts.push_back(it->second->y);
            pts.push_back(it->second->z);
        }
        ep.clear();
        el.clear();
        uv.clear();
        for (int f = 0; f < num_frames; ++f)
            for (size_t j = 0; j < v_pts_2d[f].size(); ++j) {
                ep.push_back(f);
                el.push_back(id2slot.at(v_pts_2d_to_3d_idx[f][j]));
                uv.push_back(v_pts_2d[f][j]->x);
                uv.push_back(v_pts_2d[f][j]->y);
            }
        pr = mvo_ba_problem{};
        pr.n_poses = num_frames;
        pr.n_points = (int)slot2id.size();
        pr.n_edges = (int)ep.size();
        pr.pose_T_w_c = poses.data();
        pr.points = pts.data();
        pr.edge_pose = ep.data();
        pr.edge_point = el.data();
        pr.edge_uv = uv.data();
        pr.focal = K.at<double>(0, 0);  // CameraParameters(K(0,0), (K(0,2), K(1,2)), 0): fy is not used (:219-222)
        pr.cx = K.at<double>(0, 2);
        pr.cy = K.at<double>(1, 2);
        for (int i = 0; i < 4; ++i) pr.info[i] = information_matrix.at<double>(i / 2, i % 2);
        pr.huber_delta = 1.0;
        pr.fix_points = is_fix_map_pts ? 1 : 0;
        pr.pose_fixed = nullptr;  // g2o_ba.cpp:210-211: no pose is fixed
        pr.max_iterations = 50;
    }
    // g2o_ba.cpp:298-316: poses back into the caller's matrices, points back as float
    void scatter(std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses, bool is_update_map_pts) const {
        for (size_t i = 0; i < v_camera_g2o_poses.size(); ++i)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) v_camera_g2o_poses[i]->at<double>(r, c) = poses[16 * i + 4 * r + c];
        if (is_update_map_pts)
            for (size_t s = 0; s < slot2id.size(); ++s) {
                cv::Point3f* p = pts_3d[slot2id[s]];
                p->x = (float)pts[3 * s];
                p->y = (float)pts[3 * s + 1];
                p->z = (float)pts[3 * s + 2];
            }
    }
};

inline void bundleAdjustment(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx,
                             const cv::Mat& K, std::unordered_map<int, cv::Point3f*>& pts_3d,
                             vector<cv::Mat*>& v_camera_g2o_poses, const cv::Mat& information_matrix,
                             bool is_fix_map_pts = false, bool is_update_map_pts = true) {
    FlatBundle fb;
    fb.flatten(v_pts_2d, v_pts_2d_to_3d_idx, K, pts_3d, v_camera_g2o_poses, information_matrix, is_fix_map_pts);
    mvo_ba_stats st;
    mvo_check(mvo_bundle_adjustment(hot_path_ctx(), &fb.pr, &st), "bundleAdjustment");
    fb.scatter(pts_3d, v_camera_g2o_poses, is_update_map_pts);
}

// The same call in two halves for drivers that overlap the solve with other host work (e.g. the extraction of the
// next frame): begin() = graph -> flat window -> upload -> launch; end() = wait + write-back.  The pointer lists must
// stay valid in between.  `last` keeps the statistics of the finished solve.
class BundleAdjustmentJob {
public:
    BundleAdjustmentJob() = default;
    BundleAdjustmentJob(const BundleAdjustmentJob&) = delete;
    BundleAdjustmentJob& operator=(const BundleAdjustmentJob&) = delete;
    // a job that goes out of scope between its two halves (exception, early return) still owns a launch that writes into
    // the ctx's pooled workspace: wait for it; the results are dropped, errors cannot leave a destructor
    ~BundleAdjustmentJob() {
        if (!active_) return;
        active_ = false;
        (void)mvo_bundle_adjustment_end(ctx_, nullptr, nullptr);
    }
    void begin(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx, const cv::Mat& K,
               std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses,
               const cv::Mat& information_matrix, bool is_fix_map_pts = false, bool is_update_map_pts = true) {
        fb_.flatten(v_pts_2d, v_pts_2d_to_3d_idx, K, pts_3d, v_camera_g2o_poses, information_matrix, is_fix_map_pts);
        pts_3d_ = &pts_3d;
        poses_ = &v_camera_g2o_poses;
        update_ = is_update_map_pts;
        ctx_ = hot_path_ctx();
        mvo_check(mvo_bundle_adjustment_begin(ctx_, &fb_.pr), "bundleAdjustment (begin)");
        active_ = true;
    }
    void end() {
        if (!active_) return;
        active_ = false;
        mvo_check(mvo_bundle_adjustment_end(ctx_, &fb_.pr, &last), "bundleAdjustment (end)");
        fb_.scatter(*pts_3d_, *poses_, update_);
    }
    bool active() const { return active_; }
    mvo_ba_stats last{};

private:
    FlatBundle fb_;
    std::unordered_map<int, cv::Point3f*>* pts_3d_ = nullptr;
    vector<cv::Mat*>* poses_ = nullptr;
    mvo_ctx* ctx_ = nullptr;  // the ctx the launch was made on (the binding of the thread may change in between)
    bool update_ = true, active_ = false;
};

// g2o_ba.h:16-21 (dead code in the reference: vo.cpp:456 `if (1)`): single pose + its points, no robust kernel.
// Provided for interface completeness on top of the same solver (Huber delta large = no robustification).
inline void optimizeSingleFrame(const vector<cv::Point2f*>& points_2d, const cv::Mat& K, vector<cv::Point3f*>& points_3d,
                                cv::Mat& cam_pose_in_world, bool is_fix_map_pts, bool is_update_map_pts) {
    vector<double> pose(16), pts, uv;
    vector<int> ep, el;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) pose[4 * r + c] = cam_pose_in_world.at<double>(r, c);
    for (size_t i = 0; i < points_3d.size(); ++i) {
        pts.push_back(points_3d[i]->x);
        pts.push_back(points_3d[i]->y);
        pts.push_back(points_3d[i]->z);
        ep.push_back(0);
        el.push_back((int)i);
        uv.push_back(points_2d[i]->x);
        uv.push_back(points_2d[i]->y);
    }
    mvo_ba_problem pr{};
    pr.n_poses = 1;
    pr.n_points = (int)points_3d.size();
    pr.n_edges = (int)ep.size();
    pr.pose_T_w_c = pose.data();
    pr.points = pts.data();
    pr.edge_pose = ep.data();
    pr.edge_point = el.data();
    pr.edge_uv = uv.data();
    pr.focal = K.at<double>(0, 0);
    pr.cx = K.at<double>(0, 2);
    pr.cy = K.at<double>(1, 2);
    pr.info[0] = pr.info[3] = 1.0;
    pr.huber_delta = 1e100;
    pr.fix_points = is_fix_map_pts ? 1 : 0;
    pr.max_iterations = 50;
    mvo_check(mvo_bundle_adjustment(hot_path_ctx(), &pr, nullptr), "optimizeSingleFrame");
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) cam_pose_in_world.at<double>(r, c) = pose[4 * r + c];
    if (is_update_map_pts)
        for (size_t i = 0; i < points_3d.size(); ++i) {
            points_3d[i]->x = (float)pts[3 * i];
            points_3d[i]->y = (float)pts[3 * i + 1];
            points_3d[i]->z = (float)pts[3 * i + 2];
        }
}

}  // namespace optimization
}  // namespace my_slam
#endif
