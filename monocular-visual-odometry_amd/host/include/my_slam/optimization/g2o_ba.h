// my_slam/optimization/g2o_ba.h -- drop-in for the reference's include/my_slam/optimization/g2o_ba.h:16-30
// (implementation src/optimization/g2o_ba.cpp:34-317).  Same signatures: raw pointers into live Frame / MapPoint
// storage come in, poses (and optionally points) are overwritten in place.  The graph the reference hands to g2o
// is flattened here and solved by libmvo_hip.so (one persistent LM launch on the MI355X).
#ifndef MY_SLAM_G2O_BA_H
#define MY_SLAM_G2O_BA_H
#include "my_slam/common_include.h"

namespace my_slam {
namespace optimization {

inline void bundleAdjustment(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx,
                             const cv::Mat& K, std::unordered_map<int, cv::Point3f*>& pts_3d,
                             vector<cv::Mat*>& v_camera_g2o_poses, const cv::Mat& information_matrix,
                             bool is_fix_map_pts = false, bool is_update_map_pts = true) {
    const int num_frames = (int)v_camera_g2o_poses.size();
    vector<double> poses(16 * (size_t)num_frames);
    for (int i = 0; i < num_frames; ++i)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) poses[16 * i + 4 * r + c] = v_camera_g2o_poses[i]->at<double>(r, c);
    // landmarks in the unordered_map's iteration order, like the vertex ids of g2o_ba.cpp:225-243
    std::unordered_map<int, int> id2slot;
    vector<int> slot2id;
    vector<double> pts;
    for (auto it = pts_3d.begin(); it != pts_3d.end(); ++it) {
        id2slot[it->first] = (int)slot2id.size();
        slot2id.push_back(it->first);
        pts.push_back(it->second->x);
        pts.push_back(it->second->y);
        pts.push_back(it->second->z);
    }
    vector<int> ep, el;
    vector<double> uv;
    for (int f = 0; f < num_frames; ++f)
        for (size_t j = 0; j < v_pts_2d[f].size(); ++j) {
            ep.push_back(f);
            el.push_back(id2slot.at(v_pts_2d_to_3d_idx[f][j]));
            uv.push_back(v_pts_2d[f][j]->x);
            uv.push_back(v_pts_2d[f][j]->y);
        }
    mvo_ba_problem pr{};
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
    mvo_ba_stats st;
    mvo_check(mvo_bundle_adjustment(hot_path_ctx(), &pr, &st), "bundleAdjustment");
    // 1. camera poses (g2o_ba.cpp:298-305)
    for (int i = 0; i < num_frames; ++i)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) v_camera_g2o_poses[i]->at<double>(r, c) = poses[16 * i + 4 * r + c];
    // 2. points (g2o_ba.cpp:308-316), double -> float
    if (is_update_map_pts)
        for (size_t s = 0; s < slot2id.size(); ++s) {
            cv::Point3f* p = pts_3d[slot2id[s]];
            p->x = (float)pts[3 * s];
            p->y = (float)pts[3 * s + 1];
            p->z = (float)pts[3 * s + 2];
        }
}

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
