// host/src/g2o_ba_mvo.cpp -- REPLACES src/optimization/g2o_ba.cpp of the reference: defines exactly the two functions that
// include/my_slam/optimization/g2o_ba.h:16-30 declares, against the reference's OWN header (no header of the reference is
// shadowed or edited: drop this file into the build instead of g2o_ba.cpp, add <repo>/include and this directory to the
// include path, link libmvo_hip.so -- INTEGRATION.md).  g2o, CSparse and Sophus are no longer needed by this target.
// The same file is compiled against this repo's OpenCV-less mirror of the header for the tests (host/include).
#include "my_slam/optimization/g2o_ba.h"

#include "flat_bundle.h"
#include "mvo_hot_path.h"

namespace my_slam {
namespace optimization {

void bundleAdjustment(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx,
                             const cv::Mat& K, std::unordered_map<int, cv::Point3f*>& pts_3d,
                             vector<cv::Mat*>& v_camera_g2o_poses, const cv::Mat& information_matrix,
                             bool is_fix_map_pts, bool is_update_map_pts) {
    FlatBundle fb;
    fb.flatten(v_pts_2d, v_pts_2d_to_3d_idx, K, pts_3d, v_camera_g2o_poses, information_matrix, is_fix_map_pts);
    mvo_ba_stats st;
    mvo_check(mvo_bundle_adjustment(hot_path_ctx(), &fb.pr, &st), "bundleAdjustment");
    fb.scatter(pts_3d, v_camera_g2o_poses, is_update_map_pts);
}

// g2o_ba.h:16-21 (dead code in the reference: vo.cpp:456 `if (1)`): single pose + its points, no robust kernel.
// Provided for interface completeness on top of the same solver (Huber delta large = no robustification).
void optimizeSingleFrame(const vector<cv::Point2f*>& points_2d, const cv::Mat& K, vector<cv::Point3f*>& points_3d,
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
