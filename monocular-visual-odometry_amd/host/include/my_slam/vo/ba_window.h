// my_slam/vo/ba_window.h -- the sliding-window marshalling of VisualOdometry::callBundleAdjustment_
// (reference src/vo/vo.cpp:384-478) as a free function over the same containers: frames_buff_ (std::deque of
// Frame::Ptr, newest last; vo.h:64,77-86), the Map, the camera matrix K.  Same rules: the newest
// min(num_prev_frames_to_opti_by_ba, buffered - 1) frames, newest first; frames with fewer than 3 map-point
// connections are skipped; connections to deleted map points are skipped; raw pointers into
// Frame::keypoints_[i].pt, MapPoint::pos_ and Frame::T_w_c_ are handed to optimization::bundleAdjustment, which
// overwrites poses (and points when is_ba_fix_map_points is false) in place.
#ifndef MY_SLAM_BA_WINDOW_H
#define MY_SLAM_BA_WINDOW_H
#include <deque>

#include "my_slam/basics/config.h"
#include "my_slam/optimization/g2o_ba.h"
#include "my_slam/vo/map.h"

namespace my_slam {
namespace vo {

// The pointer lists optimization::bundleAdjustment takes (g2o_ba.h:23-30), one entry of the outer vectors per window frame.
// The member names are the argument names of the reference's call (vo.cpp:458-462) so that call sites read alike.
struct BaWindow {
    vector<vector<cv::Point2f*>> v_pts_2d;                           // per frame: measured pixels (into Frame::keypoints_)
    vector<vector<int>> v_pts_2d_to_3d_idx;                          // per frame: map-point id of every measurement
    std::unordered_map<int, cv::Point3f*> um_pts_3d_in_prev_frames;  // id -> landmark position (into MapPoint::pos_)
    vector<cv::Point3f*> v_pts_3d_only_in_curr;                      // landmarks seen by the newest frame (optimizeSingleFrame's list)
    vector<cv::Mat*> v_camera_poses;                                 // per frame: its T_w_c_
    vector<int> frame_ids;
};

// Window = the newest `window_len` frames of the buffer, newest first, at most all but the oldest buffered frame
// (vo.cpp:395-396).  Per frame, in the iteration order of its connection table: one measurement per connection whose map
// point still exists (vo.cpp:440-441); frames with fewer than three connections contribute nothing, not even a pose
// (vo.cpp:423-426).  Order matters downstream (vertex ids, summation order), hence the explicit walk.
inline BaWindow buildBundleAdjustmentWindow(const std::deque<Frame::Ptr>& frames_buff, const Map::Ptr& map, int window_len) {
    BaWindow w;
    // (no reserve() on the id -> landmark table: its iteration order becomes the landmark order of the graph, g2o_ba.cpp:225-243,
    // and must be what the reference's default-constructed table of vo.cpp:404 would yield)
    const int buffered = (int)frames_buff.size();
    const int take = std::min(window_len, buffered - 1);
    bool newest = true;
    for (auto it = frames_buff.rbegin(); it != frames_buff.rend() && (int)(it - frames_buff.rbegin()) < take; ++it, newest = false) {
        Frame& frame = **it;
        if (frame.inliers_to_mappt_connections_.size() < 3) continue;
        vector<cv::Point2f*> pixels;
        vector<int> ids;
        pixels.reserve(frame.inliers_to_mappt_connections_.size());
        ids.reserve(frame.inliers_to_mappt_connections_.size());
        for (const auto& conn : frame.inliers_to_mappt_connections_) {  // (keypoint index -> {.., map-point id})
            const int id = conn.second.pt_map_idx;
            const auto found = map->map_points_.find(id);
            if (found == map->map_points_.end()) continue;
            cv::Point3f* position = &found->second->pos_;
            pixels.push_back(&frame.keypoints_[conn.first].pt);
            ids.push_back(id);
            w.um_pts_3d_in_prev_frames[id] = position;
            if (newest) w.v_pts_3d_only_in_curr.push_back(position);
        }
        w.v_pts_2d.push_back(std::move(pixels));
        w.v_pts_2d_to_3d_idx.push_back(std::move(ids));
        w.v_camera_poses.push_back(&frame.T_w_c_);
        w.frame_ids.push_back(frame.id_);
    }
    return w;
}

// VisualOdometry::callBundleAdjustment_ (vo.cpp:384-478): parameters latched from Config on first use.
inline void callBundleAdjustment(const std::deque<Frame::Ptr>& frames_buff, const Map::Ptr& map, const cv::Mat& K) {
    static const bool is_enable_ba = basics::Config::getBool("is_enable_ba");
    static const int num_prev_frames_to_opti_by_ba = basics::Config::get<int>("num_prev_frames_to_opti_by_ba");
    static const vector<double> im = basics::str2vecdouble(basics::Config::get<string>("information_matrix"));
    static const bool is_ba_fix_map_points = basics::Config::getBool("is_ba_fix_map_points");
    static const bool is_ba_update_map_points = !is_ba_fix_map_points;
    if (!is_enable_ba) return;
    cv::Mat information_matrix(2, 2, CV_64FC1);
    for (int i = 0; i < 4; ++i) information_matrix.at<double>(i / 2, i % 2) = im[i];
    BaWindow w = buildBundleAdjustmentWindow(frames_buff, map, num_prev_frames_to_opti_by_ba);
    if (w.v_camera_poses.empty()) return;
    optimization::bundleAdjustment(w.v_pts_2d, w.v_pts_2d_to_3d_idx, K, w.um_pts_3d_in_prev_frames, w.v_camera_poses,
                                   information_matrix, is_ba_fix_map_points, is_ba_update_map_points);
}

}  // namespace vo
}  // namespace my_slam
#endif
