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

struct BaWindow {  // what vo.cpp:408-449 builds
    vector<vector<cv::Point2f*>> v_pts_2d;
    vector<vector<int>> v_pts_2d_to_3d_idx;
    std::unordered_map<int, cv::Point3f*> um_pts_3d_in_prev_frames;
    vector<cv::Point3f*> v_pts_3d_only_in_curr;
    vector<cv::Mat*> v_camera_poses;
    vector<int> frame_ids;
};

inline BaWindow buildBundleAdjustmentWindow(const std::deque<Frame::Ptr>& frames_buff, const Map::Ptr& map,
                                            int num_prev_frames_to_opti_by_ba) {
    BaWindow w;
    const int kTotalFrames = (int)frames_buff.size();
    const int kNumFramesForBA = std::min(num_prev_frames_to_opti_by_ba, kTotalFrames - 1);  // vo.cpp:395-396
    int ith_frame = 0;
    for (int ith_frame_in_buff = kTotalFrames - 1; ith_frame_in_buff >= kTotalFrames - kNumFramesForBA;
         ith_frame_in_buff--, ith_frame++) {
        Frame::Ptr frame = frames_buff[ith_frame_in_buff];
        const int num_mappt_in_frame = (int)frame->inliers_to_mappt_connections_.size();
        if (num_mappt_in_frame < 3) continue;  // Too few mappoints. Not optimizing this frame (vo.cpp:423-426)
        w.v_pts_2d.push_back(vector<cv::Point2f*>());
        w.v_pts_2d_to_3d_idx.push_back(vector<int>());
        w.v_camera_poses.push_back(&frame->T_w_c_);
        w.frame_ids.push_back(frame->id_);
        for (auto ite = frame->inliers_to_mappt_connections_.begin(); ite != frame->inliers_to_mappt_connections_.end(); ++ite) {
            const int kpt_idx = ite->first;
            const int mappt_idx = ite->second.pt_map_idx;
            auto mp = map->map_points_.find(mappt_idx);
            if (mp == map->map_points_.end()) continue;  // point has been deleted (vo.cpp:440-441)
            w.v_pts_2d.back().push_back(&(frame->keypoints_[kpt_idx].pt));
            w.v_pts_2d_to_3d_idx.back().push_back(mappt_idx);
            cv::Point3f* p = &(mp->second->pos_);
            w.um_pts_3d_in_prev_frames[mappt_idx] = p;
            if (ith_frame == 0) w.v_pts_3d_only_in_curr.push_back(p);
        }
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
