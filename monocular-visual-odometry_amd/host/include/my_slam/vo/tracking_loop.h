// my_slam/vo/tracking_loop.h -- the DOING_TRACKING branch of VisualOdometry::addFrame (reference
// src/vo/vo_addFrame.cpp:70-124) composed from the mirrored hot-path pieces, plus the three small members it calls
// (checkLargeMoveForAddKeyFrame_ vo.cpp:247-266, pushCurrPointsToMap_ vo.cpp:528-576, optimizeMap_ vo.cpp:488-526).
// It exists so that the rows can be exercised TOGETHER the way the reference chains them (tests/): the state
// machine, initialisation and display of the reference are not part of this repository.
#ifndef MY_SLAM_TRACKING_LOOP_H
#define MY_SLAM_TRACKING_LOOP_H
#include <deque>

#include "my_slam/vo/ba_window.h"
#include "my_slam/vo/keyframe.h"
#include "my_slam/vo/pnp_tracking.h"

namespace my_slam {
namespace vo {

struct TrackingState {  // the members of VisualOdometry the tracking branch touches (vo.h:58-77)
    Map::Ptr map_{new Map()};
    MapOnDevice dev_map_;
    std::deque<Frame::Ptr> frames_buff_;
    Frame::Ptr ref_, prev_;
    double map_point_erase_ratio_ = 0.1;  // the function-local static of optimizeMap_
    static constexpr size_t kBuffSize_ = 20;
    void pushFrameToBuff(Frame::Ptr frame) {
        frames_buff_.push_back(frame);
        if (frames_buff_.size() > kBuffSize_) frames_buff_.pop_front();
    }
};

inline cv::Point3f preTranslatePoint3f(const cv::Point3f& p, const cv::Mat& T) {  // opencv_funcs.cpp:67-78
    const double q[4] = {p.x, p.y, p.z, 1};
    double res[3] = {0, 0, 0};
    for (int r = 0; r < 3; r++)
        for (int j = 0; j < 4; j++) res[r] += T.at<double>(r, j) * q[j];
    return cv::Point3f((float)res[0], (float)res[1], (float)res[2]);
}

inline bool isInFrame(const Frame::Ptr& f, const cv::Point3f& p_world, const cv::Mat& K) {  // frame.cpp:29-36
    const cv::Point3f pc = preTranslatePoint3f(p_world, basics::invT(f->T_w_c_));
    if (pc.z < 0) return false;
    const float u = (float)(K.at<double>(0, 0) * pc.x / pc.z + K.at<double>(0, 2));
    const float v = (float)(K.at<double>(1, 1) * pc.y / pc.z + K.at<double>(1, 2));
    return u > 0 && v > 0 && u < f->rgb_img_.cols && v < f->rgb_img_.rows;
}

inline bool checkLargeMoveForAddKeyFrame(const Frame::Ptr& curr, const Frame::Ptr& ref) {
    static const double min_dist_between_two_keyframes = basics::Config::get<double>("min_dist_between_two_keyframes");
    const cv::Mat T_key_to_curr = getMotionFromFrame1to2(ref, curr);  // ref->T_w_c_.inv() * curr->T_w_c_
    double s = 0;
    for (int i = 0; i < 3; ++i) s = s + T_key_to_curr.at<double>(i, 3) * T_key_to_curr.at<double>(i, 3);
    return std::sqrt(s) > min_dist_between_two_keyframes;
}

inline void pushCurrPointsToMap(TrackingState& st, const Frame::Ptr& curr) {
    for (size_t i = 0; i < curr->inliers_matches_for_3d_.size(); i++) {
        const cv::DMatch& dm = curr->inliers_matches_for_3d_[i];
        const int pt_idx = dm.trainIdx;
        int map_point_id;
        if (st.ref_->isMappoint(dm.queryIdx)) {
            map_point_id = st.ref_->inliers_to_mappt_connections_[dm.queryIdx].pt_map_idx;
        } else {
            const cv::Point3f world_pos = preTranslatePoint3f(curr->inliers_pts3d_[i], curr->T_w_c_);
            cv::Mat desc(1, 32, CV_8UC1), norm(3, 1, CV_64FC1);
            std::memcpy(desc.data, curr->descriptors_.ptr<unsigned char>(pt_idx), 32);
            double len = 0;
            const double w[3] = {world_pos.x, world_pos.y, world_pos.z};
            for (int r = 0; r < 3; ++r) {
                norm.at<double>(r, 0) = w[r] - curr->T_w_c_.at<double>(r, 3);
                len += norm.at<double>(r, 0) * norm.at<double>(r, 0);
            }
            len = std::sqrt(len);
            for (int r = 0; r < 3; ++r) norm.at<double>(r, 0) /= len;
            const vector<unsigned char> rgb = pt_idx < (int)curr->kpts_colors_.size() ? curr->kpts_colors_[pt_idx]
                                                                                       : vector<unsigned char>{0, 0, 0};
            MapPoint::Ptr map_point(new MapPoint(world_pos, desc, norm, rgb[0], rgb[1], rgb[2]));
            map_point_id = map_point->id_;
            st.map_->insertMapPoint(map_point);
        }
        curr->inliers_to_mappt_connections_.insert({pt_idx, PtConn{dm.queryIdx, map_point_id}});
    }
}

inline void optimizeMap(TrackingState& st, const Frame::Ptr& curr, const cv::Mat& K) {
    const double default_erase = 0.1;
    for (auto iter = st.map_->map_points_.begin(); iter != st.map_->map_points_.end();) {
        const MapPoint::Ptr& p = iter->second;
        if (!isInFrame(curr, p->pos_, K)) {
            iter = st.map_->map_points_.erase(iter);
            continue;
        }
        const float match_ratio = float(p->matched_times_) / p->visible_times_;
        if (match_ratio < st.map_point_erase_ratio_) {
            iter = st.map_->map_points_.erase(iter);
            continue;
        }
        double n[3], len = 0, dot = 0;  // getViewAngle_ (vo.cpp:578-584)
        const double w[3] = {p->pos_.x, p->pos_.y, p->pos_.z};
        for (int r = 0; r < 3; ++r) {
            n[r] = w[r] - curr->T_w_c_.at<double>(r, 3);
            len += n[r] * n[r];
        }
        len = std::sqrt(len);
        for (int r = 0; r < 3; ++r) dot += n[r] / len * p->norm_.at<double>(r, 0);
        if (std::acos(dot) > M_PI / 4.) {
            iter = st.map_->map_points_.erase(iter);
            continue;
        }
        iter++;
    }
    if (st.map_->map_points_.size() > 1000)
        st.map_point_erase_ratio_ += 0.05;
    else
        st.map_point_erase_ratio_ = default_erase;
}

// One frame of the tracking state (vo_addFrame.cpp:9-27 + 70-140); curr must carry its image size, keypoints and
// descriptors (Frame::calcKeyPoints / calcDescriptors in the real pipeline).  Returns is_pnp_good; *is_keyframe
// tells whether the frame was inserted as a keyframe.
inline bool trackFrame(TrackingState& st, const Frame::Ptr& curr, const cv::Mat& K, bool* is_keyframe = nullptr) {
    st.pushFrameToBuff(curr);
    if (is_keyframe) *is_keyframe = false;
    curr->T_w_c_ = st.ref_->T_w_c_.clone();  // Initial estimation of the current pose
    if (!st.prev_) st.prev_ = st.ref_;
    const bool is_pnp_good = poseEstimationPnP(st.dev_map_, st.map_, curr, st.prev_, K);
    if (is_pnp_good) {
        callBundleAdjustment(st.frames_buff_, st.map_, K);
        if (checkLargeMoveForAddKeyFrame(curr, st.ref_)) {
            triangulateWithReferenceKeyframe(curr, st.ref_, K);
            pushCurrPointsToMap(st, curr);
            optimizeMap(st, curr, K);
            st.map_->insertKeyFrame(curr);  // addKeyFrame_
            st.ref_ = curr;
            if (is_keyframe) *is_keyframe = true;
        }
    }
    st.prev_ = curr;
    return is_pnp_good;
}

}  // namespace vo
}  // namespace my_slam
#endif
