// my_slam/vo/frame.h -- Frame / PtConn with the reference's field names (include/my_slam/vo/frame.h:16-96); only the
// members the hot path reads or writes are kept (camera / triangulation helpers live outside the hot path).
#ifndef MY_SLAM_FRAME_H
#define MY_SLAM_FRAME_H
#include "my_slam/common_include.h"
#include "my_slam/geometry/feature_match.h"

namespace my_slam {
namespace vo {

typedef struct PtConn_ {
    int pt_ref_idx;
    int pt_map_idx;
} PtConn;

class Frame {
public:
    typedef std::shared_ptr<Frame> Ptr;
    static int& factory_id() {
        static int id = 0;
        return id;
    }

public:
    int id_ = 0;
    double time_stamp_ = -1;

    // -- image features
    cv::Mat rgb_img_;
    vector<cv::KeyPoint> keypoints_;
    cv::Mat descriptors_;
    vector<vector<unsigned char>> kpts_colors_;  // rgb colors

    // -- Matches with reference keyframe / map
    vector<cv::DMatch> matches_with_ref_;
    vector<cv::DMatch> inliers_matches_with_ref_;
    vector<cv::DMatch> inliers_matches_for_3d_;
    vector<cv::Point3f> inliers_pts3d_;
    vector<double> triangulation_angles_of_inliers_;
    std::unordered_map<int, PtConn> inliers_to_mappt_connections_;  // curr idx -> idx in ref, and map
    vector<cv::DMatch> matches_with_map_;

    // -- Current pose (cam -> world, see vo.cpp:31,89)
    cv::Mat T_w_c_;

public:
    static Frame::Ptr createFrame(cv::Mat rgb_img, double time_stamp = -1) {
        Frame::Ptr f(new Frame());
        f->rgb_img_ = rgb_img;
        f->id_ = factory_id()++;
        f->time_stamp_ = time_stamp;
        f->T_w_c_ = cv::Mat::eye(4, 4, CV_64FC1);
        return f;
    }
    void clearNoUsed() {
        kpts_colors_.clear();
        matches_with_ref_.clear();
        inliers_matches_with_ref_.clear();
        inliers_matches_for_3d_.clear();
        matches_with_map_.clear();
    }
    void calcKeyPoints() {
        geometry::calcKeyPoints(rgb_img_, keypoints_);
        geometry::detail::pyramid_token() = (long long)id_ + 1;  // the ctx now caches THIS frame's pyramid
    }
    void calcDescriptors() {
        // same image as calcKeyPoints -> the device pyramid is reused (the reference builds it twice)
        geometry::detail::reuse_pyramid_flag() = geometry::detail::pyramid_token() == (long long)id_ + 1;
        geometry::calcDescriptors(rgb_img_, keypoints_, descriptors_);
        geometry::detail::reuse_pyramid_flag() = false;
        kpts_colors_.clear();
        for (const cv::KeyPoint& kpt : keypoints_) {  // frame.h:80-85 + basics::getPixelAt: BGR -> r,g,b
            int x = (int)std::floor(kpt.pt.x), y = (int)std::floor(kpt.pt.y);
            const unsigned char* px = rgb_img_.ptr<unsigned char>(y) + (size_t)x * rgb_img_.channels();
            if (rgb_img_.channels() >= 3)
                kpts_colors_.push_back({px[2], px[1], px[0]});
            else
                kpts_colors_.push_back({px[0], px[0], px[0]});
        }
    }
    bool isMappoint(int idx) { return inliers_to_mappt_connections_.find(idx) != inliers_to_mappt_connections_.end(); }

};

}  // namespace vo
}  // namespace my_slam
#endif
