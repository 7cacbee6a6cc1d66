// my_slam/vo/keyframe.h -- what VisualOdometry::addFrame does when it inserts a keyframe while tracking
// (reference src/vo/vo_addFrame.cpp:93-118): match against the reference keyframe, keep the matches that satisfy the
// epipolar constraint, triangulate them with the known poses, drop the badly conditioned ones
// (VisualOdometry::retainGoodTriangulationResult_, src/vo/vo.cpp:181-244).
#ifndef MY_SLAM_KEYFRAME_H
#define MY_SLAM_KEYFRAME_H
#include "my_slam/geometry/feature_match.h"
#include "my_slam/geometry/motion_estimation.h"
#include "my_slam/vo/frame.h"
#include "my_slam/vo/pnp_tracking.h"

namespace my_slam {
namespace vo {

// vo_commons.cpp:9-15: T_f1_to_f2 = T_w_to_f1.inv() * T_w_to_f2
inline cv::Mat getMotionFromFrame1to2(const Frame::Ptr f1, const Frame::Ptr f2) {
    const cv::Mat Ti = basics::invT(f1->T_w_c_);
    cv::Mat T(4, 4, CV_64FC1);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Ti.at<double>(i, k) * f2->T_w_c_.at<double>(k, j);
            T.at<double>(i, j) = s;
        }
    return T;
}

// VisualOdometry::retainGoodTriangulationResult_ (vo.cpp:181-244) on curr (inliers_matches_with_ref_, inliers_pts3d_)
inline void retainGoodTriangulationResult(const Frame::Ptr& curr, const Frame::Ptr& ref) {
    static const double min_triang_angle = basics::Config::get<double>("min_triang_angle");
    static const double max_ratio_between_max_angle_and_median_angle =
        basics::Config::get<double>("max_ratio_between_max_angle_and_median_angle");
    const int N = (int)curr->inliers_pts3d_.size();
    if (N == 0) return;
    double Tc[16], Tr[16];
    for (int i = 0; i < 16; ++i) {
        Tc[i] = curr->T_w_c_.at<double>(i / 4, i % 4);
        Tr[i] = ref->T_w_c_.at<double>(i / 4, i % 4);
    }
    vector<int> keep(N);
    vector<double> all_angles(N);
    int n_keep = 0;
    if (mvo_retain_good_triangulation(&curr->inliers_pts3d_[0].x, N, Tc, Tr, min_triang_angle,
                                      max_ratio_between_max_angle_and_median_angle, keep.data(), &n_keep,
                                      all_angles.data()) != MVO_OK)
        throw std::runtime_error("retainGoodTriangulationResult: bad arguments");
    vector<cv::Point3f> old_inlier_points = curr->inliers_pts3d_;
    curr->inliers_pts3d_.clear();
    vector<double>& angles = curr->triangulation_angles_of_inliers_;
    angles.clear();
    for (int q = 0; q < n_keep; ++q) {
        const int i = keep[q];
        curr->inliers_matches_for_3d_.push_back(curr->inliers_matches_with_ref_[i]);
        curr->inliers_pts3d_.push_back(old_inlier_points[i]);
        angles.push_back(all_angles[i]);
    }
}

// vo_addFrame.cpp:96-118 up to (not including) pushCurrPointsToMap_
inline void triangulateWithReferenceKeyframe(const Frame::Ptr& curr, const Frame::Ptr& ref, const cv::Mat& K) {
    static const float max_matching_pixel_dist_in_triangulation =
        basics::Config::get<float>("max_matching_pixel_dist_in_triangulation");
    static const int method_index = (int)basics::Config::get<float>("feature_match_method_index_pnp");
    geometry::matchFeatures(ref->descriptors_, curr->descriptors_, curr->matches_with_ref_, method_index, false,
                            ref->keypoints_, curr->keypoints_, max_matching_pixel_dist_in_triangulation);
    curr->inliers_matches_with_ref_ =
        geometry::helperFindInlierMatchesByEpipolarCons(ref->keypoints_, curr->keypoints_, curr->matches_with_ref_, K);
    curr->inliers_pts3d_ = geometry::helperTriangulatePoints(ref->keypoints_, curr->keypoints_, curr->inliers_matches_with_ref_,
                                                             getMotionFromFrame1to2(curr, ref), K);
    retainGoodTriangulationResult(curr, ref);
}

}  // namespace vo
}  // namespace my_slam
#endif
