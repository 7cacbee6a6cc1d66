// my_slam/geometry/motion_estimation.h -- the two helpers VisualOdometry calls when it inserts a keyframe
// (reference include/my_slam/geometry/motion_estimation.h, src/geometry/motion_estimation.cpp:182-247, call sites
// src/vo/vo_addFrame.cpp:104-116), same names and signatures, executed by libmvo_hip.so:
//   helperFindInlierMatchesByEpipolarCons  inlier mask of cv::findEssentialMat(RANSAC) (epipolar_geometry.cpp:17-47)
//   helperTriangulatePoints                pixel2CamNormPlane + cv::triangulatePoints + transCoord
#ifndef MY_SLAM_MOTION_ESTIMATION_H
#define MY_SLAM_MOTION_ESTIMATION_H
#include "my_slam/basics/config.h"
#include "my_slam/common_include.h"

namespace my_slam {
namespace geometry {

// geometry::extractPtsFromMatches (feature_match.cpp): queryIdx indexes points_1, trainIdx points_2
inline void extractPtsFromMatches(const vector<cv::KeyPoint>& keypoints_1, const vector<cv::KeyPoint>& keypoints_2,
                                  const vector<cv::DMatch>& matches, vector<cv::Point2f>& pts1, vector<cv::Point2f>& pts2) {
    pts1.clear();
    pts2.clear();
    for (const cv::DMatch& m : matches) {
        pts1.push_back(keypoints_1[m.queryIdx].pt);
        pts2.push_back(keypoints_2[m.trainIdx].pt);
    }
}

inline vector<cv::DMatch> helperFindInlierMatchesByEpipolarCons(const vector<cv::KeyPoint>& keypoints_1,
                                                                const vector<cv::KeyPoint>& keypoints_2,
                                                                const vector<cv::DMatch>& matches, const cv::Mat& K) {
    // epipolar_geometry.cpp:31-32: latched on first use
    static const double findEssentialMat_prob = basics::Config::get<double>("findEssentialMat_prob");
    static const double findEssentialMat_threshold = basics::Config::get<double>("findEssentialMat_threshold");
    vector<cv::Point2f> pts1, pts2;
    extractPtsFromMatches(keypoints_1, keypoints_2, matches, pts1, pts2);
    const int n = (int)pts1.size();
    vector<int> inl(n > 0 ? n : 1);
    int cnt = 0;
    mvo_check(mvo_find_essential_inliers(hot_path_ctx(), n ? &pts1[0].x : nullptr, n ? &pts2[0].x : nullptr, n,
                                         K.at<double>(0, 0), K.at<double>(1, 1), K.at<double>(0, 2), K.at<double>(1, 2),
                                         findEssentialMat_prob, findEssentialMat_threshold, inl.data(), (int)inl.size(), &cnt),
              "helperFindInlierMatchesByEpipolarCons");
    vector<cv::DMatch> inlier_matches;
    for (int i = 0; i < cnt; ++i) {  // motion_estimation.cpp:174-179
        const cv::DMatch& m = matches[inl[i]];
        inlier_matches.push_back(cv::DMatch(m.queryIdx, m.trainIdx, m.distance));
    }
    return inlier_matches;
}

inline vector<cv::Point3f> helperTriangulatePoints(const vector<cv::KeyPoint>& prev_kpts, const vector<cv::KeyPoint>& curr_kpts,
                                                   const vector<cv::DMatch>& curr_inlier_matches,
                                                   const cv::Mat& R_curr_to_prev, const cv::Mat& t_curr_to_prev,
                                                   const cv::Mat& K) {
    vector<cv::Point2f> pts1, pts2;
    extractPtsFromMatches(prev_kpts, curr_kpts, curr_inlier_matches, pts1, pts2);
    const int n = (int)pts1.size();
    vector<cv::Point3f> pts_3d_in_curr(n);
    double R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = R_curr_to_prev.at<double>(i / 3, i % 3);
    for (int i = 0; i < 3; ++i) t[i] = t_curr_to_prev.ptr<double>(i)[0];
    mvo_check(mvo_triangulate_points(hot_path_ctx(), n ? &pts1[0].x : nullptr, n ? &pts2[0].x : nullptr, n, K.at<double>(0, 0),
                                     K.at<double>(1, 1), K.at<double>(0, 2), K.at<double>(1, 2), R, t, nullptr,
                                     n ? &pts_3d_in_curr[0].x : nullptr),
              "helperTriangulatePoints");
    return pts_3d_in_curr;
}

inline vector<cv::Point3f> helperTriangulatePoints(const vector<cv::KeyPoint>& prev_kpts, const vector<cv::KeyPoint>& curr_kpts,
                                                   const vector<cv::DMatch>& curr_inlier_matches, const cv::Mat& T_curr_to_prev,
                                                   const cv::Mat& K) {
    cv::Mat R(3, 3, CV_64FC1), t(3, 1, CV_64FC1);  // basics::getRtFromT
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R.at<double>(i, j) = T_curr_to_prev.at<double>(i, j);
        t.at<double>(i, 0) = T_curr_to_prev.at<double>(i, 3);
    }
    return helperTriangulatePoints(prev_kpts, curr_kpts, curr_inlier_matches, R, t, K);
}

}  // namespace geometry
}  // namespace my_slam
#endif
