// my_slam/geometry/feature_match.h -- OpenCV-less MIRROR of the reference's include/my_slam/geometry/feature_match.h:12-54 for
// this repo's tests and driver: the same nine declarations.  They are defined in host/src/feature_match_mvo.cpp, the
// translation unit that replaces src/geometry/feature_match.cpp in a reference build -- there it is compiled against the
// reference's own header; this file is NOT put on the reference's include path.
#ifndef MY_SLAM_FEATURE_MATCH_H
#define MY_SLAM_FEATURE_MATCH_H
#include "my_slam/basics/config.h"
#include "my_slam/common_include.h"

namespace my_slam {
namespace geometry {

namespace detail {
// (mirror only: Frame::calcKeyPoints / calcDescriptors of this repo's Frame tell the translation unit that the pyramid cached
// in the ctx belongs to the same image, so that calcDescriptors does not build it a second time as the reference does)
bool& reuse_pyramid_flag();
long long& pyramid_token();
}  // namespace detail

void calcKeyPoints(const cv::Mat& image, vector<cv::KeyPoint>& keypoints);

/* Compute the descriptors of keypoints. Meanwhile, keypoints might be changed (feature_match.h:15-17). */
void calcDescriptors(const cv::Mat& image, vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors);

void matchFeatures(const cv::Mat1b& descriptors_1, const cv::Mat1b& descriptors_2, vector<cv::DMatch>& matches,
                   int method_index = 1, bool is_print_res = false,
                   // Below are optional arguments for feature_matching_method_index==3
                   const vector<cv::KeyPoint>& keypoints_1 = vector<cv::KeyPoint>(),
                   const vector<cv::KeyPoint>& keypoints_2 = vector<cv::KeyPoint>(), float max_matching_pixel_dist = 0.0);

vector<cv::DMatch> matchByRadiusAndBruteForce(const vector<cv::KeyPoint>& keypoints_1, const vector<cv::KeyPoint>& keypoints_2,
                                              const cv::Mat1b& descriptors_1, const cv::Mat1b& descriptors_2,
                                              float max_matching_pixel_dist);

// Remove duplicate matches: sort by trainIdx, keep the first of every run (feature_match.cpp:241-260).
void removeDuplicatedMatches(vector<cv::DMatch>& matches);

// Use a grid to remove the keypoints that are too close to each other.
void selectUniformKptsByGrid(vector<cv::KeyPoint>& keypoints, int image_rows, int image_cols);

// --------------------- Other assistant functions ---------------------
double computeMeanDistBetweenKeypoints(const vector<cv::KeyPoint>& kpts1, const vector<cv::KeyPoint>& kpts2,
                                       const vector<cv::DMatch>& matches);

// --------------------- Datatype conversion ---------------------
vector<cv::DMatch> inliers2DMatches(const vector<int> inliers);
vector<cv::KeyPoint> pts2Keypts(const vector<cv::Point2f> pts);

}  // namespace geometry
}  // namespace my_slam
#endif
