// my_slam/geometry/feature_match.h -- drop-in for the reference's include/my_slam/geometry/feature_match.h:12-46
// (implementation src/geometry/feature_match.cpp:11-260): the same free functions, same argument meaning, same
// latching of the parameters on first call, same exception on a wrong method index -- executed by libmvo_hip.so.
#ifndef MY_SLAM_FEATURE_MATCH_H
#define MY_SLAM_FEATURE_MATCH_H
#include "my_slam/basics/config.h"
#include "my_slam/common_include.h"

namespace my_slam {
namespace geometry {

namespace detail {
// feature_match.cpp:16-19, 42-45, 56-59: parameters are read once (function-local statics in the reference)
inline void latch_orb_params() {
    static bool done = false;
    if (done) return;
    mvo_orb_params p;
    p.nfeatures = basics::Config::get<int>("number_of_keypoints_to_extract");
    p.scale_factor = (float)basics::Config::get<double>("scale_factor");
    p.nlevels = basics::Config::get<int>("level_pyramid");
    p.fast_threshold = basics::Config::get<int>("score_threshold");
    p.max_keypoints = basics::Config::get<int>("max_number_of_keypoints");
    p.grid_size = basics::Config::get<int>("kpts_uniform_selection_grid_size");
    p.grid_max_per_cell = basics::Config::get<int>("kpts_uniform_selection_max_pts_per_grid");
    mvo_check(mvo_orb_configure(hot_path_ctx(), &p), "mvo_orb_configure");
    done = true;
}
// the pyramid built by calcKeyPoints may be reused by calcDescriptors when the caller guarantees it is the same
// image (Frame::calcKeyPoints / calcDescriptors do); the free functions rebuild it by default.
inline bool& reuse_pyramid_flag() {
    static thread_local bool f = false;
    return f;
}
// who owns the pyramid cached in the ctx (nullptr after any direct call of the free function)
inline const void*& pyramid_token() {
    static thread_local const void* t = nullptr;
    return t;
}
}  // namespace detail

inline void calcKeyPoints(const cv::Mat& image, vector<cv::KeyPoint>& keypoints) {
    detail::latch_orb_params();
    detail::pyramid_token() = nullptr;
    const int cap = basics::Config::get<int>("max_number_of_keypoints") + 16;
    keypoints.resize(cap);
    int n = 0;
    mvo_check(mvo_calc_keypoints(hot_path_ctx(), image.data, image.cols, image.rows, (int)image.step, image.channels(),
                                 reinterpret_cast<mvo_keypoint*>(keypoints.data()), cap, &n),
              "calcKeyPoints");
    keypoints.resize(n);
}

/* Compute the descriptors of keypoints. Meanwhile, keypoints might be changed (feature_match.h:15-17). */
inline void calcDescriptors(const cv::Mat& image, vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
    detail::latch_orb_params();
    int n = (int)keypoints.size();
    descriptors.create(n > 0 ? n : 1, 32, CV_8UC1);
    mvo_check(mvo_calc_descriptors(hot_path_ctx(), image.data, image.cols, image.rows, (int)image.step,
                                   image.channels(), detail::reuse_pyramid_flag() ? 1 : 0,
                                   reinterpret_cast<mvo_keypoint*>(keypoints.data()), &n, descriptors.data, nullptr),
              "calcDescriptors");
    keypoints.resize(n);
    descriptors.rows = n;
}

inline void removeDuplicatedMatches(vector<cv::DMatch>& matches) {
    int n = (int)matches.size();
    mvo_remove_duplicated_matches(reinterpret_cast<mvo_dmatch*>(matches.data()), &n);
    matches.resize(n);
}

inline void selectUniformKptsByGrid(vector<cv::KeyPoint>& keypoints, int image_rows, int image_cols) {
    detail::latch_orb_params();
    int n = (int)keypoints.size();
    mvo_check(mvo_select_uniform_kpts_by_grid(hot_path_ctx(), reinterpret_cast<mvo_keypoint*>(keypoints.data()), &n,
                                              image_rows, image_cols),
              "selectUniformKptsByGrid");
    keypoints.resize(n);
}

inline void matchFeatures(const cv::Mat1b& descriptors_1, const cv::Mat1b& descriptors_2, vector<cv::DMatch>& matches,
                          int method_index = 1, bool is_print_res = false,
                          // Below are optional arguments for feature_matching_method_index==3
                          const vector<cv::KeyPoint>& keypoints_1 = vector<cv::KeyPoint>(),
                          const vector<cv::KeyPoint>& keypoints_2 = vector<cv::KeyPoint>(),
                          float max_matching_pixel_dist = 0.0) {
    // feature_match.cpp:137-139: the three ratios are read with get<int>
    static const double xiang_gao_method_match_ratio = basics::Config::get<int>("xiang_gao_method_match_ratio");
    static const double lowe_method_dist_ratio = basics::Config::get<int>("lowe_method_dist_ratio");
    matches.clear();
    if (method_index < 1 || method_index > 3)
        throw std::runtime_error("feature_match.cpp::matchFeatures: wrong method index.");  // :225
    vector<float> xy1, xy2;
    if (method_index == 3) {
        for (const cv::KeyPoint& k : keypoints_1) {
            xy1.push_back(k.pt.x);
            xy1.push_back(k.pt.y);
        }
        for (const cv::KeyPoint& k : keypoints_2) {
            xy2.push_back(k.pt.x);
            xy2.push_back(k.pt.y);
        }
    }
    const int n1 = descriptors_1.rows, n2 = descriptors_2.rows;
    matches.resize(n1 > 0 ? n1 : 1);
    int n = 0;
    mvo_check(mvo_match_features(hot_path_ctx(), descriptors_1.data, n1, descriptors_2.data, n2, method_index,
                                 xiang_gao_method_match_ratio, lowe_method_dist_ratio, xy1.data(), xy2.data(),
                                 max_matching_pixel_dist, reinterpret_cast<mvo_dmatch*>(matches.data()),
                                 (int)matches.size(), &n),
              "matchFeatures");
    matches.resize(n);
    if (is_print_res) {
        printf("Matching features:\n");
        printf("Using method %d\n", method_index);
        printf("Number of matches: %d\n", int(matches.size()));
    }
}

}  // namespace geometry
}  // namespace my_slam
#endif
