// my_slam/vo/pnp_tracking.h -- the tracking step between matching and bundle adjustment, with the reference's
// names and bookkeeping, executed by libmvo_hip.so:
//   geometry::solvePnPRansac / geometry::Rodrigues   the two cv:: calls of src/vo/vo.cpp:326-329 and :334 (also
//                                                     visible as cv::solvePnPRansac / cv::Rodrigues in builds
//                                                     without OpenCV, so vo.cpp compiles unchanged there)
//   vo::MapOnDevice                                  the map's positions + descriptors resident in HBM
//   vo::getMappointsInCurrentView                    VisualOdometry::getMappointsInCurrentView_ (vo.cpp:16-49)
//   vo::poseEstimationPnP                            VisualOdometry::poseEstimationPnP_ (vo.cpp:270-383)
#ifndef MY_SLAM_PNP_TRACKING_H
#define MY_SLAM_PNP_TRACKING_H
#include <algorithm>

#include "my_slam/basics/config.h"
#include "my_slam/geometry/feature_match.h"
#include "my_slam/vo/frame.h"
#include "my_slam/vo/map.h"

namespace my_slam {
namespace geometry {

// cv::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs (must be empty), rvec, tvec,
//                    useExtrinsicGuess (must be false), iterationsCount, reprojectionError, confidence, inliers)
// rvec / tvec: 3x1 CV_64F; inliers as vector<int> (the cv::Mat form of vo.cpp:322, type 32SC1, wraps this one).
inline bool solvePnPRansac(const vector<cv::Point3f>& objectPoints, const vector<cv::Point2f>& imagePoints,
                           const cv::Mat& cameraMatrix, cv::Mat& rvec, cv::Mat& tvec, int iterationsCount,
                           float reprojectionError, double confidence, vector<int>& inliers) {
    if (objectPoints.size() != imagePoints.size()) throw std::runtime_error("solvePnPRansac: size mismatch");
    const int n = (int)objectPoints.size();
    rvec.create(3, 1, CV_64FC1);
    tvec.create(3, 1, CV_64FC1);
    inliers.assign(n > 0 ? n : 1, 0);
    int n_inl = 0, found = 0;
    static_assert(sizeof(cv::Point3f) == 12 && sizeof(cv::Point2f) == 8, "cv::Point layouts");
    mvo_check(mvo_solve_pnp_ransac(hot_path_ctx(), n ? &objectPoints[0].x : nullptr, n ? &imagePoints[0].x : nullptr, n,
                                   cameraMatrix.at<double>(0, 0), cameraMatrix.at<double>(1, 1),
                                   cameraMatrix.at<double>(0, 2), cameraMatrix.at<double>(1, 2), iterationsCount,
                                   reprojectionError, confidence, rvec.ptr<double>(0), tvec.ptr<double>(0), inliers.data(),
                                   (int)inliers.size(), &n_inl, &found),
              "solvePnPRansac");
    inliers.resize(n_inl);
    return found != 0;
}

inline void Rodrigues(const cv::Mat& rvec, cv::Mat& R) {
    R.create(3, 3, CV_64FC1);
    const double r[3] = {rvec.ptr<double>(0)[0], rvec.ptr<double>(1)[0], rvec.ptr<double>(2)[0]};
    double out[9];
    if (mvo_rodrigues(r, out) != MVO_OK) throw std::runtime_error("Rodrigues: bad arguments");
    for (int i = 0; i < 9; ++i) R.at<double>(i / 3, i % 3) = out[i];
}

}  // namespace geometry

namespace basics {
// basics::convertRt2T (src/basics/opencv_funcs.cpp) and cv::Mat::inv() of a 4x4 pose
inline cv::Mat convertRt2T(const cv::Mat& R, const cv::Mat& t) {
    cv::Mat T = cv::Mat::eye(4, 4, CV_64FC1);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T.at<double>(i, j) = R.at<double>(i, j);
        T.at<double>(i, 3) = t.ptr<double>(i)[0];
    }
    return T;
}
inline cv::Mat invT(const cv::Mat& T) {
    cv::Mat Ti(4, 4, CV_64FC1);
    double in[16];
    for (int i = 0; i < 16; ++i) in[i] = T.at<double>(i / 4, i % 4);
    if (mvo_invert_pose(in, Ti.ptr<double>(0)) != MVO_OK) throw std::runtime_error("inv: singular pose matrix");
    return Ti;
}
}  // namespace basics

namespace vo {

// The map mirrored in HBM.  sync() re-reads Map::map_points_ in ITS iteration order (the order every index below
// refers to): descriptors travel only when the set of points changed, positions every time (bundle adjustment
// moves them through raw pointers, vo.cpp:448).
class MapOnDevice {
public:
    ~MapOnDevice() {
        if (map_) mvo_map_release(hot_path_ctx(), map_);
    }
    void sync(const Map::Ptr& map) {
        if (!map_) mvo_check(mvo_map_create(hot_path_ctx(), &map_), "mvo_map_create");
        const size_t n = map->map_points_.size();
        bool same = n == order_.size();
        size_t i = 0;
        for (auto& kv : map->map_points_) {
            if (same && order_[i]->id_ != kv.second->id_) same = false;
            if (!same) break;
            ++i;
        }
        pos_.resize(3 * n);
        if (!same) {
            order_.clear();
            desc_.resize(32 * n);
            i = 0;
            for (auto& kv : map->map_points_) {
                order_.push_back(kv.second);
                std::memcpy(&desc_[32 * i], kv.second->descriptor_.data, 32);
                ++i;
            }
        }
        for (i = 0; i < n; ++i) {
            pos_[3 * i] = order_[i]->pos_.x;
            pos_[3 * i + 1] = order_[i]->pos_.y;
            pos_[3 * i + 2] = order_[i]->pos_.z;
        }
        if (!same)
            mvo_check(mvo_map_upload(hot_path_ctx(), map_, pos_.data(), desc_.data(), (int)n), "mvo_map_upload");
        else
            mvo_check(mvo_map_update_positions(hot_path_ctx(), map_, pos_.data(), 0, (int)n), "mvo_map_update_positions");
    }
    mvo_map* handle() const { return map_; }
    const vector<MapPoint::Ptr>& order() const { return order_; }
    const unsigned char* descriptor(int i) const { return &desc_[32 * (size_t)i]; }

private:
    mvo_map* map_ = nullptr;
    vector<MapPoint::Ptr> order_;
    vector<float> pos_;
    vector<unsigned char> desc_;
};

// VisualOdometry::getMappointsInCurrentView_ (vo.cpp:16-49).  Same three outputs (+ visible_times_++); the
// descriptors of the candidates additionally stay in HBM at *d_descriptors (may be null) for the matcher.
inline void getMappointsInCurrentView(MapOnDevice& dev_map, const Map::Ptr& map, const Frame::Ptr& curr, const cv::Mat& K,
                                      vector<MapPoint::Ptr>& candidate_mappoints_in_map,
                                      vector<cv::Point2f>& candidate_2d_pts_in_image,
                                      cv::Mat& corresponding_mappoints_descriptors, const void** d_descriptors = nullptr) {
    candidate_mappoints_in_map.clear();
    dev_map.sync(map);
    const int m = (int)dev_map.order().size();
    vector<int> idx(m > 0 ? m : 1);
    vector<cv::Point2f> px(m > 0 ? m : 1);
    int n = 0;
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = curr->T_w_c_.at<double>(i / 4, i % 4);
    const void* d = nullptr;
    mvo_check(mvo_map_points_in_view(hot_path_ctx(), dev_map.handle(), T, K.at<double>(0, 0), K.at<double>(1, 1),
                                     K.at<double>(0, 2), K.at<double>(1, 2), curr->rgb_img_.cols, curr->rgb_img_.rows,
                                     idx.data(), &px[0].x, m, &n, &d),
              "getMappointsInCurrentView");
    if (d_descriptors) *d_descriptors = d;
    corresponding_mappoints_descriptors.create(n > 0 ? n : 1, 32, CV_8UC1);
    corresponding_mappoints_descriptors.rows = n;
    for (int i = 0; i < n; ++i) {
        const MapPoint::Ptr& p_world = dev_map.order()[idx[i]];
        candidate_mappoints_in_map.push_back(p_world);
        candidate_2d_pts_in_image.push_back(px[i]);  // (the reference does not clear this vector either)
        std::memcpy(corresponding_mappoints_descriptors.ptr<unsigned char>(i), dev_map.descriptor(idx[i]), 32);
        p_world->visible_times_++;
    }
}

// VisualOdometry::poseEstimationPnP_ (vo.cpp:270-383): matches the map points in view against the frame, solves
// PnP with RANSAC, records the inlier connections and sets curr->T_w_c_.  Returns is_pnp_good.
inline bool poseEstimationPnP(MapOnDevice& dev_map, const Map::Ptr& map, const Frame::Ptr& curr, const Frame::Ptr& prev,
                              const cv::Mat& K) {
    vector<MapPoint::Ptr> candidate_mappoints_in_map;
    vector<cv::Point2f> candidate_2d_pts_in_image;
    cv::Mat corresponding_mappoints_descriptors;
    getMappointsInCurrentView(dev_map, map, curr, K, candidate_mappoints_in_map, candidate_2d_pts_in_image,
                              corresponding_mappoints_descriptors);
    vector<cv::KeyPoint> candidate_2d_kpts_in_image;  // geometry::pts2Keypts (feature_match.cpp:293-303)
    for (const cv::Point2f& pt : candidate_2d_pts_in_image) candidate_2d_kpts_in_image.push_back(cv::KeyPoint(pt, 10));

    static const float max_matching_pixel_dist_in_pnp = basics::Config::get<float>("max_matching_pixel_dist_in_pnp");
    static const int method_index = (int)basics::Config::get<float>("feature_match_method_index_pnp");
    geometry::matchFeatures(corresponding_mappoints_descriptors, curr->descriptors_, curr->matches_with_map_, method_index,
                            false, candidate_2d_kpts_in_image, curr->keypoints_, max_matching_pixel_dist_in_pnp);
    const int num_matches = (int)curr->matches_with_map_.size();
    vector<cv::Point3f> pts_3d;
    vector<cv::Point2f> pts_2d;
    for (int i = 0; i < num_matches; i++) {
        const cv::DMatch& match = curr->matches_with_map_[i];
        pts_3d.push_back(candidate_mappoints_in_map[match.queryIdx]->pos_);
        pts_2d.push_back(curr->keypoints_[match.trainIdx].pt);
    }
    constexpr int kMinPtsForPnP = 5;
    static const double max_possible_dist_to_prev_keyframe = basics::Config::get<double>("max_possible_dist_to_prev_keyframe");
    bool is_pnp_good = num_matches >= kMinPtsForPnP;
    if (is_pnp_good) {
        vector<int> pnp_inliers;
        cv::Mat R_vec, t, R;
        const bool found = geometry::solvePnPRansac(pts_3d, pts_2d, K, R_vec, t, 100, 2.0f, 0.999, pnp_inliers);
        if (!found) throw std::runtime_error("solvePnPRansac found no pose (cv::Rodrigues would throw on the empty rvec)");
        geometry::Rodrigues(R_vec, R);
        vector<cv::DMatch> tmp_matches_with_map;
        for (int good_idx : pnp_inliers) {
            const cv::DMatch& match = curr->matches_with_map_[good_idx];
            tmp_matches_with_map.push_back(match);
            const MapPoint::Ptr& inlier_mappoint = candidate_mappoints_in_map[match.queryIdx];
            inlier_mappoint->matched_times_++;
            curr->inliers_to_mappt_connections_[match.trainIdx] = PtConn{-1, inlier_mappoint->id_};
        }
        curr->matches_with_map_.swap(tmp_matches_with_map);
        curr->T_w_c_ = basics::invT(basics::convertRt2T(R, t));
        double d2 = 0;  // basics::calcMatNorm(t_curr - t_prev), vo.cpp:352-356
        for (int i = 0; i < 3; ++i) {
            const double d = curr->T_w_c_.at<double>(i, 3) - prev->T_w_c_.at<double>(i, 3);
            d2 += d * d;
        }
        if (std::sqrt(d2) >= max_possible_dist_to_prev_keyframe) is_pnp_good = false;
    }
    if (!is_pnp_good) curr->T_w_c_ = prev->T_w_c_.clone();
    return is_pnp_good;
}

}  // namespace vo
}  // namespace my_slam

#ifndef MVO_HAVE_OPENCV
namespace cv {
// The exact call of vo.cpp:326-329 for builds without OpenCV (inliers: n x 1 matrix of int, read with at<int>(i, 0)).
inline bool solvePnPRansac(const std::vector<Point3f>& objectPoints, const std::vector<Point2f>& imagePoints,
                           const Mat& cameraMatrix, const Mat& distCoeffs, Mat& rvec, Mat& tvec, bool useExtrinsicGuess,
                           int iterationsCount, float reprojectionError, double confidence, Mat& inliers) {
    if (!distCoeffs.empty() || useExtrinsicGuess)
        throw std::runtime_error("solvePnPRansac: distortion / extrinsic guess are outside the reference's use (vo.cpp:323-329)");
    std::vector<int> inl;
    const bool ok = my_slam::geometry::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, rvec, tvec, iterationsCount,
                                                      reprojectionError, confidence, inl);
    inliers.create((int)inl.size() > 0 ? (int)inl.size() : 1, 1, CV_32SC1);  // "type = 32SC1" (vo.cpp:322)
    inliers.rows = (int)inl.size();
    for (size_t i = 0; i < inl.size(); ++i) inliers.at<int>((int)i, 0) = inl[i];
    return ok;
}
inline void Rodrigues(const Mat& src, Mat& dst) { my_slam::geometry::Rodrigues(src, dst); }
}  // namespace cv
#endif
#endif
