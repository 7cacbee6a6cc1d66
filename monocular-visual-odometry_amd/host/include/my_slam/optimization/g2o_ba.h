// my_slam/optimization/g2o_ba.h -- OpenCV-less MIRROR of the reference's include/my_slam/optimization/g2o_ba.h:16-30 for
// this repo's tests and driver: the same two declarations (defined in host/src/g2o_ba_mvo.cpp, the translation unit that
// replaces src/optimization/g2o_ba.cpp in a reference build -- there it is compiled against the reference's own header, this
// file is NOT put on the reference's include path), plus the two-halves form this repo's frame loop uses.
#ifndef MY_SLAM_G2O_BA_H
#define MY_SLAM_G2O_BA_H
#include "my_slam/common_include.h"

#include "flat_bundle.h"  // (host/src)

namespace my_slam {
namespace optimization {

void optimizeSingleFrame(const vector<cv::Point2f*>& points_2d, const cv::Mat& K, vector<cv::Point3f*>& points_3d,
                         cv::Mat& cam_pose_in_world, bool is_fix_map_pts, bool is_update_map_pts);

void bundleAdjustment(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx,
                      const cv::Mat& K, std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses,
                      const cv::Mat& information_matrix, bool is_fix_map_pts = false, bool is_update_map_pts = true);

// The same call in two halves for drivers that overlap the solve with other host work (e.g. the extraction of the
// next frame): begin() = graph -> flat window -> upload -> launch; end() = wait + write-back.  The pointer lists must
// stay valid in between.  `last` keeps the statistics of the finished solve.
class BundleAdjustmentJob {
public:
    BundleAdjustmentJob() = default;
    BundleAdjustmentJob(const BundleAdjustmentJob&) = delete;
    BundleAdjustmentJob& operator=(const BundleAdjustmentJob&) = delete;
    // a job that goes out of scope between its two halves (exception, early return) still owns a launch that writes into
    // the ctx's pooled workspace: wait for it; the results are dropped, errors cannot leave a destructor
    ~BundleAdjustmentJob() {
        if (!active_) return;
        active_ = false;
        (void)mvo_bundle_adjustment_end(ctx_, nullptr, nullptr);
    }
    void begin(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx, const cv::Mat& K,
               std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses,
               const cv::Mat& information_matrix, bool is_fix_map_pts = false, bool is_update_map_pts = true) {
        fb_.flatten(v_pts_2d, v_pts_2d_to_3d_idx, K, pts_3d, v_camera_g2o_poses, information_matrix, is_fix_map_pts);
        pts_3d_ = &pts_3d;
        poses_ = &v_camera_g2o_poses;
        update_ = is_update_map_pts;
        ctx_ = hot_path_ctx();
        mvo_check(mvo_bundle_adjustment_begin(ctx_, &fb_.pr), "bundleAdjustment (begin)");
        active_ = true;
    }
    void end() {
        if (!active_) return;
        active_ = false;
        mvo_check(mvo_bundle_adjustment_end(ctx_, &fb_.pr, &last), "bundleAdjustment (end)");
        fb_.scatter(*pts_3d_, *poses_, update_);
    }
    bool active() const { return active_; }
    mvo_ba_stats last{};

private:
    FlatBundle fb_;
    std::unordered_map<int, cv::Point3f*>* pts_3d_ = nullptr;
    vector<cv::Mat*>* poses_ = nullptr;
    mvo_ctx* ctx_ = nullptr;  // the ctx the launch was made on (the binding of the thread may change in between)
    bool update_ = true, active_ = false;
};

}  // namespace optimization
}  // namespace my_slam
#endif
