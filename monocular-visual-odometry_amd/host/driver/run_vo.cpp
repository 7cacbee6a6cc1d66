// host/driver/run_vo.cpp -- headless run_vo: the reference's main loop (run_vo.cpp:58-154) without the displays, over the
// reference's own inputs and outputs: config.yaml (dataset section, camera intrinsics, max_num_imgs_to_proc,
// save_predicted_traj_to), `<dataset_dir>/rgb_%05d.png`, and the 12-number-per-line trajectory file
// (src/vo/vo_io.cpp:13-117).  Every frame goes through the hot path on the MI355X: Frame::calcKeyPoints /
// calcDescriptors, map points in view, matchFeatures, solvePnPRansac, the sliding-window bundle adjustment, and on
// keyframes the epipolar inlier filter + triangulation + culling (my_slam/vo/tracking_loop.h = the DOING_TRACKING branch of
// VisualOdometry::addFrame, vo_addFrame.cpp:70-124).
//
// What is NOT here is the reference's INITIALIZATION state (essential / homography model selection, recoverPose --
// SURVEY.md section 2 marks it out of scope): the map is seeded instead from two frames whose poses are taken from the
// dataset's ground-truth trajectory (`true_traj_filename`, config.yaml:24; keys `init_keyframe_0/1` pick the frames,
// default 0 and 5): their matches are filtered and triangulated exactly as a keyframe insertion does it
// (vo_addFrame.cpp:96-118), which also fixes the scale.  From then on nothing of the ground truth is used.
// Optional key `save_frame_log_to`: a binary per-frame record of what the rows produced (keypoints, descriptors, the map's
// iteration order, inlier matches, keyframe products, pose) for tests/test_gpu_run_vo.py, which composes the same run from
// the oracle and compares stage by stage.
//   run_vo <config.yaml>
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "my_slam/basics/image_io.h"
#include "my_slam/vo/tracking_loop.h"
#include "my_slam/vo/vo_io.h"

using namespace my_slam;

namespace {
// record = 4-character tag, int64 byte count, payload; a frame starts with "FRAM"
struct FrameLog {
    FILE* f = nullptr;
    ~FrameLog() {
        if (f) fclose(f);
    }
    void put(const char* tag, const void* p, size_t bytes) {
        if (!f) return;
        const long long n = (long long)bytes;
        fwrite(tag, 1, 4, f);
        fwrite(&n, 8, 1, f);
        if (bytes) fwrite(p, 1, bytes, f);
    }
    template <class T>
    void vec(const char* tag, const vector<T>& v) {
        put(tag, v.data(), v.size() * sizeof(T));
    }
    void frame(int img_id, const vo::Frame::Ptr& fr) {
        static_assert(sizeof(cv::KeyPoint) == 28 && sizeof(cv::DMatch) == 16, "record layouts");
        const int head[2] = {img_id, fr->id_};
        put("FRAM", head, sizeof head);
        vec("KPTS", fr->keypoints_);
        put("DESC", fr->descriptors_.data, (size_t)fr->descriptors_.rows * 32);
    }
    void tracked(const vo::TrackingState& st, const vo::Frame::Ptr& fr, bool good, bool is_keyframe) {
        vector<int> order;
        for (const vo::MapPoint::Ptr& p : st.dev_map_.order()) order.push_back(p->id_);
        vec("MORD", order);  // iteration order of Map::map_points_ when the frame looked at the map
        vec("MMAP", fr->matches_with_map_);  // the PnP inliers' matches (vo.cpp:336-349)
        const int flags[2] = {good ? 1 : 0, is_keyframe ? 1 : 0};
        put("FLAG", flags, sizeof flags);
    }
    void keyframe(const vo::Map::Ptr& map, const vo::Frame::Ptr& fr) {
        vec("MREF", fr->matches_with_ref_);
        vec("IREF", fr->inliers_matches_with_ref_);
        vec("I3DM", fr->inliers_matches_for_3d_);
        vec("I3DP", fr->inliers_pts3d_);
        vector<int> ids;
        vector<float> pos;
        for (auto& kv : map->map_points_) {
            ids.push_back(kv.first);
            pos.push_back(kv.second->pos_.x);
            pos.push_back(kv.second->pos_.y);
            pos.push_back(kv.second->pos_.z);
        }
        vec("MIDS", ids);  // the map after the insertion (+ culling)
        vec("MPOS", pos);
    }
    void pose(const vo::Frame::Ptr& fr) {
        double T[16];
        for (int i = 0; i < 16; ++i) T[i] = fr->T_w_c_.at<double>(i / 4, i % 4);
        put("POSE", T, sizeof T);
    }
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: run_vo <config.yaml>\n");
        return 2;
    }
    try {
        basics::Config::setParameterFile(argv[1]);
        const string dataset_name = basics::Config::get<string>("dataset_name");
        const string sec = dataset_name + ".";
        const string dataset_dir = basics::Config::get<string>(sec + "dataset_dir");
        const int num_images = basics::Config::get<int>(sec + "num_images");
        const vector<string> image_paths = vo::readImagePaths(dataset_dir, num_images, "/rgb_%05d.png");
        const cv::Mat K = vo::readCameraIntrinsics(dataset_name);
        const int max_num_imgs_to_proc = basics::Config::get<int>("max_num_imgs_to_proc");
        const vector<cv::Mat> truth = vo::readPoseFromFile(basics::Config::get<string>(sec + "true_traj_filename"));
        const int k0 = basics::Config::has("init_keyframe_0") ? basics::Config::get<int>("init_keyframe_0") : 0;
        const int k1 = basics::Config::has("init_keyframe_1") ? basics::Config::get<int>("init_keyframe_1") : 5;
        if (k0 < 0 || k1 <= k0 || k1 >= (int)truth.size()) throw std::runtime_error("init_keyframe_0/1 outside the ground-truth trajectory");

        FrameLog log;
        if (basics::Config::has("save_frame_log_to")) {
            log.f = fopen(basics::Config::get<string>("save_frame_log_to").c_str(), "wb");
            if (!log.f) throw std::runtime_error("cannot open save_frame_log_to");
        }
        vo::TrackingState st;
        vector<cv::Mat> cam_pose_history;
        int n_tracked = 0, n_lost = 0, n_keyframes = 0;
        const int n_proc = std::min(max_num_imgs_to_proc, (int)image_paths.size());
        for (int img_id = 0; img_id < n_proc; img_id++) {
            cv::Mat rgb_img = basics::imread(image_paths[img_id]);
            if (rgb_img.data == nullptr) {
                printf("The image file %s is empty. Finished.\n", image_paths[img_id].c_str());
                break;
            }
            vo::Frame::Ptr frame = vo::Frame::createFrame(rgb_img);
            frame->calcKeyPoints();  // vo_addFrame.cpp:24-25
            frame->calcDescriptors();
            log.frame(img_id, frame);
            if (img_id < k1) {
                // before the map exists: pose = the last known one (the reference's INITIALIZATION keeps the first pose)
                frame->T_w_c_ = (img_id >= k0 ? truth[k0] : cv::Mat::eye(4, 4, CV_64FC1)).clone();
                if (img_id == k0) {
                    st.pushFrameToBuff(frame);
                    st.map_->insertKeyFrame(frame);
                    st.ref_ = st.prev_ = frame;
                }
            } else if (img_id == k1) {
                // seed the map: the two keyframes get their ground-truth poses, their matches are filtered, triangulated
                // and culled like any keyframe insertion (vo_addFrame.cpp:96-118), then pushed to the map (vo.cpp:528-576)
                frame->T_w_c_ = truth[k1].clone();
                st.pushFrameToBuff(frame);
                vo::triangulateWithReferenceKeyframe(frame, st.ref_, K);
                vo::pushCurrPointsToMap(st, frame);
                st.map_->insertKeyFrame(frame);
                log.keyframe(st.map_, frame);
                st.ref_ = st.prev_ = frame;
                printf("map seeded from frames %d and %d: %d map points\n", k0, k1, (int)st.map_->map_points_.size());
                if (st.map_->map_points_.size() < 10) throw std::runtime_error("too few map points after seeding");
            } else {
                bool is_keyframe = false;
                const bool good = vo::trackFrame(st, frame, K, &is_keyframe);
                n_tracked += good ? 1 : 0;
                n_lost += good ? 0 : 1;
                n_keyframes += is_keyframe ? 1 : 0;
                log.tracked(st, frame, good, is_keyframe);
                if (is_keyframe) log.keyframe(st.map_, frame);
            }
            log.pose(frame);
            cam_pose_history.push_back(frame->T_w_c_.clone());  // run_vo.cpp:139-142
            frame->clearNoUsed();
        }
        const string save_predicted_traj_to = basics::Config::get<string>("save_predicted_traj_to");
        vo::writePoseToFile(save_predicted_traj_to, cam_pose_history);
        printf("frames %d, tracked %d, lost %d, keyframes %d, map points %d -> %s\n", (int)cam_pose_history.size(), n_tracked, n_lost,
               n_keyframes, (int)st.map_->map_points_.size(), save_predicted_traj_to.c_str());
    } catch (const std::exception& e) {
        fprintf(stderr, "run_vo: %s\n", e.what());
        return 1;
    }
    return 0;
}
