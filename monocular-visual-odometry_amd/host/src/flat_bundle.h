// host/src/flat_bundle.h -- PRIVATE to the drop-in translation unit g2o_ba_mvo.cpp and to this repo's mirror headers: the graph
// optimization::bundleAdjustment hands to g2o (src/optimization/g2o_ba.cpp:193-271), flattened for mvo_bundle_adjustment, and
// the write-back of g2o_ba.cpp:298-316.  Only cv::Mat::at<double>, cv::Point2f / Point3f members are touched: compiles against
// real OpenCV and against the mirror's mini_cv.h alike.
#ifndef MVO_FLAT_BUNDLE_H
#define MVO_FLAT_BUNDLE_H
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "mvo_hip.h"

namespace my_slam {
namespace optimization {
using std::vector;

// The graph of g2o_ba.cpp:193-271 as flat arrays: pose vertices in the order of v_camera_g2o_poses, landmark vertices
// in the iteration order of the unordered_map (like the vertex ids of g2o_ba.cpp:225-243), one edge per observation.
struct FlatBundle {
    vector<double> poses, pts, uv;
    vector<int> ep, el, slot2id;
    vector<int> table_;  // (id -> slot, when the ids are dense)
    mvo_ba_problem pr{};

    void flatten(const vector<vector<cv::Point2f*>>& v_pts_2d, const vector<vector<int>>& v_pts_2d_to_3d_idx, const cv::Mat& K,
                 std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses,
                 const cv::Mat& information_matrix, bool is_fix_map_pts) {
        const int num_frames = (int)v_camera_g2o_poses.size();
        poses.resize(16 * (size_t)num_frames);
        for (int i = 0; i < num_frames; ++i)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) poses[16 * i + 4 * r + c] = v_camera_g2o_poses[i]->at<double>(r, c);
        // landmark id -> position in the flat list.  Map-point ids are small non-negative integers handed out by a counter
        // (mappoint.cpp:11): a direct table when they are dense enough, a hash map otherwise (9.4 k look-ups per BA5 window)
        const size_t npts = pts_3d.size();
        int max_id = -1, min_id = 0;
        for (const auto& kv : pts_3d) {
            max_id = kv.first > max_id ? kv.first : max_id;
            min_id = kv.first < min_id ? kv.first : min_id;
        }
        const bool direct = min_id >= 0 && (size_t)max_id < 8 * npts + 4096;
        std::unordered_map<int, int> id2slot;
        if (direct) table_.assign((size_t)max_id + 1, -1);
        else id2slot.reserve(npts);
        slot2id.clear();
        slot2id.reserve(npts);
        pts.clear();
        pts.reserve(3 * npts);
        for (auto it = pts_3d.begin(); it != pts_3d.end(); ++it) {
            if (direct) table_[(size_t)it->first] = (int)slot2id.size();
            else id2slot[it->first] = (int)slot2id.size();
            slot2id.push_back(it->first);
            pts.push_back(it->second->x);
            pts.push_back(it->second->y);
            pts.push_back(it->second->z);
        }
        size_t nobs = 0;
        for (int f = 0; f < num_frames; ++f) nobs += v_pts_2d[f].size();
        ep.clear();
        el.clear();
        uv.clear();
        ep.reserve(nobs);
        el.reserve(nobs);
        uv.reserve(2 * nobs);
        for (int f = 0; f < num_frames; ++f)
            for (size_t j = 0; j < v_pts_2d[f].size(); ++j) {
                const int id = v_pts_2d_to_3d_idx[f][j];
                int slot = -1;
                if (direct) slot = (id >= 0 && id <= max_id) ? table_[(size_t)id] : -1;
                else slot = id2slot.at(id);
                if (slot < 0) throw std::out_of_range("bundleAdjustment: an observation refers to a landmark that is not in pts_3d");
                ep.push_back(f);
                el.push_back(slot);
                uv.push_back(v_pts_2d[f][j]->x);
                uv.push_back(v_pts_2d[f][j]->y);
            }
        pr = mvo_ba_problem{};
        pr.n_poses = num_frames;
        pr.n_points = (int)slot2id.size();
        pr.n_edges = (int)ep.size();
        pr.pose_T_w_c = poses.data();
        pr.points = pts.data();
        pr.edge_pose = ep.data();
        pr.edge_point = el.data();
        pr.edge_uv = uv.data();
        pr.focal = K.at<double>(0, 0);  // CameraParameters(K(0,0), (K(0,2), K(1,2)), 0): fy is not used (:219-222)
        pr.cx = K.at<double>(0, 2);
        pr.cy = K.at<double>(1, 2);
        for (int i = 0; i < 4; ++i) pr.info[i] = information_matrix.at<double>(i / 2, i % 2);
        pr.huber_delta = 1.0;
        pr.fix_points = is_fix_map_pts ? 1 : 0;
        pr.pose_fixed = nullptr;  // g2o_ba.cpp:210-211: no pose is fixed
        pr.max_iterations = 50;
    }
    // g2o_ba.cpp:298-316: poses back into the caller's matrices, points back as float
    void scatter(std::unordered_map<int, cv::Point3f*>& pts_3d, vector<cv::Mat*>& v_camera_g2o_poses, bool is_update_map_pts) const {
        for (size_t i = 0; i < v_camera_g2o_poses.size(); ++i)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) v_camera_g2o_poses[i]->at<double>(r, c) = poses[16 * i + 4 * r + c];
        if (is_update_map_pts)
            for (size_t s = 0; s < slot2id.size(); ++s) {
                cv::Point3f* p = pts_3d[slot2id[s]];
                p->x = (float)pts[3 * s];
                p->y = (float)pts[3 * s + 1];
                p->z = (float)pts[3 * s + 2];
            }
    }
};

}  // namespace optimization
}  // namespace my_slam
#endif
