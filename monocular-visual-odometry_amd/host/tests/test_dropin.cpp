// host/tests/test_dropin.cpp -- exercises the C++ mirror of the reference interface the way the reference's own
// call sites do (vo_addFrame.cpp:24-25,42-46; vo.cpp:283-289,458-462; g2o_ba.cpp:193-289) and dumps the results
// so that tests/test_gpu_host_adapter.py can compare them with the CPU oracle.
//   test_dropin <img0.raw> <img1.raw> <w> <h> <channels> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>

#include "my_slam/geometry/feature_match.h"
#include "my_slam/optimization/g2o_ba.h"
#include "my_slam/optimization/g2o_facade.h"
#include "my_slam/vo/frame.h"
#include "my_slam/vo/ba_window.h"
#include "my_slam/vo/mappoint.h"

using namespace my_slam;

static cv::Mat load_raw(const char* path, int w, int h, int ch) {
    cv::Mat m(h, w, ch == 1 ? CV_8UC1 : CV_8UC3);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(m.data), (std::streamsize)w * h * ch)) {
        fprintf(stderr, "cannot read %s\n", path);
        exit(2);
    }
    return m;
}
template <class T>
static void dump(std::ofstream& o, const T* p, size_t n) {
    unsigned long long cnt = n;
    o.write(reinterpret_cast<const char*>(&cnt), 8);
    o.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]), ch = atoi(argv[5]);
    basics::Config::set("max_number_of_keypoints", "1000");
    std::ofstream out(argv[6], std::ios::binary);
    try {
        // ---- run_vo.cpp:122-123 / vo_addFrame.cpp:24-25
        vo::Frame::Ptr f0 = vo::Frame::createFrame(load_raw(argv[1], w, h, ch));
        vo::Frame::Ptr f1 = vo::Frame::createFrame(load_raw(argv[2], w, h, ch));
        for (auto& f : {f0, f1}) {
            f->calcKeyPoints();
            f->calcDescriptors();
            dump(out, f->keypoints_.data(), f->keypoints_.size());
            dump(out, f->descriptors_.data, (size_t)f->descriptors_.rows * 32);
            vector<unsigned char> rgb;
            for (auto& c : f->kpts_colors_) rgb.insert(rgb.end(), c.begin(), c.end());
            dump(out, rgb.data(), rgb.size());
        }
        // free functions without pyramid reuse must agree with the Frame methods
        vector<cv::KeyPoint> k2;
        cv::Mat d2;
        geometry::calcKeyPoints(f0->rgb_img_, k2);
        geometry::calcDescriptors(f0->rgb_img_, k2, d2);
        if (k2.size() != f0->keypoints_.size() || memcmp(d2.data, f0->descriptors_.data, k2.size() * 32)) {
            fprintf(stderr, "free functions disagree with Frame methods\n");
            return 3;
        }
        // ---- a second host thread has its own ctx (hot_path_ctx() is thread_local): the Config parameters must be
        // latched into THAT ctx as well -- with a process-wide latch it would extract with the built-in defaults
        // (1500 keypoints instead of the configured 1000) and silently differ from the reference
        {
            vector<cv::KeyPoint> kt;
            cv::Mat dt;
            std::string err;
            std::thread th([&] {
                try {
                    geometry::calcKeyPoints(f0->rgb_img_, kt);
                    geometry::calcDescriptors(f0->rgb_img_, kt, dt);
                } catch (const std::exception& e) {
                    err = e.what();
                }
            });
            th.join();
            if (!err.empty() || kt.size() != f0->keypoints_.size() || kt.size() > 1001 ||
                memcmp(dt.data, f0->descriptors_.data, kt.size() * 32)) {
                fprintf(stderr, "second host thread: %zu keypoints vs %zu (%s)\n", kt.size(), f0->keypoints_.size(), err.c_str());
                return 7;
            }
        }
        // ---- vo_addFrame.cpp:42-46: matchFeatures with each method
        for (int method = 1; method <= 3; ++method) {
            vector<cv::DMatch> m;
            geometry::matchFeatures(f0->descriptors_, f1->descriptors_, m, method, false, f0->keypoints_, f1->keypoints_, 50.f);
            dump(out, m.data(), m.size());
        }
        bool threw = false;
        try {
            vector<cv::DMatch> m;
            geometry::matchFeatures(f0->descriptors_, f1->descriptors_, m, 9);
        } catch (const std::runtime_error&) {
            threw = true;
        }
        if (!threw) return 4;
        // descriptors handed over as a ROI of a wider matrix (step 40 != 32: not continuous) must match like their packed copy;
        // a width other than cv::ORB's 32 bytes is refused, not silently misread
        {
            const int n0 = f0->descriptors_.rows, n1 = f1->descriptors_.rows;
            vector<unsigned char> wide0((size_t)n0 * 40 + 40, 0xAB), wide1((size_t)n1 * 40 + 40, 0xCD);
            for (int r = 0; r < n0; ++r) memcpy(&wide0[(size_t)r * 40], f0->descriptors_.ptr<unsigned char>(r), 32);
            for (int r = 0; r < n1; ++r) memcpy(&wide1[(size_t)r * 40], f1->descriptors_.ptr<unsigned char>(r), 32);
            cv::Mat roi0(n0, 32, CV_8UC1, wide0.data(), 40), roi1(n1, 32, CV_8UC1, wide1.data(), 40);
            for (int method = 1; method <= 3; ++method) {
                vector<cv::DMatch> a, b;
                geometry::matchFeatures(f0->descriptors_, f1->descriptors_, a, method, false, f0->keypoints_, f1->keypoints_, 50.f);
                geometry::matchFeatures(roi0, roi1, b, method, false, f0->keypoints_, f1->keypoints_, 50.f);
                if (a.size() != b.size() || a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(cv::DMatch))) {
                    fprintf(stderr, "method %d: ROI descriptors give %zu matches, packed ones %zu\n", method, b.size(), a.size());
                    return 8;
                }
            }
            cv::Mat wrong(n0, 40, CV_8UC1, wide0.data(), 40);
            threw = false;
            try {
                vector<cv::DMatch> m;
                geometry::matchFeatures(wrong, roi1, m, 2);
            } catch (const std::runtime_error&) {
                threw = true;
            }
            if (!threw) return 9;
        }

        // ---- vo.cpp:384-462: a 3-frame window, pointers into Frame / MapPoint storage
        const double fx = 517.3, cx = 325.1, cy = 249.7;
        cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
        K.at<double>(0, 0) = fx, K.at<double>(1, 1) = 516.5, K.at<double>(0, 2) = cx, K.at<double>(1, 2) = cy;
        cv::Mat info = cv::Mat::eye(2, 2, CV_64FC1);
        std::vector<vo::MapPoint::Ptr> map_pts;
        std::vector<vo::Frame::Ptr> frames;
        unsigned s = 12345;
        auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24); };
        for (int i = 0; i < 150; ++i) {
            double z = 0.8 + 2 * rnd(), u = 40 + 560 * rnd(), v = 40 + 400 * rnd();
            map_pts.push_back(vo::MapPoint::Ptr(new vo::MapPoint(
                cv::Point3f((float)((u - cx) / fx * z), (float)((v - cy) / fx * z), (float)z), cv::Mat(), cv::Mat())));
        }
        for (int f = 0; f < 3; ++f) {
            vo::Frame::Ptr fr = vo::Frame::createFrame(cv::Mat());
            fr->T_w_c_.at<double>(0, 3) = 0.05 * f + 0.004;  // cam->world translation (perturbed truth)
            fr->T_w_c_.at<double>(1, 3) = -0.003;
            for (size_t i = 0; i < map_pts.size(); ++i) {
                const cv::Point3f& X = map_pts[i]->pos_;
                double xc = X.x - 0.05 * f, yc = X.y, zc = X.z;  // true pose: pure x translation
                fr->keypoints_.push_back(cv::KeyPoint((float)(fx * xc / zc + cx + rnd() - 0.5), (float)(fx * yc / zc + cy + rnd() - 0.5), 31));
                fr->inliers_to_mappt_connections_[(int)i] = {-1, map_pts[i]->id_};
            }
            frames.push_back(fr);
        }
        auto run = [&](bool fix, bool update, vector<cv::Mat>& poses_out, vector<cv::Point3f>& pts_out) {
            vector<vector<cv::Point2f*>> v_pts_2d;
            vector<vector<int>> v_idx;
            std::unordered_map<int, cv::Point3f*> um;
            vector<cv::Mat*> v_poses;
            vector<cv::Mat> poses;
            vector<cv::Point3f> pts;
            for (auto& mp : map_pts) pts.push_back(mp->pos_);
            for (auto& fr : frames) poses.push_back(fr->T_w_c_.clone());
            for (size_t f = 0; f < frames.size(); ++f) {
                v_pts_2d.push_back({});
                v_idx.push_back({});
                v_poses.push_back(&poses[f]);
                for (auto& kv : frames[f]->inliers_to_mappt_connections_) {
                    v_pts_2d.back().push_back(&frames[f]->keypoints_[kv.first].pt);
                    v_idx.back().push_back(kv.second.pt_map_idx);
                    um[kv.second.pt_map_idx] = &pts[kv.second.pt_map_idx - map_pts[0]->id_];
                }
            }
            optimization::bundleAdjustment(v_pts_2d, v_idx, K, um, v_poses, info, fix, update);
            poses_out = poses;
            pts_out = pts;
        };
        vector<cv::Mat> P1, P2;
        vector<cv::Point3f> X1, X2;
        run(true, false, P1, X1);   // shipped default: is_ba_fix_map_points = true
        for (auto& P : P1) dump(out, P.ptr<double>(0), 16);
        // the same window through the g2o-shaped facade (g2o_ba.cpp:193-289 call sequence)
        {
            typedef g2o::BlockSolver<g2o::BlockSolverTraits<6, 3>> Block;
            Block::LinearSolverType* linearSolver = new g2o::LinearSolverDense<Block::PoseMatrixType>();
            Block* solver_ptr = new Block(linearSolver);
            g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
            g2o::SparseOptimizer optimizer;
            optimizer.setAlgorithm(solver);
            int vertex_id = 0;
            vector<g2o::VertexSE3Expmap*> g2o_poses;
            for (auto& fr : frames) {
                g2o::VertexSE3Expmap* pose = new g2o::VertexSE3Expmap();
                pose->setId(vertex_id++);
                Eigen::Matrix3d R;  // T_c_w of a pure translation: R = I, t = -t_wc
                Eigen::Vector3d t(-fr->T_w_c_.at<double>(0, 3), -fr->T_w_c_.at<double>(1, 3), -fr->T_w_c_.at<double>(2, 3));
                pose->setEstimate(g2o::SE3Quat(R, t));
                optimizer.addVertex(pose);
                g2o_poses.push_back(pose);
            }
            g2o::CameraParameters* camera = new g2o::CameraParameters(fx, Eigen::Vector2d(cx, cy), 0);
            camera->setId(0);
            optimizer.addParameter(camera);
            std::unordered_map<int, int> id2v;
            for (auto& mp : map_pts) {
                g2o::VertexSBAPointXYZ* point = new g2o::VertexSBAPointXYZ();
                point->setId(vertex_id);
                point->setFixed(true);
                id2v[mp->id_] = vertex_id++;
                point->setEstimate(Eigen::Vector3d(mp->pos_.x, mp->pos_.y, mp->pos_.z));
                point->setMarginalized(true);
                optimizer.addVertex(point);
            }
            int edge_id = 0;
            for (size_t f = 0; f < frames.size(); ++f)
                for (auto& kv : frames[f]->inliers_to_mappt_connections_) {
                    g2o::EdgeProjectXYZ2UV* edge = new g2o::EdgeProjectXYZ2UV();
                    edge->setId(edge_id++);
                    edge->setVertex(0, dynamic_cast<g2o::VertexSBAPointXYZ*>(optimizer.vertex(id2v[kv.second.pt_map_idx])));
                    edge->setVertex(1, dynamic_cast<g2o::VertexSE3Expmap*>(optimizer.vertex((int)f)));
                    const cv::Point2f& p = frames[f]->keypoints_[kv.first].pt;
                    edge->setMeasurement(Eigen::Vector2d(p.x, p.y));
                    edge->setParameterId(0, 0);
                    edge->setInformation(Eigen::Matrix2d::Identity());
                    edge->setRobustKernel(new g2o::RobustKernelHuber());
                    optimizer.addEdge(edge);
                }
            optimizer.initializeOptimization();
            optimizer.optimize(50);
            for (size_t f = 0; f < frames.size(); ++f) {
                const Eigen::Vector3d& t = g2o_poses[f]->estimate().translation();
                const Eigen::Matrix3d& R = g2o_poses[f]->estimate().rotation();
                // camera centre = -R^T t must equal the translation column of the adapter's T_w_c
                for (int r = 0; r < 3; ++r) {
                    double c = -(R(0, r) * t[0] + R(1, r) * t[1] + R(2, r) * t[2]);
                    if (std::fabs(c - P1[f].at<double>(r, 3)) > 1e-9) {
                        fprintf(stderr, "g2o facade disagrees with bundleAdjustment: %g vs %g\n", c, P1[f].at<double>(r, 3));
                        return 5;
                    }
                }
            }
        }
        // ---- vo.cpp:384-478: the marshalling itself (deque of frames + Map), skip rules included
        {
            vo::Map::Ptr map(new vo::Map());
            for (auto& mp : map_pts) map->insertMapPoint(mp);
            std::deque<vo::Frame::Ptr> buff;
            vo::Frame::Ptr old_frame = vo::Frame::createFrame(cv::Mat());  // falls out of the 5-frame window
            buff.push_back(old_frame);
            vo::Frame::Ptr sparse = vo::Frame::createFrame(cv::Mat());     // < 3 connections -> skipped
            sparse->keypoints_.resize(2);
            sparse->inliers_to_mappt_connections_[0] = {-1, map_pts[0]->id_};
            sparse->inliers_to_mappt_connections_[1] = {-1, map_pts[1]->id_};
            for (int rep = 0; rep < 2; ++rep) buff.push_back(rep == 0 ? frames[2] : sparse);
            buff.push_back(frames[1]);
            buff.push_back(frames[0]);
            // a deleted map point: connection is skipped
            const int deleted_id = map_pts[5]->id_;
            map->map_points_.erase(deleted_id);
            vo::BaWindow w = vo::buildBundleAdjustmentWindow(buff, map, 5);
            // buffered 5 frames -> min(5, 4) = 4 newest: frames[0], frames[1], sparse (skipped), frames[2]
            bool ok = w.v_camera_poses.size() == 3 && w.frame_ids[0] == frames[0]->id_ && w.frame_ids[1] == frames[1]->id_ &&
                      w.frame_ids[2] == frames[2]->id_ && w.v_pts_2d[0].size() == map_pts.size() - 1 &&
                      w.um_pts_3d_in_prev_frames.count(deleted_id) == 0 && w.v_pts_3d_only_in_curr.size() == map_pts.size() - 1 &&
                      w.v_camera_poses[0] == &frames[0]->T_w_c_;
            if (!ok) {
                fprintf(stderr, "buildBundleAdjustmentWindow: wrong window\n");
                return 6;
            }
            vector<cv::Mat> before;
            for (auto& fr : frames) before.push_back(fr->T_w_c_.clone());
            vo::callBundleAdjustment(buff, map, K);  // shipped default: pose-only, poses overwritten in place
            for (size_t f = 0; f < frames.size(); ++f) {
                // same window as P1 except for the one deleted landmark: poses agree to the pixel-noise level
                for (int r = 0; r < 3; ++r)
                    if (std::fabs(frames[f]->T_w_c_.at<double>(r, 3) - P1[f].at<double>(r, 3)) > 1e-3) {
                        fprintf(stderr, "callBundleAdjustment: pose %zu differs from the direct call\n", f);
                        return 7;
                    }
                frames[f]->T_w_c_ = before[f];
            }
            map->insertMapPoint(map_pts[5]);
        }
        run(false, true, P2, X2);   // full BA: points move and are written back as f32
        for (auto& P : P2) dump(out, P.ptr<double>(0), 16);
        dump(out, X2.data(), X2.size());
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    printf("test_dropin OK\n");
    return 0;
}
