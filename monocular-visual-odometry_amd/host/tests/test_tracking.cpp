// host/tests/test_tracking.cpp -- drives the tracking mirror (my_slam/vo/pnp_tracking.h) the way
// VisualOdometry::poseEstimationPnP_ does (src/vo/vo.cpp:270-383) on a scene read from a file, and dumps what it
// produced so that tests/test_gpu_host_adapter.py can compare it with the CPU oracle.
//   test_tracking <scene.bin> <out.bin>
// scene.bin: int32 M, N, cols, rows; double K[4] (fx fy cx cy); double T_w_c[16] (curr guess), T_prev[16];
//            float map_pos[M*3]; uint8 map_desc[M*32]; float kp_xy[N*2]; uint8 desc[N*32]
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "my_slam/vo/pnp_tracking.h"

using namespace my_slam;

template <class T>
static void rd(std::ifstream& f, T* p, size_t n) {
    if (!f.read(reinterpret_cast<char*>(p), (std::streamsize)(n * sizeof(T)))) {
        fprintf(stderr, "short scene file\n");
        exit(2);
    }
}
template <class T>
static void dump(std::ofstream& o, const T* p, size_t n) {
    unsigned long long cnt = n;
    o.write(reinterpret_cast<const char*>(&cnt), 8);
    o.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    int hdr[4];
    rd(in, hdr, 4);
    const int M = hdr[0], N = hdr[1], cols = hdr[2], rows = hdr[3];
    double K4[4], Tc[16], Tp[16];
    rd(in, K4, 4);
    rd(in, Tc, 16);
    rd(in, Tp, 16);
    vector<float> pos(3 * (size_t)M), xy(2 * (size_t)N);
    vector<unsigned char> mdesc(32 * (size_t)M), desc(32 * (size_t)N);
    rd(in, pos.data(), pos.size());
    rd(in, mdesc.data(), mdesc.size());
    rd(in, xy.data(), xy.size());
    rd(in, desc.data(), desc.size());
    try {
        cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
        K.at<double>(0, 0) = K4[0];
        K.at<double>(1, 1) = K4[1];
        K.at<double>(0, 2) = K4[2];
        K.at<double>(1, 2) = K4[3];
        vo::Map::Ptr map(new vo::Map());
        for (int i = 0; i < M; ++i) {
            cv::Mat d(1, 32, CV_8UC1);
            memcpy(d.data, &mdesc[32 * (size_t)i], 32);
            map->insertMapPoint(vo::MapPoint::Ptr(new vo::MapPoint(cv::Point3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), d, cv::Mat())));
        }
        vo::Frame::Ptr curr = vo::Frame::createFrame(cv::Mat(rows, cols, CV_8UC3)), prev = vo::Frame::createFrame(cv::Mat());
        for (int i = 0; i < 16; ++i) {
            curr->T_w_c_.at<double>(i / 4, i % 4) = Tc[i];
            prev->T_w_c_.at<double>(i / 4, i % 4) = Tp[i];
        }
        for (int i = 0; i < N; ++i) curr->keypoints_.push_back(cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31));
        curr->descriptors_.create(N, 32, CV_8UC1);
        memcpy(curr->descriptors_.data, desc.data(), desc.size());

        vo::MapOnDevice dev_map;
        // -- getMappointsInCurrentView_ alone first (vo.cpp:16-49): ids in the map's own iteration order
        vector<vo::MapPoint::Ptr> cand;
        vector<cv::Point2f> cand_px;
        cv::Mat cand_desc;
        vo::getMappointsInCurrentView(dev_map, map, curr, K, cand, cand_px, cand_desc);
        vector<int> ids, order_ids;
        for (auto& p : cand) ids.push_back(p->id_);
        for (auto& p : dev_map.order()) order_ids.push_back(p->id_);
        dump(out, order_ids.data(), order_ids.size());
        dump(out, ids.data(), ids.size());
        dump(out, cand_px.empty() ? nullptr : &cand_px[0].x, cand_px.size() * 2);
        dump(out, cand_desc.data, (size_t)cand_desc.rows * 32);
        // -- poseEstimationPnP_ (vo.cpp:270-383)
        const bool good = vo::poseEstimationPnP(dev_map, map, curr, prev, K);
        int flag = good;
        dump(out, &flag, 1);
        dump(out, curr->matches_with_map_.data(), curr->matches_with_map_.size());
        dump(out, curr->T_w_c_.ptr<double>(0), 16);
        vector<int> conn;  // keypoint index -> map point id
        for (auto& kv : curr->inliers_to_mappt_connections_) {
            conn.push_back(kv.first);
            conn.push_back(kv.second.pt_map_idx);
        }
        dump(out, conn.data(), conn.size());
        vector<int> times;  // visible_times_, matched_times_ per map point id order 0..M-1
        times.resize(2 * (size_t)M);
        for (auto& kv : map->map_points_) {
            times[2 * kv.first] = kv.second->visible_times_;
            times[2 * kv.first + 1] = kv.second->matched_times_;
        }
        dump(out, times.data(), times.size());
        // the literal cv:: call of vo.cpp:326-334 compiles against the mirror too
        vector<cv::Point3f> p3 = {cand[0]->pos_, cand[1]->pos_, cand[2]->pos_, cand[3]->pos_, cand[4]->pos_};
        vector<cv::Point2f> p2 = {cand_px[0], cand_px[1], cand_px[2], cand_px[3], cand_px[4]};
        cv::Mat pnp_inliers_mask, R_vec, t, R;
        cv::solvePnPRansac(p3, p2, K, cv::Mat(), R_vec, t, false, 100, 2.0, 0.999, pnp_inliers_mask);
        cv::Rodrigues(R_vec, R);
        if (pnp_inliers_mask.rows != 5 || pnp_inliers_mask.at<int>(4, 0) != 4) return 3;
        dump(out, R.ptr<double>(0), 9);
        dump(out, t.ptr<double>(0), 3);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
