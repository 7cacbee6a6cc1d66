// host/tests/test_vo_loop.cpp -- runs the mirrored tracking branch (my_slam/vo/tracking_loop.h = reference
// src/vo/vo_addFrame.cpp:70-124) over a feature-level sequence: PnP tracking against the device-resident map, the
// sliding-window BA, keyframe insertion with epipolar filter + triangulation + culling, map growth and pruning.
//   test_vo_loop <scene.bin> <out.bin>
// scene.bin: int32 F, M0, cols, rows; double K[4]; per frame: double T_w_c_gt[16]; int32 N; float xy[N*2];
//            uint8 desc[N*32]; then M0 x {float pos[3]; int32 kp_in_frame1; int32 kp_in_frame0}.
// Frames 0 and 1 are the two keyframes an initialisation would have produced (ground-truth poses, frame 1 = ref).
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "my_slam/vo/tracking_loop.h"

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
    o.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    int hdr[4];
    rd(in, hdr, 4);
    const int F = hdr[0], M0 = hdr[1], cols = hdr[2], rows = hdr[3];
    double K4[4];
    rd(in, K4, 4);
    try {
        cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
        K.at<double>(0, 0) = K4[0];
        K.at<double>(1, 1) = K4[1];
        K.at<double>(0, 2) = K4[2];
        K.at<double>(1, 2) = K4[3];
        vector<vo::Frame::Ptr> frames;
        vector<cv::Mat> gt;
        for (int f = 0; f < F; ++f) {
            double T[16];
            int N;
            rd(in, T, 16);
            rd(in, &N, 1);
            vo::Frame::Ptr fr = vo::Frame::createFrame(cv::Mat(rows, cols, CV_8UC3));
            cv::Mat Tg(4, 4, CV_64FC1);
            for (int i = 0; i < 16; ++i) Tg.at<double>(i / 4, i % 4) = T[i];
            gt.push_back(Tg);
            vector<float> xy(2 * (size_t)N);
            rd(in, xy.data(), xy.size());
            for (int i = 0; i < N; ++i) fr->keypoints_.push_back(cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31));
            fr->descriptors_.create(N, 32, CV_8UC1);
            rd(in, fr->descriptors_.data, (size_t)N * 32);
            frames.push_back(fr);
        }
        vo::TrackingState st;
        frames[0]->T_w_c_ = gt[0].clone();
        frames[1]->T_w_c_ = gt[1].clone();
        for (int m = 0; m < M0; ++m) {
            float pos[3];
            int k1, k0;
            rd(in, pos, 3);
            rd(in, &k1, 1);
            rd(in, &k0, 1);
            cv::Mat d(1, 32, CV_8UC1), norm(3, 1, CV_64FC1);
            memcpy(d.data, frames[1]->descriptors_.ptr<unsigned char>(k1), 32);
            double len = 0;
            for (int r = 0; r < 3; ++r) {
                norm.at<double>(r, 0) = pos[r] - gt[1].at<double>(r, 3);
                len += norm.at<double>(r, 0) * norm.at<double>(r, 0);
            }
            for (int r = 0; r < 3; ++r) norm.at<double>(r, 0) /= std::sqrt(len);
            vo::MapPoint::Ptr mp(new vo::MapPoint(cv::Point3f(pos[0], pos[1], pos[2]), d, norm));
            st.map_->insertMapPoint(mp);
            frames[1]->inliers_to_mappt_connections_[k1] = vo::PtConn{k0, mp->id_};
        }
        st.pushFrameToBuff(frames[0]);
        st.pushFrameToBuff(frames[1]);
        st.map_->insertKeyFrame(frames[0]);
        st.map_->insertKeyFrame(frames[1]);
        st.ref_ = frames[1];
        st.prev_ = frames[1];
        for (int f = 2; f < F; ++f) {
            bool kf = false;
            const bool good = vo::trackFrame(st, frames[f], K, &kf);
            int rec[4] = {good ? 1 : 0, kf ? 1 : 0, (int)st.map_->map_points_.size(),
                          (int)frames[f]->inliers_to_mappt_connections_.size()};
            dump(out, rec, 4);
            dump(out, frames[f]->T_w_c_.ptr<double>(0), 16);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
