// host/tests/test_keyframe.cpp -- drives the keyframe mirror (my_slam/vo/keyframe.h) the way VisualOdometry::addFrame
// does (src/vo/vo_addFrame.cpp:96-118) and dumps the results for tests/test_gpu_host_adapter.py.
//   test_keyframe <scene.bin> <out.bin>
// scene.bin: int32 N1, N2; double K[4]; double T_w_ref[16], T_w_cur[16]; float kp_ref[N1*2]; uint8 desc_ref[N1*32];
//            float kp_cur[N2*2]; uint8 desc_cur[N2*32]
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "my_slam/vo/keyframe.h"

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
    int hdr[2];
    rd(in, hdr, 2);
    const int N1 = hdr[0], N2 = hdr[1];
    double K4[4], Tr[16], Tc[16];
    rd(in, K4, 4);
    rd(in, Tr, 16);
    rd(in, Tc, 16);
    try {
        cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
        K.at<double>(0, 0) = K4[0];
        K.at<double>(1, 1) = K4[1];
        K.at<double>(0, 2) = K4[2];
        K.at<double>(1, 2) = K4[3];
        vo::Frame::Ptr ref = vo::Frame::createFrame(cv::Mat()), curr = vo::Frame::createFrame(cv::Mat());
        for (int i = 0; i < 16; ++i) {
            ref->T_w_c_.at<double>(i / 4, i % 4) = Tr[i];
            curr->T_w_c_.at<double>(i / 4, i % 4) = Tc[i];
        }
        for (auto& fn : {std::make_pair(ref, N1), std::make_pair(curr, N2)}) {
            vector<float> xy(2 * (size_t)fn.second);
            rd(in, xy.data(), xy.size());
            for (int i = 0; i < fn.second; ++i) fn.first->keypoints_.push_back(cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31));
            fn.first->descriptors_.create(fn.second, 32, CV_8UC1);
            rd(in, fn.first->descriptors_.data, (size_t)fn.second * 32);
        }
        vo::triangulateWithReferenceKeyframe(curr, ref, K);
        dump(out, curr->matches_with_ref_.data(), curr->matches_with_ref_.size());
        dump(out, curr->inliers_matches_with_ref_.data(), curr->inliers_matches_with_ref_.size());
        dump(out, curr->inliers_matches_for_3d_.data(), curr->inliers_matches_for_3d_.size());
        dump(out, curr->inliers_pts3d_.empty() ? nullptr : &curr->inliers_pts3d_[0].x, curr->inliers_pts3d_.size() * 3);
        dump(out, curr->triangulation_angles_of_inliers_.data(), curr->triangulation_angles_of_inliers_.size());
        cv::Mat T = vo::getMotionFromFrame1to2(curr, ref);
        dump(out, T.ptr<double>(0), 16);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
