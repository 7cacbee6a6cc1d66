// host/tests/test_callsites.cpp -- link + behaviour check of the replaced headers: every free function that the
// reference declares in include/my_slam/geometry/feature_match.h:12-54 and include/my_slam/optimization/g2o_ba.h:16-30
// is called here with the argument types of the reference's own call sites (src/vo/vo.cpp:139,277,283,458-470,
// src/vo/vo_addFrame.cpp:42,99, test/test_epipolor_geometry.cpp:91-98), so that a function missing from the drop-in
// headers fails THIS build instead of a maintainer's (tests/test_boundary.py keeps the list in step with the reference
// headers).  Run on the MI355X it also dumps results for tests/test_gpu_host_adapter.py:
//   test_callsites <out.bin>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>

#include "my_slam/geometry/feature_match.h"
#include "my_slam/optimization/g2o_ba.h"

using namespace my_slam;

template <class T>
static void dump(std::ofstream& o, const T* p, size_t n) {
    unsigned long long cnt = n;
    o.write(reinterpret_cast<const char*>(&cnt), 8);
    o.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ofstream out(argv[1], std::ios::binary);
    std::mt19937 rng(5);
    try {
        // ---- two keypoint / descriptor sets (what Frame::calcKeyPoints / calcDescriptors deliver)
        const int N1 = 300, N2 = 340;
        vector<cv::KeyPoint> k1, k2;
        cv::Mat d1(N1, 32, CV_8UC1), d2(N2, 32, CV_8UC1);
        std::uniform_real_distribution<float> ux(0, 640), uy(0, 480);
        for (int i = 0; i < N1; ++i) k1.push_back(cv::KeyPoint(ux(rng), uy(rng), 31));
        for (int i = 0; i < N2; ++i) {
            if (i < N1 && i % 3) k2.push_back(cv::KeyPoint(k1[i].pt.x + 2.f, k1[i].pt.y - 1.5f, 31));
            else k2.push_back(cv::KeyPoint(ux(rng), uy(rng), 31));
        }
        for (int i = 0; i < N1 * 32; ++i) d1.data[i] = (unsigned char)(rng() & 255);
        for (int i = 0; i < N2; ++i)
            for (int b = 0; b < 32; ++b)
                d2.data[32 * i + b] = (i < N1 && i % 3) ? (unsigned char)(d1.data[32 * i + b] ^ ((rng() & 31) == 0 ? 4 : 0)) : (unsigned char)(rng() & 255);

        // feature_match.h:20-28 (callers vo_addFrame.cpp:42,99; vo.cpp:283): all three methods, default arguments too
        vector<cv::DMatch> m1, m2, m3;
        geometry::matchFeatures(d1, d2, m1);
        geometry::matchFeatures(d1, d2, m2, 2, false);
        geometry::matchFeatures(d1, d2, m3, 3, false, k1, k2, 10.0f);
        // feature_match.h:30-35
        vector<cv::DMatch> mr = geometry::matchByRadiusAndBruteForce(k1, k2, d1, d2, 10.0f);
        // feature_match.h:40-44
        vector<cv::DMatch> mdup = mr;
        mdup.insert(mdup.end(), mr.begin(), mr.begin() + (mr.size() > 5 ? 5 : mr.size()));
        geometry::removeDuplicatedMatches(mdup);
        vector<cv::KeyPoint> kgrid = k1;
        geometry::selectUniformKptsByGrid(kgrid, 480, 640);
        // feature_match.h:47-49 (caller vo.cpp:139)
        const double mean_dist = geometry::computeMeanDistBetweenKeypoints(k1, k2, m1);
        // feature_match.h:52-53 (caller vo.cpp:277: pts2Keypts of the projected map points)
        vector<int> inl = {0, 3, 7};
        vector<cv::DMatch> minl = geometry::inliers2DMatches(inl);
        vector<cv::Point2f> pts = {cv::Point2f(1.5f, 2.5f), cv::Point2f(30.f, 40.f)};
        vector<cv::KeyPoint> kp = geometry::pts2Keypts(pts);
        if (minl.size() != 3 || minl[1].queryIdx != 3 || minl[1].trainIdx != 3 || minl[1].distance != 0.f || kp.size() != 2 ||
            kp[1].pt.x != 30.f || kp[1].size != 10.f) {
            fprintf(stderr, "datatype conversions differ from feature_match.cpp:281-303\n");
            return 3;
        }
        dump(out, k1.data(), k1.size());
        dump(out, k2.data(), k2.size());
        dump(out, d1.data, (size_t)N1 * 32);
        dump(out, d2.data, (size_t)N2 * 32);
        dump(out, m1.data(), m1.size());
        dump(out, m2.data(), m2.size());
        dump(out, m3.data(), m3.size());
        dump(out, mr.data(), mr.size());
        dump(out, mdup.data(), mdup.size());
        dump(out, &mean_dist, 1);

        // ---- g2o_ba.h:16-21 optimizeSingleFrame (dead code in the reference, vo.cpp:456-470): one pose + its points
        const int NP = 60;
        const double f = 517.3, cx = 325.1, cy = 249.7;
        cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
        K.at<double>(0, 0) = f;
        K.at<double>(1, 1) = 516.5;
        K.at<double>(0, 2) = cx;
        K.at<double>(1, 2) = cy;
        vector<cv::Point3f> p3(NP);
        vector<cv::Point2f> p2(NP);
        std::normal_distribution<double> noise(0.0, 0.4);
        std::uniform_real_distribution<double> uz(1.0, 4.0), uu(40, 600), uv(40, 440);
        for (int i = 0; i < NP; ++i) {
            const double z = uz(rng), u = uu(rng), v = uv(rng);
            p3[i] = cv::Point3f((float)((u - cx) / f * z), (float)((v - cy) / f * z), (float)z);  // camera at the origin
            p2[i] = cv::Point2f((float)(u + noise(rng)), (float)(v + noise(rng)));
        }
        cv::Mat T = cv::Mat::eye(4, 4, CV_64FC1);  // start 2 cm / ~0.6 deg off
        T.at<double>(0, 3) = 0.02;
        T.at<double>(1, 3) = -0.01;
        T.at<double>(0, 1) = -0.01;
        T.at<double>(1, 0) = 0.01;
        vector<cv::Point2f*> pp2;
        vector<cv::Point3f*> pp3;
        for (int i = 0; i < NP; ++i) {
            pp2.push_back(&p2[i]);
            pp3.push_back(&p3[i]);
        }
        vector<cv::Point3f> p3_before = p3;
        cv::Mat T0 = T.clone();
        dump(out, reinterpret_cast<const float*>(p2.data()), (size_t)2 * NP);
        dump(out, reinterpret_cast<const float*>(p3_before.data()), (size_t)3 * NP);
        dump(out, T0.ptr<double>(), 16);
        optimization::optimizeSingleFrame(pp2, K, pp3, T, true, false);   // pose only
        dump(out, T.ptr<double>(), 16);
        for (int i = 0; i < NP; ++i)
            if (p3[i].x != p3_before[i].x || p3[i].z != p3_before[i].z) {
                fprintf(stderr, "optimizeSingleFrame changed fixed / not-updated points\n");
                return 3;
            }
        cv::Mat T2 = T0.clone();
        optimization::optimizeSingleFrame(pp2, K, pp3, T2, false, true);  // pose + points, written back as float
        dump(out, T2.ptr<double>(), 16);
        dump(out, reinterpret_cast<const float*>(p3.data()), (size_t)3 * NP);

        // ---- g2o_ba.h:23-30 bundleAdjustment with the pointer lists of vo.cpp:426-449
        vector<vector<cv::Point2f*>> v_pts_2d(1, pp2);
        vector<vector<int>> v_idx(1);
        std::unordered_map<int, cv::Point3f*> um;
        p3 = p3_before;
        for (int i = 0; i < NP; ++i) {
            v_idx[0].push_back(100 + i);
            um[100 + i] = &p3[i];
        }
        cv::Mat T3 = T0.clone();
        vector<cv::Mat*> poses = {&T3};
        cv::Mat info = cv::Mat::eye(2, 2, CV_64FC1);
        optimization::bundleAdjustment(v_pts_2d, v_idx, K, um, poses, info, true, false);
        dump(out, T3.ptr<double>(), 16);
    } catch (const std::exception& e) {
        fprintf(stderr, "test_callsites: %s\n", e.what());
        return 1;
    }
    return 0;
}
