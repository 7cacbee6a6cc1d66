// tests/golden/reference_ba_driver.cpp -- pins the bundle adjustment to the REAL reference without a g2o Python binding: a
// small program that feeds one window (plain text, written by make_reference_golden.py --export-ba-text) to
// my_slam::optimization::bundleAdjustment through the reference's OWN header and writes what comes back.  Link it with
//   * the reference's src/optimization/g2o_ba.cpp (+ src/basics/eigen_funcs.cpp, OpenCV, g2o, Sophus, Eigen) -> the reference
//     result:  g++ -std=c++11 reference_ba_driver.cpp $REF/src/optimization/g2o_ba.cpp $REF/src/basics/eigen_funcs.cpp \
//              -I$REF/include `pkg-config --cflags --libs opencv` -lg2o_core -lg2o_stuff -lg2o_types_sba -lg2o_solver_dense ...
//     ./reference_ba_driver ba_3x40.txt reference_ba_3x40.txt;  python make_reference_golden.py --import-ba-text reference_ba_3x40.txt
//   * this repo's host/src/g2o_ba_mvo.cpp + libmvo_hip.so -> the MI355X result through the same call.
// The window is marshalled the way VisualOdometry::callBundleAdjustment_ does it (src/vo/vo.cpp:408-462): pointers into
// cv::Point2f / cv::Point3f / cv::Mat storage, landmark ids as keys of the unordered_map.
// Text format: "F L E f cx cy i00 i01 i10 i11 fix update" / F lines of 16 doubles (T_w_c row-major) / L lines "x y z" (float) /
// E lines "frame landmark u v" (float pixels).  Output: F lines of 16 doubles, L lines of 3 floats.
#include <cstdio>
#include <cstdlib>

#include "my_slam/optimization/g2o_ba.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* in = std::fopen(argv[1], "r");
    if (!in) return 3;
    int F, L, E, fix, update;
    double f, cx, cy, info[4];
    if (std::fscanf(in, "%d %d %d %lf %lf %lf %lf %lf %lf %lf %d %d", &F, &L, &E, &f, &cx, &cy, &info[0], &info[1], &info[2], &info[3], &fix, &update) != 12) return 4;
    std::vector<cv::Mat> poses;
    for (int i = 0; i < F; ++i) {
        cv::Mat T(4, 4, CV_64FC1);
        for (int k = 0; k < 16; ++k)
            if (std::fscanf(in, "%lf", &T.at<double>(k / 4, k % 4)) != 1) return 4;
        poses.push_back(T);
    }
    std::vector<cv::Point3f> points(L);
    for (int l = 0; l < L; ++l)
        if (std::fscanf(in, "%f %f %f", &points[l].x, &points[l].y, &points[l].z) != 3) return 4;
    std::vector<std::vector<cv::Point2f>> pixels(F);
    std::vector<std::vector<int>> ids(F);
    for (int e = 0; e < E; ++e) {
        int p, l;
        float u, v;
        if (std::fscanf(in, "%d %d %f %f", &p, &l, &u, &v) != 4) return 4;
        pixels[p].push_back(cv::Point2f(u, v));
        ids[p].push_back(l);
    }
    std::fclose(in);
    std::vector<std::vector<cv::Point2f*>> v_pts_2d(F);
    std::unordered_map<int, cv::Point3f*> pts_3d;
    std::vector<cv::Mat*> v_camera_poses;
    for (int p = 0; p < F; ++p) {
        for (size_t j = 0; j < pixels[p].size(); ++j) {
            v_pts_2d[p].push_back(&pixels[p][j]);
            pts_3d[ids[p][j]] = &points[ids[p][j]];
        }
        v_camera_poses.push_back(&poses[p]);
    }
    cv::Mat K(3, 3, CV_64FC1), information(2, 2, CV_64FC1);
    const double k9[9] = {f, 0, cx, 0, f, cy, 0, 0, 1};
    for (int k = 0; k < 9; ++k) K.at<double>(k / 3, k % 3) = k9[k];
    for (int k = 0; k < 4; ++k) information.at<double>(k / 2, k % 2) = info[k];
    my_slam::optimization::bundleAdjustment(v_pts_2d, ids, K, pts_3d, v_camera_poses, information, fix != 0, update != 0);
    FILE* out = std::fopen(argv[2], "w");
    if (!out) return 5;
    for (int i = 0; i < F; ++i) {
        for (int k = 0; k < 16; ++k) std::fprintf(out, "%.17g ", poses[i].at<double>(k / 4, k % 4));
        std::fprintf(out, "\n");
    }
    for (int l = 0; l < L; ++l) std::fprintf(out, "%.9g %.9g %.9g\n", points[l].x, points[l].y, points[l].z);
    std::fclose(out);
    return 0;
}
