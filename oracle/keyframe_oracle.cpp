// oracle/keyframe_oracle.cpp -- CPU ORACLE for the keyframe row: what
// VisualOdometry::addFrame does when it inserts a keyframe (src/vo/vo_addFrame.cpp:93-124):
//   * geometry::helperTriangulatePoints (src/geometry/motion_estimation.cpp:214-247) =
//     pixel2CamNormPlane (camera.cpp:10-15) + doTriangulation (epipolar_geometry.cpp:130-175, cv::triangulatePoints)
//     + basics::transCoord (opencv_funcs.cpp:121-125)
//   * VisualOdometry::retainGoodTriangulationResult_ (src/vo/vo.cpp:181-244)
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: cv::triangulatePoints lives in OpenCV calib3d; restated
// here: per point the 4x4 DLT system A (rows x*P[2]-P[0], y*P[2]-P[1] for both views) and the right singular
// vector of its smallest singular value, with the canonical Jacobi SVD of linalg_oracle.h.
#pragma GCC optimize("no-tree-slp-vectorize")
#include <algorithm>
#include <vector>

#include "linalg_oracle.h"
#include "oracle.h"

namespace {
using namespace orc_linalg;

// cv::triangulatePoints for one correspondence on the normalised planes; P1 = [I | 0] (float in the reference, the
// values are exact), P2 = [R | t] double.  Output as OpenCV stores it for Point2f inputs: 4 floats.
void triangulate_one(const float np1[2], const float np2[2], const double R[9], const double t[3], float X[4]) {
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double P2[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
    double A[16];
    const double* P[2] = {P1, P2};
    const float* pt[2] = {np1, np2};
    for (int j = 0; j < 2; j++) {
        const double x = pt[j][0], y = pt[j][1];
        for (int k = 0; k < 4; k++) {
            A[(j * 2 + 0) * 4 + k] = x * P[j][8 + k] - P[j][k];
            A[(j * 2 + 1) * 4 + k] = y * P[j][8 + k] - P[j][4 + k];
        }
    }
    double At[16], Vt[16], W[4];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i * 4 + k] = A[k * 4 + i];
    jacobi_svd(At, 4, 4, Vt, W);
    for (int k = 0; k < 4; k++) X[k] = (float)Vt[12 + k];
}

}  // namespace

extern "C" {

// helperTriangulatePoints: kp1 / kp2 = matched pixel coordinates (n x 2 float, KeyPoint::pt of the previous /
// current keyframe), R, t = T_curr_to_prev.  Outputs n x 3 float: points in the previous camera frame
// (doTriangulation) and in the "current" one (transCoord), either may be NULL.
int orc_triangulate_points(const float* kp1, const float* kp2, int n, const double* K4, const double* R, const double* t,
                           float* pts_prev, float* pts_curr) {
    for (int i = 0; i < n; i++) {
        const float np1[2] = {(float)((kp1[2 * i] - K4[2]) / K4[0]), (float)((kp1[2 * i + 1] - K4[3]) / K4[1])};
        const float np2[2] = {(float)((kp2[2 * i] - K4[2]) / K4[0]), (float)((kp2[2 * i + 1] - K4[3]) / K4[1])};
        float X[4];
        triangulate_one(np1, np2, R, t, X);
        const float w = X[3];
        const float p[3] = {X[0] / w, X[1] / w, X[2] / w};  // x /= x.at<float>(3, 0)
        if (pts_prev) {
            pts_prev[3 * i] = p[0];
            pts_prev[3 * i + 1] = p[1];
            pts_prev[3 * i + 2] = p[2];
        }
        if (pts_curr)
            for (int r = 0; r < 3; r++) {
                double s = R[3 * r] * (double)p[0] + R[3 * r + 1] * (double)p[1];
                s = s + R[3 * r + 2] * (double)p[2];
                pts_curr[3 * i + r] = (float)(s + t[r]);
            }
    }
    return 0;
}

// retainGoodTriangulationResult_: angle (degrees) between the rays from the point to the two camera centres; keeps
// i unless angle < min_angle or angle / median > max_ratio.  angles (n doubles, all points) may be NULL.  Returns
// the number kept; keep receives their indices.
int orc_retain_good_triangulation(const float* pts_curr, int n, const double* T_w_c_curr, const double* T_w_c_ref,
                                  double min_angle, double max_ratio, int32_t* keep, double* angles) {
    if (n == 0) return 0;
    std::vector<double> a(n);
    for (int i = 0; i < n; i++) {
        const double p[4] = {pts_curr[3 * i], pts_curr[3 * i + 1], pts_curr[3 * i + 2], 1};
        float pw[3];
        for (int r = 0; r < 3; r++) {  // basics::preTranslatePoint3f
            double res = 0;
            for (int j = 0; j < 4; j++) res += T_w_c_curr[4 * r + j] * p[j];
            pw[r] = (float)res;
        }
        double v1[3], v2[3], dot = 0, s1 = 0, s2 = 0;
        for (int r = 0; r < 3; r++) {
            v1[r] = T_w_c_curr[4 * r + 3] - (double)pw[r];
            v2[r] = T_w_c_ref[4 * r + 3] - (double)pw[r];
        }
        for (int r = 0; r < 3; r++) dot += v1[r] * v2[r];
        for (int r = 0; r < 3; r++) s1 = s1 + v1[r] * v1[r];
        for (int r = 0; r < 3; r++) s2 = s2 + v2[r] * v2[r];
        const double len = sqrt(s1) * sqrt(s2);
        a[i] = acos(dot / len) / 3.1415926 * 180.0;
    }
    std::vector<double> sorted(a);
    std::sort(sorted.begin(), sorted.end());
    const double median = sorted[n / 2];
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        if (angles) angles[i] = a[i];
        if (a[i] < min_angle || a[i] / median > max_ratio) continue;
        keep[cnt++] = i;
    }
    return cnt;
}
}
