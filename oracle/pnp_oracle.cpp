// oracle/pnp_oracle.cpp -- CPU ORACLE for the tracking rows:
//   * VisualOdometry::getMappointsInCurrentView_     (src/vo/vo.cpp:16-49, src/basics/opencv_funcs.cpp:67-78,
//                                                      src/geometry/camera.cpp:23-28)
//   * cv::solvePnPRansac as called at                 src/vo/vo.cpp:318-329 (useExtrinsicGuess=false, 100
//     iterations, 2.0 px, confidence 0.999, flags = SOLVEPNP_ITERATIVE, no distortion) and cv::Rodrigues at :334.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: solvePnPRansac lives in OpenCV calib3d, which is
// not vendored and not installed.  What is restated here is the published upstream structure
// (RANSACPointSetRegistrator with RNG(-1) and 5-point EPnP hypotheses, then cvFindExtrinsicCameraParams2 = DLT
// initialisation + CvLevMarq(20 it, FLT_EPSILON) on the inliers), with OUR canonical arithmetic wherever OpenCV
// calls its SVD: a one-sided (Hestenes) Jacobi, round-robin pair order for n = 6 and 12, cyclic order otherwise,
// right singular vectors taken from the accumulated rotations.  Documented deviations:
//   - a hypothesis is carried as (R, t), not as (rvec, tvec): no Rodrigues round trip before scoring;
//   - planar inlier sets (W[2]/W[1] < 1e-3) and sets with < 6 inliers start the refinement from the best RANSAC
//     model instead of cv::findHomography / an under-determined DLT.
// g++ 11 -O3 -fPIC SLP-vectorises the float narrowing in epnp_subset() away (the (float) round trip of the
// normalised image coordinates disappears, a 1e-8 relative change); -O1/-O2, the sanitizer build and clang agree
// with each other, so SLP vectorisation is switched off for this file.
#pragma GCC optimize("no-tree-slp-vectorize")
#include <float.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "linalg_oracle.h"
#include "oracle.h"

namespace {
using namespace orc_linalg;

double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// ------------------------------------------------------------------------------------------- EPnP
struct Epnp {
    double uc, vc, fu, fv;
    int n;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    void choose_control_points() {
        for (int j = 0; j < 3; j++) cws[0][j] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        double ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // PW0^T PW0
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double s = 0;
                for (int i = 0; i < n; i++) s += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
                ptp[a * 3 + b] = s;
            }
        double At[9], Vt[9], dc[3];
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) At[i * 3 + k] = ptp[k * 3 + i];
        jacobi_svd(At, 3, 3, Vt, dc);
        for (int i = 1; i < 4; i++) {
            double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * Vt[3 * (i - 1) + j];
        }
    }
    void compute_barycentric_coordinates() {
        double cc[9], ci[9], eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        svd_solve(cc, 3, 3, eye, 3, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = &pws[3 * i];
            double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) +
                           ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    void compute_L_6x10(const double* ut, double* l) {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                b++;
                if (b > 3) {
                    a++;
                    b = a + 1;
                }
            }
        }
        for (int i = 0; i < 6; i++) {
            double* row = l + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
    }
    void compute_rho(double* rho) {
        rho[0] = dist2(cws[0], cws[1]);
        rho[1] = dist2(cws[0], cws[2]);
        rho[2] = dist2(cws[0], cws[3]);
        rho[3] = dist2(cws[1], cws[2]);
        rho[4] = dist2(cws[1], cws[3]);
        rho[5] = dist2(cws[2], cws[3]);
    }
    static void find_betas(const double* l, const double* rho, int variant, double* betas) {
        static const int cols1[4] = {0, 1, 3, 6}, cols23[5] = {0, 1, 2, 3, 4};
        const int nc = variant == 1 ? 4 : variant == 2 ? 3 : 5;
        const int* cols = variant == 1 ? cols1 : cols23;
        double L[6 * 5], b[5];
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < nc; k++) L[i * nc + k] = l[10 * i + cols[k]];
        svd_solve(L, 6, nc, rho, 1, b);
        if (variant == 1) {
            if (b[0] < 0) {
                betas[0] = sqrt(-b[0]);
                betas[1] = -b[1] / betas[0];
                betas[2] = -b[2] / betas[0];
                betas[3] = -b[3] / betas[0];
            } else {
                betas[0] = sqrt(b[0]);
                betas[1] = b[1] / betas[0];
                betas[2] = b[2] / betas[0];
                betas[3] = b[3] / betas[0];
            }
            return;
        }
        if (b[0] < 0) {
            betas[0] = sqrt(-b[0]);
            betas[1] = (b[2] < 0) ? sqrt(-b[2]) : 0.0;
        } else {
            betas[0] = sqrt(b[0]);
            betas[1] = (b[2] > 0) ? sqrt(b[2]) : 0.0;
        }
        if (b[1] < 0) betas[0] = -betas[0];
        betas[2] = variant == 2 ? 0.0 : b[3] / betas[0];
        betas[3] = 0.0;
    }
    // Householder QR least squares of the 6 x 4 Gauss-Newton system (epnp::qr_solve).
    static void qr_solve_6x4(double* A, double* b, double* X) {
        const int nr = 6, nc = 4;
        double A1[4], A2[4];
        for (int k = 0; k < nc; k++) {
            double eta = fabs(A[k * nc + k]);
            for (int i = k + 1; i < nr; i++) {
                double e = fabs(A[i * nc + k]);
                if (eta < e) eta = e;
            }
            if (eta == 0) {
                for (int i = 0; i < nc; i++) X[i] = 0;  // singular: no update
                return;
            }
            double sum2 = 0, inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) {
                A[i * nc + k] *= inv_eta;
                sum2 += A[i * nc + k] * A[i * nc + k];
            }
            double sigma = sqrt(sum2);
            if (A[k * nc + k] < 0) sigma = -sigma;
            A[k * nc + k] += sigma;
            A1[k] = sigma * A[k * nc + k];
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double sum = 0;
                for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
                double tau = sum / A1[k];
                for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
            }
        }
        for (int j = 0; j < nc; j++) {
            double tau = 0;
            for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
            tau /= A1[j];
            for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
        }
        X[nc - 1] = b[nc - 1] / A2[nc - 1];
        for (int i = nc - 2; i >= 0; i--) {
            double sum = 0;
            for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
            X[i] = (b[i] - sum) / A2[i];
        }
    }
    static void gauss_newton(const double* l, const double* rho, double* betas) {
        for (int it = 0; it < 5; it++) {
            double A[24], b[6], x[4];
            for (int i = 0; i < 6; i++) {
                const double* r = l + i * 10;
                double* a = A + i * 4;
                a[0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
                a[1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
                a[2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
                a[3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
                b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] +
                                 r[3] * betas[0] * betas[2] + r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                                 r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] + r[8] * betas[2] * betas[3] +
                                 r[9] * betas[3] * betas[3]);
            }
            qr_solve_6x4(A, b, x);
            for (int i = 0; i < 4; i++) betas[i] += x[i];
        }
    }
    double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++) {
            const double* v = ut + 12 * (11 - i);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
        }
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < 3; j++) pcs[3 * i + j] = -pcs[3 * i + j];
        }
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) {
                pc0[j] += pcs[3 * i + j];
                pw0[j] += pws[3 * i + j];
            }
        for (int j = 0; j < 3; j++) {
            pc0[j] /= n;
            pw0[j] /= n;
        }
        double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, U[9], W[3], V[9];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) {
                abt[3 * j] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i] - pw0[0]);
                abt[3 * j + 1] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 1] - pw0[1]);
                abt[3 * j + 2] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 2] - pw0[2]);
            }
        svd3(abt, U, W, V);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[i][j] = dot3(U + 3 * i, V + 3 * j);
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                           R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) {
            R[2][0] = -R[2][0];
            R[2][1] = -R[2][1];
            R[2][2] = -R[2][2];
        }
        for (int i = 0; i < 3; i++) t[i] = pc0[i] - dot3(R[i], pw0);
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* pw = &pws[3 * i];
            double Xc = dot3(R[0], pw) + t[0];
            double Yc = dot3(R[1], pw) + t[1];
            double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            double ue = uc + fu * Xc * inv_Zc;
            double ve = vc + fv * Yc * inv_Zc;
            double u = us[2 * i], v = us[2 * i + 1];
            sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    }
    void compute_pose(double Rout[9], double tout[3], double* ut_out) {
        choose_control_points();
        compute_barycentric_coordinates();
        std::vector<double> M(2 * n * 12);
        for (int i = 0; i < n; i++) {
            double* M1 = &M[2 * i * 12];
            double* M2 = M1 + 12;
            const double* as = &alphas[4 * i];
            double u = us[2 * i], v = us[2 * i + 1];
            for (int k = 0; k < 4; k++) {
                M1[3 * k] = as[k] * fu;
                M1[3 * k + 1] = 0.0;
                M1[3 * k + 2] = as[k] * (uc - u);
                M2[3 * k] = 0.0;
                M2[3 * k + 1] = as[k] * fv;
                M2[3 * k + 2] = as[k] * (vc - v);
            }
        }
        double mtm[144], ut[144], d[12];
        for (int a = 0; a < 12; a++)
            for (int b = 0; b < 12; b++) {
                double s = 0;
                for (int r = 0; r < 2 * n; r++) s += M[r * 12 + a] * M[r * 12 + b];
                mtm[a * 12 + b] = s;
            }
        jacobi_svd(mtm, 12, 12, ut, d);  // MtM is symmetric: rows == columns
        if (ut_out) memcpy(ut_out, ut, sizeof(ut));
        double l[60], rho[6];
        compute_L_6x10(ut, l);
        compute_rho(rho);
        double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
        for (int v = 1; v <= 3; v++) {
            find_betas(l, rho, v, Betas[v]);
            gauss_newton(l, rho, Betas[v]);
            rep[v] = compute_R_and_t(ut, Betas[v], Rs[v], ts[v]);
        }
        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) Rout[3 * i + j] = Rs[N][i][j];
            tout[i] = ts[N][i];
        }
    }
};

// solvePnP(SOLVEPNP_EPNP) on the subset idx[0..cnt): undistortPoints (no distortion, float output) + epnp.
void epnp_subset(const float* p3, const float* p2, const int32_t* idx, int cnt, const double K[4], double R[9],
                 double t[3], double* ut_out) {
    Epnp e;
    e.fu = K[0];
    e.fv = K[1];
    e.uc = K[2];
    e.vc = K[3];
    e.n = cnt;
    e.pws.resize(3 * cnt);
    e.us.resize(2 * cnt);
    e.alphas.resize(4 * cnt);
    e.pcs.resize(3 * cnt);
    const double ifx = 1. / K[0], ify = 1. / K[1];
    for (int i = 0; i < cnt; i++) {
        const int s = idx[i];
        for (int j = 0; j < 3; j++) e.pws[3 * i + j] = p3[3 * s + j];
        float xn = (float)(((double)p2[2 * s] - K[2]) * ifx);
        float yn = (float)(((double)p2[2 * s + 1] - K[3]) * ify);
        e.us[2 * i] = xn * e.fu + e.uc;
        e.us[2 * i + 1] = yn * e.fv + e.vc;
    }
    e.compute_pose(R, t, ut_out);
}

// PnPRansacCallback::computeError + RANSACPointSetRegistrator::findInliers for a model carried as (R, t).
int score_model(const float* p3, const float* p2, int n, const double K[4], const double R[9], const double t[3],
                float thr2, uint8_t* mask) {
    int good = 0;
    for (int i = 0; i < n; i++) {
        const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const float pu = (float)(x * K[0] + K[2]);
        const float pv = (float)(y * K[1] + K[3]);
        const float du = p2[2 * i] - pu, dv = p2[2 * i + 1] - pv;
        float a = du * du, b = dv * dv;
        float err = a + b;
        int f = err <= thr2;
        if (mask) mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

// ------------------------------------------------------------------------------------------- Rodrigues
void rodrigues_fwd(const double r[3], double R[9], double* J /* 3 x 9 or null */) {
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    const double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        const double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,  0,  rx, 0,  rx, ry + ry,
                                 rz,      0,  rz, 0,  0,       0,  rx, 0,  0,  ry, rx, ry, rz + rz};
        const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        const double rr[3] = {rx, ry, rz};
        for (int i = 0; i < 3; i++) {
            const double ri = rr[i];
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri,
                         a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x[i * 9 + k];
        }
    }
}

void rodrigues_inv(const double Rin[9], double r[3]) {
    double U[9], W[3], V[9], R[9];
    svd3(Rin, U, W, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    const double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            x = y = z = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            x = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            y = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            z = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= sqrt(x * x + y * y + z * z);
            x *= theta;
            y *= theta;
            z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth;
        y *= vth;
        z *= vth;
    }
    r[0] = x;
    r[1] = y;
    r[2] = z;
}

// ------------------------------------------------------------------------------------------- refinement
// cvProjectPoints2 (no distortion) residuals proj - meas for all points, optionally the 2n x 6 Jacobian.
void project_residuals(const double* M, const double* m, int n, const double K[4], const double p[6], double* err,
                       double* J) {
    double R[9], dRdr[27];
    rodrigues_fwd(p, R, J ? dRdr : nullptr);
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    for (int i = 0; i < n; i++) {
        const double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + p[3];
        double y = R[3] * X + R[4] * Y + R[5] * Z + p[4];
        double z = R[6] * X + R[7] * Y + R[8] * Z + p[5];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        err[2 * i] = (x * fx + cx) - m[2 * i];
        err[2 * i + 1] = (y * fy + cy) - m[2 * i + 1];
        if (J) {
            double* Jx = J + (2 * i) * 6;
            double* Jy = Jx + 6;
            const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                     X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                     X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                     X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                Jx[j] = fx * dxdr;
                Jy[j] = fy * dydr;
            }
            const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                Jx[3 + j] = fx * dxdt[j];
                Jy[3 + j] = fy * dydt[j];
            }
        }
    }
}

double norm_l2(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

// CvLevMarq(6 params, 2n errs, {MAX_ITER+EPS, 20, FLT_EPSILON}) driven as cvFindExtrinsicCameraParams2 does.
// Returns the number of accepted iterations; evals gets the number of residual evaluations.
int lm_refine(const double* M, const double* m, int n, const double K[4], double param[6], int* evals) {
    std::vector<double> err(2 * n), J(2 * n * 6);
    double prev[6], JtJ[36], JtErr[6];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0, n_eval = 0;
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    auto step = [&]() {
        const double lambda = exp(lambdaLg10 * log(10.));
        double A[36], x[6];
        memcpy(A, JtJ, sizeof(A));
        for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
        svd_solve(A, 6, 6, JtErr, 1, x);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - x[i];
    };
    // STARTED -> CALC_J
    project_residuals(M, m, n, K, param, err.data(), J.data());
    n_eval++;
    for (;;) {
        // CALC_J
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) {
                double s = 0;
                for (int r = 0; r < 2 * n; r++) s += J[r * 6 + a] * J[r * 6 + b];
                JtJ[a * 6 + b] = s;
            }
            double s = 0;
            for (int r = 0; r < 2 * n; r++) s += J[r * 6 + a] * err[r];
            JtErr[a] = s;
        }
        memcpy(prev, param, sizeof(prev));
        step();
        if (iters == 0) prevErrNorm = norm_l2(err.data(), 2 * n);
        // CHECK_ERR
        bool done = false;
        for (;;) {
            project_residuals(M, m, n, K, param, err.data(), nullptr);
            n_eval++;
            errNorm = norm_l2(err.data(), 2 * n);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    step();
                    continue;
                }
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            double d[6];
            for (int i = 0; i < 6; i++) d[i] = param[i] - prev[i];
            if (++iters >= max_iter || norm_l2(d, 6) / norm_l2(prev, 6) < epsilon) done = true;
            break;
        }
        if (done) break;
        prevErrNorm = errNorm;
        project_residuals(M, m, n, K, param, err.data(), J.data());
        n_eval++;
    }
    if (evals) *evals = n_eval;
    return iters;
}

// cvFindExtrinsicCameraParams2(useExtrinsicGuess = 0) on the points; init_model (R, t) is used when the
// structure is planar or has fewer than 6 points (deviation, see header).  Returns 1 if the DLT was used.
int solve_pnp_iterative(const double* M, const double* m, int n, const double K[4], const double init_R[9],
                        const double init_t[3], double param[6], int* lm_iters, int* lm_evals) {
    const double ifx = 1. / K[0], ify = 1. / K[1];
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) Mc[j] += M[3 * i + j];
    for (int j = 0; j < 3; j++) Mc[j] /= n;
    double MM[9], At[9], Vt[9], W[3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double s = 0;
            for (int i = 0; i < n; i++) s += (M[3 * i + a] - Mc[a]) * (M[3 * i + b] - Mc[b]);
            MM[a * 3 + b] = s;
        }
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) At[i * 3 + k] = MM[k * 3 + i];
    jacobi_svd(At, 3, 3, Vt, W);
    const bool planar = W[2] / W[1] < 1e-3;
    int used_dlt = 0;
    double R[9], t[3];
    if (planar || n < 6) {
        memcpy(R, init_R, sizeof(R));
        memcpy(t, init_t, sizeof(t));
    } else {
        used_dlt = 1;
        double LL[144], LV[144], LW[12];
        std::vector<double> L(2 * n * 12);
        for (int i = 0; i < n; i++) {
            const double x = -((m[2 * i] - K[2]) * ifx), y = -((m[2 * i + 1] - K[3]) * ify);
            const double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
            double* l = &L[2 * i * 12];
            l[0] = l[16] = X;
            l[1] = l[17] = Y;
            l[2] = l[18] = Z;
            l[3] = l[19] = 1.;
            l[4] = l[5] = l[6] = l[7] = 0.;
            l[12] = l[13] = l[14] = l[15] = 0.;
            l[8] = x * X;
            l[9] = x * Y;
            l[10] = x * Z;
            l[11] = x;
            l[20] = y * X;
            l[21] = y * Y;
            l[22] = y * Z;
            l[23] = y;
        }
        for (int a = 0; a < 12; a++)
            for (int b = 0; b < 12; b++) {
                double s = 0;
                for (int r = 0; r < 2 * n; r++) s += L[r * 12 + a] * L[r * 12 + b];
                LL[a * 12 + b] = s;
            }
        jacobi_svd(LL, 12, 12, LV, LW);
        double RR[9], tt[3];
        const double* v = LV + 11 * 12;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) RR[3 * i + j] = v[4 * i + j];
            tt[i] = v[4 * i + 3];
        }
        const double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
                           RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
        if (det < 0) {
            for (int i = 0; i < 9; i++) RR[i] = -RR[i];
            for (int i = 0; i < 3; i++) tt[i] = -tt[i];
        }
        const double sc = norm_l2(RR, 9);
        double U[9], W3[3], V[9];
        svd3(RR, U, W3, V);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
        const double f = norm_l2(R, 9) / sc;
        for (int i = 0; i < 3; i++) t[i] = tt[i] * f;
    }
    rodrigues_inv(R, param);
    param[3] = t[0];
    param[4] = t[1];
    param[5] = t[2];
    int ev = 0;
    int it = lm_refine(M, m, n, K, param, &ev);
    if (lm_iters) *lm_iters = it;
    if (lm_evals) *lm_evals = ev;
    return used_dlt;
}

}  // namespace

// =========================================================================================== C interface
extern "C" {

int orc_invert4x4(const double* T, double* out) { return invert4x4_lu(T, out); }

int orc_map_in_view(const float* pos, int n, const double* T_w_c, double fx, double fy, double cx, double cy,
                    int cols, int rows, int32_t* idx, float* px) {
    double T[16];
    if (!invert4x4_lu(T_w_c, T)) return -1;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        const double p[4] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 1};
        double res[3] = {0, 0, 0};
        for (int r = 0; r < 3; r++)
            for (int j = 0; j < 4; j++) res[r] += T[r * 4 + j] * p[j];
        const float pcx = (float)res[0], pcy = (float)res[1], pcz = (float)res[2];
        bool in = true;
        if (pcz < 0) in = false;
        const float u = (float)(fx * pcx / pcz + cx), v = (float)(fy * pcy / pcz + cy);
        if (!(u > 0 && v > 0 && u < (float)cols && v < (float)rows)) in = false;
        if (in) {
            idx[cnt] = i;
            px[2 * cnt] = u;
            px[2 * cnt + 1] = v;
            cnt++;
        }
    }
    return cnt;
}

int orc_pnp_subsets(int count, int model_points, int n_iters, int32_t* idx) {
    CvRng rng((uint64_t)-1);
    if (count < model_points) return -1;
    for (int it = 0; it < n_iters; it++) {
        int32_t* s = idx + it * model_points;
        for (int i = 0; i < model_points;) {
            int v, j;
            for (;;) {
                v = s[i] = rng.uniform(0, count);
                for (j = 0; j < i; j++)
                    if (v == s[j]) break;
                if (j == i) break;
            }
            i++;
        }
    }
    return 0;
}

int orc_epnp(const float* p3, const float* p2, const int32_t* idx, int cnt, const double* K4, double* R,
             double* t, double* ut) {
    epnp_subset(p3, p2, idx, cnt, K4, R, t, ut);
    return 0;
}

int orc_pnp_score(const float* p3, const float* p2, int n, const double* K4, const double* R, const double* t,
                  float reproj, uint8_t* mask) {
    const float thr2 = (float)((double)reproj * (double)reproj);
    return score_model(p3, p2, n, K4, R, t, thr2, mask);
}

int orc_rodrigues(const double* r, double* R, double* J) {
    rodrigues_fwd(r, R, J);
    return 0;
}
int orc_rodrigues_inv(const double* R, double* r) {
    rodrigues_inv(R, r);
    return 0;
}

int orc_solve_pnp_iterative(const double* M, const double* m, int n, const double* K4, const double* init_R,
                            const double* init_t, double* param, int* lm_iters, int* lm_evals) {
    return solve_pnp_iterative(M, m, n, K4, init_R, init_t, param, lm_iters, lm_evals);
}

// cv::solvePnPRansac(pts3d, pts2d, K, no distortion, rvec, tvec, false, iters, reproj, conf, inliers).
// Returns 1 (pose found) / 0; debug outputs may be null: models [iters x 12], counts [iters], info[4] =
// {best iteration, iterations actually run, dlt used, lm iterations}.
int orc_solve_pnp_ransac(const float* p3, const float* p2, int n, const double* K4, int iters, float reproj,
                         double confidence, double* rvec, double* tvec, int32_t* inliers, int* n_inliers,
                         double* models, int32_t* counts, int32_t* info) {
    const int model_points = 5;
    if (n_inliers) *n_inliers = 0;
    if (n < model_points || iters < 1) return 0;
    const float thr2 = (float)((double)reproj * (double)reproj);
    if (n == model_points) {  // solvePnPRansac: "model_points == npoints" -> one kernel run, every point an inlier
        int32_t all[5] = {0, 1, 2, 3, 4};
        double R[9], t[3];
        epnp_subset(p3, p2, all, model_points, K4, R, t, nullptr);
        rodrigues_inv(R, rvec);
        for (int i = 0; i < 3; i++) tvec[i] = t[i];
        for (int i = 0; i < n; i++)
            if (inliers) inliers[i] = i;
        if (n_inliers) *n_inliers = n;
        if (models) {
            memcpy(models, R, sizeof(R));
            memcpy(models + 9, t, sizeof(t));
        }
        if (counts) counts[0] = n;
        if (info) info[0] = 0, info[1] = 1, info[2] = 0, info[3] = 0;
        return 1;
    }
    std::vector<int32_t> subsets((size_t)iters * model_points);
    orc_pnp_subsets(n, model_points, iters, subsets.data());
    std::vector<uint8_t> mask(n), best_mask(n, 0);
    double bestR[9], bestT[3];
    int niters = iters, max_good = 0, best_it = -1, it = 0;
    for (; it < niters; it++) {
        double R[9], t[3];
        epnp_subset(p3, p2, &subsets[(size_t)it * model_points], model_points, K4, R, t, nullptr);
        const int good = score_model(p3, p2, n, K4, R, t, thr2, mask.data());
        if (models) {
            memcpy(models + it * 12, R, sizeof(R));
            memcpy(models + it * 12 + 9, t, sizeof(t));
        }
        if (counts) counts[it] = good;
        if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
            best_mask.swap(mask);
            memcpy(bestR, R, sizeof(R));
            memcpy(bestT, t, sizeof(t));
            max_good = good;
            best_it = it;
            niters = ransac_update_num_iters(confidence, (double)(n - good) / n, model_points, niters);
        }
    }
    if (info) {
        info[0] = best_it;
        info[1] = it;
        info[2] = info[3] = 0;
    }
    if (max_good <= 0) return 0;
    std::vector<double> M, m;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (best_mask[i]) {
            for (int j = 0; j < 3; j++) M.push_back(p3[3 * i + j]);
            m.push_back(p2[2 * i]);
            m.push_back(p2[2 * i + 1]);
            if (inliers) inliers[cnt] = i;
            cnt++;
        }
    if (n_inliers) *n_inliers = cnt;
    double param[6];
    int lm_it = 0, lm_ev = 0;
    int dlt = solve_pnp_iterative(M.data(), m.data(), cnt, K4, bestR, bestT, param, &lm_it, &lm_ev);
    if (info) {
        info[2] = dlt;
        info[3] = lm_it;
    }
    for (int i = 0; i < 3; i++) {
        rvec[i] = param[i];
        tvec[i] = param[3 + i];
    }
    return 1;
}
}
