// oracle/ba_oracle.cpp -- CPU ORACLE (test infrastructure, see oracle.h): scalar f64 restatement of
// optimization::bundleAdjustment (reference src/optimization/g2o_ba.cpp:172-317) including the g2o
// machinery it drives (not vendored): SparseOptimizer::optimize(50) with
// OptimizationAlgorithmLevenberg over BlockSolver<6,3> + LinearSolverDense, VertexSE3Expmap,
// VertexSBAPointXYZ, EdgeProjectXYZ2UV with RobustKernelHuber -- semantics per SURVEY.md Appendix A.3.
// PARITY UNPINNED (see oracle.h).  Since round 6 the dense reduced system is solved as g2o's LinearSolverDense solves it
// (Eigen::LDLT: its pivot order, its sign rule, its pseudo-inverse of D -- ldltEigen below) and a failed solve leaves the
// solver's x what it was, as BlockSolver / OptimizationAlgorithmLevenberg do (the LM loop applies and scores that stale
// step).  The simplification of rounds 1-5 -- unpivoted LDL^T, zero step after a failed solve -- is kept as rule 0
// (orc_ba_set_solver_rule) to quantify the difference.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct Pose {  // g2o::SE3Quat, world -> camera
    double q[4];  // w, x, y, z
    double t[3];
};

void quatNormalize(double* q) {
    if (q[0] < 0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

// Eigen::Quaterniond(Matrix3d)
void quatFromR(const double R[9], double* q) {
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = std::sqrt(tr + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

// Eigen::Quaterniond::toRotationMatrix
void quatToR(const double* q, double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}

bool inv3(const double A[9], double I[9]) {
    double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
    double id = 1.0 / det;
    I[0] = c0 * id;
    I[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c1 * id;
    I[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c2 * id;
    I[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return det != 0 && std::isfinite(id);
}

// 4x4 [R t; 0 1] inverse (cv::Mat::inv on an affine matrix) -> R^-1, -R^-1 t
void invertRt(const double T[16], double Ri[9], double ti[3]) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    inv3(R, Ri);
    for (int i = 0; i < 3; ++i) ti[i] = -(Ri[3 * i] * T[3] + Ri[3 * i + 1] * T[7] + Ri[3 * i + 2] * T[11]);
}

// g2o::SE3Quat::exp(update) * T   (VertexSE3Expmap::oplusImpl)
void poseOplus(Pose& P, const double* u) {
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
        std::memcpy(V, R, sizeof(R));
    } else {
        double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
               c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double dq[4], dt[3];
    quatFromR(R, dq);
    quatNormalize(dq);
    for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    // result = exp * P : t = dt + dR * P.t ; r = dq * P.q ; normalize
    double dR[9];
    quatToR(dq, dR);
    double nt[3];
    for (int i = 0; i < 3; ++i) nt[i] = dt[i] + dR[3 * i] * P.t[0] + dR[3 * i + 1] * P.t[1] + dR[3 * i + 2] * P.t[2];
    const double* a = dq;
    const double* b = P.q;
    double nq[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                    a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                    a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                    a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    std::memcpy(P.q, nq, sizeof(nq));
    std::memcpy(P.t, nt, sizeof(nt));
    quatNormalize(P.q);
}

struct Problem {
    int F, L, E;
    std::vector<Pose> poses;
    std::vector<double> pts;
    const int32_t *ep, *el;
    const double* uv;
    double f, cx, cy, info[4], delta;
    std::vector<uint8_t> poseFixed, ptFixed, edgeActive;
    std::vector<int> poseSlot, ptSlot;  // index into the free-variable vector, -1 if fixed
    int nFreePose = 0, nFreePt = 0;
};

void huber(double e, double delta, double rho[3]) {
    double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e;
        rho[1] = 1.;
        rho[2] = 0.;
    } else {
        double sqrte = std::sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

// EdgeProjectXYZ2UV::computeError; returns chi2 = e^T Omega e
double edgeError(const Problem& P, int e, double err[2], double Rm[9], double Xc[3]) {
    const Pose& T = P.poses[P.ep[e]];
    const double* X = &P.pts[3 * P.el[e]];
    quatToR(T.q, Rm);
    for (int i = 0; i < 3; ++i) Xc[i] = Rm[3 * i] * X[0] + Rm[3 * i + 1] * X[1] + Rm[3 * i + 2] * X[2] + T.t[i];
    err[0] = P.uv[2 * e] - (Xc[0] / Xc[2] * P.f + P.cx);
    err[1] = P.uv[2 * e + 1] - (Xc[1] / Xc[2] * P.f + P.cy);
    return err[0] * (P.info[0] * err[0] + P.info[1] * err[1]) + err[1] * (P.info[2] * err[0] + P.info[3] * err[1]);
}

double robustChi2(const Problem& P) {
    double s = 0;
    for (int e = 0; e < P.E; ++e) {
        if (!P.edgeActive[e]) continue;
        double err[2], R[9], Xc[3], rho[3];
        huber(edgeError(P, e, err, R, Xc), P.delta, rho);
        s += rho[0];
    }
    return s;
}

struct System {
    std::vector<double> Hpp, bp;  // F x 36, F x 6
    std::vector<double> Hll, bl;  // L x 9,  L x 3
    std::vector<double> W;        // E x 18 (6x3, pose rows)
};

// BlockSolver::buildSystem: linearizeOplus + constructQuadraticForm over the active edges
void buildSystem(const Problem& P, System& S) {
    S.Hpp.assign((size_t)P.F * 36, 0);
    S.bp.assign((size_t)P.F * 6, 0);
    S.Hll.assign((size_t)P.L * 9, 0);
    S.bl.assign((size_t)P.L * 3, 0);
    S.W.assign((size_t)P.E * 18, 0);
    for (int e = 0; e < P.E; ++e) {
        if (!P.edgeActive[e]) continue;
        double err[2], R[9], Xc[3], rho[3];
        double chi = edgeError(P, e, err, R, Xc);
        huber(chi, P.delta, rho);
        const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = P.f;
        double tmp[6] = {f, 0, -x / z * f, 0, f, -y / z * f};
        double Jx[6];  // 2x3 = -1/z * tmp * R
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c)
                Jx[3 * r + c] = -1. / z * (tmp[3 * r] * R[c] + tmp[3 * r + 1] * R[3 + c] + tmp[3 * r + 2] * R[6 + c]);
        double Jp[12] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f,  -1. / z * f, 0,           x / z2 * f,
                         (1 + y * y / z2) * f, -x * y / z2 * f,   -x / z * f, 0,           -1. / z * f, y / z2 * f};
        double Om[4] = {rho[1] * P.info[0], rho[1] * P.info[1], rho[1] * P.info[2], rho[1] * P.info[3]};
        // omega_r = -Omega * e * rho'
        double orr[2] = {-(P.info[0] * err[0] + P.info[1] * err[1]) * rho[1],
                         -(P.info[2] * err[0] + P.info[3] * err[1]) * rho[1]};
        const int p = P.ep[e], l = P.el[e];
        const bool pf = !P.poseFixed[p], lf = !P.ptFixed[l];
        double OJp[12], OJx[6];
        for (int c = 0; c < 6; ++c) {
            OJp[c] = Om[0] * Jp[c] + Om[1] * Jp[6 + c];
            OJp[6 + c] = Om[2] * Jp[c] + Om[3] * Jp[6 + c];
        }
        for (int c = 0; c < 3; ++c) {
            OJx[c] = Om[0] * Jx[c] + Om[1] * Jx[3 + c];
            OJx[3 + c] = Om[2] * Jx[c] + Om[3] * Jx[3 + c];
        }
        if (pf) {
            for (int i = 0; i < 6; ++i) {
                S.bp[6 * p + i] += Jp[i] * orr[0] + Jp[6 + i] * orr[1];
                for (int j = 0; j < 6; ++j) S.Hpp[36 * (size_t)p + 6 * i + j] += Jp[i] * OJp[j] + Jp[6 + i] * OJp[6 + j];
            }
        }
        if (lf) {
            for (int i = 0; i < 3; ++i) {
                S.bl[3 * l + i] += Jx[i] * orr[0] + Jx[3 + i] * orr[1];
                for (int j = 0; j < 3; ++j) S.Hll[9 * (size_t)l + 3 * i + j] += Jx[i] * OJx[j] + Jx[3 + i] * OJx[3 + j];
            }
        }
        if (pf && lf)
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) S.W[18 * (size_t)e + 3 * i + j] = Jp[i] * OJx[j] + Jp[6 + i] * OJx[3 + j];
    }
}

// ---- the dense solve of the reduced system: g2o::LinearSolverDense::solve = Eigen::LDLT<MatrixXd>::compute(H);
// if (isPositive()) x = solve(b), else leave x untouched and return false  (g2o_ba.cpp:196-197 picks that solver).
// Two rules, selected by orc_ba_set_solver_rule():
//   1 (default) "eigen": Eigen 3.3 LDLT.h, ldlt_inplace<Lower>::unblocked restated -- at step k the largest |diagonal entry|
//      among positions k.. (FIRST maximum; the entries k.. still hold the INPUT matrix's values: the left-looking update only
//      touches column k) is swapped to k, d_k = a_kk - sum_j l_kj^2 d_j, column k -= A20 (D l_k), column k /= d_k unless
//      d_k == 0; the factorisation never stops at a bad pivot; sign: Zero -> PositiveSemiDef on the first d > 0 /
//      NegativeSemiDef on the first d < 0, a later pivot of the other sign -> Indefinite; isPositive() = PositiveSemiDef or
//      Zero, i.e. "no negative pivot" (zero pivots pass); solve = P^T L^-T D^+ L^-1 P b with D^+ zeroing rows whose
//      |d| <= DBL_MIN.  Inner products in index order (Eigen's own order depends on its SIMD width: unpinned).
//   0 "legacy" (rounds 1-5): unpivoted LDL^T, fails at the first pivot that is not > 0.
int g_solver_rule = 1;
int g_counters[4] = {0, 0, 0, 0};  // of the last orc_bundle_adjustment: failed solves, failed solves whose stale step was accepted

bool ldltEigen(std::vector<double>& A, const std::vector<double>& b, std::vector<double>& x, int n) {
    auto M = [&](int i, int j) -> double& { return A[(size_t)i * n + j]; };
    std::vector<int> tr(n);
    std::vector<double> temp(n);
    int sign = 0;  // 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite
    for (int k = 0; k < n; ++k) {
        int big = k;
        double best = std::fabs(M(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M(i, i)) > best) best = std::fabs(M(i, i)), big = i;
        tr[k] = big;
        if (big != k) {  // symmetric transposition on the lower triangle
            for (int j = 0; j < k; ++j) std::swap(M(k, j), M(big, j));
            for (int i = big + 1; i < n; ++i) std::swap(M(i, k), M(i, big));
            std::swap(M(k, k), M(big, big));
            for (int i = k + 1; i < big; ++i) std::swap(M(i, k), M(big, i));
        }
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = M(j, j) * M(k, j);
            double s = 0;
            for (int j = 0; j < k; ++j) s += M(k, j) * temp[j];
            M(k, k) -= s;
            for (int i = k + 1; i < n; ++i) {
                double t = 0;
                for (int j = 0; j < k; ++j) t += M(i, j) * temp[j];
                M(i, k) -= t;
            }
        }
        const double akk = M(k, k);
        const bool valid = std::fabs(akk) > 0;
        if (k == 0 && !valid) {  // the whole diagonal is zero: ZeroSign, identity transpositions
            for (int j = 0; j < n; ++j) tr[j] = j;
            break;
        }
        if (valid)
            for (int i = k + 1; i < n; ++i) M(i, k) /= akk;
        if (sign == 1) {
            if (akk < 0) sign = 2;
        } else if (sign == -1) {
            if (akk > 0) sign = 2;
        } else if (sign == 0) {
            if (akk > 0) sign = 1;
            else if (akk < 0) sign = -1;
        }
    }
    if (!(sign == 1 || sign == 0)) return false;  // LinearSolverDense: x stays what it was
    x = b;
    for (int k = 0; k < n; ++k) std::swap(x[k], x[tr[k]]);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < i; ++k) x[i] -= M(i, k) * x[k];
    for (int i = 0; i < n; ++i) x[i] = std::fabs(M(i, i)) > DBL_MIN ? x[i] / M(i, i) : 0.0;
    for (int i = n - 1; i >= 0; --i)
        for (int k = i + 1; k < n; ++k) x[i] -= M(k, i) * x[k];
    for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[tr[k]]);
    return true;
}

// unpivoted LDL^T of a dense symmetric n x n (row-major), solves in place; false if a pivot <= 0
bool ldltSolve(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < i; ++k) b[i] -= A[(size_t)i * n + k] * b[k];
    for (int i = 0; i < n; ++i) b[i] /= A[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; --i)
        for (int k = i + 1; k < n; ++k) b[i] -= A[(size_t)k * n + i] * b[k];
    return true;
}

// BlockSolver::solve with setLambda(lambda): Schur complement on the free points, dense solve of the
// reduced pose system, back-substitution.  dxp: F x 6, dxl: L x 3 (zero for fixed vertices).
bool solveSystem(const Problem& P, const System& S, double lambda, std::vector<double>& dxp,
                 std::vector<double>& dxl) {
    const int n = 6 * P.nFreePose;
    std::vector<double> A((size_t)n * n, 0), g(n, 0);
    if (g_solver_rule == 0 || dxp.size() != (size_t)P.F * 6) {  // (rule 1: x keeps its last content when the solve fails)
        dxp.assign((size_t)P.F * 6, 0);
        dxl.assign((size_t)P.L * 3, 0);
    }
    for (int p = 0; p < P.F; ++p) {
        int s = P.poseSlot[p];
        if (s < 0) continue;
        for (int i = 0; i < 6; ++i) {
            g[6 * s + i] = S.bp[6 * p + i];
            for (int j = 0; j < 6; ++j)
                A[(size_t)(6 * s + i) * n + 6 * s + j] = S.Hpp[36 * (size_t)p + 6 * i + j] + (i == j ? lambda : 0.0);
        }
    }
    // edges grouped by point
    std::vector<std::vector<int>> ptEdges(P.L);
    for (int e = 0; e < P.E; ++e)
        if (P.edgeActive[e] && !P.ptFixed[P.el[e]] && !P.poseFixed[P.ep[e]]) ptEdges[P.el[e]].push_back(e);
    std::vector<double> Dinv((size_t)P.L * 9, 0);
    for (int l = 0; l < P.L; ++l) {
        if (P.ptFixed[l]) continue;
        double D[9];
        for (int i = 0; i < 9; ++i) D[i] = S.Hll[9 * (size_t)l + i] + (i % 4 == 0 ? lambda : 0.0);
        double* Di = &Dinv[9 * (size_t)l];
        inv3(D, Di);
        const double* bl = &S.bl[3 * l];
        for (int e1 : ptEdges[l]) {
            const double* W1 = &S.W[18 * (size_t)e1];
            double WD[18];  // W1 * Dinv (6x3)
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j)
                    WD[3 * i + j] = W1[3 * i] * Di[j] + W1[3 * i + 1] * Di[3 + j] + W1[3 * i + 2] * Di[6 + j];
            int s1 = P.poseSlot[P.ep[e1]];
            for (int i = 0; i < 6; ++i) g[6 * s1 + i] -= WD[3 * i] * bl[0] + WD[3 * i + 1] * bl[1] + WD[3 * i + 2] * bl[2];
            for (int e2 : ptEdges[l]) {
                const double* W2 = &S.W[18 * (size_t)e2];
                int s2 = P.poseSlot[P.ep[e2]];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j)
                        A[(size_t)(6 * s1 + i) * n + 6 * s2 + j] -=
                            WD[3 * i] * W2[3 * j] + WD[3 * i + 1] * W2[3 * j + 1] + WD[3 * i + 2] * W2[3 * j + 2];
            }
        }
    }
    if (n > 0 && g_solver_rule == 1) {
        // BlockSolver::solve fills the upper block triangle of H_schur (i1 <= i2), LinearSolverDense::solve mirrors it
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (i / 6 != j / 6) A[(size_t)i * n + j] = A[(size_t)j * n + i];
        std::vector<double> xs;
        if (!ldltEigen(A, g, xs, n)) return false;
        g = xs;
    } else if (n > 0 && !ldltSolve(A, g, n)) {
        return false;
    }
    for (int p = 0; p < P.F; ++p)
        if (P.poseSlot[p] >= 0)
            for (int i = 0; i < 6; ++i) dxp[6 * p + i] = g[6 * P.poseSlot[p] + i];
    for (int l = 0; l < P.L; ++l) {
        if (P.ptFixed[l]) continue;
        double r[3] = {S.bl[3 * l], S.bl[3 * l + 1], S.bl[3 * l + 2]};
        for (int e : ptEdges[l]) {
            const double* W = &S.W[18 * (size_t)e];
            const double* xp = &dxp[6 * P.ep[e]];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 6; ++i) r[j] -= W[3 * i + j] * xp[i];
        }
        const double* Di = &Dinv[9 * (size_t)l];
        for (int i = 0; i < 3; ++i) dxl[3 * l + i] = Di[3 * i] * r[0] + Di[3 * i + 1] * r[1] + Di[3 * i + 2] * r[2];
    }
    return true;
}

void loadProblem(const orc_ba_problem* in, Problem& P) {
    P.F = in->n_poses;
    P.L = in->n_points;
    P.E = in->n_edges;
    P.ep = in->edge_pose;
    P.el = in->edge_point;
    P.uv = in->edge_uv;
    P.f = in->focal;
    P.cx = in->cx;
    P.cy = in->cy;
    std::memcpy(P.info, in->info, sizeof(P.info));
    P.delta = in->huber_delta;
    P.poses.resize(P.F);
    for (int i = 0; i < P.F; ++i) {  // g2o_ba.cpp:185-190, 208-215: T_w_c.inv() -> SE3Quat(R, t)
        double Ri[9];
        invertRt(in->pose_T_w_c + 16 * i, Ri, P.poses[i].t);
        quatFromR(Ri, P.poses[i].q);
        quatNormalize(P.poses[i].q);
    }
    P.pts.assign(in->points, in->points + 3 * (size_t)P.L);
    P.poseFixed.assign(P.F, 0);
    if (in->pose_fixed)
        for (int i = 0; i < P.F; ++i) P.poseFixed[i] = in->pose_fixed[i] ? 1 : 0;
    P.ptFixed.assign(P.L, in->fix_points ? 1 : 0);
    P.poseSlot.assign(P.F, -1);
    P.ptSlot.assign(P.L, -1);
    P.nFreePose = P.nFreePt = 0;
    for (int i = 0; i < P.F; ++i)
        if (!P.poseFixed[i]) P.poseSlot[i] = P.nFreePose++;
    for (int i = 0; i < P.L; ++i)
        if (!P.ptFixed[i]) P.ptSlot[i] = P.nFreePt++;
    P.edgeActive.assign(P.E, 1);  // initializeOptimization drops edges whose vertices are all fixed
    for (int e = 0; e < P.E; ++e)
        if (P.poseFixed[P.ep[e]] && P.ptFixed[P.el[e]]) P.edgeActive[e] = 0;
}

}  // namespace

extern "C" {

int orc_bundle_adjustment(orc_ba_problem* in, orc_ba_stats* st) {
    for (int e = 0; e < in->n_edges; ++e)
        if (in->edge_pose[e] < 0 || in->edge_pose[e] >= in->n_poses || in->edge_point[e] < 0 ||
            in->edge_point[e] >= in->n_points)
            return -1;
    Problem P;
    loadProblem(in, P);
    System S;
    g_counters[0] = g_counters[1] = 0;
    double lambda = 0, ni = 2;
    int it = 0, trials = 0, terminated = 0;
    std::vector<double> dxp, dxl;  // the solver's x: lives across trials and iterations
    double chi0 = robustChi2(P), chiFinal = chi0;
    const bool anyFree = (P.nFreePose + P.nFreePt) > 0;
    for (it = 0; anyFree && it < in->max_iterations; ++it) {
        double currentChi = robustChi2(P), tempChi = currentChi;
        buildSystem(P, S);
        if (it == 0) {  // computeLambdaInit: tau * max diagonal over the free vertices
            double maxDiag = 0;
            for (int p = 0; p < P.F; ++p)
                if (!P.poseFixed[p])
                    for (int j = 0; j < 6; ++j) maxDiag = std::max(std::fabs(S.Hpp[36 * (size_t)p + 7 * j]), maxDiag);
            for (int l = 0; l < P.L; ++l)
                if (!P.ptFixed[l])
                    for (int j = 0; j < 3; ++j) maxDiag = std::max(std::fabs(S.Hll[9 * (size_t)l + 4 * j]), maxDiag);
            lambda = 1e-5 * maxDiag;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            std::vector<Pose> savedPoses = P.poses;  // _optimizer->push()
            std::vector<double> savedPts = P.pts;
            bool ok2 = solveSystem(P, S, lambda, dxp, dxl);
            trials++;
            if (!ok2) g_counters[0]++;
            if (!ok2 && g_solver_rule == 0) {
                std::fill(dxp.begin(), dxp.end(), 0.0);
                std::fill(dxl.begin(), dxl.end(), 0.0);
            }
            // (rule 1: OptimizationAlgorithmLevenberg::solve applies _solver->x() whether or not the solve succeeded; after a
            // failed solve that is still the PREVIOUS solution -- BlockSolver::solve returns before touching x -- and
            // computeScale() below sees it too.  Zero before the first successful solve.)
            for (int p = 0; p < P.F; ++p)
                if (!P.poseFixed[p]) poseOplus(P.poses[p], &dxp[6 * p]);
            for (int l = 0; l < P.L; ++l)
                if (!P.ptFixed[l])
                    for (int i = 0; i < 3; ++i) P.pts[3 * l + i] += dxl[3 * l + i];
            tempChi = robustChi2(P);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;  // computeScale
            for (int p = 0; p < P.F; ++p)
                if (!P.poseFixed[p])
                    for (int i = 0; i < 6; ++i) scale += dxp[6 * p + i] * (lambda * dxp[6 * p + i] + S.bp[6 * p + i]);
            for (int l = 0; l < P.L; ++l)
                if (!P.ptFixed[l])
                    for (int i = 0; i < 3; ++i) scale += dxl[3 * l + i] * (lambda * dxl[3 * l + i] + S.bl[3 * l + i]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                if (!ok2) g_counters[1]++;
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = ok2 ? tempChi : robustChi2(P);  // (the next iteration evaluates the state it finds)
            } else {
                lambda *= ni;
                ni *= 2;
                P.poses = savedPoses;  // pop
                P.pts = savedPts;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        chiFinal = currentChi;
        if (qmax == 10 || rho == 0) {
            terminated = 1;
            ++it;
            break;
        }
    }
    // write-back (g2o_ba.cpp:298-316): SE3Quat -> (R, t) -> 4x4 -> inverse
    for (int i = 0; i < P.F; ++i) {
        double R[9], Ri[9], T[16] = {0}, ti[3];
        quatToR(P.poses[i].q, R);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
            T[4 * r + 3] = P.poses[i].t[r];
        }
        T[15] = 1;
        invertRt(T, Ri, ti);
        double* o = in->pose_T_w_c + 16 * i;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) o[4 * r + c] = Ri[3 * r + c];
            o[4 * r + 3] = ti[r];
        }
        o[12] = o[13] = o[14] = 0;
        o[15] = 1;
    }
    std::copy(P.pts.begin(), P.pts.end(), in->points);
    if (st) {
        st->iterations = it;
        st->trials = trials;
        st->terminated = terminated;
        st->chi2_initial = chi0;
        st->chi2_final = chiFinal;
        st->lambda_final = lambda;
    }
    return 0;
}

// Eigen::LDLT<MatrixXd>(A).isPositive() ? x = solve(b) : x untouched -- the restatement the LM loop uses, for known-answer tests
int orc_ldlt_eigen(const double* A, const double* b, int n, double* x) {
    std::vector<double> M(A, A + (size_t)n * n), rhs(b, b + n), xs;
    if (!ldltEigen(M, rhs, xs, n)) return 0;
    std::copy(xs.begin(), xs.end(), x);
    return 1;
}
void orc_ba_set_solver_rule(int rule) { g_solver_rule = rule ? 1 : 0; }
int orc_ba_get_solver_rule(void) { return g_solver_rule; }
void orc_ba_last_counters(int32_t* out) { out[0] = g_counters[0]; out[1] = g_counters[1]; }

int orc_ba_linearize(const orc_ba_problem* in, double* H, double* b, double* chi2, int ncap) {
    Problem P;
    loadProblem(in, P);
    const int n = 6 * P.nFreePose + 3 * P.nFreePt;
    if (n > ncap) return -3;
    System S;
    buildSystem(P, S);
    std::fill(H, H + (size_t)n * n, 0.0);
    std::fill(b, b + n, 0.0);
    const int po = 6 * P.nFreePose;
    for (int p = 0; p < P.F; ++p) {
        int s = P.poseSlot[p];
        if (s < 0) continue;
        for (int i = 0; i < 6; ++i) {
            b[6 * s + i] = S.bp[6 * p + i];
            for (int j = 0; j < 6; ++j) H[(size_t)(6 * s + i) * n + 6 * s + j] = S.Hpp[36 * (size_t)p + 6 * i + j];
        }
    }
    for (int l = 0; l < P.L; ++l) {
        int s = P.ptSlot[l];
        if (s < 0) continue;
        for (int i = 0; i < 3; ++i) {
            b[po + 3 * s + i] = S.bl[3 * l + i];
            for (int j = 0; j < 3; ++j) H[(size_t)(po + 3 * s + i) * n + po + 3 * s + j] = S.Hll[9 * (size_t)l + 3 * i + j];
        }
    }
    for (int e = 0; e < P.E; ++e) {
        int sp = P.poseSlot[P.ep[e]], sl = P.ptSlot[P.el[e]];
        if (!P.edgeActive[e] || sp < 0 || sl < 0) continue;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 3; ++j) {
                H[(size_t)(6 * sp + i) * n + po + 3 * sl + j] += S.W[18 * (size_t)e + 3 * i + j];
                H[(size_t)(po + 3 * sl + j) * n + 6 * sp + i] += S.W[18 * (size_t)e + 3 * i + j];
            }
    }
    if (chi2) *chi2 = robustChi2(P);
    return n;
}

}  // extern "C"
