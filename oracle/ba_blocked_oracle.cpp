// oracle/ba_blocked_oracle.cpp -- CPU ORACLE (test infrastructure, see oracle.h): the SAME algorithm as
// ba_oracle.cpp -- optimization::bundleAdjustment (reference src/optimization/g2o_ba.cpp:172-317) driving g2o's
// Levenberg-Marquardt with Schur complement and Huber kernel, SURVEY.md Appendix A.3 -- restated with the BLOCKED
// summation order the device declares, so that the MI355X solve can be checked bit for bit over all
// 50 iterations: every accept/reject decision, every lambda, every chi2, every pose and landmark.
//
// Why a second order: floating-point sums have no canonical order in the reference (g2o adds edge contributions in
// edge-id order, Eigen's fixed-size products fuse or not depending on the build), and on windows without a fixed
// vertex the LM path is chaotic at rounding level (tests/test_oracle_ba.py: permuting the edges of the sequential
// oracle moves the poses by ~2e-3).  The algorithm, its decisions and its tolerances are those of ba_oracle.cpp;
// only the association of the sums differs:
//   * landmarks are cut into G contiguous ranges (`wg_pt_start`, the device's plan); edges are ordered by
//     (range of their landmark, pose, original index);
//   * inside a range a Gram sum (pose blocks M^T M, Schur blocks U^T U) is one chain of std::fma over rows / columns in
//     that order, the Schur columns optionally cut into `nsplit` consecutive pieces added in order; ranges are added in
//     order starting from 0;
//   * scalar sums (chi2, predicted decrease): per-"thread" partials (index mod 512), a 64-lane xor butterfly, 8 waves in
//     order, ranges in order;
//   * whitened formulation (Omega = Lc^T Lc folded into the Jacobians), (H_ll + lambda I)^-1 = C C^T, reduced system by
//     right-looking LDL^T with r = 1 / d, l = c r, fma updates -- on the matrix PERMUTED the way Eigen::LDLT permutes it
//     (g2o's LinearSolverDense; the pivot order depends on the input diagonal only, see eigenPivotOrder); landmark
//     back-substitution per observation (C^T b_l - sum_e Y_e^T (A~_e dx), edges in ascending order);
//   * a failed factorisation leaves the solver's x what it was (g2o applies and scores that STALE step: see the trial loop);
//     its predicted decrease: landmark part per range (lane = landmark mod 64, one 64-lane butterfly) riding along with the
//     Schur exchange (added like its entries), pose part as one 64-lane butterfly, then + 1e-3;
//   * sin / cos by the fixed polynomial below instead of libm.
// PARITY UNPINNED like the rest of the oracle (oracle.h).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

constexpr int kThreads = 512, kWaves = 8, kHP = 28;

void quatNormalize(double* q) {
    if (q[0] < 0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
void quatFromR(const double R[9], double* q) {  // Eigen::Quaterniond(Matrix3d)
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
void quatToR(const double* q, double R[9]) {  // Eigen toRotationMatrix
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
void inv3(const double* A, double* I) {
    double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    double id = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    I[0] = c0 * id;
    I[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c1 * id;
    I[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c2 * id;
    I[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}
void invertRt(const double T[16], double Ri[9], double ti[3]) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    inv3(R, Ri);
    for (int i = 0; i < 3; ++i) ti[i] = -(Ri[3 * i] * T[3] + Ri[3 * i + 1] * T[7] + Ri[3 * i + 2] * T[11]);
}
// Cody-Waite reduction by pi/2 in three parts + degree-13 / 14 minimax polynomials on [-pi/4, pi/4]
void sincosFixed(double x, double* s, double* c) {
    const double k = std::rint(x * 0.63661977236758134308);
    double r = x - k * 1.57079632673412561417e+00;
    r = r - k * 6.07710050650619224932e-11;
    r = r - k * 2.02226624879595063154e-21;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 +
                      z * (8.33333333332248946124e-03 +
                           z * (-1.98412698298579493134e-04 +
                                z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 +
                      z * (-1.38888888888741095749e-03 +
                           z * (2.48015872894767294178e-05 +
                                z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = r + r * z * ps;
    const double cr = 1.0 - 0.5 * z + z * z * pc;
    const int q = ((int)k) & 3;
    *s = q == 0 ? sr : (q == 1 ? cr : (q == 2 ? -sr : -cr));
    *c = q == 0 ? cr : (q == 1 ? -sr : (q == 2 ? -cr : sr));
}
// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(u) * T   (pose = q[4] t[3] in P[0..6])
void poseOplus(double* P, const double* u) {
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
            V[i] = R[i];
        }
    } else {
        double st, ct;
        sincosFixed(theta, &st, &ct);
        double a = st / theta, b = (1 - ct) / (theta * theta), c = (theta - st) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double dq[4], dt[3], dR[9], nt[3];
    quatFromR(R, dq);
    quatNormalize(dq);
    for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    quatToR(dq, dR);
    const double* q = P;
    const double* t = P + 4;
    for (int i = 0; i < 3; ++i) nt[i] = dt[i] + dR[3 * i] * t[0] + dR[3 * i + 1] * t[1] + dR[3 * i + 2] * t[2];
    double nq[4] = {dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2] - dq[3] * q[3],
                    dq[0] * q[1] + dq[1] * q[0] + dq[2] * q[3] - dq[3] * q[2],
                    dq[0] * q[2] - dq[1] * q[3] + dq[2] * q[0] + dq[3] * q[1],
                    dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1] + dq[3] * q[0]};
    for (int i = 0; i < 4; ++i) P[i] = nq[i];
    for (int i = 0; i < 3; ++i) P[4 + i] = nt[i];
    quatNormalize(P);
}
void huber(double e, double delta, double& rho0, double& rho1) {
    double dsqr = delta * delta;
    if (e <= dsqr) {
        rho0 = e;
        rho1 = 1.;
    } else {
        double sqrte = std::sqrt(e);
        rho0 = 2 * sqrte * delta - dsqr;
        rho1 = delta / sqrte;
    }
}

// a block reduction as the device does it: 512 per-thread partials -> xor butterfly inside each 64-lane wave -> the 8 wave
// results added in order
double blockSum(const double* part) {
    double s = 0;
    for (int w = 0; w < kWaves; ++w) {
        double v[64], t[64];
        std::memcpy(v, part + 64 * w, sizeof(v));
        for (int o = 32; o > 0; o >>= 1) {
            for (int i = 0; i < 64; ++i) t[i] = v[i] + v[i ^ o];
            std::memcpy(v, t, sizeof(v));
        }
        s += v[0];
    }
    return s;
}

// Eigen::LDLT's pivot order (LDLT.h, ldlt_inplace<Lower>::unblocked): at step k the position of the largest |diagonal
// entry| among k .. n-1 -- FIRST maximum -- is swapped with k.  The factorisation is left-looking: the entries behind k
// still hold the INPUT diagonal when they are compared, so the order is a function of diag(S) alone: perm[k] = index (in S)
// of the row that ends up at position k.  A NaN on the diagonal: identity (the factorisation then fails at that pivot).
void eigenPivotOrder(const double* diag, int n, int* perm) {
    std::vector<double> a(n);
    bool nan = false;
    for (int i = 0; i < n; ++i) {
        perm[i] = i;
        a[i] = std::fabs(diag[i]);
        nan = nan || a[i] != a[i];
    }
    if (nan) return;
    for (int k = 0; k < n; ++k) {
        int big = k;
        for (int i = k + 1; i < n; ++i)
            if (a[i] > a[big]) big = i;
        std::swap(a[k], a[big]);
        std::swap(perm[k], perm[big]);
    }
}
// one wave's xor butterfly over 64 per-lane values (the device's wave_sum_d)
double waveSum(const double* part) {
    double v[64], t[64];
    std::memcpy(v, part, sizeof(v));
    for (int o = 32; o > 0; o >>= 1) {
        for (int i = 0; i < 64; ++i) t[i] = v[i] + v[i ^ o];
        std::memcpy(v, t, sizeof(v));
    }
    return v[0];
}

struct Range {  // one landmark range ("workgroup") of the plan
    int pt_lo = 0, Lg = 0, e_lo = 0, Eg = 0;
    std::vector<int> pose_start;  // F + 1 local offsets: edges of pose p inside the range
};

struct Blocked {
    // problem
    int F = 0, L = 0, E = 0, G = 1, nfree = 0, n = 0, nsplit = 1, groups = 1;
    bool fix_points = false;
    double f = 0, cx = 0, cy = 0, delta = 1, lc00 = 1, lc01 = 0, lc11 = 1;
    std::vector<int> pose_slot, slot_pose;
    std::vector<Range> rg;
    std::vector<int> e_pose, e_point;  // sorted by (range, pose, original)
    std::vector<double> e_uv;
    std::vector<std::vector<int>> pt_edges;  // per landmark: sorted edge indices (global, ascending)
    // state
    std::vector<double> P, Pbak;   // F x 8
    std::vector<double> Rm, Tm;    // F x 9, F x 3
    std::vector<double> pts, bak;  // L x 3
    // linearisation
    std::vector<double> M;    // E x 14
    std::vector<double> X;    // E x 6
    std::vector<double> Hll;  // L x 6
    std::vector<double> bl;   // L x 3
    std::vector<double> Hpp, bp;  // F x 36, F x 6

    double edgeError(int e, double* Xc, double* ew) const {
        const int p = e_pose[e], l = e_point[e];
        const double* R = &Rm[9 * p];
        const double* t = &Tm[3 * p];
        const double X0 = pts[3 * l], X1 = pts[3 * l + 1], X2 = pts[3 * l + 2];
        Xc[0] = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0];
        Xc[1] = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
        Xc[2] = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
        const double e0 = e_uv[2 * e] - (Xc[0] / Xc[2] * f + cx);
        const double e1 = e_uv[2 * e + 1] - (Xc[1] / Xc[2] * f + cy);
        ew[0] = lc00 * e0 + lc01 * e1;
        ew[1] = lc11 * e1;
        return ew[0] * ew[0] + ew[1] * ew[1];
    }
    double rangeChi2(const Range& r) const {
        double part[kThreads] = {0};
        for (int el = 0; el < r.Eg; ++el) {
            double Xc[3], ew[2], r0, r1;
            huber(edgeError(r.e_lo + el, Xc, ew), delta, r0, r1);
            part[el % kThreads] += r0;
        }
        return blockSum(part);
    }
    double robustChi2() const {
        if (G == 1) return rangeChi2(rg[0]);
        double c = 0;
        for (int g = 0; g < G; ++g) c += rangeChi2(rg[g]);
        return c;
    }
    void refreshRt() {
        for (int p = 0; p < F; ++p) {
            quatToR(&P[8 * p], &Rm[9 * p]);
            for (int i = 0; i < 3; ++i) Tm[3 * p + i] = P[8 * p + 4 + i];
        }
    }
};

}  // namespace

extern "C" {

void orc_eigen_pivot_order(const double* diag, int n, int32_t* perm) {
    std::vector<int> p(n);
    eigenPivotOrder(diag, n, p.data());
    for (int i = 0; i < n; ++i) perm[i] = p[i];
}

// wg_pt_start: G + 1 landmark range starts (the device's plan); trace (may be NULL): trace_cap rows {lambda, chi2, rho,
// accepted} per trial.  Same problem struct, same outputs as orc_bundle_adjustment.
int orc_bundle_adjustment_blocked(orc_ba_problem* in, int G, const int32_t* wg_pt_start, int nsplit, orc_ba_stats* st,
                                  double* trace, int trace_cap, int* trace_n) {
    for (int e = 0; e < in->n_edges; ++e)
        if (in->edge_pose[e] < 0 || in->edge_pose[e] >= in->n_poses || in->edge_point[e] < 0 || in->edge_point[e] >= in->n_points)
            return -1;
    if (G < 1 || wg_pt_start[0] != 0 || wg_pt_start[G] != in->n_points) return -2;
    Blocked B;
    B.F = in->n_poses;
    B.L = in->n_points;
    B.G = G;
    B.nsplit = (nsplit & 0xffff) < 1 ? 1 : (nsplit & 0xffff);
    B.groups = (nsplit >> 16) > 1 ? (nsplit >> 16) : 1;  // (bits 16 .. of `nsplit`: groups of the Schur exchange, see oracle.h)
    if (G % B.groups != 0) return -2;
    B.fix_points = in->fix_points != 0;
    B.f = in->focal;
    B.cx = in->cx;
    B.cy = in->cy;
    B.delta = in->huber_delta;
    {
        const double a = in->info[0], b = in->info[1], d = in->info[3];
        B.lc00 = std::sqrt(a);
        B.lc01 = b / B.lc00;
        B.lc11 = std::sqrt(d - B.lc01 * B.lc01);
    }
    const int F = B.F, L = B.L;
    B.pose_slot.assign(F, -1);
    for (int i = 0; i < F; ++i)
        if (!(in->pose_fixed && in->pose_fixed[i])) {
            B.pose_slot[i] = (int)B.slot_pose.size();
            B.slot_pose.push_back(i);
        }
    B.nfree = (int)B.slot_pose.size();
    B.n = 6 * B.nfree;
    const int n = B.n, nfree = B.nfree;
    // ---- the plan: active edges, ranges, edges sorted by (range, pose, original order)
    std::vector<int> owner(L, 0);
    for (int g = 0; g < G; ++g) {
        if (wg_pt_start[g + 1] < wg_pt_start[g]) return -2;
        for (int l = wg_pt_start[g]; l < wg_pt_start[g + 1]; ++l) owner[l] = g;
    }
    std::vector<int> act;
    for (int e = 0; e < in->n_edges; ++e) {
        if (B.pose_slot[in->edge_pose[e]] < 0 && B.fix_points) continue;  // initializeOptimization: all-fixed edges drop out
        act.push_back(e);
    }
    B.E = (int)act.size();
    const int E = B.E;
    std::vector<int> cnt((size_t)G * std::max(F, 1), 0);
    for (int e : act) cnt[(size_t)owner[in->edge_point[e]] * F + in->edge_pose[e]]++;
    B.rg.resize(G);
    std::vector<int> cur((size_t)G * std::max(F, 1), 0);
    {
        int acc = 0;
        for (int g = 0; g < G; ++g) {
            Range& r = B.rg[g];
            r.pt_lo = wg_pt_start[g];
            r.Lg = wg_pt_start[g + 1] - wg_pt_start[g];
            r.e_lo = acc;
            r.pose_start.assign(F + 1, 0);
            for (int p = 0; p < F; ++p) {
                r.pose_start[p] = acc - r.e_lo;
                cur[(size_t)g * F + p] = acc;
                acc += cnt[(size_t)g * F + p];
            }
            r.pose_start[F] = acc - r.e_lo;
            r.Eg = acc - r.e_lo;
        }
    }
    B.e_pose.resize(E);
    B.e_point.resize(E);
    B.e_uv.resize(2 * (size_t)E);
    for (int e : act) {
        const int k = cur[(size_t)owner[in->edge_point[e]] * F + in->edge_pose[e]]++;
        B.e_pose[k] = in->edge_pose[e];
        B.e_point[k] = in->edge_point[e];
        B.e_uv[2 * (size_t)k] = in->edge_uv[2 * (size_t)e];
        B.e_uv[2 * (size_t)k + 1] = in->edge_uv[2 * (size_t)e + 1];
    }
    B.pt_edges.assign(L, {});
    for (int k = 0; k < E; ++k) B.pt_edges[B.e_point[k]].push_back(k);
    // ---- state (g2o_ba.cpp:185-190, 208-215: T_w_c.inv() -> SE3Quat(R, t))
    B.P.assign(8 * (size_t)F, 0);
    B.Pbak = B.P;
    B.Rm.assign(9 * (size_t)F, 0);
    B.Tm.assign(3 * (size_t)F, 0);
    for (int p = 0; p < F; ++p) {
        double Ri[9], ti[3], q[4];
        invertRt(in->pose_T_w_c + 16 * p, Ri, ti);
        quatFromR(Ri, q);
        quatNormalize(q);
        for (int i = 0; i < 4; ++i) B.P[8 * p + i] = q[i];
        for (int i = 0; i < 3; ++i) B.P[8 * p + 4 + i] = ti[i];
    }
    B.refreshRt();
    B.pts.assign(in->points, in->points + 3 * (size_t)L);
    B.bak = B.pts;
    B.M.assign(14 * (size_t)E, 0);
    B.X.assign(6 * (size_t)E, 0);
    B.Hll.assign(6 * (size_t)L, 0);
    B.bl.assign(3 * (size_t)L, 0);
    B.Hpp.assign(36 * (size_t)F, 0);
    B.bp.assign(6 * (size_t)F, 0);
    std::vector<double> Cc(6 * (size_t)L, 0), cl(3 * (size_t)L, 0), rr(3 * (size_t)L, 0);
    const int nrow = n + 1;                      // rows of U: pose entries, then the rhs row
    std::vector<double> U;                       // per range: (3 Lg) columns x nrow
    std::vector<size_t> u_off(G + 1, 0);
    for (int g = 0; g < G; ++g) u_off[g + 1] = u_off[g] + (size_t)3 * B.rg[g].Lg * nrow;
    U.assign(u_off[G], 0);
    const int nlow = n * (n + 1) / 2 + n;
    std::vector<double> Gsum(std::max(nlow, 1), 0);
    std::vector<double> S((size_t)(n + 1) * (n + 1), 0), sol(std::max(n, 1), 0), dx(6 * (size_t)std::max(F, 1), 0);
    // the solver's x as g2o keeps it: it lives across trials and iterations and is only overwritten by a SUCCESSFUL solve
    // (sol / dx: pose part, xl: landmark part; zero before the first one)
    std::vector<double> xl(3 * (size_t)std::max(L, 1), 0), solNew(std::max(n, 1), 0);
    std::vector<int> perm(std::max(n, 1), 0);

    double lambda = 0, ni = 2;
    int it = 0, trials = 0, terminated = 0, ntrace = 0;
    double currentChi = B.robustChi2();
    const double chi0 = currentChi;
    const bool any_free = nfree > 0 || !B.fix_points;
    const bool do_schur = !B.fix_points && n > 0;

    for (it = 0; any_free && it < in->max_iterations; ++it) {
        // ---- LIN (EdgeProjectXYZ2UV::linearizeOplus, whitened)
        for (int e = 0; e < E; ++e) {
            double Xc[3], ew[2], r0, r1;
            const double chi = B.edgeError(e, Xc, ew);
            huber(chi, B.delta, r0, r1);
            const double sw = std::sqrt(r1);
            const int p = B.e_pose[e];
            const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = B.f;
            double* Mr = &B.M[14 * (size_t)e];
            if (B.pose_slot[p] >= 0) {
                const double J0[6] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f};
                const double J1[6] = {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};
                for (int c = 0; c < 6; ++c) {
                    Mr[c] = sw * (B.lc00 * J0[c] + B.lc01 * J1[c]);
                    Mr[7 + c] = sw * (B.lc11 * J1[c]);
                }
            } else {
                for (int c = 0; c < 6; ++c) Mr[c] = Mr[7 + c] = 0;
            }
            Mr[6] = sw * ew[0];
            Mr[13] = sw * ew[1];
            if (!B.fix_points) {
                const double* R = &B.Rm[9 * p];
                const double t0[3] = {f, 0, -x / z * f}, t1[3] = {0, f, -y / z * f};
                double* Xr = &B.X[6 * (size_t)e];
                for (int c = 0; c < 3; ++c) {
                    double j0 = -1. / z * (t0[0] * R[c] + t0[1] * R[3 + c] + t0[2] * R[6 + c]);
                    double j1 = -1. / z * (t1[0] * R[c] + t1[1] * R[3 + c] + t1[2] * R[6 + c]);
                    Xr[c] = sw * (B.lc00 * j0 + B.lc01 * j1);
                    Xr[3 + c] = sw * (B.lc11 * j1);
                }
            }
        }
        // ---- pose blocks [H_pp | -b_p] = M^T M: one fma chain per range over the rows of the pose, ranges in order
        for (int sl = 0; sl < nfree; ++sl) {
            const int p = B.slot_pose[sl];
            for (int i = 0; i < 7; ++i)
                for (int j = 0; j <= i; ++j) {
                    // (the device's first iteration adds the ranges' chains in range order; from the second one on the pose
                    // blocks ride along with the Schur exchange and are added the way that one adds: per group, then the groups)
                    const int Kp = (B.groups > 1 && do_schur && it > 0) ? B.groups : 1;
                    double tot = 0;
                    for (int k = 0; k < Kp; ++k) {
                        double gs = 0;
                        for (int g = k; g < G; g += Kp) {
                            const Range& r = B.rg[g];
                            double acc = 0;
                            for (int el = r.pose_start[p]; el < r.pose_start[p + 1]; ++el) {
                                const double* Mr = &B.M[14 * (size_t)(r.e_lo + el)];
                                acc = std::fma(Mr[i], Mr[j], acc);
                                acc = std::fma(Mr[7 + i], Mr[7 + j], acc);
                            }
                            gs = G == 1 ? acc : gs + acc;
                        }
                        tot = Kp == 1 ? gs : tot + gs;
                    }
                    if (i < 6) {
                        B.Hpp[36 * p + 6 * i + j] = tot;
                        B.Hpp[36 * p + 6 * j + i] = tot;
                    } else if (j < 6) {
                        B.bp[6 * p + j] = -tot;
                    }
                }
        }
        // ---- landmark blocks
        double maxdiag = 0;
        if (!B.fix_points)
            for (int l = 0; l < L; ++l) {
                double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
                for (int e : B.pt_edges[l]) {
                    const double* Xr = &B.X[6 * (size_t)e];
                    const double e0 = B.M[14 * (size_t)e + 6], e1 = B.M[14 * (size_t)e + 13];
                    h[0] += Xr[0] * Xr[0] + Xr[3] * Xr[3];
                    h[1] += Xr[0] * Xr[1] + Xr[3] * Xr[4];
                    h[2] += Xr[0] * Xr[2] + Xr[3] * Xr[5];
                    h[3] += Xr[1] * Xr[1] + Xr[4] * Xr[4];
                    h[4] += Xr[1] * Xr[2] + Xr[4] * Xr[5];
                    h[5] += Xr[2] * Xr[2] + Xr[5] * Xr[5];
                    b[0] -= Xr[0] * e0 + Xr[3] * e1;
                    b[1] -= Xr[1] * e0 + Xr[4] * e1;
                    b[2] -= Xr[2] * e0 + Xr[5] * e1;
                }
                for (int i = 0; i < 6; ++i) B.Hll[6 * (size_t)l + i] = h[i];
                for (int i = 0; i < 3; ++i) B.bl[3 * (size_t)l + i] = b[i];
                maxdiag = std::fmax(maxdiag, std::fmax(std::fabs(h[0]), std::fmax(std::fabs(h[3]), std::fabs(h[5]))));
            }
        if (it == 0) {  // computeLambdaInit
            double m = maxdiag;
            for (int p = 0; p < F; ++p)
                if (B.pose_slot[p] >= 0)
                    for (int j = 0; j < 6; ++j) m = std::fmax(m, std::fabs(B.Hpp[36 * p + 7 * j]));
            lambda = 1e-5 * m;
            ni = 2;
        }

        double rho = 0;
        int qmax = 0;
        do {
            // ---- (H_ll + lambda I)^-1 = C C^T, C^T b_l, U
            if (!B.fix_points) {
                for (int l = 0; l < L; ++l) {
                    const double* h = &B.Hll[6 * (size_t)l];
                    const double D[9] = {h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda};
                    double Di[9];
                    inv3(D, Di);
                    const double c00 = std::sqrt(Di[0]), c10 = Di[3] / c00, c20 = Di[6] / c00;
                    const double c11 = std::sqrt(Di[4] - c10 * c10), c21 = (Di[7] - c20 * c10) / c11;
                    const double c22 = std::sqrt(Di[8] - c20 * c20 - c21 * c21);
                    double* cc = &Cc[6 * (size_t)l];
                    cc[0] = c00;
                    cc[1] = c10;
                    cc[2] = c11;
                    cc[3] = c20;
                    cc[4] = c21;
                    cc[5] = c22;
                    const double* b = &B.bl[3 * (size_t)l];
                    cl[3 * (size_t)l] = c00 * b[0] + c10 * b[1] + c20 * b[2];
                    cl[3 * (size_t)l + 1] = c11 * b[1] + c21 * b[2];
                    cl[3 * (size_t)l + 2] = c22 * b[2];
                }
                if (do_schur)
                    for (int g = 0; g < G; ++g) {
                        const Range& r = B.rg[g];
                        double* Ug = &U[u_off[g]];
                        for (int ll = 0; ll < r.Lg; ++ll) {
                            const int l = r.pt_lo + ll;
                            const double* cc = &Cc[6 * (size_t)l];
                            for (int k = 0; k < 3; ++k) {
                                double* col = Ug + (size_t)(3 * ll + k) * nrow;
                                for (int row = 0; row < n; ++row) col[row] = 0.0;
                                col[n] = cl[3 * (size_t)l + k];
                            }
                            for (int e : B.pt_edges[l]) {
                                const int sl = B.pose_slot[B.e_pose[e]];
                                if (sl < 0) continue;
                                const double* Xr = &B.X[6 * (size_t)e];
                                const double* A = &B.M[14 * (size_t)e];
                                const double Y[6] = {Xr[0] * cc[0] + Xr[1] * cc[1] + Xr[2] * cc[3], Xr[1] * cc[2] + Xr[2] * cc[4], Xr[2] * cc[5],
                                                     Xr[3] * cc[0] + Xr[4] * cc[1] + Xr[5] * cc[3], Xr[4] * cc[2] + Xr[5] * cc[4], Xr[5] * cc[5]};
                                for (int k = 0; k < 3; ++k) {
                                    double* col = Ug + (size_t)(3 * ll + k) * nrow;
                                    for (int c = 0; c < 6; ++c) col[6 * sl + c] = col[6 * sl + c] + (A[c] * Y[k] + A[7 + c] * Y[3 + k]);
                                }
                            }
                        }
                    }
            }
            // ---- predicted decrease of the step the solver's x holds BEFORE this trial's solve, landmark part (needed if the
            // solve fails): per range, per-"thread" partials + block sum, then added over the ranges like a Schur entry
            double staleL = 0;
            if (do_schur) {
                const int K = B.groups;
                double total = 0;
                for (int k = 0; k < K; ++k) {
                    double sum = 0;
                    for (int g = k; g < G; g += K) {
                        const Range& r = B.rg[g];
                        double part[64] = {0};  // (one wave of the range's workgroup forms it: lane = landmark mod 64)
                        for (int ll = 0; ll < r.Lg; ++ll) {
                            const int l = r.pt_lo + ll;
                            for (int c = 0; c < 3; ++c)
                                part[ll % 64] += xl[3 * (size_t)l + c] * (lambda * xl[3 * (size_t)l + c] + B.bl[3 * (size_t)l + c]);
                        }
                        const double tot = waveSum(part);
                        sum = G == 1 ? tot : sum + tot;
                    }
                    total = K == 1 ? sum : total + sum;
                }
                staleL = total;
            }
            // ---- partial Schur systems: fma chains over the columns of a range (nsplit consecutive pieces), ranges in order
            if (do_schur) {
                for (int idx = 0; idx < nlow; ++idx) {
                    int i, j;
                    if (idx < n * (n + 1) / 2) {
                        i = (int)((std::sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                        while (i * (i + 1) / 2 > idx) --i;
                        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
                        j = idx - i * (i + 1) / 2;
                    } else {
                        i = n;
                        j = idx - n * (n + 1) / 2;
                    }
                    // ranges in order within a group (the ranges g = k mod K), then the K group sums in order (K = 1: all ranges)
                    const int K = B.groups;
                    double total = 0;
                    for (int k = 0; k < K; ++k) {
                        double sum = 0;
                        for (int g = k; g < G; g += K) {
                            const Range& r = B.rg[g];
                            const double* Ug = &U[u_off[g]];
                            const int ncol = 3 * r.Lg, msteps = (ncol + 3) / 4, msplit = (msteps + B.nsplit - 1) / B.nsplit;
                            double tot = 0;
                            for (int sp = 0; sp < B.nsplit; ++sp) {
                                double acc = 0;
                                const int c0 = std::min(4 * sp * msplit, ncol), c1 = std::min(4 * (sp + 1) * msplit, ncol);
                                for (int col = c0; col < c1; ++col) acc = std::fma(Ug[(size_t)col * nrow + j], Ug[(size_t)col * nrow + i], acc);
                                tot = sp == 0 ? acc : tot + acc;
                            }
                            sum = G == 1 ? tot : sum + tot;
                        }
                        total = K == 1 ? sum : total + sum;
                    }
                    Gsum[idx] = total;
                }
            }
            // ---- reduced system (lower triangle + rhs row n) in Eigen's pivot order, right-looking LDL^T, back-substitution
            int ok2 = 1;
            if (n > 0) {
                const int ld = n + 1;
                auto entry = [&](int i, int k) {  // S[i][k], k <= i < n
                    const double gsum = do_schur ? Gsum[i * (i + 1) / 2 + k] : 0.0;
                    const int pi = B.slot_pose[i / 6], pk = B.slot_pose[k / 6];
                    return ((pi == pk) ? B.Hpp[36 * pi + 6 * (i % 6) + (k % 6)] + (i == k ? lambda : 0.0) : 0.0) - gsum;
                };
                {
                    std::vector<double> diag(n);
                    for (int i = 0; i < n; ++i) diag[i] = entry(i, i);
                    eigenPivotOrder(diag.data(), n, perm.data());
                }
                std::fill(S.begin(), S.end(), 0.0);
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k <= i; ++k) S[(size_t)i * ld + k] = entry(std::max(perm[i], perm[k]), std::min(perm[i], perm[k]));
                for (int k = 0; k < n; ++k) {
                    const int pk = perm[k];
                    const double gsum = do_schur ? Gsum[n * (n + 1) / 2 + pk] : 0.0;
                    S[(size_t)n * ld + k] = B.bp[6 * B.slot_pose[pk / 6] + pk % 6] - gsum;
                }
                for (int j = 0; j < n && ok2; ++j) {
                    const double d = S[(size_t)j * ld + j];
                    // Eigen: "not positive" = some pivot < 0.  (A pivot in [0, 2^-500) or beyond 2^500, or a NaN, is not usable by the
                    // device's reciprocal chain either and fails the solve as well -- Eigen would carry on with it; unreachable here.)
                    if (!((d >= 0x1p-500) && (d <= 0x1p+500))) {
                        ok2 = 0;
                        break;
                    }
                    const double r = 1.0 / d;
                    for (int i = j + 1; i <= n; ++i) {
                        const double l = S[(size_t)i * ld + j] * r;
                        const int kend = i < n ? i : n - 1;
                        for (int k = j + 1; k <= kend; ++k) S[(size_t)i * ld + k] = std::fma(-l, S[(size_t)k * ld + j], S[(size_t)i * ld + k]);
                        S[(size_t)i * ld + n] = l;  // parked next to the row: column j of L is l_i (overwritten below)
                        // keep c_i until every row of this step has used it: store l in a side column
                    }
                    for (int i = j + 1; i <= n; ++i) S[(size_t)i * ld + j] = S[(size_t)i * ld + n];
                }
                if (ok2) {
                    for (int j = 0; j < n; ++j) solNew[j] = S[(size_t)n * ld + j];  // z = D^-1 L^-1 g
                    for (int i = n - 1; i >= 1; --i)
                        for (int j = 0; j < i; ++j) solNew[j] = std::fma(-S[(size_t)i * ld + j], solNew[i], solNew[j]);
                    for (int k = 0; k < n; ++k) sol[perm[k]] = solNew[k];  // x = P^T x'
                }
            }
            if (ok2)
                for (int p = 0; p < F; ++p)
                    for (int c = 0; c < 6; ++c) dx[6 * p + c] = B.pose_slot[p] >= 0 ? sol[6 * B.pose_slot[p] + c] : 0.0;
            const double lambda_used = lambda;
            ++trials;
            // ---- a failed solve: OptimizationAlgorithmLevenberg::solve applies _solver->x() all the same -- still the PREVIOUS
            // solution (LinearSolverDense::solve leaves x alone when LDLT is "not positive", BlockSolver::solve returns before the
            // landmark part) --, sets tempChi = DBL_MAX and scores the step with computeScale() of that stale x.  DBL_MAX is
            // finite: whenever the stale scale is negative, rho is positive and the stale step is ACCEPTED (lambda / 3).
            double staleRho = 0;
            bool staleApply = false;
            if (!ok2) {
                double part[64] = {0};
                for (int t = 0; t < 6 * F; ++t)
                    if (B.pose_slot[t / 6] >= 0) part[t % 64] += dx[t] * (lambda * dx[t] + B.bp[t]);
                double sc = staleL + waveSum(part);
                sc += 1e-3;
                staleRho = (currentChi - 1.7976931348623157e308) / sc;
                staleApply = staleRho > 0;
                if (!staleApply) {  // rejected: nothing to apply (push / update / pop leaves the state as it was)
                    rho = staleRho;
                    if (trace && ntrace < trace_cap) {
                        trace[4 * ntrace] = lambda_used;
                        trace[4 * ntrace + 1] = 1.7976931348623157e308;
                        trace[4 * ntrace + 2] = rho;
                        trace[4 * ntrace + 3] = 0.0;
                    }
                    ++ntrace;
                    lambda *= ni;
                    ni *= 2;
                    ++qmax;
                    continue;
                }
            }
            // ---- back-substitution of the landmarks, computeScale, push + apply
            double scale_r[256];
            if (G > 256) return -3;
            for (int g = 0; g < G; ++g) {
                const Range& r = B.rg[g];
                double part[kThreads] = {0};
                if (g == 0)
                    for (int t = 0; t < 6 * F && t < kThreads; ++t)
                        if (B.pose_slot[t / 6] >= 0) part[t] += dx[t] * (lambda * dx[t] + B.bp[t]);
                if (!B.fix_points) {
                    // r = C^T (b_l - W^T dx_p) = C^T b_l - sum over the landmark's observations, in ascending edge order, of
                    // Y^T (A~ dx_pose) with Y = X~ C: the 2-vector A~ dx as two fma chains over the six pose coordinates
                    for (int ll = 0; ll < r.Lg && ok2; ++ll) {
                        const int l = r.pt_lo + ll;
                        const double* cc = &Cc[6 * (size_t)l];
                        double rq[3] = {cl[3 * (size_t)l], cl[3 * (size_t)l + 1], cl[3 * (size_t)l + 2]};
                        if (do_schur)
                            for (int e : B.pt_edges[l]) {
                                const int sl = B.pose_slot[B.e_pose[e]];
                                if (sl < 0) continue;
                                const double* A = &B.M[14 * (size_t)e];
                                const double* Xr = &B.X[6 * (size_t)e];
                                double s0 = 0, s1 = 0;
                                for (int c = 0; c < 6; ++c) {
                                    s0 = std::fma(A[c], sol[6 * sl + c], s0);
                                    s1 = std::fma(A[7 + c], sol[6 * sl + c], s1);
                                }
                                const double Y[6] = {Xr[0] * cc[0] + Xr[1] * cc[1] + Xr[2] * cc[3], Xr[1] * cc[2] + Xr[2] * cc[4], Xr[2] * cc[5],
                                                     Xr[3] * cc[0] + Xr[4] * cc[1] + Xr[5] * cc[3], Xr[4] * cc[2] + Xr[5] * cc[4], Xr[5] * cc[5]};
                                for (int k = 0; k < 3; ++k) rq[k] = rq[k] - (Y[k] * s0 + Y[3 + k] * s1);
                            }
                        for (int k = 0; k < 3; ++k) rr[3 * (size_t)l + k] = rq[k];
                    }
                    for (int ll = 0; ll < r.Lg; ++ll) {
                        const int l = r.pt_lo + ll;
                        const double* rv = &rr[3 * (size_t)l];
                        const double* cc = &Cc[6 * (size_t)l];
                        double d[3] = {cc[0] * rv[0], cc[1] * rv[0] + cc[2] * rv[1], cc[3] * rv[0] + cc[4] * rv[1] + cc[5] * rv[2]};
                        if (!ok2)
                            for (int c = 0; c < 3; ++c) d[c] = xl[3 * (size_t)l + c];  // the stale step
                        for (int c = 0; c < 3; ++c) {
                            part[ll % kThreads] += d[c] * (lambda * d[c] + B.bl[3 * (size_t)l + c]);
                            xl[3 * (size_t)l + c] = d[c];
                            B.bak[3 * (size_t)l + c] = B.pts[3 * (size_t)l + c];
                            B.pts[3 * (size_t)l + c] += d[c];
                        }
                    }
                }
                scale_r[g] = blockSum(part);
            }
            B.Pbak = B.P;
            for (int p = 0; p < F; ++p)
                if (B.pose_slot[p] >= 0) poseOplus(&B.P[8 * p], &dx[6 * p]);
            B.refreshRt();
            double scale = 0, tempChi = 0;
            if (G == 1) {
                scale = scale_r[0];
                tempChi = B.rangeChi2(B.rg[0]);
            } else {
                for (int g = 0; g < G; ++g) {
                    tempChi += B.rangeChi2(B.rg[g]);
                    scale += scale_r[g];
                }
            }
            scale += 1e-3;
            const double chiAtState = tempChi;  // (what the next iteration's computeActiveErrors sees)
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = ok2 ? (currentChi - tempChi) / scale : staleRho;
            const bool accept = rho > 0 && std::isfinite(tempChi);
            if (trace && ntrace < trace_cap) {
                trace[4 * ntrace] = lambda_used;
                trace[4 * ntrace + 1] = tempChi;
                trace[4 * ntrace + 2] = rho;
                trace[4 * ntrace + 3] = accept ? 1.0 : 0.0;
            }
            ++ntrace;
            if (accept) {
                double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                alpha = std::fmin(alpha, 2. / 3.);
                lambda *= std::fmax(1. / 3., alpha);
                ni = 2;
                currentChi = chiAtState;
            } else {
                lambda *= ni;
                ni *= 2;
                B.P = B.Pbak;
                B.refreshRt();
                if (!B.fix_points) B.pts = B.bak;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) {
            terminated = 1;
            ++it;
            break;
        }
    }
    // ---- write-back (g2o_ba.cpp:298-316)
    for (int p = 0; p < F; ++p) {
        double R[9], T[16] = {0}, Ri[9], ti[3];
        quatToR(&B.P[8 * p], R);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
            T[4 * r + 3] = B.P[8 * p + 4 + r];
        }
        T[15] = 1;
        invertRt(T, Ri, ti);
        double* o = in->pose_T_w_c + 16 * p;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) o[4 * r + c] = Ri[3 * r + c];
            o[4 * r + 3] = ti[r];
        }
        o[12] = o[13] = o[14] = 0;
        o[15] = 1;
    }
    std::copy(B.pts.begin(), B.pts.end(), in->points);
    if (st) {
        st->iterations = it;
        st->trials = trials;
        st->terminated = terminated;
        st->chi2_initial = chi0;
        st->chi2_final = currentChi;
        st->lambda_final = lambda;
    }
    if (trace_n) *trace_n = ntrace;
    return 0;
}

}  // extern "C"
