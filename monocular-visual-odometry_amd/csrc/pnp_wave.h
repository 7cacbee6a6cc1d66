// csrc/pnp_wave.h -- single-wave (64 lanes) building blocks of the PnP RANSAC kernels (track_kernels.hip).
//
// cv::solvePnPRansac as the reference calls it (src/vo/vo.cpp:318-329): one wave per RANSAC hypothesis (5-point
// EPnP + scoring of all pairs) and one wave for the final DLT + Levenberg-Marquardt refinement on the inliers.
//
// Style: SPMD with the lane loop spelled out.  Code outside PW_LANES is uniform (every lane of the workgroup
// computes the same values in registers); PW_LANES(l, NL) { ... } is the per-lane part of a workgroup of NL lanes;
// PW_WAVES(w, NW) { ... } is code that is uniform inside a wave but differs between the NW waves (no PW_SYNC
// inside); data crosses lanes only through the LDS struct, separated by PW_SYNC().  The including translation
// unit defines
//     PW_FN           function qualifier            (__device__ __forceinline__)
//     PW_LANES(l, NL) the lane loop                 (one trip with l = threadIdx.x; blockDim.x == NL)
//     PW_WAVES(w, NW) the wave loop                 (one trip with w = threadIdx.x / 64)
//     PW_SYNC()       LDS hand-over between phases  (__syncthreads())
//     PW_UNROLL       loop unrolling pragma
// tests/sim/pnp_wave_sim.cpp compiles the same header with an explicit 64-trip lane loop so that the kernel logic
// can be checked bit for bit against the CPU restatement on a machine without a GPU; the library itself has no CPU path.
//
// All arithmetic is IEEE double add/mul/div/sqrt in a fixed order (-ffp-contract=off), so the RANSAC stage is
// bit-reproducible; sin/cos/acos/exp/log only appear in the refinement.
#ifndef MVO_PNP_WAVE_H
#define MVO_PNP_WAVE_H
#include <float.h>
#include <math.h>
#include <stdint.h>

#if !defined(PW_FN) || !defined(PW_LANES) || !defined(PW_WAVES) || !defined(PW_SYNC) || !defined(PW_UNROLL)
#error "define PW_FN, PW_LANES, PW_WAVES, PW_SYNC and PW_UNROLL before including pnp_wave.h"
#endif

namespace pw {

constexpr int kWave = 64;
constexpr int kHypLanes = 192;  // one hypothesis per workgroup of 3 waves: the three EPnP beta variants run side by side
constexpr int kRefLanes = 64;   // the refinement runs on one wave
constexpr int kModelPoints = 5;  // solvePnPRansac: SOLVEPNP_ITERATIVE -> EPnP kernel on 5 points
constexpr int kPartStride = kRefLanes + 1;  // per-lane partial sums in LDS, padded against bank conflicts

PW_FN double hypot2(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

// Givens parameters of one Hestenes step for two rows with squared norms a, b and inner product p.
PW_FN bool jacobi_cs(double a, double b, double p, double* c, double* s) {
    const double eps = DBL_EPSILON * 10;
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    const double beta = a - b, gamma = hypot2(p, beta);
    if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        *s = sqrt(delta / gamma);
        *c = p / (gamma * *s * 2);
    } else {
        *c = sqrt((gamma + beta) / (gamma * 2));
        *s = p / (gamma * *c * 2);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Wave-uniform one-sided Jacobi SVD of a small matrix held in registers.  At: N rows of length M = the columns
// of the matrix.  On exit row i = sigma_i u_i, Vt row i = v_i, W descending (stable for equal values).
// `ls` (optional): N * (M + N) doubles of LDS private to the calling wave -- the final ordering then goes through it
// (a permuted store and an ordered load) instead of through a second register copy of both matrices.
template <int N, int M>
PW_FN void svd_small(double (&At)[N][M], double (&Vt)[N][N], double (&W)[N], double* ls = nullptr) {
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        PW_UNROLL
        for (int k = 0; k < N; k++) Vt[i][k] = i == k ? 1.0 : 0.0;
    }
    const int max_iter = M > 30 ? M : 30;
    for (int it = 0; it < max_iter; it++) {
        bool changed = false;
        PW_UNROLL
        for (int i = 0; i < N - 1; i++) {
            PW_UNROLL
            for (int j = i + 1; j < N; j++) {
                double a = 0, b = 0, p = 0, c, s;
                PW_UNROLL
                for (int k = 0; k < M; k++) a += At[i][k] * At[i][k];
                PW_UNROLL
                for (int k = 0; k < M; k++) b += At[j][k] * At[j][k];
                PW_UNROLL
                for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
                if (jacobi_cs(a, b, p, &c, &s)) {
                    PW_UNROLL
                    for (int k = 0; k < M; k++) {
                        const double t0 = c * At[i][k] + s * At[j][k];
                        const double t1 = c * At[j][k] - s * At[i][k];
                        At[i][k] = t0;
                        At[j][k] = t1;
                    }
                    PW_UNROLL
                    for (int k = 0; k < N; k++) {
                        const double t0 = c * Vt[i][k] + s * Vt[j][k];
                        const double t1 = c * Vt[j][k] - s * Vt[i][k];
                        Vt[i][k] = t0;
                        Vt[j][k] = t1;
                    }
                    changed = true;
                }
            }
        }
        if (!changed) break;
    }
    if (ls) {
        double W0[N];
        int rank[N];
        PW_UNROLL
        for (int i = 0; i < N; i++) {
            double sd = 0;
            PW_UNROLL
            for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
            W0[i] = sqrt(sd);
            W[i] = W0[i];
        }
        PW_UNROLL
        for (int i = 0; i < N; i++) {
            int r = 0;
            PW_UNROLL
            for (int j = 0; j < N; j++) r += (W0[j] > W0[i]) || (W0[j] == W0[i] && j < i);
            rank[i] = r;
        }
        PW_UNROLL
        for (int i = 0; i < N; i++) {
            double* row = ls + rank[i] * (M + N);
            PW_UNROLL
            for (int k = 0; k < M; k++) row[k] = At[i][k];
            PW_UNROLL
            for (int k = 0; k < N; k++) row[M + k] = Vt[i][k];
        }
        PW_UNROLL
        for (int p = 0; p < N; p++) {
            PW_UNROLL
            for (int i = 0; i < N; i++) W[p] = rank[i] == p ? W0[i] : W[p];
            const double* row = ls + p * (M + N);
            PW_UNROLL
            for (int k = 0; k < M; k++) At[p][k] = row[k];
            PW_UNROLL
            for (int k = 0; k < N; k++) Vt[p][k] = row[M + k];
        }
        return;
    }
    double W0[N], A0[N][M], V0[N][N];
    int rank[N];
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        double sd = 0;
        PW_UNROLL
        for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
        W0[i] = sqrt(sd);
        W[i] = W0[i];
        PW_UNROLL
        for (int k = 0; k < M; k++) A0[i][k] = At[i][k];
        PW_UNROLL
        for (int k = 0; k < N; k++) V0[i][k] = Vt[i][k];
    }
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        int r = 0;
        PW_UNROLL
        for (int j = 0; j < N; j++) r += (W0[j] > W0[i]) || (W0[j] == W0[i] && j < i);
        rank[i] = r;
    }
    PW_UNROLL
    for (int p = 0; p < N; p++) {
        PW_UNROLL
        for (int i = 0; i < N; i++) {
            const bool hit = rank[i] == p;
            W[p] = hit ? W0[i] : W[p];
            PW_UNROLL
            for (int k = 0; k < M; k++) At[p][k] = hit ? A0[i][k] : At[p][k];
            PW_UNROLL
            for (int k = 0; k < N; k++) Vt[p][k] = hit ? V0[i][k] : Vt[p][k];
        }
    }
}

// cv::solve / cvInvert with DECOMP_SVD: X (N x NB) = pinv(A (M x N)) B (M x NB), wave-uniform.
template <int M, int N, int NB>
PW_FN void svd_solve_small(const double (&A)[M][N], const double (&B)[M][NB], double (&X)[N][NB], double* ls = nullptr) {
    double At[N][M], Vt[N][N], W[N];
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        PW_UNROLL
        for (int k = 0; k < M; k++) At[i][k] = A[k][i];
    }
    svd_small<N, M>(At, Vt, W, ls);
    double thr = 0;
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        const double s = W[i] > DBL_MIN ? 1 / W[i] : 0;
        PW_UNROLL
        for (int k = 0; k < M; k++) At[i][k] *= s;
        thr += W[i];
    }
    thr *= DBL_EPSILON * 2;
    PW_UNROLL
    for (int k = 0; k < N; k++) {
        PW_UNROLL
        for (int c = 0; c < NB; c++) X[k][c] = 0;
    }
    PW_UNROLL
    for (int i = 0; i < N; i++) {
        if (fabs(W[i]) <= thr) continue;
        const double wi = 1 / W[i];
        PW_UNROLL
        for (int c = 0; c < NB; c++) {
            double s = 0;
            PW_UNROLL
            for (int j = 0; j < M; j++) s += At[i][j] * B[j][c];
            s *= wi;
            PW_UNROLL
            for (int k = 0; k < N; k++) X[k][c] += s * Vt[i][k];
        }
    }
}

// 3 x 3: A = U diag(W) V^T, columns of U / V are the singular vectors; a zero third singular value gets
// u2 = u0 x u1.
PW_FN void svd3(const double (&A)[3][3], double (&U)[3][3], double (&W)[3], double (&V)[3][3]) {
    double At[3][3], Vt[3][3];
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        PW_UNROLL
        for (int k = 0; k < 3; k++) At[i][k] = A[k][i];
    }
    svd_small<3, 3>(At, Vt, W);
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        const double s = W[i] > DBL_MIN ? 1 / W[i] : 0;
        PW_UNROLL
        for (int k = 0; k < 3; k++) At[i][k] *= s;
    }
    if (!(W[2] > DBL_MIN)) {
        At[2][0] = At[0][1] * At[1][2] - At[0][2] * At[1][1];
        At[2][1] = At[0][2] * At[1][0] - At[0][0] * At[1][2];
        At[2][2] = At[0][0] * At[1][1] - At[0][1] * At[1][0];
    }
    PW_UNROLL
    for (int r = 0; r < 3; r++) {
        PW_UNROLL
        for (int k = 0; k < 3; k++) {
            U[r][k] = At[k][r];
            V[r][k] = Vt[k][r];
        }
    }
}

// R = U V^T from the SVD of A (cv::SVD + GEMM as the callers use it).
PW_FN void nearest_rotation_uvt(const double (&A)[3][3], double (&R)[3][3]) {
    double U[3][3], W[3], V[3][3];
    svd3(A, U, W, V);
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        PW_UNROLL
        for (int j = 0; j < 3; j++) R[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Lane-parallel one-sided Jacobi on N rows of length M in LDS (N = 6, 10 or 12), round-robin pair schedule: the N/2
// pairs of a round touch disjoint rows, lane k < N/2 derives the rotation of pair k, then all lanes apply the
// rotations to the rows of At and Vt.  perm[p] = row holding the p-th largest singular value.
PW_FN void rr_pair(int n, int r, int k, int* i, int* j) {
    int a, b;
    if (k == 0) {
        a = n - 1;
        b = r;
    } else {
        a = (r + k) % (n - 1);
        b = (r - k + (n - 1)) % (n - 1);
    }
    *i = a < b ? a : b;
    *j = a < b ? b : a;
}

struct JacobiLds {
    double dots[18];  // per pair: |Ai|^2, |Aj|^2, Ai.Aj
    double cs[12];    // (c, s) per pair
    double W[12];
    int rot[6];
    int perm[12];
    int changed;
};

template <int N, int M, int NL>
PW_FN void jacobi_rr(double* At, double* Vt, JacobiLds& js) {  // At: N rows of length M; Vt: N x N or nullptr
    const int NV = Vt ? N : 0;
    if (Vt) {
        PW_LANES(l, NL) {
            for (int e = l; e < N * N; e += NL) Vt[e] = (e / N == e % N) ? 1.0 : 0.0;
        }
    }
    PW_SYNC();
    const int max_iter = M > 30 ? M : 30;
    for (int it = 0; it < max_iter; it++) {
        PW_LANES(l, NL) {
            if (l == 0) js.changed = 0;
        }
        PW_SYNC();
        for (int r = 0; r < N - 1; r++) {
            // three inner products per pair (|Ai|^2, |Aj|^2, Ai.Aj): one lane each when the workgroup has several
            // waves to hide the extra hand-over, one lane per pair (the three chains interleave) on a single wave
            if (NL > kWave) {
                PW_LANES(l, NL) {
                    if (l < 3 * (N / 2)) {
                        int i, j;
                        rr_pair(N, r, l / 3, &i, &j);
                        const double* x = At + M * (l % 3 == 1 ? j : i);
                        const double* y = At + M * (l % 3 == 0 ? i : j);
                        double d = 0;
                        for (int k = 0; k < M; k++) d += x[k] * y[k];
                        js.dots[l] = d;
                    }
                }
                PW_SYNC();
            }
            PW_LANES(l, NL) {
                if (l < N / 2) {
                    double a = 0, b = 0, p = 0, c = 1, s = 0;
                    if (NL > kWave) {
                        a = js.dots[3 * l];
                        b = js.dots[3 * l + 1];
                        p = js.dots[3 * l + 2];
                    } else {
                        int i, j;
                        rr_pair(N, r, l, &i, &j);
                        for (int k = 0; k < M; k++) a += At[i * M + k] * At[i * M + k];
                        for (int k = 0; k < M; k++) b += At[j * M + k] * At[j * M + k];
                        for (int k = 0; k < M; k++) p += At[i * M + k] * At[j * M + k];
                    }
                    const bool rot = jacobi_cs(a, b, p, &c, &s);
                    js.cs[2 * l] = c;
                    js.cs[2 * l + 1] = s;
                    js.rot[l] = rot;
                    if (rot) js.changed = 1;
                }
            }
            PW_SYNC();
            PW_LANES(l, NL) {
                for (int e = l; e < (N / 2) * (M + NV); e += NL) {  // (pair, column of At | column of Vt)
                    const int k = e / (M + NV), rem = e % (M + NV);
                    if (js.rot[k]) {
                        int i, j;
                        rr_pair(N, r, k, &i, &j);
                        const double c = js.cs[2 * k], s = js.cs[2 * k + 1];
                        double* pi = rem < M ? At + i * M + rem : Vt + i * N + (rem - M);
                        double* pj = rem < M ? At + j * M + rem : Vt + j * N + (rem - M);
                        const double x = *pi, y = *pj;
                        *pi = c * x + s * y;
                        *pj = c * y - s * x;
                    }
                }
            }
            PW_SYNC();
        }
        const int changed = js.changed;
        PW_SYNC();
        if (!changed) break;
    }
    PW_LANES(l, NL) {
        if (l < N) {
            double sd = 0;
            for (int k = 0; k < M; k++) sd += At[l * M + k] * At[l * M + k];
            js.W[l] = sqrt(sd);
        }
    }
    PW_SYNC();
    PW_LANES(l, NL) {
        if (l < N) {
            int src = l;
            for (int i = 0; i < N; i++) {
                int rank = 0;
                for (int j = 0; j < N; j++) rank += (js.W[j] > js.W[i]) || (js.W[j] == js.W[i] && j < i);
                if (rank == l) src = i;
            }
            js.perm[l] = src;
        }
    }
    PW_SYNC();
}

// ---------------------------------------------------------------------------------------------------------
// EPnP on 5 correspondences (OpenCV calib3d epnp.cpp, restated; DESIGN.md section 9 states the canonical
// arithmetic: every SVD is the one-sided Jacobi above, every sum runs in index order).
struct Camera {
    double fu, fv, uc, vc;
};

struct EpnpPoints {
    double pws[kModelPoints][3];
    double us[kModelPoints][2];
    double alphas[kModelPoints][4];
    double cws[4][3];
};

struct HypLds {
    // operands every lane reads (uniform addresses = LDS broadcasts) instead of carrying private copies: the 6 x 10 system,
    // the four null-space vectors and the 5 points with their control points -- ~330 registers per lane otherwise
    EpnpPoints e;
    double l6[6][10];
    double v4[4][12];
    double rho[6];
    double sort_ls[3][5 * 11];  // per beta variant (= per wave): staging of svd_small's final ordering
    double At[144];  // M^T M, then the rotated rows
    double Vt[144];  // accumulated rotations = eigenvectors of M^T M
    double M[2 * kModelPoints * 12];
    double alphas[kModelPoints * 4];
    double us[kModelPoints * 2];
    double var_err[3];  // per beta variant: mean reprojection error, R, t
    double var_R[3][9];
    double var_t[3][3];
    JacobiLds js;
    int cnt[kHypLanes];
};

PW_FN double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PW_FN double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// epnp::qr_solve on the 6 x 4 Gauss-Newton system (Householder), wave-uniform.
PW_FN void qr_solve_6x4(double (&A)[6][4], double (&b)[6], double (&X)[4]) {
    double A1[4], A2[4];
    bool singular = false;
    PW_UNROLL
    for (int k = 0; k < 4; k++) {
        if (singular) continue;
        double eta = fabs(A[k][k]);
        PW_UNROLL
        for (int i = k + 1; i < 6; i++) {
            const double e = fabs(A[i][k]);
            if (eta < e) eta = e;
        }
        if (eta == 0) {
            singular = true;
            continue;
        }
        double sum2 = 0;
        const double inv_eta = 1. / eta;
        PW_UNROLL
        for (int i = k; i < 6; i++) {
            A[i][k] *= inv_eta;
            sum2 += A[i][k] * A[i][k];
        }
        double sigma = sqrt(sum2);
        if (A[k][k] < 0) sigma = -sigma;
        A[k][k] += sigma;
        A1[k] = sigma * A[k][k];
        A2[k] = -eta * sigma;
        PW_UNROLL
        for (int j = k + 1; j < 4; j++) {
            double sum = 0;
            PW_UNROLL
            for (int i = k; i < 6; i++) sum += A[i][k] * A[i][j];
            const double tau = sum / A1[k];
            PW_UNROLL
            for (int i = k; i < 6; i++) A[i][j] -= tau * A[i][k];
        }
    }
    if (singular) {
        PW_UNROLL
        for (int i = 0; i < 4; i++) X[i] = 0;
        return;
    }
    PW_UNROLL
    for (int j = 0; j < 4; j++) {
        double tau = 0;
        PW_UNROLL
        for (int i = j; i < 6; i++) tau += A[i][j] * b[i];
        tau /= A1[j];
        PW_UNROLL
        for (int i = j; i < 6; i++) b[i] -= tau * A[i][j];
    }
    X[3] = b[3] / A2[3];
    PW_UNROLL
    for (int i = 2; i >= 0; i--) {
        double sum = 0;
        PW_UNROLL
        for (int j = i + 1; j < 4; j++) sum += A[i][j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

template <int NC>
PW_FN void solve_betas_system(const double (&l)[6][10], const double (&rho)[6], const int (&cols)[NC], double (&b)[NC], double* ls) {
    double L[6][NC], rhs[6][1], x[NC][1];
    PW_UNROLL
    for (int i = 0; i < 6; i++) {
        PW_UNROLL
        for (int k = 0; k < NC; k++) L[i][k] = l[i][cols[k]];
        rhs[i][0] = rho[i];
    }
    svd_solve_small<6, NC, 1>(L, rhs, x, ls);
    PW_UNROLL
    for (int k = 0; k < NC; k++) b[k] = x[k][0];
}

PW_FN void find_betas(const double (&l)[6][10], const double (&rho)[6], int variant, double (&betas)[4], double* ls) {
    if (variant == 1) {  // [B11 B12 B13 B14]
        const int cols[4] = {0, 1, 3, 6};
        double b[4];
        solve_betas_system<4>(l, rho, cols, b, ls);
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
    double b0, b1, b2, b3 = 0;
    if (variant == 2) {  // [B11 B12 B22]
        const int cols[3] = {0, 1, 2};
        double b[3];
        solve_betas_system<3>(l, rho, cols, b, ls);
        b0 = b[0];
        b1 = b[1];
        b2 = b[2];
    } else {  // [B11 B12 B22 B13 B23]
        const int cols[5] = {0, 1, 2, 3, 4};
        double b[5];
        solve_betas_system<5>(l, rho, cols, b, ls);
        b0 = b[0];
        b1 = b[1];
        b2 = b[2];
        b3 = b[3];
    }
    if (b0 < 0) {
        betas[0] = sqrt(-b0);
        betas[1] = (b2 < 0) ? sqrt(-b2) : 0.0;
    } else {
        betas[0] = sqrt(b0);
        betas[1] = (b2 > 0) ? sqrt(b2) : 0.0;
    }
    if (b1 < 0) betas[0] = -betas[0];
    betas[2] = variant == 2 ? 0.0 : b3 / betas[0];
    betas[3] = 0.0;
}

PW_FN void gauss_newton(const double (&l)[6][10], const double (&rho)[6], double (&betas)[4]) {
    for (int it = 0; it < 5; it++) {
        double A[6][4], b[6], x[4];
        PW_UNROLL
        for (int i = 0; i < 6; i++) {
            const double* r = l[i];
            A[i][0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            A[i][1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            A[i][2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            A[i][3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] +
                             r[3] * betas[0] * betas[2] + r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                             r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] + r[8] * betas[2] * betas[3] +
                             r[9] * betas[3] * betas[3]);
        }
        qr_solve_6x4(A, b, x);
        PW_UNROLL
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// epnp::compute_R_and_t: control points in the camera frame from the betas, Arun alignment, mean reprojection
// error.  v[i] = eigenvector of the (i+1)-th smallest eigenvalue.
PW_FN double compute_R_and_t(const EpnpPoints& e, const Camera& cam, const double (&v)[4][12], const double (&betas)[4],
                             double (&R)[3][3], double (&t)[3]) {
    constexpr int n = kModelPoints;
    double ccs[4][3], pcs[n][3];
    PW_UNROLL
    for (int j = 0; j < 4; j++) ccs[j][0] = ccs[j][1] = ccs[j][2] = 0.0;
    PW_UNROLL
    for (int i = 0; i < 4; i++) {
        PW_UNROLL
        for (int j = 0; j < 4; j++) {
            PW_UNROLL
            for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[i][3 * j + k];
        }
    }
    PW_UNROLL
    for (int i = 0; i < n; i++) {
        const double* a = e.alphas[i];
        PW_UNROLL
        for (int j = 0; j < 3; j++) pcs[i][j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[0][2] < 0.0) {
        PW_UNROLL
        for (int i = 0; i < n; i++) {
            PW_UNROLL
            for (int j = 0; j < 3; j++) pcs[i][j] = -pcs[i][j];
        }
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    PW_UNROLL
    for (int i = 0; i < n; i++) {
        PW_UNROLL
        for (int j = 0; j < 3; j++) {
            pc0[j] += pcs[i][j];
            pw0[j] += e.pws[i][j];
        }
    }
    PW_UNROLL
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    PW_UNROLL
    for (int i = 0; i < n; i++) {
        PW_UNROLL
        for (int j = 0; j < 3; j++) {
            abt[j][0] += (pcs[i][j] - pc0[j]) * (e.pws[i][0] - pw0[0]);
            abt[j][1] += (pcs[i][j] - pc0[j]) * (e.pws[i][1] - pw0[1]);
            abt[j][2] += (pcs[i][j] - pc0[j]) * (e.pws[i][2] - pw0[2]);
        }
    }
    double U[3][3], W[3], V[3][3];
    svd3(abt, U, W, V);
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        PW_UNROLL
        for (int j = 0; j < 3; j++) R[i][j] = dot3(U[i], V[j]);
    }
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
        R[2][0] = -R[2][0];
        R[2][1] = -R[2][1];
        R[2][2] = -R[2][2];
    }
    PW_UNROLL
    for (int i = 0; i < 3; i++) t[i] = pc0[i] - dot3(R[i], pw0);
    double sum2 = 0.0;
    PW_UNROLL
    for (int i = 0; i < n; i++) {
        const double* pw = e.pws[i];
        const double Xc = dot3(R[0], pw) + t[0];
        const double Yc = dot3(R[1], pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        const double ue = cam.uc + cam.fu * Xc * inv_Zc;
        const double ve = cam.vc + cam.fv * Yc * inv_Zc;
        const double u = e.us[i][0], vv = e.us[i][1];
        sum2 += sqrt((u - ue) * (u - ue) + (vv - ve) * (vv - ve));
    }
    return sum2 / n;
}

// One RANSAC hypothesis: solvePnP(SOLVEPNP_EPNP) on the 5 pairs idx[0..5) -> (R row-major, t).
PW_FN void epnp_hypothesis(HypLds& s, const float* p3, const float* p2, const int32_t* idx, const Camera& cam,
                           double (&Rout)[3][3], double (&tout)[3]) {
    constexpr int n = kModelPoints;
    EpnpPoints e;
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    PW_UNROLL
    for (int i = 0; i < n; i++) {
        const int q = idx[i];
        PW_UNROLL
        for (int j = 0; j < 3; j++) e.pws[i][j] = p3[3 * q + j];
        // cv::undistortPoints without distortion keeps the input depth: normalised coordinates in float
        const float xn = (float)(((double)p2[2 * q] - cam.uc) * ifx);
        const float yn = (float)(((double)p2[2 * q + 1] - cam.vc) * ify);
        e.us[i][0] = xn * cam.fu + cam.uc;
        e.us[i][1] = yn * cam.fv + cam.vc;
    }
    // choose_control_points: centroid + PCA axes
    PW_UNROLL
    for (int j = 0; j < 3; j++) {
        double c = 0;
        PW_UNROLL
        for (int i = 0; i < n; i++) c += e.pws[i][j];
        e.cws[0][j] = c / n;
    }
    {
        double At[3][3], Vt[3][3], dc[3];
        PW_UNROLL
        for (int a = 0; a < 3; a++) {
            PW_UNROLL
            for (int b = 0; b < 3; b++) {
                double sum = 0;
                PW_UNROLL
                for (int i = 0; i < n; i++) sum += (e.pws[i][a] - e.cws[0][a]) * (e.pws[i][b] - e.cws[0][b]);
                At[b][a] = sum;
            }
        }
        svd_small<3, 3>(At, Vt, dc);
        PW_UNROLL
        for (int i = 1; i < 4; i++) {
            const double k = sqrt(dc[i - 1] / n);
            PW_UNROLL
            for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * Vt[i - 1][j];
        }
    }
    // compute_barycentric_coordinates
    {
        double cc[3][3], ci[3][3];
        const double eye[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        PW_UNROLL
        for (int i = 0; i < 3; i++) {
            PW_UNROLL
            for (int j = 1; j < 4; j++) cc[i][j - 1] = e.cws[j][i] - e.cws[0][i];
        }
        svd_solve_small<3, 3, 3>(cc, eye, ci);
        PW_UNROLL
        for (int i = 0; i < n; i++) {
            const double* pi = e.pws[i];
            double* a = e.alphas[i];
            PW_UNROLL
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[j][0] * (pi[0] - e.cws[0][0]) + ci[j][1] * (pi[1] - e.cws[0][1]) +
                           ci[j][2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    // M (2n x 12) and M^T M through LDS, one entry per lane and pass; the points themselves stay in LDS from here on
    PW_LANES(l, kHypLanes) {
        if (l == 0) {
            s.e = e;
            PW_UNROLL
            for (int i = 0; i < n; i++) {
                PW_UNROLL
                for (int k = 0; k < 4; k++) s.alphas[4 * i + k] = e.alphas[i][k];
                s.us[2 * i] = e.us[i][0];
                s.us[2 * i + 1] = e.us[i][1];
            }
        }
    }
    PW_SYNC();
    PW_LANES(l, kHypLanes) {
        for (int el = l; el < 2 * n * 12; el += kHypLanes) {
            const int r = el / 12, c = el % 12, i = r / 2, k = c / 3, comp = c % 3;
            const double a = s.alphas[4 * i + k];
            double val;
            if ((r & 1) == 0)
                val = comp == 0 ? a * cam.fu : comp == 1 ? 0.0 : a * (cam.uc - s.us[2 * i]);
            else
                val = comp == 0 ? 0.0 : comp == 1 ? a * cam.fv : a * (cam.vc - s.us[2 * i + 1]);
            s.M[el] = val;
        }
    }
    PW_SYNC();
    PW_LANES(l, kHypLanes) {
        for (int el = l; el < 144; el += kHypLanes) {
            const int a = el / 12, b = el % 12;
            double sum = 0;
            for (int r = 0; r < 2 * n; r++) sum += s.M[r * 12 + a] * s.M[r * 12 + b];
            s.At[el] = sum;
        }
    }
    PW_SYNC();
    jacobi_rr<12, 12, kHypLanes>(s.At, s.Vt, s.js);
    // compute_L_6x10 from the four eigenvectors of the smallest eigenvalues
    PW_LANES(l, kHypLanes) {
        if (l < 60) {
            const int i = l / 10, c = l % 10;
            const int xs[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, ys[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            const double* vx = s.Vt + 12 * s.js.perm[11 - xs[c]];
            const double* vy = s.Vt + 12 * s.js.perm[11 - ys[c]];
            double dx[3], dy[3];
            for (int k = 0; k < 3; k++) {
                dx[k] = vx[3 * pa[i] + k] - vx[3 * pb[i] + k];
                dy[k] = vy[3 * pa[i] + k] - vy[3 * pb[i] + k];
            }
            const double d = dot3(dx, dy);
            s.l6[i][c] = xs[c] == ys[c] ? d : 2.0f * d;
        } else if (l >= 64 && l < 64 + 48) {  // the four eigenvectors of the smallest eigenvalues
            const int i = (l - 64) / 12, k = (l - 64) % 12;
            s.v4[i][k] = s.Vt[12 * s.js.perm[11 - i] + k];
        } else if (l >= 128 && l < 128 + 6) {
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            s.rho[l - 128] = dist2(s.e.cws[pa[l - 128]], s.e.cws[pb[l - 128]]);
        }
    }
    PW_SYNC();
    const double(&l6)[6][10] = s.l6;
    const double(&v)[4][12] = s.v4;
    const double(&rho)[6] = s.rho;
    // the three beta initialisations (epnp::find_betas_approx_1/2/3 + Gauss-Newton + compute_R_and_t) are
    // independent: wave w takes variant w + 1
    PW_WAVES(w, kHypLanes / kWave) {
        for (int variant = 1 + w; variant <= 3; variant += kHypLanes / kWave) {
            double betas[4], R[3][3], t[3];
            find_betas(l6, rho, variant, betas, s.sort_ls[variant - 1]);
            gauss_newton(l6, rho, betas);
            s.var_err[variant - 1] = compute_R_and_t(s.e, cam, v, betas, R, t);
            PW_UNROLL
            for (int i = 0; i < 3; i++) {
                PW_UNROLL
                for (int j = 0; j < 3; j++) s.var_R[variant - 1][3 * i + j] = R[i][j];
                s.var_t[variant - 1][i] = t[i];
            }
        }
    }
    PW_SYNC();
    // N = 1; if (rep_errors[2] < rep_errors[1]) N = 2; if (rep_errors[3] < rep_errors[N]) N = 3;
    int N = 0;
    if (s.var_err[1] < s.var_err[0]) N = 1;
    if (s.var_err[2] < s.var_err[N]) N = 2;
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        PW_UNROLL
        for (int j = 0; j < 3; j++) Rout[i][j] = s.var_R[N][3 * i + j];
        tout[i] = s.var_t[N][i];
    }
    PW_SYNC();
}

// PnPRansacCallback::computeError + findInliers for a model carried as (R, t): mask[i] = 1 iff the squared
// reprojection error (float, as cv::projectPoints / norm(NORM_L2SQR) produce it) is <= thr2.  Returns the count.
PW_FN int score_model(HypLds& s, const float* p3, const float* p2, int n, const Camera& cam, const double (&R)[3][3],
                      const double (&t)[3], float thr2, uint8_t* mask) {
    PW_LANES(l, kHypLanes) {
        int good = 0;
        for (int i = l; i < n; i += kHypLanes) {
            const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
            double x = R[0][0] * X + R[0][1] * Y + R[0][2] * Z + t[0];
            double y = R[1][0] * X + R[1][1] * Y + R[1][2] * Z + t[1];
            double z = R[2][0] * X + R[2][1] * Y + R[2][2] * Z + t[2];
            z = z ? 1. / z : 1;
            x *= z;
            y *= z;
            const float pu = (float)(x * cam.fu + cam.uc);
            const float pv = (float)(y * cam.fv + cam.vc);
            const float du = p2[2 * i] - pu, dv = p2[2 * i + 1] - pv;
            const float a = du * du, b = dv * dv;
            const float err = a + b;
            const int f = err <= thr2;
            mask[i] = (uint8_t)f;
            good += f;
        }
        s.cnt[l] = good;
    }
    PW_SYNC();
    int total = 0;
    for (int q = 0; q < kHypLanes; q++) total += s.cnt[q];
    PW_SYNC();
    return total;
}

// ---------------------------------------------------------------------------------------------------------
// geometry::helperTriangulatePoints for ONE match (motion_estimation.cpp:214-247): pixel2CamNormPlane on both
// pixels, cv::triangulatePoints with P1 = [I | 0], P2 = [R | t] (the 4x4 DLT system, right singular vector of the
// smallest singular value, stored as float like OpenCV does for Point2f input), division by w in float, then
// basics::transCoord.  Per-lane code: every lane works on its own match.
PW_FN void triangulate_match(const float* kp1, const float* kp2, const Camera& cam, const double (&R)[9],
                             const double (&t)[3], float (&p_prev)[3], float (&p_curr)[3]) {
    const float n1[2] = {(float)((kp1[0] - cam.uc) / cam.fu), (float)((kp1[1] - cam.vc) / cam.fv)};
    const float n2[2] = {(float)((kp2[0] - cam.uc) / cam.fu), (float)((kp2[1] - cam.vc) / cam.fv)};
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double P2[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
    double At[4][4], Vt[4][4], W[4];
    {
        const double x = n1[0], y = n1[1];
        PW_UNROLL
        for (int k = 0; k < 4; k++) {
            At[k][0] = x * P1[8 + k] - P1[k];
            At[k][1] = y * P1[8 + k] - P1[4 + k];
        }
    }
    {
        const double x = n2[0], y = n2[1];
        PW_UNROLL
        for (int k = 0; k < 4; k++) {
            At[k][2] = x * P2[8 + k] - P2[k];
            At[k][3] = y * P2[8 + k] - P2[4 + k];
        }
    }
    svd_small<4, 4>(At, Vt, W);
    const float X[4] = {(float)Vt[3][0], (float)Vt[3][1], (float)Vt[3][2], (float)Vt[3][3]};
    PW_UNROLL
    for (int r = 0; r < 3; r++) p_prev[r] = X[r] / X[3];
    PW_UNROLL
    for (int r = 0; r < 3; r++) {
        double s = R[3 * r] * (double)p_prev[0] + R[3 * r + 1] * (double)p_prev[1];
        s = s + R[3 * r + 2] * (double)p_prev[2];
        p_curr[r] = (float)(s + t[r]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The sequential bookkeeping of RANSACPointSetRegistrator::run over the inlier counts of the hypotheses, in
// iteration order: a hypothesis becomes the best model when its count exceeds max(best so far, modelPoints - 1),
// and every new best shortens the loop through cv::RANSACUpdateNumIters.  Uniform.  pow / log differ in the last
// ulp between math libraries; the host repeats this loop with its own libm on the same counts and re-runs the
// refinement in the (never observed) case that it would have chosen differently.
PW_FN int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

PW_FN void ransac_replay(const int32_t* counts, int n_hyp, int n, double confidence, int* best, int* iters_run) {
    int niters = n_hyp, max_good = 0, b = -1, it = 0;
    for (; it < niters; it++) {
        const int good = counts[it];
        if (good > (max_good > kModelPoints - 1 ? max_good : kModelPoints - 1)) {
            max_good = good;
            b = it;
            niters = ransac_update_num_iters(confidence, (double)(n - good) / n, kModelPoints, niters);
        }
    }
    *best = b;
    *iters_run = it;
}

// ---------------------------------------------------------------------------------------------------------
// Refinement: cvFindExtrinsicCameraParams2(useExtrinsicGuess = 0) on the inliers = DLT start + CvLevMarq.
struct RefLds {
    double part[78 * kPartStride];  // per-lane partial sums
    double red[80];                 // reduced sums
    double At[144];
    double Vt[144];
    JacobiLds js;
    int cnt[kRefLanes];
    int offs[kRefLanes + 1];
};

PW_FN void rodrigues_fwd(const double (&r)[3], double (&R)[9], double (&J)[27], bool want_j) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < DBL_EPSILON) {
        PW_UNROLL
        for (int k = 0; k < 9; k++) R[k] = I[k];
        if (want_j) {
            PW_UNROLL
            for (int k = 0; k < 27; k++) J[k] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    const double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    PW_UNROLL
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (want_j) {
        const double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,  0,  rx, 0,  rx, ry + ry,
                                 rz,      0,  rz, 0,  0,       0,  rx, 0,  0,  ry, rx, ry, rz + rz};
        const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        const double rr[3] = {rx, ry, rz};
        PW_UNROLL
        for (int i = 0; i < 3; i++) {
            const double ri = rr[i];
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri,
                         a4 = s * itheta;
            PW_UNROLL
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x[i * 9 + k];
        }
    }
}

PW_FN void rodrigues_inv(const double (&Rin)[3][3], double (&r)[3]) {
    double R[3][3];
    nearest_rotation_uvt(Rin, R);
    double x = R[2][1] - R[1][2], y = R[0][2] - R[2][0], z = R[1][0] - R[0][1];
    const double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0][0] + R[1][1] + R[2][2] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            x = y = z = 0;
        } else {
            double t = (R[0][0] + 1) * 0.5;
            x = sqrt(t > 0. ? t : 0.);
            t = (R[1][1] + 1) * 0.5;
            y = sqrt(t > 0. ? t : 0.) * (R[0][1] < 0 ? -1. : 1.);
            t = (R[2][2] + 1) * 0.5;
            z = sqrt(t > 0. ? t : 0.) * (R[0][2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[1][2] > 0) != (y * z > 0)) z = -z;
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

// Sum over lanes of NV per-lane values already stored at part[v * kPartStride + lane] -> red[v].
PW_FN void reduce_parts(RefLds& s, int nv) {
    PW_SYNC();
    PW_LANES(l, kRefLanes) {
        for (int v = l; v < nv; v += kRefLanes) {
            double sum = 0;
            for (int q = 0; q < kRefLanes; q++) sum += s.part[v * kPartStride + q];
            s.red[v] = sum;
        }
    }
    PW_SYNC();
}

// cvProjectPoints2 residuals (projection - measurement) of the compacted inliers at param = (rvec, tvec).
// with_j: also J^T J (21 upper entries -> red[0..21)), J^T err (red[21..27)); always |err|^2 -> red[27].
PW_FN void lm_evaluate(RefLds& s, const double* Mg, const double* mg, int cnt, const Camera& cam,
                       const double (&param)[6], bool with_j) {
    double R[9], dRdr[27];
    const double rv[3] = {param[0], param[1], param[2]};
    rodrigues_fwd(rv, R, dRdr, with_j);
    PW_LANES(l, kRefLanes) {
        double acc[28];
        for (int k = 0; k < 28; k++) acc[k] = 0;
        for (int i = l; i < cnt; i += kRefLanes) {
            const double X = Mg[3 * i], Y = Mg[3 * i + 1], Z = Mg[3 * i + 2];
            double x = R[0] * X + R[1] * Y + R[2] * Z + param[3];
            double y = R[3] * X + R[4] * Y + R[5] * Z + param[4];
            double z = R[6] * X + R[7] * Y + R[8] * Z + param[5];
            z = z ? 1. / z : 1;
            x *= z;
            y *= z;
            const double ex = (x * cam.fu + cam.uc) - mg[2 * i];
            const double ey = (y * cam.fv + cam.vc) - mg[2 * i + 1];
            acc[27] += ex * ex;
            acc[27] += ey * ey;
            if (with_j) {
                double Jx[6], Jy[6];
                const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                         X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
                const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                         X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
                const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                         X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
                for (int j = 0; j < 3; j++) {
                    const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                    const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                    Jx[j] = cam.fu * dxdr;
                    Jy[j] = cam.fv * dydr;
                }
                const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
                for (int j = 0; j < 3; j++) {
                    Jx[3 + j] = cam.fu * dxdt[j];
                    Jy[3 + j] = cam.fv * dydt[j];
                }
                int k = 0;
                for (int a = 0; a < 6; a++)
                    for (int b = a; b < 6; b++, k++) {
                        acc[k] += Jx[a] * Jx[b];
                        acc[k] += Jy[a] * Jy[b];
                    }
                for (int a = 0; a < 6; a++) {
                    acc[21 + a] += Jx[a] * ex;
                    acc[21 + a] += Jy[a] * ey;
                }
            }
        }
        for (int k = 0; k < 28; k++) s.part[k * kPartStride + l] = acc[k];
    }
    reduce_parts(s, 28);
}

// CvLevMarq::step: param = prev - pinv(JtJ with diag * (1 + lambda)) JtErr, the 6 x 6 SVD on the lanes.
PW_FN void lm_step(RefLds& s, const double (&JtJ)[21], const double (&JtErr)[6], int lambdaLg10,
                   const double (&prev)[6], double (&param)[6]) {
    const double lambda = exp(lambdaLg10 * log(10.));
    PW_SYNC();
    {
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++, k++) {
                const double v = a == b ? JtJ[k] * (1. + lambda) : JtJ[k];
                s.At[a * 6 + b] = v;
                s.At[b * 6 + a] = v;
            }
    }
    PW_SYNC();
    jacobi_rr<6, 6, kRefLanes>(s.At, s.Vt, s.js);
    double W[6], thr = 0, x[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < 6; p++) {
        W[p] = s.js.W[s.js.perm[p]];
        thr += W[p];
    }
    thr *= DBL_EPSILON * 2;
    for (int p = 0; p < 6; p++) {
        if (fabs(W[p]) <= thr) continue;
        const double* u = s.At + 6 * s.js.perm[p];
        const double* v = s.Vt + 6 * s.js.perm[p];
        const double nrm = W[p] > DBL_MIN ? 1 / W[p] : 0;
        const double wi = 1 / W[p];
        double sum = 0;
        for (int j = 0; j < 6; j++) sum += (u[j] * nrm) * JtErr[j];
        sum *= wi;
        for (int k = 0; k < 6; k++) x[k] += sum * v[k];
    }
    for (int k = 0; k < 6; k++) param[k] = prev[k] - x[k];
    PW_SYNC();
}

struct RefineResult {
    double param[6];  // rvec, tvec
    int n_inliers;
    int used_dlt;
    int lm_iters;
    int lm_evals;
};

// mode 0: full refinement of the pairs flagged in mask (n bytes), started from the DLT (or from (R0, t0) for
//         planar / under-determined sets).  mode 1: only convert (R0, t0) to (rvec, tvec) -- the
//         "npoints == model_points" shortcut of solvePnPRansac.
// Mg / mg: global scratch for the compacted inliers as doubles (3 and 2 per pair).
PW_FN void refine_pose(RefLds& s, const float* p3, const float* p2, const uint8_t* mask, int n, const Camera& cam,
                       const double (&R0)[3][3], const double (&t0)[3], int mode, double* Mg, double* mg,
                       RefineResult& out) {
    out.used_dlt = 0;
    out.lm_iters = 0;
    out.lm_evals = 0;
    if (mode == 1) {
        double r[3];
        rodrigues_inv(R0, r);
        for (int k = 0; k < 3; k++) {
            out.param[k] = r[k];
            out.param[3 + k] = t0[k];
        }
        out.n_inliers = n;
        return;
    }
    // ordered compaction of the inliers: lane l owns the contiguous block [l * blk, (l + 1) * blk)
    const int blk = (n + kRefLanes - 1) / kRefLanes;
    PW_LANES(l, kRefLanes) {
        int c = 0;
        const int lo = l * blk, hi = (lo + blk) < n ? (lo + blk) : n;
        for (int i = lo; i < hi; i++) c += mask[i] != 0;
        s.cnt[l] = c;
    }
    PW_SYNC();
    PW_LANES(l, kRefLanes) {
        if (l == 0) {
            int run = 0;
            for (int q = 0; q < kRefLanes; q++) {
                s.offs[q] = run;
                run += s.cnt[q];
            }
            s.offs[kRefLanes] = run;
        }
    }
    PW_SYNC();
    const int cnt = s.offs[kRefLanes];
    out.n_inliers = cnt;
    PW_LANES(l, kRefLanes) {
        int o = s.offs[l];
        const int lo = l * blk, hi = (lo + blk) < n ? (lo + blk) : n;
        for (int i = lo; i < hi; i++)
            if (mask[i]) {
                Mg[3 * o] = p3[3 * i];
                Mg[3 * o + 1] = p3[3 * i + 1];
                Mg[3 * o + 2] = p3[3 * i + 2];
                mg[2 * o] = p2[2 * i];
                mg[2 * o + 1] = p2[2 * i + 1];
                o++;
            }
    }
    PW_SYNC();
    // planarity test: singular values of the centred second-moment matrix of the object points
    PW_LANES(l, kRefLanes) {
        double a[3] = {0, 0, 0};
        for (int i = l; i < cnt; i += kRefLanes)
            for (int j = 0; j < 3; j++) a[j] += Mg[3 * i + j];
        for (int j = 0; j < 3; j++) s.part[j * kPartStride + l] = a[j];
    }
    reduce_parts(s, 3);
    double Mc[3];
    for (int j = 0; j < 3; j++) Mc[j] = s.red[j] / cnt;
    PW_LANES(l, kRefLanes) {
        double a[6] = {0, 0, 0, 0, 0, 0};
        for (int i = l; i < cnt; i += kRefLanes) {
            const double d0 = Mg[3 * i] - Mc[0], d1 = Mg[3 * i + 1] - Mc[1], d2 = Mg[3 * i + 2] - Mc[2];
            a[0] += d0 * d0;
            a[1] += d0 * d1;
            a[2] += d0 * d2;
            a[3] += d1 * d1;
            a[4] += d1 * d2;
            a[5] += d2 * d2;
        }
        for (int j = 0; j < 6; j++) s.part[j * kPartStride + l] = a[j];
    }
    reduce_parts(s, 6);
    bool planar;
    {
        double At[3][3] = {{s.red[0], s.red[1], s.red[2]}, {s.red[1], s.red[3], s.red[4]}, {s.red[2], s.red[4], s.red[5]}};
        double Vt[3][3], W[3];
        svd_small<3, 3>(At, Vt, W);
        planar = W[2] / W[1] < 1e-3;
    }
    double R[3][3], t[3];
    if (planar || cnt < 6) {
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) R[i][j] = R0[i][j];
            t[i] = t0[i];
        }
    } else {
        out.used_dlt = 1;
        const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
        // L^T L (12 x 12, 78 unique sums over the 2 cnt rows of the DLT system)
        PW_LANES(l, kRefLanes) {
            double acc[78];
            for (int k = 0; k < 78; k++) acc[k] = 0;
            for (int i = l; i < cnt; i += kRefLanes) {
                const double x = -((mg[2 * i] - cam.uc) * ifx), y = -((mg[2 * i + 1] - cam.vc) * ify);
                const double X = Mg[3 * i], Y = Mg[3 * i + 1], Z = Mg[3 * i + 2];
                const double l0[12] = {X, Y, Z, 1., 0., 0., 0., 0., x * X, x * Y, x * Z, x};
                const double l1[12] = {0., 0., 0., 0., X, Y, Z, 1., y * X, y * Y, y * Z, y};
                int k = 0;
                for (int a = 0; a < 12; a++)
                    for (int b = a; b < 12; b++, k++) {
                        acc[k] += l0[a] * l0[b];
                        acc[k] += l1[a] * l1[b];
                    }
            }
            for (int k = 0; k < 78; k++) s.part[k * kPartStride + l] = acc[k];
        }
        reduce_parts(s, 78);
        PW_LANES(l, kRefLanes) {
            for (int el = l; el < 144; el += kRefLanes) {
                const int a = el / 12, b = el % 12, lo = a < b ? a : b, hi = a < b ? b : a;
                s.At[el] = s.red[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];
            }
        }
        PW_SYNC();
        jacobi_rr<12, 12, kRefLanes>(s.At, s.Vt, s.js);
        const double* v = s.Vt + 12 * s.js.perm[11];
        double RR[3][3], tt[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) RR[i][j] = v[4 * i + j];
            tt[i] = v[4 * i + 3];
        }
        const double det = RR[0][0] * (RR[1][1] * RR[2][2] - RR[1][2] * RR[2][1]) -
                           RR[0][1] * (RR[1][0] * RR[2][2] - RR[1][2] * RR[2][0]) +
                           RR[0][2] * (RR[1][0] * RR[2][1] - RR[1][1] * RR[2][0]);
        if (det < 0) {
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) RR[i][j] = -RR[i][j];
                tt[i] = -tt[i];
            }
        }
        double sc = 0, nr = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) sc += RR[i][j] * RR[i][j];
        sc = sqrt(sc);
        nearest_rotation_uvt(RR, R);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) nr += R[i][j] * R[i][j];
        const double f = sqrt(nr) / sc;
        for (int i = 0; i < 3; i++) t[i] = tt[i] * f;
        PW_SYNC();
    }
    double param[6], prev[6], r0[3];
    rodrigues_inv(R, r0);
    for (int k = 0; k < 3; k++) {
        param[k] = r0[k];
        param[3 + k] = t[k];
    }
    // CvLevMarq(6, 2 cnt, {COUNT + EPS, 20, FLT_EPSILON}) as cvFindExtrinsicCameraParams2 drives it
    double JtJ[21], JtErr[6], prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0, evals = 0;
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    lm_evaluate(s, Mg, mg, cnt, cam, param, true);
    evals++;
    for (;;) {
        for (int k = 0; k < 21; k++) JtJ[k] = s.red[k];
        for (int k = 0; k < 6; k++) JtErr[k] = s.red[21 + k];
        const double norm0 = sqrt(s.red[27]);
        for (int k = 0; k < 6; k++) prev[k] = param[k];
        lm_step(s, JtJ, JtErr, lambdaLg10, prev, param);
        if (iters == 0) prevErrNorm = norm0;
        bool done = false;
        for (;;) {
            lm_evaluate(s, Mg, mg, cnt, cam, param, false);
            evals++;
            errNorm = sqrt(s.red[27]);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    lm_step(s, JtJ, JtErr, lambdaLg10, prev, param);
                    continue;
                }
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            double dn = 0, pn = 0;
            for (int k = 0; k < 6; k++) {
                dn += (param[k] - prev[k]) * (param[k] - prev[k]);
                pn += prev[k] * prev[k];
            }
            if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < epsilon) done = true;
            break;
        }
        if (done) break;
        prevErrNorm = errNorm;
        lm_evaluate(s, Mg, mg, cnt, cam, param, true);
        evals++;
    }
    for (int k = 0; k < 6; k++) out.param[k] = param[k];
    out.lm_iters = iters;
    out.lm_evals = evals;
}

}  // namespace pw
#endif
