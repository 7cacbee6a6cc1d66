// oracle/linalg_oracle.h -- small dense linear algebra shared by the tracking / keyframe oracles (pnp_oracle.cpp,
// keyframe_oracle.cpp).  TEST INFRASTRUCTURE ONLY (see oracle.h).  This is OUR canonical arithmetic wherever OpenCV
// calls cv::SVD / cv::solve(DECOMP_SVD) / cv::invert: a one-sided (Hestenes) Jacobi with OpenCV's rotation formulas
// and 10*DBL_EPSILON test, round-robin pair order for 6 and 12 rows, cyclic order otherwise, right singular vectors
// from the accumulated rotations, stable descending order.
#ifndef MVO_ORACLE_LINALG_H
#define MVO_ORACLE_LINALG_H
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace orc_linalg {

// ------------------------------------------------------------------------------------------- cv::RNG
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ------------------------------------------------------------------------------------------- Jacobi SVD
inline double cv_hypot(double a, double b) {
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

// Round-robin schedule for an even number of rows n: round r in 0..n-2, slot k in 0..n/2-1 -> disjoint pairs
// (i < j).  The pairs of one round touch disjoint rows, so a round may be executed in any order (or in parallel).
inline void rr_pair(int n, int r, int k, int* i, int* j) {
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

// Rotates rows i, j of At (length m) and of Vt (length n) if they are not orthogonal yet.
inline bool jacobi_pair(double* At, int m, double* Vt, int n, int i, int j) {
    const double eps = DBL_EPSILON * 10;
    double* Ai = At + i * m;
    double* Aj = At + j * m;
    double a = 0, b = 0, p = 0;
    for (int k = 0; k < m; k++) a += Ai[k] * Ai[k];
    for (int k = 0; k < m; k++) b += Aj[k] * Aj[k];
    for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    double beta = a - b, gamma = cv_hypot(p, beta), c, s;
    if (beta < 0) {
        double delta = (gamma - beta) * 0.5;
        s = sqrt(delta / gamma);
        c = p / (gamma * s * 2);
    } else {
        c = sqrt((gamma + beta) / (gamma * 2));
        s = p / (gamma * c * 2);
    }
    for (int k = 0; k < m; k++) {
        double t0 = c * Ai[k] + s * Aj[k];
        double t1 = c * Aj[k] - s * Ai[k];
        Ai[k] = t0;
        Aj[k] = t1;
    }
    double* Vi = Vt + i * n;
    double* Vj = Vt + j * n;
    for (int k = 0; k < n; k++) {
        double t0 = c * Vi[k] + s * Vj[k];
        double t1 = c * Vj[k] - s * Vi[k];
        Vi[k] = t0;
        Vj[k] = t1;
    }
    return true;
}

// At: n rows of length m (the COLUMNS of the matrix being decomposed).  On exit row i = sigma_i * u_i, Vt row i
// = v_i, W sorted descending (rows permuted with it).
// (cv::RNG lives here too: RANSACPointSetRegistrator seeds it with (uint64)-1 for solvePnPRansac and findEssentialMat)
inline void jacobi_svd(double* At, int n, int m, double* Vt, double* W) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) Vt[i * n + k] = i == k ? 1.0 : 0.0;
    const int max_iter = m > 30 ? m : 30;
    for (int it = 0; it < max_iter; it++) {
        bool changed = false;
        if (n == 12 || n == 6) {
            for (int r = 0; r < n - 1; r++)
                for (int k = 0; k < n / 2; k++) {
                    int i, j;
                    rr_pair(n, r, k, &i, &j);
                    changed |= jacobi_pair(At, m, Vt, n, i, j);
                }
        } else {
            for (int i = 0; i < n - 1; i++)
                for (int j = i + 1; j < n; j++) changed |= jacobi_pair(At, m, Vt, n, i, j);
        }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        W[i] = sqrt(sd);
    }
    // descending, equal values keep their order (rank sort; cv::SVD's selection sort agrees whenever the singular
    // values are distinct)
    double At0[12 * 12], Vt0[12 * 12], W0[12];
    memcpy(At0, At, sizeof(double) * n * m);
    memcpy(Vt0, Vt, sizeof(double) * n * n);
    memcpy(W0, W, sizeof(double) * n);
    for (int i = 0; i < n; i++) {
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (W0[j] > W0[i]) || (W0[j] == W0[i] && j < i);
        W[rank] = W0[i];
        memcpy(At + rank * m, At0 + i * m, sizeof(double) * m);
        memcpy(Vt + rank * n, Vt0 + i * n, sizeof(double) * n);
    }
}

// rows of At -> unit left singular vectors (zero for sigma <= DBL_MIN)
inline void svd_normalize(double* At, int n, int m, const double* W) {
    for (int i = 0; i < n; i++) {
        double s = W[i] > DBL_MIN ? 1 / W[i] : 0;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

// cv::solve / cvInvert with DECOMP_SVD: X (n x nb) = pinv(A (m x n, m >= n)) * B (m x nb).
inline void svd_solve(const double* A, int m, int n, const double* B, int nb, double* X) {
    double At[12 * 12], Vt[12 * 12], W[12];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) At[i * m + k] = A[k * n + i];
    jacobi_svd(At, n, m, Vt, W);
    svd_normalize(At, n, m, W);
    double thr = 0;
    for (int i = 0; i < n; i++) thr += W[i];
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n * nb; i++) X[i] = 0;
    for (int i = 0; i < n; i++) {
        if (fabs(W[i]) <= thr) continue;
        double wi = 1 / W[i];
        for (int c = 0; c < nb; c++) {
            double s = 0;
            for (int j = 0; j < m; j++) s += At[i * m + j] * B[j * nb + c];
            s *= wi;
            for (int k = 0; k < n; k++) X[k * nb + c] += s * Vt[i * n + k];
        }
    }
}

// 3x3: A = U diag(W) V^T; U, V row-major, columns = singular vectors.  A zero third singular value gets the
// cross product of the first two left vectors.
inline void svd3(const double A[9], double U[9], double W[3], double V[9]) {
    double At[9], Vt[9];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) At[i * 3 + k] = A[k * 3 + i];
    jacobi_svd(At, 3, 3, Vt, W);
    svd_normalize(At, 3, 3, W);
    if (!(W[2] > DBL_MIN)) {
        At[6] = At[1] * At[5] - At[2] * At[4];
        At[7] = At[2] * At[3] - At[0] * At[5];
        At[8] = At[0] * At[4] - At[1] * At[3];
    }
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) {
            U[r * 3 + k] = At[k * 3 + r];
            V[r * 3 + k] = Vt[k * 3 + r];
        }
}

// cv::Mat::inv() of a 4x4 double matrix: hal::LU64f with partial pivoting on [A | I].
inline int invert4x4_lu(const double* T, double* out) {
    double A[16], B[16];
    memcpy(A, T, sizeof(A));
    for (int i = 0; i < 16; i++) B[i] = (i % 5 == 0) ? 1 : 0;
    const int m = 4;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
        if (fabs(A[k * m + i]) < DBL_EPSILON * 100) return 0;
        if (k != i)
            for (int j = 0; j < m; j++) {
                double t = A[i * m + j];
                A[i * m + j] = A[k * m + j];
                A[k * m + j] = t;
                t = B[i * m + j];
                B[i * m + j] = B[k * m + j];
                B[k * m + j] = t;
            }
        const double d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; j++) {
            const double alpha = A[j * m + i] * d;
            for (int c = i + 1; c < m; c++) A[j * m + c] += alpha * A[i * m + c];
            for (int c = 0; c < m; c++) B[j * m + c] += alpha * B[i * m + c];
        }
    }
    for (int i = m - 1; i >= 0; i--)
        for (int j = 0; j < m; j++) {
            double s = B[i * m + j];
            for (int c = i + 1; c < m; c++) s -= A[i * m + c] * B[c * m + j];
            B[i * m + j] = s / A[i * m + i];
        }
    memcpy(out, B, sizeof(B));
    return 1;
}


}  // namespace orc_linalg
#endif
