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

// ------------------------------------------------------------------------------------------- five-point solver
// cv::findEssentialMat's kernel (OpenCV calib3d five-point.cpp, EMEstimatorCallback::runKernel), restated with its
// structure: null space of the 5 x 9 epipolar system, the 10 cubic constraints det(E) = 0 and
// E E^T E - 1/2 tr(E E^T) E = 0 as a 10 x 20 matrix in Nister's monomial order, Gauss-Jordan, the 3 x 3 polynomial
// matrix B(z), its determinant (degree 10), one E per real root.  Where OpenCV has generated code / library calls the
// canonical arithmetic is ours: generic polynomial products in a fixed order instead of getCoeffMat's expanded
// expressions, Gauss-Jordan with partial pivoting instead of inv() * right block, a Sturm-sequence isolation +
// bisection instead of cv::solvePoly, the Jacobi SVD of linalg_oracle.h.
// Monomials: degree <= 1: x y z 1; degree <= 2: x2 xy xz x y2 yz y z2 z 1; degree <= 3 (columns of A):
// x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1.
const int kM1[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
const int kM2[10][3] = {{2, 0, 0}, {1, 1, 0}, {1, 0, 1}, {1, 0, 0}, {0, 2, 0}, {0, 1, 1}, {0, 1, 0}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
const int kM3[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                        {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
int find_mono(const int (*tab)[3], int n, int i, int j, int k) {
    for (int q = 0; q < n; q++)
        if (tab[q][0] == i && tab[q][1] == j && tab[q][2] == k) return q;
    return -1;
}
// out (degree 2) += a (degree 1) * b (degree 1)
void mul11_acc(const double* a, const double* b, double* out) {
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++)
            out[find_mono(kM2, 10, kM1[p][0] + kM1[q][0], kM1[p][1] + kM1[q][1], kM1[p][2] + kM1[q][2])] += a[p] * b[q];
}
// out (degree 3) += a (degree 2) * b (degree 1)
void mul21_acc(const double* a, const double* b, double* out) {
    for (int p = 0; p < 10; p++)
        for (int q = 0; q < 4; q++)
            out[find_mono(kM3, 20, kM2[p][0] + kM1[q][0], kM2[p][1] + kM1[q][1], kM2[p][2] + kM1[q][2])] += a[p] * b[q];
}

double horner(const double* c, int deg, double t) {  // c[0] is the leading coefficient
    double v = c[0];
    for (int i = 1; i <= deg; i++) v = v * t + c[i];
    return v;
}

struct SturmChain {
    double c[12][11];
    int deg[12];
    int len;
};
// number of sign changes of the chain at t (zeros ignored)
int sturm_count(const SturmChain& s, double t) {
    int changes = 0, last = 0;
    for (int k = 0; k < s.len; k++) {
        const double v = horner(s.c[k], s.deg[k], t);
        const int sg = v > 0 ? 1 : v < 0 ? -1 : 0;
        if (sg != 0) {
            if (last != 0 && sg != last) changes++;
            last = sg;
        }
    }
    return changes;
}
// Real roots of c[0] z^10 + ... + c[10], ascending.  Returns the count (<= 10).
int real_roots_deg10(const double* cin, double* roots) {
    SturmChain s;
    int d = 10, off = 0;
    while (d > 0 && cin[off] == 0) {
        off++;
        d--;
    }
    if (d < 1) return 0;
    for (int i = 0; i <= d; i++) s.c[0][i] = cin[off + i];
    s.deg[0] = d;
    for (int i = 0; i < d; i++) s.c[1][i] = s.c[0][i] * (double)(d - i);
    s.deg[1] = d - 1;
    s.len = 2;
    while (s.deg[s.len - 1] > 0) {
        const double* A = s.c[s.len - 2];
        const double* B = s.c[s.len - 1];
        const int a = s.deg[s.len - 2], b = s.deg[s.len - 1];
        double R[11];
        for (int i = 0; i <= a; i++) R[i] = A[i];
        for (int i = 0; i <= a - b; i++) {
            const double q = R[i] / B[0];
            for (int j = 0; j <= b; j++) R[i + j] -= q * B[j];
        }
        int lead = a - b + 1, rd = b - 1;  // remainder = R[lead .. a], degree b - 1
        while (rd >= 0 && R[lead] == 0) {
            lead++;
            rd--;
        }
        if (rd < 0) break;  // exact division: p and p' share a factor, the chain ends here
        for (int i = 0; i <= rd; i++) s.c[s.len][i] = -R[lead + i];
        s.deg[s.len] = rd;
        s.len++;
    }
    double bound = 0;
    for (int i = 1; i <= d; i++) {
        const double r = fabs(s.c[0][i] / s.c[0][0]);
        if (r > bound) bound = r;
    }
    bound = 1 + bound;
    if (!(bound < 1e300)) return 0;
    // level-synchronous isolation: intervals (lo, hi] with the chain's sign-change counts at both ends
    double lo[24], hi[24], ilo[12], ihi[12];
    int vlo[24], vhi[24], n_int = 1, n_iso = 0;
    lo[0] = -bound;
    hi[0] = bound;
    vlo[0] = sturm_count(s, lo[0]);
    vhi[0] = sturm_count(s, hi[0]);
    for (int level = 0; level < 64 && n_int > 0; level++) {
        double nlo[24], nhi[24];
        int nvlo[24], nvhi[24], nn = 0;
        for (int q = 0; q < n_int; q++) {
            const int cnt = vlo[q] - vhi[q];
            if (cnt <= 0) continue;
            if (cnt == 1) {
                if (n_iso < 10) {
                    ilo[n_iso] = lo[q];
                    ihi[n_iso] = hi[q];
                    n_iso++;
                }
                continue;
            }
            const double mid = 0.5 * (lo[q] + hi[q]);
            const int vm = sturm_count(s, mid);
            if (nn + 2 <= 24) {
                nlo[nn] = lo[q];
                nhi[nn] = mid;
                nvlo[nn] = vlo[q];
                nvhi[nn] = vm;
                nn++;
                nlo[nn] = mid;
                nhi[nn] = hi[q];
                nvlo[nn] = vm;
                nvhi[nn] = vhi[q];
                nn++;
            }
        }
        n_int = nn;
        for (int q = 0; q < nn; q++) {
            lo[q] = nlo[q];
            hi[q] = nhi[q];
            vlo[q] = nvlo[q];
            vhi[q] = nvhi[q];
        }
    }
    // ascending order of the isolated intervals (insertion sort on lo)
    for (int i = 1; i < n_iso; i++) {
        const double a = ilo[i], b = ihi[i];
        int j = i - 1;
        while (j >= 0 && ilo[j] > a) {
            ilo[j + 1] = ilo[j];
            ihi[j + 1] = ihi[j];
            j--;
        }
        ilo[j + 1] = a;
        ihi[j + 1] = b;
    }
    int nr = 0;
    for (int q = 0; q < n_iso; q++) {
        double a = ilo[q], b = ihi[q];
        const double fa = horner(s.c[0], d, a), fb = horner(s.c[0], d, b);
        if (fb == 0) {
            roots[nr++] = b;
            continue;
        }
        if ((fa > 0) == (fb > 0)) continue;  // no sign change: a numerically multiple root, not reported
        for (int it = 0; it < 128; it++) {
            const double mid = 0.5 * (a + b);
            if (!(mid > a && mid < b)) break;
            const double fm = horner(s.c[0], d, mid);
            if (fm == 0) {
                a = b = mid;
                break;
            }
            if ((fm > 0) == (fa > 0))
                a = mid;
            else
                b = mid;
        }
        roots[nr++] = 0.5 * (a + b);
    }
    return nr;
}

// q1, q2: the 5 normalised correspondences (x1, y1), (x2, y2).  E: up to 10 candidates (row-major, unit Frobenius
// norm) with x2^T E x1 = 0.  Returns the count.  dbg (optional, 4*9 + 200 + 11 + 10 doubles): null space, reduced A,
// polynomial, roots.
int five_point(const double* q1, const double* q2, double* E, double* dbg) {
    // Null space of Q (5 x 9) as cv::SVD::compute(Q, W, U, Vt, FULL_UV) produces rows 5..8 of Vt: one-sided Jacobi
    // on the 5 rows of Q (padded with a zero row to 6 for the round-robin schedule) gives the 5 right singular
    // vectors; the 4 missing rows are completed the way JacobiSVDImpl_ does it -- a +-1/m vector drawn from
    // cv::RNG(0x12345678) (bit 8 of next()), two rounds of projecting out every earlier row, normalisation.
    double At[9 * 9], Vt6[36], Wq[6];
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int k = 0; k < 9; k++) At[i * 9 + k] = row[k];
    }
    for (int k = 0; k < 9; k++) At[5 * 9 + k] = 0.0;
    jacobi_svd(At, 6, 9, Vt6, Wq);
    for (int i = 0; i < 5; i++) {
        const double sc = Wq[i] > DBL_MIN ? 1 / Wq[i] : 0;
        for (int k = 0; k < 9; k++) At[i * 9 + k] *= sc;
    }
    {
        CvRng rng(0x12345678);
        const int m = 9;
        for (int i = 5; i < 9; i++) {
            double sd = 0;
            for (int ii = 0; ii < 100 && sd <= DBL_MIN; ii++) {
                const double val0 = 1. / m;
                for (int k = 0; k < m; k++) At[i * m + k] = (rng.next() & 256) != 0 ? val0 : -val0;
                for (int iter = 0; iter < 2; iter++)
                    for (int j = 0; j < i; j++) {
                        sd = 0;
                        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[j * m + k];
                        double asum = 0;
                        for (int k = 0; k < m; k++) {
                            const double t = At[i * m + k] - sd * At[j * m + k];
                            At[i * m + k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > DBL_EPSILON * 10 * 100 ? 1 / asum : 0;
                        for (int k = 0; k < m; k++) At[i * m + k] *= asum;
                    }
                sd = 0;
                for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
                sd = sqrt(sd);
            }
            const double sc = sd > DBL_MIN ? 1 / sd : 0.;
            for (int k = 0; k < m; k++) At[i * m + k] *= sc;
        }
    }
    double basis[4][9];
    for (int b = 0; b < 4; b++)
        for (int k = 0; k < 9; k++) basis[b][k] = At[(5 + b) * 9 + k];
    if (dbg) memcpy(dbg, basis, sizeof(basis));
    // E(x, y, z) entries as linear polynomials
    double Ep[3][3][4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            for (int b = 0; b < 4; b++) Ep[r][c][b] = basis[b][3 * r + c];
    double A[10][20];
    {
        double EEt[3][3][10], L[3][3][10], tr[10];
        for (int r = 0; r < 3; r++)
            for (int c = r; c < 3; c++) {
                for (int q = 0; q < 10; q++) EEt[r][c][q] = 0;
                for (int k = 0; k < 3; k++) mul11_acc(Ep[r][k], Ep[c][k], EEt[r][c]);
                if (c != r)
                    for (int q = 0; q < 10; q++) EEt[c][r][q] = EEt[r][c][q];
            }
        for (int q = 0; q < 10; q++) tr[q] = (EEt[0][0][q] + EEt[1][1][q]) + EEt[2][2][q];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                for (int q = 0; q < 10; q++) L[r][c][q] = r == c ? EEt[r][c][q] - 0.5 * tr[q] : EEt[r][c][q];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double* row = A[1 + 3 * r + c];
                for (int q = 0; q < 20; q++) row[q] = 0;
                for (int k = 0; k < 3; k++) mul21_acc(L[r][k], Ep[k][c], row);
            }
        double m[3][10], t1[10], t2[10];
        const int mi[3][4] = {{1, 2, 2, 1}, {0, 2, 2, 0}, {0, 1, 1, 0}};  // minor c: E1a E2b - E1b' E2a' (columns)
        for (int c = 0; c < 3; c++) {
            for (int q = 0; q < 10; q++) t1[q] = t2[q] = 0;
            mul11_acc(Ep[1][mi[c][0]], Ep[2][mi[c][1]], t1);
            mul11_acc(Ep[1][mi[c][2]], Ep[2][mi[c][3]], t2);
            for (int q = 0; q < 10; q++) m[c][q] = t1[q] - t2[q];
        }
        double P[3][20];
        for (int c = 0; c < 3; c++) {
            for (int q = 0; q < 20; q++) P[c][q] = 0;
            mul21_acc(m[c], Ep[0][c], P[c]);
        }
        for (int q = 0; q < 20; q++) A[0][q] = (P[0][q] - P[1][q]) + P[2][q];
    }
    // Gauss-Jordan with partial pivoting on the left 10 x 10 block
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
        if (!(fabs(A[piv][col]) > 0)) return 0;
        if (piv != col)
            for (int c = 0; c < 20; c++) {
                const double t = A[col][c];
                A[col][c] = A[piv][c];
                A[piv][c] = t;
            }
        const double inv = 1.0 / A[col][col];
        for (int c = col + 1; c < 20; c++) A[col][c] *= inv;
        A[col][col] = 1.0;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            const double f = A[r][col];
            for (int c = col + 1; c < 20; c++) A[r][c] -= f * A[col][c];
            A[r][col] = 0.0;
        }
    }
    if (dbg) memcpy(dbg + 36, A, sizeof(A));
    // B(z): rows (4,5), (6,7), (8,9) -> [x: z3 z2 z 1 | y: z3 z2 z 1 | 1: z4 z3 z2 z 1]
    double B[3][13];
    for (int i = 0; i < 3; i++) {
        const double* r1 = &A[4 + 2 * i][10];
        const double* r2 = &A[5 + 2 * i][10];
        double row1[13], row2[13];
        for (int q = 0; q < 13; q++) row1[q] = row2[q] = 0.0;
        for (int q = 0; q < 3; q++) {
            row1[1 + q] = r1[q];
            row1[5 + q] = r1[3 + q];
            row2[q] = r2[q];
            row2[4 + q] = r2[3 + q];
        }
        for (int q = 0; q < 4; q++) {
            row1[9 + q] = r1[6 + q];
            row2[8 + q] = r2[6 + q];
        }
        for (int q = 0; q < 13; q++) B[i][q] = row1[q] - row2[q];
    }
    // det B(z): p1 B1_0 + p2 B1_1 + p3 B1_2 with p1 = Bx1 By2 - Bx2 By1, p2 = Bx2 By0 - Bx0 By2, p3 = Bx0 By1 - Bx1 By0
    auto conv = [](const double* u, int nu, const double* v, int nv, double* out) {
        for (int q = 0; q < nu + nv - 1; q++) out[q] = 0;
        for (int a = 0; a < nu; a++)
            for (int b = 0; b < nv; b++) out[a + b] += u[a] * v[b];
    };
    double p[3][7], c11[11];
    const int pr[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int k = 0; k < 3; k++) {
        double u[7], v[7];
        conv(&B[pr[k][0]][0], 4, &B[pr[k][1]][4], 4, u);
        conv(&B[pr[k][1]][0], 4, &B[pr[k][0]][4], 4, v);
        for (int q = 0; q < 7; q++) p[k][q] = u[q] - v[q];
    }
    {
        double t0[11], t1[11], t2[11];
        conv(p[0], 7, &B[0][8], 5, t0);
        conv(p[1], 7, &B[1][8], 5, t1);
        conv(p[2], 7, &B[2][8], 5, t2);
        for (int q = 0; q < 11; q++) c11[q] = (t0[q] + t1[q]) + t2[q];
    }
    if (dbg) memcpy(dbg + 236, c11, sizeof(c11));
    double roots[10];
    const int nr = real_roots_deg10(c11, roots);
    if (dbg)
        for (int q = 0; q < 10; q++) dbg[247 + q] = q < nr ? roots[q] : NAN;
    int count = 0;
    for (int q = 0; q < nr; q++) {
        const double z1 = roots[q], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double Bz[9], U[9], W3[3], V[9];
        for (int j = 0; j < 3; j++) {
            const double* br = B[j];
            Bz[3 * j] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            Bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            Bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        svd3(Bz, U, W3, V);
        const double w = V[8];  // Vt(2, 2)
        if (fabs(w) < 1e-10) continue;
        const double x = V[2] / w, y = V[5] / w;
        double* e = E + 9 * count;
        double nrm = 0;
        for (int k = 0; k < 9; k++) {
            e[k] = ((basis[0][k] * x + basis[1][k] * y) + basis[2][k] * z1) + basis[3][k];
            nrm += e[k] * e[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++) e[k] /= nrm;
        count++;
    }
    return count;
}

// EMEstimatorCallback::computeError + findInliers: Sampson distance (float) <= thr2.
int score_essential(const double* q1, const double* q2, int n, const double* E, float thr2, uint8_t* mask) {
    int good = 0;
    for (int i = 0; i < n; i++) {
        const double x1[3] = {q1[2 * i], q1[2 * i + 1], 1.}, x2[3] = {q2[2 * i], q2[2 * i + 1], 1.};
        double Ex1[3], Etx2[3];
        for (int r = 0; r < 3; r++) Ex1[r] = E[3 * r] * x1[0] + E[3 * r + 1] * x1[1] + E[3 * r + 2] * x1[2];
        for (int c = 0; c < 3; c++) Etx2[c] = E[c] * x2[0] + E[3 + c] * x2[1] + E[6 + c] * x2[2];
        const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
        const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
        const float err = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
        const int f = err <= thr2;
        if (mask) mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

int ransac_update_num_iters_kf(double p, double ep, int model_points, int max_iters) {
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

void draw_subsets5(int count, int n_iters, int32_t* idx) {
    CvRng rng((uint64_t)-1);
    for (int it = 0; it < n_iters; it++) {
        int32_t* s = idx + it * 5;
        for (int i = 0; i < 5;) {
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
}

}  // namespace

extern "C" {

/* five-point kernel on explicit normalised correspondences (5 x 2 doubles each); E gets up to 10 x 9 doubles. */
int orc_five_point(const double* q1, const double* q2, double* E, double* dbg) { return five_point(q1, q2, E, dbg); }

int orc_real_roots_deg10(const double* c, double* roots) { return real_roots_deg10(c, roots); }

/* geometry::helperFindInlierMatchesByEpipolarCons (motion_estimation.cpp:182-198) = the inlier mask of
 * cv::findEssentialMat(pts1, pts2, focal = (fx + fy) / 2, pp = Point2f(cx, cy), RANSAC, prob, threshold)
 * (epipolar_geometry.cpp:17-47).  kp1 / kp2: matched pixels (n x 2 float).  Returns the inlier count; inliers
 * (ascending indices), optional debug: counts [max_iters x 10] (-1 = no such model), info[4] = {best iteration,
 * best model, iterations run, total models}. */
int orc_find_essential_inliers(const float* kp1, const float* kp2, int n, const double* K4, double prob, double threshold,
                               int max_iters, int32_t* inliers, int32_t* counts, int32_t* info, double* bestE) {
    if (info) info[0] = info[1] = -1, info[2] = info[3] = 0;
    if (n < 5) return 0;
    const double focal = (K4[0] + K4[1]) / 2;
    const double cx = (double)(float)K4[2], cy = (double)(float)K4[3];
    std::vector<double> q1(2 * (size_t)n), q2(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        q1[2 * i] = ((double)kp1[2 * i] - cx) / focal;
        q1[2 * i + 1] = ((double)kp1[2 * i + 1] - cy) / focal;
        q2[2 * i] = ((double)kp2[2 * i] - cx) / focal;
        q2[2 * i + 1] = ((double)kp2[2 * i + 1] - cy) / focal;
    }
    const double thr = threshold / ((focal + focal) / 2);
    const float thr2 = (float)(thr * thr);
    if (n == 5) {  // count == modelPoints: a kernel run, every point an inlier
        double E[90];
        const int nm = five_point(q1.data(), q2.data(), E, nullptr);
        if (info) info[2] = 1, info[3] = nm, info[0] = nm > 0 ? 0 : -1, info[1] = nm > 0 ? 0 : -1;
        if (nm <= 0) return 0;
        if (bestE) memcpy(bestE, E, 9 * sizeof(double));
        for (int i = 0; i < n; i++) inliers[i] = i;
        return n;
    }
    std::vector<int32_t> subsets((size_t)max_iters * 5);
    draw_subsets5(n, max_iters, subsets.data());
    std::vector<uint8_t> mask(n), best_mask(n, 0);
    int niters = max_iters, max_good = 0, it = 0, total_models = 0;
    for (; it < niters; it++) {
        double s1[10], s2[10], E[90];
        for (int k = 0; k < 5; k++) {
            const int s = subsets[(size_t)it * 5 + k];
            s1[2 * k] = q1[2 * s];
            s1[2 * k + 1] = q1[2 * s + 1];
            s2[2 * k] = q2[2 * s];
            s2[2 * k + 1] = q2[2 * s + 1];
        }
        const int nm = five_point(s1, s2, E, nullptr);
        if (counts)
            for (int m = 0; m < 10; m++) counts[it * 10 + m] = -1;
        total_models += nm;
        for (int m = 0; m < nm; m++) {
            const int good = score_essential(q1.data(), q2.data(), n, E + 9 * m, thr2, mask.data());
            if (counts) counts[it * 10 + m] = good;
            if (good > (max_good > 4 ? max_good : 4)) {
                best_mask.swap(mask);
                max_good = good;
                if (bestE) memcpy(bestE, E + 9 * m, 9 * sizeof(double));
                if (info) info[0] = it, info[1] = m;
                niters = ransac_update_num_iters_kf(prob, (double)(n - good) / n, 5, niters);
            }
        }
    }
    if (info) info[2] = it, info[3] = total_models;
    if (max_good <= 0) return 0;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (best_mask[i]) inliers[cnt++] = i;
    return cnt;
}

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
