// csrc/em_wave.h -- one RANSAC hypothesis of cv::findEssentialMat on a single wave (keyframe_kernels in
// track_kernels.hip).  Reference call site: geometry::helperFindInlierMatchesByEpipolarCons
// (src/geometry/motion_estimation.cpp:182-198 -> epipolar_geometry.cpp:17-47), run when a keyframe is inserted
// (src/vo/vo_addFrame.cpp:104-106).
//
// The five-point kernel with OpenCV's structure (calib3d five-point.cpp): null space of the 5 x 9 epipolar system,
// the ten cubic constraints det(E) = 0, E E^T E - 1/2 tr(E E^T) E = 0 in Nister's monomial order, Gauss-Jordan,
// the 3 x 3 polynomial matrix B(z), its degree-10 determinant, one E per real root, Sampson scoring of all matches.
// Same SPMD style as pnp_wave.h (PW_LANES / PW_SYNC / uniform code); canonical arithmetic: DESIGN.md section 9.
#ifndef MVO_EM_WAVE_H
#define MVO_EM_WAVE_H
#include "pnp_wave.h"

namespace pw {

constexpr int kEmLanes = 64;  // one hypothesis per wave
constexpr int kEmMaxModels = 10;

// products of monomials: degree<=1 (x y z 1) x degree<=1 -> index into x2 xy xz x y2 yz y z2 z 1; degree<=2 x
// degree<=1 -> index into x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1 (the columns of A)
PW_FN int mono2(int p, int q) {
    const int t[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
    return t[p][q];
}
PW_FN int mono3(int p, int q) {
    const int t[10][4] = {{0, 2, 4, 5},     {2, 3, 8, 9},     {4, 8, 10, 11},   {5, 9, 11, 12},   {3, 1, 6, 7},
                          {8, 6, 13, 14},   {9, 7, 14, 15},   {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}};
    return t[p][q];
}

struct EmLds {
    double At[6 * 9];     // rows of Q (+ a zero row), rotated in place
    double Vn[9 * 9];     // rows 0..4 right singular vectors of Q, rows 5..8 the null space basis X, Y, Z, W
    double EEt[6 * 10];   // E E^T, upper triangle (00 01 02 11 12 22), degree-2 polynomials
    double L[9 * 10];     // E E^T - 1/2 tr I
    double minor[3 * 10];
    double P[3 * 20];
    double A[10 * 20];
    double B[3 * 13];
    double chain[12 * 11];  // Sturm chain, leading coefficient first
    int deg[12];
    int chain_len;
    double lo[2][24], hi[2][24];  // isolation intervals, double buffered
    int vlo[2][24], vhi[2][24], vm[24];
    double ilo[12], ihi[12], root_raw[12];
    int root_ok[12];
    double Ecand[kEmMaxModels * 9];
    int valid[kEmMaxModels];
    JacobiLds js;
    int cnt[kEmLanes];
};

PW_FN double horner(const double* c, int deg, double t) {  // c[0] is the leading coefficient
    double v = c[0];
    for (int i = 1; i <= deg; i++) v = v * t + c[i];
    return v;
}

PW_FN int sturm_count(const EmLds& s, double t) {
    int changes = 0, last = 0;
    for (int k = 0; k < s.chain_len; k++) {
        const double v = horner(s.chain + 11 * k, s.deg[k], t);
        const int sg = v > 0 ? 1 : v < 0 ? -1 : 0;
        if (sg != 0) {
            if (last != 0 && sg != last) changes++;
            last = sg;
        }
    }
    return changes;
}

// cv::RNG as JacobiSVDImpl_ uses it for the missing singular vectors: RNG rng(0x12345678), bit 8 of next()
struct MwcState {
    uint64_t state;
};
PW_FN unsigned mwc_next(MwcState& r) {
    r.state = (uint64_t)(unsigned)r.state * 4164903690U + (unsigned)(r.state >> 32);
    return (unsigned)r.state;
}

// Real roots (ascending) of c[0] z^10 + ... + c[10] into s.root_raw; returns their number.  Sturm chain, level-
// synchronous bisection of the root-counting intervals (all intervals of a level in parallel), then one lane per
// isolated root bisects on the sign of the polynomial down to adjacent doubles.
PW_FN int real_roots_deg10(EmLds& s, const double (&cin)[11]) {
    int d = 10, off = 0;
    while (d > 0 && cin[off] == 0) {
        off++;
        d--;
    }
    if (d < 1) return 0;
    PW_SYNC();
    for (int i = 0; i <= d; i++) s.chain[i] = cin[off + i];
    s.deg[0] = d;
    for (int i = 0; i < d; i++) s.chain[11 + i] = s.chain[i] * (double)(d - i);
    s.deg[1] = d - 1;
    int len = 2;
    while (s.deg[len - 1] > 0) {
        const double* A = s.chain + 11 * (len - 2);
        const double* B = s.chain + 11 * (len - 1);
        const int a = s.deg[len - 2], b = s.deg[len - 1];
        double R[11];
        for (int i = 0; i <= a; i++) R[i] = A[i];
        for (int i = 0; i <= a - b; i++) {
            const double q = R[i] / B[0];
            for (int j = 0; j <= b; j++) R[i + j] -= q * B[j];
        }
        int lead = a - b + 1, rd = b - 1;
        while (rd >= 0 && R[lead] == 0) {
            lead++;
            rd--;
        }
        if (rd < 0) break;
        for (int i = 0; i <= rd; i++) s.chain[11 * len + i] = -R[lead + i];
        s.deg[len] = rd;
        len++;
    }
    s.chain_len = len;
    PW_SYNC();
    double bound = 0;
    for (int i = 1; i <= d; i++) {
        const double r = fabs(s.chain[i] / s.chain[0]);
        if (r > bound) bound = r;
    }
    bound = 1 + bound;
    if (!(bound < 1e300)) return 0;
    int n_int = 1, n_iso = 0, cur = 0;
    s.lo[0][0] = -bound;
    s.hi[0][0] = bound;
    s.vlo[0][0] = sturm_count(s, -bound);
    s.vhi[0][0] = sturm_count(s, bound);
    PW_SYNC();
    for (int level = 0; level < 64 && n_int > 0; level++) {
        PW_LANES(l, kEmLanes) {
            if (l < n_int && s.vlo[cur][l] - s.vhi[cur][l] >= 2) s.vm[l] = sturm_count(s, 0.5 * (s.lo[cur][l] + s.hi[cur][l]));
        }
        PW_SYNC();
        const int nxt = cur ^ 1;
        int nn = 0;
        for (int q = 0; q < n_int; q++) {
            const int cnt = s.vlo[cur][q] - s.vhi[cur][q];
            if (cnt <= 0) continue;
            if (cnt == 1) {
                if (n_iso < 10) {
                    s.ilo[n_iso] = s.lo[cur][q];
                    s.ihi[n_iso] = s.hi[cur][q];
                    n_iso++;
                }
                continue;
            }
            const double mid = 0.5 * (s.lo[cur][q] + s.hi[cur][q]);
            const int vm = s.vm[q];
            if (nn + 2 <= 24) {
                s.lo[nxt][nn] = s.lo[cur][q];
                s.hi[nxt][nn] = mid;
                s.vlo[nxt][nn] = s.vlo[cur][q];
                s.vhi[nxt][nn] = vm;
                nn++;
                s.lo[nxt][nn] = mid;
                s.hi[nxt][nn] = s.hi[cur][q];
                s.vlo[nxt][nn] = vm;
                s.vhi[nxt][nn] = s.vhi[cur][q];
                nn++;
            }
        }
        n_int = nn;
        cur = nxt;
        PW_SYNC();
    }
    for (int i = 1; i < n_iso; i++) {  // ascending order of the isolated intervals
        const double a = s.ilo[i], b = s.ihi[i];
        int j = i - 1;
        while (j >= 0 && s.ilo[j] > a) {
            s.ilo[j + 1] = s.ilo[j];
            s.ihi[j + 1] = s.ihi[j];
            j--;
        }
        s.ilo[j + 1] = a;
        s.ihi[j + 1] = b;
    }
    PW_SYNC();
    PW_LANES(l, kEmLanes) {
        if (l < n_iso) {
            double a = s.ilo[l], b = s.ihi[l];
            const double fa = horner(s.chain, d, a), fb = horner(s.chain, d, b);
            int ok = 1;
            double root = 0;
            if (fb == 0) {
                root = b;
            } else if ((fa > 0) == (fb > 0)) {
                ok = 0;
            } else {
                for (int it = 0; it < 128; it++) {
                    const double mid = 0.5 * (a + b);
                    if (!(mid > a && mid < b)) break;
                    const double fm = horner(s.chain, d, mid);
                    if (fm == 0) {
                        a = b = mid;
                        break;
                    }
                    if ((fm > 0) == (fa > 0))
                        a = mid;
                    else
                        b = mid;
                }
                root = 0.5 * (a + b);
            }
            s.root_raw[l] = root;
            s.root_ok[l] = ok;
        }
    }
    PW_SYNC();
    int nr = 0;
    for (int q = 0; q < n_iso; q++)
        if (s.root_ok[q]) {
            const double r = s.root_raw[q];
            s.root_raw[nr++] = r;
        }
    PW_SYNC();
    return nr;
}

// q1 / q2: normalised coordinates of ALL matches (n x 2 doubles each); idx: the 5 matches of this hypothesis.
// E_out: up to 10 candidates x 9 doubles (row-major, unit norm, x2^T E x1 = 0).  Returns their number.
PW_FN int five_point_hypothesis(EmLds& s, const double* q1, const double* q2, const int32_t* idx, double* E_out) {
    // ---- null space of the 5 x 9 system: Jacobi on its rows, then the completion cv::SVD(FULL_UV) performs
    PW_LANES(l, kEmLanes) {
        if (l < 54) {
            const int r = l / 9, k = l % 9;
            double v = 0.0;
            if (r < 5) {
                const int i = idx[r];
                const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
                v = k == 0 ? x2 * x1 : k == 1 ? x2 * y1 : k == 2 ? x2 : k == 3 ? y2 * x1 : k == 4 ? y2 * y1 : k == 5 ? y2 : k == 6 ? x1 : k == 7 ? y1 : 1.0;
            }
            s.At[l] = v;
        }
    }
    PW_SYNC();
    jacobi_rr<6, 9, kEmLanes>(s.At, nullptr, s.js);
    PW_LANES(l, kEmLanes) {
        if (l < 45) {
            const int p = l / 9, k = l % 9, src = s.js.perm[p];
            const double w = s.js.W[src];
            const double sc = w > DBL_MIN ? 1 / w : 0;
            s.Vn[l] = s.At[src * 9 + k] * sc;
        }
    }
    PW_SYNC();
    {
        MwcState rng{0x12345678ULL};
        const int m = 9;
        for (int i = 5; i < 9; i++) {
            double cur[9], sd = 0;
            for (int ii = 0; ii < 100 && sd <= DBL_MIN; ii++) {
                const double val0 = 1. / m;
                PW_UNROLL
                for (int k = 0; k < 9; k++) cur[k] = (mwc_next(rng) & 256) != 0 ? val0 : -val0;
                for (int iter = 0; iter < 2; iter++)
                    for (int j = 0; j < i; j++) {
                        const double* vj = s.Vn + 9 * j;
                        sd = 0;
                        PW_UNROLL
                        for (int k = 0; k < 9; k++) sd += cur[k] * vj[k];
                        double asum = 0;
                        PW_UNROLL
                        for (int k = 0; k < 9; k++) {
                            const double t = cur[k] - sd * vj[k];
                            cur[k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > DBL_EPSILON * 10 * 100 ? 1 / asum : 0;
                        PW_UNROLL
                        for (int k = 0; k < 9; k++) cur[k] *= asum;
                    }
                sd = 0;
                PW_UNROLL
                for (int k = 0; k < 9; k++) sd += cur[k] * cur[k];
                sd = sqrt(sd);
            }
            const double sc = sd > DBL_MIN ? 1 / sd : 0.;
            PW_SYNC();
            PW_UNROLL
            for (int k = 0; k < 9; k++) s.Vn[9 * i + k] = cur[k] * sc;
            PW_SYNC();
        }
    }
    // E(x, y, z)[r][c] = x X + y Y + z Z + W: coefficient b of entry (r, c) = Vn[5 + b][3 r + c]
#define EM_EP(r, c, b) s.Vn[(5 + (b)) * 9 + 3 * (r) + (c)]
    // ---- E E^T (upper triangle) and the 2 x 2 minors of rows 1, 2
    PW_LANES(l, kEmLanes) {
        for (int t = l; t < 60 + 30; t += kEmLanes) {
            if (t < 60) {
                const int pi = t / 10, q = t % 10;
                const int r = pi < 3 ? 0 : pi < 5 ? 1 : 2, c = pi < 3 ? pi : pi < 5 ? pi - 2 : 2;
                double sum = 0;
                for (int k = 0; k < 3; k++)
                    for (int p = 0; p < 4; p++)
                        for (int qq = 0; qq < 4; qq++)
                            if (mono2(p, qq) == q) sum += EM_EP(r, k, p) * EM_EP(c, k, qq);
                s.EEt[t] = sum;
            } else {
                const int c = (t - 60) / 10, q = (t - 60) % 10;
                const int a0 = c == 0 ? 1 : 0, b0 = c == 2 ? 1 : 2;  // minor c = E1,a0 E2,b0 - E1,b0 E2,a0
                double t1 = 0, t2 = 0;
                for (int p = 0; p < 4; p++)
                    for (int qq = 0; qq < 4; qq++)
                        if (mono2(p, qq) == q) {
                            t1 += EM_EP(1, a0, p) * EM_EP(2, b0, qq);
                            t2 += EM_EP(1, b0, p) * EM_EP(2, a0, qq);
                        }
                s.minor[t - 60] = t1 - t2;
            }
        }
    }
    PW_SYNC();
    // ---- L = E E^T - 1/2 tr I and P_c = minor_c * E0c
    PW_LANES(l, kEmLanes) {
        for (int t = l; t < 90 + 60; t += kEmLanes) {
            if (t < 90) {
                const int rc = t / 10, q = t % 10, r = rc / 3, c = rc % 3;
                const int lo = r < c ? r : c, hi = r < c ? c : r;
                const int pi = lo == 0 ? hi : lo == 1 ? 2 + hi : 5;
                const double e = s.EEt[pi * 10 + q];
                const double tr = (s.EEt[0 * 10 + q] + s.EEt[3 * 10 + q]) + s.EEt[5 * 10 + q];
                s.L[t] = r == c ? e - 0.5 * tr : e;
            } else {
                const int c = (t - 90) / 20, q = (t - 90) % 20;
                double sum = 0;
                for (int p = 0; p < 10; p++)
                    for (int qq = 0; qq < 4; qq++)
                        if (mono3(p, qq) == q) sum += s.minor[c * 10 + p] * EM_EP(0, c, qq);
                s.P[t - 90] = sum;
            }
        }
    }
    PW_SYNC();
    // ---- the 10 x 20 constraint matrix: row 0 = det(E), rows 1..9 = (L E)[r][c]
    PW_LANES(l, kEmLanes) {
        for (int t = l; t < 200; t += kEmLanes) {
            const int row = t / 20, q = t % 20;
            double v;
            if (row == 0) {
                v = (s.P[q] - s.P[20 + q]) + s.P[40 + q];
            } else {
                const int r = (row - 1) / 3, c = (row - 1) % 3;
                v = 0;
                for (int k = 0; k < 3; k++)
                    for (int p = 0; p < 10; p++)
                        for (int qq = 0; qq < 4; qq++)
                            if (mono3(p, qq) == q) v += s.L[(3 * r + k) * 10 + p] * EM_EP(k, c, qq);
            }
            s.A[t] = v;
        }
    }
    PW_SYNC();
    // ---- Gauss-Jordan with partial pivoting on the left 10 x 10 block
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (fabs(s.A[r * 20 + col]) > fabs(s.A[piv * 20 + col])) piv = r;
        if (!(fabs(s.A[piv * 20 + col]) > 0)) return 0;
        PW_SYNC();
        if (piv != col) {
            PW_LANES(l, kEmLanes) {
                if (l < 20) {
                    const double t = s.A[col * 20 + l];
                    s.A[col * 20 + l] = s.A[piv * 20 + l];
                    s.A[piv * 20 + l] = t;
                }
            }
            PW_SYNC();
        }
        const double inv = 1.0 / s.A[col * 20 + col];
        PW_SYNC();
        PW_LANES(l, kEmLanes) {
            if (l > col && l < 20) s.A[col * 20 + l] *= inv;
            if (l == col) s.A[col * 20 + l] = 1.0;
        }
        PW_SYNC();
        PW_LANES(l, kEmLanes) {
            for (int t = l; t < 200; t += kEmLanes) {
                const int r = t / 20, c = t % 20;
                if (r != col && c > col) s.A[t] -= s.A[r * 20 + col] * s.A[col * 20 + c];
            }
        }
        PW_SYNC();
        PW_LANES(l, kEmLanes) {
            if (l < 10 && l != col) s.A[l * 20 + col] = 0.0;
        }
        PW_SYNC();
    }
    // ---- B(z) from rows (4,5), (6,7), (8,9): [x: z3 z2 z 1 | y: z3 z2 z 1 | 1: z4 z3 z2 z 1]
    double B[3][13];
    PW_UNROLL
    for (int i = 0; i < 3; i++) {
        const double* r1 = s.A + (4 + 2 * i) * 20 + 10;
        const double* r2 = s.A + (5 + 2 * i) * 20 + 10;
        double row1[13], row2[13];
        PW_UNROLL
        for (int q = 0; q < 13; q++) row1[q] = row2[q] = 0.0;
        PW_UNROLL
        for (int q = 0; q < 3; q++) {
            row1[1 + q] = r1[q];
            row1[5 + q] = r1[3 + q];
            row2[q] = r2[q];
            row2[4 + q] = r2[3 + q];
        }
        PW_UNROLL
        for (int q = 0; q < 4; q++) {
            row1[9 + q] = r1[6 + q];
            row2[8 + q] = r2[6 + q];
        }
        PW_UNROLL
        for (int q = 0; q < 13; q++) {
            B[i][q] = row1[q] - row2[q];
            s.B[13 * i + q] = B[i][q];
        }
    }
    // ---- det B(z) = p1 B1_0 + p2 B1_1 + p3 B1_2
    double c11[11];
    {
        double p[3][7];
        PW_UNROLL
        for (int k = 0; k < 3; k++) {
            const int i0 = k == 0 ? 1 : k == 1 ? 2 : 0, i1 = k == 0 ? 2 : k == 1 ? 0 : 1;
            double u[7], v[7];
            PW_UNROLL
            for (int q = 0; q < 7; q++) u[q] = v[q] = 0;
            PW_UNROLL
            for (int a = 0; a < 4; a++) {
                PW_UNROLL
                for (int b = 0; b < 4; b++) u[a + b] += B[i0][a] * B[i1][4 + b];
            }
            PW_UNROLL
            for (int a = 0; a < 4; a++) {
                PW_UNROLL
                for (int b = 0; b < 4; b++) v[a + b] += B[i1][a] * B[i0][4 + b];
            }
            PW_UNROLL
            for (int q = 0; q < 7; q++) p[k][q] = u[q] - v[q];
        }
        double t[3][11];
        PW_UNROLL
        for (int k = 0; k < 3; k++) {
            PW_UNROLL
            for (int q = 0; q < 11; q++) t[k][q] = 0;
            PW_UNROLL
            for (int a = 0; a < 7; a++) {
                PW_UNROLL
                for (int b = 0; b < 5; b++) t[k][a + b] += p[k][a] * B[k][8 + b];
            }
        }
        PW_UNROLL
        for (int q = 0; q < 11; q++) c11[q] = (t[0][q] + t[1][q]) + t[2][q];
    }
    const int nr = real_roots_deg10(s, c11);
    // ---- one E per root
    PW_LANES(l, kEmLanes) {
        if (l < nr) {
            const double z1 = s.root_raw[l], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
            double Bz[3][3], U[3][3], W3[3], V[3][3];
            for (int j = 0; j < 3; j++) {
                const double* br = s.B + 13 * j;
                Bz[j][0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
                Bz[j][1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
                Bz[j][2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
            }
            svd3(Bz, U, W3, V);
            const double w = V[2][2];
            int ok = 0;
            if (!(fabs(w) < 1e-10)) {
                ok = 1;
                const double x = V[0][2] / w, y = V[1][2] / w;
                double e[9], nrm = 0;
                for (int k = 0; k < 9; k++) {
                    e[k] = ((s.Vn[5 * 9 + k] * x + s.Vn[6 * 9 + k] * y) + s.Vn[7 * 9 + k] * z1) + s.Vn[8 * 9 + k];
                    nrm += e[k] * e[k];
                }
                nrm = sqrt(nrm);
                for (int k = 0; k < 9; k++) s.Ecand[9 * l + k] = e[k] / nrm;
            }
            s.valid[l] = ok;
        }
    }
    PW_SYNC();
    int count = 0;
    for (int q = 0; q < nr; q++)
        if (s.valid[q]) {
            PW_LANES(l, kEmLanes) {
                if (l < 9) E_out[9 * count + l] = s.Ecand[9 * q + l];
            }
            count++;
        }
    PW_SYNC();
    return count;
#undef EM_EP
}

// EMEstimatorCallback::computeError + findInliers for one candidate: Sampson distance (float) <= thr2.
PW_FN bool sampson_inlier(const double* E, double x1a, double x1b, double x2a, double x2b, float thr2) {
    const double x1[3] = {x1a, x1b, 1.}, x2[3] = {x2a, x2b, 1.};
    double Ex1[3], Etx2[3];
    for (int r = 0; r < 3; r++) Ex1[r] = E[3 * r] * x1[0] + E[3 * r + 1] * x1[1] + E[3 * r + 2] * x1[2];
    for (int c = 0; c < 3; c++) Etx2[c] = E[c] * x2[0] + E[3 + c] * x2[1] + E[6 + c] * x2[2];
    const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
    const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
    const float err = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    return err <= thr2;
}

// counts[m] = inliers of candidate m over all n matches (m < nm), -1 for the unused slots.
PW_FN void score_essentials(EmLds& s, const double* q1, const double* q2, int n, const double* E, int nm, float thr2,
                            int32_t* counts) {
    for (int m = 0; m < kEmMaxModels; m++) {
        if (m >= nm) {
            PW_LANES(l, kEmLanes) {
                if (l == 0) counts[m] = -1;
            }
            continue;
        }
        PW_LANES(l, kEmLanes) {
            int good = 0;
            for (int i = l; i < n; i += kEmLanes)
                good += sampson_inlier(E + 9 * m, q1[2 * i], q1[2 * i + 1], q2[2 * i], q2[2 * i + 1], thr2) ? 1 : 0;
            s.cnt[l] = good;
        }
        PW_SYNC();
        int total = 0;
        for (int q = 0; q < kEmLanes; q++) total += s.cnt[q];
        PW_SYNC();
        PW_LANES(l, kEmLanes) {
            if (l == 0) counts[m] = total;
        }
    }
}

}  // namespace pw
#endif
