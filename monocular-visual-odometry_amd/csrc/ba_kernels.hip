// csrc/ba_kernels.hip -- sliding-window bundle adjustment on gfx950: replaces the g2o stack that the reference's
// optimization::bundleAdjustment drives (src/optimization/g2o_ba.cpp:193-289: SparseOptimizer::optimize(50) with
// OptimizationAlgorithmLevenberg, BlockSolver<6,3>, LinearSolverDense, EdgeProjectXYZ2UV + RobustKernelHuber).
//
// One persistent workgroup (16 waves) runs the WHOLE Levenberg-Marquardt loop -- 50 outer iterations with their
// data-dependent accept/reject trials -- in a single launch: the problem (a few MB) stays L2/LDS resident and
// there is no host round trip per trial.  Several windows (sequences) run concurrently on different CUs.
// Both dense contractions run on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//   * pose blocks:  [H_pp | -b_p] = M^T M with M = sqrt(rho') * L_Omega * [J_pose | e] (2 rows per edge, K = 2 E_p)
//   * Schur:        [sum_l W_l D_l^-1 W_l^T | sum_l W_l D_l^-1 b_l] = UT^T UT with UT[3l+k][6p+i] = (W_lp C_l)[i][k],
//                   D_l^-1 = C_l C_l^T, and one extra column C_l^T b_l (a (6F+1)-wide SYRK over K = 3 L)
// everything else (edge linearisation, 3x3 point blocks, back-substitution, SE3 exp update, robust chi2) is
// per-edge / per-point VALU work; the reduced 6F x 6F system is LDL^T-factorised by one wave in LDS.
// Arithmetic is f64 throughout like g2o.  Parity target: <= 1e-4 relative on poses / landmarks vs the oracle.
#include "mvo_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define BA_THREADS 1024
#define BA_WAVES 16
#define BA_MAX_POSES 20

struct BaStatsDev {
    int iterations, trials, terminated, pad;
    double chi2_initial, chi2_final, lambda_final;
};

struct BaDev {
    int F, L, E, nfree, n, NT, W, KS, fix_points, max_it, use_mfma;
    double f, cx, cy, delta;
    double lc00, lc01, lc11;  // upper Cholesky factor of the information matrix: Omega = Lc^T Lc
    const double* poses_in;   // F x 16
    double* poses_out;        // F x 16
    const double* pts_in;     // L x 3 initial landmarks (never written: the window can be re-solved)
    double* pts;              // L x 3 working copy / result
    double* pts_bak;          // L x 3
    const int* e_pose;        // E (sorted by pose)
    const int* e_point;       // E
    const double* e_uv;       // E x 2
    const int* pose_edge_start;  // F + 1
    const int* pose_slot;        // F: index among the free poses or -1
    const int* pt_edge_start;    // L + 1
    const int* pt_edge_list;     // E
    const unsigned char* pt_free;  // L
    double* M;     // E x 16: two rows [A~(6) | e~ | 0] per edge
    double* Xt;    // E x 6:  X~ = sqrt(rho') Lc J_point (2 x 3)
    double* Hll;   // L x 6 (xx xy xz yy yz zz)
    double* bl;    // L x 3
    double* Dinv;  // L x 6
    double* Cc;    // L x 6 (c00 c10 c11 c20 c21 c22)
    double* dxl;   // L x 3
    double* UT;    // 3L x W
    double* part;  // max(16, items) x 256
    BaStatsDev* stats;
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// deterministic block reductions (fixed lane order inside a wave, waves summed in index order)
__device__ double block_sum(double v, double* scratch) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < BA_WAVES; ++w) s += scratch[w];
    return s;
}
__device__ double block_max(double v, double* scratch) {
    v = wave_max_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < BA_WAVES; ++w) s = fmax(s, scratch[w]);
    return s;
}

__device__ void quat_normalize(double* q) {
    if (q[0] < 0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
__device__ void quat_from_R(const double* R, double* q) {  // Eigen::Quaterniond(Matrix3d)
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
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
        double t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
__device__ void quat_to_R(const double* q, double* R) {  // Eigen toRotationMatrix
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
__device__ void inv3(const double* A, double* I) {
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
// [R t; 0 1]^-1 of a row-major 4x4 -> Ri (9), ti (3)
__device__ void invert_Rt(const double* T, double* Ri, double* ti) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    inv3(R, Ri);
    for (int i = 0; i < 3; ++i) ti[i] = -(Ri[3 * i] * T[3] + Ri[3 * i + 1] * T[7] + Ri[3 * i + 2] * T[11]);
}
// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(u) * T   (pose = q[4] t[3])
__device__ void pose_oplus(double* P, const double* u) {
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
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
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta),
               c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double dq[4], dt[3], dR[9], nt[3];
    quat_from_R(R, dq);
    quat_normalize(dq);
    for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    quat_to_R(dq, dR);
    const double* q = P;
    const double* t = P + 4;
    for (int i = 0; i < 3; ++i) nt[i] = dt[i] + dR[3 * i] * t[0] + dR[3 * i + 1] * t[1] + dR[3 * i + 2] * t[2];
    double nq[4] = {dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2] - dq[3] * q[3],
                    dq[0] * q[1] + dq[1] * q[0] + dq[2] * q[3] - dq[3] * q[2],
                    dq[0] * q[2] - dq[1] * q[3] + dq[2] * q[0] + dq[3] * q[1],
                    dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1] + dq[3] * q[0]};
    for (int i = 0; i < 4; ++i) P[i] = nq[i];
    for (int i = 0; i < 3; ++i) P[4 + i] = nt[i];
    quat_normalize(P);
}

__device__ __forceinline__ void huber(double e, double delta, double& rho0, double& rho1) {
    double dsqr = delta * delta;
    if (e <= dsqr) {
        rho0 = e;
        rho1 = 1.;
    } else {
        double sqrte = sqrt(e);
        rho0 = 2 * sqrte * delta - dsqr;
        rho1 = delta / sqrte;
    }
}

// EdgeProjectXYZ2UV::computeError with the information factor applied: returns chi2, fills whitened error
__device__ __forceinline__ double edge_error(const BaDev& B, int e, const double* sR, const double* sT, double* Xc,
                                             double* ew) {
    const int p = B.e_pose[e], l = B.e_point[e];
    const double* R = sR + 9 * p;
    const double* t = sT + 3 * p;
    const double X0 = B.pts[3 * l], X1 = B.pts[3 * l + 1], X2 = B.pts[3 * l + 2];
    Xc[0] = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0];
    Xc[1] = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
    Xc[2] = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
    const double e0 = B.e_uv[2 * e] - (Xc[0] / Xc[2] * B.f + B.cx);
    const double e1 = B.e_uv[2 * e + 1] - (Xc[1] / Xc[2] * B.f + B.cy);
    ew[0] = B.lc00 * e0 + B.lc01 * e1;
    ew[1] = B.lc11 * e1;
    return ew[0] * ew[0] + ew[1] * ew[1];
}

__device__ double robust_chi2(const BaDev& B, const double* sR, const double* sT, double* scratch) {
    double s = 0;
    for (int e = threadIdx.x; e < B.E; e += BA_THREADS) {
        double Xc[3], ew[2], r0, r1;
        huber(edge_error(B, e, sR, sT, Xc, ew), B.delta, r0, r1);
        s += r0;
    }
    return block_sum(s, scratch);
}

__global__ __launch_bounds__(BA_THREADS) void k_ba_lm(BaDev B) {
    extern __shared__ __attribute__((aligned(16))) double S[];  // n x (n+1) reduced system, then scratch
    __shared__ double sP[BA_MAX_POSES * 8], sPbak[BA_MAX_POSES * 8];  // q[4] t[3] pad
    __shared__ double sR[BA_MAX_POSES * 9], sT[BA_MAX_POSES * 3];
    __shared__ double sHpp[BA_MAX_POSES * 36], sBp[BA_MAX_POSES * 6], sDx[BA_MAX_POSES * 6];
    __shared__ double sPart[BA_WAVES * 49];
    __shared__ double sScr[BA_WAVES];
    __shared__ double sLcol[6 * BA_MAX_POSES], sCol[6 * BA_MAX_POSES];
    __shared__ int sFlag[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = B.n, ld = n + 1;

    // ---- load poses: T_w_c.inv() -> SE3Quat(R, t)  (g2o_ba.cpp:185-190, 208-215)
    if (tid < B.F) {
        double Ri[9], ti[3], q[4];
        invert_Rt(B.poses_in + 16 * tid, Ri, ti);
        quat_from_R(Ri, q);
        quat_normalize(q);
        for (int i = 0; i < 4; ++i) sP[8 * tid + i] = q[i];
        for (int i = 0; i < 3; ++i) sP[8 * tid + 4 + i] = ti[i];
        quat_to_R(q, sR + 9 * tid);
        for (int i = 0; i < 3; ++i) sT[3 * tid + i] = ti[i];
    }
    for (int i = tid; i < 3 * B.L; i += BA_THREADS) B.pts[i] = B.pts_in[i];
    __syncthreads();

    double lambda = 0, ni = 2;
    int it = 0, trials = 0, terminated = 0;
    double currentChi = robust_chi2(B, sR, sT, sScr);
    const double chi0 = currentChi;
    const bool any_free = B.nfree > 0 || !B.fix_points;

    for (it = 0; any_free && it < B.max_it; ++it) {
        // ================= LIN: per-edge whitened Jacobians (EdgeProjectXYZ2UV::linearizeOplus)
        for (int e = tid; e < B.E; e += BA_THREADS) {
            double Xc[3], ew[2], r0, r1;
            const double chi = edge_error(B, e, sR, sT, Xc, ew);
            huber(chi, B.delta, r0, r1);
            const double sw = sqrt(r1);
            const int p = B.e_pose[e];
            const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = B.f;
            double* Mr = B.M + 16 * (size_t)e;
            if (B.pose_slot[p] >= 0) {
                const double J0[6] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f};
                const double J1[6] = {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    Mr[c] = sw * (B.lc00 * J0[c] + B.lc01 * J1[c]);
                    Mr[8 + c] = sw * (B.lc11 * J1[c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) Mr[c] = Mr[8 + c] = 0;
            }
            Mr[6] = sw * ew[0];
            Mr[14] = sw * ew[1];
            Mr[7] = Mr[15] = 0;
            if (!B.fix_points) {
                const double* R = sR + 9 * p;
                const double t0[3] = {f, 0, -x / z * f}, t1[3] = {0, f, -y / z * f};
                double* Xr = B.Xt + 6 * (size_t)e;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double j0 = -1. / z * (t0[0] * R[c] + t0[1] * R[3 + c] + t0[2] * R[6 + c]);
                    double j1 = -1. / z * (t1[0] * R[c] + t1[1] * R[3 + c] + t1[2] * R[6 + c]);
                    Xr[c] = sw * (B.lc00 * j0 + B.lc01 * j1);
                    Xr[3 + c] = sw * (B.lc11 * j1);
                }
            }
        }
        __syncthreads();
        // ================= HPP: [H_pp | -b_p] = M^T M per free pose
        for (int p = 0; p < B.F; ++p) {
            if (B.pose_slot[p] < 0) continue;
            const int s = B.pose_edge_start[p], e = B.pose_edge_start[p + 1];
            if (B.use_mfma) {
                const int steps = (2 * (e - s) + 3) / 4;
                v4d acc = {0, 0, 0, 0};
                const int col = lane & 15;
                for (int st = wave; st < steps; st += BA_WAVES) {
                    const int row = 4 * st + (lane >> 4);
                    const int ed = s + (row >> 1);
                    double v = (ed < e && col < 8) ? B.M[16 * (size_t)ed + 8 * (row & 1) + col] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rg = (lane >> 4) + 4 * j;
                    if (rg < 7 && col < 7) sPart[wave * 49 + rg * 7 + col] = acc[j];
                }
                __syncthreads();
                if (tid < 49) {
                    double g = 0;
#pragma unroll
                    for (int w = 0; w < BA_WAVES; ++w) g += sPart[w * 49 + tid];
                    const int i = tid / 7, j = tid % 7;
                    if (i < 6 && j < 6) sHpp[36 * p + 6 * i + j] = g;
                    if (i < 6 && j == 6) sBp[6 * p + i] = -g;
                }
            } else {
                if (tid < 49) {
                    const int i = tid / 7, j = tid % 7;
                    double g = 0;
                    for (int r = 2 * s; r < 2 * e; ++r) g += B.M[8 * (size_t)r + i] * B.M[8 * (size_t)r + j];
                    if (i < 6 && j < 6) sHpp[36 * p + 6 * i + j] = g;
                    if (i < 6 && j == 6) sBp[6 * p + i] = -g;
                }
            }
            __syncthreads();
        }
        // ================= PT: 3x3 point blocks H_ll, b_l
        double maxdiag = 0;
        if (!B.fix_points) {
            for (int l = tid; l < B.L; l += BA_THREADS) {
                if (!B.pt_free[l]) continue;
                double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
                for (int k = B.pt_edge_start[l]; k < B.pt_edge_start[l + 1]; ++k) {
                    const int e = B.pt_edge_list[k];
                    const double* X = B.Xt + 6 * (size_t)e;
                    const double e0 = B.M[16 * (size_t)e + 6], e1 = B.M[16 * (size_t)e + 14];
                    h[0] += X[0] * X[0] + X[3] * X[3];
                    h[1] += X[0] * X[1] + X[3] * X[4];
                    h[2] += X[0] * X[2] + X[3] * X[5];
                    h[3] += X[1] * X[1] + X[4] * X[4];
                    h[4] += X[1] * X[2] + X[4] * X[5];
                    h[5] += X[2] * X[2] + X[5] * X[5];
                    b[0] -= X[0] * e0 + X[3] * e1;
                    b[1] -= X[1] * e0 + X[4] * e1;
                    b[2] -= X[2] * e0 + X[5] * e1;
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) B.Hll[6 * (size_t)l + i] = h[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) B.bl[3 * (size_t)l + i] = b[i];
                maxdiag = fmax(maxdiag, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))));
            }
        }
        if (it == 0) {  // computeLambdaInit: tau * max |diag H| over the free vertices
            if (tid < 6 * B.F && B.pose_slot[tid / 6] >= 0)
                maxdiag = fmax(maxdiag, fabs(sHpp[36 * (tid / 6) + 7 * (tid % 6)]));
            lambda = 1e-5 * block_max(maxdiag, sScr);
            ni = 2;
        }
        __syncthreads();

        double rho = 0;
        int qmax = 0;
        do {
            // ============= T1: D^-1 = (H_ll + lambda I)^-1 = C C^T, UT rows of every free point
            if (!B.fix_points) {
                for (int l = tid; l < B.L; l += BA_THREADS) {
                    if (!B.pt_free[l]) continue;
                    const double* h = B.Hll + 6 * (size_t)l;
                    const double D[9] = {h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda};
                    double Di[9];
                    inv3(D, Di);
                    double* di = B.Dinv + 6 * (size_t)l;
                    di[0] = Di[0];
                    di[1] = Di[1];
                    di[2] = Di[2];
                    di[3] = Di[4];
                    di[4] = Di[5];
                    di[5] = Di[8];
                    // Cholesky D^-1 = C C^T (lower)
                    const double c00 = sqrt(Di[0]), c10 = Di[3] / c00, c20 = Di[6] / c00;
                    const double c11 = sqrt(Di[4] - c10 * c10), c21 = (Di[7] - c20 * c10) / c11;
                    const double c22 = sqrt(Di[8] - c20 * c20 - c21 * c21);
                    double* cc = B.Cc + 6 * (size_t)l;
                    cc[0] = c00;
                    cc[1] = c10;
                    cc[2] = c11;
                    cc[3] = c20;
                    cc[4] = c21;
                    cc[5] = c22;
                    const double* b = B.bl + 3 * (size_t)l;
                    double* u0 = B.UT + (size_t)(3 * l) * B.W;
                    double* u1 = u0 + B.W;
                    double* u2 = u1 + B.W;
                    // extra column n: C^T b_l
                    u0[n] = c00 * b[0] + c10 * b[1] + c20 * b[2];
                    u1[n] = c11 * b[1] + c21 * b[2];
                    u2[n] = c22 * b[2];
                    const int k0 = B.pt_edge_start[l], k1 = B.pt_edge_start[l + 1];
                    for (int k = k0; k < k1; ++k) {
                        const int sl = B.pose_slot[B.e_pose[B.pt_edge_list[k]]];
                        if (sl < 0) continue;
#pragma unroll
                        for (int i = 0; i < 6; ++i) u0[6 * sl + i] = u1[6 * sl + i] = u2[6 * sl + i] = 0;
                    }
                    for (int k = k0; k < k1; ++k) {
                        const int e = B.pt_edge_list[k];
                        const int sl = B.pose_slot[B.e_pose[e]];
                        if (sl < 0) continue;
                        const double* X = B.Xt + 6 * (size_t)e;
                        const double* A = B.M + 16 * (size_t)e;
                        // Y = X~ C (2 x 3)
                        const double y00 = X[0] * c00 + X[1] * c10 + X[2] * c20, y01 = X[1] * c11 + X[2] * c21,
                                     y02 = X[2] * c22;
                        const double y10 = X[3] * c00 + X[4] * c10 + X[5] * c20, y11 = X[4] * c11 + X[5] * c21,
                                     y12 = X[5] * c22;
#pragma unroll
                        for (int i = 0; i < 6; ++i) {  // U = A~^T Y (6 x 3), accumulated (duplicate edges add up)
                            u0[6 * sl + i] += A[i] * y00 + A[8 + i] * y10;
                            u1[6 * sl + i] += A[i] * y01 + A[8 + i] * y11;
                            u2[6 * sl + i] += A[i] * y02 + A[8 + i] * y12;
                        }
                    }
                }
            }
            __syncthreads();
            // ============= T2: G = UT^T UT (upper tiles) on the matrix cores, K = 3 L split over the waves
            const int ntile = B.NT * (B.NT + 1) / 2;
            const int K = 3 * B.L;
            if (!B.fix_points && n > 0) {
                if (B.use_mfma) {
                    const int Kc = ((K + B.KS - 1) / B.KS + 3) & ~3;
                    for (int item = wave; item < ntile * B.KS; item += BA_WAVES) {
                        const int tl = item / B.KS, ck = item - tl * B.KS;
                        int ti = 0, rem = tl;  // tl -> (ti <= tj), row-major over the upper triangle
                        while (rem >= B.NT - ti) {
                            rem -= B.NT - ti;
                            ++ti;
                        }
                        const int tj = ti + rem;
                        v4d acc = {0, 0, 0, 0};
                        const int kend = min(K, (ck + 1) * Kc);
                        for (int kk0 = ck * Kc; kk0 < kend; kk0 += 4) {
                            const int kk = kk0 + (lane >> 4);
                            double a = 0, b = 0;
                            if (kk < kend) {
                                a = B.UT[(size_t)kk * B.W + 16 * ti + (lane & 15)];
                                b = B.UT[(size_t)kk * B.W + 16 * tj + (lane & 15)];
                            }
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            B.part[256 * (size_t)item + 16 * ((lane >> 4) + 4 * j) + (lane & 15)] = acc[j];
                    }
                }
            }
            __syncthreads();
            // ============= T3: assemble S = H_pp + lambda I - G, g = b_p - G[:, n]; LDL^T by wave 0
            for (int idx = tid; idx < n * ld; idx += BA_THREADS) {
                const int i = idx / ld, j = idx - i * ld;
                double g = 0;
                if (!B.fix_points) {
                    if (B.use_mfma) {
                        int a = i, b = j;
                        if (a / 16 > b / 16) {
                            a = j;
                            b = i;
                        }
                        const int ti = a / 16, tj = b / 16;
                        const int tl = ti * B.NT - ti * (ti - 1) / 2 + (tj - ti);
                        for (int ck = 0; ck < B.KS; ++ck) g += B.part[256 * (size_t)(tl * B.KS + ck) + 16 * (a % 16) + (b % 16)];
                    } else {
                        for (int kk = 0; kk < K; ++kk) g += B.UT[(size_t)kk * B.W + i] * B.UT[(size_t)kk * B.W + j];
                    }
                }
                double v;
                // slot -> pose lookup through sCol is avoided: free poses keep their order, find pose of slot i/6
                int pi = -1, pj = -1;
                {
                    int si = i / 6, sj = j / 6, c = 0;
                    for (int p = 0; p < B.F; ++p) {
                        if (B.pose_slot[p] < 0) continue;
                        if (c == si) pi = p;
                        if (c == sj) pj = p;
                        ++c;
                    }
                }
                if (j == n)
                    v = sBp[6 * pi + i % 6] - g;
                else
                    v = ((pi == pj) ? sHpp[36 * pi + 6 * (i % 6) + (j % 6)] + (i == j ? lambda : 0.0) : 0.0) - g;
                S[idx] = v;
            }
            __syncthreads();
            if (wave == 0) {
                int ok = 1;
                for (int j = 0; j < n; ++j) {
                    const double d = S[j * ld + j];
                    if (!(d > 0) || !isfinite(d)) {
                        ok = 0;
                        break;
                    }
                    const double gj = S[j * ld + n];
                    for (int i = j + 1 + lane; i < n; i += 64) {
                        const double cij = S[i * ld + j];
                        sCol[i] = cij;
                        sLcol[i] = cij / d;
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int i = j + 1 + lane; i < n; i += 64) {
                        const double li = sLcol[i];
                        for (int k = j + 1; k <= i; ++k) S[i * ld + k] -= li * sCol[k];
                        S[i * ld + n] -= li * gj;
                        S[i * ld + j] = li;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (ok) {
                    for (int i = lane; i < n; i += 64) S[i * ld + n] /= S[i * ld + i];
                    __builtin_amdgcn_wave_barrier();
                    for (int j = n - 1; j >= 0; --j) {
                        const double xj = S[j * ld + n];
                        for (int i = lane; i < j; i += 64) S[i * ld + n] -= S[j * ld + i] * xj;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (lane == 0) sFlag[0] = ok;
            }
            __syncthreads();
            const int ok2 = sFlag[0];
            if (tid < 6 * B.F) {
                const int sl = B.pose_slot[tid / 6];
                sDx[tid] = (ok2 && sl >= 0) ? S[(6 * sl + tid % 6) * ld + n] : 0.0;
            }
            __syncthreads();
            ++trials;
            // ============= T4/T5: back-substitute the points, computeScale, push + apply the update
            double scale = 0;
            if (tid < 6 * B.F && B.pose_slot[tid / 6] >= 0) scale += sDx[tid] * (lambda * sDx[tid] + sBp[tid]);
            if (!B.fix_points) {
                for (int l = tid; l < B.L; l += BA_THREADS) {
                    if (!B.pt_free[l]) continue;
                    const double* b = B.bl + 3 * (size_t)l;
                    double r[3] = {b[0], b[1], b[2]};
                    for (int k = B.pt_edge_start[l]; k < B.pt_edge_start[l + 1]; ++k) {
                        const int e = B.pt_edge_list[k];
                        const int p = B.e_pose[e];
                        if (B.pose_slot[p] < 0) continue;
                        const double* A = B.M + 16 * (size_t)e;
                        const double* X = B.Xt + 6 * (size_t)e;
                        double a0 = 0, a1 = 0;  // A~ dx_p
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            a0 += A[i] * sDx[6 * p + i];
                            a1 += A[8 + i] * sDx[6 * p + i];
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) r[c] -= X[c] * a0 + X[3 + c] * a1;  // W^T dx_p
                    }
                    const double* di = B.Dinv + 6 * (size_t)l;
                    double d[3] = {di[0] * r[0] + di[1] * r[1] + di[2] * r[2], di[1] * r[0] + di[3] * r[1] + di[4] * r[2],
                                   di[2] * r[0] + di[4] * r[1] + di[5] * r[2]};
                    if (!ok2) d[0] = d[1] = d[2] = 0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        scale += d[c] * (lambda * d[c] + b[c]);
                        B.pts_bak[3 * (size_t)l + c] = B.pts[3 * (size_t)l + c];
                        B.pts[3 * (size_t)l + c] += d[c];
                    }
                }
            }
            if (tid < B.F) {
                for (int i = 0; i < 8; ++i) sPbak[8 * tid + i] = sP[8 * tid + i];
                if (B.pose_slot[tid] >= 0) {
                    pose_oplus(sP + 8 * tid, sDx + 6 * tid);
                    quat_to_R(sP + 8 * tid, sR + 9 * tid);
                    for (int i = 0; i < 3; ++i) sT[3 * tid + i] = sP[8 * tid + 4 + i];
                }
            }
            scale = block_sum(scale, sScr) + 1e-3;
            __syncthreads();
            // ============= T6/T7: robust chi2 at the trial state, accept / reject
            double tempChi = robust_chi2(B, sR, sT, sScr);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = (currentChi - tempChi) / scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                __syncthreads();
                if (tid < B.F) {  // pop
                    for (int i = 0; i < 8; ++i) sP[8 * tid + i] = sPbak[8 * tid + i];
                    quat_to_R(sP + 8 * tid, sR + 9 * tid);
                    for (int i = 0; i < 3; ++i) sT[3 * tid + i] = sP[8 * tid + 4 + i];
                }
                if (!B.fix_points)
                    for (int l = tid; l < B.L; l += BA_THREADS)
                        if (B.pt_free[l])
                            for (int c = 0; c < 3; ++c) B.pts[3 * (size_t)l + c] = B.pts_bak[3 * (size_t)l + c];
            }
            __syncthreads();
            ++qmax;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) {
            terminated = 1;
            ++it;
            break;
        }
    }
    // ---- write-back (g2o_ba.cpp:298-305): SE3Quat -> (R, t) -> 4x4 -> inverse
    if (tid < B.F) {
        double R[9], T[16] = {0}, Ri[9], ti[3];
        quat_to_R(sP + 8 * tid, R);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
            T[4 * r + 3] = sP[8 * tid + 4 + r];
        }
        T[15] = 1;
        invert_Rt(T, Ri, ti);
        double* o = B.poses_out + 16 * tid;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) o[4 * r + c] = Ri[3 * r + c];
            o[4 * r + 3] = ti[r];
        }
        o[12] = o[13] = o[14] = 0;
        o[15] = 1;
    }
    if (tid == 0) {
        B.stats->iterations = it;
        B.stats->trials = trials;
        B.stats->terminated = terminated;
        B.stats->chi2_initial = chi0;
        B.stats->chi2_final = currentChi;
        B.stats->lambda_final = lambda;
    }
}

// ================================================================================================ host side
namespace {
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    }
};
}  // namespace

int g_ba_use_mfma = 1;  // debug knob (mvo_debug_set)

// A window whose inputs are resident in HBM: upload once (mvo_ba_prepare), solve any number of times.
struct mvo_ba_handle {
    char* dev = nullptr;  // one allocation holding inputs, CSR tables and workspace
    size_t bytes = 0;
    BaDev B{};
    int F = 0, L = 0;
    size_t o_stats = 0, o_pout = 0, o_pts = 0;
    size_t lds = 16;
    bool fix_points = false;
};

int ba_prepare_device(mvo_ctx* ctx, const mvo_ba_problem* p, mvo_ba_handle** out) {
    *out = nullptr;
    const int F = p->n_poses, L = p->n_points;
    if (F > BA_MAX_POSES) return mvo_set_err(ctx, MVO_ERR_INVALID, "more than 20 poses in the window (vo.h kBuffSize_)", hipSuccess);
    // information matrix must be symmetric positive definite: Omega = Lc^T Lc
    const double a = p->info[0], b = p->info[1], c = p->info[2], d = p->info[3];
    if (!(a > 0) || std::fabs(b - c) > 1e-12 * (std::fabs(a) + std::fabs(d)) || !(a * d - b * b > 0))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "information matrix must be symmetric positive definite", hipSuccess);
    const double lc00 = std::sqrt(a), lc01 = b / lc00, lc11 = std::sqrt(d - lc01 * lc01);
    // drop edges whose vertices are all fixed (SparseOptimizer::initializeOptimization), sort the rest by pose
    std::vector<int> pose_slot(F, -1);
    int nfree = 0;
    for (int i = 0; i < F; ++i)
        if (!(p->pose_fixed && p->pose_fixed[i])) pose_slot[i] = nfree++;
    std::vector<int> pstart(F + 1, 0);
    for (int e = 0; e < p->n_edges; ++e) {
        if (pose_slot[p->edge_pose[e]] < 0 && p->fix_points) continue;
        pstart[p->edge_pose[e] + 1]++;
    }
    for (int i = 0; i < F; ++i) pstart[i + 1] += pstart[i];
    const int E = pstart[F];
    std::vector<int> order(E, 0);
    {
        std::vector<int> cur(pstart.begin(), pstart.end() - 1);
        for (int e = 0; e < p->n_edges; ++e) {
            if (pose_slot[p->edge_pose[e]] < 0 && p->fix_points) continue;
            order[cur[p->edge_pose[e]]++] = e;
        }
    }
    std::vector<int> e_pose(E), e_point(E), ptstart(L + 1, 0), ptlist(E);
    std::vector<double> e_uv(2 * (size_t)E);
    for (int k = 0; k < E; ++k) {
        e_pose[k] = p->edge_pose[order[k]];
        e_point[k] = p->edge_point[order[k]];
        e_uv[2 * k] = p->edge_uv[2 * order[k]];
        e_uv[2 * k + 1] = p->edge_uv[2 * order[k] + 1];
        ptstart[e_point[k] + 1]++;
    }
    for (int i = 0; i < L; ++i) ptstart[i + 1] += ptstart[i];
    {
        std::vector<int> cur(ptstart.begin(), ptstart.end() - 1);
        for (int k = 0; k < E; ++k) ptlist[cur[e_point[k]]++] = k;
    }
    std::vector<unsigned char> pt_free(L, p->fix_points ? 0 : 1);

    const int n = 6 * nfree;
    const int NT = (n + 1 + 15) / 16, W = NT * 16;
    const int ntile = NT * (NT + 1) / 2;
    const int KS = std::max(1, BA_WAVES / ntile);
    Carver cv;
    const size_t o_stats = cv.take(sizeof(BaStatsDev));
    const size_t o_pin = cv.take((size_t)F * 16 * 8), o_pout = cv.take((size_t)F * 16 * 8);
    const size_t o_ptsin = cv.take((size_t)L * 3 * 8);
    const size_t o_ep = cv.take((size_t)E * 4), o_el = cv.take((size_t)E * 4), o_uv = cv.take((size_t)E * 16);
    const size_t o_ps = cv.take((size_t)(F + 1) * 4), o_slot = cv.take((size_t)F * 4 + 4);
    const size_t o_pts_s = cv.take((size_t)(L + 1) * 4), o_ptl = cv.take((size_t)E * 4 + 4), o_pf = cv.take((size_t)L + 4);
    const size_t upload_end = cv.off;
    const size_t o_pts = cv.take((size_t)L * 3 * 8), o_bak = cv.take((size_t)L * 3 * 8);
    const size_t o_M = cv.take((size_t)E * 16 * 8 + 256), o_X = cv.take((size_t)E * 6 * 8 + 256);
    const size_t o_H = cv.take((size_t)L * 6 * 8), o_b = cv.take((size_t)L * 3 * 8), o_D = cv.take((size_t)L * 6 * 8);
    const size_t o_C = cv.take((size_t)L * 6 * 8), o_dx = cv.take((size_t)L * 3 * 8);
    const size_t o_part = cv.take((size_t)std::max(BA_WAVES, ntile * KS) * 256 * 8);
    const size_t o_UT = cv.take(p->fix_points ? 256 : (size_t)3 * L * W * 8 + 256);
    const size_t total = cv.off;
    mvo_ba_handle* H = new mvo_ba_handle();
    hipError_t he = hipMalloc((void**)&H->dev, total);
    if (he != hipSuccess) {
        delete H;
        return mvo_set_err(ctx, MVO_ERR_HIP, "hipMalloc(BA window)", he);
    }
    H->bytes = total;
    int r = mvo_ensure_pinned(ctx, upload_end);
    if (r) {
        (void)hipFree(H->dev);
        delete H;
        return r;
    }
    uint8_t* h = ctx->h_pin;
    std::memset(h, 0, upload_end);
    if (F) std::memcpy(h + o_pin, p->pose_T_w_c, (size_t)F * 16 * 8);
    if (L) std::memcpy(h + o_ptsin, p->points, (size_t)L * 3 * 8);
    if (E) {
        std::memcpy(h + o_ep, e_pose.data(), (size_t)E * 4);
        std::memcpy(h + o_el, e_point.data(), (size_t)E * 4);
        std::memcpy(h + o_uv, e_uv.data(), (size_t)E * 16);
        std::memcpy(h + o_ptl, ptlist.data(), (size_t)E * 4);
    }
    std::memcpy(h + o_ps, pstart.data(), (size_t)(F + 1) * 4);
    if (F) std::memcpy(h + o_slot, pose_slot.data(), (size_t)F * 4);
    std::memcpy(h + o_pts_s, ptstart.data(), (size_t)(L + 1) * 4);
    if (L) std::memcpy(h + o_pf, pt_free.data(), (size_t)L);
    char* D = H->dev;
    hipError_t e1 = hipMemcpyAsync(D, h, upload_end, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = p->fix_points ? hipSuccess : hipMemsetAsync(D + o_UT, 0, (size_t)3 * L * W * 8, ctx->stream);
    hipError_t e3 = hipStreamSynchronize(ctx->stream);  // the pinned staging buffer is reused by later calls
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        (void)hipFree(H->dev);
        delete H;
        return mvo_set_err(ctx, MVO_ERR_HIP, "BA upload", e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3));
    }
    BaDev& B = H->B;
    B.F = F;
    B.L = L;
    B.E = E;
    B.nfree = nfree;
    B.n = n;
    B.NT = NT;
    B.W = W;
    B.KS = KS;
    B.fix_points = p->fix_points ? 1 : 0;
    B.max_it = p->max_iterations;
    B.use_mfma = g_ba_use_mfma;
    B.f = p->focal;
    B.cx = p->cx;
    B.cy = p->cy;
    B.delta = p->huber_delta;
    B.lc00 = lc00;
    B.lc01 = lc01;
    B.lc11 = lc11;
    B.poses_in = (const double*)(D + o_pin);
    B.poses_out = (double*)(D + o_pout);
    B.pts_in = (const double*)(D + o_ptsin);
    B.pts = (double*)(D + o_pts);
    B.pts_bak = (double*)(D + o_bak);
    B.e_pose = (const int*)(D + o_ep);
    B.e_point = (const int*)(D + o_el);
    B.e_uv = (const double*)(D + o_uv);
    B.pose_edge_start = (const int*)(D + o_ps);
    B.pose_slot = (const int*)(D + o_slot);
    B.pt_edge_start = (const int*)(D + o_pts_s);
    B.pt_edge_list = (const int*)(D + o_ptl);
    B.pt_free = (const unsigned char*)(D + o_pf);
    B.M = (double*)(D + o_M);
    B.Xt = (double*)(D + o_X);
    B.Hll = (double*)(D + o_H);
    B.bl = (double*)(D + o_b);
    B.Dinv = (double*)(D + o_D);
    B.Cc = (double*)(D + o_C);
    B.dxl = (double*)(D + o_dx);
    B.UT = (double*)(D + o_UT);
    B.part = (double*)(D + o_part);
    B.stats = (BaStatsDev*)(D + o_stats);
    H->F = F;
    H->L = L;
    H->o_stats = o_stats;
    H->o_pout = o_pout;
    H->o_pts = o_pts;
    H->fix_points = p->fix_points != 0;
    H->lds = std::max<size_t>((size_t)n * (n + 1) * 8, 16);
    *out = H;
    return MVO_OK;
}

// One full LM solve from the resident initial state; results stay on the device.
int ba_run_device(mvo_ctx* ctx, mvo_ba_handle* H) {
    if (H->F == 0 && (H->L == 0 || H->fix_points)) return MVO_OK;
    H->B.use_mfma = g_ba_use_mfma;
    if (H->lds > 32768)
        MVO_HIP(hipFuncSetAttribute((const void*)k_ba_lm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H->lds));
    {
        ProfScope ps(ctx, "k_ba_lm");
        hipLaunchKernelGGL(k_ba_lm, dim3(1), dim3(BA_THREADS), H->lds, ctx->stream, H->B);
    }
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int ba_fetch_device(mvo_ctx* ctx, mvo_ba_handle* H, double* poses, double* points, mvo_ba_stats* st) {
    const size_t need = sizeof(BaStatsDev) + (size_t)H->F * 128 + (size_t)H->L * 24 + 768;
    int r = mvo_ensure_pinned(ctx, need);
    if (r) return r;
    uint8_t* h = ctx->h_pin;
    uint8_t* hp = h + 256;
    uint8_t* hx = hp + (((size_t)H->F * 128 + 255) & ~(size_t)255);
    const bool ran = !(H->F == 0 && (H->L == 0 || H->fix_points));
    if (st) std::memset(st, 0, sizeof(*st));
    if (ran) {
        MVO_HIP(hipMemcpyAsync(h, H->dev + H->o_stats, sizeof(BaStatsDev), hipMemcpyDeviceToHost, ctx->stream));
        if (poses && H->F)
            MVO_HIP(hipMemcpyAsync(hp, H->dev + H->o_pout, (size_t)H->F * 128, hipMemcpyDeviceToHost, ctx->stream));
        if (points && H->L && !H->fix_points)
            MVO_HIP(hipMemcpyAsync(hx, H->dev + H->o_pts, (size_t)H->L * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    if (!ran) return MVO_OK;
    if (poses && H->F) std::memcpy(poses, hp, (size_t)H->F * 128);
    if (points && H->L && !H->fix_points) std::memcpy(points, hx, (size_t)H->L * 24);
    if (st) {
        const BaStatsDev* s = (const BaStatsDev*)h;
        st->iterations = s->iterations;
        st->trials = s->trials;
        st->terminated = s->terminated;
        st->chi2_initial = s->chi2_initial;
        st->chi2_final = s->chi2_final;
        st->lambda_final = s->lambda_final;
    }
    return MVO_OK;
}

void ba_release_device(mvo_ba_handle* H) {
    if (!H) return;
    if (H->dev) (void)hipFree(H->dev);
    delete H;
}

int ba_solve_device(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st) {
    mvo_ba_handle* H = nullptr;
    int r = ba_prepare_device(ctx, p, &H);
    if (r) return r;
    r = ba_run_device(ctx, H);
    if (!r) r = ba_fetch_device(ctx, H, p->pose_T_w_c, p->points, st);
    ba_release_device(H);
    return r;
}
