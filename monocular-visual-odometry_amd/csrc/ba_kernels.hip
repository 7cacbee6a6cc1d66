// csrc/ba_kernels.hip -- sliding-window bundle adjustment on gfx950: replaces the g2o stack that the reference's
// optimization::bundleAdjustment drives (src/optimization/g2o_ba.cpp:193-289: SparseOptimizer::optimize(50) with
// OptimizationAlgorithmLevenberg, BlockSolver<6,3>, LinearSolverDense, EdgeProjectXYZ2UV + RobustKernelHuber).
//
// ONE persistent launch runs the whole Levenberg-Marquardt loop (50 outer iterations with their data-dependent
// accept/reject trials): there is no host round trip per trial.  The window is split over G workgroups (one per
// CU) by LANDMARK: a workgroup owns a contiguous range of landmarks and every observation (edge) of them, and
// keeps ALL its per-edge / per-landmark state -- whitened Jacobians, landmark blocks, the landmarks themselves --
// in its 160 KB LDS for the lifetime of the launch; G is chosen so that this fits.  Per-edge and per-landmark
// phases are therefore workgroup-local LDS work.  Only three small reductions cross workgroups, each through a
// counter barrier (agent-scope atomics, write-through partials, no L2 flush):
//   * [H_pp | -b_p] pose blocks  = sum over edges of M^T M,   M = sqrt(rho') L_Omega [J_pose | e]     (7x7 / pose)
//   * Schur blocks   sum_l W_l D_l^-1 W_l^T and sum_l W_l D_l^-1 b_l = U^T U with U = (W_l C_l), D_l^-1 = C_l C_l^T
//   * robust chi2 / predicted decrease
// The first two are accumulated on the f64 matrix cores (v_mfma_f64_16x16x4_f64): one MFMA per pair of edges /
// per landmark (3 of the 4 k-slots carry the columns of U_l); after a barrier EVERY workgroup redundantly sums the
// G partials in a fixed order, factorises the reduced 6F x 6F system (LDL^T, one wave, LDS) and takes the same
// accept/reject decision -- no broadcast step.  All arithmetic is f64 like g2o and every reduction has a fixed
// order (bit-reproducible runs).  Parity target: <= 1e-4 relative on poses / landmarks vs the oracle.
#include "mvo_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define BA_THREADS 512
#define BA_WAVES 8
#define BA_MAX_POSES 20
#define BA_MFMA_MAX_NT 4  // matrix-core Schur path up to 64 rows (10 free poses + rhs); beyond: VALU loops
#define BA_MSTRIDE 14     // doubles per edge in M: two rows [A~(6) | e~]
#define BA_LDS_BUDGET (140 * 1024)  // dynamic part; ~18 KB of static LDS come on top (160 KB per CU)
#define BA_MAX_WGS 128
#define BA_TILE_SLOTS(ntile) ((ntile) <= 3 ? 4 : 1)  // LDS slots for the wave reduction tree (big systems: 1, sequential)

#define BA_NPHASE 16
struct BaStatsDev {
    int iterations, trials, terminated, error;
    double chi2_initial, chi2_final, lambda_final;
    long long phase[BA_NPHASE];  // shader-clock cycles per phase as seen by thread 0 of workgroup 0
};
// phase ids: 0 LIN, 1 HPP, 2 PT + pose-block exchange, 3 T1, 4 Schur MFMA, 5 publish + barrier, 6 assemble S,
// 7 LDL^T, 8 back-substitution + update, 9 chi2, 10 chi2 exchange + decision, 11 whole kernel
#define PH_BEGIN() long long ph_t = (long long)__builtin_amdgcn_s_memtime()
#define PH_END(id)                                                   \
    do {                                                             \
        long long ph_n = (long long)__builtin_amdgcn_s_memtime();    \
        ph[id] += ph_n - ph_t;                                       \
        ph_t = ph_n;                                                 \
    } while (0)

struct BaDev {
    int F, L, E, G, nfree, n, NT, ntile, fix_points, max_it, use_mfma, maxEg, maxLg, has_dups;
    double f, cx, cy, delta;
    double lc00, lc01, lc11;  // upper Cholesky factor of the information matrix: Omega = Lc^T Lc
    const double* poses_in;   // F x 16
    double* poses_out;        // F x 16
    const double* pts_in;     // L x 3
    double* pts_out;          // L x 3
    const int* wg_pt_start;   // G + 1   (landmark ranges)
    const int* wg_edge_start; // G + 1   (edges sorted by owner workgroup, then pose)
    const int* wg_pose_start; // G x (F + 1): absolute edge index where pose p starts inside workgroup g
    const int* e_pose;        // E
    const int* e_point;       // E (global landmark index)
    const double* e_uv;       // E x 2
    const int* pt_edge_start; // L + 1 -> pt_edge_list
    const int* pt_edge_list;  // E absolute edge indices, grouped by landmark
    const short* eof;         // L x nfree: LOCAL index of the first edge (landmark, pose slot), -1 if none
    const short* dup_next;    // E: next LOCAL edge with the same (landmark, pose), -1 if none
    const int* pose_slot;     // F
    const int* slot_pose;     // nfree
    // cross-workgroup exchange (agent-scope atomics only)
    double* xHpp;             // G x F x 49
    u64* xGg;                 // G x 2 npk granules: packed Schur partials (lower triangle + rhs)
    u64* xRg;                 // 2 x npk granules: the same entries summed over the workgroups
    u64* xCg;                 // 2 (parity) x G x 4 granules: chi2 / predicted-decrease partials
    double* xSc;              // G x 4: initial chi2 (slot 0) and landmark max diagonal (slot 2), counter-barrier phases
    unsigned* barrier;        // monotonically increasing arrival counter (zeroed before every launch)
    BaStatsDev* stats;
    // pinned host mirrors of stats / poses_out (written next to the device copies at the end of the solve, so that
    // fetching the result needs a synchronisation but no copy dispatch); may be null
    BaStatsDev* h_stats;
    double* h_poses;
    // The exchange region (barrier counter | summed-entry granules | chi2 granules | partial granules) must be zero
    // when a launch starts.  There are two of them, used by alternate launches: every launch clears the OTHER one
    // in its prologue (nobody reads it meanwhile), so no memset dispatch precedes the kernel.
    u64* zero_other;
    unsigned zero_words;
};

// ------------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ void xstore(double* p, double v) {  // write-through (sc1) store
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xload(const double* p) {  // L1-bypassing (sc1) load
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}

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
// deterministic block reductions (fixed lane order inside a wave, waves combined in index order)
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

// Data-tagged hand-off (no counter, no second round trip): a double travels as two 8-byte granules
// {tag : 32 | half : 32}, each written by ONE write-through store; the consumer polls the granules themselves until
// both carry the expected tag.  Tags are > 0 and increase by one per exchange; the buffers are zeroed per launch.
__device__ __forceinline__ void gstore_d(u64* g, unsigned tag, double v) {
    const u64 b = (u64)__double_as_longlong(v);
    __hip_atomic_store(g, ((u64)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((u64)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool gload_d(const u64* g, unsigned tag, double& v) {
    for (unsigned spin = 0; spin < (1u << 22); ++spin) {
        const u64 a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag) {
            v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    v = 0;
    return false;
}

// Grid barrier over the G co-resident workgroups: arrive on a monotonic counter, poll relaxed, bounded spin.
// Every cross-workgroup datum is written with xstore (write-through) BEFORE and read with xload AFTER it, so no
// release / acquire fence (L2 write-back / L1 invalidate) is needed.
__device__ bool grid_barrier(const BaDev& B, unsigned& epoch, int* sFlag) {
    if (B.G == 1) {
        __syncthreads();
        return true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(B.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)B.G;
        int ok = 0;
        for (unsigned spin = 0; spin < (1u << 24); ++spin) {
            if (__hip_atomic_load(B.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        sFlag[1] = ok;
    }
    __syncthreads();
    return sFlag[1] != 0;
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


__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// Solves the reduced n x n system [S | g] (row stride ld, in LDS) with ONE wave: lane i keeps row i in registers
// (padded with identity rows up to NR), Gaussian elimination without pivoting = the LDL^T of the SPD system;
// pivot rows are broadcast with v_readlane.  Returns 0 when a pivot is not positive (g2o: LDLT "not positive"
// -> the step is rejected).  x (n entries) is written to xout.
template <int NR>
__device__ __forceinline__ int solve_rows_in_regs(const double* S, int n, int ld, int lane, double* xout) {
    double a[NR + 1];
#pragma unroll
    for (int k = 0; k < NR; ++k) a[k] = (lane < n && k < n) ? S[lane * ld + k] : (k == lane ? 1.0 : 0.0);
    a[NR] = lane < n ? S[lane * ld + n] : 0.0;
    int ok = 1;
    double rinv = 0;  // lane j keeps 1 / pivot_j for the back-substitution
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double d = readlane_d(a[j], j);
        ok &= (d > 0) && isfinite(d);
        const double r = 1.0 / d;  // uniform
        rinv = lane == j ? r : rinv;
        const double l = lane > j ? a[j] * r : 0.0;
#pragma unroll
        for (int k = j + 1; k <= NR; ++k) a[k] -= l * readlane_d(a[k], j);
        __builtin_amdgcn_sched_barrier(0);  // no hoisting of the next step's readlanes (SGPR pressure -> spills)
    }
    double x = 0;
    a[NR] *= rinv;  // z = D^-1 y
#pragma unroll
    for (int j = NR - 1; j >= 0; --j) {
        const double xj = readlane_d(a[NR], j);
        x = lane == j ? xj : x;
        a[NR] -= (a[j] * rinv) * xj;  // rows i < j: U[i][j] / d_i; rows >= j are not read again
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < n) xout[lane] = x;
    return ok;
}


// element `idx` of the packed order "lower triangle row by row (i >= j), then the rhs column" -> (i, j)
__device__ __forceinline__ void packed_ij(int idx, int n, int nlow, int& i, int& j) {
    if (idx < nlow - n) {
        i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > idx) --i;
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        j = idx - i * (i + 1) / 2;
    } else {
        i = idx - (nlow - n);
        j = n;
    }
}
// where G[i][j] (j <= i, or j == n) lives inside the upper-triangle tile set
__device__ __forceinline__ int tile_offset(int i, int j, int n, int NT) {
    int a = j, b = i;
    if (j == n) {
        a = i;
        b = n;
    }
    const int ti = a / 16, tj = b / 16;
    const int tl = ti * NT - ti * (ti - 1) / 2 + (tj - ti);
    return tl * 256 + 16 * (a % 16) + (b % 16);
}

// the 64-row instantiation needs ~130 VGPRs of its own: kept out of line so that it does not inflate the register
// pressure of the common (<= 5 free poses) path
__device__ __noinline__ int solve_rows_64(const double* S, int n, int ld, int lane, double* xout) {
    return solve_rows_in_regs<64>(S, n, ld, lane, xout);
}

// LDS layout of one workgroup, carved from the dynamic segment.
struct WgLds {
    double* S;     // n x (n+1) reduced system
    double* M;     // maxEg x 14
    double* uv;    // maxEg x 2
    double* pts;   // maxLg x 3
    double* X;     // maxEg x 6   X~ = sqrt(rho') Lc J_point
    double* Y;     // maxEg x 6   Y  = X~ C
    double* bak;   // maxLg x 3
    double* Hll;   // maxLg x 6
    double* bl;    // maxLg x 3
    double* Cc;    // maxLg x 6   Cholesky factor of (H_ll + lambda I)^-1
    double* cl;    // maxLg x 3   C^T b_l
    double* tile;  // 4 x ntile x 256 (slot 0 = result)
    double* te;    // maxEg x 3   per-edge back-substitution terms
    short* epose;  // maxEg
    short* ept;    // maxEg  local landmark index
    short* dup;    // maxEg  next local edge with the same (landmark, pose)
    short* ptl;    // maxEg  local edge indices grouped by landmark
    short* pts0;   // maxLg + 1 offsets into ptl
    short* eof;    // maxLg x nfree
};
// tile area: reduction-tree slots, reused as the stage-1 staging buffer (<= nlow + BA_MAX_WGS doubles)
__host__ __device__ inline size_t ba_tile_doubles(int n, int ntile) {
    size_t a = (size_t)ntile * 256 * BA_TILE_SLOTS(ntile), b = (size_t)n * (n + 1) / 2 + n + BA_MAX_WGS;
    return a > b ? a : b;
}
__host__ __device__ inline size_t wg_lds_doubles(int n, int ntile, int maxEg, int maxLg, int fix_points) {
    size_t d = (size_t)n * (n + 1) + (size_t)maxEg * (BA_MSTRIDE + 2) + (size_t)maxLg * 3;
    if (!fix_points) d += (size_t)maxEg * 15 + (size_t)maxLg * (3 + 6 + 3 + 6 + 3) + ba_tile_doubles(n, ntile);
    return d;
}
__host__ __device__ inline size_t wg_lds_bytes(int n, int ntile, int nfree, int maxEg, int maxLg, int fix_points) {
    size_t shorts = (size_t)maxEg * 4 + (size_t)maxLg + 1 + (size_t)maxLg * (nfree > 0 ? nfree : 1);
    return wg_lds_doubles(n, ntile, maxEg, maxLg, fix_points) * 8 + ((shorts * 2 + 15) & ~(size_t)15) + 64;
}

// EdgeProjectXYZ2UV::computeError with the information factor applied: returns chi2, fills the whitened error
__device__ __forceinline__ double edge_error(const BaDev& B, const WgLds& W, int el, const double* sR,
                                             const double* sT, double* Xc, double* ew) {
    const int p = W.epose[el], l = W.ept[el];
    const double* R = sR + 9 * p;
    const double* t = sT + 3 * p;
    const double X0 = W.pts[3 * l], X1 = W.pts[3 * l + 1], X2 = W.pts[3 * l + 2];
    Xc[0] = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0];
    Xc[1] = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
    Xc[2] = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
    const double e0 = W.uv[2 * el] - (Xc[0] / Xc[2] * B.f + B.cx);
    const double e1 = W.uv[2 * el + 1] - (Xc[1] / Xc[2] * B.f + B.cy);
    ew[0] = B.lc00 * e0 + B.lc01 * e1;
    ew[1] = B.lc11 * e1;
    return ew[0] * ew[0] + ew[1] * ew[1];
}

__device__ double robust_chi2_local(const BaDev& B, const WgLds& W, int Eg, const double* sR, const double* sT,
                                    double* scratch) {
    double s = 0;
    for (int el = threadIdx.x; el < Eg; el += BA_THREADS) {
        double Xc[3], ew[2], r0, r1;
        huber(edge_error(B, W, el, sR, sT, Xc, ew), B.delta, r0, r1);
        s += r0;
    }
    return block_sum(s, scratch);
}

// row `row` of U_l (landmark l, k-slot k < 3): pose slot row/6, component row%6 -> sum over the (landmark, pose)
// edges of A~^T Y; row n is C_l^T b_l (the rhs column); rows beyond are zero padding.
__device__ __forceinline__ double u_entry(const WgLds& W, int nfree, int n, int l, int row, int k) {
    if (row < n) {
        const int sl = row / 6, c = row - 6 * sl;
        int el = W.eof[l * nfree + sl];
        double v = 0;
        while (el >= 0) {
            v += W.M[BA_MSTRIDE * el + c] * W.Y[6 * el + k] + W.M[BA_MSTRIDE * el + 7 + c] * W.Y[6 * el + 3 + k];
            el = W.dup[el];
        }
        return v;
    }
    return row == n ? W.cl[3 * l + k] : 0.0;
}

// branch-free variant for windows without duplicate (landmark, pose) observations (the normal case): every load
// is unconditional (clamped index), the selection happens on values
__device__ __forceinline__ double u_entry_nodup(const WgLds& W, int nfree, int l, bool valid, bool is_rhs, int sl, int c,
                                                int k) {
    const int el = W.eof[l * nfree + sl];
    const int e = el < 0 ? 0 : el;
    const double m0 = W.M[BA_MSTRIDE * e + c], m1 = W.M[BA_MSTRIDE * e + 7 + c];
    const double y0 = W.Y[6 * e + k], y1 = W.Y[6 * e + 3 + k];
    const double clv = W.cl[3 * l + k];
    double v = m0 * y0 + m1 * y1;
    v = (valid && el >= 0) ? v : 0.0;
    return is_rhs ? clv : v;
}

// partial G = U^T U over the own landmarks: every wave takes every 8th landmark, one MFMA per landmark and
// tile pair (k-slots 0..2 = columns of U_l, slot 3 = 0); the waves' accumulators are combined in wave order.
template <int NT>
__device__ void schur_mfma(const BaDev& B, const WgLds& W, int Lg, int lane, int wave, long long* ph) {
    long long ph_t = (long long)__builtin_amdgcn_s_memtime();
    constexpr int NPAIR = NT * (NT + 1) / 2;
    v4d acc[NPAIR];
#pragma unroll
    for (int a = 0; a < NPAIR; ++a) acc[a] = (v4d){0, 0, 0, 0};
    const int k = lane >> 4, i = lane & 15;
    const int kk = k < 3 ? k : 0;
    if (!B.has_dups) {
        int sl[NT], cc[NT];
        bool valid[NT], rhs[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = 16 * t + i;
            valid[t] = row < B.n && k < 3;
            rhs[t] = row == B.n && k < 3;
            sl[t] = row < B.n ? row / 6 : 0;
            cc[t] = row < B.n ? row - 6 * sl[t] : 0;
        }
        // four landmarks per step: their LDS chains are independent and overlap
        constexpr int UL = 4;
        for (int l0 = wave; l0 < Lg; l0 += UL * BA_WAVES) {
            double op[UL][NT];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int l = l0 + u * BA_WAVES;
                const bool has = l < Lg;
                const int lc = has ? l : l0;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    op[u][t] = u_entry_nodup(W, B.nfree, lc, valid[t] && has, rhs[t] && has, sl[t], cc[t], kk);
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                int a = 0;
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = ti; tj < NT; ++tj, ++a)
                        acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[u][ti], op[u][tj], acc[a], 0, 0, 0);
            }
        }
    } else {
        for (int l = wave; l < Lg; l += BA_WAVES) {
            double op[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) op[t] = k < 3 ? u_entry(W, B.nfree, B.n, l, 16 * t + i, k) : 0.0;
            int a = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj, ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[ti], op[tj], acc[a], 0, 0, 0);
        }
    }
    PH_END(12);
    __syncthreads();
    PH_END(13);
    const int base = 16 * (lane >> 4) + (lane & 15);
    if (BA_TILE_SLOTS(NPAIR) == 4) {
        // fixed reduction tree over the 8 waves through NPAIR*256-double slots: (0..3) += (4..7), (0,1) += (2,3), 0 += 1
#pragma unroll
        for (int half = BA_WAVES / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
                double* slot = W.tile + (size_t)(wave - half) * NPAIR * 256;
#pragma unroll
                for (int a = 0; a < NPAIR; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) slot[a * 256 + base + 64 * j] = acc[a][j];
            }
            __syncthreads();
            if (wave < half) {
                const double* slot = W.tile + (size_t)wave * NPAIR * 256;
#pragma unroll
                for (int a = 0; a < NPAIR; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a][j] += slot[a * 256 + base + 64 * j];
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < NPAIR; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) W.tile[a * 256 + base + 64 * j] = acc[a][j];
        }
        __syncthreads();
    } else {  // one slot: the waves add their accumulators one after the other (wave order)
        for (int w = 0; w < BA_WAVES; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < NPAIR; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int idx = a * 256 + base + 64 * j;
                        W.tile[idx] = (w == 0 ? 0.0 : W.tile[idx]) + acc[a][j];
                    }
            }
            __syncthreads();
        }
    }
    PH_END(14);
}

// the same sums with plain loops (validation path, and windows with more than 10 free poses)
__device__ void schur_valu(const BaDev& B, const WgLds& W, int Lg) {
    for (int idx = threadIdx.x; idx < B.ntile * 256; idx += BA_THREADS) {
        const int tl = idx / 256, r = (idx % 256) / 16, c = idx % 16;
        int ti = 0, rem = tl;
        while (rem >= B.NT - ti) {
            rem -= B.NT - ti;
            ++ti;
        }
        const int tj = ti + rem, ra = 16 * ti + r, rb = 16 * tj + c;
        double s = 0;
        if (ra <= B.n && rb <= B.n)
            for (int l = 0; l < Lg; ++l)
                for (int k = 0; k < 3; ++k) s += u_entry(W, B.nfree, B.n, l, ra, k) * u_entry(W, B.nfree, B.n, l, rb, k);
        W.tile[idx] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(BA_THREADS) void k_ba_lm(const BaDev* __restrict__ Bp) {
    const BaDev& B = *Bp;
    for (unsigned i = blockIdx.x * BA_THREADS + threadIdx.x; i < B.zero_words; i += gridDim.x * BA_THREADS)
        B.zero_other[i] = 0;  // the other launch parity's exchange region (see BaDev::zero_other)
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double sP[BA_MAX_POSES * 8], sPbak[BA_MAX_POSES * 8];  // q[4] t[3] pad
    __shared__ double sR[BA_MAX_POSES * 9], sT[BA_MAX_POSES * 3];
    __shared__ double sHpp[BA_MAX_POSES * 36], sBp[BA_MAX_POSES * 6], sDx[BA_MAX_POSES * 6];
    __shared__ double sScr[BA_WAVES];
    __shared__ double sLcol[6 * BA_MAX_POSES], sCol[6 * BA_MAX_POSES], sSol[6 * BA_MAX_POSES];
    __shared__ double sX[BA_MAX_WGS * 2];
    __shared__ int sSlot[BA_MAX_POSES], sSlotPose[BA_MAX_POSES], sPoseStart[BA_MAX_POSES + 1];
    __shared__ int sFlag[4];
    if (threadIdx.x == 0) sFlag[2] = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int n = B.n, ld = n + 1;
    const int pt_lo = B.wg_pt_start[g], Lg = B.wg_pt_start[g + 1] - pt_lo;
    const int e_lo = B.wg_edge_start[g], Eg = B.wg_edge_start[g + 1] - e_lo;
    unsigned epoch = 0, tagA = 0, tagB = 0;

    // ---- carve the dynamic LDS
    WgLds W;
    {
        double* d = dyn;
        W.S = d;
        d += (size_t)n * (n + 1);
        W.M = d;
        d += (size_t)B.maxEg * BA_MSTRIDE;
        W.uv = d;
        d += (size_t)B.maxEg * 2;
        W.pts = d;
        d += (size_t)B.maxLg * 3;
        W.X = W.Y = W.bak = W.Hll = W.bl = W.Cc = W.cl = W.tile = W.te = nullptr;
        if (!B.fix_points) {
            W.X = d;
            d += (size_t)B.maxEg * 6;
            W.Y = d;
            d += (size_t)B.maxEg * 6;
            W.bak = d;
            d += (size_t)B.maxLg * 3;
            W.Hll = d;
            d += (size_t)B.maxLg * 6;
            W.bl = d;
            d += (size_t)B.maxLg * 3;
            W.Cc = d;
            d += (size_t)B.maxLg * 6;
            W.cl = d;
            d += (size_t)B.maxLg * 3;
            W.tile = d;
            d += ba_tile_doubles(n, B.ntile);  // slot 0 holds the result
            W.te = d;
            d += (size_t)B.maxEg * 3;
        }
        short* s = reinterpret_cast<short*>(d);
        W.epose = s;
        s += B.maxEg;
        W.ept = s;
        s += B.maxEg;
        W.dup = s;
        s += B.maxEg;
        W.ptl = s;
        s += B.maxEg;
        W.pts0 = s;
        s += B.maxLg + 1;
        W.eof = s;
    }
    // ---- load the workgroup's slice: edges, landmarks, adjacency; every workgroup holds all poses
    for (int el = tid; el < Eg; el += BA_THREADS) {
        W.epose[el] = (short)B.e_pose[e_lo + el];
        W.ept[el] = (short)(B.e_point[e_lo + el] - pt_lo);
        W.dup[el] = B.dup_next[e_lo + el];
        W.ptl[el] = (short)(B.pt_edge_list[e_lo + el] - e_lo);
        W.uv[2 * el] = B.e_uv[2 * (size_t)(e_lo + el)];
        W.uv[2 * el + 1] = B.e_uv[2 * (size_t)(e_lo + el) + 1];
    }
    for (int i = tid; i < 3 * Lg; i += BA_THREADS) W.pts[i] = B.pts_in[3 * (size_t)pt_lo + i];
    for (int i = tid; i <= Lg; i += BA_THREADS) W.pts0[i] = (short)(B.pt_edge_start[pt_lo + i] - e_lo);
    for (int i = tid; i < Lg * B.nfree; i += BA_THREADS) W.eof[i] = B.eof[(size_t)pt_lo * B.nfree + i];
    if (tid < B.F) {  // T_w_c.inv() -> SE3Quat(R, t)  (g2o_ba.cpp:185-190, 208-215)
        double Ri[9], ti[3], q[4];
        invert_Rt(B.poses_in + 16 * tid, Ri, ti);
        quat_from_R(Ri, q);
        quat_normalize(q);
        for (int i = 0; i < 4; ++i) sP[8 * tid + i] = q[i];
        for (int i = 0; i < 3; ++i) sP[8 * tid + 4 + i] = ti[i];
        quat_to_R(q, sR + 9 * tid);
        for (int i = 0; i < 3; ++i) sT[3 * tid + i] = ti[i];
        sSlot[tid] = B.pose_slot[tid];
    }
    if (tid < B.nfree) sSlotPose[tid] = B.slot_pose[tid];
    if (tid <= B.F) sPoseStart[tid] = B.wg_pose_start[g * (B.F + 1) + tid] - e_lo;
    __syncthreads();

    double lambda = 0, ni = 2;
    int it = 0, trials = 0, terminated = 0, error = 0;
    long long ph[BA_NPHASE] = {0};
    const long long ph_start = (long long)__builtin_amdgcn_s_memtime();
    // ---- initial robust chi2 (all workgroups)
    double currentChi;
    {
        double c = robust_chi2_local(B, W, Eg, sR, sT, sScr);
        if (B.G > 1) {
            if (tid == 0) xstore(B.xSc + 4 * g, c);
            if (!grid_barrier(B, epoch, sFlag)) error = 1;
            c = 0;
            for (int w = 0; w < B.G; ++w) c += xload(B.xSc + 4 * w);
            if (!grid_barrier(B, epoch, sFlag)) error = 1;  // everybody has read slot 0
        }
        currentChi = c;
    }
    const double chi0 = currentChi;
    const bool any_free = B.nfree > 0 || !B.fix_points;
    const bool do_schur = !B.fix_points && n > 0;
    const int nlow = n * (n + 1) / 2 + n;  // packed lower triangle + rhs column
    const int npk = (nlow + 15) & ~15;     // row pitch of the packed partials
    const int slice = (nlow + B.G - 1) / B.G;  // entries each workgroup reduces in stage 1

    for (it = 0; any_free && !error && it < B.max_it; ++it) {
        // ================= LIN: whitened Jacobians of the own edges (EdgeProjectXYZ2UV::linearizeOplus)
        PH_BEGIN();
        for (int el = tid; el < Eg; el += BA_THREADS) {
            double Xc[3], ew[2], r0, r1;
            const double chi = edge_error(B, W, el, sR, sT, Xc, ew);
            huber(chi, B.delta, r0, r1);
            const double sw = sqrt(r1);
            const int p = W.epose[el];
            const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = B.f;
            double* Mr = W.M + BA_MSTRIDE * el;
            if (sSlot[p] >= 0) {
                const double J0[6] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f};
                const double J1[6] = {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    Mr[c] = sw * (B.lc00 * J0[c] + B.lc01 * J1[c]);
                    Mr[7 + c] = sw * (B.lc11 * J1[c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) Mr[c] = Mr[7 + c] = 0;
            }
            Mr[6] = sw * ew[0];
            Mr[13] = sw * ew[1];
            if (!B.fix_points) {
                const double* R = sR + 9 * p;
                const double t0[3] = {f, 0, -x / z * f}, t1[3] = {0, f, -y / z * f};
                double* Xr = W.X + 6 * el;
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
        PH_END(0);
        // ================= HPP: partial [H_pp | -b_p] = M^T M over the own edges of every free pose.
        // Matrix-core path: wave w owns the poses p = w, w + 8, ... and accumulates each of them alone (no
        // cross-wave reduction, one barrier for the whole phase).
        if (B.use_mfma) {
            const int col = lane & 15;
            for (int p = wave; p < B.F; p += BA_WAVES) {
                if (sSlot[p] < 0) continue;
                const int s = sPoseStart[p], e = sPoseStart[p + 1];
                const int steps = (2 * (e - s) + 3) / 4;
                v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};  // two chains hide the MFMA latency
                for (int st = 0; st < steps; st += 2) {
                    const int row0 = 4 * st + (lane >> 4), row1 = row0 + 4;
                    const int ed0 = s + (row0 >> 1), ed1 = s + (row1 >> 1);
                    const double v0 = (ed0 < e && col < 7) ? W.M[BA_MSTRIDE * ed0 + 7 * (row0 & 1) + col] : 0.0;
                    const double v1 = (ed1 < e && col < 7) ? W.M[BA_MSTRIDE * ed1 + 7 * (row1 & 1) + col] : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rg = (lane >> 4) + 4 * j;
                    if (rg < 7 && col < 7) {
                        const double gsum = acc0[j] + acc1[j];
                        if (B.G > 1) {
                            xstore(B.xHpp + ((size_t)g * B.F + p) * 49 + rg * 7 + col, gsum);
                        } else {
                            if (rg < 6 && col < 6) sHpp[36 * p + 6 * rg + col] = gsum;
                            if (rg < 6 && col == 6) sBp[6 * p + rg] = -gsum;
                        }
                    }
                }
            }
        } else {
            for (int p = 0; p < B.F; ++p) {
                if (sSlot[p] < 0) continue;
                const int s = sPoseStart[p], e = sPoseStart[p + 1];
                if (tid < 49) {
                    const int i = tid / 7, j = tid % 7;
                    double gsum = 0;
                    for (int r = 2 * s; r < 2 * e; ++r) gsum += W.M[7 * r + i] * W.M[7 * r + j];
                    if (B.G > 1) {
                        xstore(B.xHpp + ((size_t)g * B.F + p) * 49 + tid, gsum);
                    } else {
                        if (i < 6 && j < 6) sHpp[36 * p + 6 * i + j] = gsum;
                        if (i < 6 && j == 6) sBp[6 * p + i] = -gsum;
                    }
                }
            }
        }
        __syncthreads();
        PH_END(1);
        // ================= PT: 3x3 landmark blocks H_ll, b_l of the own landmarks
        double maxdiag = 0;
        if (!B.fix_points) {
            for (int l = tid; l < Lg; l += BA_THREADS) {
                double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
                for (int k = W.pts0[l]; k < W.pts0[l + 1]; ++k) {
                    const int el = W.ptl[k];
                    const double* X = W.X + 6 * el;
                    const double e0 = W.M[BA_MSTRIDE * el + 6], e1 = W.M[BA_MSTRIDE * el + 13];
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
                for (int i = 0; i < 6; ++i) W.Hll[6 * l + i] = h[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) W.bl[3 * l + i] = b[i];
                maxdiag = fmax(maxdiag, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))));
            }
        }
        if (it == 0 && B.G > 1) {
            double m = block_max(maxdiag, sScr);
            if (tid == 0) xstore(B.xSc + 4 * g + 2, m);
        }
        // ---- exchange: pose-block partials (+ the landmark max diagonal at iteration 0)
        if (B.G > 1) {
            if (!grid_barrier(B, epoch, sFlag)) error = 1;
            if (tid < 49 * B.F) {
                const int p = tid / 49, r = tid % 49;
                if (sSlot[p] >= 0) {
                    double gsum = 0;
                    const double* src = B.xHpp + (size_t)p * 49 + r;
                    const size_t stride = (size_t)B.F * 49;
                    for (int w0 = 0; w0 < B.G; w0 += 16) {
                        double v[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) v[q] = (w0 + q < B.G) ? xload(src + (size_t)(w0 + q) * stride) : 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) gsum += v[q];
                    }
                    const int i = r / 7, j = r % 7;
                    if (i < 6 && j < 6) sHpp[36 * p + 6 * i + j] = gsum;
                    if (i < 6 && j == 6) sBp[6 * p + i] = -gsum;
                }
            }
            if (it == 0) {
                maxdiag = tid < B.G ? xload(B.xSc + 4 * tid + 2) : 0.0;  // block_max below combines them
            }
            __syncthreads();
        }
        if (it == 0) {  // computeLambdaInit: tau * max |diag H| over the free vertices
            double m = maxdiag;
            if (tid < 6 * B.F && sSlot[tid / 6] >= 0) m = fmax(m, fabs(sHpp[36 * (tid / 6) + 7 * (tid % 6)]));
            lambda = 1e-5 * block_max(m, sScr);
            ni = 2;
        }
        __syncthreads();
        PH_END(2);

        double rho = 0;
        int qmax = 0;
        do {
            // ============= T1: (H_ll + lambda I)^-1 = C C^T, C^T b_l; Y = X~ C for the own edges
            if (!B.fix_points) {
                for (int l = tid; l < Lg; l += BA_THREADS) {
                    const double* h = W.Hll + 6 * l;
                    const double D[9] = {h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda};
                    double Di[9];
                    inv3(D, Di);
                    const double c00 = sqrt(Di[0]), c10 = Di[3] / c00, c20 = Di[6] / c00;
                    const double c11 = sqrt(Di[4] - c10 * c10), c21 = (Di[7] - c20 * c10) / c11;
                    const double c22 = sqrt(Di[8] - c20 * c20 - c21 * c21);
                    double* cc = W.Cc + 6 * l;
                    cc[0] = c00;
                    cc[1] = c10;
                    cc[2] = c11;
                    cc[3] = c20;
                    cc[4] = c21;
                    cc[5] = c22;
                    const double* b = W.bl + 3 * l;
                    W.cl[3 * l] = c00 * b[0] + c10 * b[1] + c20 * b[2];
                    W.cl[3 * l + 1] = c11 * b[1] + c21 * b[2];
                    W.cl[3 * l + 2] = c22 * b[2];
                }
                __syncthreads();
                for (int el = tid; el < Eg; el += BA_THREADS) {
                    const double* X = W.X + 6 * el;
                    const double* cc = W.Cc + 6 * W.ept[el];
                    double* Yr = W.Y + 6 * el;
                    Yr[0] = X[0] * cc[0] + X[1] * cc[1] + X[2] * cc[3];
                    Yr[1] = X[1] * cc[2] + X[2] * cc[4];
                    Yr[2] = X[2] * cc[5];
                    Yr[3] = X[3] * cc[0] + X[4] * cc[1] + X[5] * cc[3];
                    Yr[4] = X[4] * cc[2] + X[5] * cc[4];
                    Yr[5] = X[5] * cc[5];
                }
                __syncthreads();
            }
            PH_END(3);
            // ============= T2: partial Schur blocks of the own landmarks, published for the other workgroups
            if (do_schur) {
                if (B.use_mfma && B.NT <= BA_MFMA_MAX_NT) {
                    if (B.NT == 1) schur_mfma<1>(B, W, Lg, lane, wave, ph);
                    else if (B.NT == 2) schur_mfma<2>(B, W, Lg, lane, wave, ph);
                    else if (B.NT == 3) schur_mfma<3>(B, W, Lg, lane, wave, ph);
                    else schur_mfma<4>(B, W, Lg, lane, wave, ph);
                } else {
                    schur_valu(B, W, Lg);
                }
                PH_END(4);
                if (B.G > 1) {
                    ++tagA;
                    // publish the needed entries (lower triangle + rhs) in packed order
                    for (int idx = tid; idx < nlow; idx += BA_THREADS) {
                        int i, j;
                        packed_ij(idx, n, nlow, i, j);
                        gstore_d(B.xGg + 2 * ((size_t)g * npk + idx), tagA, W.tile[tile_offset(i, j, n, B.NT)]);
                    }
                    __syncthreads();  // the tile is reused as staging buffer below
                    // stage 1 of the cross-workgroup sum: this workgroup reduces its SLICE of the packed entries
                    // over all G partials (one load per thread), in workgroup order, and republishes the slice
                    const int sl0 = g * slice, sln = max(0, min(slice, nlow - sl0));
                    double* red = W.tile;  // the published tile is no longer needed
                    for (int q = tid; q < sln * B.G; q += BA_THREADS) {
                        const int w = q / sln, el = q - w * sln;
                        if (!gload_d(B.xGg + 2 * ((size_t)w * npk + sl0 + el), tagA, red[q])) sFlag[2] = 1;
                    }
                    __syncthreads();
                    for (int el = tid; el < sln; el += BA_THREADS) {
                        double sum = 0;
                        for (int w = 0; w < B.G; ++w) sum += red[w * sln + el];
                        gstore_d(B.xRg + 2 * (size_t)(sl0 + el), tagA, sum);
                    }
                    // no barrier: the consumers below poll the tagged granules of the entries they need
                }
            }
            PH_END(5);
            // ============= T3: every workgroup assembles S = H_pp + lambda I - G, g = b_p - G[:, n]; LDL^T by wave 0
            for (int idx = tid; idx < nlow; idx += BA_THREADS) {
                int i, j;
                packed_ij(idx, n, nlow, i, j);
                double gsum = 0;
                if (do_schur) {
                    if (B.G > 1) {
                        if (!gload_d(B.xRg + 2 * (size_t)idx, tagA, gsum)) sFlag[2] = 1;
                    } else {
                        gsum = W.tile[tile_offset(i, j, n, B.NT)];
                    }
                }
                const int pi = sSlotPose[i / 6];
                if (j == n) {
                    W.S[i * ld + n] = sBp[6 * pi + i % 6] - gsum;
                } else {
                    const int pj = sSlotPose[j / 6];
                    const double v = ((pi == pj) ? sHpp[36 * pi + 6 * (i % 6) + (j % 6)] + (i == j ? lambda : 0.0) : 0.0) - gsum;
                    W.S[i * ld + j] = v;
                    W.S[j * ld + i] = v;
                }
            }
            __syncthreads();
            if (sFlag[2]) error = 1;
            PH_END(6);
            if (wave == 0) {
                int ok = 1;
                if (n <= 32) {
                    ok = solve_rows_in_regs<32>(W.S, n, ld, lane, sSol);
                } else if (n <= 64) {
                    ok = solve_rows_64(W.S, n, ld, lane, sSol);
                } else {  // more than 10 free poses: unpivoted LDL^T in LDS
                    double* S = W.S;
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
                        for (int i = lane; i < n; i += 64) sSol[i] = S[i * ld + n];
                    }
                }
                if (lane == 0) sFlag[0] = ok;
            }
            __syncthreads();
            const int ok2 = sFlag[0];
            if (tid < 6 * B.F) {
                const int sl = sSlot[tid / 6];
                sDx[tid] = (ok2 && sl >= 0) ? sSol[6 * sl + tid % 6] : 0.0;
            }
            __syncthreads();
            PH_END(7);
            ++trials;
            // ============= T4/T5: back-substitute the own landmarks, computeScale, push + apply the update
            double scale = 0;
            if (g == 0 && tid < 6 * B.F && sSlot[tid / 6] >= 0) scale += sDx[tid] * (lambda * sDx[tid] + sBp[tid]);
            if (!B.fix_points) {
                // per edge (all threads): t_e = Y_e^T (A~_e dx_p), parked in the first 3 slots of X's row? no:
                // X~ is needed by the next trial -> use the rhs slots of S that the solver no longer needs
                for (int el = tid; el < Eg; el += BA_THREADS) {
                    const int p = W.epose[el];
                    const double* A = W.M + BA_MSTRIDE * el;
                    const double* Yr = W.Y + 6 * el;
                    double a0 = 0, a1 = 0;  // A~ dx_p (zero rows for a fixed pose)
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        a0 += A[i] * sDx[6 * p + i];
                        a1 += A[7 + i] * sDx[6 * p + i];
                    }
                    double* te = W.te + 3 * el;
#pragma unroll
                    for (int c = 0; c < 3; ++c) te[c] = Yr[c] * a0 + Yr[3 + c] * a1;
                }
                __syncthreads();
                for (int l = tid; l < Lg; l += BA_THREADS) {
                    double r[3] = {W.cl[3 * l], W.cl[3 * l + 1], W.cl[3 * l + 2]};  // C^T (b_l - sum W^T dx_p)
                    for (int k = W.pts0[l]; k < W.pts0[l + 1]; ++k) {
                        const double* te = W.te + 3 * W.ptl[k];
#pragma unroll
                        for (int c = 0; c < 3; ++c) r[c] -= te[c];
                    }
                    const double* cc = W.Cc + 6 * l;
                    double d[3] = {cc[0] * r[0], cc[1] * r[0] + cc[2] * r[1], cc[3] * r[0] + cc[4] * r[1] + cc[5] * r[2]};
                    if (!ok2) d[0] = d[1] = d[2] = 0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        scale += d[c] * (lambda * d[c] + W.bl[3 * l + c]);
                        W.bak[3 * l + c] = W.pts[3 * l + c];
                        W.pts[3 * l + c] += d[c];
                    }
                }
            }
            if (tid < B.F) {
                for (int i = 0; i < 8; ++i) sPbak[8 * tid + i] = sP[8 * tid + i];
                if (sSlot[tid] >= 0) {
                    pose_oplus(sP + 8 * tid, sDx + 6 * tid);
                    quat_to_R(sP + 8 * tid, sR + 9 * tid);
                    for (int i = 0; i < 3; ++i) sT[3 * tid + i] = sP[8 * tid + 4 + i];
                }
            }
            scale = block_sum(scale, sScr);
            __syncthreads();
            PH_END(8);
            // ============= T6/T7: robust chi2 at the trial state, identical accept / reject decision everywhere
            double tempChi = robust_chi2_local(B, W, Eg, sR, sT, sScr);
            PH_END(9);
            if (B.G > 1) {
                ++tagB;
                u64* slot = B.xCg + (size_t)(tagB & 1) * B.G * 4;
                if (tid == 0) {
                    gstore_d(slot + 4 * g, tagB, tempChi);
                    gstore_d(slot + 4 * g + 2, tagB, scale);
                }
                // every workgroup waits for the tagged partials of ALL workgroups: this is also the barrier that
                // keeps a fast workgroup from overwriting exchange buffers a slow one still reads (a workgroup can
                // be at most one chi2 exchange ahead, hence the two parity slots)
                if (tid < B.G) {
                    double c = 0, sc = 0;
                    if (!gload_d(slot + 4 * tid, tagB, c) || !gload_d(slot + 4 * tid + 2, tagB, sc)) sFlag[2] = 1;
                    sX[2 * tid] = c;
                    sX[2 * tid + 1] = sc;
                }
                __syncthreads();
                if (sFlag[2]) error = 1;
                tempChi = 0;
                scale = 0;
                for (int w = 0; w < B.G; ++w) {
                    tempChi += sX[2 * w];
                    scale += sX[2 * w + 1];
                }
            }
            scale += 1e-3;
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
                    for (int i = tid; i < 3 * Lg; i += BA_THREADS) W.pts[i] = W.bak[i];
            }
            __syncthreads();
            PH_END(10);
            ++qmax;
        } while (rho < 0 && qmax < 10 && !error);
        if (qmax == 10 || rho == 0 || error) {
            terminated = 1;
            ++it;
            break;
        }
        // (the next iteration's xHpp writes are behind this trial's chi2 hand-off, which every workgroup enters only
        // after it has finished reading the current xHpp)
    }
    // ---- write-back (g2o_ba.cpp:298-316): SE3Quat -> (R, t) -> 4x4 -> inverse; landmarks of the own range
    if (g == 0 && tid < B.F) {
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
        if (B.h_poses) {
            double* ho = B.h_poses + 16 * tid;
            for (int k = 0; k < 16; ++k) ho[k] = o[k];
        }
    }
    for (int i = tid; i < 3 * Lg; i += BA_THREADS) B.pts_out[3 * (size_t)pt_lo + i] = W.pts[i];
    if (g == 0 && tid == 0) {
        B.stats->iterations = it;
        B.stats->trials = trials;
        B.stats->terminated = terminated;
        B.stats->error = error;
        B.stats->chi2_initial = chi0;
        B.stats->chi2_final = currentChi;
        B.stats->lambda_final = lambda;
        ph[11] = (long long)__builtin_amdgcn_s_memtime() - ph_start;
        for (int i = 0; i < BA_NPHASE; ++i) B.stats->phase[i] = ph[i];
        if (B.h_stats) *B.h_stats = *B.stats;
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

int g_ba_use_mfma = 1;  // debug knobs (mvo_debug_set)
int g_ba_wgs = 0;       // 0 = automatic

// Co-residency guard: the workgroups of one BA launch meet at grid barriers, so all of them must be resident
// (one per CU: the LDS slice is > 80 KB).  Launches from different ctx / host threads draw their workgroups from
// a per-device budget of CUs and wait (on the host, before launching) while it is exhausted.
#include <condition_variable>
#include <mutex>
namespace {
struct CuBudget {
    std::mutex m;
    std::condition_variable cv;
    int avail[16];
    bool init = false;
} g_budget;
void budget_acquire(int device, int n) {
    std::unique_lock<std::mutex> lk(g_budget.m);
    if (!g_budget.init) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
        for (int d = 0; d < 16; ++d) {
            int cus = 256;
            if (d < ndev && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) cus = 256;
            g_budget.avail[d] = cus;
        }
        (void)hipGetLastError();
        g_budget.init = true;
    }
    device &= 15;
    g_budget.cv.wait(lk, [&] { return g_budget.avail[device] >= n; });
    g_budget.avail[device] -= n;
}
void budget_release(int device, int n) {
    {
        std::lock_guard<std::mutex> lk(g_budget.m);
        g_budget.avail[device & 15] += n;
    }
    g_budget.cv.notify_all();
}
}  // namespace

// A window whose inputs are resident in HBM: upload once (mvo_ba_prepare), solve any number of times.
struct mvo_ba_handle {
    int device = 0;
    int tokens = 0;  // CUs currently reserved for an in-flight launch of this window
    char* dev = nullptr;  // one allocation holding inputs, adjacency tables and exchange buffers
    size_t bytes = 0;
    BaDev B{};
    int F = 0, L = 0;
    size_t o_stats = 0, o_pout = 0, o_pts = 0, o_bar = 0, o_desc = 0, zero_bytes = 64, o_region1 = 0;
    int parity = 0;  // which exchange region / descriptor the next launch uses
    size_t lds = 16;
    bool fix_points = false;
    char* pin = nullptr;       // pinned host memory: BaStatsDev, then F x 16 doubles
    int uploaded_mfma = -1;    // value of B.use_mfma in the device copy of the descriptor
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
    std::vector<int> pose_slot(F, -1), slot_pose;
    for (int i = 0; i < F; ++i)
        if (!(p->pose_fixed && p->pose_fixed[i])) {
            pose_slot[i] = (int)slot_pose.size();
            slot_pose.push_back(i);
        }
    const int nfree = (int)slot_pose.size();
    // active edges: SparseOptimizer::initializeOptimization drops edges whose vertices are all fixed
    std::vector<int> act;
    act.reserve(p->n_edges);
    std::vector<int> deg(L, 0);
    for (int e = 0; e < p->n_edges; ++e) {
        if (pose_slot[p->edge_pose[e]] < 0 && p->fix_points) continue;
        act.push_back(e);
        deg[p->edge_point[e]]++;
    }
    const int E = (int)act.size();
    const int n = 6 * nfree;
    const int NT = (n + 1 + 15) / 16, ntile = NT * (NT + 1) / 2;
    // ---- choose G and the landmark ranges (balanced by edge count); the slice of every workgroup must fit in LDS
    int G = 1;
    while (G < 32 && E > 160 * G) G *= 2;  // aim at 160-320 edges per workgroup
    if (g_ba_wgs > 0) G = g_ba_wgs;
    if (const char* env = std::getenv("MVO_BA_WGS")) G = std::max(1, std::atoi(env));
    G = std::max(1, std::min(G, BA_MAX_WGS));
    std::vector<int> wg_pt, wg_edge;
    int maxEg = 0, maxLg = 0;
    for (;;) {
        wg_pt.assign(G + 1, 0);
        wg_edge.assign(G + 1, 0);
        int l = 0, eacc = 0;
        for (int g = 0; g < G; ++g) {
            wg_pt[g] = l;
            wg_edge[g] = eacc;
            const long target = (long)E * (g + 1) / G;
            while (l < L && (g == G - 1 || eacc < target)) eacc += deg[l++];
        }
        wg_pt[G] = L;
        wg_edge[G] = E;
        maxEg = maxLg = 0;
        for (int g = 0; g < G; ++g) {
            maxEg = std::max(maxEg, wg_edge[g + 1] - wg_edge[g]);
            maxLg = std::max(maxLg, wg_pt[g + 1] - wg_pt[g]);
        }
        if (wg_lds_bytes(n, ntile, nfree, maxEg, maxLg, p->fix_points) <= BA_LDS_BUDGET && maxEg < 32000 && maxLg < 32000)
            break;
        if (G >= BA_MAX_WGS)
            return mvo_set_err(ctx, MVO_ERR_CAPACITY, "BA window too large for the LDS-resident solver", hipSuccess);
        G = std::min(2 * G, BA_MAX_WGS);
    }
    // ---- edges sorted by (owner workgroup, pose); adjacency tables
    std::vector<int> owner(L, 0);
    for (int g = 0; g < G; ++g)
        for (int l = wg_pt[g]; l < wg_pt[g + 1]; ++l) owner[l] = g;
    std::vector<int> wg_pose((size_t)G * (F + 1), 0);
    {
        std::vector<int> cnt((size_t)G * std::max(F, 1), 0);
        for (int e : act) cnt[(size_t)owner[p->edge_point[e]] * F + p->edge_pose[e]]++;
        int acc = 0;
        for (int g = 0; g < G; ++g) {
            for (int q = 0; q < F; ++q) {
                wg_pose[(size_t)g * (F + 1) + q] = acc;
                acc += cnt[(size_t)g * F + q];
            }
            wg_pose[(size_t)g * (F + 1) + F] = acc;
        }
    }
    std::vector<int> e_pose(E), e_point(E), ptstart(L + 1, 0), ptlist(E);
    std::vector<double> e_uv(2 * (size_t)E);
    {
        std::vector<int> cur((size_t)G * std::max(F, 1));
        for (int g = 0; g < G; ++g)
            for (int q = 0; q < F; ++q) cur[(size_t)g * F + q] = wg_pose[(size_t)g * (F + 1) + q];
        for (int e : act) {
            const int k = cur[(size_t)owner[p->edge_point[e]] * F + p->edge_pose[e]]++;
            e_pose[k] = p->edge_pose[e];
            e_point[k] = p->edge_point[e];
            e_uv[2 * (size_t)k] = p->edge_uv[2 * (size_t)e];
            e_uv[2 * (size_t)k + 1] = p->edge_uv[2 * (size_t)e + 1];
        }
    }
    for (int k = 0; k < E; ++k) ptstart[e_point[k] + 1]++;
    for (int i = 0; i < L; ++i) ptstart[i + 1] += ptstart[i];
    {
        std::vector<int> cur(ptstart.begin(), ptstart.end() - 1);
        for (int k = 0; k < E; ++k) ptlist[cur[e_point[k]]++] = k;
    }
    std::vector<short> eof((size_t)std::max(L, 1) * std::max(nfree, 1), -1), dup(std::max(E, 1), -1);
    int has_dups = 0;
    for (int k = E - 1; k >= 0; --k) {  // descending so that the chains run in ascending edge order
        const int sl = pose_slot[e_pose[k]];
        if (sl < 0) continue;
        const int lk = k - wg_edge[owner[e_point[k]]];
        short& head = eof[(size_t)e_point[k] * nfree + sl];
        dup[k] = head;
        if (head >= 0) has_dups = 1;
        head = (short)lk;
    }

    Carver cv;
    const size_t o_stats = cv.take(sizeof(BaStatsDev));
    const size_t o_pin = cv.take((size_t)F * 128), o_ptsin = cv.take((size_t)L * 24);
    const size_t o_wpt = cv.take((size_t)(G + 1) * 4), o_wed = cv.take((size_t)(G + 1) * 4);
    const size_t o_wps = cv.take((size_t)G * (F + 1) * 4);
    const size_t o_ep = cv.take((size_t)E * 4 + 4), o_el = cv.take((size_t)E * 4 + 4), o_uv = cv.take((size_t)E * 16 + 16);
    const size_t o_pts_s = cv.take((size_t)(L + 1) * 4), o_ptl = cv.take((size_t)E * 4 + 4);
    const size_t o_eof = cv.take(eof.size() * 2), o_dup = cv.take(dup.size() * 2);
    const size_t o_slot = cv.take((size_t)F * 4 + 4), o_sp = cv.take((size_t)nfree * 4 + 4);
    const size_t upload_end = cv.off;
    const size_t o_pout = cv.take((size_t)F * 128), o_pts = cv.take((size_t)L * 24);
    const size_t o_xh = cv.take((size_t)G * F * 49 * 8 + 8);

    const size_t o_xs = cv.take((size_t)G * 32), o_desc = cv.take(2 * ((sizeof(BaDev) + 255) & ~(size_t)255));
    // zeroed before every launch: barrier counter | summed-entry granules | chi2 granules (contiguous)
    const size_t npk_h = ((size_t)n * (n + 1) / 2 + n + 15) & ~(size_t)15;
    const size_t o_bar = cv.take(256), o_xr = cv.take(npk_h * 16), o_xc = cv.take((size_t)2 * G * 4 * 8);
    const size_t o_xg = cv.take((size_t)G * npk_h * 16);
    const size_t zero_bytes = cv.off - o_bar;
    const size_t o_region1 = cv.take(zero_bytes);  // the second exchange region (same layout)
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
    if (F) std::memcpy(h + o_pin, p->pose_T_w_c, (size_t)F * 128);
    if (L) std::memcpy(h + o_ptsin, p->points, (size_t)L * 24);
    std::memcpy(h + o_wpt, wg_pt.data(), (size_t)(G + 1) * 4);
    std::memcpy(h + o_wed, wg_edge.data(), (size_t)(G + 1) * 4);
    std::memcpy(h + o_wps, wg_pose.data(), wg_pose.size() * 4);
    if (E) {
        std::memcpy(h + o_ep, e_pose.data(), (size_t)E * 4);
        std::memcpy(h + o_el, e_point.data(), (size_t)E * 4);
        std::memcpy(h + o_uv, e_uv.data(), (size_t)E * 16);
        std::memcpy(h + o_ptl, ptlist.data(), (size_t)E * 4);
    }
    std::memcpy(h + o_pts_s, ptstart.data(), (size_t)(L + 1) * 4);
    std::memcpy(h + o_eof, eof.data(), eof.size() * 2);
    std::memcpy(h + o_dup, dup.data(), dup.size() * 2);
    if (F) std::memcpy(h + o_slot, pose_slot.data(), (size_t)F * 4);
    if (nfree) std::memcpy(h + o_sp, slot_pose.data(), (size_t)nfree * 4);
    char* D = H->dev;
    hipError_t e1 = hipMemcpyAsync(D, h, upload_end, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = hipMemsetAsync(D + upload_end, 0, total - upload_end, ctx->stream);
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
    B.G = G;
    B.nfree = nfree;
    B.n = n;
    B.NT = NT;
    B.ntile = ntile;
    B.fix_points = p->fix_points ? 1 : 0;
    B.max_it = p->max_iterations;
    B.use_mfma = g_ba_use_mfma;
    B.maxEg = maxEg;
    B.maxLg = maxLg;
    B.has_dups = has_dups;
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
    B.pts_out = (double*)(D + o_pts);
    B.wg_pt_start = (const int*)(D + o_wpt);
    B.wg_edge_start = (const int*)(D + o_wed);
    B.wg_pose_start = (const int*)(D + o_wps);
    B.e_pose = (const int*)(D + o_ep);
    B.e_point = (const int*)(D + o_el);
    B.e_uv = (const double*)(D + o_uv);
    B.pt_edge_start = (const int*)(D + o_pts_s);
    B.pt_edge_list = (const int*)(D + o_ptl);
    B.eof = (const short*)(D + o_eof);
    B.dup_next = (const short*)(D + o_dup);
    B.pose_slot = (const int*)(D + o_slot);
    B.slot_pose = (const int*)(D + o_sp);
    B.xHpp = (double*)(D + o_xh);
    B.xGg = (u64*)(D + o_xg);
    B.xRg = (u64*)(D + o_xr);
    B.xCg = (u64*)(D + o_xc);
    B.xSc = (double*)(D + o_xs);
    B.barrier = (unsigned*)(D + o_bar);
    B.stats = (BaStatsDev*)(D + o_stats);
    {
        const size_t pin_bytes = ((sizeof(BaStatsDev) + 255) & ~(size_t)255) + (size_t)std::max(F, 1) * 128;
        hipError_t pe = hipHostMalloc((void**)&H->pin, pin_bytes, hipHostMallocDefault);
        if (pe != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(H->dev);
            delete H;
            return mvo_set_err(ctx, MVO_ERR_HIP, "hipHostMalloc (BA result mirror)", pe);
        }
        std::memset(H->pin, 0, pin_bytes);
        B.h_stats = (BaStatsDev*)H->pin;
        B.h_poses = (double*)(H->pin + ((sizeof(BaStatsDev) + 255) & ~(size_t)255));
    }
    H->F = F;
    H->L = L;
    H->o_stats = o_stats;
    H->o_pout = o_pout;
    H->o_pts = o_pts;
    H->o_bar = o_bar;
    H->o_desc = o_desc;
    H->zero_bytes = zero_bytes;
    H->o_region1 = o_region1;
    B.zero_other = (u64*)(D + o_region1);
    B.zero_words = (unsigned)(zero_bytes / 8);
    // both regions start out clean
    MVO_HIP(hipMemsetAsync(D + o_bar, 0, zero_bytes, ctx->stream));
    MVO_HIP(hipMemsetAsync(D + o_region1, 0, zero_bytes, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    H->fix_points = p->fix_points != 0;
    H->lds = wg_lds_bytes(n, ntile, nfree, maxEg, maxLg, p->fix_points);
    H->device = ctx->device;
    *out = H;
    return MVO_OK;
}

// One full LM solve from the resident initial state; results stay on the device.
int ba_run_device(mvo_ctx* ctx, mvo_ba_handle* H) {
    if (H->F == 0 && (H->L == 0 || H->fix_points)) return MVO_OK;
    H->B.use_mfma = g_ba_use_mfma;
    if (H->B.G > 1 && H->tokens == 0) {
        budget_acquire(H->device, H->B.G);
        H->tokens = H->B.G;
    }
    {
        // raise the dynamic-LDS limit ONCE per device to the solver's budget (a per-launch value would race
        // between host threads launching windows of different sizes)
        static std::mutex attr_mutex;
        static bool attr_done[16] = {false};
        std::lock_guard<std::mutex> lk(attr_mutex);
        if (!attr_done[H->device & 15]) {
            MVO_HIP(hipFuncSetAttribute((const void*)k_ba_lm, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_BUDGET));
            attr_done[H->device & 15] = true;
        }
    }
    const size_t desc_stride = (sizeof(BaDev) + 255) & ~(size_t)255;
    if (H->uploaded_mfma != H->B.use_mfma) {  // the descriptors are constant but for the debug knob: upload them once
        BaDev other = H->B;  // parity 1: the exchange pointers moved into the second region, clears the first
        const ptrdiff_t shift = (ptrdiff_t)H->o_region1 - (ptrdiff_t)H->o_bar;
        other.barrier = (unsigned*)((char*)H->B.barrier + shift);
        other.xRg = (u64*)((char*)H->B.xRg + shift);
        other.xCg = (u64*)((char*)H->B.xCg + shift);
        other.xGg = (u64*)((char*)H->B.xGg + shift);
        other.zero_other = (u64*)(H->dev + H->o_bar);
        MVO_HIP(hipMemcpyAsync(H->dev + H->o_desc, &H->B, sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream));
        MVO_HIP(hipMemcpyAsync(H->dev + H->o_desc + desc_stride, &other, sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream));
        MVO_HIP(hipStreamSynchronize(ctx->stream));  // (pageable sources)
        H->uploaded_mfma = H->B.use_mfma;
    }
    const char* d_desc = H->dev + H->o_desc + (H->parity ? desc_stride : 0);
    H->parity ^= 1;
    {
        ProfScope ps(ctx, "k_ba_lm");
        hipLaunchKernelGGL(k_ba_lm, dim3(H->B.G), dim3(BA_THREADS), H->lds, ctx->stream, (const BaDev*)d_desc);
    }
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int ba_fetch_device(mvo_ctx* ctx, mvo_ba_handle* H, double* poses, double* points, mvo_ba_stats* st) {
    const bool ran = !(H->F == 0 && (H->L == 0 || H->fix_points));
    if (st) std::memset(st, 0, sizeof(*st));
    const bool want_pts = ran && points && H->L && !H->fix_points;
    uint8_t* hx = nullptr;
    if (want_pts) {  // landmarks are the only part that still travels by copy
        int r = mvo_ensure_pinned(ctx, (size_t)H->L * 24 + 256);
        if (r) return r;
        hx = ctx->h_pin;
        MVO_HIP(hipMemcpyAsync(hx, H->dev + H->o_pts, (size_t)H->L * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    hipError_t sync_err = hipStreamSynchronize(ctx->stream);
    if (H->tokens) {
        budget_release(H->device, H->tokens);
        H->tokens = 0;
    }
    MVO_HIP(sync_err);
    const uint8_t* h = (const uint8_t*)H->B.h_stats;   // stats and poses were written by the kernel itself
    const uint8_t* hp = (const uint8_t*)H->B.h_poses;
    if (!ran) return MVO_OK;
    const BaStatsDev* s = (const BaStatsDev*)h;
    for (int i = 0; i < BA_NPHASE && i < 16; ++i) ctx->ba_phase[i] = s->phase[i];
    ctx->ba_wgs = H->B.G;
    if (s->error) return mvo_set_err(ctx, MVO_ERR_HIP, "BA grid barrier timed out (workgroups not co-resident)", hipSuccess);
    if (poses && H->F) std::memcpy(poses, hp, (size_t)H->F * 128);
    if (points && H->L && !H->fix_points) std::memcpy(points, hx, (size_t)H->L * 24);
    if (st) {
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
    if (H->tokens) budget_release(H->device, H->tokens);  // (the caller has synchronised the stream)
    if (H->dev) (void)hipFree(H->dev);
    if (H->pin) (void)hipHostFree(H->pin);
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
