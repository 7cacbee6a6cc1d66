// csrc/ba_kernels.hip -- sliding-window bundle adjustment on gfx950: replaces the g2o stack that the reference's
// optimization::bundleAdjustment drives (src/optimization/g2o_ba.cpp:193-289: SparseOptimizer::optimize(50) with
// OptimizationAlgorithmLevenberg, BlockSolver<6,3>, LinearSolverDense, EdgeProjectXYZ2UV + RobustKernelHuber).
//
// ONE persistent launch runs the whole Levenberg-Marquardt loop of up to BA_MAX_BATCH independent windows (50 outer
// iterations with their data-dependent accept/reject trials): there is no host round trip per trial.  A window is
// split over G workgroups (one per CU) by LANDMARK: a workgroup owns a contiguous range of landmarks and every
// observation (edge) of them and keeps its per-edge / per-landmark state -- whitened Jacobians, landmark blocks, the
// landmarks themselves, the Schur operands U -- in LDS for the lifetime of the launch.  Window w of a batch runs on
// the workgroups blockIdx % nwin == w, i.e. with 8 windows on the CUs of ONE XCD each (placement for speed only).
//
// Cross-workgroup traffic per LM trial: the packed partial Schur system (lower triangle + rhs) is published as
// data-tagged 8-byte granules, every workgroup sums ONE slice of it over the G partials in workgroup order and
// republishes the slice, every workgroup reads the summed entries, solves the reduced 6F x 6F system itself (no
// broadcast hop) -- rows in Eigen::LDLT's pivot order, as g2o's LinearSolverDense takes them -- and, after the update,
// exchanges its robust chi2 / predicted-decrease partial all-to-all (this is also the barrier of the trial).  A solve
// that fails leaves the solver's x what it was; g2o applies and scores that stale step (ba_window, "T3" onwards).  Tags carry the launch sequence number: nothing is ever zeroed between launches.
//
// Arithmetic: f64 like g2o; the Gram-type sums run on the f64 matrix cores (v_mfma_f64_16x16x4_f64) as ONE chain of
// instructions per 16x16 tile, which is bit for bit a chain of IEEE fused multiply-adds over the rows in storage order
// (tools/probes/mfma_probe.hip); every other sum has a fixed order as well (ba_types.h).  The whole solve is
// therefore bit-reproducible and can be restated bit for bit by a scalar CPU program.  This file is compiled
// with -ffp-contract=off; fused operations are spelled __builtin_fma where they are part of the canonical arithmetic.
#include "ba_types.h"
#include "mvo_internal.h"

typedef ba_u64 u64;
// Pointers into device memory that are LOADED from the descriptor (not kernel arguments) are generic to the compiler: it emits
// FLAT instructions for them, which also count against the LDS counter (every later wait for an LDS read then waits for them too).
// The per-trial accesses to such memory go through this explicitly global type instead (global_load / global_store).
#ifndef MVO_KERNEL_SIM
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const u64 gcu64;
typedef __attribute__((address_space(1))) u64 gu64;
#define BA_AS_GLOBAL_D(p) ((gdouble*)(p))
#define BA_AS_GLOBAL_CU64(p) ((gcu64*)(p))
#define BA_AS_GLOBAL_U64(p) ((gu64*)(p))
#else
typedef double gdouble;
typedef const u64 gcu64;
typedef u64 gu64;
#define BA_AS_GLOBAL_D(p) (p)
#define BA_AS_GLOBAL_CU64(p) (p)
#define BA_AS_GLOBAL_U64(p) (p)
#endif

// per-phase cycle counters: only in the instrumented instantiation of the kernel (debug knob "ba_profile"); the
// production kernel carries none of it (s_memtime drains the memory counters, the 16 counters cost 32 SGPRs)
#define PH_BEGIN() long long ph_t = PROF ? (long long)__builtin_amdgcn_s_memtime() : 0
#define PH_END(id)                                                       \
    do {                                                                 \
        if (PROF) {                                                      \
            long long ph_n = (long long)__builtin_amdgcn_s_memtime();    \
            ph[id] += ph_n - ph_t;                                       \
            ph_t = ph_n;                                                 \
        }                                                                \
    } while (0)

// timeline of one LM trial (instrumented kernel only): shader-clock stamps of thread 0 of workgroup 0 during trial
// BA_STAMP_TRIAL, left in the rows behind the LM trace (tools/ba_probe.py prints them)
#define BA_STAMP_TRIAL 6
#define STAMP(k)                                                                                      \
    do {                                                                                              \
        if (PROF && tid == 0 && trials == BA_STAMP_TRIAL) sStamp[k] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

// ------------------------------------------------------------------------------------------------ small helpers
// The value lane (lane ^ O) holds, O = 32, 16, 8, 4, 2, 1 -- what __shfl_xor(v, O) returns, without its round trip through the LDS
// crossbar (ds_bpermute: ~100 cycles per step; a block sum sits on the critical path of every LM trial several times):
// v_permlane32_swap / v_permlane16_swap (gfx950) for the two wide steps, DPP row operations for the four inside a row of 16.
template <int O>
__device__ __forceinline__ double xor_partner_d(double v) {
#ifndef MVO_KERNEL_SIM
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    unsigned plo, phi;
    if (O == 32) {  // swap(a, a): first result = the lower half twice, second = the upper half twice
        const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const bool up = (threadIdx.x & 32) != 0;
        plo = up ? l2[0] : l2[1];
        phi = up ? h2[0] : h2[1];
    } else if (O == 16) {  // swap(a, a): first result = rows 0 0 2 2, second = rows 1 1 3 3
        const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const bool odd = (threadIdx.x & 16) != 0;
        plo = odd ? l2[0] : l2[1];
        phi = odd ? h2[0] : h2[1];
    } else if (O == 8) {  // row_ror:8
        plo = __builtin_amdgcn_update_dpp(0u, lo, 0x128, 0xf, 0xf, false);
        phi = __builtin_amdgcn_update_dpp(0u, hi, 0x128, 0xf, 0xf, false);
    } else if (O == 4) {  // banks 0, 2 read lane + 4 (row_shl:4), banks 1, 3 read lane - 4 (row_shr:4)
        plo = __builtin_amdgcn_update_dpp(0u, lo, 0x104, 0xf, 0x5, false);
        plo = __builtin_amdgcn_update_dpp(plo, lo, 0x114, 0xf, 0xa, false);
        phi = __builtin_amdgcn_update_dpp(0u, hi, 0x104, 0xf, 0x5, false);
        phi = __builtin_amdgcn_update_dpp(phi, hi, 0x114, 0xf, 0xa, false);
    } else {  // quad_perm [2, 3, 0, 1] / [1, 0, 3, 2]
        plo = __builtin_amdgcn_update_dpp(0u, lo, O == 2 ? 0x4e : 0xb1, 0xf, 0xf, false);
        phi = __builtin_amdgcn_update_dpp(0u, hi, O == 2 ? 0x4e : 0xb1, 0xf, 0xf, false);
    }
    return __hiloint2double((int)phi, (int)plo);
#else
    return __shfl_xor(v, O);
#endif
}
// xor butterfly 32, 16, .. 1 (part of the canonical arithmetic: ba_types.h)
__device__ __forceinline__ double wave_sum_d(double v) {
    v += xor_partner_d<32>(v);
    v += xor_partner_d<16>(v);
    v += xor_partner_d<8>(v);
    v += xor_partner_d<4>(v);
    v += xor_partner_d<2>(v);
    v += xor_partner_d<1>(v);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
    v = fmax(v, xor_partner_d<32>(v));
    v = fmax(v, xor_partner_d<16>(v));
    v = fmax(v, xor_partner_d<8>(v));
    v = fmax(v, xor_partner_d<4>(v));
    v = fmax(v, xor_partner_d<2>(v));
    v = fmax(v, xor_partner_d<1>(v));
    return v;
}
// deterministic block reductions (xor butterfly inside a wave, waves combined in index order)
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

// Data-tagged hand-off (no counter, no fence): a double travels as two 8-byte granules {tag : 32 | half : 32}, each
// written by ONE write-through (sc1) store; the consumer polls the granules themselves with L1-bypassing loads until
// both carry the expected tag.  Tags are unique per (launch, exchange).
// `same_l2` (uniform): all workgroups of the window were found on ONE XCD at run time (they compare their XCC ids in
// the first exchange, which always uses the write-through form) -> the stores may stay in that XCD's L2 (plain
// stores), where the consumers' L1-bypassing loads find them an order of magnitude sooner than behind the fabric.
// Placement changes only which of the two store flavours is used, never the result.
__device__ __forceinline__ void gstore_d(u64* g_, unsigned tag, double v, bool same_l2) {
    gu64* g = BA_AS_GLOBAL_U64(g_);  // (exchange areas: device memory, always)
    const u64 b = (u64)__double_as_longlong(v);
    const u64 lo = ((u64)tag << 32) | (b & 0xffffffffull), hi = ((u64)tag << 32) | (b >> 32);
    if (same_l2) {
        __hip_atomic_store(g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(g + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_store(g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool gtry_d(const u64* g_, unsigned tag, double& v) {
    gcu64* g = BA_AS_GLOBAL_CU64(g_);
    const u64 a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}
#ifndef MVO_KERNEL_SIM
#define BA_SPIN_LIMIT (1u << 21)
#else
#define BA_SPIN_LIMIT (1u << 28)  // emulated workgroups run at very different speeds (tests/sim)
#endif
// Every thread fetches its share of `count` tagged values into LDS: item q comes from the granule pair
// src[2 * (q / per * stride + q % per)] (per = values per producer row, stride = row pitch in values).  Four
// independent loads in flight per thread and pass; bounded retries.
__device__ bool gather_tagged(const u64* src, int count, int per, size_t stride, unsigned tag, double* dst,
                              long long* spins = nullptr) {
    bool fine = true;
    for (int q0 = threadIdx.x; q0 < count; q0 += 4 * BA_THREADS) {
        unsigned pending = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q0 + u * BA_THREADS < count) pending |= 1u << u;
        for (unsigned spin = 0; pending; ++spin) {
            double v[4];
            bool got[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * BA_THREADS;
                got[u] = false;
                if (pending & (1u << u)) {
                    const int row = q / per, col = q - row * per;
                    got[u] = gtry_d(src + 2 * ((size_t)row * stride + col), tag, v[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (got[u]) {
                    dst[q0 + u * BA_THREADS] = v[u];
                    pending &= ~(1u << u);
                }
            if (pending) {
                if (spin > BA_SPIN_LIMIT) {
                    fine = false;
                    break;
                }
                if (spins) ++*spins;
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    return fine;
}

// (every loop of the pose helpers is unrolled: static indices keep the small arrays in registers -- indexed dynamically
// they live in scratch memory, and the pose update is a serial chain on the critical path of an LM trial)
__device__ __forceinline__ void quat_normalize(double* q) {
    const bool neg = q[0] < 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = neg ? -q[i] : q[i];
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
__device__ __forceinline__ void quat_from_R(const double* R, double* q) {  // Eigen::Quaterniond(Matrix3d)
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {  // the branch on the largest diagonal element, written out per case (static indices only)
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
        if (i == 0) {
            double t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R[7] - R[5]) * t;
            q[2] = (R[3] + R[1]) * t;
            q[3] = (R[6] + R[2]) * t;
        } else if (i == 1) {
            double t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R[2] - R[6]) * t;
            q[3] = (R[7] + R[5]) * t;
            q[1] = (R[1] + R[3]) * t;
        } else {
            double t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[3] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R[3] - R[1]) * t;
            q[1] = (R[2] + R[6]) * t;
            q[2] = (R[5] + R[7]) * t;
        }
    }
}
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {  // Eigen toRotationMatrix
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
__device__ __forceinline__ void inv3(const double* A, double* I) {
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
__device__ __forceinline__ void invert_Rt(const double* T, double* Ri, double* ti) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    inv3(R, Ri);
#pragma unroll
    for (int i = 0; i < 3; ++i) ti[i] = -(Ri[3 * i] * T[3] + Ri[3 * i + 1] * T[7] + Ri[3 * i + 2] * T[11]);
}

// sin / cos as part of the canonical arithmetic (the libm of the host and the device library differ in the last
// bit): Cody-Waite reduction by pi/2 in three parts, then the classic degree-13 / degree-14 minimax polynomials on
// [-pi/4, pi/4], plain multiplies and adds in a fixed order.  |x| < 1e5; accurate to ~1 ulp.  The CPU restatement used by the tests holds the
// same lines.
__device__ __forceinline__ void ba_sincos(double x, double* s, double* c) {
    const double k = rint(x * 0.63661977236758134308);  // 2 / pi
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

// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(u) * T   (pose = q[4] t[3])
__device__ __forceinline__ void pose_oplus(double* P, const double* u) {
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
            V[i] = R[i];
        }
    } else {
        double st, ct;
        ba_sincos(theta, &st, &ct);
        double a = st / theta, b = (1 - ct) / (theta * theta), c = (theta - st) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double dq[4], dt[3], dR[9], nt[3];
    quat_from_R(R, dq);
    quat_normalize(dq);
#pragma unroll
    for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    quat_to_R(dq, dR);
    const double q[4] = {P[0], P[1], P[2], P[3]};
    const double t[3] = {P[4], P[5], P[6]};
#pragma unroll
    for (int i = 0; i < 3; ++i) nt[i] = dt[i] + dR[3 * i] * t[0] + dR[3 * i + 1] * t[1] + dR[3 * i + 2] * t[2];
    double nq[4] = {dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2] - dq[3] * q[3],
                    dq[0] * q[1] + dq[1] * q[0] + dq[2] * q[3] - dq[3] * q[2],
                    dq[0] * q[2] - dq[1] * q[3] + dq[2] * q[0] + dq[3] * q[1],
                    dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1] + dq[3] * q[0]};
    quat_normalize(nq);
#pragma unroll
    for (int i = 0; i < 4; ++i) P[i] = nq[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) P[4 + i] = nt[i];
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

__device__ __forceinline__ void asm_waitcnt_vm0() {
#ifndef MVO_KERNEL_SIM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
// the XCD this workgroup runs on (placement is used for speed only, never for results)
__device__ __forceinline__ unsigned ba_xcc_id() {
#ifndef MVO_KERNEL_SIM
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return xcc & 15u;
#else
    return blockIdx.x & 7u;
#endif
}

// Minimum of a 32-bit value over the wave, in every lane.
__device__ __forceinline__ int wave_min_i(int v) {
#ifndef MVO_KERNEL_SIM
    const int big = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false));  // row_shr:8   -> lane 15 of every row
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3 -> lane 63
    return __builtin_amdgcn_readlane(v, 63);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
#endif
}

// Sum of a 32-bit value over the aligned group of `width` (4 / 8 / 16) consecutive lanes, in every lane of the group.
__device__ __forceinline__ int group_sum_i(int v, int width) {
#ifndef MVO_KERNEL_SIM
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]: lane ^ 1
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);   // quad_perm [2, 3, 0, 1]: lane ^ 2 -> every lane: its quad's sum
    if (width >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror: the other quad of the 8
    if (width >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); // row_mirror: the other half of the 16
    return v;
#else
    for (int o = 1; o < width; o <<= 1) v += __shfl_xor(v, o);
    return v;
#endif
}

// The pivot order of Eigen::LDLT (g2o's LinearSolverDense, g2o_ba.cpp:196-197).  Eigen's factorisation (LDLT.h,
// ldlt_inplace<Lower>::unblocked) swaps, at step k, the largest |diagonal entry| among the positions k .. n-1 -- the FIRST
// maximum -- to position k; it is left-looking, so the entries behind k still hold the INPUT diagonal when they are compared:
// the order is a function of diag(S) alone.  Without equal entries that is the descending order: dense ranks by counting,
// made by the whole workgroup (ba_window).  With equal entries -- the rule in pose-only windows: the x and y translation entries
// of a pose block are the same sums -- the swap sequence itself is replayed by ONE wave on integer keys (dense rank << 7 |
// position: the wave minimum is the first maximum); position lane and lane + 64 per lane (n <= 128).  rank[i]: number of
// entries strictly larger than entry i.  perm[k] = row of S that ends up at position k.
// (out of line: run by one wave on the rare path; inlined, its registers would count against the whole LM loop)
__device__ __attribute__((noinline)) void ba_pivot_replay(const short* rank, int n, short* perm, int lane) {
    const int done = 0x7fffffff;
    if (n <= 64) {
        // one position per lane: rk = dense rank of the element sitting there (-1 once the position is final), el = the element.
        // Dense ranks are the cumulative group sizes, so the group that is being taken at step k is known without a reduction: a new
        // one starts at step k exactly when some element has rank k, else the running one continues; its member at the lowest
        // position is the first maximum: one ballot + s_ff1 instead of a six-step wave minimum.
        int rk = lane < n ? (int)rank[lane] : -1, el = lane, out = 0, cur = 0;
        for (int k = 0; k < n; ++k) {
            const unsigned long long fresh = __ballot(rk == k), running = __ballot(rk == cur);
            const unsigned long long m = fresh ? fresh : running;
            cur = fresh ? k : cur;
            const int big = m ? (int)__builtin_ctzll(m) : k;         // position of the first maximum among k .. n-1
            const int kr = __builtin_amdgcn_readlane(rk, k), ke = __builtin_amdgcn_readlane(el, k), eb = __builtin_amdgcn_readlane(el, big);
            rk = lane == big ? kr : rk;                              // what sat at position k goes to position `big` ...
            el = lane == big ? ke : el;
            rk = lane == k ? -1 : rk;                                // ... and position k is final
            out = lane == k ? eb : out;
        }
        if (lane < n) perm[lane] = (short)out;
        return;
    }
    const int p0 = lane, p1 = lane + 64;
    int k0 = p0 < n ? ((int)rank[p0] << 7) | p0 : done, k1 = p1 < n ? ((int)rank[p1] << 7) | p1 : done;
    int e0 = p0, e1 = p1;
    for (int k = 0; k < n; ++k) {
        const int big = wave_min_i(min(k0, k1)) & 127;
        const int ek = k < 64 ? __builtin_amdgcn_readlane(e0, k) : __builtin_amdgcn_readlane(e1, k - 64);
        const int kk = k < 64 ? __builtin_amdgcn_readlane(k0, k) : __builtin_amdgcn_readlane(k1, k - 64);
        const int eb = big < 64 ? __builtin_amdgcn_readlane(e0, big) : __builtin_amdgcn_readlane(e1, big - 64);
        const int kmoved = (kk & ~127) | big;
        if (p0 == big) e0 = ek, k0 = kmoved;
        if (p1 == big) e1 = ek, k1 = kmoved;
        if (p0 == k) e0 = eb, k0 = done;
        if (p1 == k) e1 = eb, k1 = done;
    }
    if (p0 < n) perm[p0] = (short)e0;
    if (p1 < n) perm[p1] = (short)e1;
}

#include "ba_solve.h"  // readlane_d, ba_rcp_pivot, the one-wave solver of the 5-pose class (solve_wave_32), the block solver (solve_block)

// computeScale() of the landmark part of the solver's current x over one range: sum over its landmarks l (lane = l mod 64, then one
// 64-lane butterfly) of x_l . (lambda x_l + b_l).  x: device memory (L1-bypassing loads), b_l: LDS at offset bl_off of the dynamic
// segment.  Out of line and run by ONE wave beside the Schur chains: its registers do not count against the LM loop's.
__device__ __attribute__((noinline)) double ba_stale_scale_landmarks(gcu64* dxl, int bl_off, int Lg, double lambda, int lane) {
    const double* bl = ba_dyn_lds + bl_off;
    double sp = 0;
    for (int l0 = lane; l0 < Lg; l0 += 4 * 64) {
        double xs[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int l = l0 + 64 * u;
                xs[u][c] = l < Lg ? __longlong_as_double((long long)__hip_atomic_load(dxl + 3 * l + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + 64 * u;
            if (l < Lg) {
#pragma unroll
                for (int c = 0; c < 3; ++c) sp += xs[u][c] * (lambda * xs[u][c] + bl[3 * l + c]);
            }
        }
    }
    return wave_sum_d(sp);
}


// the same arithmetic for more than 63 unknowns (> 10 free poses): one wave, matrix in LDS (row pitch n + 2)
__device__ __attribute__((noinline)) int solve_lds(int sl_off, int cb_off, int n, int lane) {
    double* S = ba_dyn_lds + sl_off;
    double* col = ba_dyn_lds + cb_off;  // n + 1 <= 121 doubles of scratch; the solution is left in col[0 .. n)
    const int ld = n + 2;
    int ok = 1;
    for (int j = 0; j < n; ++j) {
        const double d = S[j * ld + j];
        if (!(d >= BA_PIVOT_MIN && d <= BA_PIVOT_MAX)) {
            ok = 0;
            break;
        }
        const double r = ba_rcp_pivot(d);
        for (int i = j + 1 + lane; i <= n; i += 64) col[i] = S[i * ld + j];
        __builtin_amdgcn_wave_barrier();
        for (int i = j + 1 + lane; i <= n; i += 64) {
            const double l = col[i] * r;
            const int kend = i < n ? i : n - 1;
            for (int k = j + 1; k <= kend; ++k) S[i * ld + k] = __builtin_fma(-l, col[k], S[i * ld + k]);
            S[i * ld + j] = l;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (ok) {
        for (int j = lane; j < n; j += 64) col[j] = S[n * ld + j];  // z
        __builtin_amdgcn_wave_barrier();
        for (int i = n - 1; i >= 1; --i) {
            const double xi = col[i];
            for (int j = lane; j < i; j += 64) col[j] = __builtin_fma(-S[i * ld + j], xi, col[j]);
            __builtin_amdgcn_wave_barrier();
        }
    }
    return ok;
}

// packed order of the reduced system: lower triangle row by row (j <= i < n), then the rhs row (n, j)
__device__ __forceinline__ void packed_ij(int idx, int n, int& i, int& j) {
    const int ntri = n * (n + 1) / 2;
    if (idx < ntri) {
        i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > idx) --i;
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        j = idx - i * (i + 1) / 2;
    } else {
        i = n;
        j = idx - ntri;
    }
}

// LDS layout of one workgroup, carved from the dynamic segment.  What a workgroup keeps for the lifetime of the launch is
// per LANDMARK (position + backup, H_ll, b_l, C, C^T b_l) and a few bytes per edge (measured pixel, indices); the
// whitened Jacobian rows of an edge -- A~ (2 x 6) and X~ (2 x 3) -- live in the REGISTERS of the thread that owns the
// edge (edge el < 512: thread el; the edges behind 512 of a fuller range keep theirs in the E2 area) and visit LDS only as
// staging copies for the two phases that read them across threads (landmark blocks, pose-block chains).  `U` is a multi-purpose area: staging during linearisation, one
// chunk of the Schur operands at a time during a trial, the per-edge back-substitution terms afterwards.
struct WgLds {
    double* SL;     // reduced system / transposed L; split-chain tiles and the slice reduction are staged here as well
    double* colbuf; // 2 x 64 column broadcast + 64 scratch
    double* pan;    // BA_PANEL_DOUBLES: panel rows of the block solver
    double* Rl;     // nlow + nhp (+16) packed entries of the summed Schur system (+ pose blocks)
    double* hpl;    // nhp own pose-block partials (waiting for the next Schur exchange, or the only ones when G = 1)
    double* uv;     // maxEg x 2 (u and v in separate arrays); not carved when the measurements stay in device memory:
    double* uvg;    // this range's (u, v) pairs in device memory (BaDev::uv_global)
    double* pts;    // maxLg x 3
    double* bak;    // maxLg x 3
    double* Hll;    // maxLg x 6 (pitch BA_XS)
    double* bl;     // maxLg x 3
    double* Cc;     // maxLg x 6 (pitch BA_XS)   Cholesky factor of (H_ll + lambda I)^-1
    double* cl;     // maxLg x 3   C^T b_l
    double* dxl;    // Lg x 3 in DEVICE memory (BaDev::dxl_dev, this range's part): landmark part of the solver's x = the step of the
                    // last SUCCESSFUL solve (g2o keeps it when a solve fails).  Written by the landmark's thread when a step is
                    // applied, read (L1-bypassing) by the last wave beside the Schur chains and when a stale step is applied.
    double* U;      // `uarea` doubles: see above
    double* E2;     // rows [a0 | a1 | x | e~] (pitch BA_E2S) of the edges that do not keep them in registers: all edges
                    // (SLOTS = 0) or the edges 512 .. Eg - 1
    short* epose;   // maxEg
    short* ept;     // maxEg  local landmark index
    short* dup;     // maxEg  rank of the edge among the observations of its (landmark, pose): 0 for the first
    short* ptl;     // maxEg  local edge indices grouped by landmark
    short* pti;     // maxEg  position of every edge in ptl (the inverse permutation)
    short* pts0;    // maxLg + 1 offsets into ptl
    short* eof;     // maxLg x nfree
};

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
    const double mu = B.uv_global ? W.uvg[2 * el] : W.uv[el], mv = B.uv_global ? W.uvg[2 * el + 1] : W.uv[B.maxEg + el];
    const double e0 = mu - (Xc[0] / Xc[2] * B.f + B.cx);
    const double e1 = mv - (Xc[1] / Xc[2] * B.f + B.cy);
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

// U buffer addressing: column `col` (= 3 landmark + k), row `row`.  The column pitch `ldu` is ODD (rows + 1): the
// per-landmark phases run one lane per column (all lanes touch the same row of different columns); with an even pitch of
// 32 doubles every such access would land on one LDS bank.
__device__ __forceinline__ int u_index(int col, int row, int ldu) {
    return col * ldu + row;
}

// One chain of the partial Schur system: tile pair (ti, tj), MFMA steps [m0, m1) over the columns of U, four columns
// per instruction: acc[i][j] = fma chain over the columns of U[col][16 ti + i] * U[col][16 tj + j].  The columns
// behind the last landmark (up to the next multiple of four) are zero.
// The wave issues in order and stalls at every dependent MFMA, so everything else of a group of four steps -- the eight
// operand loads of the NEXT group -- is placed right behind the first MFMA, where it is issued while the matrix core
// works; two register sets alternate (no copies) and the column pitch is a compile-time constant (immediate offsets).
template <int LDU>
__device__ __forceinline__ v4d schur_chain_t(const double* U, int ti, int tj, int m0, int m1, int lane) {
    v4d acc = {0, 0, 0, 0};
    const int k = lane >> 4, i = lane & 15;
    constexpr int st = 4 * LDU;
    const double* pa = U + (size_t)(4 * m0 + k) * LDU + (16 * ti + i);
    const double* pb = U + (size_t)(4 * m0 + k) * LDU + (16 * tj + i);
    int m = m0;
#define BA_LOAD4(A, Bv) \
    A##0 = pa[0], Bv##0 = pb[0], A##1 = pa[st], Bv##1 = pb[st], A##2 = pa[2 * st], Bv##2 = pb[2 * st], A##3 = pa[3 * st], Bv##3 = pb[3 * st]; \
    pa += 4 * st;                                                                                                                           \
    pb += 4 * st
#define BA_MFMA4_LOADNEXT(A, Bv, C, D, more)                                     \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##0, Bv##0, acc, 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                           \
    if (more) { BA_LOAD4(C, D); }                                                \
    __builtin_amdgcn_sched_barrier(0);                                           \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##1, Bv##1, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##2, Bv##2, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##3, Bv##3, acc, 0, 0, 0)
    if (m + 4 <= m1) {
        double a0, a1, a2, a3, b0, b1, b2, b3, c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        BA_LOAD4(a, b);
        m += 4;
        for (;;) {
            const bool more1 = m + 4 <= m1;
            BA_MFMA4_LOADNEXT(a, b, c, d, more1);
            if (!more1) break;
            m += 4;
            const bool more2 = m + 4 <= m1;
            BA_MFMA4_LOADNEXT(c, d, a, b, more2);
            if (!more2) break;
            m += 4;
        }
    }
#undef BA_LOAD4
#undef BA_MFMA4_LOADNEXT
    for (; m < m1; ++m) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[0], pb[0], acc, 0, 0, 0);
        pa += st;
        pb += st;
    }
    return acc;
}
__device__ __forceinline__ v4d schur_chain(const double* U, int ldu, int ti, int tj, int m0, int m1, int lane) {
    switch (ldu) {
        case 17: return schur_chain_t<17>(U, ti, tj, m0, m1, lane);
        case 33: return schur_chain_t<33>(U, ti, tj, m0, m1, lane);
        case 49: return schur_chain_t<49>(U, ti, tj, m0, m1, lane);
        case 65: return schur_chain_t<65>(U, ti, tj, m0, m1, lane);
        default: break;
    }
    v4d acc = {0, 0, 0, 0};  // larger windows (> 10 free poses): plain loop
    const int k = lane >> 4, i = lane & 15;
    const double* pa = U + (size_t)(4 * m0 + k) * ldu + (16 * ti + i);
    const double* pb = U + (size_t)(4 * m0 + k) * ldu + (16 * tj + i);
    for (int m = m0; m < m1; ++m) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[0], pb[0], acc, 0, 0, 0);
        pa += 4 * ldu;
        pb += 4 * ldu;
    }
    return acc;
}

// The whitened Jacobian rows of one edge as its owner thread keeps them between the linearisation and the end of the
// iteration's trials (EdgeProjectXYZ2UV::linearizeOplus with sqrt(rho') Lc folded in).
struct EdgeRegs {
    double a0[6], a1[6];  // A~ = sqrt(rho') Lc J_pose (2 x 6); zero for a fixed pose
    double x[6];          // X~ = sqrt(rho') Lc J_point (2 x 3)
};
// Y = X~ C (2 x 3) for the lower-triangular C = {c00, c10, c11, c20, c21, c22}
__device__ __forceinline__ void edge_Y(const EdgeRegs& r, const double* cc, double* Y) {
    const double c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3], c4 = cc[4], c5 = cc[5];
    Y[0] = r.x[0] * c0 + r.x[1] * c1 + r.x[2] * c3;
    Y[1] = r.x[1] * c2 + r.x[2] * c4;
    Y[2] = r.x[2] * c5;
    Y[3] = r.x[3] * c0 + r.x[4] * c1 + r.x[5] * c3;
    Y[4] = r.x[4] * c2 + r.x[5] * c4;
    Y[5] = r.x[5] * c5;
}

// PROF: per-phase cycle counters; NR: reduced-solve flavour -- 32: one-wave register solver for n + 1 <= 32 rows; 64 / -32:
// workgroup-wide block solver for n + 1 <= 64 / 32 rows; 0: LDS solver (any n).  Separate instantiations: the solvers
// differ widely in register use.
// SLOTS: where the rows of an edge live between the linearisation and the trials -- 0: all of them in LDS (ranges that
// have the room: the latency cut of the 5-keyframe window; fewest registers), 1: in the registers of thread `edge` (<= 512
// edges per range), 2: edges behind 512 in LDS (ranges of up to 1024 edges).
// One window, one workgroup of it: `desc` = the window's descriptor (device memory, or pinned host memory for the resident
// solver service), g = this workgroup's range.  Called once per launch by k_ba_lm and once per job by k_ba_service.
struct BaRun {
    unsigned tag0;
    int use_mfma, same_l2_ok;
};
template <bool PROF, int NR, int SLOTS>
__device__ __forceinline__ void ba_window(const BaDev* desc, const BaRun batch, const int g) {
    // the descriptor is staged in LDS: read through the global pointer the compiler has to assume that every store of
    // the kernel may have changed it and re-loads fields from memory all over the LM loop
    __shared__ BaDev sB;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(desc);
        unsigned* dst = reinterpret_cast<unsigned*>(&sB);
        for (unsigned i = threadIdx.x; i < sizeof(BaDev) / 4; i += BA_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const BaDev& B = sB;
    if (g >= B.G) return;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    double* dyn = ba_dyn_lds;
    __shared__ double sScr[BA_WAVES];
    // per-pose state and the per-workgroup exchange values live at the front of the dynamic segment (sized by F and G)
    double* const sP = dyn;                       // q[4] t[3] pad
    double* const sPbak = sP + 8 * B.F;
    double* const sR = sPbak + 8 * B.F;
    double* const sT = sR + 9 * B.F;
    double* const sHpp = sT + 3 * B.F;
    double* const sBp = sHpp + 36 * B.F;
    double* const sDx = sBp + 6 * B.F;
    double* const sSol = sDx + 6 * B.F;           // 6 F (+ rhs slot)
    double* const sX = dyn + ba_pose_doubles(B.F);  // 2 G
    __shared__ int sSlot[BA_MAX_POSES], sSlotPose[BA_MAX_POSES], sPoseStart[BA_MAX_POSES + 1];
    __shared__ short sSlc[3][BA_HP_PASSES][BA_MAX_POSES];  // pose-block chains: slices of the staging passes (below)
    __shared__ int sFlag[6];
    __shared__ double sStale[BA_WAVES + 2];  // wave partials of the stale step's predicted decrease; [8]: rho of a failed solve
    __shared__ short sPermAll[2][6 * BA_MAX_POSES];  // Eigen's pivot order of the trial's reduced system: [0] by ranks, [1] by replay
    short* const sPerm = sPermAll[0];
    __shared__ short sRank[6 * BA_MAX_POSES];  // dense ranks of |diag S| of the last trial
    short* const sPermTie = sPermAll[1];  // the replayed order that belongs to sRank whenever sRank has equal entries
    __shared__ long long sStamp[PROF ? 32 : 1];  // (parked in LDS: a store to host memory in front of a barrier would be timed)
    if (threadIdx.x == 0) sFlag[1] = sFlag[2] = sFlag[3] = sFlag[4] = 0;  // [1] NaN on the diagonal, [2] exchange timed out, [3] equal diagonal entries, [4] ranks changed
    if (threadIdx.x < 6 * BA_MAX_POSES) sRank[threadIdx.x] = -1;  // (no order replayed yet)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = B.n, G = B.G, nfree = B.nfree, urows = B.ldu, ldu = B.ldu + 1, nlow = B.nlow, npk = B.npk;
    const int pt_lo = B.wg_pt_start[g], Lg = B.wg_pt_start[g + 1] - pt_lo;
    const int e_lo = B.wg_edge_start[g], Eg = B.wg_edge_start[g + 1] - e_lo;
    const unsigned tag0 = batch.tag0;
    unsigned tagA = 0, tagB = 0, tagH = 0;

    // ---- carve the dynamic LDS (the same sizes as ba_lds_bytes)
    WgLds W;
    double* stage;
    {
        double* d = dyn + ba_pose_doubles(B.F) + 2 * (size_t)G + 8;
        W.SL = d;
        d += ba_solver_doubles(n, nlow + B.nhp, G, B.npair, B.npar, B.panel, B.alias_sl) - 3 * 64 - (B.panel ? BA_PANEL_DOUBLES : 0);
        W.colbuf = d;
        d += 3 * 64;
        W.pan = d;
        d += B.panel ? BA_PANEL_DOUBLES : 0;
        W.Rl = d;
        d += nlow + B.nhp + 16;
        W.hpl = d;
        d += B.nhp + 1;
        W.uv = d;
        d += B.uv_global ? 0 : (size_t)B.maxEg * 2;
        W.uvg = B.uv_dev + 2 * (size_t)e_lo;  // (uv_global: this range's measurements in device memory)
        W.pts = d;
        d += (size_t)B.maxLg * 3;
        // (sizes are zero in pose-only mode: every pointer stays an LDS address, no null pointers in this struct)
        const size_t full = B.fix_points ? 0 : 1;
        W.bak = d;
        d += full * B.maxLg * 3;
        W.Hll = d;
        d += full * B.maxLg * BA_XS;
        W.bl = d;
        d += full * B.maxLg * 3;
        W.Cc = d;
        d += full * B.maxLg * BA_XS;
        W.cl = d;
        d += full * B.maxLg * 3;
        W.dxl = B.dxl_dev + 3 * (size_t)pt_lo;
        W.U = d;
        stage = d;
        if (B.alias_sl) W.SL = d;  // (the matrix shares the U area: alive only between the last chain and the back-substitution)
        d += B.uarea;
        W.E2 = d;
        d += (size_t)B.e2_edges * BA_E2S;
        short* s = reinterpret_cast<short*>(d);
        W.epose = s;
        s += B.maxEg;
        W.ept = s;
        s += B.maxEg;
        W.dup = s;
        s += B.maxEg;
        W.ptl = s;
        s += B.maxEg;
        W.pti = s;
        s += B.maxEg;
        W.pts0 = s;
        s += B.maxLg + 1;
        W.eof = s;
    }
    // ---- load the workgroup's slice: edges, landmarks, adjacency; every workgroup holds all poses
    for (int el = tid; el < Eg; el += BA_THREADS) {
        W.epose[el] = (short)B.e_pose[e_lo + el];
        W.ept[el] = (short)(B.e_point[e_lo + el] - pt_lo);
        W.dup[el] = B.dup_rank[e_lo + el];
        {
            const int pe = B.pt_edge_list[e_lo + el] - e_lo;
            W.ptl[el] = (short)pe;
            W.pti[pe] = (short)el;
        }
        if (B.uv_global) {  // (thread `el % 512` is also the only reader of edge el's measurement)
            W.uvg[2 * el] = B.e_uv[2 * (size_t)(e_lo + el)];
            W.uvg[2 * el + 1] = B.e_uv[2 * (size_t)(e_lo + el) + 1];
        } else {
            W.uv[el] = B.e_uv[2 * (size_t)(e_lo + el)];  // (u and v in separate arrays)
            W.uv[B.maxEg + el] = B.e_uv[2 * (size_t)(e_lo + el) + 1];
        }
    }
    for (int i = tid; i < 3 * Lg; i += BA_THREADS) W.pts[i] = B.pts_in[3 * (size_t)pt_lo + i];
    if (!B.fix_points)
        for (int l = tid; l < Lg; l += BA_THREADS)
            BA_AS_GLOBAL_D(W.dxl)[3 * l] = BA_AS_GLOBAL_D(W.dxl)[3 * l + 1] = BA_AS_GLOBAL_D(W.dxl)[3 * l + 2] = 0.0;  // (the solver's x before the first solve)
    if (tid < 6 * B.F) sDx[tid] = 0.0;
    for (int i = tid; i <= Lg; i += BA_THREADS) W.pts0[i] = (short)(B.pt_edge_start[pt_lo + i] - e_lo);
    for (int i = tid; i < Lg * nfree; i += BA_THREADS) W.eof[i] = B.eof[(size_t)pt_lo * nfree + i];
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
    if (tid < nfree) sSlotPose[tid] = B.slot_pose[tid];
    if (tid <= B.F) sPoseStart[tid] = B.wg_pose_start[g * (B.F + 1) + tid] - e_lo;
    __syncthreads();

    double lambda = 0, ni = 2;
    int it = 0, trials = 0, terminated = 0, error = 0, failed_solves = 0, stale_steps = 0;
    long long ph[PROF ? BA_NPHASE : 1] = {0};
    const long long ph_start = (long long)__builtin_amdgcn_s_memtime();
    // ---- initial robust chi2: all-to-all through the chi2 slots (tag 1 of the B series)
    double currentChi;
    bool same_l2 = false;  // all workgroups of the window on one XCD (found out in the first exchange)
    bool grp_l2 = false;   // every group g mod K of the window on one XCD of its own (K = 1: the same thing)
    const int K = B.groups > 1 ? B.groups : 1;
    {
        double c = robust_chi2_local(B, W, Eg, sR, sT, sScr);
        if (G > 1) {
            ++tagB;
            u64* slot = B.xC + (size_t)(tagB & 1) * G * 4;
            const unsigned xcc = ba_xcc_id();
            if (tid == 0) {
                gstore_d(slot + 4 * g, tag0 + tagB, c, false);
                gstore_d(slot + 4 * g + 2, tag0 + tagB, (double)xcc, false);  // (second value: where this workgroup runs)
            }
            if (!gather_tagged(slot, 2 * G, 2, 2, tag0 + tagB, sX)) sFlag[2] = 1;
            __syncthreads();
            if (sFlag[2]) error = 1;
            c = 0;
            bool one = true, grp = true;
            for (int w = 0; w < G; ++w) {
                c += sX[2 * w];
                one = one && sX[2 * w + 1] == sX[1];
                grp = grp && sX[2 * w + 1] == sX[2 * (w % K) + 1];
            }
            same_l2 = one && batch.same_l2_ok;
            grp_l2 = grp && batch.same_l2_ok;
            __syncthreads();
        }
        currentChi = c;
    }
    const double chi0 = currentChi;
    const bool any_free = nfree > 0 || !B.fix_points;
    const bool do_schur = !B.fix_points && n > 0;
    // ---- the Schur chains of this workgroup: `nsplit` consecutive column pieces of `msplit` MFMA steps each, `npar` of
    // them side by side (on different waves) per chunk of the U area, `nseq` chunks one after the other
    const int npar = B.npar, nseq = B.nseq;
    const int ncol = 3 * Lg;
    const int msteps = (ncol + 3) / 4;
    const int msplit = (msteps + B.nsplit - 1) / B.nsplit;
    const int chunk_cols = 4 * npar * msplit;
    // where the entries of this wave's first Schur chain go in the packed order (constant for the whole solve)
    const short* pkt_g = B.pk_of_tile;
    int pk4[4] = {-1, -1, -1, -1};
    if (do_schur && wave < B.npair * npar) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pk4[j] = pkt_g[(wave % B.npair) * 256 + ((lane >> 4) + 4 * j) * 16 + (lane & 15)];
    }
    // ---- this thread's edges: edge `tid` (rows in registers) and, in ranges with more than 512 edges, edge tid + 512
    // (rows in the E2 area)
    EdgeRegs er;
#pragma unroll
    for (int c = 0; c < 6; ++c) er.a0[c] = er.a1[c] = er.x[c] = 0;
    const bool have0 = SLOTS > 0 && tid < Eg, have1 = SLOTS > 1 && tid + BA_THREADS < Eg;
    const int e0_p = have0 ? W.epose[tid] : 0, e0_l = have0 ? W.ept[tid] : 0, e0_sl = have0 ? sSlot[e0_p] : -1;
    constexpr int nslot = SLOTS;
    constexpr int e2_first = SLOTS == 0 ? 0 : BA_THREADS;  // first edge of the E2 area
// BA_EDGES(BODY): BODY(el, rows, landmark, slot, e~ pair) for this thread's edges -- from its registers (edge `tid`), from the
// E2 area (SLOTS = 2: edge tid + 512; SLOTS = 0: edges tid, tid + 512, ...).  BODY is a statement macro; `break` leaves
// it.  (Spelled with macros, element by element: behind a lambda capture the compiler parks the register-resident rows
// in scratch memory.)
#define BA_EDGE_FROM_LDS(BODY, el_)                                          \
    do {                                                                     \
        EdgeRegs r1_;                                                        \
        double ee_[2];                                                       \
        const double* q_ = W.E2 + BA_E2S * ((el_)-e2_first);                 \
        _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) {                   \
            r1_.a0[c_] = q_[c_];                                             \
            r1_.a1[c_] = q_[6 + c_];                                         \
            r1_.x[c_] = q_[12 + c_];                                         \
        }                                                                    \
        ee_[0] = q_[18];                                                     \
        ee_[1] = q_[19];                                                     \
        const int p1_ = W.epose[el_];                                        \
        (void)p1_;                                                           \
        BODY(el_, r1_, W.ept[el_], sSlot[p1_], ee_)                          \
    } while (0);
#define BA_EDGES(BODY)                                                       \
    if (nslot > 0 && have0) do { BODY(tid, er, e0_l, e0_sl, ee0) } while (0); \
    if (nslot > 1 && have1) BA_EDGE_FROM_LDS(BODY, tid + BA_THREADS)         \
    if (nslot == 0)                                                          \
        for (int el0_ = tid; el0_ < Eg; el0_ += BA_THREADS) BA_EDGE_FROM_LDS(BODY, el0_)

    // ---- pose-block chains: how the rows [A~ | e~] are staged (constant for the whole solve, made once).  Pass q of
    // `hp_npass` holds slice q of EVERY pose: edges [s_p + q h_p, min(e_p, s_p + (q + 1) h_p)), h_p = ceil(n_p / npass) rounded up
    // to even (a slice is whole MFMA steps), the slices of a pass packed back to back in the staging area.  sSlc[0 / 1 / 2][q][p] =
    // first edge / one past the last edge / place in the area.
    int hp_npass = 1;  // smallest number of passes whose slices fit the area together (uniform)
    {
        const int mcap = (int)(B.uarea / BA_MSTRIDE);
        for (;; ++hp_npass) {
            int tot = 0;
            for (int p = 0; p < B.F; ++p) {
                const int n_p = sPoseStart[p + 1] - sPoseStart[p];
                tot += min(n_p, ((n_p + hp_npass - 1) / hp_npass + 1) & ~1);
            }
            if (tot <= mcap) break;
            if (hp_npass > Eg) {
                error = 1;
                break;
            }
        }
    }
    const bool hp_tables = hp_npass <= BA_HP_PASSES;
    auto hp_slice_row = [&](int q, int row) {
        if (tid < B.F) {
            int off = 0, lo = 0, hi = 0;
            for (int pp = 0; pp <= tid; ++pp) {
                const int s0 = sPoseStart[pp], n_p = sPoseStart[pp + 1] - s0;
                const int h = ((n_p + hp_npass - 1) / hp_npass + 1) & ~1;
                off += hi - lo;
                lo = min(s0 + q * h, s0 + n_p);
                hi = min(s0 + (q + 1) * h, s0 + n_p);
            }
            sSlc[0][row][tid] = (short)lo;
            sSlc[1][row][tid] = (short)hi;
            sSlc[2][row][tid] = (short)off;
        }
    };
    if (hp_tables && !error)
        for (int q = 0; q < hp_npass; ++q) hp_slice_row(q, q);
    // pairs of free poses this wave chains (two rounds at most: F <= 20 -> <= 10 pairs on 8 waves)
    int nfp = 0, hp_pa0 = -1, hp_pb0 = -1, hp_pa1 = -1, hp_pb1 = -1;
    for (int p = 0; p < B.F; ++p) {
        if (sSlot[p] < 0) continue;
        if (nfp == 2 * wave) hp_pa0 = p;
        if (nfp == 2 * wave + 1) hp_pb0 = p;
        if (nfp == 2 * (wave + BA_WAVES)) hp_pa1 = p;
        if (nfp == 2 * (wave + BA_WAVES) + 1) hp_pb1 = p;
        ++nfp;
    }
    __syncthreads();

    // pivot order of every trial (T3): the H_pp diagonal entries this thread compares -- row i = tid >> lsh against j = jj, jj + LPR --
    // as offsets into sHpp, three times 10 bits (constant for the whole solve)
    int piv_hoff = 0;
    if (NR == 32 && n > 0) {  // (the larger classes form their entries once per trial into LDS instead)
        const int lsh = NR == 32 ? 4 : (n <= 32 ? 4 : (n <= 64 ? 3 : 2)), LPR = 1 << lsh, jj = tid & (LPR - 1);
        const int qi = min(tid >> lsh, n - 1), q0 = min(jj, n - 1), q1 = min(jj + LPR, n - 1);
        piv_hoff = (36 * sSlotPose[qi / 6] + 7 * (qi % 6)) | (36 * sSlotPose[q0 / 6] + 7 * (q0 % 6)) << 10 | (36 * sSlotPose[q1 / 6] + 7 * (q1 % 6)) << 20;
    }
    for (it = 0; any_free && !error && it < B.max_it; ++it) {
        // ================= LIN: whitened Jacobians of the own edges (EdgeProjectXYZ2UV::linearizeOplus), kept in registers;
        // X~ and e~ also go to the staging area for the landmark blocks
        PH_BEGIN();
        double ee0[2] = {0, 0};  // e~ of the register edge (needed again when the chain rows are staged)
// rows of edge `el` (pose p, slot sl) into r, its whitened error into ee; X~ and e~ also into the staging area
#define BA_LINEARIZE(el, r, p, sl, ee)                                                                              \
    {                                                                                                               \
        double Xc[3], ew[2], rho0_, rho1_;                                                                          \
        const double chi = edge_error(B, W, el, sR, sT, Xc, ew);                                                    \
        huber(chi, B.delta, rho0_, rho1_);                                                                          \
        const double sw = sqrt(rho1_);                                                                              \
        const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = B.f;                                          \
        if (sl >= 0) {                                                                                              \
            const double J0[6] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f};   \
            const double J1[6] = {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};    \
            _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                         \
                r.a0[c] = sw * (B.lc00 * J0[c] + B.lc01 * J1[c]);                                                   \
                r.a1[c] = sw * (B.lc11 * J1[c]);                                                                    \
            }                                                                                                       \
        } else {                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 6; ++c) r.a0[c] = r.a1[c] = 0;                                    \
        }                                                                                                           \
        ee[0] = sw * ew[0];                                                                                         \
        ee[1] = sw * ew[1];                                                                                         \
        _Pragma("unroll") for (int c = 0; c < 6; ++c) r.x[c] = 0;                                                   \
        if (!B.fix_points) {                                                                                        \
            const double* R = sR + 9 * (p);                                                                         \
            const double t0[3] = {f, 0, -x / z * f}, t1[3] = {0, f, -y / z * f};                                    \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                         \
                double j0 = -1. / z * (t0[0] * R[c] + t0[1] * R[3 + c] + t0[2] * R[6 + c]);                         \
                double j1 = -1. / z * (t1[0] * R[c] + t1[1] * R[3 + c] + t1[2] * R[6 + c]);                         \
                r.x[c] = sw * (B.lc00 * j0 + B.lc01 * j1);                                                          \
                r.x[3 + c] = sw * (B.lc11 * j1);                                                                    \
            }                                                                                                       \
        }                                                                                                           \
    }
        if (nslot > 0 && have0) BA_LINEARIZE(tid, er, e0_p, e0_sl, ee0)
        for (int el = nslot == 0 ? tid : tid + BA_THREADS; nslot != 1 && el < Eg; el += BA_THREADS) {
            EdgeRegs r1;
            double ee[2];
            const int p1 = W.epose[el], sl1 = sSlot[p1];
            BA_LINEARIZE(el, r1, p1, sl1, ee)
            double* q = W.E2 + BA_E2S * (el - e2_first);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                q[c] = r1.a0[c];
                q[6 + c] = r1.a1[c];
                q[12 + c] = r1.x[c];
            }
            q[18] = ee[0];
            q[19] = ee[1];
        }
        __syncthreads();
        PH_END(0);
        // ================= PT: 3x3 landmark blocks H_ll, b_l of the own landmarks.  The rows [X~ | e~] of the edges are staged
        // in the order of the landmark-grouped edge list, one slice of the range's landmarks per pass (B.npt passes: the area
        // need not hold all edges at once), and read back sequentially by the landmark's thread.
        double maxdiag = 0;
        if (!B.fix_points) {
            for (int h = 0; h < B.npt; ++h) {
                const int l_lo = (int)((long long)Lg * h / B.npt), l_hi = (int)((long long)Lg * (h + 1) / B.npt);
                const int k_lo = W.pts0[l_lo], k_hi = W.pts0[l_hi];
#define BA_BODY_STAGE_X(el, r, l_, sl_, ee)                                   \
    const int k_ = W.pti[el];                                                \
    if (k_ < k_lo || k_ >= k_hi) break;                                      \
    double* Xs = stage + BA_SXS * (k_ - k_lo);                               \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) Xs[c] = r.x[c];            \
    Xs[6] = ee[0];                                                           \
    Xs[7] = ee[1];
                BA_EDGES(BA_BODY_STAGE_X)
                __syncthreads();
                for (int l = l_lo + tid; l < l_hi; l += BA_THREADS) {
                    double h6[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
                    for (int k = W.pts0[l]; k < W.pts0[l + 1]; ++k) {
                        const double* X = stage + BA_SXS * (k - k_lo);
                        const double e0 = X[6], e1 = X[7];
                        h6[0] += X[0] * X[0] + X[3] * X[3];
                        h6[1] += X[0] * X[1] + X[3] * X[4];
                        h6[2] += X[0] * X[2] + X[3] * X[5];
                        h6[3] += X[1] * X[1] + X[4] * X[4];
                        h6[4] += X[1] * X[2] + X[4] * X[5];
                        h6[5] += X[2] * X[2] + X[5] * X[5];
                        b[0] -= X[0] * e0 + X[3] * e1;
                        b[1] -= X[1] * e0 + X[4] * e1;
                        b[2] -= X[2] * e0 + X[5] * e1;
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) W.Hll[BA_XS * l + i] = h6[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) W.bl[3 * l + i] = b[i];
                    maxdiag = fmax(maxdiag, fmax(fabs(h6[0]), fmax(fabs(h6[3]), fabs(h6[5]))));
                }
                __syncthreads();  // (the staged rows are dead: the area takes the next pass / the rows of the pose-block chains)
            }
        }
        PH_END(2);
        // ================= pose blocks: partial [H_pp | -b_p] = M^T M over the own edges of every free pose, M = the rows
        // [A~ | e~] in storage order: one MFMA per 4 rows (2 edges).  The rows are staged pose group by pose group
        // (as many consecutive poses as the area holds), wave w of a group runs the chains of its poses w, w + 8, ...
        // From the second iteration on the pose-block partials ride along with the first Schur exchange of the iteration
        // (one all-to-all less per iteration); iteration 0 needs them earlier (lambda_0 = tau max |diag H|) and pose-only
        // windows have no Schur exchange: those use the exchange of their own below.
        const bool hp_deferred = do_schur && G > 1 && it > 0;
        const bool hp_local = hp_deferred || G == 1;  // results stay in this workgroup's hpl
        if (!hp_deferred) ++tagH;
        {
            // The rows are staged in `hp_npass` passes (tables made once per solve, above): pass q holds, for EVERY pose, the q-th
            // slice of its rows, all slices of a pass packed back to back.  Every chain therefore runs in every pass -- side by
            // side on different waves -- and carries its accumulator from pass to pass: the same fma chain over the pose's rows in
            // storage order as with one pass.
            const int col = lane & 15;
            v4d accp0 = {0, 0, 0, 0}, accp1 = {0, 0, 0, 0};
            for (int q = 0; q < hp_npass && !error; ++q) {
                const int tq = hp_tables ? q : 0;
                if (!hp_tables) {  // (more passes than the table holds: the pass's row is made now)
                    hp_slice_row(q, 0);
                    __syncthreads();
                }
#define BA_BODY_STAGE_M(el, r, l_, sl_, ee)                                   \
    const int p_ = W.epose[el];                                              \
    const int lo_ = sSlc[0][tq][p_], hi_ = sSlc[1][tq][p_];                  \
    if ((el) < lo_ || (el) >= hi_) break;                                    \
    double* Mr = stage + BA_MSTRIDE * (sSlc[2][tq][p_] + (el)-lo_);          \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                          \
        Mr[c] = r.a0[c];                                                     \
        Mr[7 + c] = r.a1[c];                                                 \
    }                                                                        \
    Mr[6] = ee[0];                                                           \
    Mr[13] = ee[1];
                BA_EDGES(BA_BODY_STAGE_M)
                __syncthreads();
                const bool lastq = q == hp_npass - 1;
                if (batch.use_mfma) {
                    // two free poses share a chain: columns 0..6 of the 16-wide operand are the rows [A~ | e~] of the first,
                    // columns 8..14 those of the second (zero rows once the shorter one has ended: exact no-ops), so the
                    // diagonal 7 x 7 blocks of the product are the two pose blocks, each its own fma chain over its rows
                    int ai = 0;
                    for (int pair = wave; 2 * pair < nfp; pair += BA_WAVES, ++ai) {
                        const int pa = ai == 0 ? hp_pa0 : hp_pa1, pb = ai == 0 ? hp_pb0 : hp_pb1;
                        const bool second = col >= 8;
                        const int pp = second ? pb : pa;
                        const int cc7 = col & 7;
                        const bool cv = cc7 < 7 && pp >= 0;
                        const int rowsA = 2 * (sSlc[1][tq][pa] - sSlc[0][tq][pa]);
                        const int rowsB = pb >= 0 ? 2 * (sSlc[1][tq][pb] - sSlc[0][tq][pb]) : 0;
                        const int s = pp >= 0 ? sSlc[2][tq][pp] : 0, rows = second ? rowsB : rowsA;
                        const int rmax = max(rowsA, rowsB);
                        // row 4 st + k = (edge s + 2 st + (k >> 1), residual row k & 1); edge pitch BA_MSTRIDE, row offset 7
                        const double* pm = stage + BA_MSTRIDE * (s + (lane >> 5)) + 7 * ((lane >> 4) & 1) + (cv ? cc7 : 0);
                        const int kq = lane >> 4;
                        const int myrows = cv ? rows - kq : 0;  // this lane supplies row 4 st + kq of its pose while 4 st < myrows
                        v4d acc = ai == 0 ? accp0 : accp1;
                        int st = 0;
                        // four steps per group; the operands of the NEXT group are fetched right behind the first instruction of a
                        // group (the wave stalls at every dependent MFMA anyway: 64 cycles in which the loads complete)
#define BA_HP_LOAD4(V)                                                                   \
    V##0 = 4 * st < myrows ? pm[0] : 0.0, V##1 = 4 * st + 4 < myrows ? pm[2 * BA_MSTRIDE] : 0.0, \
    V##2 = 4 * st + 8 < myrows ? pm[4 * BA_MSTRIDE] : 0.0, V##3 = 4 * st + 12 < myrows ? pm[6 * BA_MSTRIDE] : 0.0; \
    pm += 8 * BA_MSTRIDE;                                                                \
    st += 4
#define BA_HP_MFMA4_LOADNEXT(V, Wn, more)                                      \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V##0, V##0, acc, 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                          \
    if (more) { BA_HP_LOAD4(Wn); }                                              \
    __builtin_amdgcn_sched_barrier(0);                                          \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V##1, V##1, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V##2, V##2, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V##3, V##3, acc, 0, 0, 0)
                        if (4 * (st + 4) <= rmax) {
                            double a0, a1, a2, a3, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                            BA_HP_LOAD4(a);
                            for (;;) {
                                const bool more1 = 4 * (st + 4) <= rmax;
                                BA_HP_MFMA4_LOADNEXT(a, c, more1);
                                if (!more1) break;
                                const bool more2 = 4 * (st + 4) <= rmax;
                                BA_HP_MFMA4_LOADNEXT(c, a, more2);
                                if (!more2) break;
                            }
                        }
#undef BA_HP_LOAD4
#undef BA_HP_MFMA4_LOADNEXT
                        for (; 4 * st < rmax; ++st) {  // last steps, possibly with fewer than 4 rows
                            const double v = 4 * st < myrows ? pm[0] : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
                            pm += 2 * BA_MSTRIDE;
                        }
                        if (ai == 0) accp0 = acc;
                        else accp1 = acc;
                        if (lastq) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int rg = (lane >> 4) + 4 * j;  // acc[j] = entry (rg, col)
                                const bool mine = second ? (rg >= 8 && rg < 15 && pb >= 0) : rg < 7;
                                const int r7 = rg & 7;
                                if (mine && cc7 <= r7 && cc7 < 7) {
                                    const int sl = sSlot[pp], pk = r7 * (r7 + 1) / 2 + cc7;
                                    if (hp_local) W.hpl[BA_HP * sl + pk] = acc[j];
                                    else gstore_d(B.xH + 2 * ((size_t)g * B.nhp + BA_HP * sl + pk), tag0 + tagH, acc[j], same_l2);
                                }
                            }
                        }
                    }
                } else {
                    // validation path: the same fma chains on the vector ALU, running sums in hpl between the passes
                    for (int p = wave; p < B.F; p += BA_WAVES) {
                        const int sl = sSlot[p];
                        if (sl < 0 || lane >= BA_HP) continue;
                        const int lo = sSlc[0][tq][p], hi = sSlc[1][tq][p];
                        const int s = sSlc[2][tq][p], e = s + hi - lo;
                        int i = 0;
                        while ((i + 1) * (i + 2) / 2 <= lane) ++i;
                        const int j = lane - i * (i + 1) / 2;
                        double acc = q == 0 ? 0.0 : W.hpl[BA_HP * sl + lane];
                        for (int r = 2 * s; r < 2 * e; ++r)
                            acc = __builtin_fma(stage[BA_MSTRIDE * (r >> 1) + 7 * (r & 1) + i], stage[BA_MSTRIDE * (r >> 1) + 7 * (r & 1) + j], acc);
                        if (hp_local || !lastq) W.hpl[BA_HP * sl + lane] = acc;
                        else gstore_d(B.xH + 2 * ((size_t)g * B.nhp + BA_HP * sl + lane), tag0 + tagH, acc, same_l2);
                    }
                }
                __syncthreads();  // (the staged rows are dead)
            }
        }
        PH_END(1);
        // ---- exchange: pose-block partials + the landmark max diagonal, summed in workgroup order; the partials are
        // staged `hrows` workgroups at a time (the staging area is bounded for windows with many workgroups)
        if (!hp_deferred) {
            const double m = block_max(maxdiag, sScr);
            const int nhp = B.nhp, hrows = min(G, max(1, (int)(B.uarea / nhp)));
            double hsum = 0, mm = 0;
            if (G > 1) {
                if (tid == 0) gstore_d(B.xH + 2 * ((size_t)g * nhp + nhp - 1), tag0 + tagH, m, same_l2);
                for (int w0 = 0; w0 < G; w0 += hrows) {
                    const int nr = min(hrows, G - w0);
                    if (!gather_tagged(B.xH + 2 * (size_t)w0 * nhp, nr * nhp, nhp, nhp, tag0 + tagH, stage)) sFlag[2] = 1;
                    __syncthreads();
                    if (tid < nhp - 1) {
#pragma unroll 8
                        for (int w = 0; w < nr; ++w) hsum += stage[w * nhp + tid];
                    }
                    if (tid < nr) mm = fmax(mm, stage[tid * nhp + nhp - 1]);
                    __syncthreads();
                }
                if (sFlag[2]) error = 1;
            } else {
                if (tid < nhp - 1) hsum = W.hpl[tid];
                if (tid == 0) mm = m;
            }
            if (tid < BA_HP * nfree) {
                const int sl = tid / BA_HP, pk = tid - BA_HP * sl, p = sSlotPose[sl];
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= pk) ++i;
                const int j = pk - i * (i + 1) / 2;
                if (i < 6) {
                    sHpp[36 * p + 6 * i + j] = hsum;
                    sHpp[36 * p + 6 * j + i] = hsum;
                } else if (j < 6) {
                    sBp[6 * p + j] = -hsum;
                }
            }
            if (it == 0) {  // computeLambdaInit: tau * max |diag H| over the free vertices
                __syncthreads();
                if (tid < 6 * B.F && sSlot[tid / 6] >= 0) mm = fmax(mm, fabs(sHpp[36 * (tid / 6) + 7 * (tid % 6)]));
                lambda = 1e-5 * block_max(mm, sScr);
                ni = 2;
            }
            __syncthreads();
        }
        PH_END(2);

        double rho = 0;
        int qmax = 0;
        bool hp_pending = hp_deferred;
        do {
            STAMP(0);
            // ============= T1: (H_ll + lambda I)^-1 = C C^T and C^T b_l of the own landmarks
            int ist = nlow;  // where the summed stale-scale entry ends up in Rl
            if (!B.fix_points) {
                for (int l = tid; l < Lg; l += BA_THREADS) {
                    const double* h = W.Hll + BA_XS * l;
                    const double D[9] = {h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda};
                    double Di[9];
                    inv3(D, Di);
                    const double c00 = sqrt(Di[0]), c10 = Di[3] / c00, c20 = Di[6] / c00;
                    const double c11 = sqrt(Di[4] - c10 * c10), c21 = (Di[7] - c20 * c10) / c11;
                    const double c22 = sqrt(Di[8] - c20 * c20 - c21 * c21);
                    double* cc = W.Cc + BA_XS * l;
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
            }
            STAMP(1);
            PH_END(3);
            // ============= T2: partial Schur system of the own landmarks.  U_l = [W_l C_l ; (C_l^T b_l)^T ; 0] (one column per
            // landmark coordinate) is built one CHUNK of columns at a time in the U area and consumed by the MFMA chains of that
            // chunk: one chain per (tile pair, column piece); the pieces of a pair are added in piece order.
            if (do_schur) {
                ++tagA;
                double* split_stage = W.SL;  // (free until the assemble step) npair x (npar - 1) x 256
                // running sums of this wave's first two tile pairs over the chunks (windows with more than 16 tile pairs are
                // planned with one chunk: a wave then publishes every pair straight from the chain)
                v4d run0 = {0, 0, 0, 0}, run1 = {0, 0, 0, 0};
                if (!batch.use_mfma)
                    for (int pk = tid; pk < nlow; pk += BA_THREADS) W.Rl[pk] = 0.0;  // (validation path: running sums in LDS)
                const int nch = chunk_cols > 0 ? max(1, min(nseq, (4 * msteps + chunk_cols - 1) / chunk_cols)) : 1;
                for (int ch = 0; ch < nch; ++ch) {
                    const int c0 = ch * chunk_cols;                      // first column of the chunk
                    const int c1 = min(c0 + chunk_cols, 4 * msteps);     // one past its last column (pad columns included)
                    // ---- fill: rhs row + zero rows (and whole pad columns), zero blocks of (landmark, slot) pairs without an
                    // observation, the blocks A~^T Y of the first observation of every pair
                    for (int q = tid; q < c1 - c0; q += BA_THREADS) {
                        const int colg = c0 + q;
                        double* uc = W.U + (size_t)q * ldu;
                        if (colg < ncol) {
                            uc[n] = W.cl[colg];
                        } else {
                            for (int row = 0; row <= n; ++row) uc[row] = 0.0;
                        }
                        for (int row = n + 1; row < urows; ++row) uc[row] = 0.0;
                    }
                    {
                        const int l0 = c0 / 3, l1 = min(Lg, (c1 + 2) / 3);  // landmarks with a column in the chunk
                        for (int q = l0 * nfree + tid; q < l1 * nfree; q += BA_THREADS) {
                            if (W.eof[q] >= 0) continue;
                            const int l = q / nfree, sl = q - l * nfree;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const int colg = 3 * l + k;
                                if (colg < c0 || colg >= c1) continue;
#pragma unroll
                                for (int c = 0; c < 6; ++c) W.U[u_index(colg - c0, 6 * sl + c, ldu)] = 0.0;
                            }
                        }
                    }
                    if (PROF) {  // (instrumented kernel only: separates the clearing from the per-edge fill in the timeline)
                        __syncthreads();
                        STAMP(2);
                    }
#define BA_BODY_UFILL(el, r, l_, sl_, ee)                                                          \
    const int l = l_, sl = sl_;                                                                   \
    if (sl < 0 || 3 * l + 2 < c0 || 3 * l >= c1 || (has_dups && W.dup[el] != rank)) break;         \
    double Y[6];                                                                                  \
    edge_Y(r, W.Cc + BA_XS * l, Y);                                                               \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                               \
        const int colg = 3 * l + k;                                                               \
        if (!whole && (colg < c0 || colg >= c1)) continue;                                        \
        double* u = W.U + u_index(colg - c0, 6 * sl, ldu);                                        \
        _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                           \
            const double v = r.a0[c] * Y[k] + r.a1[c] * Y[3 + k];                                 \
            if (has_dups && rank > 0) u[c] = u[c] + v;                                            \
            else u[c] = 0.0 + v;                                                                  \
        }                                                                                         \
    }
                    {
                        const bool has_dups = B.max_dup > 0;
                        const bool whole = c0 == 0 && c1 >= ncol;  // (every landmark column lies inside the chunk)
                        for (int rank = 0; rank <= B.max_dup; ++rank) {
                            BA_EDGES(BA_BODY_UFILL)
                            __syncthreads();
                        }
                    }
                    STAMP(3);
                    if (ch == 0 && wave == BA_WAVES - 1) {
                        // computeScale() of the x the solver holds NOW (what g2o scores if this trial's solve fails), landmark part of
                        // this range: lane = landmark mod 64, one 64-lane butterfly.  Read from device memory by the wave that has no
                        // chain in the 5-pose class (3 tile pairs x 2 pieces on 8 waves): the latency hides beside the chains.
                        const double sp = ba_stale_scale_landmarks(BA_AS_GLOBAL_CU64(reinterpret_cast<const u64*>(W.dxl)), (int)(W.bl - dyn), Lg, lambda, lane);
                        if (lane == 0) {
                            sStale[0] = sp;
                            if (G == 1) W.Rl[nlow] = sp;
                        }
                    }
                    // ---- chains of this chunk; after the last chunk the sums are published (or kept when the window has
                    // one workgroup)
                    const int st0 = ch * npar * msplit;  // first MFMA step of the chunk
                    const bool last = ch == nch - 1;
                    if (batch.use_mfma) {
                        const int nchain = B.npair * npar;
                        int ai = 0;
                        for (int a = wave; a < nchain; a += BA_WAVES, ++ai) {
                            const int pr = a % B.npair, sp = a / B.npair;
                            int ti = 0, rem = pr;
                            while (rem >= B.NT - ti) {
                                rem -= B.NT - ti;
                                ++ti;
                            }
                            const int m0 = min(st0 + sp * msplit, msteps) - st0, m1 = min(st0 + (sp + 1) * msplit, msteps) - st0;
                            v4d acc = schur_chain(W.U, ldu, ti, ti + rem, m0, m1, lane);
                            STAMP(4);
                            if (sp > 0) {
                                double* dst = split_stage + ((size_t)pr * (npar - 1) + (sp - 1)) * 256;
#pragma unroll
                                for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = acc[j];
                            }
                            if (npar > 1) __syncthreads();  // (uniform: npair x npar <= 8, every wave makes exactly one trip)
                            if (sp == 0) {
                                v4d r = ai == 0 ? run0 : run1;
#pragma unroll
                                for (int j = 0; j < 4; ++j) r[j] = ch == 0 ? acc[j] : r[j] + acc[j];
                                for (int s2 = 1; s2 < npar; ++s2) {
                                    const double* src = split_stage + ((size_t)pr * (npar - 1) + (s2 - 1)) * 256;
#pragma unroll
                                    for (int j = 0; j < 4; ++j) r[j] += src[lane * 4 + j];
                                }
                                if (ai == 0) run0 = r;
                                else if (ai == 1) run1 = r;
                                if (last) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const int rr = (lane >> 4) + 4 * j, c = lane & 15;
                                        const int pk = a == wave ? pk4[j] : pkt_g[pr * 256 + rr * 16 + c];
                                        if (pk >= 0) {
                                            if (G > 1) gstore_d(B.xP + 2 * ((size_t)g * npk + pk), tag0 + tagA, r[j], grp_l2);
                                            else W.Rl[pk] = r[j];
                                        }
                                    }
                                }
                            }
                        }
                        // waves without a chain still meet the barrier of the split combination
                        if (npar > 1 && wave >= nchain) __syncthreads();
                    } else {  // validation path: the same chains, one packed entry per thread and pass
                        for (int pk = tid; pk < nlow; pk += BA_THREADS) {
                            int i, j;
                            packed_ij(pk, n, i, j);
                            double tot = W.Rl[pk];
                            for (int sp = 0; sp < npar; ++sp) {
                                double acc = 0;
                                const int k0 = min(4 * (st0 + sp * msplit), ncol), k1 = min(4 * (st0 + (sp + 1) * msplit), ncol);
                                for (int colg = k0; colg < k1; ++colg)
                                    acc = __builtin_fma(W.U[u_index(colg - c0, j, ldu)], W.U[u_index(colg - c0, i, ldu)], acc);
                                tot = (ch == 0 && sp == 0) ? acc : tot + acc;
                            }
                            W.Rl[pk] = tot;
                            if (last && G > 1) gstore_d(B.xP + 2 * ((size_t)g * npk + pk), tag0 + tagA, tot, grp_l2);
                        }
                    }
                    if (!last) __syncthreads();  // (the chunk and the split tiles are consumed)
                }
                STAMP(5);
                PH_END(4);
                if (G > 1) {
                    __syncthreads();
                    STAMP(6);
                    // stage 1: this workgroup reduces its SLICE of the packed entries over the partials of its group (the
                    // workgroups w = g mod K; K = 1: all of them), in workgroup order, and republishes the slice.  A window
                    // of more than one XCD's worth of workgroups is planned with one group per XCD (the launch places
                    // workgroup w on XCD w mod K): the G partials -- the bulk of the exchange -- then stay inside their XCD's
                    // L2, and only the K group sums cross the fabric.
                    const int Gk = G / K, gk = g % K, gj = g / K;
                    // (K > 1: the group sums alternate between two buffers -- a trial that ends at the failed factorisation
                    // has no all-to-all behind it, and a slice owner only knows that its OWN group has left the last trial)
                    u64* xR = B.xR + (K > 1 ? 2 * (size_t)(tagA & 1) * K * npk : 0);
                    const int nhpx = hp_pending ? B.nhp - 1 : 0, nlowx = nlow + nhpx + 1;
                    const int slicex = (nlowx + Gk - 1) / Gk;
                    ist = nlow + nhpx;
                    for (int q = tid; q < nhpx; q += BA_THREADS) gstore_d(B.xP + 2 * ((size_t)g * npk + nlow + q), tag0 + tagA, W.hpl[q], grp_l2);
                    if (tid == 0) gstore_d(B.xP + 2 * ((size_t)g * npk + ist), tag0 + tagA, sStale[0], grp_l2);  // (rides along as one more packed entry)
                    const int sl0 = gj * slicex, sln = max(0, min(slicex, nlowx - sl0));
                    if (sln > 0) {
                        // item q = (w, el): partial of the group's workgroup w (= workgroup w K + gk), entry sl0 + el
                        if (!gather_tagged(B.xP + 2 * ((size_t)gk * npk + sl0), sln * Gk, sln, (size_t)K * npk, tag0 + tagA, W.SL, PROF ? &ph[PROF ? 12 : 0] : nullptr)) sFlag[2] = 1;
                    }
                    STAMP(7);
                    __syncthreads();
                    STAMP(8);
                    for (int el = tid; el < sln; el += BA_THREADS) {
                        // sum over the workgroups in order; the operands are fetched eight at a time (left to the
                        // compiler every add waits for its own LDS read: 32 round trips)
                        double sum = 0;
                        int w = 0;
                        for (; w + 8 <= Gk; w += 8) {
                            double t8[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) t8[k] = W.SL[(w + k) * sln + el];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < 8; ++k) sum += t8[k];
                        }
                        for (; w < Gk; ++w) sum += W.SL[w * sln + el];
                        gstore_d(xR + 2 * ((size_t)gk * npk + sl0 + el), tag0 + tagA, sum, same_l2);
                    }
                    STAMP(9);
                    // stage 2: everybody reads the summed entries (K > 1: the K group sums, added in group order)
                    if (K == 1) {
                        if (!gather_tagged(xR, nlowx, nlowx, 0, tag0 + tagA, W.Rl, PROF ? &ph[PROF ? 13 : 0] : nullptr)) sFlag[2] = 1;
                    } else {
                        __syncthreads();  // (the staging of stage 1 may share the U area with this one)
                        if (!gather_tagged(xR, K * nlowx, nlowx, (size_t)npk, tag0 + tagA, stage, PROF ? &ph[PROF ? 13 : 0] : nullptr)) sFlag[2] = 1;
                        __syncthreads();
                        for (int idx = tid; idx < nlowx; idx += BA_THREADS) {
                            double t = 0;
                            for (int k = 0; k < K; ++k) t += stage[k * nlowx + idx];
                            W.Rl[idx] = t;
                        }
                    }
                    STAMP(10);
                    if (hp_pending) {  // the summed pose blocks of this iteration: [H_pp | -b_p] of every free pose
                        __syncthreads();
                        if (tid < BA_HP * nfree) {
                            const double hsum = W.Rl[nlow + tid];
                            const int sl = tid / BA_HP, pk = tid - BA_HP * sl, p = sSlotPose[sl];
                            int i = 0;
                            while ((i + 1) * (i + 2) / 2 <= pk) ++i;
                            const int j = pk - i * (i + 1) / 2;
                            if (i < 6) {
                                sHpp[36 * p + 6 * i + j] = hsum;
                                sHpp[36 * p + 6 * j + i] = hsum;
                            } else if (j < 6) {
                                sBp[6 * p + j] = -hsum;
                            }
                        }
                        hp_pending = false;
                    }
                }
                __syncthreads();
                if (sFlag[2]) error = 1;
            }
            STAMP(11);
            PH_END(5);
            // ============= T3: every workgroup assembles S = H_pp + lambda I - G, g = b_p - G[:, n] -- rows and columns in the
            // pivot order Eigen::LDLT would choose for it (sPerm, from |diag S|) -- and solves it
            int pw = 0;  // the trial's pivot order = sPermAll[pw]: by ranks, or the replayed one when entries tie (an index, not a
                         // pointer: a pointer selected between two LDS arrays is generic to the compiler)
#define perm sPermAll[pw]
            if (n > 0) {
                // dense rank of every |diagonal entry| by counting, all 512 threads: row i = tid / LPR compares its entry with the
                // entries j = jj, jj + LPR, ... (LPR = 512 / P lanes per row, P = 32 / 64 / 128 >= n); every thread forms the entries it
                // needs itself (no hand-off, no barrier before the counting)
                // (the 5-pose class -- NR == 32: n <= 31 -- has its geometry at compile time: 16 lanes per row, two entries per thread)
                const int lsh = NR == 32 ? 4 : (n <= 32 ? 4 : (n <= 64 ? 3 : 2)), LPR = 1 << lsh;
                const int i = tid >> lsh, jj = tid & (LPR - 1);  // (shifts: a division by a run-time value costs ~40 instructions)
// (a macro, not a lambda: behind a by-reference capture the compiler parks the register-resident edge rows in scratch memory)
#define BA_ADIAG(q, hoff) fabs((sHpp[hoff] + lambda) - (do_schur ? W.Rl[(q) * ((q) + 1) / 2 + (q)] : 0.0))
                int v;
                if (NR == 32) {
                    // (two entries per thread, formed before they are compared: their LDS reads are in flight together)
                    const int j0 = jj, j1 = jj + LPR;
                    // (clamped indices, values selected afterwards: no branch between the three chains of LDS reads)
                    const int qi = min(i, n - 1), q0 = min(j0, n - 1), q1 = min(j1, n - 1);
                    const double vi = BA_ADIAG(qi, piv_hoff & 1023), v0 = BA_ADIAG(q0, (piv_hoff >> 10) & 1023), v1 = BA_ADIAG(q1, piv_hoff >> 20);
                    const double ai = i < n ? vi : -1.0, b0 = j0 < n ? v0 : -2.0, b1 = j1 < n ? v1 : -2.0;
                    v = ai != ai ? 1 << 24 : 0;
                    v += (b0 > ai ? 1 : 0) + ((b0 == ai && j0 != i) ? 1 << 12 : 0);
                    v += (b1 > ai ? 1 : 0) + ((b1 == ai && j1 != i) ? 1 << 12 : 0);
                } else {
                    // larger systems (8 or 30 entries per thread): the n entries are formed ONCE, by the first n threads, into the column
                    // buffer (free until the solve), and compared from there -- one barrier more, but the counting loop reads one
                    // double per entry at a trivial address instead of re-deriving it through the slot table
                    double* ab = W.colbuf;
                    if (tid < n) ab[tid] = BA_ADIAG(tid, 36 * sSlotPose[tid / 6] + 7 * (tid % 6));
                    __syncthreads();
                    const double ai = i < n ? ab[i] : -1.0;
                    v = ai != ai ? 1 << 24 : 0;
#pragma unroll 8
                    for (int j = jj; j < n; j += LPR) {
                        const double b = ab[j];
                        v += (b > ai ? 1 : 0) + ((b == ai && j != i) ? 1 << 12 : 0);
                    }
                }
                STAMP(24);
                v = group_sum_i(v, LPR);  // bits 0..11: entries larger than mine, 12..23: entries equal to mine, 24..: NaN
                if (jj == 0 && i < n) {
                    const int rk = v & 0xfff;
                    if (sRank[i] != rk) sFlag[4] = 1;
                    sRank[i] = (short)rk;
                    sPerm[rk] = (short)i;  // (the order itself when no two entries are equal)
                    if (v >> 24) sFlag[1] = 1;
                    else if ((v >> 12) & 0xfff) sFlag[3] = 1;
                }
                STAMP(25);
                __syncthreads();
                STAMP(26);
                if (sFlag[1]) {  // (uniform) NaN: identity -- the factorisation stops at that pivot
                    if (wave == 0)
                        for (int q = lane; q < n; q += 64) sPerm[q] = (short)q;
                    __syncthreads();
                } else if (sFlag[3]) {
                    // equal entries: the order is the replay of Eigen's swaps -- a function of the dense ranks alone, so it is only
                    // replayed when the ranks differ from the last trial's (in a pose-only window they hardly ever do: adding
                    // the damping to every entry keeps their order), else the order of the last replay still stands
                    if (sFlag[4]) {
                        if (wave == 0) ba_pivot_replay(sRank, n, sPermTie, lane);
                        __syncthreads();
                    }
                    pw = 1;
                }
                STAMP(27);
            }
            if (NR != 0) {
                // register solvers: the system embedded into NR rows (identity rows behind n, the rhs as row NR - 1)
                constexpr int NRR = NR ? (NR < 0 ? -NR : NR) : 32, RR = NRR - 1, PP = NRR + 1;
                for (int q = tid; q < NRR * RR; q += BA_THREADS) {
                    const int i = q / RR, k = q - i * RR;
                    double v;
                    if (i < n) {
                        v = 0.0;
                        if (k <= i) {
                            const int a_ = perm[i], b_ = perm[k], hi = max(a_, b_), lo = min(a_, b_);  // entry (hi, lo) of S
                            const double gsum = do_schur ? W.Rl[hi * (hi + 1) / 2 + lo] : 0.0;
                            const int pi = sSlotPose[hi / 6], pj = sSlotPose[lo / 6];
                            v = ((pi == pj) ? sHpp[36 * pi + 6 * (hi % 6) + (lo % 6)] + (hi == lo ? lambda : 0.0) : 0.0) - gsum;
                        }
                    } else if (i == RR) {
                        v = 0.0;
                        if (k < n) {
                            const int kk = perm[k];
                            const double gsum = do_schur ? W.Rl[n * (n + 1) / 2 + kk] : 0.0;
                            v = sBp[6 * sSlotPose[kk / 6] + kk % 6] - gsum;
                        }
                    } else {
                        v = k == i ? 1.0 : 0.0;
                    }
                    W.SL[i * PP + k] = v;
                }
            } else {
                for (int idx = tid; idx < nlow; idx += BA_THREADS) {
                    int i, j;
                    packed_ij(idx, n, i, j);  // j <= i < n, or i == n (rhs row)
                    const int pitch = n + 2;
                    if (i == n) {
                        const int kk = perm[j];
                        const double gsum = do_schur ? W.Rl[n * (n + 1) / 2 + kk] : 0.0;
                        W.SL[n * pitch + j] = sBp[6 * sSlotPose[kk / 6] + kk % 6] - gsum;
                    } else {
                        const int a_ = perm[i], b_ = perm[j], hi = max(a_, b_), lo = min(a_, b_);
                        const double gsum = do_schur ? W.Rl[hi * (hi + 1) / 2 + lo] : 0.0;
                        const int pi = sSlotPose[hi / 6], pj = sSlotPose[lo / 6];
                        W.SL[i * pitch + j] = ((pi == pj) ? sHpp[36 * pi + 6 * (hi % 6) + (lo % 6)] + (hi == lo ? lambda : 0.0) : 0.0) - gsum;
                    }
                }
            }
            __syncthreads();
            STAMP(12);
            PH_END(6);
            if (wave == BA_WAVES - 1) {
                // what g2o's LM would make of THIS trial if the solve below fails: it applies the solver's x all the same -- still
                // the previous solution (sDx / dxl) --, sets tempChi = DBL_MAX and divides by computeScale() of that stale x.  Formed
                // beside the solve (the register solver runs on wave 0 alone): pose part as one 64-lane butterfly + the landmark
                // part that rode along with the Schur exchange, + 1e-3.
                double ps = 0;
                for (int t = lane; t < 6 * B.F; t += 64)
                    if (sSlot[t / 6] >= 0) ps += sDx[t] * (lambda * sDx[t] + sBp[t]);
                ps = wave_sum_d(ps);
                double sc = (do_schur ? W.Rl[ist] : 0.0) + ps;
                sc += 1e-3;
                if (lane == 0) sStale[BA_WAVES] = (currentChi - 1.7976931348623157e308) / sc;
            }
            {
                const int sl_off = (int)(W.SL - dyn), cb_off = (int)(W.colbuf - dyn);
                if (n > 0 && (NR == 64 || NR == -32)) {
                    // workgroup-wide block factorisation (all waves take part: barriers inside)
                    const int ok = solve_block<(NR == 64 ? 64 : 32)>(sl_off, (int)(W.pan - dyn), cb_off + 128, n, tid);
                    if (tid == 0) sFlag[0] = ok;
                    if (ok && tid < n) sSol[perm[tid]] = W.colbuf[128 + tid];  // x = P^T x'; a failed solve leaves x what it was
                } else if (wave == 0 && n > 0) {
                    int ok;
                    if (NR == 32) {
                        ok = solve_wave_32(sl_off, cb_off, n, lane);
                        if (ok && lane < n) sSol[perm[lane]] = W.colbuf[128 + lane];  // x = P^T x'; a failed solve leaves x what it was
                    } else {
                        ok = solve_lds(sl_off, cb_off, n, lane);
                        for (int j = lane; ok && j < n; j += 64) sSol[perm[j]] = W.colbuf[j];
                    }
                    if (lane == 0) sFlag[0] = ok;
                } else if (n == 0 && tid == 0) {
                    sFlag[0] = 1;
                }
            }
            __syncthreads();
#undef perm
            const int ok2 = sFlag[0];
            if (tid == 0) sFlag[1] = sFlag[3] = sFlag[4] = 0;  // (re-armed for the next trial's pivot order: read before the solve, written again many barriers later)
            if (ok2 && tid < 6 * B.F) {
                const int sl = sSlot[tid / 6];
                sDx[tid] = sl >= 0 ? sSol[6 * sl + tid % 6] : 0.0;
            }
            __syncthreads();
            STAMP(13);
            PH_END(7);
            const double lambda_used = lambda;
            ++trials;
            double stale_rho = 0;
            if (!ok2) {
                // The factorisation met a pivot that is not usable (g2o: LDLT "not positive" -> LinearSolverDense::solve returns
                // false WITHOUT touching x, BlockSolver::solve returns before the landmark part).  OptimizationAlgorithmLevenberg
                // applies the solver's x all the same -- still the PREVIOUS solution: sDx / dxl here --, sets tempChi = DBL_MAX and
                // scores the step with computeScale() of that stale x.  DBL_MAX is finite, so the sign of the stale scale decides:
                // positive (the rule) -> rho < 0, rejected; negative -> rho > 0 and the stale step is ACCEPTED (lambda / 3).  Every
                // workgroup solved the same system with the same bits and holds the same summed stale scale (its landmark part rode
                // along with the Schur exchange), so all of them know the outcome at the same time and without another exchange.  On
                // the gauge-free window of the benchmark every rejection is of this kind: lambda shrinks by 1/3 per accepted step
                // until the reduced system stops being numerically positive definite, then bounces.
                stale_rho = sStale[BA_WAVES];  // (formed by the last wave beside the solve, above)
                ++failed_solves;
                if (stale_rho > 0) ++stale_steps;
                if (!(stale_rho > 0)) {
                    // rejected: push / update / pop leave the state as it was -- nothing is applied, evaluated or restored.  The Schur
                    // exchange of the next trial is safe without the chi2 all-to-all in between: a workgroup that holds all summed
                    // entries knows that every workgroup has finished reading the partials (stage 1 precedes the republished
                    // slices), and nobody republishes a slice of the next trial before everybody has published its partials of that
                    // trial, i.e. has left this one.
                    rho = stale_rho;
                    if (B.trace && g == 0 && tid == 0 && trials <= BA_TRACE_MAX) {
                        BaTraceRow tr = {lambda_used, 1.7976931348623157e308, rho, 0.0};
                        B.trace[trials - 1] = tr;
                    }
                    lambda *= ni;
                    ni *= 2;
                    ++qmax;
                    continue;
                }
            }
            // ============= T4/T5: back-substitute the own landmarks, computeScale, push + apply the update
            double scale = 0;
            if (g == 0 && tid < 6 * B.F && sSlot[tid / 6] >= 0) scale += sDx[tid] * (lambda * sDx[tid] + sBp[tid]);
            if (!B.fix_points && do_schur && ok2) {
                // r = C^T (b_l - W^T dx_p) = C^T b_l - sum over the landmark's observations (ascending edge order) of
                // Y^T (A~ dx_pose): every edge thread leaves its three terms in the (now free) U area (zeros for an
                // observation from a fixed pose)
#define BA_BODY_BACKSUB(el, r, l_, sl_, ee)                                     \
    double t3[3] = {0.0, 0.0, 0.0};                                            \
    if (sl_ >= 0) {                                                            \
        const double* sx = sSol + 6 * sl_;                                     \
        double s0 = 0, s1 = 0;                                                 \
        _Pragma("unroll") for (int c = 0; c < 6; ++c) {                        \
            s0 = __builtin_fma(r.a0[c], sx[c], s0);                            \
            s1 = __builtin_fma(r.a1[c], sx[c], s1);                            \
        }                                                                      \
        double Y[6];                                                           \
        edge_Y(r, W.Cc + BA_XS * l_, Y);                                       \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) t3[k] = Y[k] * s0 + Y[3 + k] * s1; \
    }                                                                          \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) W.U[3 * (el) + k] = t3[k];
                BA_EDGES(BA_BODY_BACKSUB)
            }
            STAMP(14);
            __syncthreads();
            STAMP(15);
            // the poses (last wave: push + oplus, a long serial chain) and the landmarks (first waves) update side by side
            if (tid >= BA_THREADS - 64 && tid - (BA_THREADS - 64) < B.F) {
                const int p = tid - (BA_THREADS - 64);
#pragma unroll
                for (int i = 0; i < 8; ++i) sPbak[8 * p + i] = sP[8 * p + i];
                if (sSlot[p] >= 0) {
                    pose_oplus(sP + 8 * p, sDx + 6 * p);
                    quat_to_R(sP + 8 * p, sR + 9 * p);
                    for (int i = 0; i < 3; ++i) sT[3 * p + i] = sP[8 * p + 4 + i];
                }
            }
            if (!B.fix_points) {
                for (int l = tid; l < Lg; l += BA_THREADS) {
                    double r[3] = {W.cl[3 * l], W.cl[3 * l + 1], W.cl[3 * l + 2]};
                    if (do_schur && ok2) {
                        const int k1 = W.pts0[l + 1];
                        for (int k = W.pts0[l]; k < k1; k += 4) {  // (indices, then terms, fetched four edges at a time)
                            int e4[4];
                            double t4[4][3];
#pragma unroll
                            for (int u = 0; u < 4; ++u) e4[u] = k + u < k1 ? W.ptl[k + u] : -1;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int c = 0; c < 3; ++c) t4[u][c] = e4[u] >= 0 ? W.U[3 * e4[u] + c] : 0.0;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (e4[u] >= 0) {
#pragma unroll
                                    for (int c = 0; c < 3; ++c) r[c] = r[c] - t4[u][c];
                                }
                        }
                    }
                    const double* cc = W.Cc + BA_XS * l;
                    double d[3] = {cc[0] * r[0], cc[1] * r[0] + cc[2] * r[1], cc[3] * r[0] + cc[4] * r[1] + cc[5] * r[2]};
                    if (!ok2) {  // (the stale step)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            d[c] = __longlong_as_double((long long)__hip_atomic_load(BA_AS_GLOBAL_CU64(reinterpret_cast<const u64*>(W.dxl)) + 3 * l + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        scale += d[c] * (lambda * d[c] + W.bl[3 * l + c]);
                        // (x of a successful solve; a stale step that is applied leaves x what it is.  Parking the values in the idle U
                        // area and storing them at the top of the next phase was tried: 2.88 vs 2.82 ms per window, profiles/r06_ab_runs.txt)
                        if (ok2) BA_AS_GLOBAL_D(W.dxl)[3 * l + c] = d[c];
                        W.bak[3 * l + c] = W.pts[3 * l + c];
                        W.pts[3 * l + c] += d[c];
                    }
                }
            }
            STAMP(16);
            STAMP(17);
            scale = block_sum(scale, sScr);
            __syncthreads();
            STAMP(18);
            PH_END(8);
            // ============= T6/T7: robust chi2 at the trial state, identical accept / reject decision everywhere
            double tempChi = robust_chi2_local(B, W, Eg, sR, sT, sScr);
            STAMP(19);
            PH_END(9);
            if (G > 1) {
                ++tagB;
                u64* slot = B.xC + (size_t)(tagB & 1) * G * 4;
                if (tid == 0) {
                    gstore_d(slot + 4 * g, tag0 + tagB, tempChi, same_l2);
                    gstore_d(slot + 4 * g + 2, tag0 + tagB, scale, same_l2);
                }
                // every workgroup waits for the tagged partials of ALL workgroups: this is also the barrier that
                // keeps a fast workgroup from overwriting exchange buffers a slow one still reads (a workgroup can
                // be at most one chi2 exchange ahead, hence the two parity slots)
                STAMP(20);
                if (!gather_tagged(slot, 2 * G, 2, 2, tag0 + tagB, sX, PROF ? &ph[PROF ? 14 : 0] : nullptr)) sFlag[2] = 1;
                STAMP(21);
                __syncthreads();
                if (sFlag[2]) error = 1;
                tempChi = 0;
                scale = 0;
                int w = 0;
                for (; w + 8 <= G; w += 8) {  // (operands fetched eight pairs at a time, summed in workgroup order)
                    double t8[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) t8[k] = sX[2 * w + k];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        tempChi += t8[2 * k];
                        scale += t8[2 * k + 1];
                    }
                }
                for (; w < G; ++w) {
                    tempChi += sX[2 * w];
                    scale += sX[2 * w + 1];
                }
            }
            STAMP(22);
            scale += 1e-3;
            const double chi_at_state = tempChi;  // (what the next iteration's computeActiveErrors sees)
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = ok2 ? (currentChi - tempChi) / scale : stale_rho;
            const bool accept = rho > 0 && isfinite(tempChi);
            if (B.trace && g == 0 && tid == 0 && trials <= BA_TRACE_MAX) {
                BaTraceRow tr = {lambda_used, tempChi, rho, accept ? 1.0 : 0.0};
                B.trace[trials - 1] = tr;
            }
            if (accept) {
                double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = chi_at_state;
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
            STAMP(23);
            PH_END(10);
            ++qmax;
        } while (rho < 0 && qmax < 10 && !error);
        if (qmax == 10 || rho == 0 || error) {
            terminated = 1;
            ++it;
            break;
        }
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
    for (int i = tid; i < 3 * Lg; i += BA_THREADS) {
        B.pts_out[3 * (size_t)pt_lo + i] = W.pts[i];
        if (B.h_pts) B.h_pts[3 * (size_t)pt_lo + i] = W.pts[i];
    }
    if (g == 0 && tid == 0) {
        BaStatsDev st;
        st.iterations = it;
        st.trials = trials;
        st.terminated = terminated;
        st.error = error;
        st.failed_solves = failed_solves;
        st.stale_steps = stale_steps;
        st.chi2_initial = chi0;
        st.chi2_final = currentChi;
        st.lambda_final = lambda;
        for (int i = 0; i < BA_NPHASE; ++i) st.phase[i] = 0;
        if (PROF) {
            ph[11] = (long long)__builtin_amdgcn_s_memtime() - ph_start;
            for (int i = 0; i < BA_NPHASE; ++i) st.phase[i] = ph[PROF ? i : 0];
        }
        else st.phase[14] = (long long)__builtin_amdgcn_s_memtime() - ph_start;  // shader cycles of the solve (with solve_ticks: the clock it ran at)
        st.phase[15] = same_l2 ? 1 : 0;
        st.solve_ticks = (long long)(__builtin_amdgcn_s_memrealtime() - t_begin);
        if (PROF && B.trace)
            for (int i = 0; i < 32; ++i) reinterpret_cast<double*>(B.trace + 400)[i] = (double)sStamp[PROF ? i : 0];
        *B.stats = st;
        if (B.h_stats) *B.h_stats = st;
    }
}

// The launch form: window = blockIdx % stride; the stride is a multiple of 8 whenever several windows share a launch: with the
// dispatcher's round-robin placement (block b on XCD b % 8) the workgroups of a window then share one XCD (one L2).
template <bool PROF, int NR, int SLOTS>
__global__ __launch_bounds__(BA_THREADS) void k_ba_lm(BaBatch batch) {
    const int win = blockIdx.x % batch.stride;
    const int g = blockIdx.x / batch.stride;
    if (win >= batch.nwin) return;
    BaRun run = {batch.tag_base[win], batch.use_mfma, batch.same_l2_ok};
    ba_window<PROF, NR, SLOTS>(batch.win[win], run, g);
}

// The resident form (solver service): the grid is launched ONCE and stays; slot = blockIdx % nslots (two slots per XCD
// with 16 slots), workgroup g = blockIdx / nslots of its slot.  Workgroup 0 of a slot polls the slot's mailbox in pinned
// host memory; a new sequence number is a job: it republishes the job to the other workgroups of the slot through a word
// pair in device memory (same XCD: they poll its L2), everybody solves the window -- inputs are read straight from the
// pinned upload image, results go to the pinned mirrors as always --, the workgroups count themselves off, workgroup 0 writes
// the job's sequence number into the window's pinned status word and the slot waits for the next job.  A slot leaves when
// the host posts `stop`, or when neither a job nor a heartbeat of the host's scheduler has arrived for BA_SERVICE_IDLE_TICKS
// of the 100 MHz clock (safety net: the grid must never outlive its host -- and no slot may leave while the host lives).
template <int NR, int SLOTS>
__global__ __launch_bounds__(BA_THREADS) void k_ba_service(BaServiceArgs a) {
    const int slot = blockIdx.x % a.nslots;
    const int g = blockIdx.x / a.nslots;
    __shared__ unsigned long long sJob[4];  // seq, desc, (tag0 | use_mfma << 32), stop
    volatile BaMail* mail = a.mail + slot;
    u64* cmd = a.cmd + 8 * (size_t)slot;    // device memory: {seq, desc, flags, stop} republished by workgroup 0
    u64* arrived = a.arrived + slot;        // device memory: workgroups that finished the current job (monotonic)
    unsigned long long last = a.first_seq[slot];
    unsigned long long idle0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long last_beat = ~0ull;
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned long long seq, stop = 0, desc = 0, flags = 0;
            for (;;) {
                if (g == 0) {
                    seq = __hip_atomic_load(&mail->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    stop = __hip_atomic_load(&mail->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (!stop && seq == last && __builtin_amdgcn_s_memrealtime() - idle0 > BA_SERVICE_IDLE_TICKS) {
                        // nothing for this slot for a long time: leave only if the HOST has gone quiet as well (its scheduler
                        // bumps `beat` in every mailbox while the grid is resident) -- a slot that left on its own while the
                        // grid stays would swallow the next window posted to it
                        const unsigned long long beat = __hip_atomic_load(&mail->beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (beat != last_beat) {
                            last_beat = beat;
                            idle0 = __builtin_amdgcn_s_memrealtime();
                        } else {
                            stop = 1;
                        }
                    }
                    if (seq != last || stop) {
                        desc = __hip_atomic_load(&mail->desc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        flags = __hip_atomic_load(&mail->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(cmd + 1, desc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(cmd + 2, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(cmd + 3, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm_waitcnt_vm0();
                        __hip_atomic_store(cmd, stop ? ~0ull : seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                } else {
                    seq = __hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seq != last) {
                        desc = __hip_atomic_load(cmd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flags = __hip_atomic_load(cmd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stop = __hip_atomic_load(cmd + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(8);
            }
            sJob[0] = seq, sJob[1] = desc, sJob[2] = flags, sJob[3] = stop;
        }
        __syncthreads();
        const unsigned long long seq = sJob[0], desc = sJob[1], flags = sJob[2], stop = sJob[3];
        __syncthreads();
        if (stop) return;
        last = seq;
        BaRun run = {(unsigned)flags, (int)((flags >> 32) & 1), (int)((flags >> 33) & 1)};
        const BaDev* D = reinterpret_cast<const BaDev*>(desc);
        // (inlined on purpose: called out of line -- an allocation of its own, one call per window -- the body ran 5 % slower,
        // 2.97 vs 2.83 ms per window, profiles/r06_ab_runs.txt; the spill COUNT of this kernel is not what its time follows)
        ba_window<false, NR, SLOTS>(D, run, g);
        // ---- completion: every workgroup of the slot counts itself off once its results are on their way to the host (every
        // wave drains its own stores first); workgroup 0 waits for all of them and then posts the job's sequence number
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0) {
                const unsigned long long want = (unsigned long long)a.wgs_per_slot * (seq - a.first_seq[slot]);
                while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
                __hip_atomic_store(&mail->done_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        idle0 = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------ launch
typedef void (*ba_kernel_fn)(BaBatch);
template <bool PROF, int SLOTS>
static ba_kernel_fn ba_kernel_pick(int nr) {
    return nr == 32 ? k_ba_lm<PROF, 32, SLOTS> : (nr == -32 ? k_ba_lm<PROF, -32, SLOTS> : (nr == 64 ? k_ba_lm<PROF, 64, SLOTS> : k_ba_lm<PROF, 0, SLOTS>));
}
static ba_kernel_fn ba_kernel_for(int profile, int nr, int slots) {
    if (profile) return slots == 0 ? ba_kernel_pick<true, 0>(nr) : (slots > 1 ? ba_kernel_pick<true, 2>(nr) : ba_kernel_pick<true, 1>(nr));
    return slots == 0 ? ba_kernel_pick<false, 0>(nr) : (slots > 1 ? ba_kernel_pick<false, 2>(nr) : ba_kernel_pick<false, 1>(nr));
}
int ba_kernel_set_lds_limit() {
    int bad = 0;
    for (int profile = 0; profile < 2; ++profile)
        for (int nr : {32, -32, 64, 0})
            for (int slots = 0; slots <= 2; ++slots)
                bad |= hipFuncSetAttribute((const void*)ba_kernel_for(profile, nr, slots), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           BA_LDS_BUDGET) != hipSuccess;
    return bad ? -1 : 0;
}
hipError_t ba_service_launch(const BaServiceArgs& a, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)k_ba_service<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_BUDGET);
    hipLaunchKernelGGL((k_ba_service<32, 2>), dim3(a.nslots * a.wgs_per_slot), dim3(BA_THREADS), BA_LDS_BUDGET, stream, a);
    return hipGetLastError();
}
// solver class of a window with n unknowns (windows of one launch share it)
int ba_solver_class(int n) { return n + 1 <= 32 ? 32 : (n + 1 <= 64 ? 64 : 0); }
hipError_t ba_kernel_launch(const BaBatch& batch, int max_wgs, size_t lds_bytes, hipStream_t stream, int profile, int nr, int slots) {
    hipLaunchKernelGGL(ba_kernel_for(profile, nr, slots), dim3(batch.stride * max_wgs), dim3(BA_THREADS), lds_bytes, stream, batch);
    return hipGetLastError();
}
