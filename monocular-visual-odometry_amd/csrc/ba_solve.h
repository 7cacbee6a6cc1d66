// csrc/ba_solve.h -- the one-wave solver of the reduced (6F x 6F, F <= 5) system of the bundle adjustment: g2o's
// LinearSolverDense on the Schur complement (src/optimization/g2o_ba.cpp:193-200 builds that stack).  Included by
// ba_kernels.hip (the LM loop) and by tools/probes/solve_probe.hip (cycles per solve of the variants, on their own).
#ifndef MVO_BA_SOLVE_H
#define MVO_BA_SOLVE_H

__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
// The value of lane (lane & 31) + 32 * half in every lane: the 32 values of one half of the wave, seen by both halves.
// v_permlane32_swap_b32 exchanges the upper half of its first operand with the lower half of its second one; with the
// same value in both, the first result holds the lower half twice and the second one the upper half twice.  A vector-ALU
// operation: no LDS round trip.  `half` must be a compile-time constant.
__device__ __forceinline__ double half_bcast_d(double v, int half) {
#ifndef MVO_KERNEL_SIM
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[half], (int)l2[half]);
#else
    return __shfl(v, (emu_lane() & 31) + 32 * half);
#endif
}

// Correctly rounded 1 / d for the pivots of the reduced solve.  The compiler's IEEE f64 division is
//   s = div_scale(d), n = div_scale(1), r = rcp(s), two Newton steps on r, q = n r, e = fma(-s, q, n), q = div_fmas(e, r, q),
//   div_fixup(q)
// -- ten dependent operations, three of which (the two scalings and the fix-up) only act on operands near the ends of
// the exponent range, zeros, infinities and NaNs.  For a pivot inside BA_PIVOT_MIN .. BA_PIVOT_MAX they are identities
// (scale factor 1, nothing to fix), so the shorter chain below computes the SAME intermediate values and returns the same
// bits (every bitwise BA test runs through it).  A pivot outside that range is not a usable pivot: the solvers' "not
// positive" check rejects the step (the blocked oracle has the same rule), so what this function returns for it never
// reaches a result.  A range test with a fall-back to the generic division was measured too: the branch breaks the
// overlap of the reciprocal with the row updates and costs more than the three operations save.
#define BA_PIVOT_MIN 0x1p-500
#define BA_PIVOT_MAX 0x1p+500
__device__ __forceinline__ double ba_rcp_pivot(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(e, r, r);
}

// (The LDS areas are passed as offsets into the dynamic segment: generic pointers into LDS handed to an out-of-line
// function make hipcc 7.2 emit an illegal v_cmp against src_shared_base.)
#ifndef MVO_KERNEL_SIM
extern __shared__ __attribute__((aligned(16))) double ba_dyn_lds[];
#else  // tests/sim: this source compiled for the host against tests/sim/hip_emu (a test aid; the library has no CPU path)
#define ba_dyn_lds (static_cast<double*>(emu_dyn_lds()))
#endif

// ------------------------------------------------------------------------------------------------ reduced solve
// Solves the reduced (n x n, n <= 30) system with ONE wave.  SL (LDS, row pitch 33) holds row i = S[i][0..i] for i < n and
// row 31 = the rhs g^T: the lower triangle of the symmetric matrix [[S, g], [g^T, .]], embedded into 31 rows (identity
// rows behind n: exact no-ops).  Right-looking LDL^T without pivoting, canonical arithmetic per step j:
//   r = 1 / d_j,  l_i = c_i r,  a_ik = fma(-l_i, c_k, a_ik)   (c = column j).
// Lane (i = lane & 31, h = lane >> 5) keeps the entries k = 2 m + h of row i in registers (16 per lane: both halves of the
// wave work on a row).  A single wave issues in order, and a step is bound by its instruction count and by three hand-offs;
// the form below (round 4) keeps them off the path from one pivot to the next:
//   * the pivot d_{j+1} and the entry c_{j+2} of the freshly finished column j + 1 that the NEXT step's first update needs
//     travel as scalars (v_readlane): the chain pivot -> reciprocal -> l -> first update of the next column -> next pivot
//     never waits for LDS;
//   * a lane's c_i of the other half's column comes through v_permlane32_swap (vector ALU) instead of an LDS read-back;
//   * the finished column still goes through LDS for the bulk of the next step's updates (15 .. 1 per lane), de-interleaved
//     (even rows | odd rows) so that a lane's operands are consecutive and come two per read; they are fetched a whole
//     step before they are used.
// The rhs row comes out as z = D^-1 L^-1 g.  L is written transposed (row j = column j of L) over SL; x = L^-T z by a
// column sweep with v_readlane broadcasts: x_j = fma(-l_ij, x_i, x_j) for i = n-1 .. j+1.  Returns 0 when a pivot is not
// usable (g2o: LDLT "not positive" -> the step is rejected).  Same operations per entry, in the same order, as the
// round-3 form (tests/sim and the bitwise GPU tests hold both to the blocked oracle).
__device__ __forceinline__ int solve_wave_32(int sl_off, int cb_off, int n, int lane) {
    constexpr int R = 31, P = 33, H = 16;
    double* SL = ba_dyn_lds + sl_off;
    double* colbuf = ba_dyn_lds + cb_off;  // 2 buffers x 32: the finished column, rows (even | odd)
    double* xout = colbuf + 128;
    const int i = lane & 31, h = lane >> 5;
    const int pos = (i & 1) * 16 + (i >> 1);  // where row i's entry of a column goes
    int ok = 1;
    double a[H];
#pragma unroll
    for (int m = 0; m < H; ++m) a[m] = SL[i * P + 2 * m + h];
    double ck[2][H];
    // column 0 lives in half 0, register 0
    double d = readlane_d(a[0], 0);
    double sc = readlane_d(a[0], 1);  // c_1: row 1's entry of column 0
    if (h == 0) colbuf[pos] = a[0];
    double ci = half_bcast_d(a[0], 0);
#pragma unroll
    for (int m = 0; m < H; ++m) ck[0][m] = colbuf[h * 16 + m];
    double r = ba_rcp_pivot(d);
    ok &= (d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX);
    double l = ci * r;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        const int jn = j + 1, hn = jn & 1, mn = jn >> 1;
        double cin = 0, scn = 0;
        // region A: finish column j + 1 (its entries live in half hn, register mn), send pivot and next entry ahead as
        // scalars, hand the column to the other lanes
        if (jn < R) {
            if (h == hn) a[mn] = __builtin_fma(-l, sc, a[mn]);
            d = readlane_d(a[mn], jn + 32 * hn);
            scn = readlane_d(a[mn], jn + 1 + 32 * hn);  // c_{j+2}: row j + 2's entry of column j + 1
            if (h == hn) colbuf[nxt * 32 + pos] = a[mn];
            cin = half_bcast_d(a[mn], hn);
            // operands of the next step: the other half's register mn' when it holds column j + 3, and everything behind
            const int mq = (jn + 1) >> 1;  // = mn' of the next step
#pragma unroll
            for (int m = mq; m < H; ++m) ck[nxt][m] = colbuf[nxt * 32 + h * 16 + m];
        }
        __builtin_amdgcn_sched_barrier(0);
        // region B: the rest of step j; the division of step j + 1 rides along
        double rn = 0;
        if (jn < R) {
            rn = ba_rcp_pivot(d);
            ok &= ((d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX)) | (jn >= n);
        }
        // (the half that does not hold column j + 1 in register mn holds column j -- finished -- or column j + 2 there)
        if (hn == 0 && h == 1) a[mn] = __builtin_fma(-l, ck[cur][mn], a[mn]);
#pragma unroll
        for (int m = mn + 1; m < H; ++m) a[m] = __builtin_fma(-l, ck[cur][m], a[m]);
        SL[j * P + i] = l;  // column j of L (entries of the rows <= j are never read)
        const double ln = cin * rn;
        __builtin_amdgcn_sched_barrier(0);
        r = rn;
        l = ln;
        sc = scn;
    }
    // back-substitution: lane j (< 31) owns x_j  (intra-wave hand-off of L^T through LDS)
    __builtin_amdgcn_wave_barrier();
    double cl[32];
    const int lj = i < R ? i : 0;
#pragma unroll
    for (int q = 1; q < R; ++q) cl[q] = SL[lj * P + q];
    double x = SL[lj * P + R];
#pragma unroll
    for (int q = R - 1; q >= 1; --q) {
        const double xi = readlane_d(x, q);
        const double t = __builtin_fma(-cl[q], xi, x);
        x = i < q ? t : x;
    }
    if (lane < n) xout[lane] = x;
    return __builtin_amdgcn_readfirstlane(ok);
}

#endif
