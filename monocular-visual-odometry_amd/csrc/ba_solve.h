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
// Solves the reduced (n x n) system with ONE wave.  SL (LDS, row pitch NR + 1) holds row i = S[i][0..i] for i < n and
// row n = the rhs g^T: the lower triangle of the symmetric matrix [[S, g], [g^T, .]].  Right-looking LDL^T without
// pivoting, canonical arithmetic per step j: r = 1 / d_j, l_i = c_i r, a_ik = fma(-l_i, c_k, a_ik) (c = column j).
// Lane i keeps row i in registers; the system is embedded into NR - 1 rows (identity rows behind n: exact no-ops) with
// the rhs as row NR - 1, so that the whole elimination is straight-line code: at step j every lane first finishes
// its entry of column j + 1, parks it in LDS and fetches the pivot with v_readlane -- the division of step j + 1 and
// the LDS round trip of its column overlap with the remaining updates of step j.  The rhs row comes out as
// z = D^-1 L^-1 g.  L is written transposed (row j = column j of L) over SL; x = L^-T z by a column sweep with
// v_readlane broadcasts: x_j = fma(-l_ij, x_i, x_j) for i = n-1 .. j+1.  Returns 0 when a pivot is not positive
// (g2o: LDLT "not positive" -> the step is rejected).
// A step is bound by the ~45 instructions ONE wave has to issue for it (8999 cycles per solve = 290 per pivot,
// tools/probes/solve_probe.hip).  Round 4 tried to take the LDS hand-off of the finished column off the path from one pivot to
// the next -- pivot and next-column entry forwarded as scalars (three v_readlane pairs per step), the other half's c_i through
// v_permlane32_swap -- : bit-identical and SLOWER (13016 cycles): the extra scalar moves and exec-masked updates cost more
// issue slots than the hand-off costs latency.  That variant lives on in the probe for the record.
// The 32-row flavour uses both halves of the wave: lane = (row i = lane & 31, half h = lane >> 5) keeps the columns
// k = 2 m + h of its row (16 registers instead of 32), so that the prefetched column of the next step fits the register
// file next to the one in use, and a step costs half the fused multiply-adds per lane.  Same arithmetic per entry.
__device__ __forceinline__ int solve_wave_32(int sl_off, int cb_off, int n, int lane) {
    constexpr int NR = 32, R = 31, P = 33, H = 16;
    double* SL = ba_dyn_lds + sl_off;
    double* colbuf = ba_dyn_lds + cb_off;  // 2 buffers x (2 halves x 32 rows)
    double* xout = colbuf + 128;
    const int i = lane & 31, h = lane >> 5;
    int ok = 1;
    double a[H];
#pragma unroll
    for (int m = 0; m < H; ++m) a[m] = SL[i * P + 2 * m + h];
    double ck[2][H];
    // column 0 lives in half 0, register 0
    colbuf[h * 32 + i] = a[0];
    double d = readlane_d(a[0], 0);
    double ci = colbuf[i];
#pragma unroll
    for (int m = 0; m < H; ++m) ck[0][m] = colbuf[2 * m + h];
    double r = ba_rcp_pivot(d);
    ok &= (d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX);
    double l = ci * r;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        const int jn = j + 1, hn = jn & 1, mn = jn >> 1;
        double cin = 0;
        // region A: finish the entries of column j + 1, hand them to the other lanes, start fetching that column
        if (jn < R) {
            a[mn] = __builtin_fma(-l, ck[cur][mn], a[mn]);  // (half hn: column j + 1; other half: column j or j + 2)
            colbuf[nxt * 64 + h * 32 + i] = a[mn];
            d = readlane_d(a[mn], jn + 32 * hn);
            cin = colbuf[nxt * 64 + hn * 32 + i];
#pragma unroll
            for (int m = mn + 1; m < H; ++m) ck[nxt][m] = colbuf[nxt * 64 + hn * 32 + 2 * m + h];
            if (mn + 1 < H || true) ck[nxt][mn] = colbuf[nxt * 64 + hn * 32 + ((2 * mn + h) & 31)];
        }
        __builtin_amdgcn_sched_barrier(0);
        // region B: the rest of step j; the division of step j + 1 rides along
        double rn = 0;
        if (jn < R) {
            rn = ba_rcp_pivot(d);
            ok &= ((d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX)) | (jn >= n);
        }
#pragma unroll
        for (int m = mn + 1; m < H; ++m) a[m] = __builtin_fma(-l, ck[cur][m], a[m]);
        SL[j * P + i] = l;  // column j of L (entries of the rows <= j are never read)
        const double ln = cin * rn;
        __builtin_amdgcn_sched_barrier(0);
        r = rn;
        l = ln;
    }
    // back-substitution: lane j (< 31) owns x_j  (intra-wave hand-off of L^T through LDS, see solve_wave)
    __builtin_amdgcn_wave_barrier();
    double cl[NR];
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
