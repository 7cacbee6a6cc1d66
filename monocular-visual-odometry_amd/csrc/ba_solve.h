// csrc/ba_solve.h -- the one-wave solver of the reduced (6F x 6F, F <= 5) system of the bundle adjustment: g2o's
// LinearSolverDense on the Schur complement (src/optimization/g2o_ba.cpp:193-200 builds that stack).  Included by
// ba_kernels.hip (the LM loop) and by tools/probes/solve_probe.hip (cycles per solve of the variants, on their own).
#ifndef MVO_BA_SOLVE_H
#define MVO_BA_SOLVE_H

typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef BA_WAVES
#define BA_WAVES 8  // (ba_types.h)
#endif

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
// row n = the rhs g^T: the lower triangle of the symmetric matrix [[S, g], [g^T, .]].  The caller has ASSEMBLED S in the order in which
// Eigen::LDLT takes its rows (round 6: ba_window forms that pivot order from |diag S| before the assembly; the solution is scattered
// back through it), so the elimination itself takes the rows as they come.  Right-looking LDL^T,
// canonical arithmetic per step j: r = 1 / d_j, l_i = c_i r, a_ik = fma(-l_i, c_k, a_ik) (c = column j).
// Lane i keeps row i in registers; the system is embedded into NR - 1 rows (identity rows behind n: exact no-ops) with
// the rhs as row NR - 1, so that the whole elimination is straight-line code: at step j every lane first finishes
// its entry of column j + 1, parks it in LDS and fetches the pivot with v_readlane -- the division of step j + 1 and
// the LDS round trip of its column overlap with the remaining updates of step j.  The rhs row comes out as
// z = D^-1 L^-1 g.  L is written transposed (row j = column j of L) over SL; x = L^-T z by a column sweep with
// v_readlane broadcasts: x_j = fma(-l_ij, x_i, x_j) for i = n-1 .. j+1.  Returns 0 when a pivot is not usable -- negative
// (Eigen: "not positive" -> LinearSolverDense::solve returns false, the LM loop then applies and scores the solver's stale x) or outside
// BA_PIVOT_MIN .. BA_PIVOT_MAX (never met; Eigen would carry on with such a pivot).
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
            ok &= (d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX);  // (the identity rows behind n have the pivot 1)
        }
#pragma unroll
        for (int m = mn + 1; m < H; ++m) a[m] = __builtin_fma(-l, ck[cur][m], a[m]);
        SL[j * P + i] = l;  // column j of L (entries of the rows <= j are never read)
        const double ln = cin * rn;
        __builtin_amdgcn_sched_barrier(0);
        r = rn;
        l = ln;
    }
    // A factorisation that met an unusable pivot ends here: the trial is rejected whatever x would be (on the benchmark's
    // gauge-free window 37 of 87 trials end this way, all at one of the last six pivots).
    if (__builtin_amdgcn_readfirstlane(ok) == 0) return 0;
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

// The same factorisation as a BLOCK algorithm run by the whole workgroup, matrix in LDS (SL, NR rows at pitch NR + 1,
// embedded as above: identity rows behind n, the rhs as row NR - 1).  Per block of 4 columns j0 .. j0 + 3:
//   panel   (wave 0, lane = row): the four pivots one after the other, r = 1 / d, l = c r, the remaining panel columns
//           updated in registers; the rows [-l] and [c] of the panel go to LDS in MFMA operand order, the l into the matrix;
//   update  (16 x 16 tiles): S_tile += (-L_panel) C_panel^T by ONE v_mfma_f64_16x16x4_f64 -- the four columns of the
//           panel are the four k slots, added as fused multiply-adds in column order, which is exactly the canonical
//           a_ik = fma(-l_ij, c_kj, a_ik), j ascending; only entries inside the lower triangle are written back.
// The two are software-pipelined (look-ahead): the panel of block b + 1 only needs the four columns behind block b, so
// wave 0 applies update b to that strip itself and goes straight on to factor panel b + 1 into the other panel buffer,
// while waves 1 .. 7 apply update b to everything right of the strip -- one workgroup barrier per block, the serial part
// runs beside the update instead of before it.
// A wave issues one instruction every ~5 cycles and a v_readlane into a vector operand costs ~17 (block_solve_probe):
// the loop is written for instruction count.  Wave 0 keeps its row's -l and c of the last panel in registers and updates
// its strip entries with plain fma's, the c of the four strip rows coming as LDS broadcast reads of the panel buffer (the
// same fma's in the same order as the MFMA's k slots); the 4 x 4 diagonal block goes through 16 doubles of LDS once and
// EVERY lane then eliminates it redundantly (the same operations on the same values as the four lanes that own those rows
// perform on their registers: pivots, reciprocals and multipliers without one cross-lane operation).  The other waves own
// at most two tiles each, fixed for the whole factorisation (tile e of the column-major list of the lower triangle goes
// to wave 1 + e mod 7), with their LDS offsets and write-back thresholds computed once.
// Entries above the diagonal are never read for a result (tile loads fetch them, nothing stores what comes out of them),
// so the panel writes its l without a row test.
// Same bits as the scalar right-looking LDL^T (every entry receives the same fma's in the same order).  `pan`:
// BA_PANEL_DOUBLES of scratch (two panel buffers of 8 NR, the diagonal block).  Returns 0 when a pivot is not usable;
// leaves x in xout[0 .. n).
// (`STAMPS`: tools/probes/block_solve_probe.hip -- wave 0 adds the cycles of its stages to st[0 .. 4), the others theirs to st[4 .. 7).)
#define BA_SB_STAMP(k)                                                     \
    if (STAMPS) {                                                          \
        const long long sb_n = (long long)__builtin_amdgcn_s_memtime();    \
        st[k] += sb_n - sb_t;                                              \
        sb_t = sb_n;                                                       \
    }
template <int NR, bool STAMPS = false>
__device__ __forceinline__ int solve_block(int sl_off, int pan_off, int xout_off, int n, int tid, long long* st = nullptr) {
    constexpr int R = NR - 1, P = NR + 1, NB = NR / 4, NTL = NR / 16, NTILES = NTL * (NTL + 1) / 2;
    static_assert(NTILES <= 2 * (BA_WAVES - 1), "two tiles per updating wave");
    double* SL = ba_dyn_lds + sl_off;
    double* PAN = ba_dyn_lds + pan_off;  // buffer (b & 1): NR x 4 of -l_i,j0+k, then NR x 4 of c_i,j0+k
    double* DB = PAN + 16 * NR;          // the diagonal block of the panel being factored (4 x 4)
    double* xout = ba_dyn_lds + xout_off;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ok = 1;
    if (wave == 0) {
        const int i = lane < NR ? lane : R;
        double* rowp = SL + i * P;
        double nl[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};  // this row's -l and c of the panel factored last
#pragma unroll 1
        for (int b = -1; b + 1 < NB; ++b) {  // iteration b: update b on the strip (none for b = -1), then panel b + 1
            long long sb_t = STAMPS ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const int js = 4 * b + 4;  // the panel's columns js .. js + 3
            double p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = rowp[js + k];  // (updates < b: applied by the others before the last barrier)
            if (b >= 0) {
                const double* CB = PAN + (b & 1) * 8 * NR + 4 * NR + 4 * js;  // c of the rows js .. js + 3 in panel b
                double cb[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) cb[k][jj] = CB[4 * k + jj];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int k = 0; k < 4; ++k) p[k] = __builtin_fma(nl[jj], cb[k][jj], p[k]);
            }
            BA_SB_STAMP(0);
            if ((unsigned)(lane - js) < 4u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) DB[4 * (lane - js) + k] = p[k];
            }
            __builtin_amdgcn_wave_barrier();
            double D[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int k = 0; k <= m; ++k) D[m][k] = DB[4 * m + k];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = js + jj;
                double d = D[jj][jj];
                if (j >= R) d = 1.0;  // (column R of the last block: never a pivot)
                ok &= (d >= BA_PIVOT_MIN) & (d <= BA_PIVOT_MAX);  // (the identity rows behind n have the pivot 1)
                const double r = ba_rcp_pivot(d);
                c[jj] = p[jj];
                nl[jj] = -(p[jj] * r);
#pragma unroll
                for (int kk = jj + 1; kk < 4; ++kk) p[kk] = __builtin_fma(nl[jj], D[kk][jj], p[kk]);  // D[kk][jj]: S[js + kk][j] before the scaling
#pragma unroll
                for (int m = jj + 1; m < 4; ++m) {  // the rows js + m of the diagonal block, as their own lanes update them
                    const double nlm = -(D[m][jj] * r);
#pragma unroll
                    for (int kk = jj + 1; kk <= m; ++kk) D[m][kk] = __builtin_fma(nlm, D[kk][jj], D[m][kk]);
                }
            }
            BA_SB_STAMP(1);
            if (lane < NR) {
                double* LPn = PAN + ((b + 1) & 1) * 8 * NR + 4 * i;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    LPn[k] = nl[k];
                    LPn[4 * NR + k] = c[k];
                    rowp[js + k] = -nl[k];
                }
            }
            BA_SB_STAMP(2);
            __syncthreads();
            BA_SB_STAMP(3);
        }
    } else {
        // this wave's tiles: e = wave - 1 and e + 7 of (0,0) (0,1) .. (0,NTL-1) (1,1) .. as (tile column, tile row)
        const int q = lane >> 4, cidx = lane & 15;
        int tj[2] = {NTL, NTL}, ti[2] = {NTL, NTL};
        {
            int e = 0;
            for (int cj = 0; cj < NTL; ++cj)
                for (int ci = cj; ci < NTL; ++ci, ++e)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (e == wave - 1 + 7 * u) {
                            tj[u] = cj;
                            ti[u] = ci;
                        }
        }
        int acc_off[2], a_off[2], b_off[2], thr[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            acc_off[u] = (16 * ti[u] + q) * P + 16 * tj[u] + cidx;
            a_off[u] = (16 * ti[u] + cidx) * 4 + q;
            b_off[u] = 4 * NR + (16 * tj[u] + cidx) * 4 + q;
            const int col = 16 * tj[u] + cidx;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) thr[u][rr] = (col <= 16 * ti[u] + 4 * rr + q && col < R) ? col : -1;  // written back while j0 + 7 < thr
        }
#pragma unroll 1
        for (int b = -1; b + 1 < NB; ++b) {
            if (b >= 0) {
                long long sb_t = STAMPS ? (long long)__builtin_amdgcn_s_memtime() : 0;
                const int lim = 4 * b + 7, t0 = (4 * b + 8) >> 4;  // update b right of the strip: columns > j0 + 7
                const double* PB = PAN + (b & 1) * 8 * NR;
                v4d acc[2];
                double av[2], bv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (tj[u] < t0 || tj[u] >= NTL) continue;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[u][rr] = SL[acc_off[u] + 4 * rr * P];
                    av[u] = PB[a_off[u]];
                    bv[u] = PB[b_off[u]];
                }
                BA_SB_STAMP(4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (tj[u] < t0 || tj[u] >= NTL) continue;
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc[u], 0, 0, 0);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (lim < thr[u][rr]) SL[acc_off[u] + 4 * rr * P] = acc[u][rr];
                }
                BA_SB_STAMP(5);
                __syncthreads();
                BA_SB_STAMP(6);
            } else {
                __syncthreads();
            }
        }
    }
    // back-substitution x = L^-T z by wave 0: lane j owns x_j (z = the rhs row of L); row i of L is read in LDS order
    // (not after an unusable pivot: the trial is rejected whatever x would be)
    if (wave == 0 && __builtin_amdgcn_readfirstlane(ok) != 0) {
        const int j = lane < R ? lane : 0;
        double x = SL[R * P + j];
        for (int i0 = R - 1; i0 >= 1; i0 -= 8) {
            double li[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) li[u] = i0 - u >= 1 ? SL[(i0 - u) * P + j] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 - u;
                if (i < 1) break;
                const double xi = readlane_d(x, i);
                const double t = __builtin_fma(-li[u], xi, x);
                x = lane < i ? t : x;
            }
        }
        if (lane < n) xout[lane] = x;
    }
    return __builtin_amdgcn_readfirstlane(ok);
}

#endif
