// Dev probe: cycles per reduced solve (31-row embedded LDL^T on one wave) of the product's solver (csrc/ba_solve.h) and of
// a variant that forwards pivot / next-column entry as scalars (measured slower: 13016 vs 8999 cycles), results compared bit
// for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../monocular-visual-odometry_amd/csrc -o solve_probe solve_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "ba_solve.h"

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
__device__ __forceinline__ int solve_wave_32_scalar_forwarding(int sl_off, int cb_off, int n, int lane) {
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

// SL image (32 x 33) in global memory -> LDS, solve, x out; `which` 0 = round 3, 1 = round 4
__global__ __launch_bounds__(512) void probe(const double* img, double* xo, long long* cyc, int* okv, int n, int which, int reps) {
    double* SL = ba_dyn_lds;
    double* colbuf = ba_dyn_lds + 32 * 33 + 7;  // (odd offset on purpose: no alignment assumed)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long total = 0;
    int ok = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int q = threadIdx.x; q < 32 * 33; q += blockDim.x) SL[q] = img[q];
        __syncthreads();
        if (wave == 0) {
            const long long t0 = __builtin_amdgcn_s_memtime();
            ok = which ? solve_wave_32_scalar_forwarding(0, 32 * 33 + 7, n, lane) : solve_wave_32(0, 32 * 33 + 7, n, lane);
            total += (long long)__builtin_amdgcn_s_memtime() - t0;
        }
        __syncthreads();
    }
    if (wave == 0) {
        if (lane < n) xo[lane] = colbuf[128 + lane];
        if (lane == 0) {
            cyc[0] = total / reps;
            okv[0] = ok;
        }
    }
}
int main() {
    const int n = 30, NR = 32, P = 33;
    std::vector<double> img(NR * P, 0.0);
    srand(5);
    std::vector<double> M(n * n);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < NR - 1; ++i)
        for (int k = 0; k <= i; ++k) {
            double v = 0;
            if (i < n) {
                for (int q = 0; q < n; ++q) v += M[i * n + q] * M[k * n + q];
                if (i == k) v += 0.5;
            } else {
                v = i == k ? 1.0 : 0.0;
            }
            img[i * P + k] = v;
        }
    for (int k = 0; k < n; ++k) img[(NR - 1) * P + k] = rand() / (double)RAND_MAX - 0.5;
    double *d_img, *d_x;
    long long* d_c;
    int* d_ok;
    (void)hipMalloc(&d_img, img.size() * 8);
    (void)hipMalloc(&d_x, 64 * 8);
    (void)hipMalloc(&d_c, 8);
    (void)hipMalloc(&d_ok, 4);
    (void)hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice);
    double x[2][32];
    for (int which = 0; which < 2; ++which) {
        long long c = 0;
        int ok = 0;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 64 * 1024, 0, d_img, d_x, d_c, d_ok, n, which, 200);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&ok, d_ok, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(x[which], d_x, n * 8, hipMemcpyDeviceToHost);
        printf("solve %s: %lld cycles per solve, ok %d, x[0] %.17g x[29] %.17g\n", which ? "scalar-forwarding variant" : "product (ba_solve.h)", c, ok, x[which][0], x[which][29]);
    }
    printf("results %s\n", memcmp(x[0], x[1], n * 8) == 0 ? "bit-identical" : "DIFFER");
    return 0;
}
