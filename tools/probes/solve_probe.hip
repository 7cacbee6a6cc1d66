// Dev probe: cycles per reduced solve (31-row embedded LDL^T on one wave) of the round-3 form and of the round-4 form
// (csrc/ba_solve.h), and their results compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../monocular-visual-odometry_amd/csrc -o solve_probe solve_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "ba_solve.h"

// The 32-row flavour uses both halves of the wave: lane = (row i = lane & 31, half h = lane >> 5) keeps the columns
// k = 2 m + h of its row (16 registers instead of 32), so that the prefetched column of the next step fits the register
// file next to the one in use, and a step costs half the fused multiply-adds per lane.  Same arithmetic per entry.
__device__ __forceinline__ int solve_wave_32_r03(int sl_off, int cb_off, int n, int lane) {
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
            ok = which ? solve_wave_32(0, 32 * 33 + 7, n, lane) : solve_wave_32_r03(0, 32 * 33 + 7, n, lane);
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
        printf("solve %s: %lld cycles per solve, ok %d, x[0] %.17g x[29] %.17g\n", which ? "round 4" : "round 3", c, ok, x[which][0], x[which][29]);
    }
    printf("results %s\n", memcmp(x[0], x[1], n * 8) == 0 ? "bit-identical" : "DIFFER");
    return 0;
}
