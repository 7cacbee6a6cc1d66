// Dev probe: cycles of the workgroup-wide block LDL^T of the 64-row class (csrc/ba_solve.h solve_block<64>) on its own, and the
// cycles wave 0 spends per stage (strip update, the four pivots, panel write, barrier wait).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../monocular-visual-odometry_amd/csrc -o block_solve_probe block_solve_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "ba_solve.h"

constexpr int NR = 64, P = 65, SLD = NR * P, PAN = 16 * NR + 16;

template <bool STAMPS>
__global__ __launch_bounds__(512) void probe(const double* img, double* xo, long long* cyc, int* okv, int n, int reps) {
    double* SL = ba_dyn_lds;
    long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    long long total = 0;
    int ok = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int q = threadIdx.x; q < SLD; q += blockDim.x) SL[q] = img[q];
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        ok = solve_block<NR, STAMPS>(0, SLD, SLD + PAN, n, threadIdx.x, st);
        total += (long long)__builtin_amdgcn_s_memtime() - t0;
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        if (threadIdx.x < n) xo[threadIdx.x] = ba_dyn_lds[SLD + PAN + threadIdx.x];
        if (threadIdx.x == 0) {
            cyc[0] = total / reps;
            for (int k = 0; k < 4; ++k) cyc[1 + k] = st[k] / reps;
            okv[0] = ok;
        }
    }
    if ((threadIdx.x & 63) == 0 && threadIdx.x > 0)
        for (int k = 4; k < 7; ++k) cyc[8 + 3 * (threadIdx.x >> 6) + k - 4] = st[k] / reps;
}
int main() {
    const int n = 60;
    std::vector<double> img(SLD, 0.0);
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
    (void)hipMalloc(&d_c, 40 * 8);
    (void)hipMalloc(&d_ok, 4);
    (void)hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice);
    double x[2][64];
    for (int which = 0; which < 2; ++which) {
        long long c[40] = {0};
        int ok = 0;
        const size_t lds = (SLD + PAN + 64) * 8;
        for (int rep = 0; rep < 2; ++rep) {
            if (which) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(512), lds, 0, d_img, d_x, d_c, d_ok, n, 200);
            else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(512), lds, 0, d_img, d_x, d_c, d_ok, n, 200);
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(c, d_c, 40 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&ok, d_ok, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(x[which], d_x, n * 8, hipMemcpyDeviceToHost);
        printf("solve_block<64> %s: %lld cycles per solve, ok %d, x[0] %.17g x[59] %.17g\n", which ? "with stamps" : "product", c[0], ok, x[which][0], x[which][59]);
        if (which) printf("   wave 0 per solve: strip %lld  pivots %lld  panel write %lld  barrier %lld\n", c[1], c[2], c[3], c[4]);
        if (which) for (int w = 1; w < 8; ++w) printf("   wave %d per solve: fetch %lld  mfma + write %lld  barrier %lld\n", w, c[8 + 3 * w], c[9 + 3 * w], c[10 + 3 * w]);
    }
    printf("results %s\n", memcmp(x[0], x[1], n * 8) == 0 ? "bit-identical" : "DIFFER");
    return 0;
}
