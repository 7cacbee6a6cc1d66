// Whole-chip FP64 rate of the MI355X, measured: the figure bench.py's roofline block divides by (FP64_PEAK_TFLOPS = 78.6:
// 256 CU x 4 SIMD x 16 lanes-per-clock x 2 flop x 2.4 GHz for the vector ALU; one v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles
// per SIMD for the matrix cores -- the same number).  Neither is in the local micro-architecture guide, hence this probe.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_peak_probe fp64_peak_probe.hip && ./fp64_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kIters = 4096;
__global__ __launch_bounds__(256) void k_fma(double* out, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-6;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], m, c);  // 16 independent chains per lane
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma(double* out, double seed) {
    v4d acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = {seed, seed, seed, seed};
    const double a = 1.0000001 + threadIdx.x * 1e-9, b = 1e-3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);  // 4 independent chains per wave
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU
    double* d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (kind == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, 1.0);
            else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, 1.0);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = kind == 0 ? (double)blocks * 256 * kIters * 16 * 2 : (double)blocks * 4 * kIters * 4 * 2048.0;
        printf("%s: %.3f ms, %.1f TFLOP/s (best of 4 after a warm-up; %d workgroups x 256 threads)\n",
               kind == 0 ? "v_fma_f64 (vector ALU, 16 independent chains per lane)" : "v_mfma_f64_16x16x4_f64 (matrix cores, 4 independent chains per wave)",
               best, flop / (best * 1e-3) / 1e12, blocks);
    }
    return 0;
}
