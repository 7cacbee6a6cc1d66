// Dev probe: issue cost of f64 vector operations for ONE wave on a SIMD (cycles per instruction), independent vs
// dependent.   hipcc --offload-arch=gfx950 -O2 -o dp_rate_probe dp_rate_probe.hip && ./dp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, long long* cyc, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const double m = 1.0000001, c = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) {  // 8 independent chains
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double d = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 8000; ++i) d = __builtin_fma(d, m, c);  // one dependent chain
    long long t2 = __builtin_amdgcn_s_memtime();
    float f = (float)d;
    for (int i = 0; i < 8000; ++i) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);  // dependent f32 chain
    long long t3 = __builtin_amdgcn_s_memtime();
    double r = d + f;
    for (int i = 0; i < 2000; ++i) r = __builtin_amdgcn_rcp(r) + 1.5;  // dependent rcp + add
    long long t4 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3;
    }
    out[threadIdx.x] = r;
}
int main() {
    double* d; long long* c;
    hipMalloc(&d, 64 * 8); hipMalloc(&c, 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, 1.0);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on gfx9: report ticks and the ratio between the loops
    printf("ticks: 8000 independent fma_f64 %lld | 8000 dependent fma_f64 %lld | 8000 dependent fma_f32 %lld | 2000 x (rcp_f64 + add_f64) %lld\n", h[0], h[1], h[2], h[3]);
    return 0;
}
