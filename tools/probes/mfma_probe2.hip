// tools/probes/mfma_probe2.hip -- dev probe: cycles of a dependent v_mfma_f64_16x16x4_f64 chain run by ONE wave of a
// 512-thread workgroup while the other waves (a) wait at the barrier, (b) spin on LDS, (c) run the same chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int mode, int lds_bytes) {
    extern __shared__ double dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v4d acc = {0, 0, 0, 0};
    double a = 1.0 + lane * 1e-3;
    long long t0 = 0, t1 = 0;
    dyn[threadIdx.x] = a;
    __syncthreads();
    if (wave == 0 || mode == 2) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
    } else if (mode == 1) {
        double s = 0;
        for (int i = 0; i < 3000; ++i) s += dyn[(threadIdx.x + i) & 511];
        acc[1] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1];
}
int main() {
    long long* d; double* s; long long h;
    hipMalloc(&d, 64); hipMalloc(&s, 4096);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int lds : {8192, 140 * 1024})
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), lds, 0, d, s, mode, lds);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("lds %d KB, other waves %s: %.1f cycles per dependent MFMA\n", lds / 1024, mode == 0 ? "at the barrier" : mode == 1 ? "reading LDS" : "running the same chain", h / 256.0);
        }
    // many blocks (one per CU)
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 140 * 1024, 0, d, s, 0, 0);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("256 blocks, other waves at the barrier: %.1f\n", h / 256.0);
    return 0;
}
