// tools/probes/chain_probe.hip -- what does one step of a Schur chain (v_mfma_f64_16x16x4_f64, operands from LDS, the
// accumulator carried from step to step) cost, alone and with other chain waves in the workgroup?  Prints cycles per step
// for every set of active waves in `masks`.  hipcc --offload-arch=gfx950 -O3 chain_probe.hip -o chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int LDU>
__device__ __forceinline__ v4d chain(const double* U, int ti, int tj, int m0, int m1, int lane) {
    v4d acc = {0, 0, 0, 0};
    const int k = lane >> 4, i = lane & 15;
    constexpr int st = 4 * LDU;
    const double* pa = U + (size_t)(4 * m0 + k) * LDU + (16 * ti + i);
    const double* pb = U + (size_t)(4 * m0 + k) * LDU + (16 * tj + i);
    int m = m0;
#define LOAD4(A, Bv) A##0 = pa[0], Bv##0 = pb[0], A##1 = pa[st], Bv##1 = pb[st], A##2 = pa[2 * st], Bv##2 = pb[2 * st], A##3 = pa[3 * st], Bv##3 = pb[3 * st]; pa += 4 * st; pb += 4 * st
#define MFMA4(A, Bv, C, D, more)                                            \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##0, Bv##0, acc, 0, 0, 0);  \
    __builtin_amdgcn_sched_barrier(0);                                      \
    if (more) { LOAD4(C, D); }                                              \
    __builtin_amdgcn_sched_barrier(0);                                      \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##1, Bv##1, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##2, Bv##2, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A##3, Bv##3, acc, 0, 0, 0)
    if (m + 4 <= m1) {
        double a0, a1, a2, a3, b0, b1, b2, b3, c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        LOAD4(a, b);
        m += 4;
        for (;;) {
            const bool more1 = m + 4 <= m1;
            MFMA4(a, b, c, d, more1);
            if (!more1) break;
            m += 4;
            const bool more2 = m + 4 <= m1;
            MFMA4(c, d, a, b, more2);
            if (!more2) break;
            m += 4;
        }
    }
    return acc;
}
// variant: accumulator chain only (operands fixed in registers): the matrix unit by itself
__device__ __forceinline__ v4d chain_regs(int steps, double a, double b) {
    v4d acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    return acc;
}
__global__ __launch_bounds__(512) void k(unsigned mask, int steps, int mode, long long* out, double* sink) {
    extern __shared__ double U[];
    for (int i = threadIdx.x; i < 4 * steps * 33 + 64; i += 512) U[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    v4d acc = {0, 0, 0, 0};
    if ((mask >> wave) & 1) {
        t0 = __builtin_amdgcn_s_memtime();
        acc = mode == 0 ? chain<33>(U, 0, wave & 1, 0, steps, lane) : chain_regs(steps, U[lane], U[lane + 1]);
        asm volatile("s_nop 0" ::"v"(acc[0]));
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    long long* out;
    double* sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4096);
    const int steps = 64;
    const unsigned masks[] = {0x01, 0x11, 0x03, 0x05, 0x0f, 0x3f, 0xff};
    for (int mode = 0; mode < 2; ++mode)
        for (unsigned m : masks) {
            long long h[8];
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k, dim3(1), dim3(512), (4 * steps * 33 + 64) * 8, 0, m, steps, mode, out, sink);
                hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            }
            printf("%s waves %02x:", mode == 0 ? "LDS operands " : "register ops ", m);
            for (int w = 0; w < 8; ++w)
                if ((m >> w) & 1) printf(" w%d %.0f", w, (double)h[w] / steps);
            printf("  cycles/step\n");
        }
    return 0;
}
