// Dev probe: cost of an LDS hand-off inside ONE wave (write a value, read the neighbour's), of v_readlane and of
// v_permlane32_swap, in shader cycles per dependent round.   hipcc --offload-arch=gfx950 -O2 -o p lds_trip_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, long long* cyc) {
    __shared__ double buf[2][64];
    const int lane = threadIdx.x;
    double v = lane + 1.5;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) {  // write own, read the neighbour's: the next write depends on the read
        buf[i & 1][lane] = v;
        v = buf[i & 1][(lane + 1) & 63] * 1.0000001;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) {  // the same dependency through v_readlane (two 32-bit halves)
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_readlane(lo, (i + 1) & 63);
        hi = __builtin_amdgcn_readlane(hi, (i + 1) & 63);
        v = __hiloint2double(hi, lo) * 1.0000001 + lane;
    }
    long long t2 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) {  // write, then FIVE reads the next round depends on (the solver's pattern)
        buf[i & 1][lane] = v;
        const double* b = buf[i & 1];
        v = (b[(lane + 1) & 63] + b[(lane + 2) & 63] + b[(lane + 3) & 63] + b[(lane + 4) & 63] + b[(lane + 5) & 63]) * 0.2;
    }
    long long t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2;
    }
    out[lane] = v;
}
int main() {
    double* d; long long* c;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 3 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
    (void)hipDeviceSynchronize();
    long long h[3];
    (void)hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
    printf("cycles per dependent round: LDS write+read %.1f | readlane pair + fma %.1f | LDS write + 5 reads + adds %.1f\n", h[0] / 1000.0, h[1] / 1000.0, h[2] / 1000.0);
    return 0;
}
