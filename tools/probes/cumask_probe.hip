// Dev probe: where do the workgroups of a kernel launched on a CU-masked stream run?  Prints, per mask pattern, the
// histogram of XCC ids (s_getreg HW_REG_XCC_ID) over 64 workgroups.
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where(int* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // keep the workgroup alive for a while so that all of them are resident together
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (int)(xcc & 15);
        out[2 * blockIdx.x + 1] = (int)hwid;
    }
}
int main() {
    int* d;
    hipMalloc(&d, 2 * 64 * sizeof(int));
    for (int pattern = 0; pattern < 4; ++pattern) {
        for (int x = 0; x < 8; x += 3) {
            std::vector<uint32_t> mask(8, 0);  // 256 bits
            for (int i = 0; i < 256; ++i) {
                bool on = pattern == 0 ? (i % 8 == x) : pattern == 1 ? (i / 32 == x) : pattern == 2 ? ((i / 4) % 8 == x) : ((i / 2) % 8 == x);
                if (on) mask[i / 32] |= 1u << (i % 32);
            }
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
            if (e != hipSuccess) {
                printf("pattern %d x %d: hipExtStreamCreateWithCUMask failed: %s\n", pattern, x, hipGetErrorString(e));
                continue;
            }
            hipLaunchKernelGGL(where, dim3(32), dim3(512), 150 * 1024 > 65536 ? 0 : 0, s, d);
            hipStreamSynchronize(s);
            int h[128];
            hipMemcpy(h, d, sizeof(int) * 64, hipMemcpyDeviceToHost);
            int hist[16] = {0};
            for (int b = 0; b < 32; ++b) hist[h[2 * b]]++;
            printf("pattern %d (%s) x=%d: xcc histogram", pattern,
                   pattern == 0 ? "bit%8==x" : pattern == 1 ? "bit/32==x" : pattern == 2 ? "(bit/4)%8==x" : "(bit/2)%8==x", x);
            for (int k = 0; k < 8; ++k) printf(" %d", hist[k]);
            printf("\n");
            hipStreamDestroy(s);
        }
    }
    return 0;
}
