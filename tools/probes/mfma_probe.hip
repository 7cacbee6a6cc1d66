// tools/probes/mfma_probe.hip -- dev probe (run on the GPU box): arithmetic semantics and timing of the f64 building
// blocks the BA kernel relies on for a bit-reproducible result:
//   * v_mfma_f64_16x16x4_f64: is D = C + sum_k A_ik B_kj a chain of IEEE FMAs in k order?
//   * f64 division / sqrt / rcp: correctly rounded like the host's?
//   * cycles of a dependent MFMA chain, of LDS broadcast round trips, of v_readlane broadcasts.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double* A, const double* B, const double* C, double* D, int reps) {
    // A: 16x4 (row i, k), B: 4x16 (k, col j), C/D: 16x16.  lane l: a = A[l%16][l/16], b = B[l/16][l%16]
    const int l = threadIdx.x;
    v4d acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    for (int it = 0; it < reps; ++it) {
        const double a = A[it * 64 + (l & 15) * 4 + (l >> 4)], b = B[it * 64 + (l >> 4) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_ops(const double* x, const double* y, double* q, double* s, double* r, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        q[i] = x[i] / y[i];
        s[i] = sqrt(fabs(x[i]));
        r[i] = 1.0 / y[i];
    }
}
__global__ void k_time(long long* out, double* sink) {
    const int l = threadIdx.x;
    v4d acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 128; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc2, 0, 0, 0);
    }
    long long t2 = __builtin_amdgcn_s_memtime();
    // dependent f64 fma chain
    double f = a;
    for (int i = 0; i < 256; ++i) f = __builtin_fma(f, b, a);
    long long t3 = __builtin_amdgcn_s_memtime();
    // dependent division chain
    double d = a;
    for (int i = 0; i < 64; ++i) d = 1.0 / (d + b);
    long long t4 = __builtin_amdgcn_s_memtime();
    // LDS broadcast round trip: write own value, wait, read 8 x b128 broadcast, dependent
    __shared__ double buf[64];
    double g = a;
    for (int i = 0; i < 64; ++i) {
        buf[l] = g;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) sacc += buf[(i + k) & 63];
        g = sacc * 0.01 + a;
        __builtin_amdgcn_wave_barrier();
    }
    long long t5 = __builtin_amdgcn_s_memtime();
    // readlane broadcast + fma, 16 per step, dependent
    double h = a;
    for (int i = 0; i < 64; ++i) {
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int lo = __builtin_amdgcn_readlane(__double2loint(h), (k * 3) & 63), hi = __builtin_amdgcn_readlane(__double2hiint(h), (k * 3) & 63);
            sacc = __builtin_fma(__hiloint2double(hi, lo), b, sacc);
        }
        h = sacc * 0.01 + a;
    }
    long long t6 = __builtin_amdgcn_s_memtime();
    // s_barrier cost with 8 waves
    for (int i = 0; i < 64; ++i) __syncthreads();
    long long t7 = __builtin_amdgcn_s_memtime();
    if (l == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5; out[6] = t7 - t6;
    }
    sink[threadIdx.x] = acc[0] + acc2[1] + f + d + g + h;
}
static double model_chain(const double* A, const double* B, double c, int i, int j, int reps, int mode) {
    double acc = c;
    for (int it = 0; it < reps; ++it) {
        const double* a = A + it * 64; const double* b = B + it * 64;
        if (mode == 0) for (int k = 0; k < 4; ++k) acc = std::fma(a[i * 4 + k], b[k * 16 + j], acc);
        else if (mode == 1) for (int k = 3; k >= 0; --k) acc = std::fma(a[i * 4 + k], b[k * 16 + j], acc);
        else if (mode == 2) { double s = 0; for (int k = 0; k < 4; ++k) s = std::fma(a[i * 4 + k], b[k * 16 + j], s); acc += s; }
        else { for (int k = 0; k < 4; ++k) acc = acc + a[i * 4 + k] * b[k * 16 + j]; }
    }
    return acc;
}
int main() {
    srand(5);
    const int reps = 7;
    std::vector<double> A(64 * reps), B(64 * reps), C(256), D(256);
    auto rnd = [] { return ((double)rand() / RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 9 - 4); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    for (auto& v : C) v = rnd();
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    for (int rp : {1, reps}) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, rp);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int mode = 0; mode < 4; ++mode) {
            int bad = 0; double maxd = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double m = model_chain(A.data(), B.data(), C[i * 16 + j], i, j, rp, mode);
                if (std::memcmp(&m, &D[i * 16 + j], 8)) ++bad;
                maxd = std::fmax(maxd, std::fabs(m - D[i * 16 + j]));
            }
            printf("mfma reps %d model %d (0 fma k up, 1 fma k down, 2 dot then add, 3 unfused): mismatches %d / 256, max diff %g\n", rp, mode, bad, maxd);
        }
    }
    const int n = 1 << 20;
    std::vector<double> x(n), y(n), q(n), s(n), r(n);
    for (int i = 0; i < n; ++i) { x[i] = rnd() * std::ldexp(1.0, rand() % 60 - 30); y[i] = rnd() * std::ldexp(1.0, rand() % 60 - 30); if (y[i] == 0) y[i] = 1; }
    double *dx, *dy, *dq, *ds, *dr;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dr, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ops, dim3(n / 256), dim3(256), 0, 0, dx, dy, dq, ds, dr, n);
    hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    int bq = 0, bs = 0, br = 0;
    for (int i = 0; i < n; ++i) {
        double hq = x[i] / y[i], hs = std::sqrt(std::fabs(x[i])), hr = 1.0 / y[i];
        bq += std::memcmp(&hq, &q[i], 8) != 0; bs += std::memcmp(&hs, &s[i], 8) != 0; br += std::memcmp(&hr, &r[i], 8) != 0;
    }
    printf("f64 div mismatches %d, sqrt %d, 1/y %d of %d\n", bq, bs, br, n);
    long long* dt; double* sink; long long ht[8];
    hipMalloc(&dt, 64); hipMalloc(&sink, 512 * 8);
    for (int th : {64, 512}) {
        hipLaunchKernelGGL(k_time, dim3(1), dim3(th), 0, 0, dt, sink);
        hipMemcpy(ht, dt, 56, hipMemcpyDeviceToHost);
        printf("threads %d: dependent mfma %.1f cyc each; 2 interleaved chains %.1f per mfma; dep f64 fma %.1f; dep 1/(x+b) %.1f; LDS bcast step (1 write + 16 reads) %.1f; readlane x16 step %.1f; syncthreads %.1f\n",
               th, ht[0] / 256.0, ht[1] / 256.0, ht[2] / 256.0, ht[3] / 64.0, ht[4] / 64.0, ht[5] / 64.0, ht[6] / 64.0);
    }
    return 0;
}
