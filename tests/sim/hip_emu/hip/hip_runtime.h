// tests/sim/hip_emu/hip/hip_runtime.h -- TEST AID, not a backend: a stand-in for <hip/hip_runtime.h> that lets the
// SOURCE of a HIP kernel (and of its host side) be compiled for x86 and executed on the CPU, thread for thread:
//   * every workgroup of a launch runs on its own OS thread; its GPU threads are fibers (own stack each) that the
//     workgroup thread schedules cooperatively: a fiber runs until it reaches __syncthreads(), a wave operation
//     (__shfl_xor, v_readlane, MFMA ...) or a polling pause (s_sleep), where it blocks until its workgroup / wave /
//     nobody arrives;
//   * wave operations are rendez-vous points of the 64 fibers of a wave: lanes deposit their operands, the last
//     arrival releases the wave, every lane picks up what the instruction would have delivered to it;
//   * v_mfma_f64_16x16x4_f64 is restated as what the hardware was probed to compute (tools/probes/mfma_probe.hip): per
//     output element a chain of IEEE fused multiply-adds over the four k slots in order;
//   * cross-workgroup traffic goes through ordinary memory with relaxed atomics; the workgroups really run
//     concurrently (OS threads), so hand-off protocols are exercised, bounded spins can time out.
// The fiber order inside a workgroup can be reversed or shuffled (EMU_ORDER=reverse|shuffle): a missing barrier shows
// as a result that depends on the order.  Nothing here is ever linked into libmvo_hip.so: the product has no CPU path
// (tests/test_abi.py checks that); the CPU suite uses this to check kernel logic bit for bit against the oracle where
// no GPU exists.
#ifndef MVO_HIP_EMU_RUNTIME_H
#define MVO_HIP_EMU_RUNTIME_H
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <utility>

#define MVO_KERNEL_SIM 1

// ---------------------------------------------------------------------------------------------- language keywords
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local  // one OS thread per workgroup: thread-local = workgroup-local
#define __constant__                    // (plain globals; hipMemcpyToSymbol below)
#define HIP_SYMBOL(x) x

struct emu_uint3 {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

// ---------------------------------------------------------------------------------------------- scheduler entry points
void emu_syncthreads();
void emu_wave_sync();   // rendez-vous of the (live) lanes of the calling fiber's wave
void emu_yield();       // polling pause: lets the other fibers of the workgroup (and the other workgroups) run
long long emu_clock();
// per-wave exchange area (valid between two emu_wave_sync of one operation)
struct EmuWaveBuf {
    unsigned long long u64[64];
    double d[64][6];
};
EmuWaveBuf& emu_wave_buf();
inline int emu_lane() { return (int)(threadIdx.x & 63u); }

#define __syncthreads() emu_syncthreads()

// ---------------------------------------------------------------------------------------------- wave operations
inline unsigned long long emu_exchange_u64(unsigned long long v, int src_lane) {
    EmuWaveBuf& b = emu_wave_buf();
    b.u64[emu_lane()] = v;
    emu_wave_sync();
    const unsigned long long r = b.u64[src_lane & 63];
    emu_wave_sync();
    return r;
}
inline double __shfl_xor(double v, int mask) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    u = emu_exchange_u64(u, emu_lane() ^ mask);
    memcpy(&v, &u, 8);
    return v;
}
inline int __shfl_xor(int v, int mask) { return (int)emu_exchange_u64((unsigned)v, emu_lane() ^ mask); }
inline int __shfl(int v, int src) { return (int)emu_exchange_u64((unsigned)v, src); }
inline double __shfl(double v, int src) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    u = emu_exchange_u64(u, src);
    memcpy(&v, &u, 8);
    return v;
}
inline unsigned long long __ballot(int pred) {
    EmuWaveBuf& b = emu_wave_buf();
    b.u64[emu_lane()] = pred ? 1ull : 0ull;
    emu_wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (b.u64[i] & 1ull) << i;
    emu_wave_sync();
    return m;
}
inline unsigned __shfl_xor(unsigned v, int mask) { return (unsigned)emu_exchange_u64(v, emu_lane() ^ mask); }
inline unsigned long long __shfl_xor(unsigned long long v, int mask) { return emu_exchange_u64(v, emu_lane() ^ mask); }
inline long long __shfl_xor(long long v, int mask) { return (long long)emu_exchange_u64((unsigned long long)v, emu_lane() ^ mask); }
inline float __shfl_xor(float v, int mask) {
    unsigned u;
    memcpy(&u, &v, 4);
    u = (unsigned)emu_exchange_u64(u, emu_lane() ^ mask);
    memcpy(&v, &u, 4);
    return v;
}
inline unsigned __shfl(unsigned v, int src) { return (unsigned)emu_exchange_u64(v, src); }
inline float __shfl(float v, int src) {
    unsigned u;
    memcpy(&u, &v, 4);
    u = (unsigned)emu_exchange_u64(u, src);
    memcpy(&v, &u, 4);
    return v;
}
// lane l receives the value of lane l - delta; the lanes below delta keep their own
inline int __shfl_up(int v, unsigned delta) {
    const int l = emu_lane();
    return (int)emu_exchange_u64((unsigned)v, l >= (int)delta ? l - (int)delta : l);
}
inline unsigned __shfl_up(unsigned v, unsigned delta) { return (unsigned)__shfl_up((int)v, delta); }
#define __builtin_amdgcn_readlane(v, src) ((int)emu_exchange_u64((unsigned)(v), (src)))
#define __builtin_amdgcn_readfirstlane(v) ((int)emu_exchange_u64((unsigned)(v), 0))
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu_yield()
#define __builtin_amdgcn_s_memtime() ((unsigned long long)emu_clock())
#define __builtin_amdgcn_s_memrealtime() ((unsigned long long)emu_clock() / 10ull)  // 100 MHz
// v_rcp_f64 is an approximation that the callers refine by Newton steps to the correctly rounded quotient; starting the
// refinement from the correctly rounded value ends on the same bits
#define __builtin_amdgcn_rcp(d) (1.0 / (d))

typedef double emu_v4d __attribute__((ext_vector_type(4)));
// D = A (16 x 4) * B (4 x 16) + C: lane (k = lane >> 4, i = lane & 15) supplies A[i][k] and B[k][i]; register r of lane
// (q = lane >> 4, j = lane & 15) holds D[4 r + q][j].  Per element: fused multiply-adds over k = 0..3 in order.
inline emu_v4d emu_mfma_f64_16x16x4(double a, double b, emu_v4d c) {
    EmuWaveBuf& w = emu_wave_buf();
    const int lane = emu_lane();
    w.d[lane][0] = a;
    w.d[lane][1] = b;
    emu_wave_sync();
    const int q = lane >> 4, j = lane & 15;
    emu_v4d r;
    for (int reg = 0; reg < 4; ++reg) {
        const int i = 4 * reg + q;
        double acc = c[reg];
        for (int k = 0; k < 4; ++k) acc = __builtin_fma(w.d[16 * k + i][0], w.d[16 * k + j][1], acc);
        r[reg] = acc;
    }
    emu_wave_sync();
    return r;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) emu_mfma_f64_16x16x4((a), (b), (c))
// v_mfma_i32_16x16x64_i8: D = A (16 x 64) * B (64 x 16) + C in exact integers.  Lane (i = lane & 15, kb = lane >> 4) supplies the 16
// bytes A[i][16 kb .. 16 kb + 15] and, as the B operand, B[16 kb .. 16 kb + 15][i]; register r of lane (q = lane >> 4,
// j = lane & 15) holds D[4 q + r][j] (the layout csrc/match_kernels.hip relies on, validated on the MI355X by its tests).
typedef int emu_v4i __attribute__((ext_vector_type(4)));
struct EmuI8Buf {
    signed char a[64][16], b[64][16];
};
EmuI8Buf& emu_wave_i8();
inline emu_v4i emu_mfma_i32_16x16x64_i8(emu_v4i a, emu_v4i b, emu_v4i c) {
    EmuI8Buf& w = emu_wave_i8();
    const int lane = emu_lane();
    memcpy(w.a[lane], &a, 16);
    memcpy(w.b[lane], &b, 16);
    emu_wave_sync();
    const int q = lane >> 4, j = lane & 15;
    emu_v4i r;
    for (int reg = 0; reg < 4; ++reg) {
        const int i = 4 * q + reg;
        int acc = c[reg];
        for (int kb = 0; kb < 4; ++kb)
            for (int k = 0; k < 16; ++k) acc += (int)w.a[16 * kb + i][k] * (int)w.b[16 * kb + j][k];
        r[reg] = acc;
    }
    emu_wave_sync();
    return r;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, x, y, z) emu_mfma_i32_16x16x64_i8((a), (b), (c))

// ---------------------------------------------------------------------------------------------- scalar helpers
inline int __double2loint(double v) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    return (int)(unsigned)(u & 0xffffffffull);
}
inline int __double2hiint(double v) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    return (int)(unsigned)(u >> 32);
}
inline double __hiloint2double(int hi, int lo) {
    const unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    double v;
    memcpy(&v, &u, 8);
    return v;
}
inline long long __double_as_longlong(double v) {
    long long u;
    memcpy(&u, &v, 8);
    return u;
}
inline double __longlong_as_double(long long u) {
    double v;
    memcpy(&v, &u, 8);
    return v;
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
// (this build is compiled with -ffp-contract=off: the plain operators round once, like the _rn intrinsics)
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline int __float2int_rn(float v) { return (int)lrintf(v); }  // round to nearest even (the default rounding mode)
inline int __float2int_rd(float v) { return (int)floorf(v); }
inline unsigned emu_sad_u8(unsigned a, unsigned b, unsigned c) {
    for (int k = 0; k < 4; ++k) {
        const int x = (int)((a >> (8 * k)) & 255u), y = (int)((b >> (8 * k)) & 255u);
        c += (unsigned)(x > y ? x - y : y - x);
    }
    return c;
}
#define __builtin_amdgcn_sad_u8(a, b, c) emu_sad_u8((a), (b), (c))
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
// vector types of the HIP headers that the kernels use
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
inline int2 make_int2(int x, int y) { return {x, y}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline float2 make_float2(float x, float y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return {x, y}; }
using std::max;
using std::min;
inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
inline size_t min(size_t a, int b) { return a < (size_t)b ? a : (size_t)b; }
inline size_t max(size_t a, int b) { return a > (size_t)b ? a : (size_t)b; }

// ---------------------------------------------------------------------------------------------- atomics
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), __ATOMIC_RELAXED)
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---------------------------------------------------------------------------------------------- host runtime subset
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorNotReady 600
#define hipErrorUnknown 999
struct EmuStream;
struct EmuEvent;
typedef EmuStream* hipStream_t;
typedef EmuEvent* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToHost = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

hipError_t hipSetDevice(int);
hipError_t hipGetDeviceCount(int*);
inline hipError_t hipGetDevice(int* d) {
    *d = 0;
    return 0;
}
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int device);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio);
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
    *least = 0;
    *greatest = -1;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum { hipDeviceScheduleAuto = 0, hipDeviceScheduleSpin = 1, hipDeviceScheduleYield = 2, hipDeviceScheduleBlockingSync = 4 };
inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
template <class T>
inline hipError_t hipMemcpyToSymbol(T& symbol, const void* src, size_t bytes, size_t offset = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
    memcpy(reinterpret_cast<char*>(&symbol) + offset, src, bytes);
    return hipSuccess;
}

// A launch: `body` is executed by every GPU thread of the grid; asynchronous on `stream` like the real thing.
void emu_launch(std::function<void()> body, dim3 grid, dim3 block, size_t dyn_lds_bytes, hipStream_t stream);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu_launch([=]() { (kernel)(__VA_ARGS__); }, dim3(grid), dim3(block), (lds), (stream))
// dynamic LDS segment of the calling workgroup
void* emu_dyn_lds();

#endif
