// csrc/match_kernels.hip -- all-pairs descriptor matching on gfx950.
//   k_knn2      exact 2-NN in 256-bit Hamming space: replaces cv::BFMatcher("BruteForce-Hamming")::knnMatch(k=2)
//               (reference src/geometry/feature_match.cpp:141,203-208) and, as the exact 1-NN, the
//               cv::FlannBasedMatcher(LshIndexParams(5,10,2))::match call at feature_match.cpp:140,162.
//   k_radius_l1 geometry::matchByRadiusAndBruteForce (feature_match.cpp:86-124).
// Work decomposition (wave64): one LANE per query, the query's 256 bits live in 4 x u64 VGPRs; the train
// descriptor of the current step is wave-uniform, so it is fetched with scalar loads (s_load_dwordx8) and
// broadcast for free; distance = 8 x (v_xor + v_bcnt_u32).  k_knn2: one wave per (64 queries, train slice) pair,
// 32 slices -> 1024 waves for 2000 x 2000 in workgroups of four, each scanning its slice in index order; LDS fold of the
// four, then the last workgroup to arrive for a query group folds the 8 partial (best, second) pairs, keeping the
// lexicographically smallest
// (distance, index) pairs, which reproduces cv::batchDistance's tie rule exactly (equal distances keep the lower
// train index).  k_radius_l1
// keeps the single-kernel form (16 waves x 64 queries, LDS merge).  Everything is integer: results are bit-exact.
// The whole working set (<= 2 x 128 KB) is L2-resident: the bound is VALU integer throughput, not HBM.
#include "mvo_internal.h"

#include <climits>

typedef unsigned long long u64;

#define MK_WAVES 16

struct Top2 {
    int d0, i0, d1, i1;
};

__device__ __forceinline__ void top2_insert(Top2& t, int d, int j) {
    // strict '<' on both levels: an equal distance never displaces an earlier (lower-index) entry
    bool lt0 = d < t.d0, lt1 = d < t.d1;
    int nd1 = lt0 ? t.d0 : (lt1 ? d : t.d1);
    int ni1 = lt0 ? t.i0 : (lt1 ? j : t.i1);
    t.d0 = lt0 ? d : t.d0;
    t.i0 = lt0 ? j : t.i0;
    t.d1 = nd1;
    t.i1 = ni1;
}

// grid = (query groups of 64) x MK_GROUPS; one workgroup = 4 waves (one per SIMD of its CU) = 4 of the MK_SLICES train
// slices -> 1024 waves for a 2000 x 2000 call: every SIMD of the chip gets one, which is what the pair loop wants (it is
// VALU-throughput bound: 16 waves x 64 queries per workgroup put 8x the work on an eighth of the SIMDs and took 44 us).
// A wave first parks its slice in registers -- lane j holds train descriptor j of the current 64-train chunk (one
// coalesced 2 KB read) -- and then broadcasts descriptor j to all lanes with v_readlane: no memory access and no LDS in
// the pair loop.  The four partial (best, second) pairs of a query meet in LDS; the workgroup's partial goes out
// write-through at agent scope and the workgroup that arrives LAST for its query group (arrival counter per group)
// folds the MK_GROUPS partials -- one batch of loads -- and delivers the result: no merge launch.  (With one wave per
// workgroup the last wave had 32 partials to fetch, four dependent batches: 2/3 of the call's 20 us.)  Every fold keeps
// the two lexicographically smallest (distance, train index) pairs, which is exactly the strict-'<' in-order scan
// (indices are unique), so the order in which the slices finished does not matter.
#define MK_SLICES 32
#define MK_GROUPS (MK_SLICES / 4)
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ bool pair_less(int d, int i, int e, int j) {  // (d,i) < (e,j); index -1 = empty = +inf
    return j < 0 ? i >= 0 : (i >= 0 && (d < e || (d == e && i < j)));
}
// merge two sorted pairs (p.x,p.y)<=(p.z,p.w) and (r.x,r.y)<=(r.z,r.w)
__device__ __forceinline__ int4 top2_merge(int4 p, int4 r) {
    const bool pf = pair_less(p.x, p.y, r.x, r.y);
    const int b0d = pf ? p.x : r.x, b0i = pf ? p.y : r.y;  // overall best
    const int cd = pf ? r.x : p.x, ci = pf ? r.y : p.y;    // loser of the heads
    const int nd = pf ? p.z : r.z, ni = pf ? p.w : r.w;    // second of the winner's list
    const bool sf = pair_less(cd, ci, nd, ni);
    return make_int4(b0d, b0i, sf ? cd : nd, sf ? ci : ni);
}
__global__ __launch_bounds__(256) void k_knn2(const uint4* __restrict__ q, int nq, const uint4* __restrict__ t, int nt,
                                              u64* __restrict__ part, int32_t* __restrict__ arrive,
                                              int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist) {
    __shared__ int4 lpart[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.x * 64 + lane;
    const int qc = min(qi, nq - 1);
    const uint4 qa = q[2 * (size_t)qc], qb = q[2 * (size_t)qc + 1];
    const int slice = (nt + MK_SLICES - 1) / MK_SLICES;
    const int j0 = (blockIdx.y * 4 + wave) * slice, j1 = min(nt, j0 + slice);
    Top2 b = {INT_MAX, -1, INT_MAX, -1};
    for (int c0 = j0; c0 < j1; c0 += 64) {
        const int cn = min(64, j1 - c0);  // wave-uniform
        const int tl = min(c0 + lane, nt - 1);
        const uint4 ta = t[2 * (size_t)tl], tb = t[2 * (size_t)tl + 1];
        for (int j = 0; j < cn; ++j) {  // (v_readlane with a scalar lane index; the loop is not unrollable)
            int d = __popc(qa.x ^ rl(ta.x, j)) + __popc(qa.y ^ rl(ta.y, j)) + __popc(qa.z ^ rl(ta.z, j)) +
                    __popc(qa.w ^ rl(ta.w, j)) + __popc(qb.x ^ rl(tb.x, j)) + __popc(qb.y ^ rl(tb.y, j)) +
                    __popc(qb.z ^ rl(tb.z, j)) + __popc(qb.w ^ rl(tb.w, j));
            top2_insert(b, d, c0 + j);
        }
    }
    lpart[wave][lane] = make_int4(b.d0, b.i0, b.d1, b.i1);
    __syncthreads();
    if (wave != 0) return;
    int4 p = lpart[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) p = top2_merge(p, lpart[w][lane]);
    // partial of (workgroup, query): two 8-byte {distance, index} words
    u64* mine = part + 2 * ((size_t)blockIdx.y * nq + qc);
    if (qi < nq) {
        __hip_atomic_store(mine, ((u64)(uint32_t)p.y << 32) | (uint32_t)p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, ((u64)(uint32_t)p.w << 32) | (uint32_t)p.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores are complete before this workgroup is counted
    int last = 0;
    if (lane == 0) {
        last = __hip_atomic_fetch_add(arrive + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == MK_GROUPS - 1;
        if (last) __hip_atomic_store(arrive + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    u64 w0[MK_GROUPS], w1[MK_GROUPS];
#pragma unroll
    for (int s = 0; s < MK_GROUPS; ++s) {
        const u64* src = part + 2 * ((size_t)s * nq + qc);
        w0[s] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1[s] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    p = make_int4(INT_MAX, -1, INT_MAX, -1);
#pragma unroll
    for (int s = 0; s < MK_GROUPS; ++s)
        p = top2_merge(p, make_int4((int)(uint32_t)w0[s], (int)(uint32_t)(w0[s] >> 32), (int)(uint32_t)w1[s], (int)(uint32_t)(w1[s] >> 32)));
    if (qi < nq) {
        out_idx[2 * qi] = p.y;
        out_idx[2 * qi + 1] = p.w;
        out_dist[2 * qi] = p.y >= 0 ? p.x : INT_MAX;
        out_dist[2 * qi + 1] = p.w >= 0 ? p.z : INT_MAX;
    }
}

__global__ __launch_bounds__(1024) void k_radius_l1(const uint32_t* __restrict__ q, const float2* __restrict__ qxy,
                                                    int nq, const uint32_t* __restrict__ t,
                                                    const float2* __restrict__ txy, int nt, float r2,
                                                    int32_t* __restrict__ out_idx, int32_t* __restrict__ out_sum) {
    __shared__ int2 part[MK_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.x * 64 + lane;
    const int qc = min(qi, nq - 1);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = q[8 * (size_t)qc + k];
    const float2 p = qxy[qc];
    const int slice = (nt + MK_WAVES - 1) / MK_WAVES;
    const int j0 = wave * slice, j1 = min(nt, j0 + slice);
    int best = INT_MAX, bi = -1;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const float2 p2 = txy[j];
        const float dx = __fsub_rn(p.x, p2.x), dy = __fsub_rn(p.y, p2.y);
        const bool in = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) <= r2;
        const uint32_t* tj = t + 8 * (size_t)j;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = __builtin_amdgcn_sad_u8(qd[k], tj[k], s);
        const bool take = in && (int)s < best;  // strict '<': the first minimum wins
        best = take ? (int)s : best;
        bi = take ? j : bi;
    }
    part[wave][lane] = make_int2(best, bi);
    __syncthreads();
    if (wave == 0 && qi < nq) {
        int rb = part[0][lane].x, ri = part[0][lane].y;
#pragma unroll
        for (int w = 1; w < MK_WAVES; ++w) {
            int2 c = part[w][lane];
            bool take = c.y >= 0 && c.x < rb;
            rb = take ? c.x : rb;
            ri = take ? c.y : ri;
        }
        out_idx[qi] = ri;
        out_sum[qi] = rb;
    }
}

// d_out: device scratch (partials behind nq x 4 int32); final (optional): where the merged (idx, dist) block goes --
// a pinned host buffer lets the kernel deliver the result itself (no copy dispatch on the frame's critical path)
int match_launch_knn2(mvo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_out,
                      int32_t* final_out) {
    if (nq <= 0) return MVO_OK;
    // MK_GROUPS x nq 16-byte partials live behind the nq x 4 int32 result block
    u64* d_part = reinterpret_cast<u64*>(d_out + 4 * (size_t)nq);
    int32_t* dst = final_out ? final_out : d_out;
    ProfScope ps(ctx, "k_knn2");
    hipLaunchKernelGGL(k_knn2, dim3((nq + 63) / 64, MK_GROUPS), dim3(256), 0, ctx->stream, (const uint4*)d_q, nq,
                       (const uint4*)d_t, nt, d_part, ctx->d_marrive, dst, dst + 2 * (size_t)nq);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int match_launch_radius_l1(mvo_ctx* ctx, const uint8_t* d_q, const float* d_qxy, int nq, const uint8_t* d_t,
                           const float* d_txy, int nt, float max_px, int32_t* d_out) {
    if (nq <= 0) return MVO_OK;
    ProfScope ps(ctx, "k_radius_l1");
    hipLaunchKernelGGL(k_radius_l1, dim3((nq + 63) / 64), dim3(1024), 0, ctx->stream, (const uint32_t*)d_q,
                       (const float2*)d_qxy, nq, (const uint32_t*)d_t, (const float2*)d_txy, nt,
                       max_px * max_px, d_out, d_out + nq);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}
