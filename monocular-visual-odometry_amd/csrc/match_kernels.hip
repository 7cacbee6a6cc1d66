// csrc/match_kernels.hip -- all-pairs descriptor matching on gfx950.
//   k_knn2      exact 2-NN in 256-bit Hamming space: replaces cv::BFMatcher("BruteForce-Hamming")::knnMatch(k=2)
//               (reference src/geometry/feature_match.cpp:141,203-208) and, as the exact 1-NN, the
//               cv::FlannBasedMatcher(LshIndexParams(5,10,2))::match call at feature_match.cpp:140,162.
//   k_radius_l1 geometry::matchByRadiusAndBruteForce (feature_match.cpp:86-124).
//   k_knn2_mfma the same 2-NN as an i8 Gram on the matrix cores (default; see its header below).
// k_knn2 (vector ALU, kept as the second implementation the tests compare and for train sets of >= 65536 descriptors):
// one LANE per query, the query's 256 bits in 8 VGPRs; a wave parks its train slice in registers (lane j = descriptor j)
// and broadcasts descriptor j with v_readlane, distance = 8 x (v_xor + v_bcnt_u32); one wave per (64 queries, train
// slice) pair, 32 slices -> 1024 waves for 2000 x 2000 in workgroups of four; LDS fold of the four, then the last
// workgroup to arrive for a query group folds the 8 partial (best, second) pairs, keeping the lexicographically smallest
// (distance, index) pairs, which reproduces cv::batchDistance's tie rule exactly (equal distances keep the lower train
// index).  k_radius_l1 keeps the single-kernel form (16 waves x 64 queries, LDS merge).  Everything is integer: results
// are bit-exact.  The whole working set (<= 2 x 128 KB) is L2-resident: no HBM bound.
#include "mvo_internal.h"

#include <climits>
#include <cstdlib>
#include <algorithm>

typedef unsigned long long u64;

#define MK_WAVES 16

int g_match_mfma = 1;  // test hook: 0 = the vector-ALU kernel k_knn2 for every call
// trains per slice of k_knn2_mfma in THROUGHPUT-mode contexts (latency mode: MM_TS = 256): a slice's LDS image (272 B per
// train) decides how many workgroups share a CU -- 256 trains = 68 KB = two workgroups (8 waves) per CU, fine for a lone
// frame on an empty chip, a waste of the few CUs the resident solver grid leaves to the extraction
int g_match_slice_throughput = std::getenv("MVO_MATCH_SLICE") ? std::atoi(std::getenv("MVO_MATCH_SLICE")) : 256;

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
    MVO_WAIT_VM0();  // the write-through stores are complete before this workgroup is counted
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

// ---------------------------------------------------------------------------------------------------------------------
// k_knn2_mfma: the same exact 2-NN on the INTEGER MATRIX CORES.  With every descriptor bit b mapped to the byte
// s = 1 - 2 b (+1 / -1), the dot product of two descriptors is 256 - 2 hamming: a 2000 x 2000 x 256 Gram in i8 with
// exact i32 accumulation (v_mfma_i32_16x16x64_i8: four instructions per 16 x 16 tile of pairs) instead of
// 8 x (v_readlane + v_xor + v_bcnt) per pair on the vector ALU.
//   grid = (groups of 64 queries) x (train slices); workgroup = 4 waves, wave w owns queries 16 w .. 16 w + 15 of the group.
//   prologue: the workgroup expands its train slice (<= 256 descriptors) into LDS, 256 bytes per descriptor in bit order
//             (row pitch 272: the 16 rows of a tile start on different banks); every lane expands the 4 x 16 bits of ITS
//             query column that the four K = 64 steps need (B operand: lane = (column l & 15, k block l >> 4)).
//   loop:     per tile of 16 trains: 4 x (ds_read_b128 A operand + MFMA); lane (column l & 15, rows 4 (l >> 4) + r)
//             turns its four dot products into keys (hamming << 16 | train index) and folds them into its running
//             (best, second) with v_min / v_max -- the lexicographic (distance, index) order IS cv::batchDistance's tie
//             rule (strict '<' in index order: equal distances keep the lower train index).
//   epilogue: the four lanes of a column merge by shuffles, the slice partials meet like in k_knn2 (write-through
//             partial + arrival counter, the last workgroup of a query group folds and delivers).
// Exact integers throughout; needs nt < 65536 (16-bit index in the key), else k_knn2 runs.
typedef int v4i __attribute__((ext_vector_type(4)));
#define MM_TS 256      // trains per slice (LDS: 256 x 272 B = 68 KB)
#define MM_PITCH 272
__device__ __forceinline__ uint32_t mm_spread(uint32_t nib) {  // 4 bits -> 4 bytes of +1 (0x01) / -1 (0xff)
    const uint32_t x = (nib | (nib << 7) | (nib << 14) | (nib << 21)) & 0x01010101u;
    return (x * 0xffu) | 0x01010101u;
}
__device__ __forceinline__ void mm_fold(uint32_t& b0, uint32_t& b1, uint32_t key) {
    b1 = min(b1, max(b0, key));
    b0 = min(b0, key);
}
__global__ __launch_bounds__(256) void k_knn2_mfma(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t, int nt,
                                                   int slice, u64* __restrict__ part, int32_t* __restrict__ arrive,
                                                   int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist) {
    MVO_DYN_LDS_ALIGNED16(unsigned char, mm_lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.y * slice, jn = max(0, min(slice, nt - j0));  // trains j0 .. j0 + jn - 1
    const int ntile = (jn + 15) / 16;
    // ---- expand the slice: item = (train row, source dword): 32 bits -> 32 bytes
    for (int it = tid; it < ntile * 16 * 8; it += 256) {
        const int r = it >> 3, w = it & 7;
        const uint32_t bits = r < jn ? t[8 * (size_t)(j0 + r) + w] : 0u;
        uint4 lo, hi;
        lo.x = mm_spread(bits & 15u), lo.y = mm_spread((bits >> 4) & 15u), lo.z = mm_spread((bits >> 8) & 15u), lo.w = mm_spread((bits >> 12) & 15u);
        hi.x = mm_spread((bits >> 16) & 15u), hi.y = mm_spread((bits >> 20) & 15u), hi.z = mm_spread((bits >> 24) & 15u), hi.w = mm_spread(bits >> 28);
        uint4* dst = reinterpret_cast<uint4*>(mm_lds + (size_t)r * MM_PITCH + 32 * w);
        dst[0] = lo;
        dst[1] = hi;
    }
    // ---- this lane's query column: bits [64 m + 16 kb, +16) for the four K steps
    const int col = lane & 15, kb = lane >> 4;
    const int qi = blockIdx.x * 64 + 16 * wave + col;
    const int qc = min(qi, nq - 1);
    v4i bq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t wsrc = q[8 * (size_t)qc + 2 * m + (kb >> 1)];
        const uint32_t h = (kb & 1) ? (wsrc >> 16) : (wsrc & 0xffffu);
        bq[m][0] = (int)mm_spread(h & 15u);
        bq[m][1] = (int)mm_spread((h >> 4) & 15u);
        bq[m][2] = (int)mm_spread((h >> 8) & 15u);
        bq[m][3] = (int)mm_spread(h >> 12);
    }
    __syncthreads();
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu;
    for (int tile = 0; tile < ntile; ++tile) {
        const unsigned char* row = mm_lds + (size_t)(16 * tile + col) * MM_PITCH + 16 * kb;  // (A operand: row l & 15 = train)
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const v4i a = *reinterpret_cast<const v4i*>(row + 64 * m);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq[m], acc, 0, 0, 0);
        }
        // acc[r] = dot(train 16 tile + 4 kb + r, query col)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = 16 * tile + 4 * kb + r;
            const uint32_t d = (uint32_t)(256 - acc[r]) >> 1;
            const uint32_t key = jl < jn ? ((d << 16) | (uint32_t)(j0 + jl)) : 0xffffffffu;
            mm_fold(b0, b1, key);
        }
    }
    // ---- the four lanes of a column
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const uint32_t o0 = (uint32_t)__shfl_xor((int)b0, o), o1 = (uint32_t)__shfl_xor((int)b1, o);
        const uint32_t c1 = min(max(b0, o0), min(b1, o1));
        b0 = min(b0, o0);
        b1 = c1;
    }
    // ---- slice partials -> the last workgroup of the query group folds them (arrival counter per group, self re-arming)
    u64* mine = part + ((size_t)blockIdx.y * nq + qc);
    if (kb == 0 && qi < nq) __hip_atomic_store(mine, ((u64)b1 << 32) | b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MVO_WAIT_VM0();
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
        const int nsl = gridDim.y;
        const int last = __hip_atomic_fetch_add(arrive + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsl - 1;
        if (last) __hip_atomic_store(arrive + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    const int qo = blockIdx.x * 64 + tid;
    if (qo >= nq) return;
    b0 = b1 = 0xffffffffu;
    for (int s2 = 0; s2 < (int)gridDim.y; ++s2) {
        const u64 w = __hip_atomic_load(part + ((size_t)s2 * nq + qo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t o0 = (uint32_t)w, o1 = (uint32_t)(w >> 32);
        const uint32_t c1 = min(max(b0, o0), min(b1, o1));
        b0 = min(b0, o0);
        b1 = c1;
    }
    const bool h0 = b0 != 0xffffffffu, h1 = b1 != 0xffffffffu;
    out_idx[2 * qo] = h0 ? (int)(b0 & 0xffffu) : -1;
    out_idx[2 * qo + 1] = h1 ? (int)(b1 & 0xffffu) : -1;
    out_dist[2 * qo] = h0 ? (int)(b0 >> 16) : INT_MAX;
    out_dist[2 * qo + 1] = h1 ? (int)(b1 >> 16) : INT_MAX;
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
    if (g_match_mfma && nt > 0 && nt < 65536) {
        // slices of <= MM_TS trains, at most 64 of them (the partial area holds 64 x nq x 16 B)
        const int ts = ctx->ba_throughput_mode ? std::max(16, std::min(MM_TS, g_match_slice_throughput & ~15)) : MM_TS;
        int nsl = (nt + ts - 1) / ts;
        if (nsl > 64) nsl = 64;
        int slice = ((nt + nsl - 1) / nsl + 15) & ~15;
        if (slice <= MM_TS) {
            ProfScope ps(ctx, "k_knn2_mfma");
            static const int lds_ok = hipFuncSetAttribute((const void*)k_knn2_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, MM_TS * MM_PITCH);
            (void)lds_ok;
            nsl = (nt + slice - 1) / slice;
            hipLaunchKernelGGL(k_knn2_mfma, dim3((nq + 63) / 64, nsl), dim3(256), (size_t)slice * MM_PITCH, ctx->stream, (const uint32_t*)d_q, nq,
                               (const uint32_t*)d_t, nt, slice, d_part, ctx->d_marrive, dst, dst + 2 * (size_t)nq);
            MVO_HIP(hipGetLastError());
            return MVO_OK;
        }
    }
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
