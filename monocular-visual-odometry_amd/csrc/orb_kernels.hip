// csrc/orb_kernels.hip -- gfx950 kernels for ORB extraction (replaces cv::ORB::detect / ::compute as called
// from the reference's src/geometry/feature_match.cpp:22-23,34,45,48).
//
// Data layout in HBM: ONE byte buffer for the raw gray pyramid (a second one for whole-level blur, filled on debug
// request only), every level stored with its 32-px BORDER_REFLECT_101 frame, row stride rounded up to 64 B so that every
// tile row starts dword-aligned and a 64-lane wave reads one aligned 64-B segment per row.  All kernels are integer/byte
// work bounded by HBM/L2 traffic (in practice by latency: the time a wave stays resident); four or five launches per frame:
//   k_pyramid       every level of a group of up to four levels in ONE launch: workgroup = one 64x16 tile of one level;
//                   the tile's footprint is traced down the bilinear chain to the group's base (the BGR image, or the
//                   last level of the previous group) -- the regions come from a per-tile table the host makes once per image
//                   geometry --, the base region is converted once into LDS, every intermediate level's region is built in
//                   LDS, the tile is written from the region one level below (lane = column, wave = every fourth row).  No
//                   level waits for another one: no inter-workgroup dependency, no launch per level.  k_pyramid_chain (every
//                   pixel walks the chain itself, 4^depth base samples) is the fallback when the regions do not fit LDS
//   k_fast_harris   the 64x16 tile plus a 15-row / 16-column apron staged in LDS; FAST-9/16 quick test on every pixel, the few
//                   that pass are queued and scored densely; 3x3 NMS + 31-px border (one 64-bit survivor mask per tile row
//                   via wave ballot); then ONE WAVE per survivor of the tile, entirely on the staged bytes: 7x7 Harris (49
//                   lanes) + IC angle over the 749-px disc, wave sums by DPP row shifts; the finished 16-byte records go into
//                   the tile's slot of a pinned host buffer (no candidate compaction on the device, no copy dispatch)
//   k_brief         one WAVE per keypoint: the 45x56 raw window is staged in LDS, blurred there (separable 7x7
//                   fixed-point Gaussian, the keypoint's window only) and sampled: 512 rotated taps, 4 ballots = 256 bits
//   k_blur          whole-level blur (dword LDS reads, 7-tap sums as two v_dot4_u32_u8) + k_brief_sample: the same descriptors
//                   for a ctx in throughput mode (a sixth of the instructions per frame); also behind mvo_debug_get_level(blurred)
// Compiled with -ffp-contract=off: the float expressions (Harris response, fastAtan2, tap rotation) are
// canonical arithmetic and must round exactly like the oracle.
#include "mvo_internal.h"
#include "orb_pattern_31.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

typedef unsigned long long u64;

__constant__ signed char c_pattern[256 * 4];
__constant__ signed char c_disc[768 * 2];  // (u, v) of the 749 pixels of the IC-angle disc
__constant__ int c_disc_n;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ------------------------------------------------------------------------------------------------ pyramid
// The base of a level group: the caller's image (gray conversion on the fly) or an already built pyramid level.
struct PyrBase {
    const uint8_t* img;  // BGR / BGRA / gray image, or nullptr: the base is raw level `level`
    int istride, ch;
    int level;
};

// Interior pixel (x, y) of level l, D levels above the base.  cv::cvtColor(BGR2GRAY) 8-bit: (1868 B + 9617 G +
// 4899 R + 2^13) >> 14; cv::resize 8-bit fixed point with the per-geometry coefficient tables -- exact:
// INTER_LINEAR_EXACT (8-bit coefficients, one rounding), else INTER_LINEAR (11-bit coefficients, OpenCV's truncating
// vertical pass).  Every level is rounded to 8 bits exactly as if it had been stored, so the value does not depend on
// how many levels are evaluated in one go.
template <int D>
__device__ __forceinline__ int pyr_px(const PyrBase& B, const uint8_t* __restrict__ raw, const PyrInfo& P,
                                      const ResizeEntry* __restrict__ tabs, int l, int x, int y, int exact) {
    if constexpr (D == 0) {
        if (B.img) {
            const uint8_t* px = B.img + (size_t)y * B.istride + x * B.ch;
            return B.ch == 1 ? (int)px[0] : (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14;
        }
        const LevelInfo& L = P.lv[l];
        return raw[L.off + (size_t)(y + MVO_BORDER) * L.stride + MVO_BORDER + x];
    } else {
        const LevelInfo& Dl = P.lv[l];
        const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h;
        const ResizeEntry tx = tabs[Dl.tab_off + x], ty = tabs[Dl.tab_off + Dl.w + y];
        const int sx1 = min(tx.ofs + 1, sw - 1), sy1 = min(ty.ofs + 1, sh - 1);
        const int p00 = pyr_px<D - 1>(B, raw, P, tabs, l - 1, tx.ofs, ty.ofs, exact);
        const int p01 = pyr_px<D - 1>(B, raw, P, tabs, l - 1, sx1, ty.ofs, exact);
        const int p10 = pyr_px<D - 1>(B, raw, P, tabs, l - 1, tx.ofs, sy1, exact);
        const int p11 = pyr_px<D - 1>(B, raw, P, tabs, l - 1, sx1, sy1, exact);
        const int h0 = p00 * tx.c0 + p01 * tx.c1, h1 = p10 * tx.c0 + p11 * tx.c1;
        return exact ? ((ty.c0 * h0 + ty.c1 * h1 + 32768) >> 16) & 0xff
                     : ((((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
    }
}

// Fallback form (k_pyramid_chain): one thread = 4 bordered pixels (one dword) of one level, every pixel walks the whole
// chain itself (4^depth base samples).  Used when a level group's source regions do not fit the LDS budget of the
// tiled kernel below (scale factors far above the reference's 1.2) and by the tests as a second implementation.
#define PYR_GROUP 4
__global__ __launch_bounds__(256) void k_pyramid_chain(PyrBase B, uint8_t* __restrict__ raw, PyrInfo P,
                                                       const ResizeEntry* __restrict__ tabs, int l0, int nl, int exact) {
    int l = l0;
    for (int k = 1; k < PYR_GROUP; ++k)
        if (k < nl && (int)blockIdx.x >= P.lv[l0 + k].btile_off - P.lv[l0].btile_off) l = l0 + k;
    const LevelInfo L = P.lv[l];
    const int b = blockIdx.x - (L.btile_off - P.lv[l0].btile_off);
    const int tx = b % L.btiles_x, ty = b / L.btiles_x;
    const int x4 = tx * PT_W + (threadIdx.x & 15) * 4;
    const int by = ty * PT_H + (threadIdx.x >> 4);
    if (by >= L.h + 2 * MVO_BORDER) return;
    const int y = reflect101(by - MVO_BORDER, L.h);
    const int depth = l - B.level;  // workgroup-uniform
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const int bxk = x4 + k;
        uint32_t g = 0;
        if (bxk < L.w + 2 * MVO_BORDER) {
            const int x = reflect101(bxk - MVO_BORDER, L.w);
            switch (depth) {
                case 0: g = (uint32_t)pyr_px<0>(B, raw, P, tabs, l, x, y, exact); break;
                case 1: g = (uint32_t)pyr_px<1>(B, raw, P, tabs, l, x, y, exact); break;
                case 2: g = (uint32_t)pyr_px<2>(B, raw, P, tabs, l, x, y, exact); break;
                case 3: g = (uint32_t)pyr_px<3>(B, raw, P, tabs, l, x, y, exact); break;
                default: g = (uint32_t)pyr_px<4>(B, raw, P, tabs, l, x, y, exact); break;
            }
        }
        out |= g << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(raw + L.off + (size_t)by * L.stride + x4) = out;
}

// k_pyramid: one workgroup = one 64 x 16 tile (bordered coordinates) of one level of the group.  The tile's interior
// footprint (after BORDER_REFLECT_101) is traced down the bilinear chain to the base: the base region is converted /
// fetched ONCE into LDS, every intermediate level's region is built in LDS from the one below, and the tile is written
// from the region of level l-1.  A level-3 tile re-derives ~10 base-level pixels per output instead of 64 and never
// touches global memory in between; no workgroup depends on another one.
__device__ __forceinline__ int pyr_sample(const uint8_t* __restrict__ S, const PyrRegion& R, int sw, int sh,
                                          const ResizeEntry tx, const ResizeEntry ty, int exact) {
    const int sx1 = min(tx.ofs + 1, sw - 1), sy1 = min(ty.ofs + 1, sh - 1);
    const uint8_t* r0 = S + (ty.ofs - R.y0) * R.w - R.x0;
    const uint8_t* r1 = S + (sy1 - R.y0) * R.w - R.x0;
    const int h0 = r0[tx.ofs] * tx.c0 + r0[sx1] * tx.c1, h1 = r1[tx.ofs] * tx.c0 + r1[sx1] * tx.c1;
    return exact ? ((ty.c0 * h0 + ty.c1 * h1 + 32768) >> 16) & 0xff
                 : ((((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
}

__global__ __launch_bounds__(256) void k_pyramid(PyrBase B, uint8_t* __restrict__ raw, PyrInfo P,
                                                 const ResizeEntry* __restrict__ tabs, int l0, int nl, int exact,
                                                 const PyrTileRegs* __restrict__ tile_regs) {
    // region pool: dynamic LDS sized by the host for the hungriest tile of THIS launch (orb_setup_geometry: ~9 KB for the
    // 4-level 640 x 480 pyramid; the fixed 32 KB pool of round 3 let only four workgroups share a CU)
    MVO_DYN_LDS(uint8_t, lds);
    const int tid = threadIdx.x;
    int l = l0;
    for (int k = 1; k < PYR_GROUP; ++k)
        if (k < nl && (int)blockIdx.x >= P.lv[l0 + k].btile_off - P.lv[l0].btile_off) l = l0 + k;
    const LevelInfo L = P.lv[l];
    const int b = blockIdx.x - (L.btile_off - P.lv[l0].btile_off);
    const int tx = b % L.btiles_x, ty = b / L.btiles_x;
    const int bx0 = tx * PT_W, by0 = ty * PT_H;
    const int bw = L.w + 2 * MVO_BORDER, bh = L.h + 2 * MVO_BORDER;
    const int depth = l - B.level;  // workgroup-uniform
    const int ox4 = bx0 + (tid & 15) * 4, oy = by0 + (tid >> 4);  // this thread's output dword
    if (depth == 0) {  // level 0 straight from the image
        if (oy >= bh) return;
        const int y = reflect101(oy - MVO_BORDER, L.h);
        const int xi = ox4 - MVO_BORDER;  // interior x of the first of the four pixels (a multiple of 4)
        if (B.ch == 3 && xi >= 0 && xi + 4 <= L.w) {
            // four interior BGR pixels = 12 consecutive bytes: three dword loads when the row is dword-aligned
            const uint8_t* px = B.img + (size_t)y * B.istride + 3 * xi;
            if ((reinterpret_cast<uintptr_t>(px) & 3) == 0) {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(px);
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
                auto g = [](uint32_t b, uint32_t gg, uint32_t r) { return (b * 1868u + gg * 9617u + r * 4899u + 8192u) >> 14; };
                const uint32_t o = g(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255) | g(w0 >> 24, w1 & 255, (w1 >> 8) & 255) << 8 |
                                   g((w1 >> 16) & 255, w1 >> 24, w2 & 255) << 16 | g((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24) << 24;
                *reinterpret_cast<uint32_t*>(raw + L.off + (size_t)oy * L.stride + ox4) = o;
                return;
            }
        }
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ox4 + k < bw) out |= (uint32_t)pyr_px<0>(B, raw, P, tabs, l, reflect101(ox4 + k - MVO_BORDER, L.w), y, exact) << (8 * k);
        *reinterpret_cast<uint32_t*>(raw + L.off + (size_t)oy * L.stride + ox4) = out;
        return;
    }
    // ---- the regions of the levels l-1 .. l-depth: this tile's entry of the table the host made with pyr_regions() when the
    // image geometry was set up (the same function it checks the LDS fit with)
    __shared__ PyrTileRegs sregs;
    if (tid < 32) reinterpret_cast<int*>(&sregs)[tid] = reinterpret_cast<const int*>(tile_regs + (blockIdx.x + P.lv[l0].btile_off))[tid];
    __syncthreads();
    const PyrRegion* reg = sregs.reg;
    const int* regoff = sregs.off;
    // Mapping of the region loops and of the tile: lane = column (two columns per lane for regions wider than 64), wave =
    // every fourth row.  What depends on the column alone -- its resize-table entry, the clamped neighbour, the byte offset in
    // the image -- is made ONCE per lane instead of once per sample, what depends on the row alone is wave-uniform (scalar
    // loads), and no sample pays for an index -> (row, column) split: ~14 vector instructions per bilinear sample instead of
    // ~40, ~9 per gray conversion instead of ~25 (the kernel was instruction-bound on the few CUs the solver grid leaves).
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- base region
    {
        const PyrRegion R = reg[depth];
        uint8_t* dst = lds + regoff[depth];
        for (int cx = lane; cx < R.w; cx += 64) {
            if (B.img) {
                const uint8_t* col = B.img + (size_t)R.y0 * B.istride + (size_t)(R.x0 + cx) * B.ch;
                for (int ry = wv; ry < R.h; ry += 4) {
                    const uint8_t* px = col + (size_t)ry * B.istride;
                    dst[ry * R.w + cx] = (uint8_t)(B.ch == 1 ? (int)px[0] : (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
                }
            } else {
                const LevelInfo& Lb = P.lv[B.level];
                const uint8_t* col = raw + Lb.off + (size_t)(R.y0 + MVO_BORDER) * Lb.stride + MVO_BORDER + R.x0 + cx;
                for (int ry = wv; ry < R.h; ry += 4) dst[ry * R.w + cx] = col[(size_t)ry * Lb.stride];
            }
        }
    }
    __syncthreads();
    // ---- intermediate levels, bottom-up: the region of level l-d from the region of level l-d-1
    for (int d = depth - 1; d >= 1; --d) {
        const int m = l - d;  // level being produced
        const PyrRegion R = reg[d], S = reg[d + 1];
        const uint8_t* src = lds + regoff[d + 1];
        uint8_t* dst = lds + regoff[d];
        const int sw = P.lv[m - 1].w, sh = P.lv[m - 1].h, toff = P.lv[m].tab_off, mw = P.lv[m].w;
        for (int cx = lane; cx < R.w; cx += 64) {
            const ResizeEntry tx = tabs[toff + R.x0 + cx];
            const int sx0 = tx.ofs - S.x0, sx1 = min(tx.ofs + 1, sw - 1) - S.x0;
            for (int ry = wv; ry < R.h; ry += 4) {
                const ResizeEntry ty = tabs[toff + mw + R.y0 + ry];
                const uint8_t* r0 = src + (ty.ofs - S.y0) * S.w;
                const uint8_t* r1 = src + (min(ty.ofs + 1, sh - 1) - S.y0) * S.w;
                const int h0 = r0[sx0] * tx.c0 + r0[sx1] * tx.c1, h1 = r1[sx0] * tx.c0 + r1[sx1] * tx.c1;
                dst[ry * R.w + cx] = (uint8_t)(exact ? ((ty.c0 * h0 + ty.c1 * h1 + 32768) >> 16) & 0xff
                                                     : ((((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff);
            }
        }
        __syncthreads();
    }
    // ---- the tile itself, frame included: lane = bordered column bx0 + lane, wave w = rows by0 + w, w + 4, ...
    {
        const PyrRegion S = reg[1];
        const uint8_t* src = lds + regoff[1];
        const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h;
        const int ox = bx0 + lane;
        const bool live = ox < bw;  // (the columns between the bordered width and the 64-byte row stride are written as zeros)
        const ResizeEntry txE = tabs[L.tab_off + reflect101(min(ox, bw - 1) - MVO_BORDER, L.w)];
        const int sx0 = txE.ofs - S.x0, sx1 = min(txE.ofs + 1, sw - 1) - S.x0;
        uint8_t* out = raw + L.off + ox;
        for (int ry = wv; ry < PT_H; ry += 4) {
            const int oyy = by0 + ry;
            if (oyy >= bh) break;
            const ResizeEntry tyE = tabs[L.tab_off + L.w + reflect101(oyy - MVO_BORDER, L.h)];
            const uint8_t* r0 = src + (tyE.ofs - S.y0) * S.w;
            const uint8_t* r1 = src + (min(tyE.ofs + 1, sh - 1) - S.y0) * S.w;
            const int h0 = r0[sx0] * txE.c0 + r0[sx1] * txE.c1, h1 = r1[sx0] * txE.c0 + r1[sx1] * txE.c1;
            const int v = exact ? ((tyE.c0 * h0 + tyE.c1 * h1 + 32768) >> 16) & 0xff
                                : ((((tyE.c0 * (h0 >> 4)) >> 16) + ((tyE.c1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
            out[(size_t)oyy * L.stride] = (uint8_t)(live ? v : 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ FAST + NMS
#define FT_W 64
#define FT_H 16
// The staged patch: the tile plus what its survivors read around them -- 15 rows above / below (the intensity-centroid disc), 16
// columns left / right (a dword-aligned start; the disc needs 15) --, so that the whole survivor phase (7 x 7 Harris block, disc)
// works on LDS: no global round trip per survivor (the waves of this kernel spent two thirds of their cycles waiting).
#define FT_HX 16
#define FT_HY 15
#define FT_PW (FT_W + 2 * FT_HX)  // bytes per LDS row (24 dwords)
#define FT_PH (FT_H + 2 * FT_HY)
#define FT_SW (FT_W + 2)
#define FT_SH (FT_H + 2)

__device__ __forceinline__ int find_level_by(const PyrInfo& P, int idx, int which) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < MVO_MAX_LEVELS; ++k) {
        if (k < P.nlevels) {
            int start = which == 0 ? P.lv[k].tile_off : (which == 1 ? P.lv[k].cell_off : P.lv[k].btile_off);
            if (idx >= start) l = k;
        }
    }
    return l;
}

// cornerScore<16> in two halves.  d[k] = centre - circle[k].
// fast_quick_test: can the pixel be a FAST-9 corner at threshold thr at all?  (bit masks of the circle pixels darker / brighter
// than the centre by more than thr; nine consecutive set bits in either)
__device__ __forceinline__ bool fast_quick_test(const int (&d)[16], int thr) {
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        dark |= (uint32_t)(d[k] > thr) << k;
        bright |= (uint32_t)(d[k] < -thr) << k;
    }
    auto has9 = [](uint32_t m) {
        uint32_t x = m | (m << 16);
        uint32_t a = x & (x >> 1);
        uint32_t b = a & (a >> 2);
        uint32_t c = b & (b >> 4);
        return (c & (x >> 8) & 0xffffu) != 0;
    };
    return has9(dark) || has9(bright);
}
// fast_score_full: the largest threshold for which the pixel is still a FAST-9 corner (for a pixel that passed the quick test):
// max over the 16 arcs of 9 consecutive circle pixels of min(d) and of min(-d), by a sliding minimum / maximum with doubling.
__device__ __forceinline__ int fast_score_full(const int (&d)[16], int thr) {
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn2[k] = min(d[k], d[(k + 1) & 15]);
        mx2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
        mx4[k] = max(mx2[k], mx2[(k + 2) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn8[k] = min(mn4[k], mn4[(k + 4) & 15]);
        mx8[k] = max(mx4[k], mx4[(k + 4) & 15]);
    }
    int A = -256, B = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        A = max(A, min(mn8[k], d[(k + 8) & 15]));
        B = max(B, -max(mx8[k], d[(k + 8) & 15]));
    }
    int best = max(A, B);
    return best > thr ? best - 1 : 0;
}

// cv::fastAtan2 (degrees)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int wave_sum(int v) {
#ifndef MVO_KERNEL_SIM
    // Inclusive scan inside each row of 16 lanes by DPP row shifts (a lane outside the row contributes 0), then the row totals
    // are carried over by the two row broadcasts: lane 63 holds the sum of the wave and is read back as a scalar.  Six VALU
    // instructions instead of six ds_bpermute round trips (each ~60 cycles of dependent latency: five sums per survivor were the
    // longest chain of the survivor phase).  Integer adds: the order does not matter.
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8   -> lane 15 of every row: the row's total
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3 -> lane 63: the total
    return __builtin_amdgcn_readlane(v, 63);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#endif
}

// k_fast_harris: one 64 x 16 tile per workgroup, everything cv::ORB::detect computes per corner in ONE launch:
//   1. FAST-9/16 score of the tile + 1-px ring (from the staged patch in LDS: quick test everywhere, the pixels that pass are
//      queued and scored densely), 3x3 non-maximum suppression, 31-px image border: one 64-bit survivor mask per tile row
//      via wave ballot;
//   2. the tile's survivors, row-major (= the order cv::FAST emits inside the tile), one WAVE per survivor round-robin:
//      7x7 Harris response (49 lanes, Sobel sums) and the intensity-centroid angle over the 749-px disc, both read from
//      the staged patch (its apron covers the disc), wave sums by DPP -- the same integer sums and float expressions as
//      the stand-alone form had;
//   3. the finished 16-byte records go into the tile's slot in device memory, the 16 line counts beside them;
//   4. the workgroup that arrives LAST for its tile row (arrival counter per row, self re-arming) puts the row in order:
//      the global order (level, row, column) = the order cv::FAST emits interleaves the tiles of a tile row line by line
//      -- an exclusive scan over (line, tile) of the line counts gives every record its place -- and writes the row as ONE
//      contiguous list into the pinned host buffer, its length beside it.  The host appends ~100 row lists per frame
//      (orb_host.cpp) instead of walking 64000 (line, tile) cursors; rows do not wait for each other (a device-wide scan
//      would need a second launch or a serial single-workgroup tail: tried, 40-100 us of exposed latency).
//   order_rows == 0 (a ctx in throughput mode): steps 3/4 are skipped -- every tile writes its records and its count straight
//   into its own slot of the pinned buffer and the host thread interleaves the tiles of a row (8 us less kernel time per
//   frame for ~75 us more of the host thread).
__global__ __launch_bounds__(256) void k_fast_harris(const uint8_t* __restrict__ raw, PyrInfo P, int thr,
                                                     int32_t* __restrict__ row_count, DevCandidate* __restrict__ rows,
                                                     u64* __restrict__ dslots, u64* __restrict__ dline,
                                                     int32_t* __restrict__ arrive, int order_rows) {
    __shared__ uint32_t pix[FT_PH * (FT_PW / 4)];
    __shared__ uint8_t sc[FT_SH * 68];
    __shared__ u64 rowmask[FT_H];
    __shared__ int rowstart[FT_H + 1];
    const int tid = threadIdx.x;
    const int lvl = find_level_by(P, blockIdx.x, 0);
    const LevelInfo L = P.lv[lvl];
    const int t = blockIdx.x - L.tile_off;
    const int tx = t % L.tiles_x, ty = t / L.tiles_x;
    const int x0 = tx * FT_W, y0 = ty * FT_H;
    const uint8_t* base = raw + L.off;
    // stage FT_PH x FT_PW bytes; row/col origin = (y0 - FT_HY, x0 - FT_HX) in interior coordinates (inside the level's 32-px
    // frame and its 64-byte row stride for every tile)
    for (int i = tid; i < FT_PH * (FT_PW / 4); i += 256) {
        int r = i / (FT_PW / 4), c = i - r * (FT_PW / 4);
        const uint8_t* p = base + (size_t)(y0 - FT_HY + r + MVO_BORDER) * L.stride + (x0 - FT_HX + MVO_BORDER) + 4 * c;
        pix[i] = *reinterpret_cast<const uint32_t*>(p);
    }
    __syncthreads();
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(pix);
    // FAST circle (x, y), same enumeration as the oracle
    constexpr int CX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int CY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    // The quick test decides for every pixel of the tile + ring; the few that pass (a few per cent even on busy images) are
    // queued and scored DENSELY afterwards: scored in place, nearly every wave would walk the whole sliding-minimum code for
    // a handful of live lanes (that code was ~40 % of the kernel's instructions).
    __shared__ uint16_t fq[FT_SH * FT_SW];
    __shared__ int fq_n;
    if (tid == 0) fq_n = 0;
    __syncthreads();
    for (int i0 = 0; i0 < FT_SH * FT_SW; i0 += 256) {
        const int i = i0 + tid;
        bool pass = false;
        if (i < FT_SH * FT_SW) {
            int sr = i / FT_SW, scx = i - sr * FT_SW;
            const uint8_t* c = pb + (sr - 1 + FT_HY) * FT_PW + (scx - 1 + FT_HX);  // (ly, lx) = (sr-1, scx-1)
            int v = c[0];
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = v - (int)c[CX[k] + CY[k] * FT_PW];
            pass = fast_quick_test(d, thr);
            if (!pass) sc[sr * 68 + scx] = 0;
        }
        const u64 pm = __ballot(pass);
        if (pm) {  // (wave-uniform) one LDS atomic per wave, the lanes take consecutive places
            int at = 0;
            if ((tid & 63) == 0) at = __hip_atomic_fetch_add(&fq_n, (int)__popcll(pm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            at = __shfl(at, 0);
            if (pass) fq[at + (int)__popcll(pm & ((1ull << (tid & 63)) - 1))] = (uint16_t)i;
        }
    }
    __syncthreads();
    for (int q = tid; q < fq_n; q += 256) {
        const int i = fq[q];
        int sr = i / FT_SW, scx = i - sr * FT_SW;
        const uint8_t* c = pb + (sr - 1 + FT_HY) * FT_PW + (scx - 1 + FT_HX);
        int v = c[0];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - (int)c[CX[k] + CY[k] * FT_PW];
        sc[sr * 68 + scx] = (uint8_t)fast_score_full(d, thr);
    }
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the survivor loop below works on scalar addresses)
    for (int ly = wave; ly < FT_H; ly += 4) {
        const int gx = x0 + lane, gy = y0 + ly;
        const uint8_t* s = sc + (ly + 1) * 68 + (lane + 1);
        int v = s[0];
        bool flag = v > 0 && v > s[-1] && v > s[1] && v > s[-69] && v > s[-68] && v > s[-67] && v > s[67] &&
                    v > s[68] && v > s[69];
        flag = flag && gx >= 31 && gx < L.w - 31 && gy >= 31 && gy < L.h - 31;
        const u64 mask = __ballot(flag);
        if (lane == 0) rowmask[ly] = mask;
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 16 row counts
        const int cnt = tid < FT_H ? (int)__popcll(rowmask[tid]) : 0;
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < FT_H; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (tid < FT_H) rowstart[tid] = inc - cnt;
        if (tid == FT_H - 1) rowstart[FT_H] = inc;
        // the 16 line counts (<= 64 each) as bytes: two words, written through to where the row's last workgroup reads them
        u64 packed = (u64)(tid < FT_H ? cnt : 0) << (8 * (tid & 7));
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) packed |= (u64)__shfl_xor((long long)packed, o);
        if (order_rows && (tid == 0 || tid == 8))
            __hip_atomic_store(dline + 2 * (size_t)blockIdx.x + (tid >> 3), packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!order_rows && tid == FT_H - 1) row_count[blockIdx.x] = inc;
    }
    __syncthreads();
    const int total = rowstart[FT_H];
    u64* out = order_rows ? dslots + (size_t)blockIdx.x * FT_TILE_CAP * 2 : reinterpret_cast<u64*>(rows + (size_t)blockIdx.x * FT_TILE_CAP);
    const int ndisc = c_disc_n;
    // this lane's pixels of the intensity-centroid disc (k = lane, lane + 64, ...): offsets and (u, v) once per workgroup,
    // not once per survivor
    constexpr int DISC_IT = (768 + 63) / 64;
    // (u + 15, v + 15) of this lane's disc pixels, packed: all the survivor loop keeps of the table.  Offsets are taken from the
    // corner (gx - 15, gy - 15) of the patch, so that they are non-negative 32-bit lane offsets on one scalar base.
    unsigned duv[DISC_IT];
#pragma unroll
    for (int q = 0; q < DISC_IT; ++q) {
        const int k = lane + 64 * q;
        const int u = k < ndisc ? c_disc[2 * k] : 0, v = k < ndisc ? c_disc[2 * k + 1] : 0;
        duv[q] = (unsigned)(u + 15) | ((unsigned)(v + 15) << 8);  // (a padded lane: u = v = 0 -- it re-reads the centre with weight 0)
    }
    const unsigned hb = lane < 49 ? (unsigned)(lane / 7 + FT_HY - 3) * FT_PW + (unsigned)(lane % 7 + FT_HX - 3) : 0u;  // this lane's Harris block position
    for (int i = wave; i < total; i += 4) {
        // survivor i -> (row, i-th set bit of the row's mask)
        int r = 0;
#pragma unroll
        for (int k = 1; k < FT_H; ++k)
            if (i >= rowstart[k]) r = k;
        r = __builtin_amdgcn_readfirstlane(r);
        const u64 m = rowmask[r];
        const int kth = i - rowstart[r];
        const bool me = ((m >> lane) & 1) && (int)__popcll(m & ((1ull << lane) - 1)) == kth;
        // (one survivor per wave: its position is wave-uniform)
        const int lx = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(me)) - 1);
        const int gx = x0 + lx, gy = y0 + r;
        const uint8_t* lo = pb + r * FT_PW + lx;  // corner (gx - FT_HX, gy - FT_HY) of the survivor's patch in the staged bytes
        // HarrisResponses, blockSize 7: lanes 0..48 take one block position each
        int a = 0, b = 0, c = 0;
        if (lane < 49) {
            const uint8_t* q = lo + hb;
            const int p_l = q[-1], p_r = q[1], p_u = q[-FT_PW], p_d = q[FT_PW];
            const int p_ul = q[-FT_PW - 1], p_ur = q[-FT_PW + 1], p_dl = q[FT_PW - 1], p_dr = q[FT_PW + 1];
            int Ix = (p_r - p_l) * 2 + (p_ur - p_ul) + (p_dr - p_dl);
            int Iy = (p_d - p_u) * 2 + (p_dl - p_ul) + (p_dr - p_ur);
            a = Ix * Ix;
            b = Iy * Iy;
            c = Ix * Iy;
        }
        a = wave_sum(a);
        b = wave_sum(b);
        c = wave_sum(c);
        // IC_Angle: m10 = sum u*I, m01 = sum v*I over the disc
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int q = 0; q < DISC_IT; ++q) {
            unsigned e = duv[q];
            MVO_OPAQUE(e);  // (unpacked per survivor: hoisted, the twelve entries become three dozen live registers)
            const int du = (int)(e & 0xffu) - 15, dv = (int)((e >> 8) & 0xffu) - 15;
            const int val = lo[(dv + FT_HY) * FT_PW + (du + FT_HX)];
            m10 += du * val;
            m01 += dv * val;
        }
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
        if (lane == 0) {
            float scale = 1.f / ((1 << 2) * 7 * 255.f);
            float scale_sq_sq = scale * scale * scale * scale;
            float fa = (float)a, fb = (float)b, fc = (float)c;
            DevCandidate cd;
            cd.x = (int16_t)gx;
            cd.y = (int16_t)gy;
            cd.level_score = (lvl << 16) | sc[(r + 1) * 68 + (lx + 1)];
            cd.harris = (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
            cd.angle = fast_atan2_deg((float)m01, (float)m10);
            static_assert(sizeof(DevCandidate) == 16, "record = two 64-bit words");
            u64 w[2];
            __builtin_memcpy(w, &cd, 16);
            if (order_rows) {
                __hip_atomic_store(out + 2 * i, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out + 2 * i + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // ONE 16-byte store per record: the slot is host memory, every store is a PCIe write of its own
                reinterpret_cast<ulonglong2*>(out)[i] = make_ulonglong2(w[0], w[1]);
            }
        }
    }
    if (!order_rows) return;
    // ---- 4. the tile row's last workgroup orders the row
    // (dynamic LDS, sized by the host for the widest level: orb_detect_order_lds)
    MVO_DYN_LDS(int, s_dyn);
    const int mtx = P.lv[0].tiles_x;                     // level 0 is the widest
    __shared__ int s_last;
    int* s_off = s_dyn;                                  // [FT_H * mtx] exclusive scan over (line, tile)
    int* s_tpre = s_off + FT_H * mtx;                    // [mtx + 1] records before tile tx in (tile, slot) enumeration
    typedef uint16_t TileStarts[FT_H + 1];
    TileStarts* s_tstart = reinterpret_cast<TileStarts*>(s_tpre + mtx + 1);  // [mtx][17] line starts inside every tile
    __shared__ int s_wsum[4];
    MVO_WAIT_VM0();
    __syncthreads();
    const int t0 = L.tile_off + ty * L.tiles_x;  // first tile of the row: names the row
    if (tid == 0) {
        const int last = __hip_atomic_fetch_add(arrive + t0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.tiles_x - 1;
        if (last) __hip_atomic_store(arrive + t0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int ntx = L.tiles_x;
    for (int q = tid; q < ntx; q += 256) {
        const u64 a = __hip_atomic_load(dline + 2 * (size_t)(t0 + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 b = __hip_atomic_load(dline + 2 * (size_t)(t0 + q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int acc = 0;
#pragma unroll
        for (int y = 0; y < FT_H; ++y) {
            s_tstart[q][y] = (uint16_t)acc;
            acc += (int)(((y < 8 ? a : b) >> (8 * (y & 7))) & 0xff);
        }
        s_tstart[q][FT_H] = (uint16_t)acc;
    }
    __syncthreads();
    {   // exclusive scan of cnt(line y, tile q) in (y, q) order and of the tile totals: 8 consecutive entries per thread
        const int nent = FT_H * ntx;
        int v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = 8 * tid + k;
            const int y = e / ntx, q = e - y * ntx;
            v[k] = e < nent ? (int)s_tstart[q][y + 1] - (int)s_tstart[q][y] : 0;
            sum += v[k];
        }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int w2 = 0; w2 < wave; ++w2) run += s_wsum[w2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = 8 * tid + k;
            if (e < nent) s_off[e] = run;
            run += v[k];
        }
        if (tid == 0) {
            int acc = 0;
            for (int q = 0; q < ntx; ++q) {
                s_tpre[q] = acc;
                acc += s_tstart[q][FT_H];
            }
            s_tpre[ntx] = acc;
            row_count[t0] = acc;
        }
    }
    __syncthreads();
    const int row_total = s_tpre[ntx];
    u64* dst = reinterpret_cast<u64*>(rows + (size_t)t0 * FT_TILE_CAP);
    for (int k = tid; k < row_total; k += 256) {
        int lo = 0, hi = ntx;  // tile q with s_tpre[q] <= k < s_tpre[q + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_tpre[mid] <= k) lo = mid; else hi = mid;
        }
        const int q = lo, c = k - s_tpre[q];
        int y = 0;
#pragma unroll
        for (int yy = 1; yy < FT_H; ++yy)
            if (c >= (int)s_tstart[q][yy]) y = yy;
        const int d = s_off[y * ntx + q] + c - (int)s_tstart[q][y];
        const u64* src = dslots + ((size_t)(t0 + q) * FT_TILE_CAP + c) * 2;
        const u64 w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<ulonglong2*>(dst)[d] = make_ulonglong2(w0, w1);  // (one PCIe write per record)
    }
}

// ------------------------------------------------------------------------------------------------ blur
// cv::GaussianBlur(7x7, sigma 2) canonical fixed point {18,34,48,56,48,34,18}/256, 8.8 intermediate,
// (+2^15)>>16 final; the 32-px frame is copied unblurred (cv::ORB::compute blurs the level ROI in place).
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ raw, uint8_t* __restrict__ blur,
                                              PyrInfo P, int nlevels_active) {
    __shared__ uint32_t pix[22 * 18];
    __shared__ __attribute__((aligned(8))) uint16_t hb[22 * 64];
    const int tid = threadIdx.x;
    const int lvl = find_level_by(P, blockIdx.x, 2);
    if (lvl >= nlevels_active) return;
    const LevelInfo L = P.lv[lvl];
    const int t = blockIdx.x - L.btile_off;
    const int tx = t % L.btiles_x, ty = t / L.btiles_x;
    const int bx0 = tx * 64, by0 = ty * 16;  // bordered coordinates
    const int rows = L.h + 2 * MVO_BORDER, dwords = L.stride / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(raw + L.off);
    for (int i = tid; i < 22 * 18; i += 256) {
        int r = i / 18, c = i - r * 18;
        int yy = min(max(by0 - 3 + r, 0), rows - 1);
        int xx = min(max(bx0 / 4 - 1 + c, 0), dwords - 1);
        pix[i] = src[(size_t)yy * dwords + xx];
    }
    __syncthreads();
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(pix);
    constexpr int G[7] = {18, 34, 48, 56, 48, 34, 18};
#ifndef MVO_KERNEL_SIM
    // Horizontal pass: one thread = four consecutive outputs of a row from THREE dword reads (the 10 bytes they share) -- every
    // 7-tap sum is two v_dot4_u32_u8 over a byte window shifted into place by v_alignbyte.  Vertical pass: four outputs from seven
    // 8-byte reads.  (One byte / one u16 per LDS instruction, this kernel's waves spent a third of their cycles queueing for the LDS.)
    uint32_t* hb32 = reinterpret_cast<uint32_t*>(hb);  // row pitch 32 dwords (64 u16)
    constexpr uint32_t GLO = 18u | 34u << 8 | 48u << 16 | 56u << 24, GHI = 48u | 34u << 8 | 18u << 16;
    for (int item = tid; item < 22 * 16; item += 256) {
        const int r = item >> 4, g = item & 15;
        const uint32_t w0 = pix[r * 18 + g], w1 = pix[r * 18 + g + 1], w2 = pix[r * 18 + g + 2];
        // output c = 4 g + j reads the row bytes 4 g + j + 1 .. 4 g + j + 7 (p[k - 3], p = row + c + 4)
        const uint32_t a0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), GHI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), GLO, 0u, false), false);
        const uint32_t a1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), GHI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), GLO, 0u, false), false);
        const uint32_t a2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), GHI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), GLO, 0u, false), false);
        const uint32_t a3 = __builtin_amdgcn_udot4(w2, GHI, __builtin_amdgcn_udot4(w1, GLO, 0u, false), false);
        hb32[r * 32 + 2 * g] = a0 | a1 << 16;  // (each sum <= 255 * 256 fits 16 bits)
        hb32[r * 32 + 2 * g + 1] = a2 | a3 << 16;
    }
    __syncthreads();
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const int by = by0 + r;
    if (by >= rows || bx0 + c4 >= L.stride) return;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const uint2 h = *reinterpret_cast<const uint2*>(hb32 + (r + j) * 32 + 2 * (tid & 15));
        acc0 += (uint32_t)G[j] * (h.x & 0xffffu);
        acc1 += (uint32_t)G[j] * (h.x >> 16);
        acc2 += (uint32_t)G[j] * (h.y & 0xffffu);
        acc3 += (uint32_t)G[j] * (h.y >> 16);
    }
    const uint32_t accs[4] = {acc0, acc1, acc2, acc3};
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 + k;
        const int gx = bx0 + c - MVO_BORDER, gy = by - MVO_BORDER;
        const uint32_t v = (gx >= 0 && gx < L.w && gy >= 0 && gy < L.h) ? (accs[k] + 32768u) >> 16 : (uint32_t)pb[(r + 3) * 72 + c + 4];
        out |= v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(blur + L.off + (size_t)by * L.stride + bx0 + c4) = out;
#else
    for (int i = tid; i < 22 * 64; i += 256) {
        int r = i >> 6, c = i & 63;
        const uint8_t* p = pb + r * 72 + c + 4;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += G[k] * p[k - 3];
        hb[i] = (uint16_t)acc;
    }
    __syncthreads();
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const int by = by0 + r;
    if (by >= rows || bx0 + c4 >= L.stride) return;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = c4 + k;
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += G[j] * hb[(r + j) * 64 + c];
        int gx = bx0 + c - MVO_BORDER, gy = by - MVO_BORDER;
        uint32_t v = (gx >= 0 && gx < L.w && gy >= 0 && gy < L.h) ? (uint32_t)((acc + 32768) >> 16)
                                                                  : (uint32_t)pb[(r + 3) * 72 + c + 4];
        out |= v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(blur + L.off + (size_t)by * L.stride + bx0 + c4) = out;
#endif
}

// ------------------------------------------------------------------------------------------------ rBRIEF
// cv::ORB::compute blurs every level (GaussianBlur 7x7, sigma 2, over the level ROI: the 32-px frame stays unblurred)
// and samples the blurred level around each keypoint.  Only the 39-row window around a keypoint is ever sampled, so the
// wave blurs exactly that window in LDS: raw rows cy-22 .. cy+22 are staged, the horizontal pass keeps 8.8 fixed point
// (the kernel {18,34,48,56,48,34,18} sums to 256), the vertical pass rounds (+2^15) >> 16 -- the same integers as the
// whole-level k_blur, without its launch and without the 2 x pyramid bytes of traffic.
#define BW_R 19
#define BW_ROWS (2 * BW_R + 1)
#define BW_STRIDE 48
#define BW_RAW_ROWS (BW_ROWS + 6)
#define BW_RAW_DW 14  // 56 bytes per staged row: 4 bytes left of the window, 4 right
struct __attribute__((aligned(16))) BriefLds {
    uint32_t raw[BW_RAW_ROWS * BW_RAW_DW];
    uint16_t hb[BW_RAW_ROWS * BW_STRIDE];
    uint32_t out[BW_ROWS * (BW_STRIDE / 4)];
};
// WAVES keypoints per workgroup (one wave each, 8.7 KB of LDS each).  WAVES = 1 (round 4): the waves of a workgroup share
// nothing, and small workgroups let a CU hold as many of them as its wave slots allow instead of four 35-KB ones -- next to the
// resident solver grid the extraction kernels live on 6 CUs per XCD, where occupancy is throughput.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_brief(const uint8_t* __restrict__ rawpyr, const DevDescKp* __restrict__ kps,
                                                      uint8_t* __restrict__ desc, uint8_t* __restrict__ desc_host, PyrInfo P,
                                                      int n) {
    __shared__ BriefLds lds[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ki = blockIdx.x * WAVES + wave;
    if (ki >= n) return;
    const DevDescKp kp = kps[ki];
    const LevelInfo L = P.lv[__builtin_amdgcn_readfirstlane(kp.level)];  // (wave-uniform: scalar loads)
    const int cxb = kp.cx + MVO_BORDER, cyb = kp.cy + MVO_BORDER;  // bordered coordinates
    const int x0 = (cxb - BW_R) & ~3, y0 = cyb - BW_R;             // window origin (bordered)
    const int rows = L.h + 2 * MVO_BORDER, dwords = L.stride / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rawpyr + L.off);
    BriefLds& W = lds[wave];
    // clamped reads: a clamped row / dword only ever feeds the taps of frame pixels, which are copied, not blurred
    for (int i = lane; i < BW_RAW_ROWS * BW_RAW_DW; i += 64) {
        const int r = i / BW_RAW_DW, c = i - r * BW_RAW_DW;
        const int yy = min(max(y0 - 3 + r, 0), rows - 1);
        const int xx = min(max(x0 / 4 - 1 + c, 0), dwords - 1);
        W.raw[i] = src[(size_t)yy * dwords + xx];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    constexpr int G[7] = {18, 34, 48, 56, 48, 34, 18};
    // horizontal pass: one item = 4 adjacent columns of one staged row; the 10 source bytes come as three dwords
    for (int i = lane; i < BW_RAW_ROWS * (BW_STRIDE / 4); i += 64) {
        const int r = i / (BW_STRIDE / 4), cq = i - r * (BW_STRIDE / 4);  // window columns 4 cq .. 4 cq + 3
        const uint32_t* q = W.raw + r * BW_RAW_DW + cq;  // staged bytes 4 cq .. 4 cq + 11 = window columns 4 cq - 4 .. 4 cq + 7
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        // px[k] = window column 4 cq - 3 + k = staged byte 4 cq + 1 + k
        const int px[10] = {(int)((d0 >> 8) & 255), (int)((d0 >> 16) & 255), (int)(d0 >> 24), (int)(d1 & 255), (int)((d1 >> 8) & 255),
                            (int)((d1 >> 16) & 255), (int)(d1 >> 24), (int)(d2 & 255), (int)((d2 >> 8) & 255), (int)((d2 >> 16) & 255)};
        uint32_t h01, h23;
        {
            int acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[k] = 0;
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[k] += G[j] * px[k + j];
            }
            h01 = (uint32_t)acc[0] | ((uint32_t)acc[1] << 16);
            h23 = (uint32_t)acc[2] | ((uint32_t)acc[3] << 16);
        }
        uint32_t* hq = reinterpret_cast<uint32_t*>(W.hb + r * BW_STRIDE + 4 * cq);
        hq[0] = h01;
        hq[1] = h23;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // vertical pass + the unblurred frame; four columns of a row = one 8-byte read per tap row
    for (int i = lane; i < BW_ROWS * (BW_STRIDE / 4); i += 64) {
        const int r = i / (BW_STRIDE / 4), cq = i - r * (BW_STRIDE / 4), c4 = 4 * cq;
        const int gy = y0 + r - MVO_BORDER;
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint2 t = *reinterpret_cast<const uint2*>(W.hb + (r + j) * BW_STRIDE + c4);
            acc[0] += G[j] * (int)(t.x & 0xffff);
            acc[1] += G[j] * (int)(t.x >> 16);
            acc[2] += G[j] * (int)(t.y & 0xffff);
            acc[3] += G[j] * (int)(t.y >> 16);
        }
        const uint32_t rawq = W.raw[(r + 3) * BW_RAW_DW + cq + 1];  // the same four pixels, unblurred
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gx = x0 + c4 + k - MVO_BORDER;
            const uint32_t v = (gx >= 0 && gx < L.w && gy >= 0 && gy < L.h) ? (uint32_t)((acc[k] + 32768) >> 16)
                                                                            : (rawq >> (8 * k)) & 255u;
            o |= v << (8 * k);
        }
        W.out[i] = o;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(W.out) + BW_R * BW_STRIDE + (cxb - x0);
    const float a = kp.a, b = kp.b;
    u64 words[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const signed char* pt = c_pattern + 4 * (lane + 64 * k);
        float px0 = (float)pt[0], py0 = (float)pt[1], px1 = (float)pt[2], py1 = (float)pt[3];
        int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(px0, a), __fmul_rn(py0, b)));
        int iy0 = __float2int_rn(__fadd_rn(__fmul_rn(px0, b), __fmul_rn(py0, a)));
        int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(px1, a), __fmul_rn(py1, b)));
        int iy1 = __float2int_rn(__fadd_rn(__fmul_rn(px1, b), __fmul_rn(py1, a)));
        int t0 = wb[iy0 * BW_STRIDE + ix0];
        int t1 = wb[iy1 * BW_STRIDE + ix1];
        words[k] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        reinterpret_cast<u64*>(desc + (size_t)ki * 32)[lane] = words[lane];
        if (desc_host) reinterpret_cast<u64*>(desc_host + (size_t)ki * 32)[lane] = words[lane];  // pinned mirror: no copy dispatch
    }
}

// The same descriptor from a level that has been blurred as a whole (k_blur): one WAVE per keypoint, 8 byte loads per lane.
// A ctx that shares the GPU with many others (throughput mode) takes this form: blurring every level once costs a sixth of
// the instructions of blurring 2000 keypoint windows (0.8 M against ~5 M wave instructions per S640 frame), and under the
// resident solver grid the extraction lives on 6 CUs per XCD where instructions, not launches, are what a frame costs.  The
// blur is queued right behind the detection kernel: it runs while the host thread selects the keypoints.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_brief_sample(const uint8_t* __restrict__ blurpyr, const DevDescKp* __restrict__ kps,
                                                             uint8_t* __restrict__ desc, uint8_t* __restrict__ desc_host, PyrInfo P,
                                                             int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ki = blockIdx.x * WAVES + wave;
    if (ki >= n) return;
    const DevDescKp kp = kps[ki];
    const LevelInfo L = P.lv[__builtin_amdgcn_readfirstlane(kp.level)];
    const int stride = L.stride;
    const uint8_t* wb = blurpyr + L.off + (size_t)(kp.cy + MVO_BORDER) * stride + (kp.cx + MVO_BORDER);
    const float a = kp.a, b = kp.b;
    u64 words[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const signed char* pt = c_pattern + 4 * (lane + 64 * k);
        float px0 = (float)pt[0], py0 = (float)pt[1], px1 = (float)pt[2], py1 = (float)pt[3];
        int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(px0, a), __fmul_rn(py0, b)));
        int iy0 = __float2int_rn(__fadd_rn(__fmul_rn(px0, b), __fmul_rn(py0, a)));
        int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(px1, a), __fmul_rn(py1, b)));
        int iy1 = __float2int_rn(__fadd_rn(__fmul_rn(px1, b), __fmul_rn(py1, a)));
        int t0 = wb[iy0 * stride + ix0];
        int t1 = wb[iy1 * stride + ix1];
        words[k] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        reinterpret_cast<u64*>(desc + (size_t)ki * 32)[lane] = words[lane];
        if (desc_host) reinterpret_cast<u64*>(desc_host + (size_t)ki * 32)[lane] = words[lane];
    }
}

// ================================================================================================ launchers
static bool g_tables_ready[16] = {false};
int g_pyr_force_chain = 0;  // test hook: the per-pixel chain kernel instead of the LDS-tiled one
int g_pyr_full_pool = std::getenv("MVO_PYR_FULL_POOL") ? std::atoi(std::getenv("MVO_PYR_FULL_POOL")) : 0;  // A/B: the fixed 32-KB pool
int g_brief_level_blur = std::getenv("MVO_BRIEF_LEVEL_BLUR") ? std::atoi(std::getenv("MVO_BRIEF_LEVEL_BLUR")) : -1;  // -1 = by mode (throughput ctx: whole-level blur + k_brief_sample), 0 = always the fused k_brief, 1 = always the level form (A/B)
int g_brief_waves = std::getenv("MVO_BRIEF_WAVES") ? std::atoi(std::getenv("MVO_BRIEF_WAVES")) : 4;        // A/B: 1 = one keypoint per workgroup (measured: no gain under load, 3 us slower alone)

static int upload_constant_tables(mvo_ctx* ctx) {
    if (ctx->device < 16 && g_tables_ready[ctx->device]) return MVO_OK;
    MVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), MVO_ORB_PATTERN_31, sizeof(MVO_ORB_PATTERN_31)));
    // umax table of cv::ORB (halfPatchSize 15) -> explicit list of disc pixels
    int umax[17];
    const int hp = 15;
    int vmax = (int)std::floor(hp * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(hp * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = (int)std::lrint(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
    signed char disc[768 * 2] = {0};
    int n = 0;
    for (int v = -hp; v <= hp; ++v) {
        int d = umax[v < 0 ? -v : v];
        for (int u = -d; u <= d; ++u) {
            disc[2 * n] = (signed char)u;
            disc[2 * n + 1] = (signed char)v;
            ++n;
        }
    }
    MVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_disc), disc, sizeof(disc)));
    MVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_disc_n), &n, sizeof(n)));
    if (ctx->device < 16) g_tables_ready[ctx->device] = true;
    return MVO_OK;
}

int orb_launch_pyramid(mvo_ctx* ctx, const uint8_t* d_img, int stride, int channels, int nlevels) {
    int r = upload_constant_tables(ctx);
    if (r) return r;
    const PyrInfo& P = ctx->pyr;
    const int exact = ctx->orb.pyramid_interpolation != 0 ? 1 : 0;
    // groups of up to PYR_GROUP levels per launch: the first group hangs off the image, the next ones off the last level
    // of the group before (the reference's level_pyramid 4 is ONE launch)
    for (int l0 = 0; l0 < nlevels; l0 += PYR_GROUP) {
        const int nl = std::min(PYR_GROUP, nlevels - l0);
        PyrBase B{l0 == 0 ? d_img : nullptr, stride, channels, l0 == 0 ? 0 : l0 - 1};
        const int nblk = (l0 + nl < P.nlevels ? P.lv[l0 + nl].btile_off : P.n_btiles) - P.lv[l0].btile_off;
        const bool tiled = ctx->pyr_group_tiled[l0 / PYR_GROUP] && !g_pyr_force_chain;
        ProfScope ps(ctx, tiled ? "k_pyramid" : "k_pyramid_chain");
        if (tiled)
            hipLaunchKernelGGL(k_pyramid, dim3(nblk), dim3(256), g_pyr_full_pool ? PYR_LDS_BYTES : ctx->pyr_group_lds[l0 / PYR_GROUP],
                               ctx->stream, B, ctx->d_raw, P, ctx->d_tabs, l0, nl, exact, ctx->d_pyr_regs);
        else
            hipLaunchKernelGGL(k_pyramid_chain, dim3(nblk), dim3(256), 0, ctx->stream, B, ctx->d_raw, P, ctx->d_tabs, l0, nl, exact);
    }
    MVO_HIP(hipGetLastError());
    ctx->blur_valid = false;
    return MVO_OK;
}

// detection: FAST + NMS + Harris + angle in one launch; per-tile survivor counts and records land in the pinned buffer
// `host` = [int32 count x n_tiles (padded to 64 B)][DevCandidate x n_tiles x FT_TILE_CAP]
int orb_launch_detect(mvo_ctx* ctx, uint8_t* host, bool ordered) {
    const PyrInfo& P = ctx->pyr;
    int32_t* counts = reinterpret_cast<int32_t*>(host);
    DevCandidate* slots = reinterpret_cast<DevCandidate*>(host + orb_detect_counts_bytes(P.n_tiles));
    ProfScope ps(ctx, "k_fast_harris");
    // (level 0 is the widest level: its tile count per row sizes the ordering step's tables)
    const size_t lds = ordered ? ((size_t)FT_H * P.lv[0].tiles_x + P.lv[0].tiles_x + 1) * 4 + (size_t)P.lv[0].tiles_x * (FT_H + 1) * 2 : 0;
    hipLaunchKernelGGL(k_fast_harris, dim3(P.n_tiles), dim3(256), lds, ctx->stream, ctx->d_raw, P, ctx->orb.fast_threshold,
                       counts, slots, (u64*)ctx->d_fh_slots, (u64*)ctx->d_fh_line, ctx->d_fh_arrive, ordered ? 1 : 0);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

// does this ctx describe from whole blurred levels (k_blur + k_brief_sample) or from keypoint windows (k_brief)?
bool orb_brief_from_levels(const mvo_ctx* ctx) {
    return g_brief_level_blur < 0 ? ctx->ba_throughput_mode != 0 : g_brief_level_blur != 0;
}
// whole-level blur into d_blur: for k_brief_sample and for mvo_debug_get_level(blurred)
int orb_launch_blur(mvo_ctx* ctx, int nlevels) {
    const PyrInfo& P = ctx->pyr;
    int nb = nlevels < P.nlevels ? P.lv[nlevels].btile_off : P.n_btiles;
    ProfScope ps(ctx, "k_blur");
    hipLaunchKernelGGL(k_blur, dim3(nb), dim3(256), 0, ctx->stream, ctx->d_raw, ctx->d_blur, P, nlevels);
    MVO_HIP(hipGetLastError());
    ctx->blur_valid = true;
    return MVO_OK;
}

int orb_launch_brief(mvo_ctx* ctx, int n, const DevDescKp* kps, uint8_t* desc_host) {
    if (n <= 0) return MVO_OK;
    if (orb_brief_from_levels(ctx)) {
        if (!ctx->blur_valid) {  // (normally queued behind the detection already: orb_detect_device)
            int r = orb_launch_blur(ctx, ctx->pyr_levels_built);
            if (r) return r;
        }
        ProfScope ps(ctx, "k_brief");
        hipLaunchKernelGGL((k_brief_sample<4>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, ctx->d_blur, kps, ctx->d_desc, desc_host, ctx->pyr, n);
        MVO_HIP(hipGetLastError());
        return MVO_OK;
    }
    ProfScope ps(ctx, "k_brief");
    if (g_brief_waves == 4)
        hipLaunchKernelGGL((k_brief<4>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, ctx->d_raw, kps, ctx->d_desc, desc_host, ctx->pyr, n);
    else
        hipLaunchKernelGGL((k_brief<1>), dim3(n), dim3(64), 0, ctx->stream, ctx->d_raw, kps, ctx->d_desc, desc_host, ctx->pyr, n);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}
