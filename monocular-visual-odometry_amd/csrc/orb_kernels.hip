// csrc/orb_kernels.hip -- gfx950 kernels for ORB extraction (replaces cv::ORB::detect / ::compute as called
// from the reference's src/geometry/feature_match.cpp:22-23,34,45,48).
//
// Data layout in HBM: ONE byte buffer per pyramid flavour (raw gray, blurred), every level stored with its
// 32-px BORDER_REFLECT_101 frame, row stride rounded up to 64 B so that every tile row starts dword-aligned and
// a 64-lane wave reads one aligned 64-B segment per row.  All kernels are integer/byte work bounded by HBM/L2
// traffic; each is a single streaming pass with the neighbourhood staged in LDS:
//   k_gray_border   BGR -> gray + frame of level 0                      (1 B in x c, 1 B out per pixel)
//   k_resize_border level l from level l-1 (fixed-point bilinear) + frame
//   k_fast_nms      FAST-9/16 score + 3x3 NMS + 31-px border; one 64x16 tile per workgroup; emits one 64-bit
//                   survivor mask per (row, 64-px column) cell via wave ballot
//   k_scan_cells    exclusive scan of the cell counts (one workgroup) -> per-cell offsets + per-level starts
//   k_emit_cells    one thread per cell expands its mask into the canonical (level,row,col) candidate list
//   k_harris_angle  one WAVE per candidate: 7x7 Harris (49 lanes) + IC angle over the 749-px disc, wave
//                   shuffle reductions
//   k_blur          separable 7x7 fixed-point Gaussian through LDS, frame copied unblurred
//   k_brief         one WAVE per keypoint: 39x48 window staged in LDS, 512 rotated taps, 4 ballots = 256 bits
// Compiled with -ffp-contract=off: the float expressions (Harris response, fastAtan2, tap rotation) are
// canonical arithmetic and must round exactly like the oracle.
#include "mvo_internal.h"
#include "orb_pattern_31.h"

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

// ------------------------------------------------------------------------------------------------ gray
// cv::cvtColor(BGR2GRAY) 8-bit: (1868 B + 9617 G + 4899 R + 2^13) >> 14.  One thread = 4 bordered pixels.
__global__ __launch_bounds__(256) void k_gray_border(const uint8_t* __restrict__ img, int istride, int ch,
                                                     uint8_t* __restrict__ raw, LevelInfo L) {
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int by = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= L.stride || by >= L.h + 2 * MVO_BORDER) return;
    const int sy = reflect101(by - MVO_BORDER, L.h);
    const uint8_t* row = img + (size_t)sy * istride;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int bx = x4 + k;
        uint32_t g = 0;
        if (bx < L.w + 2 * MVO_BORDER) {
            int sx = reflect101(bx - MVO_BORDER, L.w);
            const uint8_t* px = row + sx * ch;
            g = ch == 1 ? px[0] : (uint32_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
        }
        out |= g << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(raw + L.off + (size_t)by * L.stride + x4) = out;
}

// ------------------------------------------------------------------------------------------------ resize
// cv::resize 8-bit fixed point; coefficient tables are built on the host once per geometry.  exact: INTER_LINEAR_EXACT
// (8-bit coefficients, one rounding), else INTER_LINEAR (11-bit coefficients, OpenCV's truncating vertical pass).
__global__ __launch_bounds__(256) void k_resize_border(uint8_t* __restrict__ raw, LevelInfo S, LevelInfo D,
                                                       const ResizeEntry* __restrict__ tabs, int exact) {
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int by = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= D.stride || by >= D.h + 2 * MVO_BORDER) return;
    const int dy = reflect101(by - MVO_BORDER, D.h);
    const ResizeEntry ty = tabs[D.tab_off + D.w + dy];
    const int sy1 = min(ty.ofs + 1, S.h - 1);
    const uint8_t* r0 = raw + S.off + (size_t)(ty.ofs + MVO_BORDER) * S.stride + MVO_BORDER;
    const uint8_t* r1 = raw + S.off + (size_t)(sy1 + MVO_BORDER) * S.stride + MVO_BORDER;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int bx = x4 + k;
        uint32_t g = 0;
        if (bx < D.w + 2 * MVO_BORDER) {
            int dx = reflect101(bx - MVO_BORDER, D.w);
            const ResizeEntry tx = tabs[D.tab_off + dx];
            int sx1 = min(tx.ofs + 1, S.w - 1);
            int h0 = r0[tx.ofs] * tx.c0 + r0[sx1] * tx.c1;
            int h1 = r1[tx.ofs] * tx.c0 + r1[sx1] * tx.c1;
            g = exact ? (uint32_t)((ty.c0 * h0 + ty.c1 * h1 + 32768) >> 16) & 0xff
                      : (uint32_t)((((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
        }
        out |= g << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(raw + D.off + (size_t)by * D.stride + x4) = out;
}

// ------------------------------------------------------------------------------------------------ FAST + NMS
#define FT_W 64
#define FT_H 16
#define FT_PW 72  // tile + 4-px halo each side, bytes per LDS row (18 dwords)
#define FT_PH (FT_H + 8)
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

// cornerScore<16>: the largest threshold for which the pixel is still a FAST-9 corner; 0 if it is not one at
// threshold thr.  d[k] = centre - circle[k].
__device__ __forceinline__ int fast_score16(const int (&d)[16], int thr) {
    // quick reject: bit masks of circle pixels darker / brighter than the centre by more than thr
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
    if (!has9(dark) && !has9(bright)) return 0;
    // sliding minimum / maximum over 9 consecutive circle pixels by doubling
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

__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t* __restrict__ raw, uint8_t* __restrict__ score,
                                                  u64* __restrict__ cell_mask, int32_t* __restrict__ cell_cnt,
                                                  PyrInfo P, int thr) {
    __shared__ uint32_t pix[FT_PH * (FT_PW / 4)];
    __shared__ uint8_t sc[FT_SH * 68];
    const int tid = threadIdx.x;
    const int lvl = find_level_by(P, blockIdx.x, 0);
    const LevelInfo L = P.lv[lvl];
    const int t = blockIdx.x - L.tile_off;
    const int tx = t % L.tiles_x, ty = t / L.tiles_x;
    const int x0 = tx * FT_W, y0 = ty * FT_H;
    const uint8_t* base = raw + L.off;
    // stage (FT_H+8) x 72 bytes; row/col origin = (y0-4, x0-4) in interior coordinates
    for (int i = tid; i < FT_PH * (FT_PW / 4); i += 256) {
        int r = i / (FT_PW / 4), c = i - r * (FT_PW / 4);
        const uint8_t* p = base + (size_t)(y0 - 4 + r + MVO_BORDER) * L.stride + (x0 - 4 + MVO_BORDER) + 4 * c;
        pix[i] = *reinterpret_cast<const uint32_t*>(p);
    }
    __syncthreads();
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(pix);
    // FAST circle (x, y), same enumeration as the oracle
    constexpr int CX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int CY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    for (int i = tid; i < FT_SH * FT_SW; i += 256) {
        int sr = i / FT_SW, scx = i - sr * FT_SW;
        const uint8_t* c = pb + (sr + 3) * FT_PW + (scx + 3);  // (ly, lx) = (sr-1, scx-1); +4 halo
        int v = c[0];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - (int)c[CX[k] + CY[k] * FT_PW];
        sc[sr * 68 + scx] = (uint8_t)fast_score16(d, thr);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int ly = wave; ly < FT_H; ly += 4) {
        const int gx = x0 + lane, gy = y0 + ly;
        const uint8_t* s = sc + (ly + 1) * 68 + (lane + 1);
        int v = s[0];
        bool flag = v > 0 && v > s[-1] && v > s[1] && v > s[-69] && v > s[-68] && v > s[-67] && v > s[67] &&
                    v > s[68] && v > s[69];
        flag = flag && gx >= 31 && gx < L.w - 31 && gy >= 31 && gy < L.h - 31;
        u64 mask = __ballot(flag);
        if (flag) score[L.off + (size_t)(gy + MVO_BORDER) * L.stride + MVO_BORDER + gx] = (uint8_t)v;
        if (lane == 0 && gy < L.h) {
            int cell = L.cell_off + gy * L.tiles_x + tx;
            cell_mask[cell] = mask;
            cell_cnt[cell] = __popcll(mask);
        }
    }
}

// ------------------------------------------------------------------------------------------------ scan + emit
// k_scan_cells: ONE workgroup turns the per-cell survivor counts into exclusive offsets (canonical order = level,
// row, 64-px column) and fills the header; k_emit_cells: one thread per cell expands its ballot mask.
__global__ __launch_bounds__(1024) void k_scan_cells(const int32_t* __restrict__ cell_cnt, int32_t* __restrict__ cell_off,
                                                     CandHeader* __restrict__ hdr, PyrInfo P) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n_cells;
    // every thread owns CH consecutive cells, held in registers (int4 loads / stores); CH*1024 >= n is guaranteed
    // by the launcher (it picks the instantiation)
    constexpr int CH = 16;
    const int passes = (n + CH * 1024 - 1) / (CH * 1024);
    int carry = 0;
    for (int ps = 0; ps < passes; ++ps) {
        const int c0 = (ps * 1024 + tid) * CH;
        int cnt[CH];
#pragma unroll
        for (int k = 0; k < CH; k += 4) {
            int4 v = make_int4(0, 0, 0, 0);
            if (c0 + k + 4 <= n) {
                v = *reinterpret_cast<const int4*>(cell_cnt + c0 + k);
            } else {
                if (c0 + k < n) v.x = cell_cnt[c0 + k];
                if (c0 + k + 1 < n) v.y = cell_cnt[c0 + k + 1];
                if (c0 + k + 2 < n) v.z = cell_cnt[c0 + k + 2];
            }
            cnt[k] = v.x;
            cnt[k + 1] = v.y;
            cnt[k + 2] = v.z;
            cnt[k + 3] = v.w;
        }
        int local = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) local += cnt[k];
        int incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            int sv = wsum[w];
            if (w < wave) wbase += sv;
            total += sv;
        }
        int off = carry + wbase + incl - local;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = c0 + k;
            if (c < n) {
#pragma unroll
                for (int l = 0; l < MVO_MAX_LEVELS; ++l)
                    if (l < P.nlevels && c == P.lv[l].cell_off) hdr->level_start[l] = off;
                cell_off[c] = off;
            }
            off += cnt[k];
        }
        carry += total;
    }
    if (tid == 0) {
        hdr->n_total = carry;
        hdr->level_start[P.nlevels] = carry;
    }
}

__global__ __launch_bounds__(256) void k_emit_cells(const u64* __restrict__ cell_mask, const int32_t* __restrict__ cell_off,
                                                    const uint8_t* __restrict__ score, DevCandidate* __restrict__ cand,
                                                    PyrInfo P, int cand_cap) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= P.n_cells) return;
    u64 m = cell_mask[c];
    if (!m) return;
    int off = cell_off[c];
    const int lvl = find_level_by(P, c, 1);
    const LevelInfo L = P.lv[lvl];
    const int rc = c - L.cell_off;
    const int y = rc / L.tiles_x, tx = rc - y * L.tiles_x;
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int x = tx * FT_W + b;
        if (off < cand_cap) {
            DevCandidate cd;
            cd.x = (int16_t)x;
            cd.y = (int16_t)y;
            cd.level_score = (lvl << 16) | score[L.off + (size_t)(y + MVO_BORDER) * L.stride + MVO_BORDER + x];
            cd.harris = 0.f;
            cd.angle = 0.f;
            cand[off] = cd;
        }
        ++off;
    }
}

// ------------------------------------------------------------------------------------------------ Harris + angle
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
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

__global__ __launch_bounds__(256) void k_harris_angle(const uint8_t* __restrict__ raw,
                                                      DevCandidate* __restrict__ cand,
                                                      const CandHeader* __restrict__ hdr, PyrInfo P, int cand_cap) {
    const int lane = threadIdx.x & 63;
    const int wave_id = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * 4;
    const int n = min(hdr->n_total, cand_cap);
    const int ndisc = c_disc_n;
    for (int ci = wave_id; ci < n; ci += nwaves) {
        const DevCandidate cd = cand[ci];
        const LevelInfo L = P.lv[cd.level_score >> 16];
        const int step = L.stride;
        const uint8_t* ctr = raw + L.off + (size_t)(cd.y + MVO_BORDER) * step + MVO_BORDER + cd.x;
        // HarrisResponses, blockSize 7: lanes 0..48 take one block position each
        int a = 0, b = 0, c = 0;
        if (lane < 49) {
            int i = lane / 7 - 3, j = lane % 7 - 3;
            const uint8_t* p = ctr + i * step + j;
            int Ix = ((int)p[1] - p[-1]) * 2 + ((int)p[-step + 1] - p[-step - 1]) + ((int)p[step + 1] - p[step - 1]);
            int Iy = ((int)p[step] - p[-step]) * 2 + ((int)p[step - 1] - p[-step - 1]) + ((int)p[step + 1] - p[-step + 1]);
            a = Ix * Ix;
            b = Iy * Iy;
            c = Ix * Iy;
        }
        a = wave_sum(a);
        b = wave_sum(b);
        c = wave_sum(c);
        // IC_Angle: m10 = sum u*I, m01 = sum v*I over the disc
        int m10 = 0, m01 = 0;
        for (int k = lane; k < ndisc; k += 64) {
            int u = c_disc[2 * k], v = c_disc[2 * k + 1];
            int val = ctr[v * step + u];
            m10 += u * val;
            m01 += v * val;
        }
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
        if (lane == 0) {
            float scale = 1.f / ((1 << 2) * 7 * 255.f);
            float scale_sq_sq = scale * scale * scale * scale;
            float fa = (float)a, fb = (float)b, fc = (float)c;
            cand[ci].harris = (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
            cand[ci].angle = fast_atan2_deg((float)m01, (float)m10);
        }
    }
}

// ------------------------------------------------------------------------------------------------ blur
// cv::GaussianBlur(7x7, sigma 2) canonical fixed point {18,34,48,56,48,34,18}/256, 8.8 intermediate,
// (+2^15)>>16 final; the 32-px frame is copied unblurred (cv::ORB::compute blurs the level ROI in place).
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ raw, uint8_t* __restrict__ blur,
                                              PyrInfo P, int nlevels_active) {
    __shared__ uint32_t pix[22 * 18];
    __shared__ uint16_t hb[22 * 64];
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
}

// ------------------------------------------------------------------------------------------------ rBRIEF
#define BW_R 19
#define BW_ROWS (2 * BW_R + 1)
#define BW_STRIDE 48
__global__ __launch_bounds__(256) void k_brief(const uint8_t* __restrict__ blur, const DevDescKp* __restrict__ kps,
                                               uint8_t* __restrict__ desc, PyrInfo P, int n) {
    __shared__ uint32_t win[4][BW_ROWS * (BW_STRIDE / 4)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ki = blockIdx.x * 4 + wave;
    if (ki >= n) return;
    const DevDescKp kp = kps[ki];
    const LevelInfo L = P.lv[kp.level];
    const int cxb = kp.cx + MVO_BORDER, cyb = kp.cy + MVO_BORDER;  // bordered coordinates
    const int x0 = (cxb - BW_R) & ~3;
    const uint8_t* src = blur + L.off + (size_t)(cyb - BW_R) * L.stride + x0;
    uint32_t* w = win[wave];
    for (int i = lane; i < BW_ROWS * (BW_STRIDE / 4); i += 64) {
        int r = i / (BW_STRIDE / 4), c = i - r * (BW_STRIDE / 4);
        w[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)r * L.stride + 4 * c);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(w) + BW_R * BW_STRIDE + (cxb - x0);
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
    if (lane < 4) reinterpret_cast<u64*>(desc + (size_t)ki * 32)[lane] = words[lane];
}

// ================================================================================================ launchers
static bool g_tables_ready[16] = {false};

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
    {
        const LevelInfo& L = P.lv[0];
        dim3 blk(64, 4), grd((L.stride / 4 + 63) / 64, (L.h + 2 * MVO_BORDER + 3) / 4);
        ProfScope ps(ctx, "k_gray_border");
        hipLaunchKernelGGL(k_gray_border, grd, blk, 0, ctx->stream, d_img, stride, channels, ctx->d_raw, L);
    }
    for (int l = 1; l < nlevels; ++l) {
        const LevelInfo& D = P.lv[l];
        dim3 blk(64, 4), grd((D.stride / 4 + 63) / 64, (D.h + 2 * MVO_BORDER + 3) / 4);
        ProfScope ps(ctx, "k_resize_border");
        hipLaunchKernelGGL(k_resize_border, grd, blk, 0, ctx->stream, ctx->d_raw, P.lv[l - 1], D, ctx->d_tabs, ctx->orb.pyramid_interpolation != 0 ? 1 : 0);
    }
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int orb_launch_detect(mvo_ctx* ctx) {
    const PyrInfo& P = ctx->pyr;
    {
        ProfScope ps(ctx, "k_fast_nms");
        hipLaunchKernelGGL(k_fast_nms, dim3(P.n_tiles), dim3(256), 0, ctx->stream, ctx->d_raw, ctx->d_score,
                           ctx->d_cell_mask, ctx->d_cell_cnt, P, ctx->orb.fast_threshold);
    }
    {
        ProfScope ps(ctx, "k_scan_cells");
        hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_cell_cnt, ctx->d_cell_off, ctx->d_hdr, P);
    }
    {
        ProfScope ps(ctx, "k_emit_cells");
        hipLaunchKernelGGL(k_emit_cells, dim3((P.n_cells + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_cell_mask,
                           ctx->d_cell_off, ctx->d_score, ctx->d_cand, P, ctx->cand_cap);
    }
    {
        ProfScope ps(ctx, "k_harris_angle");
        hipLaunchKernelGGL(k_harris_angle, dim3(1024), dim3(256), 0, ctx->stream, ctx->d_raw, ctx->d_cand,
                           ctx->d_hdr, P, ctx->cand_cap);
    }
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int orb_launch_blur(mvo_ctx* ctx, int nlevels) {
    const PyrInfo& P = ctx->pyr;
    int nb = nlevels < P.nlevels ? P.lv[nlevels].btile_off : P.n_btiles;
    ProfScope ps(ctx, "k_blur");
    hipLaunchKernelGGL(k_blur, dim3(nb), dim3(256), 0, ctx->stream, ctx->d_raw, ctx->d_blur, P, nlevels);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int orb_launch_brief(mvo_ctx* ctx, int n, const DevDescKp* kps) {
    if (n <= 0) return MVO_OK;
    ProfScope ps(ctx, "k_brief");
    hipLaunchKernelGGL(k_brief, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, ctx->d_blur, kps, ctx->d_desc, ctx->pyr, n);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}
