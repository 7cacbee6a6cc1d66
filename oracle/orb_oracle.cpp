// oracle/orb_oracle.cpp -- CPU ORACLE (test infrastructure, see oracle.h): scalar restatement of
// geometry::calcKeyPoints / selectUniformKptsByGrid / calcDescriptors
// (reference src/geometry/feature_match.cpp:11-84, include/my_slam/vo/frame.h:73-86) including the
// cv::ORB arithmetic those functions delegate to (not vendored; semantics per SURVEY.md Appendix A.1).
// PARITY UNPINNED -- no upstream golden vectors exist; every rounding rule below is the canonical
// arithmetic the HIP path is checked against bit for bit.
// Build with -ffp-contract=off: float expressions must not be fused.
#include "oracle.h"
#include "orb_pattern_31.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int kBorder = 32;        // max(edgeThreshold 31, descPatchSize 22, HARRIS_BLOCK_SIZE/2)+1
constexpr int kEdge = 31;          // edgeThreshold (feature_match.cpp:23)
constexpr int kPatch = 31;         // patchSize
constexpr int kHalfPatch = 15;
constexpr float kHarrisK = 0.04f;

inline int cvRound(double v) { return (int)std::lrint(v); }  // round-half-to-even
inline int cvFloor(double v) {
    int i = (int)v;
    return i - (i > v);
}
inline int cvCeil(double v) {
    int i = (int)v;
    return i + (i < v);
}
inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

struct Level {
    int w = 0, h = 0, stride = 0;
    float scale = 1.f;
    std::vector<uint8_t> buf;  // (h+64) x stride, interior origin at (32,32)
    uint8_t* at(int x, int y) { return &buf[(size_t)(y + kBorder) * stride + x + kBorder]; }
    const uint8_t* at(int x, int y) const { return &buf[(size_t)(y + kBorder) * stride + x + kBorder]; }
    void alloc(int w_, int h_) {
        w = w_;
        h = h_;
        stride = w + 2 * kBorder;
        buf.assign((size_t)(h + 2 * kBorder) * stride, 0);
    }
    void fillBorder() {  // copyMakeBorder(BORDER_REFLECT_101)
        for (int y = -kBorder; y < h + kBorder; ++y) {
            int sy = reflect101(y, h);
            for (int x = -kBorder; x < w + kBorder; ++x) {
                if (x >= 0 && x < w && y >= 0 && y < h) continue;
                *at(x, y) = *at(reflect101(x, w), sy);
            }
        }
    }
};

float layerScale(const orc_orb_params& p, int level) {
    // ORB_Impl stores the float scaleFactor in a double; getScale = (float)pow(scaleFactor, level)
    return (float)std::pow((double)p.scale_factor, (double)level);
}

void levelSize(int w, int h, const orc_orb_params& p, int level, int& lw, int& lh, float& sc) {
    sc = layerScale(p, level);
    lw = cvRound(w / sc);
    lh = cvRound(h / sc);
}

// cv::cvtColor(BGR2GRAY) 8-bit fixed point: (1868 B + 9617 G + 4899 R + 2^13) >> 14
void toGray(const uint8_t* img, int w, int h, int stride, int ch, Level& L) {
    L.alloc(w, h);
    L.scale = 1.f;
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = img + (size_t)y * stride;
        uint8_t* d = L.at(0, y);
        if (ch == 1) {
            std::memcpy(d, s, w);
        } else {
            for (int x = 0; x < w; ++x) {
                const uint8_t* px = s + x * ch;
                d[x] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
            }
        }
    }
    L.fillBorder();
}

// cv::ORB builds its pyramid with cv::resize.  Two arithmetic flavours exist upstream:
//   INTER_LINEAR (OpenCV < 3.4; modules/imgproc/src/resize.cpp, HResizeLinear / VResizeLinear for uchar): source coordinate in
//     FLOAT, 11-bit coefficients cvRound(f * 2048), vertical pass ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
//   INTER_LINEAR_EXACT (OpenCV >= 3.4, what cv::ORB uses there; resize.cpp bit-exact path on ufixedpoint16): source coordinate
//     in DOUBLE (softdouble), 8-bit coefficients cvRound(f * 256) with the pair summing to 256, horizontal pass exact in 8.8,
//     vertical pass rounded once: (b0 * h0 + b1 * h1 + 2^15) >> 16.
// The reference's README requires OpenCV >= 3.4.5 / 4.0, so EXACT is the canonical flavour (orc_orb_params default 1).
struct ResizeTab {
    std::vector<int> ofs;     // source index
    std::vector<short> coef;  // 2 per destination sample
};
ResizeTab makeTabExact(int ssize, int dsize) {
    ResizeTab t;
    t.ofs.resize(dsize);
    t.coef.resize(2 * dsize);
    const double scale = (double)ssize / (double)dsize;
    for (int d = 0; d < dsize; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) {
            f = 0;
            s = 0;
        }
        if (s >= ssize - 1) {
            f = 0;
            s = ssize - 1;
        }
        const int c1 = cvRound(f * 256.0);
        t.ofs[d] = s;
        t.coef[2 * d] = (short)(256 - c1);
        t.coef[2 * d + 1] = (short)c1;
    }
    return t;
}
ResizeTab makeTab(int ssize, int dsize) {
    ResizeTab t;
    t.ofs.resize(dsize);
    t.coef.resize(2 * dsize);
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cvFloor(f);
        f -= s;
        if (s < 0) {
            f = 0;
            s = 0;
        }
        if (s >= ssize - 1) {
            f = 0;
            s = ssize - 1;
        }
        t.ofs[d] = s;
        t.coef[2 * d] = (short)cvRound((1.f - f) * 2048);
        t.coef[2 * d + 1] = (short)cvRound(f * 2048);
    }
    return t;
}
void resizeLinear(const Level& S, Level& D, bool exact) {
    ResizeTab tx = exact ? makeTabExact(S.w, D.w) : makeTab(S.w, D.w), ty = exact ? makeTabExact(S.h, D.h) : makeTab(S.h, D.h);
    for (int dy = 0; dy < D.h; ++dy) {
        int sy0 = ty.ofs[dy], sy1 = std::min(sy0 + 1, S.h - 1);
        int b0 = ty.coef[2 * dy], b1 = ty.coef[2 * dy + 1];
        const uint8_t* r0 = S.at(0, sy0);
        const uint8_t* r1 = S.at(0, sy1);
        uint8_t* d = D.at(0, dy);
        for (int dx = 0; dx < D.w; ++dx) {
            int sx0 = tx.ofs[dx], sx1 = std::min(sx0 + 1, S.w - 1);
            int a0 = tx.coef[2 * dx], a1 = tx.coef[2 * dx + 1];
            int h0 = r0[sx0] * a0 + r0[sx1] * a1;
            int h1 = r1[sx0] * a0 + r1[sx1] * a1;
            d[dx] = exact ? (uint8_t)((b0 * h0 + b1 * h1 + 32768) >> 16)
                          : (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

void buildPyramid(const uint8_t* img, int w, int h, int stride, int ch, const orc_orb_params& p,
                  int nlevels, std::vector<Level>& pyr) {
    pyr.resize(nlevels);
    toGray(img, w, h, stride, ch, pyr[0]);
    for (int l = 1; l < nlevels; ++l) {
        int lw, lh;
        float sc;
        levelSize(w, h, p, l, lw, lh, sc);
        pyr[l].alloc(lw, lh);
        pyr[l].scale = sc;
        resizeLinear(pyr[l - 1], pyr[l], p.pyramid_interpolation != 0);  // each level from the PREVIOUS level
        pyr[l].fillBorder();
    }
}

// cv::GaussianBlur(7x7, sigma 2) canonical fixed point: 8-bit kernel (sum 256, error diffused from
// the tails to the centre), horizontal pass kept in 8.8, vertical pass rounded (+2^15) >> 16.
// The kernel is DERIVED the way OpenCV's bit-exact path does (getGaussianKernelBitExact + fixed-point conversion with
// error diffusion, modules/imgproc/src/smooth.dispatch.cpp): k_i = exp(-i^2 / (2 sigma^2)) normalised to 1, then from the
// outermost tap inwards v_i = cvRound(k_i * 256 + err), err = (k_i * 256 + err) - v_i, and the centre takes what is left
// of 256.  For ksize 7, sigma 2 this gives {18, 34, 48, 56, 48, 34, 18} (tests/test_oracle_orb.py checks it).
void gaussKernelFixed(int ksize, double sigma, int bits, int* out) {
    std::vector<double> k(ksize);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        k[i] = std::exp(-0.5 / (sigma * sigma) * x * x);
        sum += k[i];
    }
    const double mul = (double)(1 << bits);
    double err = 0;
    int acc = 0;
    for (int i = 0; i < ksize / 2; ++i) {
        const double adj = k[i] / sum * mul + err;
        const int v = cvRound(adj);
        err = adj - v;
        out[i] = out[ksize - 1 - i] = v;
        acc += 2 * v;
    }
    out[ksize / 2] = (1 << bits) - acc;
}
struct Gauss7 {
    int k[7];
    Gauss7() { gaussKernelFixed(7, 2.0, 8, k); }
};
const Gauss7 kGaussDerived;
const int* const kGauss7 = kGaussDerived.k;
void blurLevel(const Level& S, Level& D) {
    D = S;  // frame stays unblurred
    std::vector<int> hbuf((size_t)(S.h + 6) * S.w);
    for (int y = -3; y < S.h + 3; ++y)
        for (int x = 0; x < S.w; ++x) {
            const uint8_t* c = S.at(x, y);
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += kGauss7[k] * c[k - 3];
            hbuf[(size_t)(y + 3) * S.w + x] = acc;
        }
    for (int y = 0; y < S.h; ++y)
        for (int x = 0; x < S.w; ++x) {
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += kGauss7[k] * hbuf[(size_t)(y + k) * S.w + x];
            *D.at(x, y) = (uint8_t)((acc + 32768) >> 16);
        }
}

// FAST-9/16 circle, same enumeration as cv::FAST (x, y)
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// returns 0 for a non-corner, else cornerScore<16> = (largest t that keeps it a corner)
int fastScore(const Level& L, int x, int y, int t) {
    const uint8_t* c = L.at(x, y);
    int v = c[0];
    int d[25];
    for (int k = 0; k < 16; ++k) d[k] = v - c[kCircle[k][0] + kCircle[k][1] * L.stride];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int A = -256, B = -256;
    for (int k = 0; k < 16; ++k) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; ++j) {
            mn = std::min(mn, d[k + j]);
            mx = std::max(mx, d[k + j]);
        }
        A = std::max(A, mn);   // centre brighter than the whole arc by at least mn
        B = std::max(B, -mx);  // centre darker
    }
    int best = std::max(A, B);
    if (best <= t) return 0;
    return best - 1;
}

// HarrisResponses(blockSize 7): integer Sobel-like gradients, float response.
float harrisResponse(const Level& L, int x0, int y0) {
    const int step = L.stride;
    int a = 0, b = 0, c = 0;
    for (int i = -3; i <= 3; ++i)
        for (int j = -3; j <= 3; ++j) {
            const uint8_t* p = L.at(x0 + j, y0 + i);
            int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    float scale = 1.f / ((1 << 2) * 7 * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    float fa = (float)a, fb = (float)b, fc = (float)c;
    return (fa * fb - fc * fc - kHarrisK * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

// cv::fastAtan2 (degrees), 7th-order polynomial
float fastAtan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
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

struct Umax {
    int u[kHalfPatch + 2];
    Umax() {
        int v, v0, vmax = cvFloor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(kHalfPatch * std::sqrt(2.f) / 2);
        for (v = 0; v <= vmax; ++v) u[v] = cvRound(std::sqrt((double)kHalfPatch * kHalfPatch - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (u[v0] == u[v0 + 1]) ++v0;
            u[v] = v0;
            ++v0;
        }
    }
};
const Umax kUmax;

float icAngle(const Level& L, int x, int y) {
    const uint8_t* c = L.at(x, y);
    const int step = L.stride;
    int m01 = 0, m10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * c[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vsum = 0, d = kUmax.u[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * step], vm = c[u - v * step];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return fastAtan2((float)m01, (float)m10);
}

void featureQuota(const orc_orb_params& p, std::vector<int>& q) {
    q.assign(p.nlevels, 0);
    float factor = (float)(1.0 / (double)p.scale_factor);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; ++l) {
        q[l] = cvRound(nd);
        sum += q[l];
        nd *= factor;
    }
    q[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
}

// FAST + 3x3 NMS + 31-px border, in the order cv::FAST emits (row-major).
void levelCandidates(const Level& L, int level, int thr, std::vector<orc_candidate>& out) {
    if (L.w <= 2 * kEdge || L.h <= 2 * kEdge) return;
    const int x0 = kEdge - 1, x1 = L.w - kEdge + 1, y0 = kEdge - 1, y1 = L.h - kEdge + 1;
    const int sw = x1 - x0, sh = y1 - y0;
    std::vector<int> sc((size_t)sw * sh);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sc[(size_t)(y - y0) * sw + (x - x0)] = fastScore(L, x, y, thr);
    for (int y = kEdge; y < L.h - kEdge; ++y)
        for (int x = kEdge; x < L.w - kEdge; ++x) {
            const int* s = &sc[(size_t)(y - y0) * sw + (x - x0)];
            int v = s[0];
            if (v == 0) continue;
            if (v > s[-1] && v > s[1] && v > s[-sw - 1] && v > s[-sw] && v > s[-sw + 1] && v > s[sw - 1] &&
                v > s[sw] && v > s[sw + 1]) {
                orc_candidate c;
                c.x = (int16_t)x;
                c.y = (int16_t)y;
                c.level = level;
                c.fast_score = v;
                c.harris = harrisResponse(L, x, y);
                c.angle = icAngle(L, x, y);
                out.push_back(c);
            }
        }
}

// KeyPointsFilter::retainBest: nth_element + keep ties with the n-th response (permutes order).
template <class T, class Key>
void retainBest(std::vector<T>& v, int n, Key key) {
    if (n >= 0 && (int)v.size() > n) {
        if (n == 0) {
            v.clear();
            return;
        }
        std::nth_element(v.begin(), v.begin() + n - 1, v.end(),
                         [&](const T& a, const T& b) { return key(a) > key(b); });
        float amb = key(v[n - 1]);
        auto new_end = std::partition(v.begin() + n, v.end(), [&](const T& a) { return key(a) >= amb; });
        v.resize(new_end - v.begin());
    }
}

void detect(const uint8_t* img, int w, int h, int stride, int ch, const orc_orb_params& p,
            std::vector<orc_keypoint>& kps) {
    std::vector<Level> pyr;
    buildPyramid(img, w, h, stride, ch, p, p.nlevels, pyr);
    std::vector<int> quota;
    featureQuota(p, quota);
    kps.clear();
    for (int l = 0; l < p.nlevels; ++l) {
        std::vector<orc_candidate> c;
        levelCandidates(pyr[l], l, p.fast_threshold, c);
        retainBest(c, 2 * quota[l], [](const orc_candidate& a) { return (float)a.fast_score; });
        retainBest(c, quota[l], [](const orc_candidate& a) { return a.harris; });
        float sf = pyr[l].scale;
        for (const orc_candidate& a : c) {
            orc_keypoint k;
            k.x = (float)a.x * sf;
            k.y = (float)a.y * sf;
            k.size = kPatch * sf;
            k.angle = a.angle;
            k.response = a.harris;
            k.octave = l;
            k.class_id = -1;
            kps.push_back(k);
        }
    }
}

int gridSelect(std::vector<orc_keypoint>& kps, int rows, int cols, const orc_orb_params& p) {
    std::vector<int> grid((size_t)rows * cols, 0);
    std::vector<orc_keypoint> tmp;
    int cnt = 0;
    for (const orc_keypoint& k : kps) {
        int row = ((int)k.y) / p.grid_size, col = ((int)k.x) / p.grid_size;
        if (row < 0 || row >= rows || col < 0 || col >= cols) return -2;  // reference would index OOB
        if (grid[(size_t)row * cols + col] < p.grid_max_per_cell) {
            tmp.push_back(k);
            grid[(size_t)row * cols + col]++;
            cnt++;
            if (cnt > p.max_keypoints) break;  // feature_match.cpp:77 -- yields max+1
        }
    }
    kps.swap(tmp);
    return (int)kps.size();
}

}  // namespace

extern "C" {

int orc_orb_level_size(int w, int h, const orc_orb_params* p, int level, int* lw, int* lh, float* scale) {
    levelSize(w, h, *p, level, *lw, *lh, *scale);
    return 0;
}

int orc_orb_feature_quota(const orc_orb_params* p, int32_t* quota) {
    std::vector<int> q;
    featureQuota(*p, q);
    for (int l = 0; l < p->nlevels; ++l) quota[l] = q[l];
    return p->nlevels;
}

int orc_orb_pyramid_level(const uint8_t* img, int w, int h, int stride, int channels,
                          const orc_orb_params* p, int level, int blurred, uint8_t* out) {
    if (level < 0 || level >= p->nlevels) return -1;
    std::vector<Level> pyr;
    buildPyramid(img, w, h, stride, channels, *p, level + 1, pyr);
    Level L = pyr[level];
    if (blurred) blurLevel(pyr[level], L);
    std::memcpy(out, L.buf.data(), L.buf.size());
    return (int)L.buf.size();
}

int orc_orb_candidates(const uint8_t* img, int w, int h, int stride, int channels,
                       const orc_orb_params* p, orc_candidate* out, int cap) {
    std::vector<Level> pyr;
    buildPyramid(img, w, h, stride, channels, *p, p->nlevels, pyr);
    std::vector<orc_candidate> c;
    for (int l = 0; l < p->nlevels; ++l) levelCandidates(pyr[l], l, p->fast_threshold, c);
    if ((int)c.size() > cap) return -3;
    std::copy(c.begin(), c.end(), out);
    return (int)c.size();
}

int orc_orb_detect(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p,
                   orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> kps;
    detect(img, w, h, stride, channels, *p, kps);
    if ((int)kps.size() > cap) return -3;
    std::copy(kps.begin(), kps.end(), out);
    return (int)kps.size();
}

int orc_select_uniform_kpts_by_grid(orc_keypoint* kps, int n, int grid_rows, int grid_cols,
                                    const orc_orb_params* p) {
    std::vector<orc_keypoint> v(kps, kps + n);
    int r = gridSelect(v, grid_rows, grid_cols, *p);
    if (r < 0) return r;
    std::copy(v.begin(), v.end(), kps);
    return r;
}

int orc_calc_keypoints(const uint8_t* img, int w, int h, int stride, int channels,
                       const orc_orb_params* p, int grid_rows, int grid_cols, orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> kps;
    detect(img, w, h, stride, channels, *p, kps);
    if (grid_rows <= 0) grid_rows = h / p->grid_size;  // latched from the first image (:59-62)
    if (grid_cols <= 0) grid_cols = w / p->grid_size;
    int r = gridSelect(kps, grid_rows, grid_cols, *p);
    if (r < 0) return r;
    if (r > cap) return -3;
    std::copy(kps.begin(), kps.end(), out);
    return r;
}

int orc_calc_descriptors(const uint8_t* img, int w, int h, int stride, int channels,
                         const orc_orb_params* p, orc_keypoint* kps, int n, uint8_t* desc, uint8_t* rgb) {
    // KeyPointsFilter::runByImageBorder(keypoints, image.size(), 31): Point2f -> Point rounds.
    std::vector<orc_keypoint> keep;
    int nlevels = 0;
    for (int i = 0; i < n; ++i) {
        int xi = cvRound(kps[i].x), yi = cvRound(kps[i].y);
        if (xi >= kEdge && xi < w - kEdge && yi >= kEdge && yi < h - kEdge) keep.push_back(kps[i]);
    }
    for (const orc_keypoint& k : keep) nlevels = std::max(nlevels, std::max(k.octave, 0));
    nlevels++;
    if (keep.empty()) return 0;
    std::vector<Level> pyr, blur;
    buildPyramid(img, w, h, stride, channels, *p, nlevels, pyr);
    blur.resize(nlevels);
    for (int l = 0; l < nlevels; ++l) blurLevel(pyr[l], blur[l]);
    for (size_t j = 0; j < keep.size(); ++j) {
        const orc_keypoint& k = keep[j];
        const Level& L = blur[k.octave];
        float scale = 1.f / L.scale;
        float angle = k.angle * (float)(M_PI / 180.f);
        // canonical: double-precision cos/sin rounded to float (computed on the host in the product)
        float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
        const uint8_t* c = L.at(cvRound(k.x * scale), cvRound(k.y * scale));
        uint8_t* d = desc + 32 * j;
        const signed char* pat = MVO_ORB_PATTERN_31;
        for (int i = 0; i < 32; ++i) {
            int val = 0;
            for (int bit = 0; bit < 8; ++bit, pat += 4) {
                float x0 = pat[0] * a - pat[1] * b, y0 = pat[0] * b + pat[1] * a;
                float x1 = pat[2] * a - pat[3] * b, y1 = pat[2] * b + pat[3] * a;
                int t0 = c[cvRound(y0) * L.stride + cvRound(x0)];
                int t1 = c[cvRound(y1) * L.stride + cvRound(x1)];
                val |= (t0 < t1) << bit;
            }
            d[i] = (uint8_t)val;
        }
        if (rgb) {  // frame.h:80-85 + opencv_funcs.cpp:10-32 (BGR -> rgb)
            int x = (int)std::floor(k.x), y = (int)std::floor(k.y);
            const uint8_t* px = img + (size_t)y * stride + x * channels;
            if (channels >= 3) {
                rgb[3 * j + 0] = px[2];
                rgb[3 * j + 1] = px[1];
                rgb[3 * j + 2] = px[0];
            } else {
                rgb[3 * j + 0] = rgb[3 * j + 1] = rgb[3 * j + 2] = px[0];
            }
        }
    }
    std::copy(keep.begin(), keep.end(), kps);
    return (int)keep.size();
}

}  // extern "C"
