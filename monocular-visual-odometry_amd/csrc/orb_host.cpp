// csrc/orb_host.cpp -- host half of the ORB path: pyramid geometry, per-level feature quotas, resize
// coefficient tables, the order-sensitive selections (KeyPointsFilter::retainBest, selectUniformKptsByGrid,
// runByImageBorder) and the orchestration of the device kernels.
// These steps stay on the host ON PURPOSE: cv::ORB's retainBest is std::nth_element + std::partition and the
// reference's selectUniformKptsByGrid (src/geometry/feature_match.cpp:51-84) is first-come, so the *set* of
// surviving keypoints depends on libstdc++'s element order; <= 10^4 items, ~0.1 ms.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mvo_internal.h"

// MVO_HOST_TIMING=1: per-stage wall clock of the host half (printed every 200 frames to stderr; development aid)
struct HostTimes {
    double acc[8] = {0};
    long n = 0;
    std::chrono::steady_clock::time_point t;
    bool on = std::getenv("MVO_HOST_TIMING") != nullptr;
    void start() {
        if (on) t = std::chrono::steady_clock::now();
    }
    void lap(int k) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    }
    void frame(const char* const* names, int cnt) {
        if (!on || ++n % 200) return;
        std::fprintf(stderr, "[mvo host us/frame]");
        for (int k = 0; k < cnt; ++k) std::fprintf(stderr, " %s %.1f", names[k], acc[k] / 200), acc[k] = 0;
        std::fprintf(stderr, "\n");
    }
};
static thread_local HostTimes g_ht_detect, g_ht_describe;

namespace {

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_floor(double v) {
    int i = (int)v;
    return i - (i > v);
}
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ORB_Impl::getScale: scaleFactor is a float stored in a double member.
float layer_scale(const mvo_orb_params& p, int level) { return (float)std::pow((double)p.scale_factor, (double)level); }

void feature_quota(const mvo_orb_params& p, std::vector<int>& q) {
    q.assign(p.nlevels, 0);
    float factor = (float)(1.0 / (double)p.scale_factor);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; ++l) {
        q[l] = cv_round(nd);
        sum += q[l];
        nd *= factor;
    }
    q[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
}

// cv::resize as cv::ORB calls it, 8-bit: source offset + two coefficients per destination sample.
// exact (cv::INTER_LINEAR_EXACT, OpenCV >= 3.4): coordinate in double, 8-bit coefficients summing to 256;
// otherwise (cv::INTER_LINEAR): coordinate in float, 11-bit coefficients.
void fill_resize_tab(ResizeEntry* tab, int ssize, int dsize, bool exact) {
    if (exact) {
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
            const int c1 = cv_round(f * 256.0);
            tab[d].ofs = s;
            tab[d].c0 = (int16_t)(256 - c1);
            tab[d].c1 = (int16_t)c1;
        }
        return;
    }
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) {
            f = 0;
            s = 0;
        }
        if (s >= ssize - 1) {
            f = 0;
            s = ssize - 1;
        }
        tab[d].ofs = s;
        tab[d].c0 = (int16_t)cv_round((1.f - f) * 2048);
        tab[d].c1 = (int16_t)cv_round(f * 2048);
    }
}

template <class T>
int free_dev(T*& p) {
    if (p) mvo_free_on_current_device(p);
    p = nullptr;
    return 0;
}

// KeyPointsFilter::retainBest
template <class Key>
void retain_best(std::vector<DevCandidate>& v, int n, Key key) {
    if (n >= 0 && (int)v.size() > n) {
        if (n == 0) {
            v.clear();
            return;
        }
        std::nth_element(v.begin(), v.begin() + n - 1, v.end(),
                         [&](const DevCandidate& a, const DevCandidate& b) { return key(a) > key(b); });
        const float amb = key(v[n - 1]);
        auto new_end = std::partition(v.begin() + n, v.end(), [&](const DevCandidate& a) { return key(a) >= amb; });
        v.resize(new_end - v.begin());
    }
}

}  // namespace

int mvo_ensure_pinned(mvo_ctx* ctx, size_t bytes) {
    if (ctx->h_pin_cap >= bytes) return MVO_OK;
    if (ctx->h_pin) ba_service_free(ctx->device, ctx->h_pin, true);
    ctx->h_pin = nullptr;
    ctx->h_pin_cap = 0;
    size_t cap = round_up(bytes + bytes / 2, 1 << 16);
    MVO_HIP(hipHostMalloc((void**)&ctx->h_pin, cap, hipHostMallocDefault));
    ctx->h_pin_cap = cap;
    return MVO_OK;
}

// (Re)builds the pyramid geometry + device buffers for a (w, h) image.
int orb_setup_geometry(mvo_ctx* ctx, int w, int h) {
    if (ctx->img_w == w && ctx->img_h == h && ctx->d_raw) return MVO_OK;
    const mvo_orb_params& p = ctx->orb;
    PyrInfo P{};
    P.nlevels = p.nlevels;
    size_t off = 256, tab = 0;
    int cells = 0, tiles = 0, btiles = 0;
    for (int l = 0; l < p.nlevels; ++l) {
        LevelInfo& L = P.lv[l];
        L.scale = layer_scale(p, l);
        L.w = cv_round(w / L.scale);
        L.h = cv_round(h / L.scale);
        if (L.w < 8 || L.h < 8) return mvo_set_err(ctx, MVO_ERR_INVALID, "image too small for the pyramid", hipSuccess);
        L.stride = (int)round_up(L.w + 2 * MVO_BORDER, 64);
        L.off = (int)off;
        off = round_up(off + (size_t)L.stride * (L.h + 2 * MVO_BORDER) + 256, 256);
        L.tiles_x = (L.w + 63) / 64;
        L.tiles_y = (L.h + 15) / 16;
        if (L.tiles_x > FT_ROW_TILES) return mvo_set_err(ctx, MVO_ERR_INVALID, "image too wide", hipSuccess);
        L.tile_off = tiles;
        tiles += L.tiles_x * L.tiles_y;
        L.cell_off = cells;
        cells += L.h * L.tiles_x;
        L.btiles_x = L.stride / 64;
        L.btiles_y = (L.h + 2 * MVO_BORDER + 15) / 16;
        L.btile_off = btiles;
        btiles += L.btiles_x * L.btiles_y;
        L.tab_off = (int)tab;
        tab += L.w + L.h;
    }
    P.n_cells = cells;
    P.n_tiles = tiles;
    P.n_btiles = btiles;
    const size_t bytes = off + 4096;
    free_dev(ctx->d_raw);
    free_dev(ctx->d_blur);
    free_dev(ctx->d_tabs);
    free_dev(ctx->d_pyr_regs);
    free_dev(ctx->d_fh_slots);
    free_dev(ctx->d_fh_line);
    free_dev(ctx->d_fh_arrive);
    MVO_HIP(hipMalloc((void**)&ctx->d_fh_slots, (size_t)tiles * FT_TILE_CAP * sizeof(DevCandidate)));
    MVO_HIP(hipMalloc((void**)&ctx->d_fh_line, (size_t)tiles * 16));
    MVO_HIP(hipMalloc((void**)&ctx->d_fh_arrive, (size_t)tiles * 4));
    MVO_HIP(hipMemsetAsync(ctx->d_fh_arrive, 0, (size_t)tiles * 4, ctx->stream));
    MVO_HIP(hipMalloc((void**)&ctx->d_raw, bytes));
    MVO_HIP(hipMalloc((void**)&ctx->d_blur, bytes));
    MVO_HIP(hipMemsetAsync(ctx->d_raw, 0, bytes, ctx->stream));
    MVO_HIP(hipMemsetAsync(ctx->d_blur, 0, bytes, ctx->stream));
    MVO_HIP(hipMalloc((void**)&ctx->d_tabs, tab * sizeof(ResizeEntry)));
    std::vector<ResizeEntry> tabs(tab);
    for (int l = 1; l < p.nlevels; ++l) {
        fill_resize_tab(&tabs[P.lv[l].tab_off], P.lv[l - 1].w, P.lv[l].w, ctx->orb.pyramid_interpolation != 0);
        fill_resize_tab(&tabs[P.lv[l].tab_off + P.lv[l].w], P.lv[l - 1].h, P.lv[l].h, ctx->orb.pyramid_interpolation != 0);
    }
    MVO_HIP(hipMemcpy(ctx->d_tabs, tabs.data(), tab * sizeof(ResizeEntry), hipMemcpyHostToDevice));
    // which level groups can run the LDS-tiled pyramid kernel: every tile's source regions must fit its LDS pool
    auto reflect = [](int i, int n) {
        if (n == 1) return 0;
        while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
        return i;
    };
    std::vector<PyrTileRegs> tile_regs((size_t)std::max(btiles, 1));
    std::memset(tile_regs.data(), 0, tile_regs.size() * sizeof(PyrTileRegs));
    for (int l0 = 0; l0 < p.nlevels; l0 += 4) {
        bool fits = true;
        int need = 0;
        const int base = l0 == 0 ? 0 : l0 - 1;
        for (int l = std::max(l0, 1); l < std::min(l0 + 4, p.nlevels) && fits; ++l) {
            const LevelInfo& L = P.lv[l];
            for (int ty = 0; ty < L.btiles_y && fits; ++ty)
                for (int tx = 0; tx < L.btiles_x && fits; ++tx) {
                    int xlo = 1 << 30, xhi = -1, ylo = 1 << 30, yhi = -1;
                    for (int i = 0; i < PT_W; ++i) {
                        const int x = reflect(std::min(tx * PT_W + i, L.w + 2 * MVO_BORDER - 1) - MVO_BORDER, L.w);
                        xlo = std::min(xlo, x), xhi = std::max(xhi, x);
                    }
                    for (int i = 0; i < PT_H; ++i) {
                        const int y = reflect(std::min(ty * PT_H + i, L.h + 2 * MVO_BORDER - 1) - MVO_BORDER, L.h);
                        ylo = std::min(ylo, y), yhi = std::max(yhi, y);
                    }
                    PyrRegion reg[8];
                    int off[8];
                    const int used = pyr_regions(P, tabs.data(), l, l - base, xlo, xhi, ylo, yhi, reg, off);
                    need = std::max(need, used);
                    fits = used <= PYR_LDS_BYTES;
                    PyrTileRegs& T = tile_regs[(size_t)L.btile_off + (size_t)ty * L.btiles_x + tx];
                    for (int d = 1; d <= l - base && d <= 4; ++d) T.reg[d] = reg[d], T.off[d] = off[d];
                }
        }
        ctx->pyr_group_tiled[l0 / 4] = fits;
        ctx->pyr_group_lds[l0 / 4] = std::min(PYR_LDS_BYTES, (need + 255) & ~255);  // (dynamic LDS of the group's launch)
    }
    MVO_HIP(hipMalloc((void**)&ctx->d_pyr_regs, tile_regs.size() * sizeof(PyrTileRegs)));
    MVO_HIP(hipMemcpy(ctx->d_pyr_regs, tile_regs.data(), tile_regs.size() * sizeof(PyrTileRegs), hipMemcpyHostToDevice));
    ctx->pyr = P;
    ctx->pyr_bytes = bytes;
    ctx->img_w = w;
    ctx->img_h = h;
    ctx->pyr_valid = ctx->blur_valid = false;
    feature_quota(p, ctx->quota);
    return MVO_OK;
}

static int ensure_kp_cap(mvo_ctx* ctx, int n) {
    if (ctx->kp_cap < n) {
        // (growing drops the previous frame's device descriptors: callers keep n <= 4096 or re-extract)
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        free_dev(ctx->d_kp);
        free_dev(ctx->d_desc_buf);
        int cap = std::max(4096, n + n / 2);
        MVO_HIP(hipMalloc((void**)&ctx->d_kp, (size_t)cap * sizeof(DevDescKp)));
        MVO_HIP(hipMalloc((void**)&ctx->d_desc_buf, (size_t)cap * 64));
        ctx->kp_cap = cap;
    }
    ctx->desc_flip ^= 1;
    ctx->d_desc = ctx->d_desc_buf + (size_t)ctx->desc_flip * ctx->kp_cap * 32;
    return MVO_OK;
}

// geometry::selectUniformKptsByGrid (feature_match.cpp:51-84)
int orb_grid_select(mvo_ctx* ctx, std::vector<mvo_keypoint>& kps, int image_rows, int image_cols) {
    const mvo_orb_params& p = ctx->orb;
    if (ctx->grid_rows == 0) {  // latched from the first image (feature_match.cpp:59-62)
        ctx->grid_rows = image_rows / p.grid_size;
        ctx->grid_cols = image_cols / p.grid_size;
    }
    const int rows = ctx->grid_rows, cols = ctx->grid_cols;
    std::vector<int> grid((size_t)rows * cols, 0);
    std::vector<mvo_keypoint> tmp;
    int cnt = 0;
    for (const mvo_keypoint& k : kps) {
        int row = ((int)k.y) / p.grid_size, col = ((int)k.x) / p.grid_size;
        if (row < 0 || row >= rows || col < 0 || col >= cols)
            return mvo_set_err(ctx, MVO_ERR_INVALID, "keypoint outside the latched grid", hipSuccess);
        int& g = grid[(size_t)row * cols + col];
        if (g < p.grid_max_per_cell) {
            tmp.push_back(k);
            g++;
            cnt++;
            if (cnt > p.max_keypoints) break;  // feature_match.cpp:77: yields max+1 keypoints
        }
    }
    kps.swap(tmp);
    return MVO_OK;
}

// cv::ORB::detect on an image already in device memory; leaves the raw pyramid cached in the ctx.
int orb_detect_device(mvo_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, int channels,
                      std::vector<mvo_keypoint>& out) {
    HostTimes& ht = g_ht_detect;
    ht.start();
    int r = orb_setup_geometry(ctx, w, h);
    if (r) return r;
    const PyrInfo& P = ctx->pyr;
    ctx->pyr_valid = ctx->blur_valid = false;
    if ((r = mvo_ensure_pinned(ctx, orb_detect_host_bytes(P.n_tiles)))) return r;
    ExtractGate gate(ctx);
    if ((r = orb_launch_pyramid(ctx, d_img, stride, channels, P.nlevels))) return r;
    // the kernel delivers per-tile-row counts and ordered record lists into the pinned buffer itself
    // a ctx in throughput mode (many sequences share the GPU) leaves the ordering of a tile row to this thread: the GPU time
    // of the ordering step (~8 us of kernel tail) is worth more there than ~75 us of a host thread that has company
    const bool ordered = !ctx->ba_throughput_mode;
    if ((r = orb_launch_detect(ctx, ctx->h_pin, ordered))) return r;
    MVO_HIP(hipEventRecord(ctx->ev, ctx->stream));
    // a ctx that describes from whole blurred levels has them blurred now, behind the event: while this thread selects keypoints
    if (orb_brief_from_levels(ctx) && (r = orb_launch_blur(ctx, P.nlevels))) return r;
    ht.lap(0);
    MVO_HIP(hipEventSynchronize(ctx->ev));
    gate.release();
    ht.lap(1);
    const int32_t* counts = (const int32_t*)ctx->h_pin;
    const DevCandidate* slots = (const DevCandidate*)(ctx->h_pin + orb_detect_counts_bytes(P.n_tiles));
    std::vector<DevCandidate>& all = ctx->last_cand;
    all.clear();
    int level_start[MVO_MAX_LEVELS + 1] = {0};
    if (ordered) {
        // canonical order (level, row, column) = the order cv::FAST emits: the kernel delivers every tile row as one ordered
        // list (at the slot of the row's first tile, its length at that tile's count); the rows are appended in order
        for (int l = 0; l < P.nlevels; ++l) {
            const LevelInfo& L = P.lv[l];
            level_start[l] = (int)all.size();
            for (int ty = 0; ty < L.tiles_y; ++ty) {
                const int t0 = L.tile_off + ty * L.tiles_x;
                // the lists are freshly written by the device: pull the next row's lines in while this one is copied
                if (ty + 1 < L.tiles_y) {
                    const char* base = (const char*)(slots + (size_t)(t0 + L.tiles_x) * FT_TILE_CAP);
                    for (int b = 0, nb = counts[t0 + L.tiles_x] * (int)sizeof(DevCandidate); b < nb; b += 64) __builtin_prefetch(base + b);
                }
                const DevCandidate* row = slots + (size_t)t0 * FT_TILE_CAP;
                all.insert(all.end(), row, row + counts[t0]);
            }
        }
    } else {
        int cursor[FT_ROW_TILES];
        // canonical order (level, row, column) = the order cv::FAST emits: inside a tile row the tiles interleave line by
        // line; every tile's slot is row-major already, so one cursor per tile column restores it
        for (int l = 0; l < P.nlevels; ++l) {
            const LevelInfo& L = P.lv[l];
            level_start[l] = (int)all.size();
            for (int ty = 0; ty < L.tiles_y; ++ty) {
                const int t0 = L.tile_off + ty * L.tiles_x;
                int left = 0;
                for (int tx = 0; tx < L.tiles_x; ++tx) {
                    cursor[tx] = 0;
                    left += counts[t0 + tx];
                }
                // the slots are 4 KB apart and freshly written by the device: pull the next tile row's lines in while this
                // one is merged (the hardware prefetcher cannot follow the slot pattern)
                if (ty + 1 < L.tiles_y)
                    for (int tx = 0; tx < L.tiles_x; ++tx) {
                        const char* base = (const char*)(slots + (size_t)(t0 + L.tiles_x + tx) * FT_TILE_CAP);
                        for (int b = 0, nb = counts[t0 + L.tiles_x + tx] * (int)sizeof(DevCandidate); b < nb; b += 64)
                            __builtin_prefetch(base + b);
                    }
                for (int y = ty * 16; left > 0 && y < ty * 16 + 16; ++y)
                    for (int tx = 0; tx < L.tiles_x; ++tx) {
                        const DevCandidate* sl = slots + (size_t)(t0 + tx) * FT_TILE_CAP;
                        const int cnt = counts[t0 + tx];
                        int& c = cursor[tx];
                        while (c < cnt && sl[c].y == y) {
                            all.push_back(sl[c++]);
                            --left;
                        }
                    }
            }
        }
    }
    level_start[P.nlevels] = (int)all.size();
    const DevCandidate* cand = all.data();
    ht.lap(2);
    out.clear();
    std::vector<DevCandidate> lv;
    for (int l = 0; l < P.nlevels; ++l) {
        lv.assign(cand + level_start[l], cand + level_start[l + 1]);
        // FAST score first (keep 2x), then Harris (cv::ORB computeKeyPoints)
        retain_best(lv, 2 * ctx->quota[l], [](const DevCandidate& a) { return (float)(a.level_score & 0xffff); });
        retain_best(lv, ctx->quota[l], [](const DevCandidate& a) { return a.harris; });
        const float sf = P.lv[l].scale;
        for (const DevCandidate& a : lv) {
            mvo_keypoint k;
            k.x = (float)a.x * sf;
            k.y = (float)a.y * sf;
            k.size = 31 * sf;
            k.angle = a.angle;
            k.response = a.harris;
            k.octave = l;
            k.class_id = -1;
            out.push_back(k);
        }
    }
    ctx->pyr_valid = true;
    ctx->pyr_levels_built = P.nlevels;
    ht.lap(3);
    static const char* const names[] = {"launch", "wait", "merge", "retain"};
    ht.frame(names, 4);
    return MVO_OK;
}

// cv::ORB::compute given keypoints; the raw pyramid must be valid for levels < nlevels_needed.
int orb_describe_device(mvo_ctx* ctx, std::vector<mvo_keypoint>& kps, int w, int h, uint8_t* desc_host) {
    const PyrInfo& P = ctx->pyr;
    const int n = (int)kps.size();
    if (n == 0) return MVO_OK;
    HostTimes& ht = g_ht_describe;
    ht.start();
    int r = ensure_kp_cap(ctx, n);
    if (r) return r;
    if ((r = mvo_ensure_pinned(ctx, (size_t)n * (sizeof(DevDescKp) + 32)))) return r;
    DevDescKp* hk = (DevDescKp*)ctx->h_pin;
    for (int i = 0; i < n; ++i) {
        const mvo_keypoint& k = kps[i];
        const float scale = 1.f / P.lv[k.octave].scale;
        const float angle = k.angle * (float)(M_PI / 180.f);
        hk[i].cx = (int16_t)cv_round(k.x * scale);
        hk[i].cy = (int16_t)cv_round(k.y * scale);
        hk[i].level = k.octave;
        double sn, cs;  // one argument reduction for both (glibc: the same bits as sin() and cos())
        ::sincos((double)angle, &sn, &cs);
        hk[i].a = (float)cs;
        hk[i].b = (float)sn;
        const LevelInfo& L = P.lv[k.octave];
        // the tap window must stay inside the 32-px frame
        if (hk[i].cx < -12 || hk[i].cx > L.w + 11 || hk[i].cy < -12 || hk[i].cy > L.h + 11)
            return mvo_set_err(ctx, MVO_ERR_INVALID, "keypoint outside its pyramid level", hipSuccess);
    }
    // the kernel reads the 16-byte keypoint records straight from the pinned staging buffer (one load per wave) and
    // writes the descriptors into it as well as into device memory: no copy dispatch on the frame's critical path; the
    // buffer is not touched again before the synchronisation below
    uint8_t* hd = desc_host ? ctx->h_pin + (size_t)n * sizeof(DevDescKp) : nullptr;
    ht.lap(0);
    ExtractGate gate(ctx);
    if ((r = orb_launch_brief(ctx, n, hk, hd))) return r;
    ht.lap(1);
    MVO_HIP(hipStreamSynchronize(ctx->stream));
    gate.release();
    ht.lap(2);
    if (desc_host) std::memcpy(desc_host, hd, (size_t)n * 32);
    ht.lap(3);
    static const char* const names[] = {"prep", "launch", "wait", "copy"};
    ht.frame(names, 4);
    (void)w;
    (void)h;
    return MVO_OK;
}

// KeyPointsFilter::runByImageBorder(keypoints, image.size(), 31): Point2f -> Point rounds half-to-even
void orb_border_filter(std::vector<mvo_keypoint>& kps, int w, int h) {
    std::vector<mvo_keypoint> keep;
    keep.reserve(kps.size());
    for (const mvo_keypoint& k : kps) {
        int xi = cv_round(k.x), yi = cv_round(k.y);
        if (xi >= 31 && xi < w - 31 && yi >= 31 && yi < h - 31) keep.push_back(k);
    }
    kps.swap(keep);
}
