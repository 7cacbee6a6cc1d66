// csrc/mvo_internal.h -- context, device buffers and launch plumbing shared by the HIP translation units.
#ifndef MVO_INTERNAL_H
#define MVO_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// The dynamic LDS segment of a kernel and kernel-only attributes.  tests/sim compiles the kernel SOURCES for the host against an
// emulation of the HIP runtime (a test aid: MVO_KERNEL_SIM is defined by its stand-in for <hip/hip_runtime.h>, the library has
// no CPU path); there the segment comes from the emulated launch and the attributes mean nothing.
#ifndef MVO_KERNEL_SIM
#define MVO_DYN_LDS(T, name) extern __shared__ T name[]
#define MVO_DYN_LDS_ALIGNED16(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#define MVO_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#define MVO_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")  // this wave's global stores have left for memory
// the value of a 32-bit vector register is taken as unknown from here on: keeps loop-invariant unpacking / address arithmetic
// INSIDE a loop (hoisted, a dozen packed table entries become three dozen live registers)
#define MVO_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define MVO_DYN_LDS(T, name) T* name = static_cast<T*>(emu_dyn_lds())
#define MVO_DYN_LDS_ALIGNED16(T, name) T* name = static_cast<T*>(emu_dyn_lds())
#define MVO_WAVES_PER_EU(lo, hi)
#define MVO_WAIT_VM0() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define MVO_OPAQUE(x) (void)(x)
#endif

#include <map>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/mvo_hip.h"

#define MVO_MAX_LEVELS 8
#define MVO_BORDER 32  // ORB: max(edgeThreshold 31, descPatchSize 22, HARRIS_BLOCK/2) + 1

// One pyramid level inside the device pyramid buffers (raw, blurred share the geometry).
struct LevelInfo {
    int w, h;        // interior size
    int stride;      // bytes per bordered row, multiple of 64
    int off;         // byte offset of bordered row 0 inside the pyramid buffer
    int tiles_x;     // 64-px-wide NMS cell columns (FAST tiles are 64 x 16 interior pixels)
    int tiles_y;
    int tile_off;    // first FAST tile of this level in the flattened tile list
    int cell_off;    // first (row, tile column) cell of this level in the cell arrays
    int btiles_x;    // blur tiles (64 x 16) over the bordered extent
    int btiles_y;
    int btile_off;
    int tab_off;     // offset of this level's resize tables: [x entries w][y entries h]
    float scale;     // layerScale[level]
};
struct PyrInfo {
    int nlevels;
    int n_cells;
    int n_tiles;
    int n_btiles;
    LevelInfo lv[MVO_MAX_LEVELS];
};

// FAST+NMS survivor as the device emits it (16 bytes), canonical order: level, row, column.
struct DevCandidate {
    int16_t x, y;
    int32_t level_score;  // level << 16 | fast score
    float harris;
    float angle;
};
// Keypoint as uploaded for the rBRIEF kernel.
struct DevDescKp {
    int16_t cx, cy;  // cvRound(pt * 1/scale) in level coordinates
    int32_t level;
    float a, b;  // cos / sin of the keypoint angle, computed on the host
};
struct ResizeEntry {
    int32_t ofs;
    int16_t c0, c1;
};

// Pyramid kernel (k_pyramid): 64 x 16 output tiles; the source regions of a tile, level by level down to the base of
// its level group, are staged in LDS.  pyr_regions() is the one statement of that footprint arithmetic: the device
// uses it per tile, the host uses it to check that every tile of a group fits PYR_LDS_BYTES (otherwise the group runs
// the per-pixel chain kernel).
#define PT_W 64
#define PT_H 16
#define PYR_LDS_BYTES 32768
struct PyrRegion {
    int x0, y0, w, h;  // interior coordinates of its level
};
#ifdef __HIPCC__
#define MVO_HD __host__ __device__
#else
#define MVO_HD
#endif
// reg[d] (d = 1 .. depth) = region of level l-d needed for the interior box [xlo, xhi] x [ylo, yhi] of level l;
// off[d] = its byte offset in the LDS pool.  Returns the pool bytes used.
MVO_HD inline int pyr_regions(const PyrInfo& P, const ResizeEntry* tabs, int l, int depth, int xlo, int xhi, int ylo,
                              int yhi, PyrRegion* reg, int* off) {
    int used = 0;
    for (int d = 1; d <= depth; ++d) {
        const int m = l - d + 1;  // destination level of this step; the region lives on level m-1
        const ResizeEntry* tx = tabs + P.lv[m].tab_off;
        const ResizeEntry* ty = tx + P.lv[m].w;
        const int sw = P.lv[m - 1].w, sh = P.lv[m - 1].h;
        const int x0 = tx[xlo].ofs, x1 = tx[xhi].ofs + 1 < sw - 1 ? tx[xhi].ofs + 1 : sw - 1;
        const int y0 = ty[ylo].ofs, y1 = ty[yhi].ofs + 1 < sh - 1 ? ty[yhi].ofs + 1 : sh - 1;
        reg[d].x0 = x0;
        reg[d].y0 = y0;
        reg[d].w = x1 - x0 + 1;
        reg[d].h = y1 - y0 + 1;
        off[d] = used;
        used += (reg[d].w * reg[d].h + 15) & ~15;
        xlo = x0, xhi = x1, ylo = y0, yhi = y1;
    }
    return used;
}

// The regions of one tile as orb_setup_geometry works them out once per image geometry (the tile grid of a pyramid never
// changes between frames): k_pyramid reads its entry instead of deriving it -- a serial chain of dependent table look-ups on ONE
// thread in front of every deep tile.  128 bytes per bordered tile.
struct PyrTileRegs {
    PyrRegion reg[5];  // [1 .. depth]
    int off[5];
    int pad[7];
};
static_assert(sizeof(PyrTileRegs) == 128, "one 128-byte line per tile");

// k_fast_harris output: every 64 x 16 tile owns FT_TILE_CAP record slots (3x3 NMS leaves at most one survivor per
// 2 x 2 pixels: 256 per tile) and one count, in pinned host memory
#define FT_TILE_CAP 256
#define FT_ROW_TILES 128  // tiles per tile row the in-kernel ordering handles (images up to 8192 px wide)
inline size_t orb_detect_counts_bytes(int n_tiles) { return ((size_t)n_tiles * 4 + 63) / 64 * 64; }
inline size_t orb_detect_host_bytes(int n_tiles) {
    return orb_detect_counts_bytes(n_tiles) + (size_t)n_tiles * FT_TILE_CAP * sizeof(DevCandidate);
}

// tracking rows (track_kernels.hip / track_host.cpp)
struct TrackCamera {
    double fx, fy, cx, cy;
};
struct TrackViewArgs {
    double T[12];  // rows 0..2 of T_c_w = inv(T_w_c)
    double fx, fy, cx, cy;
    int cols, rows;
};
struct mvo_track_state;  // device buffers of the tracking rows, allocated on first use

struct ProfEntry {
    int64_t launches = 0;
    double ms = 0;
};

// Admission gate of the extraction / matching launches of THROUGHPUT-mode contexts (many sequences share the GPU): at most
// `g_extract_concurrency` such launch-and-wait sections are in flight per device (0 = no limit).  The kernels of a frame run on
// the CUs the resident solver grid leaves free; their combined throughput DROPS when too many frames interleave there
// (round 4: 14 frames at once -> 3400 frames/s of extraction, 8-9 at once -> 4600), so the excess waits at the door.
extern std::atomic<int> g_extract_concurrency;
struct mvo_ctx;
struct ExtractGate {
    int device = -1;
    explicit ExtractGate(const mvo_ctx* ctx);
    ~ExtractGate() { release(); }
    void release();
    ExtractGate(const ExtractGate&) = delete;
    ExtractGate& operator=(const ExtractGate&) = delete;
};

inline unsigned long long mvo_next_ctx_uid() {
    static std::atomic<unsigned long long> n{0};
    return ++n;
}
struct mvo_ctx {
    unsigned long long uid = mvo_next_ctx_uid();  // mvo_ctx_uid: never reused
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;  // false: a sibling ctx on its parent's stream (mvo_create_sibling)
    std::string err;
    // --- ORB
    mvo_orb_params orb{};
    bool orb_configured = false;
    int grid_rows = 0, grid_cols = 0;  // latched from the first image
    int img_w = 0, img_h = 0;
    PyrInfo pyr{};
    std::vector<int> quota;
    bool pyr_valid = false, blur_valid = false;
    bool pyr_group_tiled[MVO_MAX_LEVELS] = {false};  // per level group: every tile's regions fit the LDS pool
    int pyr_group_lds[MVO_MAX_LEVELS] = {0};         // ... and the bytes of it the hungriest tile of the group needs
    int pyr_levels_built = 0;
    uint8_t* d_img = nullptr;
    size_t d_img_cap = 0;
    uint8_t *d_raw = nullptr, *d_blur = nullptr;
    size_t pyr_bytes = 0;
    ResizeEntry* d_tabs = nullptr;
    PyrTileRegs* d_pyr_regs = nullptr;  // per bordered tile: the source regions of k_pyramid (orb_setup_geometry)
    // k_fast_harris: per-tile record slots + line counts in device memory, arrival counter per tile row (self re-arming)
    void* d_fh_slots = nullptr;
    void* d_fh_line = nullptr;
    int32_t* d_fh_arrive = nullptr;
    std::vector<DevCandidate> last_cand;  // canonical candidate list of the last detection (debug getter)
    DevDescKp* d_kp = nullptr;
    uint8_t* d_desc = nullptr;      // descriptors of the current extraction (one of the two halves below)
    uint8_t* d_desc_buf = nullptr;  // 2 x kp_cap x 32: ping-pong so that frame i-1 survives frame i
    int desc_flip = 0;
    int kp_cap = 0;
    // pinned host staging
    uint8_t* h_pin = nullptr;
    size_t h_pin_cap = 0;
    hipEvent_t ev = nullptr;
    // --- matcher
    uint8_t *d_mq = nullptr, *d_mt = nullptr;
    float *d_mqxy = nullptr, *d_mtxy = nullptr;
    int32_t* d_mout = nullptr;
    int32_t* d_marrive = nullptr;  // k_knn2 arrival counters (inside the d_mout allocation)
    int m_cap_q = 0, m_cap_t = 0;
    // --- tracking rows
    mvo_track_state* track = nullptr;
    // --- BA diagnostics of the last fetched solve
    long long ba_phase[16] = {0};
    int ba_wgs = 0, ba_trials = 0;
    int ba_throughput_mode = 0;  // mvo_ba_set_mode: 0 = latency (default), 1 = throughput (fewer, fuller workgroups per window)
    bool ba_never_resident = false;  // MVO_BA_MODE_SHARED: throughput cut, launch path only
    struct mvo_ba_pool* ba_pool = nullptr;  // pooled BA workspaces (ba_host.cpp)
    // --- profiling
    bool prof = false;
    std::map<std::string, ProfEntry> prof_acc;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
};

int mvo_set_err(mvo_ctx* c, int code, const char* what, hipError_t e);
#define MVO_HIP(call)                                                            \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) {                                                 \
            (void)hipGetLastError(); /* clear the sticky error */                \
            return mvo_set_err(ctx, MVO_ERR_HIP, #call, e__);                    \
        }                                                                        \
    } while (0)

// profiling brackets
void mvo_prof_begin(mvo_ctx* c, const char* name);
void mvo_prof_end(mvo_ctx* c);
void mvo_prof_collect(mvo_ctx* c);
struct ProfScope {
    mvo_ctx* c;
    ProfScope(mvo_ctx* c_, const char* name) : c(c_) {
        if (c->prof) mvo_prof_begin(c, name);
    }
    ~ProfScope() {
        if (c->prof) mvo_prof_end(c);
    }
};

int mvo_ensure_pinned(mvo_ctx* ctx, size_t bytes);

// orb_kernels.hip
int orb_launch_pyramid(mvo_ctx* ctx, const uint8_t* d_img, int stride, int channels, int nlevels);
int orb_launch_detect(mvo_ctx* ctx, uint8_t* host, bool ordered);
int orb_launch_blur(mvo_ctx* ctx, int nlevels);
bool orb_brief_from_levels(const mvo_ctx* ctx);
int orb_launch_brief(mvo_ctx* ctx, int n, const DevDescKp* kps, uint8_t* desc_host);
// match_kernels.hip
int match_launch_knn2(mvo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_out,
                      int32_t* final_out);
int match_launch_radius_l1(mvo_ctx* ctx, const uint8_t* d_q, const float* d_qxy, int nq, const uint8_t* d_t,
                           const float* d_txy, int nt, float max_px, int32_t* d_out);
// track_kernels.hip
int track_launch_map_in_view(mvo_ctx* ctx, const float* d_pos, const uint8_t* d_desc, int n, const TrackViewArgs& a,
                             int32_t* d_idx, float* d_px, uint8_t* d_desc_out, int32_t* d_n);
int track_launch_pnp_hypotheses(mvo_ctx* ctx, const float* d_p3, const float* d_p2, int n, const int32_t* d_subsets,
                                int n_hyp, const TrackCamera& cam, float thr2, double* d_models, int32_t* d_counts,
                                uint8_t* d_masks, double* h_models, int32_t* h_counts);
int track_launch_pnp_refine(mvo_ctx* ctx, const float* d_p3, const float* d_p2, const uint8_t* d_masks, int n,
                            const TrackCamera& cam, const double* d_models, const int32_t* d_counts, int n_hyp,
                            double confidence, int forced_best, int mode, double* d_Mg, double* d_mg,
                            uint8_t* d_best_mask, double* d_out);
int track_launch_triangulate(mvo_ctx* ctx, const float* d_kp1, const float* d_kp2, int n, const TrackCamera& cam,
                             const double* R, const double* t, float* d_prev, float* d_curr);
int track_launch_em_hypotheses(mvo_ctx* ctx, const double* d_q1, const double* d_q2, int n, const int32_t* d_subsets,
                               int n_hyp, float thr2, double* d_E, int32_t* d_nm, int32_t* d_counts);
int track_launch_em_mask(mvo_ctx* ctx, const double* d_q1, const double* d_q2, int n, const double* d_E, float thr2,
                         uint8_t* d_mask);
extern int g_pyr_force_chain;  // test hook (orb_kernels.hip)
extern int g_match_mfma;       // test hook (match_kernels.hip)
extern int g_pnp_replay_skew;  // test hook: the device replays the RANSAC loop with a wrong confidence
// track_host.cpp
void track_release(mvo_ctx* ctx);
// ba_host.cpp (planning, pooled workspaces, launch service) + ba_kernels.hip (k_ba_lm)
struct mvo_ba_handle;
int ba_solve_device(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st);
int ba_begin_device(mvo_ctx* ctx, const mvo_ba_problem* p);
int ba_end_device(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st);
int ba_solve_batch_device(mvo_ctx* ctx, mvo_ba_problem* ps, int n, mvo_ba_stats* sts);
int ba_prepare_device(mvo_ctx* ctx, const mvo_ba_problem* p, mvo_ba_handle** out);
int ba_run_device(mvo_ctx* ctx, mvo_ba_handle* H);
int ba_fetch_device(mvo_ctx* ctx, mvo_ba_handle* H, double* poses, double* points, mvo_ba_stats* st);
void ba_release_device(mvo_ctx* ctx, mvo_ba_handle* H);
void ba_pool_release(mvo_ctx* ctx);
void ba_set_trace(mvo_ctx* ctx, int on);
int ba_get_trace(mvo_ctx* ctx, mvo_ba_handle* H, double* rows, int cap, int* n);
int ba_get_plan(mvo_ctx* ctx, mvo_ba_handle* H, int* G, int* nsplit, int32_t* wg_pt, int cap);
void ba_launch_stats(int device, long long* launches, long long* windows, double* ms, int reset);
void ba_service_times(int device, double* out5);
void ba_service_park(int device);
void ba_resident_stats(int device, long long* windows, long long* grid_starts, double* cycles = nullptr, long long* path_switches = nullptr);
// hipFree / hipHostFree with the resident solver grid of `device` taken off first (both synchronise with every stream of the
// device; the grid never ends on its own)
void ba_service_free(int device, void* p, bool host);
inline void mvo_free_on_current_device(void* p) {
    int d = 0;
    (void)hipGetDevice(&d);
    ba_service_free(d, p, false);
}

#endif
