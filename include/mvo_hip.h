/* include/mvo_hip.h -- C-ABI of libmvo_hip.so: the MI355X (gfx950) implementation of the per-frame
 * hot path of felixchenfy/Monocular-Visual-Odometry (ORB extract + grid sampling, Hamming/L1 descriptor
 * matching with the reference's filters, sliding-window bundle adjustment).
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference repo).
 * Conventions: plain pointers and sizes, caller-allocated buffers, int status (0 = MVO_OK, <0 = error),
 * never throws, never prints; one mvo_ctx per host thread, one HIP stream per ctx.  Host pointers unless
 * the name ends in _dev.  There is NO CPU fallback: without a usable HIP device mvo_create fails with
 * MVO_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef MVO_HIP_H
#define MVO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_OK 0
#define MVO_ERR_INVALID (-1)   /* bad argument (e.g. wrong method index: feature_match.cpp:225 throws) */
#define MVO_ERR_NO_DEVICE (-2) /* no HIP device / kernels not loadable */
#define MVO_ERR_CAPACITY (-3)  /* caller buffer or internal candidate buffer too small */
#define MVO_ERR_HIP (-4)       /* HIP runtime error, see mvo_last_error */
#define MVO_ERR_STATE (-5)     /* call order violated (e.g. reuse_pyramid without a pyramid) */

typedef struct mvo_ctx mvo_ctx;

/* cv::KeyPoint, 28 bytes (Frame::keypoints_, include/my_slam/vo/frame.h:32). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} mvo_keypoint;

/* cv::DMatch, 16 bytes (Frame::matches_with_ref_, frame.h:38). */
typedef struct {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} mvo_dmatch;

/* The values the reference latches from config/config.yaml:65-69,94-95 in function-local statics
 * (src/geometry/feature_match.cpp:16-19, 42-45, 56-59). */
typedef struct {
    int32_t nfeatures;         /* number_of_keypoints_to_extract */
    float scale_factor;        /* scale_factor */
    int32_t nlevels;           /* level_pyramid (1..8) */
    int32_t fast_threshold;    /* score_threshold */
    int32_t max_keypoints;     /* max_number_of_keypoints */
    int32_t grid_size;         /* kpts_uniform_selection_grid_size */
    int32_t grid_max_per_cell; /* kpts_uniform_selection_max_pts_per_grid */
    /* How cv::ORB resamples its pyramid -- a property of the OpenCV version, not of config.yaml: 1 = INTER_LINEAR_EXACT
     * (OpenCV >= 3.4, what the reference's README asks for: bit-exact 8.8 fixed point), 0 = INTER_LINEAR (older
     * OpenCV: 11-bit coefficients from float coordinates, truncating vertical pass). */
    int32_t pyramid_interpolation;
} mvo_orb_params;

/* ---- context ------------------------------------------------------------------------------- */
int mvo_create(mvo_ctx** ctx, int device);
/* A second context for the SAME host thread that works on the parent's HIP stream instead of a stream of its own: own
 * workspaces, error state and mode, no further hardware queue.  For callers that give one sequence two contexts so that two
 * of its calls can be in progress at once (the bundle adjustment of frame i between its _begin and _end, the extraction of
 * frame i+1 meanwhile).  The HIP runtime spreads its streams over a fixed number of hardware queues in the order of their
 * creation; with two streams per sequence the extraction streams of 24 sequences ended up three to a queue on half of the
 * queues (measured: 3443 / 3551 -> 3874 / 3910 frames/s with sibling contexts).  Destroy the sibling before its parent. */
int mvo_create_sibling(mvo_ctx* parent, mvo_ctx** ctx);
void mvo_destroy(mvo_ctx* ctx);
/* A number that identifies this ctx for the lifetime of the process (never reused, unlike its address): what per-ctx state
 * kept OUTSIDE the library (e.g. "parameters already latched", feature_match.cpp:16-19) is keyed by.  0 for a null ctx. */
unsigned long long mvo_ctx_uid(const mvo_ctx* ctx);
const char* mvo_last_error(const mvo_ctx* ctx);
/* Blocks until all work queued on the ctx stream is done. */
int mvo_synchronize(mvo_ctx* ctx);
/* How the host threads of this process wait for `device` (hipSetDeviceFlags): 0 auto (the runtime's choice: spinning while
 * there are enough cores), 1 spin, 2 yield, 3 block on an interrupt.  A process whose waiting threads outnumber its CPUs
 * (one rank of a multi-GPU node confined to its share of the cores, 24 sequence threads each) should yield or block.  No
 * reference counterpart (OpenCV / g2o do not wait for a device). */
/* (The calling thread's current HIP device is left as it was.) */
#define MVO_WAIT_AUTO 0
#define MVO_WAIT_SPIN 1
#define MVO_WAIT_YIELD 2
#define MVO_WAIT_BLOCK 3
int mvo_set_wait_policy(int device, int policy);
/* Replaces basics::Config::get<...> latching in feature_match.cpp:16-19,42-45,56-59.  Also resets the
 * latched grid dimensions (feature_match.cpp:59-62 latches rows/cols from the FIRST image). */
int mvo_orb_configure(mvo_ctx* ctx, const mvo_orb_params* params);

/* ---- extraction ---------------------------------------------------------------------------- */
/* geometry::calcKeyPoints (src/geometry/feature_match.cpp:11-36; called from Frame::calcKeyPoints,
 * include/my_slam/vo/frame.h:73-76): cv::ORB::detect + selectUniformKptsByGrid.  image: u8, `channels`
 * = 1 (gray) or 3/4 (BGR[A] as cv::imread gives, run_vo.cpp:114).  The device pyramid built here stays
 * cached in the ctx for mvo_calc_descriptors(reuse_pyramid=1). */
int mvo_calc_keypoints(mvo_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                       int channels, mvo_keypoint* kps, int cap, int* n);
/* Same with the image already resident in HBM (bench: inputs resident when the timed region starts). */
int mvo_calc_keypoints_dev(mvo_ctx* ctx, const void* d_image, int width, int height, int stride,
                           int channels, mvo_keypoint* kps, int cap, int* n);
/* geometry::calcDescriptors (feature_match.cpp:38-49; Frame::calcDescriptors, frame.h:77-86):
 * cv::ORB::compute.  May DROP keypoints (feature_match.h:15-17): kps/n are in/out.  desc: n*32 bytes.
 * rgb (optional): n*3 bytes, the per-keypoint colour of frame.h:80-85 / basics::getPixelAt
 * (src/basics/opencv_funcs.cpp:10-32).  reuse_pyramid != 0: `image` must be the image of the last
 * mvo_calc_keypoints[_dev] call on this ctx (the pyramid is NOT rebuilt; image is still read for rgb
 * and may be NULL if rgb is NULL). */
int mvo_calc_descriptors(mvo_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                         int channels, int reuse_pyramid, mvo_keypoint* kps, int* n, uint8_t* desc,
                         uint8_t* rgb);
/* Descriptors stay on the device as well (d_desc_out receives a device pointer owned by the ctx, valid
 * until the extraction after the next one on this ctx: two buffers alternate) so the matcher can consume them without a PCIe round trip. */
int mvo_calc_descriptors_dev(mvo_ctx* ctx, mvo_keypoint* kps, int* n, uint8_t* desc,
                             const void** d_desc_out);
/* geometry::selectUniformKptsByGrid (feature_match.cpp:51-84), host-side, in place. grid dims are the
 * latched ones (first call: rows = image_rows / grid_size, cols = image_cols / grid_size). */
int mvo_select_uniform_kpts_by_grid(mvo_ctx* ctx, mvo_keypoint* kps, int* n, int image_rows,
                                    int image_cols);

/* ---- matching ------------------------------------------------------------------------------ */
/* cv::BFMatcher("BruteForce-Hamming")::knnMatch(k=2) as used at feature_match.cpp:203-208: exact
 * 2-NN, equal distances keep the lower train index first.  idx/dist: nq x 2 int32; a missing
 * neighbour is (-1, INT32_MAX). */
int mvo_match_knn2(mvo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                   int32_t* dist);
int mvo_match_knn2_dev(mvo_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, int32_t* idx,
                       int32_t* dist);
/* geometry::matchByRadiusAndBruteForce (feature_match.cpp:86-124): per query the first minimum of
 * sum|a-b| over the 32 descriptor bytes among trains within max_px pixels; idx -1 if none.
 * `sum` is 32x the reference's mean-abs-difference. qxy/txy: n x 2 float (KeyPoint::pt). */
int mvo_match_radius_l1(mvo_ctx* ctx, const uint8_t* q, const float* qxy, int nq, const uint8_t* t,
                        const float* txy, int nt, float max_px, int32_t* idx, int32_t* sum);
/* geometry::matchFeatures (feature_match.cpp:126-239; callers vo_addFrame.cpp:42,99, vo.cpp:283).
 * method 1: exact 1-NN (replaces cv::FlannBasedMatcher(LshIndexParams(5,10,2)), which is approximate
 * and RNG-seeded) + d < max(min_d * xiang_gao_ratio, 30); method 2: Lowe ratio d0 < lowe_ratio * d1;
 * method 3: radius-gated L1 + the method-1 threshold; anything else MVO_ERR_INVALID.  Always followed
 * by removeDuplicatedMatches.  The ratios are passed as the reference latches them: get<int> at
 * feature_match.cpp:137-139 turns 0.8 into 1. */
int mvo_match_features(mvo_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int method,
                       double xiang_gao_ratio, double lowe_ratio, const float* xy1, const float* xy2,
                       float max_px, mvo_dmatch* out, int cap, int* n);
/* matchFeatures methods 1 / 2 with both descriptor sets already in HBM (device pointers, e.g. the ones
 * mvo_calc_descriptors_dev returns for frame i-1 and frame i). */
int mvo_match_features_dev(mvo_ctx* ctx, const void* d_d1, int n1, const void* d_d2, int n2, int method,
                           double xiang_gao_ratio, double lowe_ratio, mvo_dmatch* out, int cap, int* n);
/* geometry::removeDuplicatedMatches (feature_match.cpp:241-260), host-side, in place. */
int mvo_remove_duplicated_matches(mvo_dmatch* m, int* n);

/* ---- bundle adjustment --------------------------------------------------------------------- */
/* The graph optimization::bundleAdjustment builds (src/optimization/g2o_ba.cpp:172-317), flattened:
 * pose vertices 0..F-1 (VertexSE3Expmap), point vertices (VertexSBAPointXYZ, marginalized), one
 * EdgeProjectXYZ2UV + RobustKernelHuber per observation, CameraParameters(f, (cx, cy), 0). */
typedef struct {
    int32_t n_poses, n_points, n_edges;
    double* pose_T_w_c;        /* n_poses x 16 row-major 4x4 cam->world (Frame::T_w_c_), in/out */
    double* points;            /* n_points x 3, in/out (the adapter stores them back as f32, :308-316) */
    const int32_t* edge_pose;  /* n_edges: index into poses */
    const int32_t* edge_point; /* n_edges: index into points */
    const double* edge_uv;     /* n_edges x 2 measured pixel */
    double focal, cx, cy;      /* K(0,0), K(0,2), K(1,2) -- fy is ignored by the reference (:219-222) */
    double info[4];            /* information_matrix (config.yaml:122) */
    double huber_delta;        /* g2o::RobustKernelHuber default delta = 1 */
    int32_t fix_points;        /* is_fix_map_pts */
    const uint8_t* pose_fixed; /* optional per-pose setFixed flags; NULL = none (as the reference) */
    int32_t max_iterations;    /* optimizer.optimize(50) */
} mvo_ba_problem;

typedef struct {
    int32_t iterations, trials, terminated;
    int32_t failed_solves;     /* trials whose reduced system was "not positive" for Eigen::LDLT (g2o: solve() returns false) */
    double chi2_initial, chi2_final, lambda_final;
    int32_t stale_steps;       /* of those: g2o applied the solver's previous x and accepted it (negative predicted decrease) */
    int32_t reserved_;
} mvo_ba_stats;

/* optimization::bundleAdjustment (g2o_ba.cpp:172-317; caller VisualOdometry::callBundleAdjustment_,
 * src/vo/vo.cpp:458-462).  The solver stack g2o_ba.cpp:193-200 builds is followed to the letter: Levenberg-Marquardt with
 * g2o's lambda / nu rules, Schur complement on the points, the reduced system through Eigen::LDLT's pivot order and sign rule
 * (LinearSolverDense), and -- as OptimizationAlgorithmLevenberg::solve does -- a FAILED linear solve still applies the solver's
 * previous x and scores it with chi2 = DBL_MAX (stats->failed_solves / ->stale_steps). */
int mvo_bundle_adjustment(mvo_ctx* ctx, mvo_ba_problem* problem, mvo_ba_stats* stats);
/* How this ctx shares the GPU (no reference counterpart: g2o and cv::ORB are single-threaded).
 * LATENCY (default): one sequence wants its frame back as fast as possible -- a window is cut into ~300 observations per
 * workgroup (the 5-keyframe window of the benchmark: 28 CUs of one XCD, shortest solve) and solved by a launch of its own;
 * the detection kernel also puts its candidates in order on the device.
 * THROUGHPUT: many sequences are in flight on this GPU -- CU time counts, not latency.  While the offered load keeps it
 * busy (128 submissions in a row at a rate x solve time of >= 8 of its 16 slots; it leaves after 80 ms below 7), 5-keyframe windows are cut into ~670 observations
 * per workgroup (14 CUs) and go to the resident solver service: a grid that stays on the device (2 x 14 CUs of every XCD)
 * and pulls windows from pinned mailboxes, no launch per window.  With less load the windows take the LATENCY cut on the
 * launch path and the CUs stay with whoever has work.  Detection leaves the interleaving of a tile row's candidates to the
 * calling thread (~75 us of host time per frame instead of ~8 us of kernel time), and descriptors are sampled from whole
 * blurred pyramid levels (one blur of every level behind the detection kernel instead of one blur per keypoint window: a fifth
 * of the instructions, one launch more).
 * The summation order of a solve (hence the last bits of its result) follows the cut; every cut is deterministic and is what
 * mvo_debug_get_ba_plan reports.  Results of the extraction do not depend on the mode.
 * SHARED: many sequences are in flight, but the bundle adjustment is not what their frames mostly wait for (tracking rows --
 * map points in view, solvePnPRansac -- or other stages in the loop: ~2500 windows/s on this GPU instead of ~5000).  Windows keep the
 * THROUGHPUT cut but always take the launch path (1...16 windows per grid): a resident grid would hold 208 CUs for slots that
 * are half empty (tracking rows in the loop: 1800 frames/s with the grid, 2470 without).  The caller knows its loop; the library's
 * load estimate only sees submission times and cannot tell a saturated launch path from a loop that is busy elsewhere. */
#define MVO_BA_MODE_LATENCY 0
#define MVO_BA_MODE_THROUGHPUT 1
#define MVO_BA_MODE_SHARED 2
int mvo_ba_set_mode(mvo_ctx* ctx, int mode);
/* Admission gate of the extraction / matching kernels (no reference counterpart).  Contexts in THROUGHPUT or SHARED mode pass a
 * process-wide, per-device counting gate around every launch-and-wait section of mvo_calc_keypoints*, mvo_calc_descriptors* and
 * the matchers: at most `n` such sections are in flight on a device at once (default 8; 0 = no limit).  With 24-32 sequences
 * next to the resident solver grid the frames' kernels share the few CUs the grid leaves; more than ~8 frames interleaving
 * there LOWER their combined throughput (14 in flight: 3400 frames/s of extraction, 8-9: 4600; values 4...12 are within 4 % of
 * each other).  LATENCY-mode contexts never wait at the gate.  Returns the previous value. */
int mvo_set_extract_concurrency(int n);

/* ---- operating knobs, in one place -------------------------------------------------------------------------------------
 * What a caller chooses (API):
 *   mvo_ba_set_mode(ctx, LATENCY | THROUGHPUT | SHARED)   how this ctx shares the GPU (above)
 *   mvo_set_extract_concurrency(n)                        admission gate of the frame kernels (above)
 *   mvo_set_wait_policy(device, AUTO|SPIN|YIELD|BLOCK)    how host threads wait for the device
 *   mvo_orb_params.pyramid_interpolation                  which OpenCV resampling the pyramid restates
 * Environment variables, read once at start-up -- defaults in [] -- for A/B measurements and development; results never
 * depend on them except through the summation plan of a BA window, which mvo_debug_get_ba_plan always reports:
 *   MVO_EXTRACT_CONCURRENCY [8]   start-up value of mvo_set_extract_concurrency
 *   MVO_BA_SERVICE [1]            resident solver service: 0 never, 1 by offered load (mvo_ba_set_mode), 2 always
 *   MVO_BA_XCD_RESERVE [4]        CUs per XCD a BA window leaves to other kernels (28 workgroups of a LATENCY window, 2 x 14 of THROUGHPUT windows)
 *   MVO_BA_WGS, MVO_BA_NSPLIT     force the workgroups / Schur column pieces of a window (= the debug keys ba_wgs, ...)
 *   MVO_BA_CU_SHARE [all]         CUs one launch-path grid may take
 *   MVO_BA_GROUPS [1], MVO_BA_ALIAS_SL [1], MVO_BA_BLOCK_SOLVER [0]   A/B switches of DESIGN.md 4.3 (grouped Schur exchange,
 *                                 reduced system inside the U area, block LDL^T for the 5-pose class)
 *   MVO_BRIEF_LEVEL_BLUR [by mode] 1 = descriptors from whole blurred levels (k_blur + k_brief_sample), 0 = per-keypoint windows
 *   MVO_BA_UV_GLOBAL [1]          0 = the measurements of a > 512-observation range never leave LDS (1: device memory when LDS would cost a chunk)
 *   MVO_PNP_CHUNK [by mode]       0 = solvePnPRansac always evaluates all hypotheses at once, n = the first n for every ctx
 *                                 (default: the first 32 for a THROUGHPUT / SHARED ctx whose previous RANSAC loop was short)
 *   MVO_BRIEF_WAVES [4], MVO_MATCH_SLICE [256], MVO_PYR_FULL_POOL [0], MVO_PNP_OCC [by mode]   kernel shape A/Bs
 *   MVO_BA_PLAN_TRACE, MVO_HOST_TIMING   development output on stderr
 * mvo_debug_set(key, value) (bottom of this file) sets the same switches at run time for the tests. */

/* The same call in two halves, so that the host thread can do other work (e.g. extract the next frame on another
 * ctx) while the window is being solved: _begin builds the window, uploads it and queues the launch, _end blocks
 * until it is done and writes poses / points / stats back like mvo_bundle_adjustment.  One solve in flight per ctx;
 * `problem` must stay valid between the two calls. */
int mvo_bundle_adjustment_begin(mvo_ctx* ctx, const mvo_ba_problem* problem);
int mvo_bundle_adjustment_end(mvo_ctx* ctx, mvo_ba_problem* problem, mvo_ba_stats* stats);
/* n independent windows (e.g. the windows of n sequences) solved in one grid (8 windows x 32 workgroups fill the 256
 * CUs; more than 8 are split over consecutive launches).  Same results as n mvo_bundle_adjustment calls.
 * Concurrency note: every BA launch of the process is issued by one service thread per device, which also batches
 * windows submitted by different ctx / host threads at the same time -- callers need no co-ordination. */
int mvo_ba_solve_batch(mvo_ctx* ctx, mvo_ba_problem* problems, int n, mvo_ba_stats* stats);

/* The same solve with the window resident in HBM: mvo_ba_prepare uploads the graph once (inputs + the
 * pose / point adjacency the kernels need), mvo_ba_solve_resident runs one full optimize(50) from the
 * resident initial state (asynchronous on the ctx stream; may be repeated), mvo_ba_fetch copies the
 * refined poses (n_poses x 16) / points (n_points x 3, untouched when fix_points) / stats back. */
typedef struct mvo_ba_handle mvo_ba_handle;
int mvo_ba_prepare(mvo_ctx* ctx, const mvo_ba_problem* problem, mvo_ba_handle** handle);
int mvo_ba_solve_resident(mvo_ctx* ctx, mvo_ba_handle* handle);
int mvo_ba_fetch(mvo_ctx* ctx, mvo_ba_handle* handle, double* poses, double* points, mvo_ba_stats* stats);
void mvo_ba_release(mvo_ctx* ctx, mvo_ba_handle* handle);

/* ---- tracking: the steps between matching and bundle adjustment -------- */
/* The map (vo::Map::map_points_, include/my_slam/vo/map.h:19: MapPoint::pos_ + descriptor_) kept resident in
 * HBM so that map descriptors never cross PCIe between frames.  The order of the uploaded arrays is the
 * host's iteration order of map_points_; indices returned below refer to it. */
typedef struct mvo_map mvo_map;
int mvo_map_create(mvo_ctx* ctx, mvo_map** map);
void mvo_map_release(mvo_ctx* ctx, mvo_map* map);
/* Replaces the device copy: pos n x 3 float (cv::Point3f), desc n x 32 bytes. */
int mvo_map_upload(mvo_ctx* ctx, mvo_map* map, const float* pos, const uint8_t* desc, int n);
/* Positions [first, first+n) only -- what bundleAdjustment writes back (g2o_ba.cpp:306-316). */
int mvo_map_update_positions(mvo_ctx* ctx, mvo_map* map, const float* pos, int first, int n);
/* VisualOdometry::getMappointsInCurrentView_ (src/vo/vo.cpp:16-49): the map points with p_cam.z >= 0 whose
 * pixel lies strictly inside the cols x rows image, in map order.  T_w_c: 4x4 row-major double (inverted
 * on the host like cv::Mat::inv()).  idx / px (n x 2, cv::Point2f) are host outputs of capacity cap;
 * *d_desc_out receives a device pointer (owned by the ctx, valid until the next call) to the gathered
 * n x 32 descriptors = `corresponding_mappoints_descriptors`, ready for mvo_match_features_dev. */
int mvo_map_points_in_view(mvo_ctx* ctx, mvo_map* map, const double* T_w_c, double fx, double fy, double cx,
                           double cy, int cols, int rows, int32_t* idx, float* px, int cap, int* n,
                           const void** d_desc_out);
/* cv::solvePnPRansac(pts_3d, pts_2d, K, cv::Mat(), R_vec, t, false, iterations, reprojection_error,
 * confidence, inliers) as called at src/vo/vo.cpp:326-329 (flags = SOLVEPNP_ITERATIVE): RANSAC over 5-point
 * EPnP hypotheses with the subsets cv::RNG((uint64)-1) draws, then DLT + Levenberg-Marquardt on the inliers.
 * pts3d n x 3 float, pts2d n x 2 float.  *found = 0 when no hypothesis reaches 5 inliers (solvePnPRansac
 * returns false) or n < 5; inliers (ascending indices into the pairs) needs capacity cap >= n. */
int mvo_solve_pnp_ransac(mvo_ctx* ctx, const float* pts3d, const float* pts2d, int n, double fx, double fy,
                         double cx, double cy, int iterations, float reprojection_error, double confidence,
                         double* rvec, double* tvec, int32_t* inliers, int cap, int* n_inliers, int* found);
/* ---- keyframe insertion: epipolar inlier filter, triangulation, culling ---- */
/* geometry::helperTriangulatePoints (src/geometry/motion_estimation.cpp:214-247, called at
 * vo_addFrame.cpp:114-116): pixel2CamNormPlane on the matched pixels of the previous / current keyframe
 * (n x 2 float each, KeyPoint::pt), cv::triangulatePoints with [I|0] and [R|t] = T_curr_to_prev, then
 * basics::transCoord.  Outputs n x 3 float (either may be NULL): the points in the previous camera frame
 * (doTriangulation, epipolar_geometry.cpp:130-175) and what helperTriangulatePoints returns. */
int mvo_triangulate_points(mvo_ctx* ctx, const float* kp_prev, const float* kp_curr, int n, double fx, double fy,
                           double cx, double cy, const double* R, const double* t, float* pts3d_in_prev,
                           float* pts3d_in_curr);
/* geometry::helperFindInlierMatchesByEpipolarCons (src/geometry/motion_estimation.cpp:182-198, called at
 * vo_addFrame.cpp:104-106) = the inlier mask of cv::findEssentialMat(pts1, pts2, focal = (fx + fy) / 2,
 * pp = Point2f(cx, cy), cv::RANSAC, prob, threshold) (epipolar_geometry.cpp:17-47): RANSAC (cv::RNG subsets, at
 * most 1000 iterations, adaptive count) over five-point hypotheses scored by the Sampson distance.  kp_prev /
 * kp_curr: the matched pixels (n x 2 float); inliers: ascending indices into the matches, capacity cap >= n. */
int mvo_find_essential_inliers(mvo_ctx* ctx, const float* kp_prev, const float* kp_curr, int n, double fx, double fy,
                               double cx, double cy, double prob, double threshold, int32_t* inliers, int cap,
                               int* n_inliers);
/* VisualOdometry::retainGoodTriangulationResult_ (src/vo/vo.cpp:181-244), host-side (acos + a sort for the
 * median): keep[i] lists the points whose triangulation angle (degrees) is >= min_triang_angle and at most
 * max_ratio_to_median times the median; angles (n, may be NULL) receives every angle. */
int mvo_retain_good_triangulation(const float* pts3d_in_curr, int n, const double* T_w_c_curr, const double* T_w_c_ref,
                                  double min_triang_angle, double max_ratio_to_median, int32_t* keep, int* n_keep,
                                  double* angles);
/* cv::Rodrigues(rvec -> R, 3x3 row-major) as used at vo.cpp:334; host-side. */
int mvo_rodrigues(const double* rvec, double* R);
/* cv::Mat::inv() of a 4x4 double matrix (LU with partial pivoting) -- the T_w_c <-> T_c_w flips at
 * vo.cpp:31,349; host-side.  MVO_ERR_INVALID if singular. */
int mvo_invert_pose(const double* T, double* T_inv);

/* ---- measurement --------------------------------------------------------------------------- */
/* BA launches issued on `device` by this process since the last reset: number of launches, windows solved by them,
 * and the sum of the launch durations (HIP events on the launch stream, milliseconds). */
int mvo_ba_launch_stats(int device, long long* launches, long long* windows, double* ms, int reset);
/* Wall-clock (ms, since the last reset of mvo_ba_launch_stats) the launch thread spent: [0] waiting for work, [1] waiting
 * for a full batch, [2] building + issuing launches, [3] waiting for the kernels, [4] publishing results. */
int mvo_debug_ba_service_times(int device, double* out5);
/* Resident solver service (throughput mode): windows it solved and how often the resident grid was put on the device since
 * the last reset of mvo_ba_launch_stats (those windows count as one launch each there, `ms` = their solve times on the
 * device clock). */
int mvo_debug_ba_resident_stats(int device, long long* windows, long long* grid_starts);
/* Shader-clock cycles the resident grid spent on those windows (sum over the windows, workgroup 0 of each, load to write-back;
 * divided by their `ms` this is the clock the solver ran at under load). */
int mvo_debug_ba_resident_cycles(int device, double* shader_cycles);
/* Times a window that cannot use the resident grid (pose-only, another solver class, rows in LDS only) made the grid leave
 * since the last reset: every such switch drains the 16 slots and relaunches the grid afterwards -- mixed workloads pay for it. */
int mvo_debug_ba_path_switches(int device, long long* n);
/* Test hook: replays the resident solver service's demand estimate (see mvo_ba_set_mode) over n submission times (seconds,
 * ascending); decisions[i] = 1 if a window submitted at times[i] would go to the resident grid.  Returns the number of
 * changes of mind.  Touches no device. */
int mvo_debug_ba_demand_replay(const double* times, int n, uint8_t* decisions);
/* When enabled every kernel launch is bracketed by hipEvents on the ctx stream; times accumulate per
 * kernel name until reset. */
int mvo_profile_enable(mvo_ctx* ctx, int on);
int mvo_profile_reset(mvo_ctx* ctx);
/* Returns the number of distinct kernels; fills up to cap entries. */
typedef struct {
    char name[48];
    int64_t launches;
    double total_ms;
} mvo_kernel_time;
int mvo_profile_get(mvo_ctx* ctx, mvo_kernel_time* out, int cap);

/* ---- debug / test hooks (used by tests/ to localise a parity failure; not part of the drop-in) ---- */
/* key "ba_mfma": 1 (default) = matrix-core contractions, 0 = plain VALU loops computing the same sums;
 * key "ba_wgs": workgroups (CUs) one BA window is split over, 0 (default) = automatic;
 * key "pnp_replay_skew": 1 = the device replays the RANSAC loop with a wrong confidence, so that the host's
 * verification of the selected hypothesis has something to correct (tests only). */
int mvo_debug_set(const char* key, int value);
/* LM trace of the last solve on this ctx (handle == NULL: mvo_bundle_adjustment; else that resident window), after
 * mvo_debug_ba_trace_enable(ctx, 1): one row {lambda the trial was solved with, robust chi2 it reached, gain ratio,
 * accepted} per trial. */
int mvo_debug_ba_trace_enable(mvo_ctx* ctx, int on);
int mvo_debug_get_ba_trace(mvo_ctx* ctx, mvo_ba_handle* handle, double* rows, int cap, int* n);
/* Summation plan of that window: number of landmark ranges (workgroups), column splits of a Schur chain, and the
 * first landmark of every range (wgs + 1 entries) -- what the oracle needs to restate the device sums bit for bit.
 * nsplit: bits 0..15 = the column splits; bits 16.. = K when the window's Schur exchange is grouped (windows of more than
 * 32 workgroups add the partials of the workgroups g = k mod K first -- one XCD each --, then the K group sums), else 0. */
int mvo_debug_get_ba_plan(mvo_ctx* ctx, mvo_ba_handle* handle, int* wgs, int* nsplit, int32_t* wg_pt_start, int cap);
/* Shader-clock cycles the last fetched BA solve spent per phase (ids in csrc/ba_kernels.hip), and the number
 * of workgroups it ran on. */
int mvo_debug_get_ba_phases(mvo_ctx* ctx, long long* cycles, int n, int* wgs);
/* Copies cached pyramid level `level` (raw gray or blurred) WITH its 32-px frame: (h+64) rows of `stride`
 * bytes.  out == NULL only queries the geometry. */
int mvo_debug_get_level(mvo_ctx* ctx, int level, int blurred, uint8_t* out, int cap, int* w, int* h, int* stride);
/* FAST+NMS survivors of the last detection in canonical (level,row,col) order as 16-byte records
 * {int16 x, y; int32 level<<16|fast_score; float harris; float angle}. */
int mvo_debug_get_candidates(mvo_ctx* ctx, void* out, int cap, int* n);

/* Record of the last mvo_solve_pnp_ransac on this ctx: models (iterations x 12: R row-major, t) and inlier
 * counts of every hypothesis, info[6] = {best iteration, iterations the sequential loop would have run,
 * DLT used, LM iterations, LM residual evaluations, hypotheses evaluated (negative when the host had to correct
 * the device's choice of hypothesis)}. */
int mvo_debug_get_pnp(mvo_ctx* ctx, double* models, int32_t* counts, int cap, int32_t* info);

/* Record of the last mvo_find_essential_inliers on this ctx: inlier counts of every evaluated hypothesis
 * (iterations x 10 candidates, -1 = no such candidate), info[5] = {best iteration, best candidate, iterations
 * the sequential loop ran, iterations evaluated, 0}.  Returns the number of evaluated iterations. */
int mvo_debug_get_essential(mvo_ctx* ctx, int32_t* counts, int cap_iters, int32_t* info);

#ifdef __cplusplus
}
#endif
#endif
