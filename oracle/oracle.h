/* oracle/oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory is a scalar, dependency-free CPU restatement of the
 * reference's per-frame hot path (ORB extract + grid sampling, Hamming / L1 matching + filters,
 * sliding-window bundle adjustment).  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load it -- as the checker, never as the thing measured or shipped.  The product
 * (monocular-visual-odometry_amd/) never includes, links or calls anything in here.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in OpenCV (cv::ORB, cv::BFMatcher,
 * cv::FlannBasedMatcher) and g2o, neither of which is vendored under /root/reference nor
 * installed in this image, and the reference's own tests hold no golden vectors for it
 * (SURVEY.md section 4 / 8c).  The oracle therefore restates the published upstream algorithms
 *  and is anchored on the reference's call sites:
 *   src/geometry/feature_match.cpp:11-260, include/my_slam/vo/frame.h:73-86,
 *   src/optimization/g2o_ba.cpp:172-317, src/vo/vo.cpp:384-478, config/config.yaml:63-123.
 * It is checked against analytic known-answer cases in tests/ (tests/golden/), not against
 * outputs of OpenCV/g2o themselves.
 */
#ifndef MVO_ORACLE_H
#define MVO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* Parameters the reference latches from config.yaml in function-local statics
 * (feature_match.cpp:16-19, 42-45, 56-59). */
typedef struct {
    int32_t nfeatures;         /* number_of_keypoints_to_extract (8000) */
    float scale_factor;        /* scale_factor (1.2) -- ORB::create takes a float */
    int32_t nlevels;           /* level_pyramid (4) */
    int32_t fast_threshold;    /* score_threshold (20) */
    int32_t max_keypoints;     /* max_number_of_keypoints (1500) */
    int32_t grid_size;         /* kpts_uniform_selection_grid_size (16) */
    int32_t grid_max_per_cell; /* kpts_uniform_selection_max_pts_per_grid (8) */
    int32_t pyramid_interpolation; /* 1 = cv::INTER_LINEAR_EXACT (cv::ORB of OpenCV >= 3.4), 0 = cv::INTER_LINEAR */
} orc_orb_params;

/* One FAST+NMS survivor inside the 31-px border of a pyramid level, in canonical order
 * (level-major, then row-major), with both scores and the IC angle attached. */
typedef struct {
    int16_t x, y;       /* level coordinates */
    int32_t level;
    int32_t fast_score; /* cornerScore<16> */
    float harris;       /* HarrisResponses(blockSize 7, k 0.04) */
    float angle;        /* IC_Angle in degrees via the fastAtan2 polynomial */
} orc_candidate;

/* --- ORB stages (each returns a count or a negative error) --------------------------------- */
int orc_orb_level_size(int w, int h, const orc_orb_params* p, int level, int* lw, int* lh, float* scale);
int orc_orb_feature_quota(const orc_orb_params* p, int32_t* quota /* nlevels */);
/* Writes level `level` of the gray pyramid WITH its 32-px BORDER_REFLECT_101 frame:
 * out must hold (lh+64)*(lw+64) bytes, row stride lw+64.  blurred!=0 -> interior is the 7x7
 * sigma=2 fixed-point Gaussian, frame left unblurred (cv::ORB::compute blurs the level ROI in place). */
int orc_orb_pyramid_level(const uint8_t* img, int w, int h, int stride, int channels,
                          const orc_orb_params* p, int level, int blurred, uint8_t* out);
int orc_orb_candidates(const uint8_t* img, int w, int h, int stride, int channels,
                       const orc_orb_params* p, orc_candidate* out, int cap);
/* geometry::calcKeyPoints (feature_match.cpp:11-36): ORB detect + selectUniformKptsByGrid.
 * grid_rows/grid_cols <= 0 -> derive from (h, w) as the first call of the reference does. */
int orc_calc_keypoints(const uint8_t* img, int w, int h, int stride, int channels,
                       const orc_orb_params* p, int grid_rows, int grid_cols,
                       orc_keypoint* out, int cap);
/* Only cv::ORB::detect (no grid sampling). */
int orc_orb_detect(const uint8_t* img, int w, int h, int stride, int channels,
                   const orc_orb_params* p, orc_keypoint* out, int cap);
/* geometry::selectUniformKptsByGrid (feature_match.cpp:51-84); in place, returns new count. */
int orc_select_uniform_kpts_by_grid(orc_keypoint* kps, int n, int grid_rows, int grid_cols,
                                    const orc_orb_params* p);
/* geometry::calcDescriptors (feature_match.cpp:38-49): may drop keypoints (in place);
 * desc gets n_out*32 bytes; rgb (optional) n_out*3 bytes as frame.h:80-85 computes them. */
int orc_calc_descriptors(const uint8_t* img, int w, int h, int stride, int channels,
                         const orc_orb_params* p, orc_keypoint* kps, int n, uint8_t* desc,
                         uint8_t* rgb);

/* --- matching ------------------------------------------------------------------------------ */
/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (feature_match.cpp:203-208): idx/dist are nq x 2,
 * missing neighbours are (-1, INT32_MAX). */
int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
/* geometry::matchByRadiusAndBruteForce (feature_match.cpp:86-124): per query the first minimum of
 * the byte-L1 sum among trains within the pixel radius; idx -1 if none. sum is 32x the reference's
 * mean-abs-difference. */
int orc_match_radius_l1(const uint8_t* q, const float* qxy, int nq, const uint8_t* t,
                        const float* txy, int nt, float max_px, int32_t* idx, int32_t* sum);

typedef struct {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} orc_dmatch; /* cv::DMatch */

/* FLANN LSH(tables, key_size, multi_probe_level) 2-NN with seeded key bits, and matchFeatures' selection rules applied
 * to a given 2-NN table: used to quantify approximate-vs-exact matching (match_oracle.cpp). */
int orc_match_knn2_lsh(const uint8_t* q, int nq, const uint8_t* t, int nt, int tables, int key_size, int probe_level,
                       uint32_t seed, int32_t* idx, int32_t* dist);
int orc_match_features_from_knn(const int32_t* idx, const int32_t* dist, int n1, int method, double xiang_gao_ratio,
                                double lowe_ratio, orc_dmatch* out, int cap);

/* geometry::matchFeatures (feature_match.cpp:126-239). method 1 uses the exact 1-NN in place of
 * FLANN-LSH (documented deviation). ratios are passed as the reference latches
 * them (get<int> -> 0.8 becomes 1; pass 0.8 for the intended behaviour). Returns match count. */
int orc_match_features(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int method,
                       double xiang_gao_ratio, double lowe_ratio, const float* xy1,
                       const float* xy2, float max_px, orc_dmatch* out, int cap);
/* geometry::removeDuplicatedMatches (feature_match.cpp:241-260), in place. */
int orc_remove_duplicated_matches(orc_dmatch* m, int n);

/* --- bundle adjustment --------------------------------------------------------------------- */
typedef struct {
    int32_t n_poses, n_points, n_edges;
    double* pose_T_w_c;        /* n_poses x 16, row-major 4x4 cam->world (Frame::T_w_c_), in/out */
    double* points;            /* n_points x 3 world xyz, in/out (the caller rounds to f32) */
    const int32_t* edge_pose;  /* n_edges */
    const int32_t* edge_point; /* n_edges */
    const double* edge_uv;     /* n_edges x 2 */
    double focal, cx, cy;      /* g2o::CameraParameters(K(0,0), (K(0,2),K(1,2)), 0): fy ignored */
    double info[4];            /* 2x2 information matrix */
    double huber_delta;        /* RobustKernelHuber default 1.0 */
    int32_t fix_points;        /* is_fix_map_pts */
    const uint8_t* pose_fixed; /* optional n_poses flags (reference: none fixed); may be NULL */
    int32_t max_iterations;    /* 50 */
} orc_ba_problem;

typedef struct {
    int32_t iterations;  /* outer LM iterations executed */
    int32_t trials;      /* total linear solves */
    int32_t terminated;  /* 1 if g2o's Terminate condition ended the run early */
    double chi2_initial, chi2_final, lambda_final;
} orc_ba_stats;

/* optimization::bundleAdjustment (g2o_ba.cpp:172-317). */
int orc_bundle_adjustment(orc_ba_problem* prob, orc_ba_stats* stats);
/* How the reduced system is solved and what a failed solve does (ba_oracle.cpp):
 *   1 (default) g2o's LinearSolverDense as it is: Eigen::LDLT with its diagonal pivoting and sign rule, x left STALE by a
 *     failed solve and applied / scored all the same (OptimizationAlgorithmLevenberg::solve);
 *   0 the simplification of rounds 1-5 (unpivoted LDL^T, zero step after a failed solve), kept to quantify the difference. */
void orc_ba_set_solver_rule(int rule);
int orc_ba_get_solver_rule(void);
/* of the last orc_bundle_adjustment: out[0] = failed solves, out[1] = failed solves whose stale step was accepted */
void orc_ba_last_counters(int32_t* out);
/* Eigen::LDLT<MatrixXd> on a dense symmetric row-major A (lower triangle read): returns isPositive(); x = solve(b) only then. */
int orc_ldlt_eigen(const double* A, const double* b, int n, double* x);
/* The order in which Eigen::LDLT takes the rows of a matrix with this diagonal: perm[k] = row at position k. */
void orc_eigen_pivot_order(const double* diag, int n, int32_t* perm);
/* The same algorithm with the BLOCKED summation order the device declares (ba_blocked_oracle.cpp): G landmark ranges
 * (wg_pt_start: G + 1 entries), nsplit column pieces per Schur chain (bits 0..15 of `nsplit`; bits 16..: K > 1 = the Schur
 * exchange adds the ranges g = k mod K per group k first, then the K group sums).  trace (may be NULL): rows {lambda, chi2, rho,
 * accepted} per LM trial.  Used to check the MI355X solve bit for bit. */
int orc_bundle_adjustment_blocked(orc_ba_problem* prob, int G, const int32_t* wg_pt_start, int nsplit, orc_ba_stats* stats,
                                  double* trace, int trace_cap, int* trace_n);
/* One linearisation at the current state: dense H (n x n, n = 6*free poses + 3*free points),
 * b, robust chi2.  For known-answer tests of the Jacobians. */
int orc_ba_linearize(const orc_ba_problem* prob, double* H, double* b, double* chi2, int ncap);

/* --- tracking rows: pnp_oracle.cpp ---------------------------------- */
/* cv::Mat::inv() of a 4x4 double matrix (LU, partial pivoting); returns 0 if singular. */
int orc_invert4x4(const double* T, double* out);
/* VisualOdometry::getMappointsInCurrentView_ (vo.cpp:16-49): indices (map iteration order) and pixels of
 * the map points in front of the camera and strictly inside the image.  Returns the count. */
int orc_map_in_view(const float* pos, int n, const double* T_w_c, double fx, double fy, double cx, double cy,
                    int cols, int rows, int32_t* idx, float* px);
/* RANSACPointSetRegistrator::getSubset driven by cv::RNG((uint64)-1): n_iters x model_points indices. */
int orc_pnp_subsets(int count, int model_points, int n_iters, int32_t* idx);
/* solvePnP(SOLVEPNP_EPNP) on a subset; K4 = {fx, fy, cx, cy}; R row-major; ut (optional) 12x12. */
int orc_epnp(const float* p3, const float* p2, const int32_t* idx, int cnt, const double* K4, double* R,
             double* t, double* ut);
/* PnPRansacCallback::computeError + findInliers: returns the inlier count, mask (optional) n bytes. */
int orc_pnp_score(const float* p3, const float* p2, int n, const double* K4, const double* R, const double* t,
                  float reproj, uint8_t* mask);
int orc_rodrigues(const double* r, double* R, double* J /* 3x9 or NULL */);
int orc_rodrigues_inv(const double* R, double* r);
/* cvFindExtrinsicCameraParams2 (DLT + CvLevMarq) on double points; param = {rvec, tvec}. */
int orc_solve_pnp_iterative(const double* M, const double* m, int n, const double* K4, const double* init_R,
                            const double* init_t, double* param, int* lm_iters, int* lm_evals);
/* cv::solvePnPRansac as called at vo.cpp:326-329.  Returns 1 if a pose was found.  Debug outputs may be
 * NULL: models [iters x 12] = (R row-major, t), counts [iters], info[4] = {best iteration, iterations run,
 * DLT used, LM iterations}. */
int orc_solve_pnp_ransac(const float* p3, const float* p2, int n, const double* K4, int iters, float reproj,
                         double confidence, double* rvec, double* tvec, int32_t* inliers, int* n_inliers,
                         double* models, int32_t* counts, int32_t* info);

/* --- keyframe row: keyframe_oracle.cpp -------------------------------- */
/* geometry::helperTriangulatePoints (motion_estimation.cpp:214-247): matched pixels of the previous / current
 * keyframe + T_curr_to_prev -> points in the previous camera frame and after basics::transCoord. */
int orc_triangulate_points(const float* kp1, const float* kp2, int n, const double* K4, const double* R, const double* t,
                           float* pts_prev, float* pts_curr);
/* VisualOdometry::retainGoodTriangulationResult_ (vo.cpp:181-244); returns the count kept. */
int orc_retain_good_triangulation(const float* pts_curr, int n, const double* T_w_c_curr, const double* T_w_c_ref,
                                  double min_angle, double max_ratio, int32_t* keep, double* angles);

/* cv::findEssentialMat's five-point kernel on 5 normalised correspondences (x2^T E x1 = 0): up to 10 unit-norm
 * row-major candidates in E; dbg (optional, 257 doubles): null space 4x9, reduced 10x20 system, the degree-10
 * polynomial (11), its real roots (10, NaN padded). */
int orc_five_point(const double* q1, const double* q2, double* E, double* dbg);
/* real roots (ascending) of c[0] z^10 + ... + c[10]: Sturm isolation + bisection */
int orc_real_roots_deg10(const double* c, double* roots);
/* geometry::helperFindInlierMatchesByEpipolarCons (motion_estimation.cpp:182-198) = inlier mask of
 * cv::findEssentialMat(..., RANSAC, prob, threshold) as called at epipolar_geometry.cpp:36-39.  Returns the
 * inlier count. */
int orc_find_essential_inliers(const float* kp1, const float* kp2, int n, const double* K4, double prob, double threshold,
                               int max_iters, int32_t* inliers, int32_t* counts, int32_t* info, double* bestE);

#ifdef __cplusplus
}
#endif
#endif
