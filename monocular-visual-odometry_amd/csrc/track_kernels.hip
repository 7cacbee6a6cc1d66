// csrc/track_kernels.hip -- tracking rows between the matcher and bundle adjustment:
//   k_map_in_view      VisualOdometry::getMappointsInCurrentView_ (src/vo/vo.cpp:16-49): project the resident map,
//                      keep what is in front of the camera and inside the image, gather the descriptors.
//   k_pnp_hypotheses   the RANSAC loop of cv::solvePnPRansac (vo.cpp:326-329): one workgroup of three waves per
//                      hypothesis runs the 5-point EPnP kernel (one wave per beta variant) and scores every pair.
//   k_pnp_refine       the final cv::solvePnP(SOLVEPNP_ITERATIVE) on the inliers: DLT start + Levenberg-Marquardt.
//   k_triangulate      geometry::helperTriangulatePoints (motion_estimation.cpp:214-247) on a keyframe's matches.
//   k_em_hypotheses    the RANSAC loop of cv::findEssentialMat (epipolar_geometry.cpp:36-39): one wave per five-point
//   k_em_mask          hypothesis; the inlier mask of the selected candidate.
// The arithmetic lives in pnp_wave.h (wave-level SPMD code); this file binds it to threads and LDS.
#include "mvo_internal.h"

#include <cstdlib>

#define PW_FN __device__ __forceinline__
#define PW_LANES(l, NL) for (int l = (int)threadIdx.x, pw_once_ = 1; pw_once_; pw_once_ = 0)
#define PW_WAVES(w, NW) for (int w = (int)(threadIdx.x >> 6), pw_once_ = 1; pw_once_; pw_once_ = 0)
#define PW_SYNC() __syncthreads()
#define PW_UNROLL _Pragma("unroll")
#include "em_wave.h"

// ------------------------------------------------------------------------------------------------ map in view
// One workgroup walks the map in chunks of 1024 points and appends the survivors in map order (the reference
// iterates Map::map_points_ and push_backs).  p_cam = (float)(T_c_w * p) accumulated in double like
// basics::preTranslatePoint3f (opencv_funcs.cpp:67-78); pixel = (float)(fx * x / z + cx) like geometry::cam2pixel
// (camera.cpp:23-28); the tests are `p_cam.z < 0` and the strict image bounds of vo.cpp:31-36.
__global__ __launch_bounds__(1024) void k_map_in_view(const float* __restrict__ pos, const uint4* __restrict__ desc,
                                                       int n, TrackViewArgs a, int32_t* __restrict__ idx,
                                                       float2* __restrict__ px, uint4* __restrict__ desc_out,
                                                       int32_t* __restrict__ n_out) {
    __shared__ int wave_cnt[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int base = 0;
    for (int start = 0; start < n; start += 1024) {
        const int i = start + (int)threadIdx.x;
        bool in = false;
        float u = 0.f, v = 0.f;
        if (i < n) {
            const double p[4] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 1.0};
            double res[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += a.T[4 * r + j] * p[j];
                res[r] = acc;
            }
            const float pcx = (float)res[0], pcy = (float)res[1], pcz = (float)res[2];
            in = !(pcz < 0);
            u = (float)(a.fx * pcx / pcz + a.cx);
            v = (float)(a.fy * pcy / pcz + a.cy);
            in = in && (u > 0 && v > 0 && u < (float)a.cols && v < (float)a.rows);
        }
        const unsigned long long b = __ballot(in);
        if (lane == 0) wave_cnt[w] = __popcll(b);
        __syncthreads();
        int off = base, total = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = wave_cnt[q];
            off += q < w ? c : 0;
            total += c;
        }
        if (in) {
            const int o = off + __popcll(b & ((1ull << lane) - 1ull));
            idx[o] = i;
            px[o] = make_float2(u, v);
            desc_out[2 * o] = desc[2 * i];
            desc_out[2 * o + 1] = desc[2 * i + 1];
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}

int track_launch_map_in_view(mvo_ctx* ctx, const float* d_pos, const uint8_t* d_desc, int n, const TrackViewArgs& a,
                             int32_t* d_idx, float* d_px, uint8_t* d_desc_out, int32_t* d_n) {
    ProfScope ps(ctx, "k_map_in_view");
    hipLaunchKernelGGL(k_map_in_view, dim3(1), dim3(1024), 0, ctx->stream, d_pos, (const uint4*)d_desc, n, a, d_idx,
                       (float2*)d_px, (uint4*)d_desc_out, d_n);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

// ------------------------------------------------------------------------------------------------ PnP RANSAC
__device__ __forceinline__ void pnp_hypothesis_body(const float* __restrict__ p3, const float* __restrict__ p2, int n,
                                                        const int32_t* __restrict__ subsets, TrackCamera cam, float thr2,
                                                        double* __restrict__ models, int32_t* __restrict__ counts,
                                                        uint8_t* __restrict__ masks, double* __restrict__ h_models,
                                                        int32_t* __restrict__ h_counts, pw::HypLds& lds) {
    const int h = blockIdx.x;
    const pw::Camera c{cam.fx, cam.fy, cam.cx, cam.cy};
    double R[3][3], t[3];
    pw::epnp_hypothesis(lds, p3, p2, subsets + pw::kModelPoints * h, c, R, t);
    const int good = pw::score_model(lds, p3, p2, n, c, R, t, thr2, masks + (size_t)h * n);
    if (threadIdx.x == 0) {
        double* m = models + 12 * (size_t)h;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) m[3 * i + j] = R[i][j];
            m[9 + i] = t[i];
        }
        counts[h] = good;
        // the same record straight into the caller's pinned buffer (the refinement kernel reads the device copy: its replay of
        // the RANSAC bookkeeping is a chain of dependent loads)
        double* hm = h_models + 12 * (size_t)h;
#pragma unroll
        for (int i = 0; i < 12; ++i) hm[i] = m[i];
        h_counts[h] = good;
    }
}

// Two builds of the same body: the compiler's free choice (256 VGPR + 52 AGPR = ONE wave per SIMD: fastest for a lone frame on an
// empty chip, 200 us) and a build held to two waves per SIMD (40 registers spilled to scratch): next to the resident solver grid
// the 100 hypotheses of a frame share a few CUs, where a workgroup that holds a CU to itself is what limits the frame rate.
__global__ __launch_bounds__(pw::kHypLanes) void k_pnp_hypotheses(const float* __restrict__ p3, const float* __restrict__ p2, int n,
                                                        const int32_t* __restrict__ subsets, TrackCamera cam, float thr2,
                                                        double* __restrict__ models, int32_t* __restrict__ counts,
                                                        uint8_t* __restrict__ masks, double* __restrict__ h_models,
                                                        int32_t* __restrict__ h_counts) {
    __shared__ pw::HypLds lds;
    pnp_hypothesis_body(p3, p2, n, subsets, cam, thr2, models, counts, masks, h_models, h_counts, lds);
}
__global__ __launch_bounds__(pw::kHypLanes) MVO_WAVES_PER_EU(2, 2) void k_pnp_hypotheses_occ2(
    const float* __restrict__ p3, const float* __restrict__ p2, int n, const int32_t* __restrict__ subsets, TrackCamera cam, float thr2,
    double* __restrict__ models, int32_t* __restrict__ counts, uint8_t* __restrict__ masks, double* __restrict__ h_models,
    int32_t* __restrict__ h_counts) {
    __shared__ pw::HypLds lds;
    pnp_hypothesis_body(p3, p2, n, subsets, cam, thr2, models, counts, masks, h_models, h_counts, lds);
}

// Picks the best hypothesis (the RANSAC loop's bookkeeping replayed over the counts, or `forced_best` >= 0), keeps
// its mask in best_mask for the host and refines it.  out: param[6] = (rvec, tvec), n_inliers, dlt used, LM
// iterations, LM evaluations, best hypothesis (-1: none reached 5 inliers), iterations the sequential loop runs.
__global__ __launch_bounds__(pw::kRefLanes) void k_pnp_refine(const float* __restrict__ p3, const float* __restrict__ p2,
                                                              const uint8_t* __restrict__ masks, int n, TrackCamera cam,
                                                              const double* __restrict__ models,
                                                              const int32_t* __restrict__ counts, int n_hyp,
                                                              double confidence, int forced_best, int mode, double* Mg,
                                                              double* mg, uint8_t* __restrict__ best_mask,
                                                              double* __restrict__ out) {
    __shared__ pw::RefLds lds;
    const pw::Camera c{cam.fx, cam.fy, cam.cx, cam.cy};
    int best = forced_best, iters_run = mode == 1 ? 1 : 0;
    if (forced_best < 0) pw::ransac_replay(counts, n_hyp, n, confidence, &best, &iters_run);
    if (best < 0) {
        if (threadIdx.x == 0) {
            for (int k = 0; k < 10; ++k) out[k] = 0;
            out[10] = -1;
            out[11] = iters_run;
        }
        return;
    }
    const uint8_t* mask = masks + (size_t)best * n;
    const double* model = models + 12 * (size_t)best;
    for (int i = threadIdx.x; i < n; i += pw::kRefLanes) best_mask[i] = mask[i];
    double R0[3][3], t0[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R0[i][j] = model[3 * i + j];
        t0[i] = model[9 + i];
    }
    pw::RefineResult res;
    pw::refine_pose(lds, p3, p2, mask, n, c, R0, t0, mode, Mg, mg, res);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = res.param[k];
        out[6] = res.n_inliers;
        out[7] = res.used_dlt;
        out[8] = res.lm_iters;
        out[9] = res.lm_evals;
        out[10] = best;
        out[11] = iters_run;
    }
}

// ------------------------------------------------------------------------------------------------ triangulation
// geometry::helperTriangulatePoints (motion_estimation.cpp:214-247): one lane per match.
struct TrackPose {
    double R[9], t[3];
};
__global__ __launch_bounds__(256) void k_triangulate(const float2* __restrict__ kp1, const float2* __restrict__ kp2, int n,
                                                      TrackCamera cam, TrackPose pose, float* __restrict__ pts_prev,
                                                      float* __restrict__ pts_curr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pw::Camera c{cam.fx, cam.fy, cam.cx, cam.cy};
    const float a[2] = {kp1[i].x, kp1[i].y}, b[2] = {kp2[i].x, kp2[i].y};
    float pp[3], pc[3];
    pw::triangulate_match(a, b, c, pose.R, pose.t, pp, pc);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        pts_prev[3 * i + r] = pp[r];
        pts_curr[3 * i + r] = pc[r];
    }
}

int track_launch_triangulate(mvo_ctx* ctx, const float* d_kp1, const float* d_kp2, int n, const TrackCamera& cam,
                             const double* R, const double* t, float* d_prev, float* d_curr) {
    TrackPose pose;
    for (int k = 0; k < 9; ++k) pose.R[k] = R[k];
    for (int k = 0; k < 3; ++k) pose.t[k] = t[k];
    ProfScope ps(ctx, "k_triangulate");
    hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float2*)d_kp1,
                       (const float2*)d_kp2, n, cam, pose, d_prev, d_curr);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

// ------------------------------------------------------------------------------------------------ essential matrix
// cv::findEssentialMat's RANSAC loop (epipolar_geometry.cpp:36-39): one wave per hypothesis solves the five-point
// problem and counts the Sampson inliers of each of its (<= 10) candidates.
__global__ __launch_bounds__(pw::kEmLanes) void k_em_hypotheses(const double* __restrict__ q1, const double* __restrict__ q2,
                                                               int n, const int32_t* __restrict__ subsets, float thr2,
                                                               double* __restrict__ E, int32_t* __restrict__ n_models,
                                                               int32_t* __restrict__ counts) {
    __shared__ pw::EmLds lds;
    const size_t h = blockIdx.x;
    const int nm = pw::five_point_hypothesis(lds, q1, q2, subsets + 5 * h, E + 90 * h);
    __threadfence_block();
    __syncthreads();  // the candidates were written by lanes 0..8, every lane reads them for the scoring
    pw::score_essentials(lds, q1, q2, n, E + 90 * h, nm, thr2, counts + 10 * h);
    if (threadIdx.x == 0) n_models[h] = nm;
}

// inlier mask of the selected candidate (the bestMask of RANSACPointSetRegistrator::run)
__global__ __launch_bounds__(256) void k_em_mask(const double* __restrict__ q1, const double* __restrict__ q2, int n,
                                                 const double* __restrict__ E, float thr2, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    mask[i] = pw::sampson_inlier(e, q1[2 * i], q1[2 * i + 1], q2[2 * i], q2[2 * i + 1], thr2) ? 1 : 0;
}

int track_launch_em_hypotheses(mvo_ctx* ctx, const double* d_q1, const double* d_q2, int n, const int32_t* d_subsets,
                               int n_hyp, float thr2, double* d_E, int32_t* d_nm, int32_t* d_counts) {
    ProfScope ps(ctx, "k_em_hypotheses");
    hipLaunchKernelGGL(k_em_hypotheses, dim3(n_hyp), dim3(pw::kEmLanes), 0, ctx->stream, d_q1, d_q2, n, d_subsets, thr2,
                       d_E, d_nm, d_counts);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int track_launch_em_mask(mvo_ctx* ctx, const double* d_q1, const double* d_q2, int n, const double* d_E, float thr2,
                         uint8_t* d_mask) {
    ProfScope ps(ctx, "k_em_mask");
    hipLaunchKernelGGL(k_em_mask, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_q1, d_q2, n, d_E, thr2, d_mask);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int track_launch_pnp_hypotheses(mvo_ctx* ctx, const float* d_p3, const float* d_p2, int n, const int32_t* d_subsets,
                                int n_hyp, const TrackCamera& cam, float thr2, double* d_models, int32_t* d_counts,
                                uint8_t* d_masks, double* h_models, int32_t* h_counts) {
    ProfScope ps(ctx, "k_pnp_hypotheses");
    static const int env_occ = std::getenv("MVO_PNP_OCC") ? std::atoi(std::getenv("MVO_PNP_OCC")) : 0;  // 0: by mode, 1 / 2: forced
    const bool occ2 = env_occ ? env_occ == 2 : ctx->ba_throughput_mode != 0;
    hipLaunchKernelGGL(occ2 ? k_pnp_hypotheses_occ2 : k_pnp_hypotheses, dim3(n_hyp), dim3(pw::kHypLanes), 0, ctx->stream, d_p3, d_p2, n, d_subsets, cam, thr2,
                       d_models, d_counts, d_masks, h_models, h_counts);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}

int track_launch_pnp_refine(mvo_ctx* ctx, const float* d_p3, const float* d_p2, const uint8_t* d_masks, int n,
                            const TrackCamera& cam, const double* d_models, const int32_t* d_counts, int n_hyp,
                            double confidence, int forced_best, int mode, double* d_Mg, double* d_mg,
                            uint8_t* d_best_mask, double* d_out) {
    ProfScope ps(ctx, "k_pnp_refine");
    hipLaunchKernelGGL(k_pnp_refine, dim3(1), dim3(pw::kRefLanes), 0, ctx->stream, d_p3, d_p2, d_masks, n, cam, d_models,
                       d_counts, n_hyp, confidence, forced_best, mode, d_Mg, d_mg, d_best_mask, d_out);
    MVO_HIP(hipGetLastError());
    return MVO_OK;
}
