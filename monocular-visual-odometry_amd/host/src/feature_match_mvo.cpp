// host/src/feature_match_mvo.cpp -- REPLACES src/geometry/feature_match.cpp of the reference: defines exactly the nine
// functions that include/my_slam/geometry/feature_match.h:12-54 declares, against the reference's OWN headers (nothing of
// the reference is shadowed or edited: drop this file into the build instead of feature_match.cpp, add <repo>/include and
// this directory to the include path, link libmvo_hip.so -- INTEGRATION.md).  Same argument meaning, same latching of the
// parameters on first use, same exception on a wrong method index (feature_match.cpp:11-303) -- executed on the MI355X.
// cv::Mat / cv::KeyPoint / cv::DMatch are only touched through rows / cols / data / step / channels() / create() and the
// public fields, so the same file compiles against this repo's OpenCV-less mirror of the header for the tests (host/include).
#include "my_slam/geometry/feature_match.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <unordered_map>

#include "my_slam/basics/config.h"
#include "mvo_hot_path.h"

static_assert(sizeof(cv::KeyPoint) == sizeof(mvo_keypoint), "cv::KeyPoint is handed to the C-ABI as mvo_keypoint (28 bytes)");
static_assert(sizeof(cv::DMatch) == sizeof(mvo_dmatch), "cv::DMatch is handed to the C-ABI as mvo_dmatch (16 bytes)");

namespace my_slam {
namespace geometry {

namespace detail {
// State the adapters keep per ctx (a host thread may bind several ctxs one after the other through
// hot_path_ctx_binding()): "parameters latched" and the owner of the pyramid cached inside that ctx.
struct CtxState {
    bool configured = false;
    bool reuse_pyramid = false;
    long long pyramid_token = 0;
};
CtxState& ctx_state() {
    // keyed by the ctx's unique id, not its address: a ctx created where a destroyed one lived must not inherit "configured"
    static thread_local std::unordered_map<unsigned long long, CtxState> m;
    return m[mvo_ctx_uid(hot_path_ctx())];
}
// feature_match.cpp:16-19, 42-45, 56-59: parameters are read once (function-local statics in the reference); here once
// per ctx, so that every ctx a thread uses receives the configuration
void latch_orb_params() {
    CtxState& cs = ctx_state();
    if (cs.configured) return;
    mvo_orb_params p;
    p.nfeatures = basics::Config::get<int>("number_of_keypoints_to_extract");
    p.scale_factor = (float)basics::Config::get<double>("scale_factor");
    p.nlevels = basics::Config::get<int>("level_pyramid");
    p.fast_threshold = basics::Config::get<int>("score_threshold");
    p.max_keypoints = basics::Config::get<int>("max_number_of_keypoints");
    p.grid_size = basics::Config::get<int>("kpts_uniform_selection_grid_size");
    p.grid_max_per_cell = basics::Config::get<int>("kpts_uniform_selection_max_pts_per_grid");
    // not a config.yaml key: which cv::resize flavour cv::ORB uses depends on the OpenCV version (>= 3.4: EXACT)
    p.pyramid_interpolation = 1;
    try {  // (the reference's Config throws on a missing key, config.cpp:34-35)
        p.pyramid_interpolation = basics::Config::get<int>("orb_pyramid_interpolation");
    } catch (const std::exception&) {
    }
    mvo_check(mvo_orb_configure(hot_path_ctx(), &p), "mvo_orb_configure");
    cs.configured = true;
}
// the pyramid built by calcKeyPoints may be reused by calcDescriptors when the caller guarantees it is the same
// image (Frame::calcKeyPoints / calcDescriptors do); the free functions rebuild it by default.
bool& reuse_pyramid_flag() { return ctx_state().reuse_pyramid; }
// who owns the pyramid cached in the ctx: the Frame's unique id + 1 (ids are never reused, unlike addresses), 0 after
// any direct call of the free functions
long long& pyramid_token() { return ctx_state().pyramid_token; }

// The C-ABI takes n x 32 contiguous descriptor bytes (ORB: 256 bits).  A cv::Mat may be a ROI / a row range of a wider
// matrix (step != 32): its rows are packed into `tmp`; any other width is refused -- cv::batchDistance would take it, the
// Hamming kernels of this path are written for cv::ORB's 32 bytes (feature_match.cpp:45-48 is the only producer).
const uint8_t* packed_descriptors(const cv::Mat& d, std::vector<uint8_t>& tmp, const char* who) {
    if (d.rows > 0 && d.cols != 32) throw std::runtime_error(std::string(who) + ": descriptors must be n x 32 bytes (cv::ORB)");
    if (d.rows <= 0 || d.step == 32) return d.data;
    tmp.resize((size_t)d.rows * 32);
    for (int r = 0; r < d.rows; ++r) std::memcpy(tmp.data() + 32 * (size_t)r, d.ptr<unsigned char>(r), 32);
    return tmp.data();
}
}  // namespace detail

void calcKeyPoints(const cv::Mat& image, vector<cv::KeyPoint>& keypoints) {
    detail::latch_orb_params();
    detail::pyramid_token() = 0;
    const int cap = basics::Config::get<int>("max_number_of_keypoints") + 16;
    keypoints.resize(cap);
    int n = 0;
    mvo_check(mvo_calc_keypoints(hot_path_ctx(), image.data, image.cols, image.rows, (int)image.step, image.channels(),
                                 reinterpret_cast<mvo_keypoint*>(keypoints.data()), cap, &n),
              "calcKeyPoints");
    keypoints.resize(n);
}

/* Compute the descriptors of keypoints. Meanwhile, keypoints might be changed (feature_match.h:15-17). */
void calcDescriptors(const cv::Mat& image, vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
    detail::latch_orb_params();
    // without the reuse flag this call rebuilds the ctx pyramid from `image`: whoever owned the cached one lost it
    if (!detail::reuse_pyramid_flag()) detail::pyramid_token() = 0;
    int n = (int)keypoints.size();
    descriptors.create(n > 0 ? n : 1, 32, CV_8UC1);
    mvo_check(mvo_calc_descriptors(hot_path_ctx(), image.data, image.cols, image.rows, (int)image.step,
                                   image.channels(), detail::reuse_pyramid_flag() ? 1 : 0,
                                   reinterpret_cast<mvo_keypoint*>(keypoints.data()), &n, descriptors.data, nullptr),
              "calcDescriptors");
    keypoints.resize(n);
    // cv::ORB::compute hands back exactly one row per surviving keypoint: shrink through the matrix header's own operation
    // (rowRange keeps dataend / datalimit of a real cv::Mat consistent; writing `rows` directly does not)
    descriptors = descriptors.rowRange(0, n);
}

void removeDuplicatedMatches(vector<cv::DMatch>& matches) {
    int n = (int)matches.size();
    mvo_remove_duplicated_matches(reinterpret_cast<mvo_dmatch*>(matches.data()), &n);
    matches.resize(n);
}

void selectUniformKptsByGrid(vector<cv::KeyPoint>& keypoints, int image_rows, int image_cols) {
    detail::latch_orb_params();
    int n = (int)keypoints.size();
    mvo_check(mvo_select_uniform_kpts_by_grid(hot_path_ctx(), reinterpret_cast<mvo_keypoint*>(keypoints.data()), &n,
                                              image_rows, image_cols),
              "selectUniformKptsByGrid");
    keypoints.resize(n);
}

void matchFeatures(const cv::Mat1b& descriptors_1, const cv::Mat1b& descriptors_2, vector<cv::DMatch>& matches,
                          int method_index, bool is_print_res,
                          // Below are optional arguments for feature_matching_method_index==3
                          const vector<cv::KeyPoint>& keypoints_1, const vector<cv::KeyPoint>& keypoints_2,
                          float max_matching_pixel_dist) {
    // feature_match.cpp:137-139: the three ratios are read with get<int>
    static const double xiang_gao_method_match_ratio = basics::Config::get<int>("xiang_gao_method_match_ratio");
    static const double lowe_method_dist_ratio = basics::Config::get<int>("lowe_method_dist_ratio");
    matches.clear();
    if (method_index < 1 || method_index > 3)
        throw std::runtime_error("feature_match.cpp::matchFeatures: wrong method index.");  // :225
    vector<float> xy1, xy2;
    if (method_index == 3) {
        for (const cv::KeyPoint& k : keypoints_1) {
            xy1.push_back(k.pt.x);
            xy1.push_back(k.pt.y);
        }
        for (const cv::KeyPoint& k : keypoints_2) {
            xy2.push_back(k.pt.x);
            xy2.push_back(k.pt.y);
        }
    }
    const int n1 = descriptors_1.rows, n2 = descriptors_2.rows;
    std::vector<uint8_t> pack1, pack2;
    const uint8_t* d1 = detail::packed_descriptors(descriptors_1, pack1, "matchFeatures");
    const uint8_t* d2 = detail::packed_descriptors(descriptors_2, pack2, "matchFeatures");
    matches.resize(n1 > 0 ? n1 : 1);
    int n = 0;
    mvo_check(mvo_match_features(hot_path_ctx(), d1, n1, d2, n2, method_index,
                                 xiang_gao_method_match_ratio, lowe_method_dist_ratio, xy1.data(), xy2.data(),
                                 max_matching_pixel_dist, reinterpret_cast<mvo_dmatch*>(matches.data()),
                                 (int)matches.size(), &n),
              "matchFeatures");
    matches.resize(n);
    if (is_print_res) {
        printf("Matching features:\n");
        printf("Using method %d\n", method_index);
        printf("Number of matches: %d\n", int(matches.size()));
    }
}

// feature_match.cpp:86-124: per keypoint of image 1 the first minimum of the mean absolute descriptor difference
// among the keypoints of image 2 within max_matching_pixel_dist pixels (the device returns the byte sum: / 32 here).
vector<cv::DMatch> matchByRadiusAndBruteForce(const vector<cv::KeyPoint>& keypoints_1,
                                                     const vector<cv::KeyPoint>& keypoints_2,
                                                     const cv::Mat1b& descriptors_1, const cv::Mat1b& descriptors_2,
                                                     float max_matching_pixel_dist) {
    const int N1 = (int)keypoints_1.size(), N2 = (int)keypoints_2.size();
    if (N1 != descriptors_1.rows || N2 != descriptors_2.rows)  // the reference asserts (:94)
        throw std::runtime_error("matchByRadiusAndBruteForce: keypoints and descriptors differ in number");
    vector<float> xy1, xy2;
    for (const cv::KeyPoint& k : keypoints_1) {
        xy1.push_back(k.pt.x);
        xy1.push_back(k.pt.y);
    }
    for (const cv::KeyPoint& k : keypoints_2) {
        xy2.push_back(k.pt.x);
        xy2.push_back(k.pt.y);
    }
    vector<int32_t> idx((size_t)(N1 > 0 ? N1 : 1)), sum((size_t)(N1 > 0 ? N1 : 1));
    std::vector<uint8_t> pack1, pack2;
    const uint8_t* d1 = detail::packed_descriptors(descriptors_1, pack1, "matchByRadiusAndBruteForce");
    const uint8_t* d2 = detail::packed_descriptors(descriptors_2, pack2, "matchByRadiusAndBruteForce");
    mvo_check(mvo_match_radius_l1(hot_path_ctx(), d1, xy1.data(), N1, d2, xy2.data(), N2,
                                  max_matching_pixel_dist, idx.data(), sum.data()),
              "matchByRadiusAndBruteForce");
    vector<cv::DMatch> matches;
    for (int i = 0; i < N1; ++i)
        if (idx[i] >= 0) matches.push_back(cv::DMatch(i, idx[i], static_cast<float>((double)sum[i] / descriptors_1.cols)));
    return matches;
}

// --------------------- Other assistant functions (feature_match.cpp:263-278; caller vo.cpp:139) ---------------------
double computeMeanDistBetweenKeypoints(const vector<cv::KeyPoint>& kpts1, const vector<cv::KeyPoint>& kpts2,
                                              const vector<cv::DMatch>& matches) {
    vector<double> dists_between_kpts;
    for (const cv::DMatch& d : matches) {
        const cv::Point2f p1 = kpts1[d.queryIdx].pt, p2 = kpts2[d.trainIdx].pt;
        const double dx = p1.x - p2.x, dy = p1.y - p2.y;  // basics::calcDist (opencv_funcs.cpp:132-136)
        dists_between_kpts.push_back(sqrt(dx * dx + dy * dy));
    }
    double mean_dist = 0;
    for (double d : dists_between_kpts) mean_dist += d;
    mean_dist /= dists_between_kpts.size();  // (0 / 0 = NaN for an empty list, like the reference)
    return mean_dist;
}

// --------------------- Datatype conversion (feature_match.cpp:281-303; caller vo.cpp:277) ---------------------
vector<cv::DMatch> inliers2DMatches(const vector<int> inliers) {
    vector<cv::DMatch> matches;
    for (auto idx : inliers) matches.push_back(cv::DMatch(idx, idx, 0.0));
    return matches;
}
vector<cv::KeyPoint> pts2Keypts(const vector<cv::Point2f> pts) {
    vector<cv::KeyPoint> keypts;
    for (cv::Point2f pt : pts) keypts.push_back(cv::KeyPoint(pt, 10));
    return keypts;
}

}  // namespace geometry
}  // namespace my_slam
