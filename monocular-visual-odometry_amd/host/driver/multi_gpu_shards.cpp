// host/driver/multi_gpu_shards.cpp -- SURVEY.md 8(e) from the C++ host: one sequence shard per GPU, no data-path
// collective, ONE RCCL all-gather of the trajectory blocks (frames x 12 f64, the row format of the reference's
// trajectory file: tx ty tz R[:,0] R[:,1] R[:,2], src/vo/vo_io.cpp:58-75) at the end.
//
//   multi_gpu_shards [ranks = visible GPUs] [frames = 24] [out_prefix]
//
// One host thread per rank: mvo_create(&ctx, rank) -- the ctx (its HIP stream, its device buffers) is the only state a
// shard has on its GPU; it is bound to the thread so that the reference-shaped free functions (Frame::calcKeyPoints /
// calcDescriptors -> geometry::calcKeyPoints ..., geometry::matchFeatures; feature_match.h:12-46) use it.  Every rank
// renders its own frames (seed 1234 + rank), extracts and matches them frame by frame and turns the matches of a frame into a
// pose row.  The pose model of this EXAMPLE is the image-plane shift of the matched keypoints (no map, no PnP -- the
// headless run_vo next to this file runs the full tracking rows); what it shows is the sharding: contexts per device, no
// exchange while the sequences run, and the gather.  bench.py does the same with torch.distributed (one process per GPU).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "mvo_hot_path.h"
#include "my_slam/geometry/feature_match.h"
#include "my_slam/vo/frame.h"
#include "my_slam/vo/vo_io.h"

using namespace my_slam;

namespace {
constexpr int W = 640, H = 480;

// frame i of rank r: a fixed random-rectangle scene (seed 1234 + r) seen through a window that moves 3 px per frame
cv::Mat renderFrame(int rank, int i) {
    cv::Mat img(H, W, CV_8UC3);
    unsigned s = 1234u + (unsigned)rank;
    auto rnd = [&s]() { return (s = s * 1664525u + 1013904223u) >> 8; };
    std::fill(img.data, img.data + (size_t)W * H * 3, (unsigned char)128);
    for (int k = 0; k < 900; ++k) {
        const int w = 6 + (int)(rnd() % 50), h = 6 + (int)(rnd() % 50), g = 20 + (int)(rnd() % 215);
        const int x0 = (int)(rnd() % 1400) - 3 * i - 300, y0 = (int)(rnd() % 700) - 100;
        for (int y = std::max(0, y0); y < std::min(H, y0 + h); ++y)
            for (int x = std::max(0, x0); x < std::min(W, x0 + w); ++x)
                for (int c = 0; c < 3; ++c) img.data[((size_t)y * W + x) * 3 + c] = (unsigned char)g;
    }
    return img;
}

struct Rank {
    int id = 0, frames = 0, error = 0;
    ncclComm_t comm{};
    std::vector<double> traj;       // frames x 12: this rank's rows
    std::vector<double> gathered;   // ranks x frames x 12
    long kp = 0, matches = 0;
};

void runRank(Rank& R, int n_ranks) {
    try {
        mvo_ctx* ctx = nullptr;
        if (mvo_create(&ctx, R.id) != MVO_OK) throw std::runtime_error("mvo_create failed (no CPU fallback)");
        hot_path_ctx_binding() = ctx;  // the free functions of this thread use the rank's ctx
        cv::Mat T = cv::Mat::eye(4, 4, CV_64FC1);
        vo::Frame::Ptr prev;
        R.traj.assign((size_t)R.frames * 12, 0.0);
        for (int i = 0; i < R.frames; ++i) {
            vo::Frame::Ptr f = vo::Frame::createFrame(renderFrame(R.id, i));
            f->calcKeyPoints();
            f->calcDescriptors();
            R.kp += (long)f->keypoints_.size();
            if (prev) {
                vector<cv::DMatch> m;
                geometry::matchFeatures(prev->descriptors_, f->descriptors_, m, 2);
                R.matches += (long)m.size();
                vector<float> dx, dy;
                for (const cv::DMatch& d : m) {
                    dx.push_back(f->keypoints_[d.trainIdx].pt.x - prev->keypoints_[d.queryIdx].pt.x);
                    dy.push_back(f->keypoints_[d.trainIdx].pt.y - prev->keypoints_[d.queryIdx].pt.y);
                }
                if (!dx.empty()) {  // median image shift -> translation of the example's pose
                    std::nth_element(dx.begin(), dx.begin() + dx.size() / 2, dx.end());
                    std::nth_element(dy.begin(), dy.begin() + dy.size() / 2, dy.end());
                    T.at<double>(0, 3) -= dx[dx.size() / 2];
                    T.at<double>(1, 3) -= dy[dy.size() / 2];
                }
            }
            double* row = &R.traj[(size_t)i * 12];  // vo_io.cpp:58-75
            for (int k = 0; k < 3; ++k) row[k] = T.at<double>(k, 3);
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) row[3 + 3 * c + r] = T.at<double>(r, c);
            prev = f;
        }
        // ---- the one collective: all-gather of the trajectory blocks over RCCL (xGMI between the GPUs of a node)
        const size_t n = (size_t)R.frames * 12;
        hipStream_t st;
        double *d_send = nullptr, *d_recv = nullptr;
        if (hipSetDevice(R.id) != hipSuccess || hipStreamCreate(&st) != hipSuccess || hipMalloc((void**)&d_send, n * 8) != hipSuccess ||
            hipMalloc((void**)&d_recv, n * 8 * n_ranks) != hipSuccess)
            throw std::runtime_error("device buffers for the gather");
        (void)hipMemcpyAsync(d_send, R.traj.data(), n * 8, hipMemcpyHostToDevice, st);
        if (ncclAllGather(d_send, d_recv, n, ncclDouble, R.comm, st) != ncclSuccess) throw std::runtime_error("ncclAllGather");
        R.gathered.assign(n * n_ranks, 0.0);
        (void)hipMemcpyAsync(R.gathered.data(), d_recv, n * 8 * n_ranks, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) throw std::runtime_error("gather stream");
        (void)hipFree(d_send);
        (void)hipFree(d_recv);
        (void)hipStreamDestroy(st);
        hot_path_ctx_binding() = nullptr;
        mvo_destroy(ctx);
    } catch (const std::exception& e) {
        fprintf(stderr, "rank %d: %s\n", R.id, e.what());
        R.error = 1;
    }
}
}  // namespace

int main(int argc, char** argv) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
        fprintf(stderr, "multi_gpu_shards: no HIP device (there is no CPU fallback)\n");
        return 1;
    }
    const int n_ranks = argc > 1 ? std::min(std::max(1, atoi(argv[1])), n_dev) : n_dev;
    const int frames = argc > 2 ? std::max(2, atoi(argv[2])) : 24;
    const std::string prefix = argc > 3 ? argv[3] : "";
    std::vector<int> devs(n_ranks);
    for (int r = 0; r < n_ranks; ++r) devs[r] = r;
    std::vector<ncclComm_t> comms(n_ranks);
    if (ncclCommInitAll(comms.data(), n_ranks, devs.data()) != ncclSuccess) {
        fprintf(stderr, "multi_gpu_shards: ncclCommInitAll failed\n");
        return 1;
    }
    std::vector<Rank> ranks(n_ranks);
    std::vector<std::thread> th;
    for (int r = 0; r < n_ranks; ++r) {
        ranks[r].id = r;
        ranks[r].frames = frames;
        ranks[r].comm = comms[r];
        th.emplace_back(runRank, std::ref(ranks[r]), n_ranks);
    }
    for (std::thread& t : th) t.join();
    for (ncclComm_t c : comms) (void)ncclCommDestroy(c);
    int bad = 0;
    for (const Rank& R : ranks) bad |= R.error;
    if (bad) return 1;
    // every rank holds all trajectories; they must agree with what each rank computed
    for (int r = 0; r < n_ranks; ++r)
        for (int q = 0; q < n_ranks; ++q)
            if (!std::equal(ranks[q].traj.begin(), ranks[q].traj.end(), ranks[r].gathered.begin() + (size_t)q * frames * 12)) bad = 1;
    for (int q = 0; q < n_ranks && !prefix.empty(); ++q) {
        vector<cv::Mat> poses;
        for (int i = 0; i < frames; ++i) {
            const double* row = &ranks[0].gathered[((size_t)q * frames + i) * 12];
            cv::Mat T = cv::Mat::eye(4, 4, CV_64FC1);
            for (int k = 0; k < 3; ++k) T.at<double>(k, 3) = row[k];
            for (int c = 0; c < 3; ++c)
                for (int rr = 0; rr < 3; ++rr) T.at<double>(rr, c) = row[3 + 3 * c + rr];
            poses.push_back(T);
        }
        vo::writePoseToFile(prefix + "_shard" + std::to_string(q) + ".txt", poses);  // the reference's trajectory file format
    }
    long kp = 0, m = 0;
    for (const Rank& R : ranks) kp += R.kp, m += R.matches;
    printf("%s: %d rank(s) x %d frames x 12 gathered on every rank; %ld keypoints, %ld matches; last row of shard 0: %.1f %.1f %.1f\n",
           bad ? "MISMATCH" : "ok", n_ranks, frames, kp, m, ranks[0].gathered[(size_t)(frames - 1) * 12], ranks[0].gathered[(size_t)(frames - 1) * 12 + 1],
           ranks[0].gathered[(size_t)(frames - 1) * 12 + 2]);
    return bad;
}
