// my_slam/common_include.h -- what every header of the hot-path mirror needs (reference: include/my_slam/common_include.h).
#ifndef MY_SLAM_COMMON_INCLUDE_H
#define MY_SLAM_COMMON_INCLUDE_H
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "my_slam/mini_cv.h"
#include "mvo_hip.h"

namespace my_slam {
using std::string;
using std::vector;

// One process-wide mvo_ctx per host thread: plays the role of the function-local statics (cv::ORB objects,
// matchers, latched parameters) the reference keeps inside feature_match.cpp / vo.cpp.
// A driver that manages its own contexts (one per sequence, or two per sequence when it overlaps extraction with
// bundle adjustment) binds the one the adapters of the calling thread shall use; nullptr = the thread's default.
inline mvo_ctx*& hot_path_ctx_binding() {
    static thread_local mvo_ctx* bound = nullptr;
    return bound;
}
inline mvo_ctx* hot_path_ctx() {
    if (hot_path_ctx_binding()) return hot_path_ctx_binding();
    struct Holder {
        mvo_ctx* c = nullptr;
        Holder() {
            int r = mvo_create(&c, 0);
            if (r != MVO_OK) throw std::runtime_error("mvo_create failed: no usable MI355X / HIP device (no CPU fallback)");
        }
        ~Holder() { mvo_destroy(c); }
    };
    static thread_local Holder h;
    return h.c;
}
inline void mvo_check(int r, const char* where) {
    if (r != MVO_OK) throw std::runtime_error(string(where) + ": " + mvo_last_error(hot_path_ctx()));
}
}  // namespace my_slam
#endif
