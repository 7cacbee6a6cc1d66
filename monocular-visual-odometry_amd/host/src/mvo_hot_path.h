// host/src/mvo_hot_path.h -- PRIVATE to the drop-in translation units (feature_match_mvo.cpp, g2o_ba_mvo.cpp) and to this
// repo's OpenCV-less mirror headers: which mvo_ctx the free functions of the calling thread use.  The reference's functions
// take no context argument (they keep cv::ORB objects, matchers and latched parameters in function-local statics,
// src/geometry/feature_match.cpp:16-23,42-45,56-62,137-141); a ctx per host thread plays that role here.
// Not under my_slam/: adding this directory to an include path shadows none of the reference's headers.
#ifndef MVO_HOT_PATH_H
#define MVO_HOT_PATH_H
#include <stdexcept>
#include <string>

#include "mvo_hip.h"

namespace my_slam {
// A driver that manages its own contexts (one per sequence, or two per sequence when it overlaps extraction with bundle
// adjustment) binds the one the functions of the calling thread shall use; nullptr = the thread's default.
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
    if (r != MVO_OK) throw std::runtime_error(std::string(where) + ": " + mvo_last_error(hot_path_ctx()));
}
}  // namespace my_slam
#endif
