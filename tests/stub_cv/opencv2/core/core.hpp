// tests/stub_cv/opencv2/core/core.hpp -- TEST AID.  OpenCV is not installed in this image; this tree supplies the cv:: names
// the reference's OWN headers mention, so that tests/test_dropin_headers.py can compile the two drop-in translation units
// (host/src/feature_match_mvo.cpp, g2o_ba_mvo.cpp) and syntax-check reference sources against /root/reference/include
// UNMODIFIED.  The value types come from the mirror's mini_cv.h (layout-compatible KeyPoint / DMatch / Mat subset); what the
// reference headers need on top of it (Mat_<T> with the comma initialiser of camera.h:31, FileStorage / FileNode of
// config.h:40,57-70) is declared here -- declarations only where nothing in the test links against them.
#ifndef MVO_STUB_OPENCV_CORE_HPP
#define MVO_STUB_OPENCV_CORE_HPP
#include <string>
#include <vector>

#include "../../../../monocular-visual-odometry_amd/host/include/my_slam/mini_cv.h"

namespace cv {
template <class T>
class MatCommaInitializer_ {
public:
    explicit MatCommaInitializer_(Mat* m) : m_(m) {}
    template <class U>
    MatCommaInitializer_& operator,(U v) {
        m_->ptr<T>(k_ / m_->cols)[k_ % m_->cols] = static_cast<T>(v);
        ++k_;
        return *this;
    }
    operator Mat() const { return *m_; }
    int k_ = 0;

private:
    Mat* m_;
};
template <class T>
class Mat_ : public Mat {
public:
    Mat_() {}
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 8 ? CV_64FC1 : (sizeof(T) == 4 ? CV_32SC1 : CV_8UC1)) {}
    template <class U>
    MatCommaInitializer_<T> operator<<(U v) {
        MatCommaInitializer_<T> ci(this);
        return (ci, v);
    }
};
class FileNode {
public:
    bool empty() const;
    template <class T>
    operator T() const;
};
template <class T>
void operator>>(const FileNode& n, std::vector<T>& v);
class FileStorage {
public:
    enum { READ = 0 };
    FileStorage();
    FileStorage(const std::string& filename, int flags);
    bool isOpened() const;
    void release();
    FileNode operator[](const std::string& key) const;
};
}  // namespace cv
#endif
