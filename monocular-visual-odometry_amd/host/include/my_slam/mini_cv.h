// my_slam/mini_cv.h -- the handful of cv:: types the hot-path interface is written in (cv::KeyPoint, cv::DMatch,
// cv::Point2f/3f, cv::Mat for u8 images / Nx32 descriptors / small f64 matrices), for builds WITHOUT OpenCV
// (none is installed in the MI355X image).  With OpenCV available define MVO_HAVE_OPENCV and the real headers are
// used instead: the adapters only touch rows / cols / data / step / channels() / create() / at<T>() / ptr<T>().
// Layouts match OpenCV's (KeyPoint 28 B, DMatch 16 B) so the vectors are handed to the C-ABI without copies.
#ifndef MY_SLAM_MINI_CV_H
#define MY_SLAM_MINI_CV_H
#ifdef MVO_HAVE_OPENCV
#include <opencv2/core.hpp>
#else
#include <cfloat>
#include <cmath>
#include <utility>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_32S 4
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Point3f {
    float x, y, z;
    Point3f() : x(0), y(0), z(0) {}
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct Vec3b {
    uchar val[3];
    uchar& operator[](int i) { return val[i]; }
    const uchar& operator[](int i) const { return val[i]; }
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1)
        : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
        : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(FLT_MAX) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};
static_assert(sizeof(DMatch) == 16, "cv::DMatch layout");

// Dense 2-D matrix with shared storage (copy = shallow like cv::Mat, clone() = deep).
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), type_(type) {  // wraps, no copy
        step = step_ ? step_ : (size_t)c * elemSize();
        data = static_cast<uchar*>(ext);
    }
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data) return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elemSize();
        store_ = std::make_shared<std::vector<uchar>>((size_t)r * step, 0);
        data = store_->data();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type) {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; ++i) m.at<double>(i, i) = 1.0;
        return m;
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : depth() == CV_32S ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <class T>
    T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T>
    const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <class T>
    T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T>
    const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    // rows [r0, r1) sharing the storage (cv::Mat::rowRange)
    Mat rowRange(int r0, int r1) const {
        Mat m = *this;
        m.rows = r1 - r0;
        m.data = data ? data + (size_t)r0 * step : nullptr;
        return m;
    }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr<uchar>(r), ptr<uchar>(r), (size_t)cols * elemSize());
        return m;
    }
    // inverse of a small square f64 matrix (Gauss-Jordan with partial pivoting): what T_w_c_.inv() needs (frame.cpp:22,29)
    Mat inv() const {
        const int n = rows;
        Mat a = clone(), b = eye(n, n, CV_64FC1);
        for (int c = 0; c < n; ++c) {
            int p = c;
            for (int r = c + 1; r < n; ++r)
                if (std::fabs(a.at<double>(r, c)) > std::fabs(a.at<double>(p, c))) p = r;
            for (int k = 0; k < n; ++k) {
                std::swap(a.at<double>(c, k), a.at<double>(p, k));
                std::swap(b.at<double>(c, k), b.at<double>(p, k));
            }
            const double d = a.at<double>(c, c);
            for (int k = 0; k < n; ++k) {
                a.at<double>(c, k) /= d;
                b.at<double>(c, k) /= d;
            }
            for (int r = 0; r < n; ++r) {
                if (r == c) continue;
                const double f = a.at<double>(r, c);
                for (int k = 0; k < n; ++k) {
                    a.at<double>(r, k) -= f * a.at<double>(c, k);
                    b.at<double>(r, k) -= f * b.at<double>(c, k);
                }
            }
        }
        return b;
    }
    void copyTo(Mat& dst) const {
        dst.create(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(dst.ptr<uchar>(r), ptr<uchar>(r), (size_t)cols * elemSize());
    }

private:
    int type_ = 0;
    std::shared_ptr<std::vector<uchar>> store_;
};
typedef Mat Mat1b;

}  // namespace cv
#endif  // MVO_HAVE_OPENCV
#endif
