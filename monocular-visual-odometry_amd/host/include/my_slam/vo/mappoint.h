// my_slam/vo/mappoint.h -- MapPoint with the reference's field names (include/my_slam/vo/mappoint.h:14-35,
// src/vo/mappoint.cpp:11-24).
#ifndef MY_SLAM_MAPPOINT_H
#define MY_SLAM_MAPPOINT_H
#include "my_slam/common_include.h"

namespace my_slam {
namespace vo {

class MapPoint {
public:
    typedef std::shared_ptr<MapPoint> Ptr;
    static int& factory_id() {
        static int id = 0;
        return id;
    }
    int id_;
    cv::Point3f pos_;
    cv::Mat norm_;
    vector<unsigned char> color_;  // r,g,b
    cv::Mat descriptor_;           // 1 x 32
    bool good_ = true;
    int matched_times_ = 1, visible_times_ = 1;

    MapPoint(const cv::Point3f& pos, const cv::Mat& descriptor, const cv::Mat& norm, unsigned char r = 0,
             unsigned char g = 0, unsigned char b = 0)
        : id_(factory_id()++), pos_(pos), norm_(norm), color_({r, g, b}), descriptor_(descriptor) {}
    void setPos(const cv::Point3f& pos) { pos_ = pos; }
};

}  // namespace vo
}  // namespace my_slam
#endif
