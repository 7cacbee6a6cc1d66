// my_slam/vo/map.h -- Map with the reference's members (include/my_slam/vo/map.h:14-27, src/vo/map.cpp:9-48): the
// BA window marshalling reads map_points_ through it.
#ifndef MY_SLAM_MAP_H
#define MY_SLAM_MAP_H
#include "my_slam/common_include.h"
#include "my_slam/vo/frame.h"
#include "my_slam/vo/mappoint.h"

namespace my_slam {
namespace vo {

class Map {
public:
    typedef std::shared_ptr<Map> Ptr;
    std::unordered_map<int, Frame::Ptr> keyframes_;
    std::unordered_map<int, MapPoint::Ptr> map_points_;

    void insertKeyFrame(Frame::Ptr frame) { keyframes_[frame->id_] = frame; }
    void insertMapPoint(MapPoint::Ptr map_point) { map_points_[map_point->id_] = map_point; }
    Frame::Ptr findKeyFrame(int frame_id) {
        auto it = keyframes_.find(frame_id);
        return it == keyframes_.end() ? Frame::Ptr() : it->second;
    }
    bool hasKeyFrame(int frame_id) { return keyframes_.find(frame_id) != keyframes_.end(); }
};

}  // namespace vo
}  // namespace my_slam
#endif
