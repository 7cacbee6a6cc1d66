// my_slam/common_include.h -- what every header of the hot-path mirror needs (reference: include/my_slam/common_include.h).
#ifndef MY_SLAM_COMMON_INCLUDE_H
#define MY_SLAM_COMMON_INCLUDE_H
#include <cmath>
#include <cstdio>
#include <deque>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "my_slam/mini_cv.h"
#include "mvo_hip.h"
#include "mvo_hot_path.h"  // (host/src: the ctx binding of the calling thread, shared with the drop-in translation units)

// (at global scope, like include/my_slam/common_include.h:24-27 of the reference)
using std::cout;
using std::endl;
using std::string;
using std::vector;
#endif
