// my_slam/vo/vo_io.h -- the reference's file formats (src/vo/vo_io.cpp:13-117, config/config.yaml:1-60): image path
// list, camera intrinsics of the selected dataset section, and the trajectory text format (one pose per line:
// tx ty tz R[:,0] R[:,1] R[:,2]) that run_vo writes (`save_predicted_traj_to`) and reads back for the ground truth.
#ifndef MY_SLAM_VO_IO_H
#define MY_SLAM_VO_IO_H
#include <fstream>
#include <sstream>

#include "my_slam/basics/config.h"
#include "my_slam/common_include.h"

namespace my_slam {
namespace vo {

// vo_io.cpp:13-39: dataset_dir + printf-style formatting ("/rgb_%05d.png") for i in [0, num_images)
inline vector<string> readImagePaths(const string& dataset_dir, int num_images, const string& image_formatting,
                                     bool is_print_res = false) {
    vector<string> image_paths;
    for (int i = 0; i < num_images; i++) {
        char buf[512];
        snprintf(buf, sizeof(buf), (dataset_dir + image_formatting).c_str(), i);
        image_paths.push_back(buf);
    }
    if (is_print_res) {
        printf("\nReading from dataset_dir: %s\nNumber of images: %d\n", dataset_dir.c_str(), (int)image_paths.size());
        for (const string& s : image_paths) printf("%s\n", s.c_str());
    }
    return image_paths;
}

// vo_io.cpp:41-49 on the dataset section named by `dataset_name` (run_vo.cpp reads config.get(dataset_name))
inline cv::Mat readCameraIntrinsics(const string& dataset_section) {
    const string p = dataset_section.empty() ? "" : dataset_section + ".";
    cv::Mat K = cv::Mat::eye(3, 3, CV_64FC1);
    K.at<double>(0, 0) = basics::Config::get<double>(p + "camera_info.fx");
    K.at<double>(1, 1) = basics::Config::get<double>(p + "camera_info.fy");
    K.at<double>(0, 2) = basics::Config::get<double>(p + "camera_info.cx");
    K.at<double>(1, 2) = basics::Config::get<double>(p + "camera_info.cy");
    return K;
}

// vo_io.cpp:51-78
inline void writePoseToFile(const string filename, const vector<cv::Mat>& list_T) {
    std::ofstream fout(filename);
    if (!fout.is_open()) {
        printf("my WARNING: failed to store camera trajectory to the wrong file name of:\n    %s\n", filename.c_str());
        return;
    }
    for (const cv::Mat& T : list_T) {
        fout << T.at<double>(0, 3) << " " << T.at<double>(1, 3) << " " << T.at<double>(2, 3) << " ";
        for (int i = 0; i < 3; i++)  // order: 1st column, 2nd column, 3rd column
            for (int j = 0; j < 3; j++) fout << T.at<double>(j, i) << " ";
        fout << '\n';
    }
}

// vo_io.cpp:80-117
inline vector<cv::Mat> readPoseFromFile(const string filename) {
    vector<cv::Mat> list_T;
    std::ifstream fin(filename);
    if (!fin.is_open()) throw std::runtime_error("readPoseFromFile: cannot open " + filename);
    constexpr int kNumValsPerRow = 12;
    double pose[kNumValsPerRow], val;
    int cnt = 0;
    while (fin >> val) {
        pose[cnt++] = val;
        if (cnt == kNumValsPerRow) {
            cnt = 0;
            cv::Mat T = cv::Mat::eye(4, 4, CV_64FC1);
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 3; r++) T.at<double>(r, c) = pose[3 + 3 * c + r];
            for (int r = 0; r < 3; r++) T.at<double>(r, 3) = pose[r];
            list_T.push_back(T);
        }
    }
    return list_T;
}

}  // namespace vo
}  // namespace my_slam
#endif
