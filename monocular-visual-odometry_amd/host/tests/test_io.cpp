// host/tests/test_io.cpp -- the reference's wire formats through the mirror (my_slam/vo/vo_io.h, basics/config.h):
//   test_io <config.yaml> <traj_in.txt> <traj_out.txt>
// prints the values run_vo.cpp reads from the config (dataset section, intrinsics, a few VO parameters) and copies the
// trajectory through readPoseFromFile / writePoseToFile.  No GPU involved.
#include <cstdio>

#include "my_slam/vo/vo_io.h"

using namespace my_slam;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        basics::Config::setParameterFile(argv[1]);
        const string dataset = basics::Config::get<string>("dataset_name");
        cv::Mat K = vo::readCameraIntrinsics(dataset);
        printf("dataset_name=%s\n", dataset.c_str());
        printf("dataset_dir=%s\n", basics::Config::get<string>(dataset + ".dataset_dir").c_str());
        printf("num_images=%d\n", basics::Config::get<int>(dataset + ".num_images"));
        printf("K=%.10g %.10g %.10g %.10g\n", K.at<double>(0, 0), K.at<double>(1, 1), K.at<double>(0, 2), K.at<double>(1, 2));
        printf("max_number_of_keypoints=%d\n", basics::Config::get<int>("max_number_of_keypoints"));
        printf("scale_factor=%.10g\n", basics::Config::get<double>("scale_factor"));
        printf("lowe_method_dist_ratio_as_int=%d\n", basics::Config::get<int>("lowe_method_dist_ratio"));
        printf("is_ba_fix_map_points=%d\n", (int)basics::Config::getBool("is_ba_fix_map_points"));
        printf("information_matrix=%s\n", basics::Config::get<string>("information_matrix").c_str());
        printf("findEssentialMat_prob=%.10g\n", basics::Config::get<double>("findEssentialMat_prob"));
        vector<string> paths = vo::readImagePaths(basics::Config::get<string>(dataset + ".dataset_dir"), 3, "/rgb_%05d.png");
        printf("image2=%s\n", paths[2].c_str());
        vector<cv::Mat> traj = vo::readPoseFromFile(argv[2]);
        printf("poses=%d\n", (int)traj.size());
        vo::writePoseToFile(argv[3], traj);
        bool threw = false;
        try {
            basics::Config::get<int>("no_such_key");
        } catch (const std::runtime_error&) {
            threw = true;
        }
        printf("missing_key_throws=%d\n", (int)threw);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
