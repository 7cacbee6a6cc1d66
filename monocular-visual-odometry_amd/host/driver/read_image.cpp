// host/driver/read_image.cpp -- test helper: read_image <in.png|pgm|ppm> <out.raw> writes "w h" + the BGR bytes
// basics::imread decoded (tests/test_host_io.py compares them with PIL's decoding).  Needs no GPU.
#include <cstdio>
#include <fstream>

#include "my_slam/basics/image_io.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    cv::Mat img = my_slam::basics::imread(argv[1]);
    if (img.empty()) return 1;
    std::ofstream o(argv[2], std::ios::binary);
    const int wh[2] = {img.cols, img.rows};
    o.write(reinterpret_cast<const char*>(wh), 8);
    for (int y = 0; y < img.rows; ++y) o.write(reinterpret_cast<const char*>(img.ptr<unsigned char>(y)), (std::streamsize)img.cols * 3);
    return 0;
}
