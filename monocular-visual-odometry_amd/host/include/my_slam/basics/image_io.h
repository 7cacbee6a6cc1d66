// my_slam/basics/image_io.h -- what the headless driver needs in place of cv::imread (reference run_vo.cpp:114 reads
// `rgb_%05d.png`): 8-bit PNG (gray, gray+alpha, RGB, RGBA; non-interlaced; zlib inflates the IDAT stream) and binary
// PGM / PPM.  Like cv::imread(path) (IMREAD_COLOR) the result is always 3-channel BGR, 8 bits; an unreadable file gives
// an empty Mat (run_vo.cpp:115-119 stops the run there).  With OpenCV present (MVO_HAVE_OPENCV) use cv::imread.
#ifndef MY_SLAM_IMAGE_IO_H
#define MY_SLAM_IMAGE_IO_H
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "my_slam/common_include.h"

namespace my_slam {
namespace basics {

namespace detail {
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace detail

inline cv::Mat readPng(const vector<unsigned char>& file) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8)) return cv::Mat();
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    vector<unsigned char> idat;
    while (pos + 12 <= file.size()) {
        const uint32_t len = detail::be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        const unsigned char* data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) return cv::Mat();
        if (!std::memcmp(type, "IHDR", 4)) {
            w = detail::be32(data);
            h = detail::be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !ch || interlace) return cv::Mat();  // (palette / 16-bit / Adam7 are not needed here)
    const size_t stride = (size_t)w * ch;
    vector<unsigned char> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return cv::Mat();
    vector<unsigned char> prev(stride, 0), cur(stride);
    cv::Mat img((int)h, (int)w, CV_8UC3);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* line = &raw[(stride + 1) * y];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int v = line[1 + i];
            switch (filter) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += detail::paeth(a, b, c); break;
                default: break;
            }
            cur[i] = (unsigned char)v;
        }
        unsigned char* dst = img.ptr<unsigned char>((int)y);
        for (uint32_t x = 0; x < w; ++x) {
            const unsigned char* px = &cur[(size_t)x * ch];
            if (ch <= 2) {
                dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = px[0];
            } else {  // RGB[A] -> BGR
                dst[3 * x] = px[2];
                dst[3 * x + 1] = px[1];
                dst[3 * x + 2] = px[0];
            }
        }
        prev.swap(cur);
    }
    return img;
}

inline cv::Mat readPnm(const vector<unsigned char>& file) {
    if (file.size() < 8 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) return cv::Mat();
    const int ch = file[1] == '5' ? 1 : 3;
    size_t pos = 2;
    int vals[3], nv = 0;
    while (nv < 3 && pos < file.size()) {
        while (pos < file.size() && (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t')) ++pos;
        if (pos < file.size() && file[pos] == '#') {
            while (pos < file.size() && file[pos] != '\n') ++pos;
            continue;
        }
        int v = 0;
        bool any = false;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') {
            v = 10 * v + (file[pos++] - '0');
            any = true;
        }
        if (!any) return cv::Mat();
        vals[nv++] = v;
    }
    ++pos;  // the single whitespace behind maxval
    const int w = vals[0], h = vals[1];
    if (nv < 3 || vals[2] != 255 || w <= 0 || h <= 0 || pos + (size_t)w * h * ch > file.size()) return cv::Mat();
    cv::Mat img(h, w, CV_8UC3);
    for (int y = 0; y < h; ++y) {
        const unsigned char* src = &file[pos + (size_t)y * w * ch];
        unsigned char* dst = img.ptr<unsigned char>(y);
        for (int x = 0; x < w; ++x) {
            if (ch == 1) {
                dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = src[x];
            } else {
                dst[3 * x] = src[3 * x + 2];
                dst[3 * x + 1] = src[3 * x + 1];
                dst[3 * x + 2] = src[3 * x];
            }
        }
    }
    return img;
}

// cv::imread(path) stand-in: BGR, 8 bits, 3 channels; empty Mat if the file is missing or not one of the formats above
inline cv::Mat imread(const string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return cv::Mat();
    vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    cv::Mat img = readPng(file);
    if (img.empty()) img = readPnm(file);
    return img;
}

}  // namespace basics
}  // namespace my_slam
#endif
