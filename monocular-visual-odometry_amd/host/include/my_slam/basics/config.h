// my_slam/basics/config.h -- basics::Config stand-in (reference include/my_slam/basics/config.h:16-50,
// src/basics/config.cpp:11-46).  The reference reads config/config.yaml through cv::FileStorage; here the keys of
// the hot path are set programmatically (Config::set) or parsed from flat "key: value" lines.  get<int> of a real
// value rounds like cv::FileNode does -- that is how lowe_method_dist_ratio 0.8 becomes 1 at feature_match.cpp:138.
#ifndef MY_SLAM_CONFIG_H
#define MY_SLAM_CONFIG_H
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "my_slam/common_include.h"

namespace my_slam {
namespace basics {

class Config {
public:
    static std::map<string, string>& table() {
        static std::map<string, string> t = {
            // config/config.yaml:63-123
            {"number_of_keypoints_to_extract", "8000"}, {"max_number_of_keypoints", "1500"}, {"scale_factor", "1.2"},
            {"level_pyramid", "4"}, {"score_threshold", "20"}, {"xiang_gao_method_match_ratio", "2"},
            {"lowe_method_dist_ratio", "0.8"}, {"method_3_feature_dist_threshold", "50.0"},
            {"kpts_uniform_selection_grid_size", "16"}, {"kpts_uniform_selection_max_pts_per_grid", "8"},
            {"is_enable_ba", "true"}, {"num_prev_frames_to_opti_by_ba", "5"}, {"information_matrix", "1.0 0.0 0.0 1.0"},
            {"is_ba_fix_map_points", "true"}, {"feature_match_method_index_pnp", "1"},
            {"max_matching_pixel_dist_in_pnp", "50"}, {"max_possible_dist_to_prev_keyframe", "0.3"}, {"max_matching_pixel_dist_in_triangulation", "100"},
            {"findEssentialMat_prob", "0.999"}, {"findEssentialMat_threshold", "1.0"}, {"min_triang_angle", "1.0"},
            {"max_ratio_between_max_angle_and_median_angle", "20"}, {"min_dist_between_two_keyframes", "0.03"}};
        return t;
    }
    static void set(const string& key, const string& value) { table()[key] = value; }
    // YAML subset of config/config.yaml: "key: value" lines, '#' comments, quotes stripped; one level of nesting
    // (the dataset sections "matlab:" / "fr1_desk:", config.yaml:12-47) is stored as "section.key".
    static void setParameterFile(const string& filename) {
        std::ifstream f(filename);
        if (!f) throw std::runtime_error("parameter file " + filename + " does not exist.");
        string line, section;
        auto trim = [](string& s) {
            size_t a = s.find_first_not_of(" \t\""), b = s.find_last_not_of(" \t\"\r");
            s = a == string::npos ? "" : s.substr(a, b - a + 1);
        };
        while (std::getline(f, line)) {
            size_t h = line.find('#');
            if (h != string::npos) line.erase(h);
            size_t c = line.find(':');
            if (c == string::npos || line.empty() || line[0] == '%') continue;
            const bool nested = line[0] == ' ' || line[0] == '\t';
            string k = line.substr(0, c), v = line.substr(c + 1);
            trim(k);
            trim(v);
            if (k.empty()) continue;
            if (!nested) section.clear();
            if (v.empty()) {  // "matlab:" opens a section
                if (!nested) section = k;
                continue;
            }
            table()[nested && !section.empty() ? section + "." + k : k] = v;
        }
    }
    static bool has(const string& key) { return table().find(key) != table().end(); }
    template <typename T>
    static T get(const string& key);
    static bool getBool(const string& key) {
        string v = raw(key);
        return v == "true" || v == "1";
    }

private:
    static const string& raw(const string& key) {
        auto it = table().find(key);
        if (it == table().end()) throw std::runtime_error("Key " + key + " does not exist");  // config.cpp:34-35
        return it->second;
    }
};
template <>
inline double Config::get<double>(const string& key) { return std::atof(raw(key).c_str()); }
template <>
inline float Config::get<float>(const string& key) { return (float)std::atof(raw(key).c_str()); }
template <>
inline int Config::get<int>(const string& key) { return (int)std::lrint(std::atof(raw(key).c_str())); }
template <>
inline string Config::get<string>(const string& key) { return raw(key); }

inline vector<double> str2vecdouble(const string& s) {
    std::istringstream is(s);
    vector<double> v;
    double d;
    while (is >> d) v.push_back(d);
    return v;
}

}  // namespace basics
}  // namespace my_slam
#endif
