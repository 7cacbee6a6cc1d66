// oracle/match_oracle.cpp -- CPU ORACLE (test infrastructure, see oracle.h): scalar restatement of
// geometry::matchFeatures / matchByRadiusAndBruteForce / removeDuplicatedMatches
// (reference src/geometry/feature_match.cpp:86-260) and of cv::BFMatcher(NORM_HAMMING)::knnMatch
// (not vendored; tie rule per SURVEY.md Appendix A.2).  PARITY UNPINNED (see oracle.h).
#include "oracle.h"

#include <algorithm>
#include <climits>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace {

inline int hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int k = 0; k < 4; ++k) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * k, 8);
        std::memcpy(&y, b + 8 * k, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

}  // namespace

extern "C" {

// batchDistance knn: scan trains in index order; a candidate enters only if strictly smaller than
// the current k-th best and is inserted after entries with distance <= its own -> equal distances
// keep the lower train index first.
int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    for (int i = 0; i < nq; ++i) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; ++j) {
            int d = hamming256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < d0) {
                d1 = d0;
                i1 = i0;
                d0 = d;
                i0 = j;
            } else if (d < d1) {
                d1 = d;
                i1 = j;
            }
        }
        idx[2 * i] = i0;
        idx[2 * i + 1] = i1;
        dist[2 * i] = d0;
        dist[2 * i + 1] = d1;
    }
    return nq;
}

// feature_match.cpp:86-124.  The reference's feature distance is sum|a-b| / 32 in double and the
// comparison is a strict '<' (first minimum wins); comparing the integer sums is equivalent.
int orc_match_radius_l1(const uint8_t* q, const float* qxy, int nq, const uint8_t* t, const float* txy,
                        int nt, float max_px, int32_t* idx, int32_t* sum) {
    float r2 = max_px * max_px;
    for (int i = 0; i < nq; ++i) {
        float x = qxy[2 * i], y = qxy[2 * i + 1];
        int best = INT_MAX, bi = -1;
        for (int j = 0; j < nt; ++j) {
            float x2 = txy[2 * j], y2 = txy[2 * j + 1];
            if ((x - x2) * (x - x2) + (y - y2) * (y - y2) <= r2) {
                int s = 0;
                for (int k = 0; k < 32; ++k) {
                    int a = q[32 * (size_t)i + k], b = t[32 * (size_t)j + k];
                    s += a > b ? a - b : b - a;
                }
                if (s < best) {
                    best = s;
                    bi = j;
                }
            }
        }
        idx[i] = bi;
        sum[i] = best;
    }
    return nq;
}

int orc_remove_duplicated_matches(orc_dmatch* m, int n) {
    std::vector<orc_dmatch> v(m, m + n);
    std::sort(v.begin(), v.end(),
              [](const orc_dmatch& a, const orc_dmatch& b) { return a.trainIdx < b.trainIdx; });
    std::vector<orc_dmatch> res;
    if (!v.empty()) res.push_back(v[0]);
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i].trainIdx != v[i - 1].trainIdx) res.push_back(v[i]);
    std::copy(res.begin(), res.end(), m);
    return (int)res.size();
}

int orc_match_features(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int method,
                       double xiang_gao_ratio, double lowe_ratio, const float* xy1, const float* xy2,
                       float max_px, orc_dmatch* out, int cap) {
    std::vector<orc_dmatch> matches;
    double min_dis = 9999999, max_dis = 0;
    if (method == 1 || method == 3) {
        std::vector<orc_dmatch> all;
        if (method == 3) {
            std::vector<int32_t> idx(n1), sum(n1);
            orc_match_radius_l1(d1, xy1, n1, d2, xy2, n2, max_px, idx.data(), sum.data());
            for (int i = 0; i < n1; ++i)
                if (idx[i] >= 0) all.push_back({i, idx[i], -1, (float)((double)sum[i] / 32)});
        } else {
            // FLANN-LSH 1-NN replaced by the exact nearest neighbour
            std::vector<int32_t> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
            orc_match_knn2(d1, n1, d2, n2, idx.data(), dist.data());
            for (int i = 0; i < n1; ++i)
                if (idx[2 * i] >= 0) all.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
        for (const orc_dmatch& m : all) {
            double dist = m.distance;
            if (dist < min_dis) min_dis = dist;
            if (dist > max_dis) max_dis = dist;
        }
        double thr = std::max<float>(min_dis * xiang_gao_ratio, 30.0);
        for (const orc_dmatch& m : all)
            if (m.distance < thr) matches.push_back(m);
    } else if (method == 2) {
        std::vector<int32_t> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
        orc_match_knn2(d1, n1, d2, n2, idx.data(), dist.data());
        for (int i = 0; i < n1; ++i) {
            if (idx[2 * i + 1] < 0) continue;  // reference reads knn[i][1] out of bounds when nt < 2
            double d = (float)dist[2 * i];
            if (d < lowe_ratio * (float)dist[2 * i + 1]) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
    } else {
        return -1;  // reference throws std::runtime_error (feature_match.cpp:225)
    }
    int n = orc_remove_duplicated_matches(matches.data(), (int)matches.size());
    if (n > cap) return -3;
    std::copy(matches.begin(), matches.begin() + n, out);
    return n;
}

}  // extern "C"
