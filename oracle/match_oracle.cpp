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

// FLANN's LSH index as cv::FlannBasedMatcher(new cv::flann::LshIndexParams(tables, key_size, multi_probe_level)) builds
// it (feature_match.cpp:140; FLANN 1.6.10 lsh_index.h / lsh_table.h, not vendored -- restated from the published
// algorithm): every table keys a descriptor by `key_size` of its 256 bits; a query looks into the bucket of its own key
// and of every key at most `multi_probe_level` bit flips away, in every table; the candidates are ranked by exact
// Hamming distance, ties to the lower train index (KNNUniqueResultSet orders by (distance, index)).  FLANN draws the
// key bits with its own unseeded generator, so its choice is not reproducible: this restatement draws them from a seeded
// LCG and is used to QUANTIFY what the approximate search changes against the exact one
// (tests/test_oracle_match.py), not as a bit-exact model of one particular FLANN run.
int orc_match_knn2_lsh(const uint8_t* q, int nq, const uint8_t* t, int nt, int tables, int key_size, int probe_level,
                       uint32_t seed, int32_t* idx, int32_t* dist) {
    if (tables < 1 || key_size < 1 || key_size > 20 || probe_level < 0) return -1;
    uint32_t rng = seed ? seed : 1u;
    auto next = [&]() {
        rng = rng * 1664525u + 1013904223u;
        return rng >> 8;
    };
    std::vector<std::vector<int>> bits(tables);
    for (auto& b : bits) {  // key_size distinct bit positions per table (partial Fisher-Yates over 0..255)
        int perm[256];
        for (int i = 0; i < 256; ++i) perm[i] = i;
        for (int i = 0; i < key_size; ++i) std::swap(perm[i], perm[i + next() % (256 - i)]);
        b.assign(perm, perm + key_size);
        std::sort(b.begin(), b.end());
    }
    auto key_of = [&](const uint8_t* d, const std::vector<int>& b) {
        uint32_t k = 0;
        for (int i = 0; i < key_size; ++i) k |= (uint32_t)((d[b[i] >> 3] >> (b[i] & 7)) & 1) << i;
        return k;
    };
    std::vector<uint32_t> masks;  // every xor mask with at most probe_level bits set
    for (uint32_t m = 0; m < (1u << key_size); ++m)
        if (__builtin_popcount(m) <= probe_level) masks.push_back(m);
    std::vector<std::vector<std::vector<int32_t>>> bucket(tables, std::vector<std::vector<int32_t>>((size_t)1 << key_size));
    for (int tb = 0; tb < tables; ++tb)
        for (int j = 0; j < nt; ++j) bucket[tb][key_of(t + 32 * (size_t)j, bits[tb])].push_back(j);
    std::vector<uint8_t> seen(nt);
    for (int i = 0; i < nq; ++i) {
        std::fill(seen.begin(), seen.end(), 0);
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int tb = 0; tb < tables; ++tb) {
            const uint32_t key = key_of(q + 32 * (size_t)i, bits[tb]);
            for (uint32_t m : masks)
                for (int32_t j : bucket[tb][key ^ m]) {
                    if (seen[j]) continue;
                    seen[j] = 1;
                    const int d = hamming256(q + 32 * (size_t)i, t + 32 * (size_t)j);
                    if (d < d0 || (d == d0 && j < i0)) {
                        d1 = d0, i1 = i0, d0 = d, i0 = j;
                    } else if (d < d1 || (d == d1 && j < i1)) {
                        d1 = d, i1 = j;
                    }
                }
        }
        idx[2 * i] = i0, idx[2 * i + 1] = i1;
        dist[2 * i] = d0, dist[2 * i + 1] = d1;
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

// matchFeatures on a given 2-NN table (methods 1 and 2): the selection rules of feature_match.cpp:140-236 applied to
// the neighbours an exact or an approximate search returned (a query without a neighbour has index -1)
int orc_match_features_from_knn(const int32_t* idx, const int32_t* dist, int n1, int method, double xiang_gao_ratio,
                                double lowe_ratio, orc_dmatch* out, int cap) {
    std::vector<orc_dmatch> matches;
    if (method == 1) {
        double min_dis = 9999999, max_dis = 0;
        for (int i = 0; i < n1; ++i) {
            if (idx[2 * i] < 0) continue;
            double d = (float)dist[2 * i];
            if (d < min_dis) min_dis = d;
            if (d > max_dis) max_dis = d;
        }
        double thr = std::max<float>(min_dis * xiang_gao_ratio, 30.0);
        for (int i = 0; i < n1; ++i)
            if (idx[2 * i] >= 0 && (float)dist[2 * i] < thr) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
    } else if (method == 2) {
        for (int i = 0; i < n1; ++i) {
            if (idx[2 * i + 1] < 0) continue;
            double d = (float)dist[2 * i];
            if (d < lowe_ratio * (float)dist[2 * i + 1]) matches.push_back({i, idx[2 * i], 0, (float)dist[2 * i]});
        }
    } else {
        return -1;
    }
    int n = orc_remove_duplicated_matches(matches.data(), (int)matches.size());
    if (n > cap) return -3;
    std::copy(matches.begin(), matches.begin() + n, out);
    return n;
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
