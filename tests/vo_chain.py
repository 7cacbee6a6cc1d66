"""tests/vo_chain.py -- TEST AID: the run of host/driver/run_vo.cpp composed from the CPU oracle (oracle_py), row by row
the way the reference chains them:

    run_vo.cpp:110-148          imread -> createFrame -> addFrame -> pose history
    vo_addFrame.cpp:9-27        calcKeyPoints / calcDescriptors
    vo_addFrame.cpp:70-124      DOING_TRACKING: poseEstimationPnP_ -> callBundleAdjustment_ -> keyframe insertion
    vo.cpp:16-49, 270-383       map points in view, matchFeatures, solvePnPRansac, inlier bookkeeping
    vo.cpp:384-478              the sliding window (pose-only or full)
    vo_addFrame.cpp:96-118      matchFeatures with the reference keyframe, epipolar inliers, triangulation, culling
    vo.cpp:528-576, 488-526     pushCurrPointsToMap_, optimizeMap_

Nothing of the product is used here: every row is the oracle's.  The one thing the chain cannot know by itself is the
ITERATION ORDER of the host's std::unordered_map<int, MapPoint::Ptr> (the order every candidate / match index refers to,
vo.cpp:24): it is a property of the container, not of the hot path, so the caller passes it in (`map_order`), taken from the
binary's frame log; without it the ids are used in ascending order.
"""
import math

import numpy as np


def read_frame_log(path):
    """The records of run_vo's `save_frame_log_to` -> list (one dict per frame) of raw byte strings by tag."""
    frames = []
    with open(path, "rb") as f:
        while True:
            tag = f.read(4)
            if len(tag) < 4:
                break
            n = int(np.frombuffer(f.read(8), "<i8")[0])
            payload = f.read(n)
            assert len(payload) == n, "truncated frame log"
            tag = tag.decode()
            if tag == "FRAM":
                frames.append({})
            frames[-1][tag] = payload
    return frames


class MapPoint:
    def __init__(self, id_, pos, norm, desc):
        self.id, self.pos, self.norm, self.desc = id_, pos, norm, desc
        self.matched_times = self.visible_times = 1


class Frame:
    def __init__(self, idx, kps, desc):
        self.idx, self.kps, self.desc = idx, kps, desc
        self.xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        self.T = np.eye(4)
        self.conn = {}                      # inliers_to_mappt_connections_: keypoint index -> (pt_ref_idx, pt_map_idx)
        self.rec = {}                       # what the comparison reads


def pre_translate_point3f(p, T):
    """basics::preTranslatePoint3f (opencv_funcs.cpp:67-78): double accumulation in column order, result as float."""
    q = (float(p[0]), float(p[1]), float(p[2]), 1.0)
    out = []
    for r in range(3):
        s = 0.0
        for j in range(4):
            s += float(T[r, j]) * q[j]
        out.append(s)
    return np.array(out, np.float64).astype(np.float32)


class OracleChain:
    def __init__(self, O, K, cols, rows, orb_params, fix_map_points=True, num_ba_frames=5, map_order=None):
        self.O, self.K, self.cols, self.rows, self.p = O, K, cols, rows, orb_params
        self.fix = fix_map_points
        self.num_ba_frames = num_ba_frames
        self.map_order = map_order          # callable(frame index, set of ids) -> ids in the host container's order
        self.map = {}                       # id -> MapPoint
        self.next_point_id = 0
        self.buff = []                      # frames_buff_ (newest last, 20 at most)
        self.ref = self.prev = None
        self.erase_ratio = 0.1
        self.frames = []

    # ---- vo_addFrame.cpp:9-27
    def create_frame(self, idx, img):
        O = self.O
        k = O.calc_keypoints(img, self.p)
        k, d = O.calc_descriptors(img, k, self.p)
        fr = Frame(idx, k, d)
        self.frames.append(fr)
        return fr

    def push_to_buff(self, fr):
        self.buff.append(fr)
        if len(self.buff) > 20:
            self.buff.pop(0)

    # ---- run_vo.cpp (this repository's seeding; see the header of host/driver/run_vo.cpp)
    def seed_first(self, fr, T):
        fr.T = np.array(T, np.float64)
        self.push_to_buff(fr)
        self.ref = self.prev = fr

    def seed_second(self, fr, T):
        fr.T = np.array(T, np.float64)
        self.push_to_buff(fr)
        self.triangulate_with_reference(fr)
        self.push_points_to_map(fr)
        self.ref = self.prev = fr

    # ---- vo.cpp:270-383
    def pose_estimation_pnp(self, fr):
        O, K = self.O, self.K
        ids = sorted(self.map) if self.map_order is None else list(self.map_order(fr.idx, set(self.map)))
        assert sorted(ids) == sorted(self.map), "map_order must enumerate exactly the map's points"
        ids = np.array(ids, np.int64)
        pos = np.array([self.map[i].pos for i in ids], np.float32).reshape(-1, 3)
        vis, px = O.map_in_view(pos, fr.T, K, self.cols, self.rows)
        cand = ids[vis]
        for i in cand:
            self.map[i].visible_times += 1
        cdesc = np.array([self.map[i].desc for i in cand], np.uint8).reshape(-1, 32)
        m = O.match_features(cdesc, fr.desc, 1, 2.0, 1.0, px, fr.xy, 50.0)
        fr.rec.update(map_order=ids, candidates=cand, n_matches=len(m))
        good = len(m) >= 5
        if good:
            p3 = pos[vis][m["queryIdx"]]
            p2 = fr.xy[m["trainIdx"]]
            res = O.solve_pnp_ransac(p3, p2, K, 100, 2.0, 0.999)
            assert res["ok"], "solvePnPRansac found no pose"
            inl = res["inliers"]
            fr.rec["matches_with_map"] = m[inl]
            for q in m[inl]:
                mp = self.map[cand[q["queryIdx"]]]
                mp.matched_times += 1
                fr.conn[int(q["trainIdx"])] = (-1, mp.id)
            T_c_w = np.eye(4)
            T_c_w[:3, :3] = O.rodrigues(res["rvec"])
            T_c_w[:3, 3] = res["tvec"]
            fr.T = O.invert4x4(T_c_w)
            fr.rec["T_pnp"] = fr.T.copy()
            d2 = 0.0
            for i in range(3):
                d = float(fr.T[i, 3]) - float(self.prev.T[i, 3])
                d2 += d * d
            if math.sqrt(d2) >= 0.3:
                good = False
        if not good:
            fr.T = self.prev.T.copy()
        return good

    # ---- vo.cpp:384-478 + g2o_ba.cpp:193-316
    def bundle_adjustment(self):
        total = len(self.buff)
        n_ba = min(self.num_ba_frames, total - 1)
        frames, ep, el, uv, slot = [], [], [], [], {}
        for b in range(total - 1, total - n_ba - 1, -1):
            fr = self.buff[b]
            if len(fr.conn) < 3:
                continue
            f = len(frames)
            frames.append(fr)
            for kpt in sorted(fr.conn):
                mid = fr.conn[kpt][1]
                if mid not in self.map:
                    continue
                ep.append(f)
                el.append(slot.setdefault(mid, len(slot)))
                uv.append(fr.xy[kpt].astype(np.float64))
        if not frames:
            return
        ids = sorted(slot, key=slot.get)
        pts = np.array([self.map[i].pos for i in ids], np.float64).reshape(-1, 3)
        poses = np.stack([fr.T for fr in frames])
        P, X, st = self.O.bundle_adjustment(poses, pts, ep, el, np.array(uv).reshape(-1, 2), self.K["fx"], self.K["cx"],
                                            self.K["cy"], fix_points=self.fix)
        for fr, T in zip(frames, P):
            fr.T = T.copy()
        if not self.fix:
            for i, x in zip(ids, X):
                self.map[i].pos = x.astype(np.float32)

    # ---- vo_commons.cpp:9-15
    @staticmethod
    def motion_from_1_to_2(O, f1, f2):
        Ti = O.invert4x4(f1.T)
        T = np.zeros((4, 4))
        for i in range(4):
            for j in range(4):
                s = 0.0
                for k in range(4):
                    s += float(Ti[i, k]) * float(f2.T[k, j])
                T[i, j] = s
        return T

    def large_move(self, fr):                                  # vo.cpp:247-266
        T = self.motion_from_1_to_2(self.O, self.ref, fr)
        s = 0.0
        for i in range(3):
            s = s + float(T[i, 3]) * float(T[i, 3])
        return math.sqrt(s) > 0.03

    # ---- vo_addFrame.cpp:96-118
    def triangulate_with_reference(self, fr):
        O, K, ref = self.O, self.K, self.ref
        m = O.match_features(ref.desc, fr.desc, 1, 2.0, 1.0, ref.xy, fr.xy, 100.0)
        a, b = ref.xy[m["queryIdx"]], fr.xy[m["trainIdx"]]
        inl = O.find_essential_inliers(a, b, K, 0.999, 1.0)["inliers"]
        mi = m[inl].copy()
        mi["imgIdx"] = -1                                       # cv::DMatch(queryIdx, trainIdx, distance), motion_estimation.cpp:174-179
        T = self.motion_from_1_to_2(O, fr, ref)
        _, p_cur = O.triangulate_points(a[inl], b[inl], K, T[:3, :3], T[:3, 3])
        keep, _ = O.retain_good_triangulation(p_cur, fr.T, ref.T, 1.0, 20.0)
        fr.rec.update(matches_with_ref=m, inliers_matches_with_ref=mi, inliers_matches_for_3d=mi[keep],
                      inliers_pts3d=p_cur[keep])

    # ---- vo.cpp:528-576
    def push_points_to_map(self, fr):
        ref = self.ref
        for dm, p in zip(fr.rec["inliers_matches_for_3d"], fr.rec["inliers_pts3d"]):
            pt_idx, q = int(dm["trainIdx"]), int(dm["queryIdx"])
            if q in ref.conn:
                mid = ref.conn[q][1]
            else:
                w = pre_translate_point3f(p, fr.T)
                n = [float(w[r]) - float(fr.T[r, 3]) for r in range(3)]
                ln = 0.0
                for r in range(3):
                    ln += n[r] * n[r]
                ln = math.sqrt(ln)
                n = np.array([v / ln for v in n])
                mid = self.next_point_id
                self.next_point_id += 1
                self.map[mid] = MapPoint(mid, w, n, fr.desc[pt_idx].copy())
            fr.conn.setdefault(pt_idx, (q, mid))               # unordered_map::insert keeps an existing entry

    # ---- vo.cpp:488-526 (+ frame.cpp:29-36, vo.cpp:578-584)
    def optimize_map(self, fr):
        K = self.K
        Ti = self.O.invert4x4(fr.T)
        for mid in list(self.map):
            mp = self.map[mid]
            pc = pre_translate_point3f(mp.pos, Ti)
            ok = not pc[2] < 0
            if ok:
                u = np.float32(K["fx"] * float(pc[0]) / float(pc[2]) + K["cx"])
                v = np.float32(K["fy"] * float(pc[1]) / float(pc[2]) + K["cy"])
                ok = bool(u > 0 and v > 0 and u < self.cols and v < self.rows)
            if not ok:
                del self.map[mid]
                continue
            ratio = np.float32(mp.matched_times) / np.float32(mp.visible_times)
            if float(ratio) < self.erase_ratio:
                del self.map[mid]
                continue
            n = [float(mp.pos[r]) - float(fr.T[r, 3]) for r in range(3)]
            ln = 0.0
            for r in range(3):
                ln += n[r] * n[r]
            ln = math.sqrt(ln)
            dot = 0.0
            for r in range(3):
                dot += n[r] / ln * float(mp.norm[r])
            ang = math.acos(dot) if -1.0 <= dot <= 1.0 else float("nan")   # std::acos: NaN outside [-1, 1], never > pi/4
            if ang > math.pi / 4.0:
                del self.map[mid]
                continue
        if len(self.map) > 1000:
            self.erase_ratio += 0.05
        else:
            self.erase_ratio = 0.1

    # ---- vo_addFrame.cpp:70-124
    def track(self, fr):
        self.push_to_buff(fr)
        fr.T = self.ref.T.copy()
        good = self.pose_estimation_pnp(fr)
        is_key = False
        if good:
            self.bundle_adjustment()
            if self.large_move(fr):
                self.triangulate_with_reference(fr)
                self.push_points_to_map(fr)
                self.optimize_map(fr)
                self.ref = fr
                is_key = True
        self.prev = fr
        fr.rec.update(good=good, is_keyframe=is_key)
        return good, is_key


def run_oracle_chain(O, images, K, truth, k0, k1, orb_params, fix_map_points=True, map_order=None):
    """The loop of host/driver/run_vo.cpp.  Returns (chain, poses [n, 4, 4] as they stood when each frame was done)."""
    rows, cols = images[0].shape[:2]
    ch = OracleChain(O, K, cols, rows, orb_params, fix_map_points, map_order=map_order)
    history = []
    for i, img in enumerate(images):
        fr = ch.create_frame(i, img)
        if i < k1:
            fr.T = (np.array(truth[k0], np.float64) if i >= k0 else np.eye(4)).copy()
            if i == k0:
                ch.seed_first(fr, truth[k0])
        elif i == k1:
            ch.seed_second(fr, truth[k1])
            fr.rec["map_after"] = {m: ch.map[m].pos.copy() for m in ch.map}
        else:
            _, is_key = ch.track(fr)
            if is_key:
                fr.rec["map_after"] = {m: ch.map[m].pos.copy() for m in ch.map}
        history.append(fr.T.copy())
    return ch, np.stack(history)
