"""mvo_amd -- MI355X-native per-frame VO hot path (ORB extract + Hamming match + sliding-window BA).

This package is plumbing around ``csrc/libmvo_hip.so`` (hand-written HIP kernels behind the C-ABI declared
in ``include/mvo_hip.h``).  The Python layer only mirrors the reference's *interface* for this path
(``my_slam::geometry`` / ``my_slam::optimization`` free functions, same names and argument meaning) so that
tests read like the reference's own call sites; the product host code for a C++ caller is the header-only
adapter under ``host/include/my_slam``.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, construction fails loudly.
The directory name contains a '-', so import it through ``__graft_entry__.load_package()`` (module name
``mvo_amd``).
"""
import ctypes as C
import os

import numpy as np

from . import synth  # noqa: F401  (synthetic workloads)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmvo_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mvo_hip.h")

MVO_OK, MVO_ERR_INVALID, MVO_ERR_NO_DEVICE, MVO_ERR_CAPACITY, MVO_ERR_HIP, MVO_ERR_STATE = 0, -1, -2, -3, -4, -5

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])   # cv::KeyPoint
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
CANDIDATE_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("level_score", "<i4"), ("harris", "<f4"), ("angle", "<f4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("fast_threshold", C.c_int32), ("max_keypoints", C.c_int32), ("grid_size", C.c_int32),
                ("grid_max_per_cell", C.c_int32), ("pyramid_interpolation", C.c_int32)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_T_w_c", C.c_void_p), ("points", C.c_void_p), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("edge_uv", C.c_void_p), ("focal", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("info", C.c_double * 4),
                ("huber_delta", C.c_double), ("fix_points", C.c_int32), ("pose_fixed", C.c_void_p),
                ("max_iterations", C.c_int32)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("terminated", C.c_int32), ("failed_solves", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("stale_steps", C.c_int32), ("reserved_", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


class MvoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmvo_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """Loads csrc/libmvo_hip.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP extension %s is missing -- run __graft_entry__.build() (hipcc, gfx950). "
                          "There is no CPU fallback." % LIB_PATH)
    try:
        # torch bundles its own libamdhip64; when both end up in one process the FIRST one loaded must be
        # torch's (otherwise torch.cuda reports "No HIP GPUs are available"), so import it before dlopen.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.mvo_last_error.restype = C.c_char_p
    lib.mvo_destroy.restype = None
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _image_args(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        h, w = img.shape
        ch = 1
    else:
        h, w, ch = img.shape
    return img, w, h, w * ch, ch


class Context:
    """One mvo_ctx (= one HIP stream).  Not thread-safe; use one per host thread."""

    def __init__(self, device=0, **orb_params):
        self.lib = load_library()
        h = C.c_void_p()
        r = self.lib.mvo_create(C.byref(h), int(device))
        if r != MVO_OK:
            raise MvoError(r, "mvo_create failed (no usable HIP device?) -- there is no CPU fallback")
        self.h = h
        self.device = int(device)
        self.params = dict(nfeatures=8000, scale_factor=1.2, nlevels=4, fast_threshold=20, max_keypoints=1500,
                           grid_size=16, grid_max_per_cell=8,  # config/config.yaml:65-69,94-95
                           pyramid_interpolation=1)             # cv::ORB of OpenCV >= 3.4: INTER_LINEAR_EXACT
        if orb_params:
            self.orb_configure(**orb_params)

    def sibling(self):
        """A second context of this host thread on THIS context's stream (mvo_create_sibling); close it before its parent."""
        c = Context.__new__(Context)
        c.lib = self.lib
        h = C.c_void_p()
        r = self.lib.mvo_create_sibling(self.h, C.byref(h))
        if r != MVO_OK:
            raise MvoError(r, "mvo_create_sibling failed")
        c.h, c.device, c.params = h, self.device, dict(self.params)
        c._parent = self   # (keeps the parent alive as long as the sibling)
        return c

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != MVO_OK:
            raise MvoError(r, (self.lib.mvo_last_error(self.h) or b"").decode())

    # ---- extraction
    def orb_configure(self, **kw):
        new = dict(self.params)
        new.update(kw)
        p = OrbParams(**new)
        self._chk(self.lib.mvo_orb_configure(self.h, C.byref(p)))
        self.params = new

    def calc_keypoints(self, image, cap=None):
        """geometry::calcKeyPoints (feature_match.cpp:11-36)."""
        img, w, h, stride, ch = _image_args(image)
        self._w, self._h = w, h
        cap = cap or (self.params["max_keypoints"] + 16)
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = C.c_int()
        self._chk(self.lib.mvo_calc_keypoints(self.h, _p(img), w, h, stride, ch, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def calc_keypoints_dev(self, d_ptr, w, h, stride, ch, cap=None):
        self._w, self._h = w, h
        cap = cap or (self.params["max_keypoints"] + 16)
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = C.c_int()
        self._chk(self.lib.mvo_calc_keypoints_dev(self.h, C.c_void_p(d_ptr), w, h, stride, ch, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def calc_descriptors(self, image, kps, reuse_pyramid=False, want_rgb=False):
        """geometry::calcDescriptors (feature_match.cpp:38-49): returns (keypoints', descriptors[, rgb])."""
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        n = C.c_int(len(kps))
        desc = np.zeros((max(len(kps), 1), 32), np.uint8)
        rgb = np.zeros((max(len(kps), 1), 3), np.uint8) if want_rgb else None
        if image is None:
            img, w, h, stride, ch = None, self._w, self._h, 0, 1
        else:
            img, w, h, stride, ch = _image_args(image)
        self._chk(self.lib.mvo_calc_descriptors(self.h, _p(img), w, h, stride, ch, int(reuse_pyramid), _p(kps),
                                                C.byref(n), _p(desc), _p(rgb)))
        if want_rgb:
            return kps[:n.value].copy(), desc[:n.value].copy(), rgb[:n.value].copy()
        return kps[:n.value].copy(), desc[:n.value].copy()

    def calc_descriptors_dev(self, kps, want_host=True):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        n = C.c_int(len(kps))
        desc = np.zeros((max(len(kps), 1), 32), np.uint8) if want_host else None
        dptr = C.c_void_p()
        self._chk(self.lib.mvo_calc_descriptors_dev(self.h, _p(kps), C.byref(n), _p(desc), C.byref(dptr)))
        return kps[:n.value].copy(), (desc[:n.value].copy() if want_host else None), dptr.value

    def select_uniform_kpts_by_grid(self, kps, image_rows, image_cols):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        n = C.c_int(len(kps))
        self._chk(self.lib.mvo_select_uniform_kpts_by_grid(self.h, _p(kps), C.byref(n), image_rows, image_cols))
        return kps[:n.value].copy()

    # ---- matching
    def match_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.zeros((len(q), 2), np.int32)
        dist = np.zeros((len(q), 2), np.int32)
        self._chk(self.lib.mvo_match_knn2(self.h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)))
        return idx, dist

    def match_knn2_dev(self, d_q, nq, d_t, nt):
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.int32)
        self._chk(self.lib.mvo_match_knn2_dev(self.h, C.c_void_p(d_q), nq, C.c_void_p(d_t), nt, _p(idx), _p(dist)))
        return idx, dist

    def match_radius_l1(self, q, qxy, t, txy, max_px):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
        txy = np.ascontiguousarray(txy, np.float32).reshape(-1, 2)
        idx = np.zeros(len(q), np.int32)
        s = np.zeros(len(q), np.int32)
        self._chk(self.lib.mvo_match_radius_l1(self.h, _p(q), _p(qxy), len(q), _p(t), _p(txy), len(t),
                                               C.c_float(max_px), _p(idx), _p(s)))
        return idx, s

    def match_features(self, d1, d2, method=1, xiang_gao_ratio=2.0, lowe_ratio=1.0, xy1=None, xy2=None, max_px=0.0):
        """geometry::matchFeatures (feature_match.cpp:126-239); ratios as the reference latches them."""
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        if xy1 is not None:
            xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
            xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        out = np.zeros(max(len(d1), 1), DMATCH_DTYPE)
        n = C.c_int()
        r = self.lib.mvo_match_features(self.h, _p(d1), len(d1), _p(d2), len(d2), int(method),
                                        C.c_double(xiang_gao_ratio), C.c_double(lowe_ratio), _p(xy1), _p(xy2),
                                        C.c_float(max_px), _p(out), len(out), C.byref(n))
        if r == MVO_ERR_INVALID and method not in (1, 2, 3):
            # the reference throws std::runtime_error here (feature_match.cpp:225)
            raise RuntimeError("feature_match.cpp::matchFeatures: wrong method index.")
        self._chk(r)
        return out[:n.value].copy()

    def match_features_dev(self, d_d1, n1, d_d2, n2, method=2, xiang_gao_ratio=2.0, lowe_ratio=1.0):
        out = np.zeros(max(n1, 1), DMATCH_DTYPE)
        n = C.c_int()
        self._chk(self.lib.mvo_match_features_dev(self.h, C.c_void_p(d_d1), n1, C.c_void_p(d_d2), n2, int(method),
                                                  C.c_double(xiang_gao_ratio), C.c_double(lowe_ratio), _p(out),
                                                  len(out), C.byref(n)))
        return out[:n.value].copy()

    # ---- bundle adjustment
    def _ba_problem(self, poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                    fix_points, pose_fixed, max_iterations):
        keep = dict(poses=np.ascontiguousarray(poses, np.float64).reshape(-1, 16).copy(),
                    points=np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy(),
                    ep=np.ascontiguousarray(edge_pose, np.int32), el=np.ascontiguousarray(edge_point, np.int32),
                    uv=np.ascontiguousarray(edge_uv, np.float64).reshape(-1, 2),
                    pf=None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8))
        pr = BaProblem()
        pr.n_poses, pr.n_points, pr.n_edges = len(keep["poses"]), len(keep["points"]), len(keep["ep"])
        pr.pose_T_w_c, pr.points = keep["poses"].ctypes.data, keep["points"].ctypes.data
        pr.edge_pose, pr.edge_point, pr.edge_uv = keep["ep"].ctypes.data, keep["el"].ctypes.data, keep["uv"].ctypes.data
        pr.focal, pr.cx, pr.cy = focal, cx, cy
        pr.info = (C.c_double * 4)(*np.asarray(info, np.float64).ravel())
        pr.huber_delta = huber_delta
        pr.fix_points = int(bool(fix_points))
        pr.pose_fixed = None if keep["pf"] is None else keep["pf"].ctypes.data
        pr.max_iterations = int(max_iterations)
        return pr, keep

    def ba_prepare(self, poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info=(1, 0, 0, 1),
                   huber_delta=1.0, fix_points=False, pose_fixed=None, max_iterations=50):
        """Uploads a window once; returns an opaque handle for ba_solve_resident / ba_fetch / ba_release."""
        pr, keep = self._ba_problem(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                                    fix_points, pose_fixed, max_iterations)
        h = C.c_void_p()
        self._chk(self.lib.mvo_ba_prepare(self.h, C.byref(pr), C.byref(h)))
        return (h, pr.n_poses, pr.n_points)

    def ba_solve_resident(self, handle):
        self._chk(self.lib.mvo_ba_solve_resident(self.h, handle[0]))

    def ba_fetch(self, handle, want_points=True):
        h, F, L = handle
        poses = np.zeros((F, 16))
        points = np.zeros((L, 3)) if want_points else None
        st = BaStats()
        self._chk(self.lib.mvo_ba_fetch(self.h, h, _p(poses), _p(points), C.byref(st)))
        return poses.reshape(-1, 4, 4), points, {k: getattr(st, k) for k, _ in BaStats._fields_}

    def ba_release(self, handle):
        self.lib.mvo_ba_release.restype = None
        self.lib.mvo_ba_release(self.h, handle[0])

    def bundle_adjustment(self, poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info=(1, 0, 0, 1),
                          huber_delta=1.0, fix_points=False, pose_fixed=None, max_iterations=50):
        """optimization::bundleAdjustment on flattened arrays.  Returns (poses [F,4,4], points [L,3], stats)."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 16).copy()
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
        ep = np.ascontiguousarray(edge_pose, np.int32)
        el = np.ascontiguousarray(edge_point, np.int32)
        uv = np.ascontiguousarray(edge_uv, np.float64).reshape(-1, 2)
        pf = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
        pr = BaProblem()
        pr.n_poses, pr.n_points, pr.n_edges = len(poses), len(points), len(ep)
        pr.pose_T_w_c, pr.points = poses.ctypes.data, points.ctypes.data
        pr.edge_pose, pr.edge_point, pr.edge_uv = ep.ctypes.data, el.ctypes.data, uv.ctypes.data
        pr.focal, pr.cx, pr.cy = focal, cx, cy
        pr.info = (C.c_double * 4)(*np.asarray(info, np.float64).ravel())
        pr.huber_delta = huber_delta
        pr.fix_points = int(bool(fix_points))
        pr.pose_fixed = None if pf is None else pf.ctypes.data
        pr.max_iterations = int(max_iterations)
        st = BaStats()
        self._chk(self.lib.mvo_bundle_adjustment(self.h, C.byref(pr), C.byref(st)))
        return poses.reshape(-1, 4, 4), points, {k: getattr(st, k) for k, _ in BaStats._fields_}

    def ba_solve_batch(self, problems, **kw):
        """mvo_ba_solve_batch: `problems` = list of argument tuples of bundle_adjustment(); the windows are solved in one
        grid (8 per launch).  Returns a list of (poses, points, stats)."""
        prs = (BaProblem * len(problems))()
        keeps = []
        for i, a in enumerate(problems):
            d = dict(info=(1, 0, 0, 1), huber_delta=1.0, fix_points=False, pose_fixed=None, max_iterations=50)
            d.update(kw)
            pr, keep = self._ba_problem(*a, d["info"], d["huber_delta"], d["fix_points"], d["pose_fixed"], d["max_iterations"])
            prs[i] = pr
            keeps.append(keep)
        sts = (BaStats * len(problems))()
        self._chk(self.lib.mvo_ba_solve_batch(self.h, prs, len(problems), sts))
        return [(k["poses"].reshape(-1, 4, 4), k["points"], {f: getattr(sts[i], f) for f, _ in BaStats._fields_})
                for i, k in enumerate(keeps)]

    def ba_set_mode(self, mode):
        """mvo_ba_set_mode: "latency" (default, ~300 observations per workgroup), "throughput" (~670, resident grid under load) or
        "shared" (the throughput cut on the launch path only)."""
        self._chk(self.lib.mvo_ba_set_mode(self.h, {"latency": 0, "throughput": 1, "shared": 2}[mode]))

    def ba_trace_enable(self, on=True):
        self._chk(self.lib.mvo_debug_ba_trace_enable(self.h, int(on)))

    def ba_trace(self, handle=None, raw_rows=0):
        """LM trace of the last solve: rows {lambda, chi2, rho, accepted} per trial (raw_rows > 0: that many raw rows)."""
        out = np.zeros((512, 4))
        n = C.c_int()
        self._chk(self.lib.mvo_debug_get_ba_trace(self.h, handle[0] if handle else None, _p(out),
                                                  -int(raw_rows) if raw_rows else 512, C.byref(n)))
        return out[:n.value].copy()

    def ba_plan(self, handle=None):
        """Summation plan of the last window: dict(wgs, nsplit, groups, wg_pt_start)."""
        g, ns = C.c_int(), C.c_int()
        pt = np.zeros(257, np.int32)
        self._chk(self.lib.mvo_debug_get_ba_plan(self.h, handle[0] if handle else None, C.byref(g), C.byref(ns), _p(pt), 257))
        return dict(wgs=g.value, nsplit=ns.value & 0xffff, groups=max(1, ns.value >> 16), wg_pt_start=pt[:g.value + 1].copy())

    # ---- tracking rows (vo.cpp:16-49, 270-357)
    def map_create(self):
        m = C.c_void_p()
        self._chk(self.lib.mvo_map_create(self.h, C.byref(m)))
        return m

    def map_release(self, m):
        self.lib.mvo_map_release.restype = None
        self.lib.mvo_map_release(self.h, m)

    def map_upload(self, m, pos, desc):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        assert len(pos) == len(desc)
        self._chk(self.lib.mvo_map_upload(self.h, m, _p(pos), _p(desc), len(pos)))

    def map_update_positions(self, m, pos, first=0):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        self._chk(self.lib.mvo_map_update_positions(self.h, m, _p(pos), int(first), len(pos)))

    def map_points_in_view(self, m, T_w_c, K, cols, rows, cap):
        """-> (idx [n] int32, px [n,2] float32, device pointer of the gathered n x 32 descriptors)."""
        T = np.ascontiguousarray(T_w_c, np.float64).reshape(4, 4)
        idx = np.zeros(max(cap, 1), np.int32)
        px = np.zeros((max(cap, 1), 2), np.float32)
        n = C.c_int()
        d = C.c_void_p()
        self._chk(self.lib.mvo_map_points_in_view(self.h, m, _p(T), C.c_double(K["fx"]), C.c_double(K["fy"]),
                                                  C.c_double(K["cx"]), C.c_double(K["cy"]), int(cols), int(rows),
                                                  _p(idx), _p(px), int(cap), C.byref(n), C.byref(d)))
        return idx[:n.value].copy(), px[:n.value].copy(), d.value

    def solve_pnp_ransac(self, pts3d, pts2d, K, iterations=100, reprojection_error=2.0, confidence=0.999):
        """cv::solvePnPRansac as vo.cpp:326-329 calls it -> dict(ok, rvec, tvec, inliers)."""
        p3 = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3)
        p2 = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
        assert len(p3) == len(p2)
        n = len(p3)
        rvec, tvec = np.zeros(3), np.zeros(3)
        inl = np.zeros(max(n, 1), np.int32)
        n_inl, found = C.c_int(), C.c_int()
        self._chk(self.lib.mvo_solve_pnp_ransac(self.h, _p(p3), _p(p2), n, C.c_double(K["fx"]), C.c_double(K["fy"]),
                                                C.c_double(K["cx"]), C.c_double(K["cy"]), int(iterations),
                                                C.c_float(reprojection_error), C.c_double(confidence), _p(rvec),
                                                _p(tvec), _p(inl), len(inl), C.byref(n_inl), C.byref(found)))
        return dict(ok=bool(found.value), rvec=rvec, tvec=tvec, inliers=inl[:n_inl.value].copy())

    def triangulate_points(self, kp_prev, kp_curr, K, R, t):
        """geometry::helperTriangulatePoints -> (points in the previous camera frame, returned points), n x 3 f32."""
        a = np.ascontiguousarray(kp_prev, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(kp_curr, np.float32).reshape(-1, 2)
        assert len(a) == len(b)
        R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        n = len(a)
        pp, pc = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self._chk(self.lib.mvo_triangulate_points(self.h, _p(a), _p(b), n, C.c_double(K["fx"]), C.c_double(K["fy"]),
                                                  C.c_double(K["cx"]), C.c_double(K["cy"]), _p(R), _p(t), _p(pp), _p(pc)))
        return pp, pc

    def find_essential_inliers(self, kp_prev, kp_curr, K, prob=0.999, threshold=1.0):
        """geometry::helperFindInlierMatchesByEpipolarCons -> ascending inlier indices into the matches."""
        a = np.ascontiguousarray(kp_prev, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(kp_curr, np.float32).reshape(-1, 2)
        assert len(a) == len(b)
        n = len(a)
        inl = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int()
        self._chk(self.lib.mvo_find_essential_inliers(self.h, _p(a), _p(b), n, C.c_double(K["fx"]), C.c_double(K["fy"]),
                                                      C.c_double(K["cx"]), C.c_double(K["cy"]), C.c_double(prob),
                                                      C.c_double(threshold), _p(inl), len(inl), C.byref(cnt)))
        return inl[:cnt.value].copy()

    def debug_essential(self):
        counts = np.zeros((1000, 10), np.int32)
        info = np.zeros(5, np.int32)
        it = self.lib.mvo_debug_get_essential(self.h, _p(counts), 1000, _p(info))
        if it < 0:
            self._chk(it)
        return dict(counts=counts[:it].copy(), best_iter=int(info[0]), best_model=int(info[1]), iters_run=int(info[2]),
                    evaluated=int(info[3]))

    def debug_pnp(self, cap=4096):
        models = np.zeros((cap, 12))
        counts = np.zeros(cap, np.int32)
        info = np.zeros(6, np.int32)
        h = self.lib.mvo_debug_get_pnp(self.h, _p(models), _p(counts), cap, _p(info))
        if h < 0:
            self._chk(h)
        return dict(models=models[:h].copy(), counts=counts[:h].copy(), best_iter=int(info[0]), iters_run=int(info[1]),
                    dlt=int(info[2]), lm_iters=int(info[3]), lm_evals=int(info[4]), n_hyp=int(info[5]))

    # ---- measurement / debug
    def last_error(self):
        return (self.lib.mvo_last_error(self.h) or b"").decode() if self.h else ""

    def synchronize(self):
        self._chk(self.lib.mvo_synchronize(self.h))

    def profile_enable(self, on=True):
        self._chk(self.lib.mvo_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(self.lib.mvo_profile_reset(self.h))

    def profile_get(self):
        arr = (KernelTime * 64)()
        n = self.lib.mvo_profile_get(self.h, arr, 64)
        return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(min(n, 64))}

    def ba_launch_stats(self, reset=False):
        a, b, ms = C.c_longlong(), C.c_longlong(), C.c_double()
        rw, rs = C.c_longlong(), C.c_longlong()
        rc = C.c_double()
        self.lib.mvo_debug_ba_resident_stats(self.device, C.byref(rw), C.byref(rs))   # (before a reset clears them)
        self.lib.mvo_debug_ba_resident_cycles(self.device, C.byref(rc))
        self.lib.mvo_ba_launch_stats(self.device, C.byref(a), C.byref(b), C.byref(ms), int(reset))
        return dict(launches=a.value, windows=b.value, ms=ms.value, resident_windows=rw.value, resident_grid_starts=rs.value,
                    resident_cycles=rc.value)

    def ba_service_times(self):
        """Wall-clock of the BA launch thread's stages since the last stats reset (ms)."""
        t = (C.c_double * 5)()
        self.lib.mvo_debug_ba_service_times(self.device, t)
        return dict(zip(("wait_for_work", "wait_for_batch", "issue", "wait_for_kernel", "publish"), [round(v, 3) for v in t]))

    def debug_ba_phases(self):
        arr = (C.c_longlong * 16)()
        g = C.c_int()
        self._chk(self.lib.mvo_debug_get_ba_phases(self.h, arr, 16, C.byref(g)))
        names = ["lin", "hpp", "pt+xchg", "t1", "schur", "publish+bar", "assemble", "ldlt", "backsub", "chi2",
                 "chi2xchg", "total", "schur.loop", "schur.wait", "schur.acc", "x15"]
        return {"wgs": g.value, **{n: arr[i] for i, n in enumerate(names)}}

    def debug_level(self, level, blurred=False):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.mvo_debug_get_level(self.h, level, int(blurred), None, 0, C.byref(w), C.byref(h), C.byref(s)))
        buf = np.zeros((h.value + 64, s.value), np.uint8)
        self._chk(self.lib.mvo_debug_get_level(self.h, level, int(blurred), _p(buf), buf.size, None, None, None))
        return buf[:, :w.value + 64].copy()

    def debug_candidates(self):
        n = C.c_int()
        self._chk(self.lib.mvo_debug_get_candidates(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), CANDIDATE_DTYPE)
        self._chk(self.lib.mvo_debug_get_candidates(self.h, _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()


def set_wait_policy(device, policy):
    """How host threads wait for `device`: "spin" | "yield" | "block" | "auto" (mvo_set_wait_policy)."""
    r = load_library().mvo_set_wait_policy(int(device), {"auto": 0, "spin": 1, "yield": 2, "block": 3}[policy])
    if r != MVO_OK:
        raise MvoError(r, "mvo_set_wait_policy")


def debug_set(key, value):
    r = load_library().mvo_debug_set(key.encode(), int(value))
    if r != MVO_OK:
        raise MvoError(r, "unknown debug key")


def rodrigues(rvec):
    """cv::Rodrigues(rvec) -> 3x3 (vo.cpp:334)."""
    r = np.ascontiguousarray(rvec, np.float64).reshape(3)
    R = np.zeros((3, 3))
    if load_library().mvo_rodrigues(_p(r), _p(R)) != MVO_OK:
        raise MvoError(MVO_ERR_INVALID, "mvo_rodrigues")
    return R


def retain_good_triangulation(pts3d_in_curr, T_w_c_curr, T_w_c_ref, min_triang_angle=1.0, max_ratio_to_median=20.0):
    """VisualOdometry::retainGoodTriangulationResult_ -> (kept indices, all angles in degrees)."""
    p = np.ascontiguousarray(pts3d_in_curr, np.float32).reshape(-1, 3)
    Tc = np.ascontiguousarray(T_w_c_curr, np.float64).reshape(4, 4)
    Tr = np.ascontiguousarray(T_w_c_ref, np.float64).reshape(4, 4)
    n = len(p)
    keep, ang, cnt = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1)), C.c_int()
    r = load_library().mvo_retain_good_triangulation(_p(p), n, _p(Tc), _p(Tr), C.c_double(min_triang_angle),
                                                     C.c_double(max_ratio_to_median), _p(keep), C.byref(cnt), _p(ang))
    if r != MVO_OK:
        raise MvoError(r, "mvo_retain_good_triangulation")
    return keep[:cnt.value].copy(), ang[:n].copy()


def remove_duplicated_matches(matches):
    """geometry::removeDuplicatedMatches (feature_match.cpp:241-260) -- host-side, needs no GPU."""
    m = np.ascontiguousarray(matches, DMATCH_DTYPE).copy()
    n = C.c_int(len(m))
    r = load_library().mvo_remove_duplicated_matches(_p(m), C.byref(n))
    if r != MVO_OK:
        raise MvoError(r, "mvo_remove_duplicated_matches")
    return m[:n.value].copy()


# ------------------------------------------------------------------------------------------------------
# Mirror of the reference's free-function interface (my_slam::geometry / my_slam::optimization) with the
# same latching behaviour as its function-local statics.  `Config` plays basics::Config (config.h:16-50).
class Config:
    """basics::Config stand-in holding config/config.yaml:63-123 values; get_int reproduces the
    cv::FileNode -> int rounding that turns lowe_method_dist_ratio 0.8 into 1 (feature_match.cpp:137-139)."""
    values = {
        "number_of_keypoints_to_extract": 8000, "max_number_of_keypoints": 1500, "scale_factor": 1.2,
        "level_pyramid": 4, "score_threshold": 20,
        "xiang_gao_method_match_ratio": 2, "lowe_method_dist_ratio": 0.8, "method_3_feature_dist_threshold": 50.0,
        "kpts_uniform_selection_grid_size": 16, "kpts_uniform_selection_max_pts_per_grid": 8,
        "num_prev_frames_to_opti_by_ba": 5, "information_matrix": "1.0 0.0 0.0 1.0", "is_ba_fix_map_points": "true",
    }

    @classmethod
    def get(cls, key):
        if key not in cls.values:
            raise RuntimeError("Key " + key + " does not exist")  # config.cpp:34-35
        return cls.values[key]

    @classmethod
    def get_int(cls, key):
        return int(np.rint(float(cls.get(key))))  # cv::FileNode real -> int rounds


_default_ctx = None


def default_context():
    """The process-wide ctx behind the free functions (the reference's function-local statics)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(nfeatures=Config.get_int("number_of_keypoints_to_extract"),
                               scale_factor=float(Config.get("scale_factor")),
                               nlevels=Config.get_int("level_pyramid"),
                               fast_threshold=Config.get_int("score_threshold"),
                               max_keypoints=Config.get_int("max_number_of_keypoints"),
                               grid_size=Config.get_int("kpts_uniform_selection_grid_size"),
                               grid_max_per_cell=Config.get_int("kpts_uniform_selection_max_pts_per_grid"))
    return _default_ctx


def reset_default_context():
    global _default_ctx
    if _default_ctx is not None:
        _default_ctx.close()
    _default_ctx = None


def calcKeyPoints(image):
    return default_context().calc_keypoints(image)


def calcDescriptors(image, keypoints):
    return default_context().calc_descriptors(image, keypoints)


def selectUniformKptsByGrid(keypoints, image_rows, image_cols):
    return default_context().select_uniform_kpts_by_grid(keypoints, image_rows, image_cols)


def matchFeatures(descriptors_1, descriptors_2, method_index=1, is_print_res=False, keypoints_1=None,
                  keypoints_2=None, max_matching_pixel_dist=0.0):
    xy1 = xy2 = None
    if keypoints_1 is not None and len(keypoints_1):
        xy1 = np.stack([keypoints_1["x"], keypoints_1["y"]], 1)
        xy2 = np.stack([keypoints_2["x"], keypoints_2["y"]], 1)
    return default_context().match_features(descriptors_1, descriptors_2, method_index,
                                            float(Config.get_int("xiang_gao_method_match_ratio")),
                                            float(Config.get_int("lowe_method_dist_ratio")), xy1, xy2,
                                            max_matching_pixel_dist)


removeDuplicatedMatches = remove_duplicated_matches


def bundleAdjustment(v_pts_2d, v_pts_2d_to_3d_idx, K, pts_3d, v_camera_g2o_poses, information_matrix,
                     is_fix_map_pts=False, is_update_map_pts=True):
    """optimization::bundleAdjustment (g2o_ba.h:23-30) with Python containers standing in for the pointer
    lists: v_pts_2d[i] = float32 [n_i, 2] pixels of frame i, v_pts_2d_to_3d_idx[i] = map-point ids,
    pts_3d = dict id -> float32[3] (mutated in place when is_update_map_pts), v_camera_g2o_poses = list of
    4x4 float64 cam->world matrices (mutated in place)."""
    ids = list(pts_3d.keys())
    slot = {pid: i for i, pid in enumerate(ids)}
    pts = np.array([pts_3d[i] for i in ids], np.float64).reshape(-1, 3)
    ep, el, uv = [], [], []
    for i, (p2, idx) in enumerate(zip(v_pts_2d, v_pts_2d_to_3d_idx)):
        for j in range(len(p2)):
            ep.append(i)
            el.append(slot[idx[j]])
            uv.append(p2[j])
    poses = np.array(v_camera_g2o_poses, np.float64).reshape(-1, 16)
    K = np.asarray(K, np.float64)
    P, X, st = default_context().bundle_adjustment(
        poses, pts, np.array(ep, np.int32), np.array(el, np.int32), np.array(uv, np.float64).reshape(-1, 2),
        K[0, 0], K[0, 2], K[1, 2], np.asarray(information_matrix, np.float64).ravel(), 1.0, is_fix_map_pts)
    for i in range(len(v_camera_g2o_poses)):
        v_camera_g2o_poses[i][...] = P[i]
    if is_update_map_pts:
        for pid, x in zip(ids, X):
            pts_3d[pid][...] = x.astype(np.float32)   # double -> cv::Point3f (g2o_ba.cpp:313-315)
    return st
