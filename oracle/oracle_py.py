"""ctypes wrapper around the CPU ORACLE (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h): imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("fast_threshold", C.c_int32), ("max_keypoints", C.c_int32), ("grid_size", C.c_int32),
                ("grid_max_per_cell", C.c_int32), ("pyramid_interpolation", C.c_int32)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CANDIDATE_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("level", "<i4"), ("fast_score", "<i4"),
                            ("harris", "<f4"), ("angle", "<f4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_T_w_c", C.c_void_p), ("points", C.c_void_p), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("edge_uv", C.c_void_p), ("focal", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("info", C.c_double * 4),
                ("huber_delta", C.c_double), ("fix_points", C.c_int32), ("pose_fixed", C.c_void_p),
                ("max_iterations", C.c_int32)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("terminated", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def default_params(**kw):
    """config/config.yaml:65-69, 94-95 defaults."""
    p = dict(nfeatures=8000, scale_factor=1.2, nlevels=4, fast_threshold=20, max_keypoints=1500,
             grid_size=16, grid_max_per_cell=8, pyramid_interpolation=1)
    p.update(kw)
    return OrbParams(**p)


def _img_args(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        h, w = img.shape
        ch = 1
    else:
        h, w, ch = img.shape
    return img, img.ctypes.data_as(C.c_void_p), w, h, w * ch, ch


def level_size(w, h, p, level):
    lw, lh, sc = C.c_int(), C.c_int(), C.c_float()
    lib().orc_orb_level_size(w, h, C.byref(p), level, C.byref(lw), C.byref(lh), C.byref(sc))
    return lw.value, lh.value, sc.value


def feature_quota(p):
    q = (C.c_int32 * p.nlevels)()
    lib().orc_orb_feature_quota(C.byref(p), q)
    return list(q)


def pyramid_level(img, p, level, blurred=False):
    img, ptr, w, h, stride, ch = _img_args(img)
    lw, lh, _ = level_size(w, h, p, level)
    out = np.zeros((lh + 64, lw + 64), np.uint8)
    r = lib().orc_orb_pyramid_level(ptr, w, h, stride, ch, C.byref(p), level, int(blurred),
                                    out.ctypes.data_as(C.c_void_p))
    assert r == out.size, r
    return out


def candidates(img, p, cap=1 << 18):
    img, ptr, w, h, stride, ch = _img_args(img)
    out = np.zeros(cap, CANDIDATE_DTYPE)
    n = lib().orc_orb_candidates(ptr, w, h, stride, ch, C.byref(p), out.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0, n
    return out[:n].copy()


def orb_detect(img, p, cap=1 << 18):
    img, ptr, w, h, stride, ch = _img_args(img)
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().orc_orb_detect(ptr, w, h, stride, ch, C.byref(p), out.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0, n
    return out[:n].copy()


def calc_keypoints(img, p, grid_rows=0, grid_cols=0, cap=1 << 18):
    img, ptr, w, h, stride, ch = _img_args(img)
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().orc_calc_keypoints(ptr, w, h, stride, ch, C.byref(p), grid_rows, grid_cols,
                                 out.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0, n
    return out[:n].copy()


def select_uniform_kpts_by_grid(kps, rows, cols, p):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
    n = lib().orc_select_uniform_kpts_by_grid(kps.ctypes.data_as(C.c_void_p), len(kps), rows, cols, C.byref(p))
    assert n >= 0, n
    return kps[:n].copy()


def calc_descriptors(img, kps, p, want_rgb=False):
    img, ptr, w, h, stride, ch = _img_args(img)
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
    desc = np.zeros((max(len(kps), 1), 32), np.uint8)
    rgb = np.zeros((max(len(kps), 1), 3), np.uint8)
    n = lib().orc_calc_descriptors(ptr, w, h, stride, ch, C.byref(p), kps.ctypes.data_as(C.c_void_p),
                                   len(kps), desc.ctypes.data_as(C.c_void_p),
                                   rgb.ctypes.data_as(C.c_void_p) if want_rgb else None)
    assert n >= 0, n
    if want_rgb:
        return kps[:n].copy(), desc[:n].copy(), rgb[:n].copy()
    return kps[:n].copy(), desc[:n].copy()


def match_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().orc_match_knn2(q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                         idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
    return idx, dist


def match_radius_l1(q, qxy, t, txy, max_px):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
    txy = np.ascontiguousarray(txy, np.float32).reshape(-1, 2)
    idx = np.zeros(len(q), np.int32)
    s = np.zeros(len(q), np.int32)
    lib().orc_match_radius_l1(q.ctypes.data_as(C.c_void_p), qxy.ctypes.data_as(C.c_void_p), len(q),
                              t.ctypes.data_as(C.c_void_p), txy.ctypes.data_as(C.c_void_p), len(t),
                              C.c_float(max_px), idx.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
    return idx, s


def match_knn2_lsh(q, t, tables=5, key_size=10, probe_level=2, seed=1):
    """2-NN through a restated FLANN LSH index (the reference's LshIndexParams(5, 10, 2)); -1 = no candidate."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    n = lib().orc_match_knn2_lsh(q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t), tables,
                                 key_size, probe_level, C.c_uint32(seed), idx.ctypes.data_as(C.c_void_p),
                                 dist.ctypes.data_as(C.c_void_p))
    assert n >= 0, n
    return idx, dist


def match_features_from_knn(idx, dist, method=1, xiang_gao_ratio=2.0, lowe_ratio=1.0):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    out = np.zeros(max(len(idx), 1), DMATCH_DTYPE)
    n = lib().orc_match_features_from_knn(idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p), len(idx),
                                          method, C.c_double(xiang_gao_ratio), C.c_double(lowe_ratio),
                                          out.ctypes.data_as(C.c_void_p), len(out))
    assert n >= 0, n
    return out[:n].copy()


def match_features(d1, d2, method=1, xiang_gao_ratio=2.0, lowe_ratio=1.0, xy1=None, xy2=None, max_px=0.0):
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
    out = np.zeros(max(len(d1), 1), DMATCH_DTYPE)
    p1 = p2 = None
    if xy1 is not None:
        xy1 = np.ascontiguousarray(xy1, np.float32)
        xy2 = np.ascontiguousarray(xy2, np.float32)
        p1, p2 = xy1.ctypes.data_as(C.c_void_p), xy2.ctypes.data_as(C.c_void_p)
    n = lib().orc_match_features(d1.ctypes.data_as(C.c_void_p), len(d1), d2.ctypes.data_as(C.c_void_p), len(d2),
                                 method, C.c_double(xiang_gao_ratio), C.c_double(lowe_ratio), p1, p2,
                                 C.c_float(max_px), out.ctypes.data_as(C.c_void_p), len(out))
    if n < 0:
        raise RuntimeError("feature_match.cpp::matchFeatures: wrong method index.")
    return out[:n].copy()


def remove_duplicated_matches(m):
    m = np.ascontiguousarray(m, DMATCH_DTYPE).copy()
    n = lib().orc_remove_duplicated_matches(m.ctypes.data_as(C.c_void_p), len(m))
    return m[:n].copy()


def _ba_problem(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                fix_points, pose_fixed, max_iterations):
    keep = dict(
        poses=np.ascontiguousarray(poses, np.float64).reshape(-1, 16).copy(),
        points=np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy(),
        ep=np.ascontiguousarray(edge_pose, np.int32), el=np.ascontiguousarray(edge_point, np.int32),
        uv=np.ascontiguousarray(edge_uv, np.float64).reshape(-1, 2),
        pf=None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8))
    pr = BaProblem()
    pr.n_poses, pr.n_points, pr.n_edges = len(keep["poses"]), len(keep["points"]), len(keep["ep"])
    pr.pose_T_w_c = keep["poses"].ctypes.data
    pr.points = keep["points"].ctypes.data
    pr.edge_pose = keep["ep"].ctypes.data
    pr.edge_point = keep["el"].ctypes.data
    pr.edge_uv = keep["uv"].ctypes.data
    pr.focal, pr.cx, pr.cy = focal, cx, cy
    pr.info = (C.c_double * 4)(*np.asarray(info, np.float64).ravel())
    pr.huber_delta = huber_delta
    pr.fix_points = int(fix_points)
    pr.pose_fixed = None if keep["pf"] is None else keep["pf"].ctypes.data
    pr.max_iterations = max_iterations
    return pr, keep


def bundle_adjustment(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info=(1, 0, 0, 1),
                      huber_delta=1.0, fix_points=False, pose_fixed=None, max_iterations=50):
    """Returns (poses_T_w_c [F,4,4] f64, points [L,3] f64, stats dict)."""
    pr, keep = _ba_problem(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                           fix_points, pose_fixed, max_iterations)
    st = BaStats()
    r = lib().orc_bundle_adjustment(C.byref(pr), C.byref(st))
    if r != 0:
        raise RuntimeError("oracle BA failed: %d" % r)
    stats = {k: getattr(st, k) for k, _ in BaStats._fields_}
    return keep["poses"].reshape(-1, 4, 4), keep["points"], stats


def bundle_adjustment_blocked(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, plan, info=(1, 0, 0, 1),
                              huber_delta=1.0, fix_points=False, pose_fixed=None, max_iterations=50):
    """The same LM solve with the blocked summation order of `plan` = dict(wgs, nsplit, wg_pt_start[, groups]) (what
    mvo_debug_get_ba_plan reports).  Returns (poses, points, stats, trace [trials, 4] = lambda, chi2, rho, accepted)."""
    pr, keep = _ba_problem(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                           fix_points, pose_fixed, max_iterations)
    st = BaStats()
    wg = np.ascontiguousarray(plan["wg_pt_start"], np.int32)
    trace = np.zeros((512, 4))
    nt = C.c_int()
    r = lib().orc_bundle_adjustment_blocked(C.byref(pr), int(plan["wgs"]), wg.ctypes.data_as(C.c_void_p),
                                            int(plan["nsplit"]) | ((int(plan.get("groups", 1)) << 16) if int(plan.get("groups", 1)) > 1 else 0),
                                            C.byref(st), trace.ctypes.data_as(C.c_void_p), 512, C.byref(nt))
    if r != 0:
        raise RuntimeError("oracle BA (blocked order) failed: %d" % r)
    stats = {k: getattr(st, k) for k, _ in BaStats._fields_}
    return keep["poses"].reshape(-1, 4, 4), keep["points"], stats, trace[:min(nt.value, 512)].copy()


def ba_linearize(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info=(1, 0, 0, 1),
                 huber_delta=1.0, fix_points=False, pose_fixed=None):
    pr, keep = _ba_problem(poses, points, edge_pose, edge_point, edge_uv, focal, cx, cy, info, huber_delta,
                           fix_points, pose_fixed, 0)
    nf = pr.n_poses if pose_fixed is None else int(pr.n_poses - np.count_nonzero(keep["pf"]))
    n = 6 * nf + (0 if fix_points else 3 * pr.n_points)
    H = np.zeros((n, n))
    b = np.zeros(n)
    chi = C.c_double()
    r = lib().orc_ba_linearize(C.byref(pr), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                               C.byref(chi), n)
    assert r == n, (r, n)
    return H, b, chi.value


# ---------------------------------------------------------------- tracking rows (pnp_oracle.cpp)
def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def _K4(K):
    return np.ascontiguousarray([K["fx"], K["fy"], K["cx"], K["cy"]], np.float64)


def invert4x4(T):
    T = np.ascontiguousarray(T, np.float64)
    out = np.zeros((4, 4))
    ok = lib().orc_invert4x4(_dp(T), _dp(out))
    return out if ok else None


def map_in_view(pos, T_w_c, K, cols, rows):
    pos = np.ascontiguousarray(pos, np.float32)
    T = np.ascontiguousarray(T_w_c, np.float64)
    n = len(pos)
    idx = np.zeros(n, np.int32)
    px = np.zeros((n, 2), np.float32)
    f = lib().orc_map_in_view
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                  C.c_int, C.c_void_p, C.c_void_p]
    cnt = f(_dp(pos), n, _dp(T), K["fx"], K["fy"], K["cx"], K["cy"], cols, rows, _dp(idx), _dp(px))
    assert cnt >= 0
    return idx[:cnt].copy(), px[:cnt].copy()


def pnp_subsets(count, n_iters, model_points=5):
    idx = np.zeros((n_iters, model_points), np.int32)
    r = lib().orc_pnp_subsets(count, model_points, n_iters, _dp(idx))
    assert r == 0
    return idx


def epnp(p3, p2, idx, K, want_ut=False):
    p3 = np.ascontiguousarray(p3, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    R, t, ut = np.zeros((3, 3)), np.zeros(3), np.zeros((12, 12))
    k4 = _K4(K)
    lib().orc_epnp(_dp(p3), _dp(p2), _dp(idx), len(idx), _dp(k4), _dp(R), _dp(t), _dp(ut))
    return (R, t, ut) if want_ut else (R, t)


def pnp_score(p3, p2, K, R, t, reproj=2.0):
    p3 = np.ascontiguousarray(p3, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    R = np.ascontiguousarray(R, np.float64)
    t = np.ascontiguousarray(t, np.float64)
    mask = np.zeros(len(p3), np.uint8)
    k4 = _K4(K)
    f = lib().orc_pnp_score
    f.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_void_p]
    cnt = f(_dp(p3), _dp(p2), len(p3), _dp(k4), _dp(R), _dp(t), reproj, _dp(mask))
    return cnt, mask


def rodrigues(r, want_jac=False):
    r = np.ascontiguousarray(r, np.float64)
    R, J = np.zeros((3, 3)), np.zeros((3, 9))
    lib().orc_rodrigues(_dp(r), _dp(R), _dp(J) if want_jac else None)
    return (R, J) if want_jac else R


def rodrigues_inv(R):
    R = np.ascontiguousarray(R, np.float64)
    r = np.zeros(3)
    lib().orc_rodrigues_inv(_dp(R), _dp(r))
    return r


def solve_pnp_iterative(M, m, K, init_R=None, init_t=None):
    M = np.ascontiguousarray(M, np.float64)
    m = np.ascontiguousarray(m, np.float64)
    R0 = np.ascontiguousarray(np.eye(3) if init_R is None else init_R, np.float64)
    t0 = np.ascontiguousarray(np.zeros(3) if init_t is None else init_t, np.float64)
    param = np.zeros(6)
    it, ev = C.c_int(), C.c_int()
    k4 = _K4(K)
    dlt = lib().orc_solve_pnp_iterative(_dp(M), _dp(m), len(M), _dp(k4), _dp(R0), _dp(t0), _dp(param),
                                        C.byref(it), C.byref(ev))
    return dict(rvec=param[:3].copy(), tvec=param[3:].copy(), dlt=dlt, lm_iters=it.value, lm_evals=ev.value)


def solve_pnp_ransac(p3, p2, K, iters=100, reproj=2.0, confidence=0.999):
    """cv::solvePnPRansac as vo.cpp:326-329 calls it.  Returns a dict with ok, rvec, tvec, inliers and the
    debug record (models [iters,12], counts [iters], best_iter, iters_run, dlt, lm_iters)."""
    p3 = np.ascontiguousarray(p3, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    n = len(p3)
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    n_inl = C.c_int()
    models = np.full((iters, 12), np.nan)
    counts = np.full(iters, -1, np.int32)
    info = np.zeros(4, np.int32)
    k4 = _K4(K)
    f = lib().orc_solve_pnp_ransac
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_double] + [C.c_void_p] * 7
    ok = f(_dp(p3), _dp(p2), n, _dp(k4), iters, reproj, confidence, _dp(rvec), _dp(tvec), _dp(inl),
           C.addressof(n_inl), _dp(models), _dp(counts), _dp(info))
    return dict(ok=bool(ok), rvec=rvec, tvec=tvec, inliers=inl[:n_inl.value].copy(), models=models, counts=counts,
                best_iter=int(info[0]), iters_run=int(info[1]), dlt=int(info[2]), lm_iters=int(info[3]))


# ---------------------------------------------------------------- keyframe row (keyframe_oracle.cpp)
def triangulate_points(kp1, kp2, K, R, t):
    """helperTriangulatePoints -> (pts in the previous camera frame, pts after transCoord), n x 3 float32 each."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    R = np.ascontiguousarray(R, np.float64)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    n = len(kp1)
    a, b = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    k4 = _K4(K)
    lib().orc_triangulate_points(_dp(kp1), _dp(kp2), n, _dp(k4), _dp(R), _dp(t), _dp(a), _dp(b))
    return a, b


def retain_good_triangulation(pts_curr, T_w_c_curr, T_w_c_ref, min_angle=1.0, max_ratio=20.0):
    """retainGoodTriangulationResult_ -> (kept indices, angles in degrees of all points)."""
    p = np.ascontiguousarray(pts_curr, np.float32).reshape(-1, 3)
    Tc = np.ascontiguousarray(T_w_c_curr, np.float64)
    Tr = np.ascontiguousarray(T_w_c_ref, np.float64)
    n = len(p)
    keep = np.zeros(max(n, 1), np.int32)
    ang = np.zeros(max(n, 1))
    f = lib().orc_retain_good_triangulation
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    cnt = f(_dp(p), n, _dp(Tc), _dp(Tr), min_angle, max_ratio, _dp(keep), _dp(ang))
    return keep[:cnt].copy(), ang[:n].copy()


def five_point(q1, q2, want_dbg=False):
    """five-point kernel on 5 normalised correspondences -> E candidates [k, 3, 3] (x2^T E x1 = 0)."""
    q1 = np.ascontiguousarray(q1, np.float64).reshape(5, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(5, 2)
    E = np.zeros((10, 3, 3))
    dbg = np.zeros(257)
    k = lib().orc_five_point(_dp(q1), _dp(q2), _dp(E), _dp(dbg))
    if want_dbg:
        return E[:k].copy(), dict(basis=dbg[:36].reshape(4, 9), A=dbg[36:236].reshape(10, 20), poly=dbg[236:247],
                                  roots=dbg[247:257])
    return E[:k].copy()


def real_roots_deg10(c):
    c = np.ascontiguousarray(c, np.float64).reshape(11)
    r = np.zeros(10)
    k = lib().orc_real_roots_deg10(_dp(c), _dp(r))
    return r[:k].copy()


def find_essential_inliers(kp1, kp2, K, prob=0.999, threshold=1.0, max_iters=1000):
    """helperFindInlierMatchesByEpipolarCons -> dict(inliers, counts [max_iters, 10], best_iter, best_model,
    iters_run, n_models, E)."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    n = len(kp1)
    inl = np.zeros(max(n, 1), np.int32)
    counts = np.full((max_iters, 10), -2, np.int32)
    info = np.zeros(4, np.int32)
    E = np.zeros((3, 3))
    k4 = _K4(K)
    f = lib().orc_find_essential_inliers
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 4
    cnt = f(_dp(kp1), _dp(kp2), n, _dp(k4), prob, threshold, max_iters, _dp(inl), _dp(counts), _dp(info), _dp(E))
    return dict(inliers=inl[:cnt].copy(), counts=counts, best_iter=int(info[0]), best_model=int(info[1]),
                iters_run=int(info[2]), n_models=int(info[3]), E=E)
