"""The C++ mirror of the reference interface (host/include/my_slam: Frame, calcKeyPoints, matchFeatures,
bundleAdjustment, g2o-shaped facade) driven like the reference's call sites, results checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_struct_equal

BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_dropin")


def test_adapter_binary_is_built_and_links_only_the_hip_library():
    assert os.path.exists(BIN), "run __graft_entry__.build()"
    ldd = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libmvo_hip.so" in ldd and "liboracle" not in ldd and "opencv" not in ldd.lower() and "g2o" not in ldd


def _read(f, dtype):
    n = int(np.frombuffer(f.read(8), "<u8")[0])
    return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype).copy()


@pytest.mark.gpu
def test_cpp_dropin_matches_oracle(mvo, O, tmp_path):
    img0 = mvo.synth.small_test_image(31, 480, 360)
    img1 = np.roll(img0, (2, 5), axis=(0, 1))
    p0, p1, out = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "out.bin"
    img0.tofile(p0)
    img1.tofile(p1)
    r = subprocess.run([BIN, str(p0), str(p1), "480", "360", "3", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    p = O.default_params(max_keypoints=1000)
    with open(out, "rb") as f:
        descs, kps = [], []
        for img in (img0, img1):
            k = _read(f, O.KEYPOINT_DTYPE)
            d = _read(f, np.uint8).reshape(-1, 32)
            rgb = _read(f, np.uint8).reshape(-1, 3)
            ko = O.calc_keypoints(img, p)
            ko, do, rgbo = O.calc_descriptors(img, ko, p, want_rgb=True)
            assert_struct_equal(k, ko, "Frame::calcKeyPoints/calcDescriptors")
            assert np.array_equal(d, do) and np.array_equal(rgb, rgbo)
            descs.append(d)
            kps.append(k)
        for method in (1, 2, 3):
            m = _read(f, O.DMATCH_DTYPE)
            xy = [np.stack([k["x"], k["y"]], 1) for k in kps]
            mo = O.match_features(descs[0], descs[1], method, 2.0, 1.0, xy[0], xy[1], 50.0)   # ratios latched as int
            assert_struct_equal(m, mo, "matchFeatures method %d" % method)
        P1 = np.stack([_read(f, "<f8").reshape(4, 4) for _ in range(3)])
        P2 = np.stack([_read(f, "<f8").reshape(4, 4) for _ in range(3)])
        X2 = _read(f, np.dtype([("xyz", "<f4", 3)]))["xyz"]        # vector<cv::Point3f>
    # pose-only BA pulled the perturbed poses back to the truth (x = 0.05 f, y = z = 0) within the pixel noise
    assert np.abs(P1[:, :3, 3] - np.array([[0, 0, 0], [0.05, 0, 0], [0.10, 0, 0]])).max() < 2e-3
    assert np.isfinite(P2).all() and np.isfinite(X2).all() and len(X2) == 150


TRACK_BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_tracking")


def test_tracking_binary_is_built_and_links_only_the_hip_library():
    assert os.path.exists(TRACK_BIN), "run __graft_entry__.build()"
    ldd = subprocess.run(["ldd", TRACK_BIN], capture_output=True, text=True).stdout
    assert "libmvo_hip.so" in ldd and "liboracle" not in ldd and "opencv" not in ldd.lower()


@pytest.mark.gpu
def test_cpp_tracking_mirror_matches_oracle(mvo, O, tmp_path):
    """my_slam/vo/pnp_tracking.h run as vo.cpp:270-383 runs: same candidates, matches, inliers, pose, bookkeeping."""
    pr = mvo.synth.tracking_problem(n_map=2500, seed=41, outlier_frac=0.0)
    K, M = pr["K"], len(pr["map_pos"])
    rng = np.random.RandomState(9)
    vis, px_true = O.map_in_view(pr["map_pos"], pr["T_w_c"], K, pr["cols"], pr["rows"])
    seen = rng.permutation(len(vis))[: int(0.7 * len(vis))]
    bits = np.unpackbits(pr["map_desc"][vis[seen]], axis=1)
    bits ^= (rng.uniform(size=bits.shape) < 0.03).astype(np.uint8)
    desc = np.concatenate([np.packbits(bits, axis=1), rng.randint(0, 256, (300, 32)).astype(np.uint8)])
    xy = np.concatenate([px_true[seen] + rng.normal(0, 0.3, (len(seen), 2)).astype(np.float32),
                         rng.uniform(0, 480, (300, 2)).astype(np.float32)]).astype(np.float32)
    # the frame's pose guess: a few millimetres / milliradians off, like the previous frame's pose
    T_guess = pr["T_w_c"].copy()
    T_guess[:3, 3] += [0.004, -0.003, 0.005]
    T_prev = pr["T_w_c"].copy()
    T_prev[:3, 3] += [0.02, 0.0, -0.01]
    scene, out = tmp_path / "scene.bin", tmp_path / "out.bin"
    with open(scene, "wb") as f:
        f.write(np.array([M, len(xy), pr["cols"], pr["rows"]], "<i4").tobytes())
        f.write(np.array([K["fx"], K["fy"], K["cx"], K["cy"]], "<f8").tobytes())
        f.write(np.ascontiguousarray(T_guess, "<f8").tobytes())
        f.write(np.ascontiguousarray(T_prev, "<f8").tobytes())
        f.write(pr["map_pos"].astype("<f4").tobytes())
        f.write(pr["map_desc"].tobytes())
        f.write(xy.tobytes())
        f.write(desc.tobytes())
    r = subprocess.run([TRACK_BIN, str(scene), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out, "rb") as f:
        order = _read(f, "<i4")                      # iteration order of the unordered_map (ids)
        ids = _read(f, "<i4")
        px = _read(f, "<f4").reshape(-1, 2)
        cdesc = _read(f, np.uint8).reshape(-1, 32)
        good = int(_read(f, "<i4")[0])
        matches = _read(f, O.DMATCH_DTYPE)
        T_out = _read(f, "<f8").reshape(4, 4)
        conn = _read(f, "<i4").reshape(-1, 2)
        times = _read(f, "<i4").reshape(-1, 2)
        R5 = _read(f, "<f8").reshape(3, 3)
        t5 = _read(f, "<f8")
    assert sorted(order.tolist()) == list(range(M))
    # -- getMappointsInCurrentView_: the oracle on the map in the adapter's order
    idx_o, px_o = O.map_in_view(pr["map_pos"][order], T_guess, K, pr["cols"], pr["rows"])
    assert np.array_equal(ids, order[idx_o]) and np.array_equal(px, px_o)
    assert np.array_equal(cdesc, pr["map_desc"][ids])
    # -- matchFeatures (method 1) + the 3D-2D pairs + solvePnPRansac
    m_o = O.match_features(pr["map_desc"][ids], desc, 1, 2.0, 1.0)
    p3 = pr["map_pos"][ids[m_o["queryIdx"]]]
    p2 = xy[m_o["trainIdx"]]
    ref = O.solve_pnp_ransac(p3, p2, K)
    assert ref["ok"] and good == 1
    assert_struct_equal(matches, m_o[ref["inliers"]], "curr_->matches_with_map_ after the inlier swap")
    T_c_w = np.eye(4)
    T_c_w[:3, :3] = O.rodrigues(ref["rvec"])
    T_c_w[:3, 3] = ref["tvec"]
    assert np.abs(T_out - np.linalg.inv(T_c_w)).max() < 1e-8
    assert np.abs(T_out - pr["T_w_c"]).max() < 2e-3
    want_conn = {int(m["trainIdx"]): int(ids[m["queryIdx"]]) for m in m_o[ref["inliers"]]}
    assert {int(a): int(b) for a, b in conn} == want_conn
    vis_times = np.ones(M, int)
    vis_times[ids] += 2                            # the program looks at the view twice (alone, then inside PnP)
    mt = np.ones(M, int)
    np.add.at(mt, ids[m_o["queryIdx"]][ref["inliers"]], 1)
    assert np.array_equal(times[:, 0], vis_times) and np.array_equal(times[:, 1], mt)
    # -- the literal cv::solvePnPRansac / cv::Rodrigues call on 5 pairs
    five = O.solve_pnp_ransac(pr["map_pos"][ids[:5]], px[:5], K)
    assert np.abs(R5 - O.rodrigues(five["rvec"])).max() < 1e-12 and np.abs(t5 - five["tvec"]).max() < 1e-12


KEY_BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_keyframe")


def test_keyframe_binary_is_built_and_links_only_the_hip_library():
    assert os.path.exists(KEY_BIN), "run __graft_entry__.build()"
    ldd = subprocess.run(["ldd", KEY_BIN], capture_output=True, text=True).stdout
    assert "libmvo_hip.so" in ldd and "liboracle" not in ldd and "opencv" not in ldd.lower()


@pytest.mark.gpu
def test_cpp_keyframe_mirror_matches_oracle(mvo, O, tmp_path):
    """my_slam/vo/keyframe.h run as vo_addFrame.cpp:96-118 runs: matchFeatures -> helperFindInlierMatchesByEpipolarCons
    -> helperTriangulatePoints -> retainGoodTriangulationResult_."""
    kf = mvo.synth.keyframe_problem(n=700, seed=51, outlier_frac=0.0)
    rng = np.random.RandomState(4)
    n = len(kf["kp_ref"])
    d_ref = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    # the current keyframe sees 80 % of the points (descriptors with a few flipped bits, shuffled) plus clutter
    seen = rng.permutation(n)[: int(0.8 * n)]
    bits = np.unpackbits(d_ref[seen], axis=1)
    bits ^= (rng.uniform(size=bits.shape) < 0.03).astype(np.uint8)
    d_cur = np.concatenate([np.packbits(bits, axis=1), rng.randint(0, 256, (150, 32)).astype(np.uint8)])
    kp_cur = np.concatenate([kf["kp_cur"][seen], rng.uniform(0, 480, (150, 2)).astype(np.float32)]).astype(np.float32)
    scene, out = tmp_path / "kscene.bin", tmp_path / "kout.bin"
    K = kf["K"]
    with open(scene, "wb") as f:
        f.write(np.array([n, len(kp_cur)], "<i4").tobytes())
        f.write(np.array([K["fx"], K["fy"], K["cx"], K["cy"]], "<f8").tobytes())
        f.write(np.ascontiguousarray(kf["T_w_ref"], "<f8").tobytes())
        f.write(np.ascontiguousarray(kf["T_w_cur"], "<f8").tobytes())
        f.write(kf["kp_ref"].tobytes())
        f.write(d_ref.tobytes())
        f.write(kp_cur.tobytes())
        f.write(d_cur.tobytes())
    r = subprocess.run([KEY_BIN, str(scene), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out, "rb") as f:
        matches = _read(f, O.DMATCH_DTYPE)
        inl_matches = _read(f, O.DMATCH_DTYPE)
        kept_matches = _read(f, O.DMATCH_DTYPE)
        pts3d = _read(f, "<f4").reshape(-1, 3)
        angles = _read(f, "<f8")
        T = _read(f, "<f8").reshape(4, 4)
    m_o = O.match_features(d_ref, d_cur, 1, 2.0, 1.0)
    assert_struct_equal(matches, m_o, "matches_with_ref_")
    a, b = kf["kp_ref"][m_o["queryIdx"]], kp_cur[m_o["trainIdx"]]
    inl = O.find_essential_inliers(a, b, K)["inliers"]
    assert np.array_equal(inl_matches["queryIdx"], m_o["queryIdx"][inl]) and np.array_equal(inl_matches["trainIdx"], m_o["trainIdx"][inl])
    assert np.array_equal(inl_matches["distance"], m_o["distance"][inl])
    T_o = np.linalg.inv(kf["T_w_cur"]) @ kf["T_w_ref"]
    assert np.abs(T - T_o).max() < 1e-12
    _, p_cur = O.triangulate_points(a[inl], b[inl], K, T[:3, :3], T[:3, 3])      # same T as the adapter used
    keep, ang = O.retain_good_triangulation(p_cur, kf["T_w_cur"], kf["T_w_ref"])
    assert np.array_equal(pts3d, p_cur[keep]) and np.array_equal(angles, ang[keep])
    assert np.array_equal(kept_matches["queryIdx"], m_o["queryIdx"][inl][keep])
    assert len(keep) > 0.5 * len(seen)


LOOP_BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_vo_loop")


@pytest.mark.gpu
def test_cpp_tracking_loop_follows_the_ground_truth(mvo, tmp_path):
    """All rows chained the way the reference chains them (vo_addFrame.cpp:70-124) over 30 frames of a 3-D scene:
    map points in view -> matchFeatures -> solvePnPRansac -> sliding-window BA -> (on keyframes) epipolar filter,
    triangulation, culling, map growth / pruning.  The estimated trajectory must follow the ground truth."""
    seq = mvo.synth.feature_sequence(n_frames=30, seed=61)
    F, K = len(seq["frames"]), seq["K"]
    f0, f1 = seq["frames"][0], seq["frames"][1]
    # what an initialisation would have left behind: points seen in both keyframes, slightly noisy positions
    rng = np.random.RandomState(7)
    ids1 = {int(p): k for k, p in enumerate(f1["point_id"]) if p >= 0}
    ids0 = {int(p): k for k, p in enumerate(f0["point_id"]) if p >= 0}
    both = sorted(set(ids1) & set(ids0))
    scene, out = tmp_path / "loop.bin", tmp_path / "loop_out.bin"
    with open(scene, "wb") as f:
        f.write(np.array([F, len(both), seq["cols"], seq["rows"]], "<i4").tobytes())
        f.write(np.array([K["fx"], K["fy"], K["cx"], K["cy"]], "<f8").tobytes())
        for fr in seq["frames"]:
            f.write(np.ascontiguousarray(fr["T_w_c"], "<f8").tobytes())
            f.write(np.array([len(fr["xy"])], "<i4").tobytes())
            f.write(fr["xy"].tobytes())
            f.write(fr["desc"].tobytes())
        for p in both:
            f.write((seq["points"][p] + rng.normal(0, 0.005, 3)).astype("<f4").tobytes())
            f.write(np.array([ids1[p], ids0[p]], "<i4").tobytes())
    r = subprocess.run([LOOP_BIN, str(scene), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    rec = np.frombuffer(raw, np.dtype([("flags", "<i4", 4), ("T", "<f8", (4, 4))]))
    assert len(rec) == F - 2
    good, kf, map_size, conns = rec["flags"].T
    assert good.all(), "PnP failed on frames %s" % np.nonzero(good == 0)[0]
    assert kf.sum() >= 5 and (conns > 100).all()
    gt = np.stack([fr["T_w_c"] for fr in seq["frames"][2:]])
    t_err = np.linalg.norm(rec["T"][:, :3, 3] - gt[:, :3, 3], axis=1)
    cos = (np.einsum("nij,nij->n", rec["T"][:, :3, :3], gt[:, :3, :3]) - 1) / 2
    r_err = np.degrees(np.arccos(np.clip(cos, -1, 1)))
    assert t_err.max() < 0.03 and r_err.max() < 0.5, (t_err.max(), r_err.max())
    # the map is alive: new points were triangulated and pushed while old ones left the view and were pruned
    assert map_size.min() > 200 and map_size[-1] != map_size[0]
    travelled = np.linalg.norm(gt[-1, :3, 3] - gt[0, :3, 3])
    assert travelled > 0.7 and t_err[-1] < 0.04 * travelled


CALLSITES_BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_callsites")


@pytest.mark.gpu
def test_every_function_of_the_replaced_headers_runs_like_the_reference(mvo, O, tmp_path):
    """feature_match.h:12-54 and g2o_ba.h:16-30 complete: matchByRadiusAndBruteForce, computeMeanDistBetweenKeypoints,
    inliers2DMatches, pts2Keypts (vo.cpp:139,277) and optimizeSingleFrame (g2o_ba.cpp:34-145: identity information,
    no robust kernel, 50 iterations) next to the functions test_dropin already covers."""
    out = tmp_path / "cs.bin"
    r = subprocess.run([CALLSITES_BIN, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out, "rb") as f:
        k1, k2 = _read(f, O.KEYPOINT_DTYPE), _read(f, O.KEYPOINT_DTYPE)
        d1, d2 = _read(f, np.uint8).reshape(-1, 32), _read(f, np.uint8).reshape(-1, 32)
        m = [_read(f, O.DMATCH_DTYPE) for _ in range(3)]
        mr, mdup = _read(f, O.DMATCH_DTYPE), _read(f, O.DMATCH_DTYPE)
        mean_dist = _read(f, "<f8")[0]
        p2 = _read(f, "<f4").reshape(-1, 2)
        p3 = _read(f, "<f4").reshape(-1, 3)
        T0, T_pose_only, T_full = (_read(f, "<f8").reshape(4, 4) for _ in range(3))
        p3_after = _read(f, "<f4").reshape(-1, 3)
        T_ba = _read(f, "<f8").reshape(4, 4)
    xy1, xy2 = np.stack([k1["x"], k1["y"]], 1), np.stack([k2["x"], k2["y"]], 1)
    for method in (1, 2, 3):
        mo = O.match_features(d1, d2, method, 2.0, 1.0, xy1, xy2, 10.0)
        assert_struct_equal(m[method - 1], mo, "matchFeatures method %d" % method)
    # matchByRadiusAndBruteForce (feature_match.cpp:86-124): first minimum of the mean |a - b| inside the radius
    idx, s = O.match_radius_l1(d1, xy1, d2, xy2, 10.0)
    keep = idx >= 0
    assert np.array_equal(mr["queryIdx"], np.nonzero(keep)[0]) and np.array_equal(mr["trainIdx"], idx[keep])
    assert np.array_equal(mr["distance"], (s[keep].astype(np.float64) / 32).astype(np.float32))
    assert len(mr) > 100 and len(np.unique(mdup["trainIdx"])) == len(mdup)
    # computeMeanDistBetweenKeypoints (feature_match.cpp:263-278)
    dx = xy1[m[0]["queryIdx"]].astype(np.float64) - xy2[m[0]["trainIdx"]].astype(np.float64)
    assert abs(mean_dist - np.sqrt((dx * dx).sum(1)).mean()) < 1e-9
    # optimizeSingleFrame == the oracle's LM without robust kernel on the one-pose window
    n = len(p3)
    args = (T0[None], p3.astype(np.float64), np.zeros(n, np.int32), np.arange(n, dtype=np.int32), p2.astype(np.float64),
            517.3, 325.1, 249.7)
    Po, _, so = O.bundle_adjustment(*args, fix_points=True, huber_delta=1e100)
    assert np.abs(T_pose_only - Po[0]).max() < 1e-6 and so["iterations"] >= 1
    Pf, Xf, _ = O.bundle_adjustment(*args, fix_points=False, huber_delta=1e100)
    # (pose + points free with one camera: the gauge is held by the damping only -> compare the gauge-invariant part)
    def reproj(P, X):
        Tcw = np.linalg.inv(P)
        pc = X @ Tcw[:3, :3].T + Tcw[:3, 3]
        return 517.3 * pc[:, :2] / pc[:, 2:] + [325.1, 249.7]
    assert np.abs(reproj(T_full, p3_after.astype(np.float64)) - reproj(Pf[0], Xf)).max() < 5e-2
    assert np.abs(reproj(T_full, p3_after.astype(np.float64)) - p2).max() < 0.5
    # bundleAdjustment with fixed points = the same pose-only problem but WITH the Huber kernel (delta 1)
    Ph, _, _ = O.bundle_adjustment(*args, fix_points=True)
    assert np.abs(T_ba - Ph[0]).max() < 1e-6
    # the pose came back to the truth (identity) within the pixel noise
    assert np.abs(T_pose_only - np.eye(4)).max() < 2e-3


MULTI = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "driver", "multi_gpu_shards")


def test_multi_gpu_example_is_built_against_rccl_and_the_hip_library_only():
    assert os.path.exists(MULTI), "run __graft_entry__.build()"
    ldd = subprocess.run(["ldd", MULTI], capture_output=True, text=True).stdout
    assert "libmvo_hip.so" in ldd and "librccl" in ldd and "liboracle" not in ldd and "torch" not in ldd


@pytest.mark.gpu
def test_cpp_host_shards_sequences_and_gathers_the_trajectories(tmp_path):
    """SURVEY 8(e) from the C++ host (host/driver/multi_gpu_shards.cpp): one mvo_ctx per GPU rank, no collective while the
    sequences run, one RCCL all-gather of the frames x 12 trajectory blocks (vo_io.cpp:58-75 row format).  A 1-GPU box runs it with
    one rank; the gathered rows must be what the rank computed and the file the reference's trajectory format."""
    prefix = str(tmp_path / "traj")
    r = subprocess.run([MULTI, "8", "6", prefix], capture_output=True, text=True, timeout=300)     # ranks are clamped to the GPUs present
    assert r.returncode == 0 and any(line.startswith("ok:") for line in r.stdout.splitlines()), r.stdout + r.stderr    # (RCCL prints a banner first)
    rows = np.loadtxt(prefix + "_shard0.txt")
    assert rows.shape == (6, 12) and np.allclose(rows[:, 3:], np.eye(3).T.reshape(-1))
    # the scene moves 3 px per frame to the left: the example's pose (median image shift) follows it
    assert np.allclose(np.diff(rows[:, 0]), 3.0, atol=0.5) and np.allclose(rows[:, 1], 0.0, atol=0.5)
