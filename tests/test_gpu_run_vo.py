"""Headless run_vo (host/driver/run_vo.cpp = reference run_vo.cpp:58-154 without the displays) on the reference's own wire
formats: config.yaml, <dataset_dir>/rgb_%05d.png, the 12-number trajectory file.  The images are real perspective views of a
textured 3-D surface (synth.Scene3D); the map is seeded from two ground-truth keyframes, everything after that runs through
the hot path on the MI355X (extraction, matching, PnP, bundle adjustment, keyframe insertion)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_struct_equal

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "driver", "run_vo")


def _write_traj(path, poses):
    with open(path, "w") as f:
        for T in poses:
            f.write(" ".join("%.17g" % v for v in np.concatenate([T[:3, 3], T[:3, :3].T.ravel()])) + "\n")


def _read_traj(path):
    rows = np.loadtxt(path).reshape(-1, 12)
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    T[:, :3, 3] = rows[:, :3]
    T[:, :3, :3] = rows[:, 3:].reshape(-1, 3, 3).transpose(0, 2, 1)
    return T


def _write_dataset(mvo, tmp_path, n, k1, extra=""):
    from PIL import Image
    scene = mvo.synth.Scene3D(amp=0.6, tilt=0.3)
    data = tmp_path / "dataset"
    data.mkdir()
    frames = []
    for i in range(n):
        img = scene.frame(i)
        frames.append(img)
        Image.fromarray(img[:, :, ::-1]).save(data / ("rgb_%05d.png" % i))          # PIL wants RGB; the files hold what imread returns as BGR
    truth = [scene.pose(i) for i in range(n)]
    _write_traj(tmp_path / "cam_traj_truth.txt", truth)
    K = scene.K
    cfg = tmp_path / "config.yaml"
    cfg.write_text("""%%YAML:1.0
dataset_name: "synthetic"
synthetic:
  dataset_dir: %s
  num_images: %d
  camera_info.fx: %r
  camera_info.fy: %r
  camera_info.cx: %r
  camera_info.cy: %r
  true_traj_filename: %s
max_num_imgs_to_proc: 300
save_predicted_traj_to: %s
init_keyframe_0: 0
init_keyframe_1: %d
max_number_of_keypoints: 1500
is_ba_fix_map_points: "true"
%s""" % (data, n, K["fx"], K["fy"], K["cx"], K["cy"], tmp_path / "cam_traj_truth.txt", tmp_path / "cam_traj.txt", k1, extra))
    return scene, frames, truth, cfg


def test_run_vo_end_to_end_on_png_frames(mvo, O, tmp_path):
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    n, k1 = 24, 5
    scene, frames, truth, cfg = _write_dataset(mvo, tmp_path, n, k1)
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lost 0" in r.stdout, r.stdout
    est = _read_traj(tmp_path / "cam_traj.txt")
    assert len(est) == n
    gt = np.stack(truth)
    # the seed keyframes carry the ground truth; every later frame is tracked.  This is a plain monocular VO (pose-only BA,
    # two-view triangulation at a few centimetres of baseline in front of a ~2 m deep scene): it drifts along the
    # translation / yaw ambiguity like the reference does -- bounded here at 30 % of the distance travelled and 3 degrees
    # (measured: 18 % / 1.6 deg); the first tracked frames sit within millimetres
    err_t = np.linalg.norm(est[k1:, :3, 3] - gt[k1:, :3, 3], axis=1)
    cosang = (np.einsum("nij,nij->n", est[k1:, :3, :3], gt[k1:, :3, :3]) - 1) / 2
    err_r = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    travelled = np.linalg.norm(gt[-1, :3, 3] - gt[k1, :3, 3])
    assert err_t.max() < 0.3 * travelled and err_r.max() < 3.0, (r.stdout, err_t, err_r)
    assert err_t[1:4].max() < 0.01, err_t[:5]
    assert np.linalg.norm(est[-1, :3, 3] - est[k1, :3, 3]) > 0.3                  # it really moved
    # the decoded frames are what the oracle sees: first-frame keypoints of the run = oracle keypoints on the same pixels
    p = O.default_params(max_keypoints=1500)
    ko = O.calc_keypoints(frames[0], p)
    assert len(ko) > 800
    # determinism: a second run writes the identical trajectory file
    first = (tmp_path / "cam_traj.txt").read_text()
    r2 = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and (tmp_path / "cam_traj.txt").read_text() == first



def test_run_vo_equals_the_oracle_chain(mvo, O, tmp_path):
    """(f)4: the SAME run composed from the oracle (tests/vo_chain.py: extract -> map points in view -> match -> PnP ->
    sliding-window BA -> keyframe insertion, reference run_vo.cpp:110-148 / vo_addFrame.cpp:70-124 / vo_io.cpp:51-77) on the
    same PNG frames and the same two seed keyframes.  Per frame, bit for bit: keypoints, descriptors, the PnP inliers'
    matches (= the inlier set and the map points they hit), the tracking / keyframe decisions, and on keyframes the matches
    with the reference keyframe, the epipolar inliers, the matches kept by the triangulation test and the ids of the map
    after insertion + culling.  Floating point: the trajectory within 1e-4 (pose-only BA is well-posed; the device sums in
    its blocked order, the oracle sequentially), triangulated points / map positions within 1e-5 relative (they are
    computed from those poses and stored as float)."""
    import vo_chain
    n, k1 = 24, 5
    log_path = tmp_path / "frames.log"
    scene, frames, truth, cfg = _write_dataset(mvo, tmp_path, n, k1, 'save_frame_log_to: %s\n' % log_path)
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    log = vo_chain.read_frame_log(log_path)
    assert len(log) == n
    est = _read_traj(tmp_path / "cam_traj.txt")

    def map_order(idx, ids):                       # the host container's iteration order (see vo_chain.py)
        order = np.frombuffer(log[idx]["MORD"], "<i4")
        assert set(order.tolist()) == ids, "frame %d: the map of the run and of the oracle chain hold different points" % idx
        return order

    ch, hist = vo_chain.run_oracle_chain(O, frames, scene.K, truth, 0, k1, O.default_params(max_keypoints=1500),
                                         fix_map_points=True, map_order=map_order)
    n_key = n_tracked = 0
    worst_pose = worst_pt = 0.0
    for i, (rec, fr) in enumerate(zip(log, ch.frames)):
        what = "frame %d: " % i
        assert np.frombuffer(rec["FRAM"], "<i4")[0] == i
        assert_struct_equal(np.frombuffer(rec["KPTS"], O.KEYPOINT_DTYPE), fr.kps, what + "keypoints")
        assert rec["DESC"] == fr.desc.tobytes(), what + "descriptors"
        if i > k1:
            good, is_key = np.frombuffer(rec["FLAG"], "<i4")
            assert (bool(good), bool(is_key)) == (fr.rec["good"], fr.rec["is_keyframe"]), what + "tracking / keyframe decision"
            assert_struct_equal(np.frombuffer(rec["MMAP"], O.DMATCH_DTYPE), fr.rec["matches_with_map"], what + "PnP inlier matches")
            n_tracked += int(good)
        if "MREF" in rec:                              # the seed keyframe k1 and every inserted keyframe
            n_key += 1
            assert_struct_equal(np.frombuffer(rec["MREF"], O.DMATCH_DTYPE), fr.rec["matches_with_ref"], what + "matches_with_ref_")
            assert_struct_equal(np.frombuffer(rec["IREF"], O.DMATCH_DTYPE), fr.rec["inliers_matches_with_ref"], what + "epipolar inliers")
            assert_struct_equal(np.frombuffer(rec["I3DM"], O.DMATCH_DTYPE), fr.rec["inliers_matches_for_3d"], what + "triangulation survivors")
            p_run = np.frombuffer(rec["I3DP"], "<f4").reshape(-1, 3)
            p_orc = fr.rec["inliers_pts3d"]
            worst_pt = max(worst_pt, float((np.abs(p_run - p_orc) / np.abs(p_orc).max(axis=1, keepdims=True)).max()))
            ids = np.frombuffer(rec["MIDS"], "<i4")
            pos = np.frombuffer(rec["MPOS"], "<f4").reshape(-1, 3)
            assert set(ids.tolist()) == set(fr.rec["map_after"]), what + "map after insertion and culling"
            ref_pos = np.stack([fr.rec["map_after"][int(m)] for m in ids])
            worst_pt = max(worst_pt, float((np.abs(pos - ref_pos) / np.abs(ref_pos).max(axis=1, keepdims=True)).max()))
        else:
            assert "map_after" not in fr.rec, what + "the oracle chain inserted a keyframe, the run did not"
        T_run = np.frombuffer(rec["POSE"], "<f8").reshape(4, 4)
        worst_pose = max(worst_pose, float(np.abs(T_run - hist[i]).max()))
    assert n_tracked == n - k1 - 1 and n_key >= 4, (n_tracked, n_key)
    assert worst_pose < 1e-4 and worst_pt < 1e-5, (worst_pose, worst_pt)
    # the trajectory FILE (vo_io.cpp:51-77: 12 numbers per line) carries the same poses
    assert np.abs(est - hist).max() < 1e-4
    print("oracle chain: %d tracked, %d keyframes, max |dT| %.3g, max rel point error %.3g" % (n_tracked, n_key, worst_pose, worst_pt))
