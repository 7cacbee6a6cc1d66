"""The reference's wire formats through the C++ mirror: config.yaml subset incl. the dataset
sections, image path formatting, the 12-number trajectory format (src/vo/vo_io.cpp:13-117).  Host only, no GPU."""
import os
import subprocess

import numpy as np

from conftest import ROOT

BIN = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "tests", "test_io")

CONFIG = """%YAML:1.0
# ===== Select dataset =====
dataset_name: "fr1_desk"
# dataset_name: "matlab"

matlab: # New Tsukuba
  dataset_dir: data/dataset_images_matlab
  num_images: 150
  camera_info.fx: 615
  camera_info.fy: 615
  camera_info.cx: 320
  camera_info.cy: 240
  is_draw_true_traj: "true" # comment with: colon

fr1_desk: # fr1_desk dataset
  dataset_dir: /data/fr1
  num_images: 98
  camera_info.fx: 517.3
  camera_info.fy: 516.5
  camera_info.cx: 325.1
  camera_info.cy: 249.7

max_num_imgs_to_proc: 300
max_number_of_keypoints: 1234
scale_factor: 1.25
lowe_method_dist_ratio: 0.8
is_ba_fix_map_points: "true" # TO DEBUG
information_matrix: "1.0 0.0 0.0 1.0"
findEssentialMat_prob: 0.999
"""


def test_config_and_trajectory_round_trip(tmp_path):
    assert os.path.exists(BIN), "run __graft_entry__.build()"
    cfg, tin, tout = tmp_path / "config.yaml", tmp_path / "in.txt", tmp_path / "out.txt"
    cfg.write_text(CONFIG)
    rng = np.random.RandomState(0)
    rows = []
    for _ in range(7):
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        t = rng.normal(size=3)
        rows.append(np.concatenate([t, q.T.ravel()]))          # tx ty tz, then the columns of R
    np.savetxt(tin, np.array(rows), fmt="%.17g")
    r = subprocess.run([BIN, str(cfg), str(tin), str(tout)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = dict(l.split("=", 1) for l in r.stdout.strip().splitlines())
    assert out["dataset_name"] == "fr1_desk" and out["dataset_dir"] == "/data/fr1" and out["num_images"] == "98"
    assert [float(v) for v in out["K"].split()] == [517.3, 516.5, 325.1, 249.7]
    assert out["max_number_of_keypoints"] == "1234" and float(out["scale_factor"]) == 1.25
    assert out["lowe_method_dist_ratio_as_int"] == "1"          # get<int>(0.8) rounds like the reference latches it
    assert out["is_ba_fix_map_points"] == "1" and out["information_matrix"] == "1.0 0.0 0.0 1.0"
    assert out["image2"] == "/data/fr1/rgb_00002.png" and out["poses"] == "7" and out["missing_key_throws"] == "1"
    back = np.loadtxt(tout)
    assert back.shape == (7, 12) and np.abs(back - np.array(rows)).max() < 1e-5   # operator<< prints 6 digits


def test_reference_config_parses_if_present(tmp_path):
    ref = "/root/reference/config/config.yaml"
    if not os.path.exists(ref):
        return                                                   # the GPU box has no reference checkout
    tin = tmp_path / "in.txt"
    tin.write_text("0 0 0 1 0 0 0 1 0 0 0 1\n")
    r = subprocess.run([BIN, ref, str(tin), str(tmp_path / "o.txt")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = dict(l.split("=", 1) for l in r.stdout.strip().splitlines())
    assert out["dataset_name"] == "matlab" and out["num_images"] == "150"
    assert [float(v) for v in out["K"].split()] == [615, 615, 320, 240]
    assert out["max_number_of_keypoints"] == "1500" and out["findEssentialMat_prob"] == "0.999"


def test_image_reader_decodes_png_and_pnm_like_pil(tmp_path):
    """basics::imread (host/include/my_slam/basics/image_io.h) = cv::imread stand-in of the headless run_vo: BGR, 8 bit,
    3 channels, for gray / RGB / RGBA PNGs (all scanline filters occur in these images) and binary PGM / PPM."""
    import subprocess
    from PIL import Image
    exe = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "driver", "read_image")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    rng = np.random.RandomState(3)
    yy, xx = np.mgrid[0:61, 0:83]
    smooth = ((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 120).astype(np.uint8)
    rgb = np.stack([smooth, np.roll(smooth, 5, 1), rng.randint(0, 256, smooth.shape).astype(np.uint8)], -1)
    cases = {"gray.png": Image.fromarray(smooth), "rgb.png": Image.fromarray(rgb),
             "rgba.png": Image.fromarray(np.dstack([rgb, np.full(smooth.shape, 200, np.uint8)]), "RGBA"),
             "gray.pgm": Image.fromarray(smooth), "rgb.ppm": Image.fromarray(rgb)}
    for name, im in cases.items():
        path, out = tmp_path / name, tmp_path / (name + ".raw")
        im.save(path)
        r = subprocess.run([exe, str(path), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr)
        raw = np.fromfile(out, np.uint8)
        w, h = np.frombuffer(raw[:8].tobytes(), "<i4")
        got = raw[8:].reshape(h, w, 3)
        want = np.asarray(im.convert("RGB"))[:, :, ::-1]            # BGR like cv::imread
        assert (w, h) == im.size and np.array_equal(got, want), name
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not an image")
    assert subprocess.run([exe, str(bad), str(tmp_path / "x.raw")]).returncode == 1      # empty Mat, like cv::imread
