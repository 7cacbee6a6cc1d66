"""Pins the oracle to the REAL reference stack: run this wherever OpenCV's Python module is importable
(`python tests/golden/make_reference_golden.py`); it writes tests/golden/reference_*.npz with the same schema as
make_golden.py's fixtures, produced by the library calls the reference makes:

  reference_orb_176x144.npz   cv::ORB::create(8000, 1.2, 3, 31, 0, 2, HARRIS_SCORE, 31, 20)->detect
                              (src/geometry/feature_match.cpp:21-34), the grid selection restated below
                              (feature_match.cpp:51-85 -- the reference's own code, not an OpenCV call), ORB->compute
                              (feature_match.cpp:42-48)
  reference_match_150x170.npz cv::BFMatcher(NORM_HAMMING).knnMatch k = 2 (the exact answer) and the reference's
                              FlannBasedMatcher(LshIndexParams(5, 10, 2)) (feature_match.cpp:140, 161, 182) side by side
  reference_ba_3x40.npz       through a g2o Python binding (g2opy) if one is importable -- or WITHOUT one: `--export-ba-text`
                              writes tests/golden/ba_3x40.txt, tests/golden/reference_ba_driver.cpp (linked with the
                              reference's own src/optimization/g2o_ba.cpp: build line in its header) turns it into
                              reference_ba_3x40.txt, `--import-ba-text reference_ba_3x40.txt` makes the .npz

None of this can run in the authoring container (no cv2, no network), so no reference_*.npz is committed yet and
the oracle stays "parity unpinned" (DESIGN.md section 2).  tests/test_reference_golden.py is the pinned-parity check: it
skips while the fixtures are absent and compares the oracle (CPU) and the HIP path (GPU) with them when present."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def keypoints_to_struct(kps):
    out = np.zeros(len(kps), KEYPOINT_DTYPE)
    for i, k in enumerate(kps):
        out[i] = (k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id)
    return out


def grid_select(kps, rows, cols, grid_size=16, per_cell=8, max_keypoints=300):
    """feature_match.cpp:51-85 (note `cnt > max` : max + 1 survive)."""
    grid = np.zeros((rows // grid_size, cols // grid_size), np.int32)
    out = []
    for k in kps:
        r, c = int(k.pt[1]) // grid_size, int(k.pt[0]) // grid_size
        if grid[r, c] < per_cell:
            out.append(k)
            grid[r, c] += 1
            if len(out) > max_keypoints:
                break
    return out


def export_ba_text(path=None):
    """ba_3x40.npz -> the plain-text window tests/golden/reference_ba_driver.cpp reads (points and pixels as float: the
    reference stores them as cv::Point3f / cv::Point2f)."""
    b = np.load(os.path.join(HERE, "ba_3x40.npz"))
    path = path or os.path.join(HERE, "ba_3x40.txt")
    f, cx, cy = [float(v) for v in b["intr"]]
    with open(path, "w") as o:
        o.write("%d %d %d %.17g %.17g %.17g 1 0 0 1 0 1\n" % (len(b["poses0"]), len(b["points0"]), len(b["edge_pose"]), f, cx, cy))
        for T in b["poses0"]:
            o.write(" ".join("%.17g" % v for v in T.ravel()) + "\n")
        for X in b["points0"].astype(np.float32):
            o.write("%.9g %.9g %.9g\n" % tuple(X))
        for p, l, uv in zip(b["edge_pose"], b["edge_point"], b["edge_uv"].astype(np.float32)):
            o.write("%d %d %.9g %.9g\n" % (p, l, uv[0], uv[1]))
    print("wrote", path)
    return 0


def import_ba_text(path):
    """The driver's output (F lines of 16 doubles, L lines of 3 floats) -> reference_ba_3x40.npz."""
    b = np.load(os.path.join(HERE, "ba_3x40.npz"))
    rows = [np.array(line.split(), np.float64) for line in open(path) if line.strip()]
    F, L = len(b["poses0"]), len(b["points0"])
    assert len(rows) == F + L, (len(rows), F, L)
    poses = np.stack(rows[:F]).reshape(F, 4, 4)
    pts = np.stack(rows[F:])
    np.savez_compressed(os.path.join(HERE, "reference_ba_3x40.npz"), poses0=b["poses0"],
                        points0=b["points0"].astype(np.float32).astype(np.float64), edge_pose=b["edge_pose"],
                        edge_point=b["edge_point"], edge_uv=b["edge_uv"].astype(np.float32).astype(np.float64), intr=b["intr"],
                        full50_poses=poses, full50_points=pts)
    print("wrote reference_ba_3x40.npz from", path)
    return 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--export-ba-text":
        return export_ba_text(sys.argv[2] if len(sys.argv) > 2 else None)
    if len(sys.argv) > 2 and sys.argv[1] == "--import-ba-text":
        return import_ba_text(sys.argv[2])
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: nothing written (the oracle stays unpinned)")
        return 1
    import __graft_entry__ as graft
    S = graft.load_package().synth
    # ---- ORB: the image of the oracle fixture, the reference's constructor arguments
    img = S.small_test_image(2024, 176, 144)
    det = cv2.ORB_create(8000, 1.2, 3, 31, 0, 2, cv2.ORB_HARRIS_SCORE, 31, 20)
    kps = grid_select(det.detect(img, None), img.shape[0], img.shape[1])
    ext = cv2.ORB_create(8000, 1.2, 3)
    kps, desc = ext.compute(img, kps)
    np.savez_compressed(os.path.join(HERE, "reference_orb_176x144.npz"), image=img, keypoints=keypoints_to_struct(kps),
                        descriptors=desc, cv2_version=np.array(cv2.__version__))
    # ---- matching: exact 2-NN and the reference's LSH matcher on the same descriptors
    g = np.load(os.path.join(HERE, "match_150x170.npz"))
    out = dict(q=g["q"], t=g["t"], q2=g["q2"], t2=g["t2"], cv2_version=np.array(cv2.__version__))
    for name, (q, t) in dict(ties=(g["q"], g["t"]), perturbed=(g["q2"], g["t2"])).items():
        knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2)
        out[name + "_bf_idx"] = np.array([[m.trainIdx for m in r] for r in knn], np.int32)
        out[name + "_bf_dist"] = np.array([[int(m.distance) for m in r] for r in knn], np.int32)
        lsh = cv2.FlannBasedMatcher(dict(algorithm=6, table_number=5, key_size=10, multi_probe_level=2), {})
        one = lsh.match(q, t)
        out[name + "_lsh_idx"] = np.array([m.trainIdx for m in one], np.int32)
        out[name + "_lsh_dist"] = np.array([int(m.distance) for m in one], np.int32)
        two = lsh.knnMatch(q, t, k=2)
        out[name + "_lsh2_idx"] = np.array([[r[j].trainIdx if j < len(r) else -1 for j in range(2)] for r in two], np.int32)
        out[name + "_lsh2_dist"] = np.array([[int(r[j].distance) if j < len(r) else -1 for j in range(2)] for r in two], np.int32)
    np.savez_compressed(os.path.join(HERE, "reference_match_150x170.npz"), **out)
    # ---- BA: g2o's Levenberg (50 iterations, Huber sqrt(5.991), no fixed pose) if a binding exists
    try:
        import g2o  # noqa: F401
    except ImportError:
        print("no g2o binding: reference_ba_3x40.npz not written")
    else:
        write_ba(g2o)
    print("reference fixtures written to", HERE)
    return 0


def write_ba(g2o):
    """src/optimization/g2o_ba.cpp:193-289 through g2opy: VertexSE3Expmap / VertexSBAPointXYZ (marginalized) /
    EdgeProjectXYZ2UV + RobustKernelHuber() (default delta 1, g2o_ba.cpp:268), OptimizationAlgorithmLevenberg over
    BlockSolver_6_3 with the dense linear solver, 50 iterations.  Poses go in as world -> camera (g2o_ba.cpp:185-190
    inverts T_w_c) and come back inverted again (:298-306)."""
    b = np.load(os.path.join(HERE, "ba_3x40.npz"))
    f, cx, cy = [float(v) for v in b["intr"]]
    opt = g2o.SparseOptimizer()
    opt.set_algorithm(g2o.OptimizationAlgorithmLevenberg(g2o.BlockSolverSE3(g2o.LinearSolverDenseSE3())))
    cam = g2o.CameraParameters(f, np.array([cx, cy]), 0)
    cam.set_id(0)
    opt.add_parameter(cam)
    F, L = len(b["poses0"]), len(b["points0"])
    for i, T in enumerate(b["poses0"]):
        v = g2o.VertexSE3Expmap()
        v.set_id(i)
        Tcw = np.linalg.inv(T)
        v.set_estimate(g2o.SE3Quat(Tcw[:3, :3], Tcw[:3, 3]))
        opt.add_vertex(v)
    for j, X in enumerate(b["points0"]):
        v = g2o.VertexSBAPointXYZ()
        v.set_id(F + j)
        v.set_marginalized(True)
        v.set_estimate(X)
        opt.add_vertex(v)
    for p, l, uv in zip(b["edge_pose"], b["edge_point"], b["edge_uv"]):
        e = g2o.EdgeProjectXYZ2UV()
        e.set_vertex(0, opt.vertex(F + int(l)))
        e.set_vertex(1, opt.vertex(int(p)))
        e.set_measurement(uv)
        e.set_parameter_id(0, 0)
        e.set_information(np.eye(2))
        e.set_robust_kernel(g2o.RobustKernelHuber())
        opt.add_edge(e)
    opt.initialize_optimization()
    opt.optimize(50)
    poses = np.stack([np.linalg.inv(np.vstack([np.hstack([opt.vertex(i).estimate().rotation().matrix(),
                                                          opt.vertex(i).estimate().translation().reshape(3, 1)]), [0, 0, 0, 1]]))
                      for i in range(F)])
    pts = np.stack([opt.vertex(F + j).estimate() for j in range(L)])
    np.savez_compressed(os.path.join(HERE, "reference_ba_3x40.npz"), poses0=b["poses0"], points0=b["points0"],
                        edge_pose=b["edge_pose"], edge_point=b["edge_point"], edge_uv=b["edge_uv"], intr=b["intr"],
                        full50_poses=poses, full50_points=pts)


if __name__ == "__main__":
    sys.exit(main())
