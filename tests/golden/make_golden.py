"""Regenerates tests/golden/*.npz from the CPU ORACLE (run in the authoring container: `python tests/golden/make_golden.py`).

The reference is C++ on OpenCV/g2o (not importable, not buildable here), so these fixtures are NOT outputs of the
reference: they freeze the oracle's canonical arithmetic (parity unpinned) so that neither the
oracle nor the HIP path can drift silently.  Inputs are stored with the outputs; sizes are kept tiny."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    O = graft.load_oracle()
    S = graft.load_package().synth
    # ---- ORB: one 176x144 BGR image, 3 levels
    img = S.small_test_image(2024, 176, 144)
    p = O.default_params(nlevels=3, max_keypoints=300)
    cand = O.candidates(img, p)
    k = O.calc_keypoints(img, p)
    k2, d, rgb = O.calc_descriptors(img, k, p, want_rgb=True)
    np.savez_compressed(os.path.join(HERE, "orb_176x144.npz"), image=img, candidates=cand, keypoints=k2, descriptors=d,
                        rgb=rgb, level2_blurred=O.pyramid_level(img, p, 2, True))
    # ---- matching: 150 x 170 descriptors with ties
    q, t = S.match_inputs("ties", 150, 170, seed=5)
    q2, t2 = S.match_inputs("perturbed", 150, 170, seed=6)
    idx, dist = O.match_knn2(q, t)
    out = dict(q=q, t=t, idx=idx, dist=dist, q2=q2, t2=t2)
    for m in (1, 2):
        out["m%d" % m] = O.match_features(q2, t2, m, 2.0, 1.0)
    np.savez_compressed(os.path.join(HERE, "match_150x170.npz"), **out)
    # ---- BA: 3 poses / 40 landmarks, pose-only (50 it) and full (3 it)
    pb = S.ba_problem(3, 40, seed=77)
    args = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    P1, _, st1 = O.bundle_adjustment(*args, fix_points=True)
    P2, X2, st2 = O.bundle_adjustment(*args, fix_points=False, max_iterations=3)
    np.savez_compressed(os.path.join(HERE, "ba_3x40.npz"), poses0=pb["poses0"], points0=pb["points0"],
                        edge_pose=pb["edge_pose"], edge_point=pb["edge_point"], edge_uv=pb["edge_uv"],
                        intr=np.array([pb["focal"], pb["cx"], pb["cy"]]), pose_only_poses=P1,
                        pose_only_chi2=np.array([st1["chi2_initial"], st1["chi2_final"]]), full3_poses=P2,
                        full3_points=X2, full3_chi2=np.array([st2["chi2_initial"], st2["chi2_final"]]))
    # ---- tracking rows: 300 map points, ~100 3D-2D pairs with 30 % wrong matches, 40 RANSAC iterations
    tp = S.tracking_problem(n_map=300, seed=99, outlier_frac=0.3)
    k4 = np.array([tp["K"][k] for k in ("fx", "fy", "cx", "cy")])
    vis_idx, vis_px = O.map_in_view(tp["map_pos"], tp["T_w_c"], tp["K"], tp["cols"], tp["rows"])
    res = O.solve_pnp_ransac(tp["pts3d"], tp["pts2d"], tp["K"], iters=40)
    np.savez_compressed(os.path.join(HERE, "track_300.npz"), map_pos=tp["map_pos"], map_desc=tp["map_desc"],
                        T_w_c=tp["T_w_c"], K4=k4, size=np.array([tp["cols"], tp["rows"]]), view_idx=vis_idx,
                        view_px=vis_px, pts3d=tp["pts3d"], pts2d=tp["pts2d"], subsets=O.pnp_subsets(len(tp["pts3d"]), 40),
                        models=res["models"], counts=res["counts"], best_iter=res["best_iter"], iters_run=res["iters_run"],
                        inliers=res["inliers"], rvec=res["rvec"], tvec=res["tvec"])
    # ---- keyframe row: 120 matches with 30 % wrong ones
    kf = S.keyframe_problem(n=120, seed=77, outlier_frac=0.3)
    kk4 = np.array([kf["K"][k] for k in ("fx", "fy", "cx", "cy")])
    er = O.find_essential_inliers(kf["kp_ref"], kf["kp_cur"], kf["K"])
    Tk = kf["T_curr_to_prev"]
    pp, pc = O.triangulate_points(kf["kp_ref"][er["inliers"]], kf["kp_cur"][er["inliers"]], kf["K"], Tk[:3, :3], Tk[:3, 3])
    keep, ang = O.retain_good_triangulation(pc, kf["T_w_cur"], kf["T_w_ref"])
    np.savez_compressed(os.path.join(HERE, "keyframe_120.npz"), kp_ref=kf["kp_ref"], kp_cur=kf["kp_cur"], K4=kk4,
                        T_w_ref=kf["T_w_ref"], T_w_cur=kf["T_w_cur"], T_curr_to_prev=Tk, inliers=er["inliers"],
                        counts=er["counts"][:er["iters_run"]], best=np.array([er["best_iter"], er["best_model"]]),
                        pts_prev=pp, pts_curr=pc, keep=keep, angles=ang)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
