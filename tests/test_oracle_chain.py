"""The oracle chain of tests/vo_chain.py (the run of run_vo composed from the CPU oracle) on its own: it must track the
synthetic sequence, insert keyframes and grow / prune the map -- otherwise comparing the GPU run with it
(tests/test_gpu_run_vo.py) would prove nothing."""
import numpy as np

import vo_chain


def test_oracle_chain_follows_the_ground_truth(mvo, O):
    n, k1 = 16, 5
    scene = mvo.synth.Scene3D(amp=0.6, tilt=0.3)
    frames = [scene.frame(i) for i in range(n)]
    truth = [scene.pose(i) for i in range(n)]
    ch, est = vo_chain.run_oracle_chain(O, frames, scene.K, truth, 0, k1, O.default_params(max_keypoints=1500))
    gt = np.stack(truth)
    err_t = np.linalg.norm(est[k1:, :3, 3] - gt[k1:, :3, 3], axis=1)
    travelled = np.linalg.norm(gt[-1, :3, 3] - gt[k1, :3, 3])
    assert err_t.max() < 0.3 * travelled and err_t[1:4].max() < 0.01, err_t
    tracked = [fr for fr in ch.frames[k1 + 1:]]
    assert all(fr.rec["good"] for fr in tracked)
    assert sum(fr.rec["is_keyframe"] for fr in tracked) >= 3
    assert all(len(fr.rec["matches_with_map"]) > 300 for fr in tracked)
    assert len(ch.map) > 500
    # an arbitrary container order changes which RANSAC subsets are drawn, not the quality of the track
    rng = np.random.RandomState(3)
    ch2, est2 = vo_chain.run_oracle_chain(O, frames, scene.K, truth, 0, k1, O.default_params(max_keypoints=1500),
                                          map_order=lambda idx, ids: rng.permutation(sorted(ids)))
    assert np.abs(est2 - est).max() < 0.02


def test_frame_log_reader(tmp_path):
    p = tmp_path / "log.bin"
    with open(p, "wb") as f:
        for i in range(3):
            for tag, payload in (("FRAM", np.array([i, i], "<i4").tobytes()), ("POSE", np.eye(4).tobytes()), ("MORD", b"")):
                f.write(tag.encode() + np.array([len(payload)], "<i8").tobytes() + payload)
    log = vo_chain.read_frame_log(p)
    assert len(log) == 3 and np.frombuffer(log[2]["FRAM"], "<i4")[0] == 2 and log[1]["MORD"] == b""
