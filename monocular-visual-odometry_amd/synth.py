"""Deterministic synthetic workloads for the VO hot path.

* S640 / S1242 image sequences: windows of one random world texture under a smooth planar camera path
  (the reference's datasets -- data/dataset_images_matlab, TUM fr1_desk -- are not in the repo and there
  is no network), replicated to BGR like cv::imread (run_vo.cpp:114).
* matcher micro-inputs (uniform / perturbed copy / tie-heavy).
* BA5 / BA10 sliding-window problems shaped like callBundleAdjustment_ builds them (vo.cpp:384-478).
"""
import numpy as np

FR1_K = dict(fx=517.3, fy=516.5, cx=325.1, cy=249.7)       # config/config.yaml:40-43
KITTI_K = dict(fx=718.856, fy=718.856, cx=607.19, cy=185.22)


def world_texture(seed=1234, size=2048, n_rect=4000):
    from scipy import ndimage
    rng = np.random.RandomState(seed)
    tex = np.zeros((size, size), np.float64)
    for octave, sigma in enumerate((32.0, 12.0, 4.0, 1.5)):
        n = ndimage.gaussian_filter(rng.uniform(-1, 1, (size, size)), sigma, mode="wrap")
        tex += n / n.std() * (40.0 / (1 + 0.5 * octave))
    tex += 128
    for _ in range(n_rect):
        w, h = rng.randint(6, 60, 2)
        x, y = rng.randint(0, size - w), rng.randint(0, size - h)
        tex[y:y + h, x:x + w] = rng.uniform(20, 235)
    return np.clip(tex, 0, 255)


class Sequence:
    """frames(i) -> HxWx3 uint8 (BGR, the three channels equal)."""

    def __init__(self, width=640, height=480, n_frames=150, seed=1234, tex_size=2048):
        self.w, self.h, self.n = width, height, n_frames
        self.seed = seed
        self.tex = world_texture(seed, tex_size)
        self.tex_size = tex_size

    def frame(self, i, channels=3):
        from scipy import ndimage
        rng = np.random.RandomState((self.seed * 1000003 + i) % (2 ** 32))
        t = i / max(self.n - 1, 1)
        # smooth planar path: ~4 px/frame translation, 0.2 deg/frame roll, +-5 % scale
        cx = self.tex_size / 2 + 4.0 * i * np.cos(0.3) + 60 * np.sin(2 * np.pi * t)
        cy = self.tex_size / 2 + 4.0 * i * np.sin(0.3) * 0.5 + 40 * np.sin(4 * np.pi * t)
        ang = np.deg2rad(0.2 * i)
        sc = 1.0 + 0.05 * np.sin(2 * np.pi * t)
        ys, xs = np.mgrid[0:self.h, 0:self.w].astype(np.float64)
        xs -= self.w / 2
        ys -= self.h / 2
        u = cx + sc * (np.cos(ang) * xs - np.sin(ang) * ys)
        v = cy + sc * (np.sin(ang) * xs + np.cos(ang) * ys)
        img = ndimage.map_coordinates(self.tex, [v, u], order=1, mode="reflect")
        img = img + rng.normal(0, 2.0, img.shape)
        g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if channels == 1:
            return g
        return np.repeat(g[:, :, None], 3, axis=2)


def small_test_image(seed=0, w=160, h=120, channels=3):
    """Corner-rich small image for fast CPU tests."""
    rng = np.random.RandomState(seed)
    img = np.full((h, w), 128.0)
    from scipy import ndimage
    img += ndimage.gaussian_filter(rng.uniform(-1, 1, (h, w)), 3.0) * 200
    for _ in range(max(8, w * h // 500)):
        rw, rh = rng.randint(4, 24, 2)
        x, y = rng.randint(0, w - rw), rng.randint(0, h - rh)
        img[y:y + rh, x:x + rw] = rng.uniform(10, 245)
    img += rng.normal(0, 1.5, img.shape)
    g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if channels == 1:
        return g
    out = np.repeat(g[:, :, None], 3, axis=2)
    # make the channels differ so that BGR2GRAY weights are exercised
    out[:, :, 0] = np.clip(out[:, :, 0].astype(int) + rng.randint(-20, 20, (h, w)), 0, 255)
    out[:, :, 2] = np.clip(out[:, :, 2].astype(int) + rng.randint(-20, 20, (h, w)), 0, 255)
    return np.ascontiguousarray(out)


# ---------------------------------------------------------------- matcher micro-inputs (seed 42)
def match_inputs(kind, nq=2000, nt=2000, seed=42):
    rng = np.random.RandomState(seed)
    if kind == "uniform":
        return rng.randint(0, 256, (nq, 32)).astype(np.uint8), rng.randint(0, 256, (nt, 32)).astype(np.uint8)
    if kind == "perturbed":
        q = rng.randint(0, 256, (nq, 32)).astype(np.uint8)
        n_copy = min(nq, max(nt - nt // 4, 1))
        perm = rng.permutation(nq)[:n_copy]
        bits = np.unpackbits(q[perm], axis=1)
        flips = rng.uniform(size=bits.shape) < 0.08
        t = np.packbits(bits ^ flips, axis=1)
        distract = rng.randint(0, 256, (nt - n_copy, 32)).astype(np.uint8)
        t = np.concatenate([t, distract])[rng.permutation(nt)]
        return q, np.ascontiguousarray(t)
    if kind == "ties":
        base = rng.randint(0, 256, (64, 32)).astype(np.uint8)
        return base[rng.randint(0, 64, nq)].copy(), base[rng.randint(0, 64, nt)].copy()
    raise ValueError(kind)


# ---------------------------------------------------------------- bundle-adjustment problems
def _rot(axis, ang):
    axis = np.asarray(axis, float)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def ba_problem(n_poses=5, n_points=2000, seed=7, width=640, height=480, K=FR1_K, pix_noise=0.5,
               outlier_frac=0.02, pose_rot_noise=0.01, pose_trans_noise=0.01, point_noise=0.01,
               f32_storage=True):
    """Returns dict(poses_gt, poses0 [F,4,4] cam->world, points_gt, points0, edge_pose, edge_point,
    edge_uv, focal, cx, cy).  Frames are listed newest->oldest like vo.cpp:417-421."""
    rng = np.random.RandomState(seed)
    f, cx, cy = K["fx"], K["cx"], K["cy"]
    poses_gt = []
    for i in range(n_poses):
        yaw = np.deg2rad(3.0) * i / max(n_poses - 1, 1)
        R_wc = _rot([0, 1, 0], yaw)
        t_wc = np.array([0.05 * i, 0.005 * np.sin(i), 0.01 * i])
        T = np.eye(4)
        T[:3, :3] = R_wc
        T[:3, 3] = t_wc
        poses_gt.append(T)
    poses_gt = np.array(poses_gt)
    # points uniform in the first camera's frustum, depth 0.5 .. 3 m
    z = rng.uniform(0.5, 3.0, n_points)
    u = rng.uniform(0.05 * width, 0.95 * width, n_points)
    v = rng.uniform(0.05 * height, 0.95 * height, n_points)
    pts = np.stack([(u - cx) / f * z, (v - cy) / f * z, z], 1)
    ep, el, uv = [], [], []
    for i in range(n_poses):
        Tcw = np.linalg.inv(poses_gt[i])
        pc = pts @ Tcw[:3, :3].T + Tcw[:3, 3]
        pu = f * pc[:, 0] / pc[:, 2] + cx
        pv = f * pc[:, 1] / pc[:, 2] + cy
        vis = (pc[:, 2] > 0.1) & (pu >= 0) & (pu < width) & (pv >= 0) & (pv < height)
        ids = np.nonzero(vis)[0]
        ids = ids[rng.permutation(len(ids))]  # unordered_map iteration order is arbitrary
        nz = rng.normal(0, pix_noise, (len(ids), 2))
        out = rng.uniform(size=len(ids)) < outlier_frac
        nz[out] += rng.uniform(-20, 20, (int(out.sum()), 2))
        ep.append(np.full(len(ids), i, np.int32))
        el.append(ids.astype(np.int32))
        uv.append(np.stack([pu[ids], pv[ids]], 1) + nz)
    ep, el, uv = np.concatenate(ep), np.concatenate(el), np.concatenate(uv)
    poses0 = poses_gt.copy()
    for i in range(n_poses):
        dR = _rot(rng.normal(size=3), rng.normal(0, pose_rot_noise))
        poses0[i, :3, :3] = dR @ poses0[i, :3, :3]
        poses0[i, :3, 3] += rng.normal(0, pose_trans_noise, 3)
    points0 = pts + rng.normal(0, point_noise, pts.shape)
    if f32_storage:  # cv::Point2f / cv::Point3f storage in Frame / MapPoint
        uv = uv.astype(np.float32).astype(np.float64)
        points0 = points0.astype(np.float32).astype(np.float64)
    return dict(poses_gt=poses_gt, poses0=poses0, points_gt=pts, points0=points0, edge_pose=ep,
                edge_point=el, edge_uv=uv, focal=f, cx=cx, cy=cy)


# ---------------------------------------------------------------- tracking inputs (vo.cpp:270-329)
def tracking_problem(n_map=3000, seed=11, width=640, height=480, K=FR1_K, pix_noise=0.4, outlier_frac=0.25,
                     behind_frac=0.15, planar=False):
    """A map, a camera pose and the 3D-2D pairs the reference feeds to cv::solvePnPRansac.

    Returns dict(map_pos [M,3] f32, map_desc [M,32] u8, T_w_c [4,4], K, cols, rows, pts3d [n,3] f32,
    pts2d [n,2] f32, inlier_gt [n] bool).  The map has points behind the camera and outside the image so
    that getMappointsInCurrentView_ has something to reject; `outlier_frac` of the pairs are wrong matches."""
    rng = np.random.RandomState(seed)
    f, cx, cy = K["fx"], K["cx"], K["cy"]
    R_wc = _rot([0.2, 1.0, 0.1], np.deg2rad(7.0))
    T = np.eye(4)
    T[:3, :3] = R_wc
    T[:3, 3] = [0.3, -0.05, 0.1]
    # points around the camera: in a generous frustum (some outside the image) plus some behind it
    z = rng.uniform(0.6, 4.0, n_map)
    if planar:
        z = 2.0 + 0.0 * z
    u = rng.uniform(-0.25 * width, 1.25 * width, n_map)
    v = rng.uniform(-0.25 * height, 1.25 * height, n_map)
    pc = np.stack([(u - cx) / f * z, (v - cy) / K["fy"] * z, z], 1)
    behind = rng.uniform(size=n_map) < behind_frac
    pc[behind, 2] *= -1
    pw = (pc @ R_wc.T + T[:3, 3]).astype(np.float32)
    desc = rng.randint(0, 256, (n_map, 32)).astype(np.uint8)
    # the pairs: visible points, measured pixel = projection + noise, a fraction replaced by wrong pixels
    Tcw = np.linalg.inv(T)
    q = pw.astype(np.float64) @ Tcw[:3, :3].T + Tcw[:3, 3]
    pu = f * q[:, 0] / q[:, 2] + cx
    pv = K["fy"] * q[:, 1] / q[:, 2] + cy
    vis = (q[:, 2] > 0) & (pu > 0) & (pv > 0) & (pu < width) & (pv < height)
    ids = np.nonzero(vis)[0]
    ids = ids[rng.permutation(len(ids))]
    uv = np.stack([pu[ids], pv[ids]], 1) + rng.normal(0, pix_noise, (len(ids), 2))
    bad = rng.uniform(size=len(ids)) < outlier_frac
    uv[bad] = np.stack([rng.uniform(0, width, int(bad.sum())), rng.uniform(0, height, int(bad.sum()))], 1)
    return dict(map_pos=pw, map_desc=desc, T_w_c=T, K=K, cols=width, rows=height, pts3d=pw[ids].copy(),
                pts2d=uv.astype(np.float32), inlier_gt=~bad, ids=ids)


# ---------------------------------------------------------------- keyframe inputs (vo_addFrame.cpp:93-124)
def keyframe_problem(n=600, seed=21, width=640, height=480, K=FR1_K, pix_noise=0.3, outlier_frac=0.2, baseline=0.25):
    """Two keyframes looking at n scene points: matched pixels (kp_ref, kp_cur, a fraction of them wrong), the two
    camera poses T_w_c and T_curr_to_prev = inv(T_w_cur) ... as getMotionFromFrame1to2(curr, ref) gives it
    (vo_commons.cpp:9-16), the ground-truth points in the reference camera frame."""
    rng = np.random.RandomState(seed)
    f, cx, cy = K["fx"], K["cx"], K["cy"]
    T_ref = np.eye(4)
    T_ref[:3, :3] = _rot([0, 1, 0], np.deg2rad(2.0))
    T_ref[:3, 3] = [0.1, 0.0, 0.05]
    T_cur = np.eye(4)
    T_cur[:3, :3] = _rot([0.1, 1.0, 0.05], np.deg2rad(6.0))
    T_cur[:3, 3] = T_ref[:3, 3] + np.array([baseline, 0.02, 0.03])
    z = rng.uniform(1.0, 5.0, n)
    u = rng.uniform(0.15 * width, 0.85 * width, n)
    v = rng.uniform(0.15 * height, 0.85 * height, n)
    p_ref = np.stack([(u - cx) / f * z, (v - cy) / K["fy"] * z, z], 1)
    p_w = p_ref @ T_ref[:3, :3].T + T_ref[:3, 3]

    def proj(T):
        Tcw = np.linalg.inv(T)
        q = p_w @ Tcw[:3, :3].T + Tcw[:3, 3]
        return np.stack([f * q[:, 0] / q[:, 2] + cx, K["fy"] * q[:, 1] / q[:, 2] + cy], 1), q

    kp_ref, _ = proj(T_ref)
    kp_cur, q_cur = proj(T_cur)
    kp_ref = kp_ref + rng.normal(0, pix_noise, kp_ref.shape)
    kp_cur = kp_cur + rng.normal(0, pix_noise, kp_cur.shape)
    bad = rng.uniform(size=n) < outlier_frac
    kp_cur[bad] = np.stack([rng.uniform(0, width, int(bad.sum())), rng.uniform(0, height, int(bad.sum()))], 1)
    T_cur_to_ref = np.linalg.inv(T_cur) @ T_ref          # getMotionFromFrame1to2(curr, ref) = T_w_curr^-1 T_w_ref
    return dict(kp_ref=kp_ref.astype(np.float32), kp_cur=kp_cur.astype(np.float32), T_w_ref=T_ref, T_w_cur=T_cur,
                T_curr_to_prev=T_cur_to_ref, K=K, p_ref=p_ref, p_cur=q_cur, inlier_gt=~bad, cols=width, rows=height)


# ---------------------------------------------------------------- feature-level sequence (vo_addFrame.cpp:70-124)
def feature_sequence(n_frames=30, n_points=5000, seed=61, width=640, height=480, K=FR1_K, step=0.03, pix_noise=0.3,
                     bit_flip=0.02, clutter=150, max_kp=1200):
    """A 3-D scene seen from a moving camera, as per-frame keypoints + descriptors (what Frame::calcKeyPoints /
    calcDescriptors would deliver), with ground-truth poses.  Returns dict(K, cols, rows, frames=[dict(T_w_c, xy
    [N,2] f32, desc [N,32] u8, point_id [N] (-1 = clutter))], points [P,3])."""
    rng = np.random.RandomState(seed)
    f, cx, cy = K["fx"], K["cx"], K["cy"]
    pts = np.stack([rng.uniform(-3.0, 4.5, n_points), rng.uniform(-2.0, 2.0, n_points), rng.uniform(1.2, 5.0, n_points)], 1)
    pdesc = rng.randint(0, 256, (n_points, 32)).astype(np.uint8)
    frames = []
    for i in range(n_frames):
        T = np.eye(4)
        T[:3, :3] = _rot([0.05, 1.0, 0.02], np.deg2rad(0.4 * i))
        T[:3, 3] = [step * i, 0.004 * np.sin(0.5 * i), 0.006 * i]
        Tcw = np.linalg.inv(T)
        q = pts @ Tcw[:3, :3].T + Tcw[:3, 3]
        u = f * q[:, 0] / q[:, 2] + cx
        v = K["fy"] * q[:, 1] / q[:, 2] + cy
        vis = np.nonzero((q[:, 2] > 0.3) & (u > 8) & (v > 8) & (u < width - 8) & (v < height - 8))[0]
        vis = vis[rng.permutation(len(vis))][:max_kp]
        xy = np.stack([u[vis], v[vis]], 1) + rng.normal(0, pix_noise, (len(vis), 2))
        bits = np.unpackbits(pdesc[vis], axis=1)
        bits ^= (rng.uniform(size=bits.shape) < bit_flip).astype(np.uint8)
        d = np.packbits(bits, axis=1)
        cxy = np.stack([rng.uniform(8, width - 8, clutter), rng.uniform(8, height - 8, clutter)], 1)
        cd = rng.randint(0, 256, (clutter, 32)).astype(np.uint8)
        order = rng.permutation(len(vis) + clutter)
        frames.append(dict(T_w_c=T, xy=np.concatenate([xy, cxy])[order].astype(np.float32),
                           desc=np.ascontiguousarray(np.concatenate([d, cd])[order]),
                           point_id=np.concatenate([vis, -np.ones(clutter, int)])[order]))
    return dict(K=K, cols=width, rows=height, frames=frames, points=pts)


# ---------------------------------------------------------------- image-level 3-D sequence (headless run_vo)
class Scene3D:
    """A textured, gently undulating surface in front of a moving camera: real perspective images with parallax and their
    ground-truth poses (T_w_c, camera -> world), for the end-to-end run of host/driver/run_vo.

    Surface z = z0 + amp * sin(kx X + 0.7) * cos(ky Y) + tilt * X (metres, camera axes: X right, Y down, Z forward);
    texture = world_texture() with one texel per pixel at depth z0.  Every pixel's ray is intersected with the surface by
    fixed-point iteration (the surface is smooth and far: 8 rounds reach < 1e-6 m)."""

    def __init__(self, width=640, height=480, K=FR1_K, seed=77, z0=2.0, amp=0.22, tilt=0.12, step=0.02, tex_size=2048):
        self.w, self.h, self.K, self.seed = width, height, K, seed
        self.z0, self.amp, self.tilt, self.step = z0, amp, tilt, step
        self.tex = world_texture(seed, tex_size)
        self.tex_size = tex_size
        self.m = z0 / K["fx"]                     # metres per texel

    def surface(self, X, Y):
        return self.z0 + self.amp * np.sin(2.1 * X + 0.7) * np.cos(1.7 * Y) + self.tilt * X

    def pose(self, i):
        T = np.eye(4)
        T[:3, :3] = _rot([0.1, 1.0, 0.05], np.deg2rad(0.12 * i)) @ _rot([0, 0, 1], np.deg2rad(0.1 * i))
        T[:3, 3] = [self.step * i, 0.004 * np.sin(0.4 * i), 0.003 * i]
        return T

    def frame(self, i, noise=1.5):
        from scipy import ndimage
        K, T = self.K, self.pose(i)
        ys, xs = np.mgrid[0:self.h, 0:self.w].astype(np.float64)
        d = np.stack([(xs - K["cx"]) / K["fx"], (ys - K["cy"]) / K["fy"], np.ones_like(xs)], -1) @ T[:3, :3].T
        C = T[:3, 3]
        t = (self.z0 - C[2]) / d[..., 2]
        for _ in range(8):
            X, Y = C[0] + t * d[..., 0], C[1] + t * d[..., 1]
            t = (self.surface(X, Y) - C[2]) / d[..., 2]
        X, Y = C[0] + t * d[..., 0], C[1] + t * d[..., 1]
        u, v = self.tex_size / 2 + X / self.m, self.tex_size / 2 + Y / self.m
        img = ndimage.map_coordinates(self.tex, [v, u], order=1, mode="reflect")
        rng = np.random.RandomState((self.seed * 7919 + i) % (2 ** 32))
        img = img + rng.normal(0, noise, img.shape)
        g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        return np.repeat(g[:, :, None], 3, axis=2)


if __name__ == "__main__":
    # python synth.py render <out.npy> <width> <height> <frames> <seed> <tex_size>: one sequence as a [frames, H, W, 3] uint8 .npy file
    # (bench.py renders the shards' sequences with worker processes of this form)
    import sys
    if len(sys.argv) == 8 and sys.argv[1] == "render":
        out_path = sys.argv[2]
        w, h, n, seed, tex = (int(v) for v in sys.argv[3:8])
        seq = Sequence(w, h, n, seed=seed, tex_size=tex)
        arr = np.lib.format.open_memmap(out_path, mode="w+", dtype=np.uint8, shape=(n, h, w, 3))
        for i in range(n):
            arr[i] = seq.frame(i)
        arr.flush()
    else:
        raise SystemExit("usage: synth.py render <out.npy> <width> <height> <frames> <seed> <tex_size>")
