#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame VO hot path (ORB extract + Hamming match + 5-frame BA) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One rank per GPU: started WITHOUT a launcher (no WORLD_SIZE in the environment) and --gpus N > 1, the script starts the N ranks
itself (spawn_ranks: torch.distributed.run on 127.0.0.1 with a free port) and relays rank 0's JSON line.  Every rank owns `--streams` independent sequence shards (S640, seed 1234 + shard id), each
driven by its own host thread through the native frame loop (host/driver/frame_loop.cpp): the path is serial inside
a sequence and embarrassingly parallel across sequences, so this is how one GPU is filled.  A "step" =
`--frames-per-step` (default 10) consecutive frames of every shard of the rank (one batch of synthetic input: 240 frames
with 24 shards; the timed region of the driver's `--steps 20` is then ~1 s instead of 0.13 s, where the start and the
tail of the region -- shards entering and leaving one after the other -- weighed 8 %), per frame:
  extract (image already resident in HBM) -> descriptors stay in HBM -> match against the previous frame's descriptors
  (2-NN Hamming + Lowe ratio + de-dup) -> bundle adjustment of a NEW BA5 window (5 poses / 2000 landmarks / ~10k edges,
  50 LM iterations): every frame the window is marshalled from Frame / MapPoint objects into pointer lists as
  src/vo/vo.cpp:408-449 does, flattened, planned, uploaded, solved and written back (g2o_ba.cpp:172-317); a shard
  rotates through `--windows` distinct windows.
Weak scaling: per-GPU work is fixed.  The only collective is one all_gather of the per-shard trajectories (frames x 12
f64, the refined pose of the newest frame of every window) after the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import json
import os
import re
import sys

# one HIP stream per sequence shard (+ one per shard for uploads): lift the runtime's default of 4 hardware queues
# BEFORE HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

METRIC = "frames/sec (extract+match+5-kf BA), 640x480 / 2000 kp, 1->8 MI355X"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # FP64 vector = matrix peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz; one
#                                v_mfma_f64_16x16x4_f64 (2048 flop) issues every 64 cycles per SIMD (tools/probes/mfma_probe.hip)
FP64_NOTE = ("fp64: the peak is MI355X's FP64 rate, the same 78.6 TFLOP/s for the vector ALU and the matrix cores; of the counted flops "
             "only the Gram chains (pose blocks, Schur complement, the block solver's updates) run on v_mfma_f64_16x16x4_f64 -- their "
             "share of the time is mfma_busy --, the rest is vector-ALU f64 work")
I8_MFMA_PEAK_TOPS = 3944.0     # MI355X_MICROARCH.md: dense I8 matrix peak (>= 3944 TOPS, 16x16x64)
VALU_PEAK_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-ops/s = 7.86e13


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=None,
                    help="sequence shards in flight per GPU; default 32 (the solver slots stay full also on hosts whose per-frame host "
                         "work is slower; 24 suffices on fast ones), 24 with --track (its PnP kernels hold a CU per workgroup: more "
                         "shards only queue)")
    ap.add_argument("--frames-per-step", type=int, default=10,
                    help="consecutive frames every shard advances in one step (a step = one batch: streams x this many frames)")
    ap.add_argument("--ba", default="full", choices=["full", "pose_only"],
                    help="full = points + poses free (reference is_fix_map_pts=false branch, no vertex fixed)")
    ap.add_argument("--ba-mode", default="rebuild", choices=["rebuild", "resident", "none"],
                    help="rebuild (the metric): a new window is marshalled, uploaded and solved every frame; resident: one "
                         "pre-uploaded window re-solved every frame (kernel-side upper bound, round-1 behaviour)")
    ap.add_argument("--ba-cut", default="auto", choices=["auto", "latency", "throughput"],
                    help="mvo_ba_set_mode of the shards: latency = 32 workgroups per BA5 window (shortest solve), throughput = 16 "
                         "(half the CUs per window); auto = throughput when more than 8 shards share the GPU")
    ap.add_argument("--windows", type=int, default=16, help="distinct BA windows per shard (rotated)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default): every shard overlaps extraction+matching of frame i+1 with the BA of frame i (second "
                         "ctx per shard for the uploads); 0: strictly serial frame loop")
    ap.add_argument("--track", action="store_true",
                    help="also run the tracking rows every frame (map points in view -> match against the map -> "
                         "solvePnPRansac, vo.cpp:270-357); off by default: BASELINE.json's metric is extract+match+BA")
    ap.add_argument("--chain", action="store_true",
                    help="the chained loop as the headline workload (what secondary.chained_fps measures): every frame runs solvePnPRansac on "
                         "the new frame's own 3D-2D pairs and ONLY its inliers become the window's edges of that frame (vo.cpp:304-357)")
    ap.add_argument("--keyframe-every", type=int, default=10,
                    help="with --track: run the keyframe row (findEssentialMat inlier filter + triangulation + culling, "
                         "vo_addFrame.cpp:93-118) on every N-th frame")
    ap.add_argument("--frames", type=int, default=150,
                    help="frames of every shard's sequence (SURVEY.md 8d config 2: 150 frames rendered from a 2048 x 2048 texture; cycled)")
    ap.add_argument("--tex-size", type=int, default=2048, help="side of the world texture a sequence is rendered from")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed loop's outputs (the `parity` object)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the single-sequence and resident-window runs")
    ap.add_argument("--cpu-frames", type=int, default=150)
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="also time the oracle with this many host threads, one sequence shard each (-1 = all cores, 0 = skip)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--max-kp", type=int, default=2000)
    ap.add_argument("--ba-poses", type=int, default=5)
    ap.add_argument("--ba-points", type=int, default=2000)
    args = ap.parse_args(argv)
    if args.streams is None:
        args.streams = 24 if args.track else 32
    return args


class FrameLoopCfg(C.Structure):  # host/driver/frame_loop.cpp: struct frame_loop_cfg
    _fields_ = [("ctx", C.c_void_p), ("ctx_ba", C.c_void_p), ("d_frames", C.POINTER(C.c_void_p)), ("n_frames", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("channels", C.c_int32),
                ("max_kp", C.c_int32), ("ba_mode", C.c_int32), ("pipeline", C.c_int32), ("fix_points", C.c_int32),
                ("K4", C.c_double * 4), ("track", C.c_int32), ("keyframe_every", C.c_int32), ("map", C.c_void_p),
                ("n_map", C.c_int32), ("T_w_c", C.c_void_p), ("pts3d", C.c_void_p), ("pts2d", C.c_void_p),
                ("n_pairs", C.c_int32), ("kf_ref", C.c_void_p), ("kf_cur", C.c_void_p), ("kf_n", C.c_int32),
                ("kf_T_curr_to_prev", C.c_void_p), ("kf_T_w_cur", C.c_void_p), ("kf_T_w_ref", C.c_void_p),
                ("ba_throughput", C.c_int32), ("h_frames", C.POINTER(C.c_void_p)), ("chain", C.c_int32)]


class FrameLoopState(C.Structure):
    _fields_ = [("frame_no", C.c_int32), ("n_kp", C.c_int32), ("n_match", C.c_int32), ("n_inliers", C.c_int32),
                ("n_tri", C.c_int32), ("ba_trials", C.c_int64), ("ba_iterations", C.c_int64), ("ba_solves", C.c_int64),
                ("ba_edges", C.c_int64), ("ns_extract", C.c_int64), ("ns_restore", C.c_int64), ("ns_build", C.c_int64),
                ("ns_begin", C.c_int64), ("ns_end", C.c_int64), ("ba_failed_solves", C.c_int64), ("ba_stale_steps", C.c_int64)]


_frame_loop = None


def frame_loop_lib():
    """host/driver/libmvo_frame_loop.so (built by __graft_entry__.build()); fails loudly when missing."""
    global _frame_loop
    if _frame_loop is None:
        path = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "driver", "libmvo_frame_loop.so")
        if not os.path.exists(path):
            raise SystemExit("native frame loop %s is missing -- run __graft_entry__.build()" % path)
        lib = C.CDLL(path)
        lib.frame_loop_create.restype = C.c_void_p
        lib.frame_loop_create.argtypes = [C.POINTER(FrameLoopCfg)]
        lib.frame_loop_add_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        lib.frame_loop_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.frame_loop_get_state.argtypes = [C.c_void_p, C.POINTER(FrameLoopState)]
        lib.frame_loop_destroy.argtypes = [C.c_void_p]
        lib.frame_loop_export_window.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8
        lib.frame_loop_capture.argtypes = [C.c_void_p, C.c_int]
        lib.frame_loop_capture.restype = None
        lib.frame_loop_get_capture.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        _frame_loop = lib
    return _frame_loop


def render_sequences(args, sids):
    """S640 as SURVEY.md 8d config 2 specifies it: every shard its own sequence (seed 1234 + shard), `--frames` (150) frames
    rendered from a `--tex-size` (2048) square texture, BGR.  32 shards x 150 frames are 4800 renderings of ~30 ms plus 32
    textures of ~2 s: done by worker PROCESSES (`python synth.py render ...`: numpy / scipy only, no torch, no HIP; one per shard,
    as many at a time as this process has CPUs) that write .npy files under /dev/shm (or the temp directory); the files are mapped
    and unlinked at once.  Returns {shard id: uint8 array [frames, H, W, 3]}."""
    import shutil
    import subprocess
    import tempfile
    synth_path = os.path.join(graft.PKG_DIR, "synth.py")
    need = len(sids) * args.frames * args.width * args.height * 3
    base = tempfile.gettempdir()
    try:   # (memory-backed when it has the room -- a container's /dev/shm may be a few MB --, else the temp directory)
        if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 1.25 * need + (64 << 20):
            base = "/dev/shm"
    except OSError:
        pass
    tmp = tempfile.mkdtemp(prefix="mvo_bench_frames_", dir=base)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    # (the workers are plain numpy programs: a profiler wrapped around bench.py -- rocprofv3 preloads its tool library into every
    # child -- must not follow them, it would write a trace per worker and slow each by seconds)
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS"))}
    if "LD_PRELOAD" in env:
        keep = [x for x in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in x.lower()]
        if keep:
            env["LD_PRELOAD"] = ":".join(keep)
        else:
            env.pop("LD_PRELOAD")
    env.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    todo = [(sid, os.path.join(tmp, "shard_%d.npy" % sid)) for sid in sids]
    out, running = {}, []
    try:
        pending = list(todo)
        while pending or running:
            while pending and len(running) < max(1, ncpu):
                sid, path = pending.pop(0)
                cmd = [sys.executable, synth_path, "render", path, str(args.width), str(args.height), str(args.frames), str(1234 + sid),
                       str(args.tex_size)]
                running.append((sid, path, subprocess.Popen(cmd, env=env)))
            sid, path, pr = running.pop(0)
            if pr.wait(timeout=900) != 0:
                raise RuntimeError("rendering the sequence of shard %d failed" % sid)
            out[sid] = np.load(path, mmap_mode="r")
    finally:
        for _, _, pr in running:
            pr.kill()
        shutil.rmtree(tmp, ignore_errors=True)               # (the mappings keep the pages until the arrays go)
    return out


def window_pool(mvo, args, shard_id, n):
    """`n` distinct synthetic BA windows of a shard (SURVEY.md 8d config 3 generator, different seeds)."""
    K = mvo.synth.FR1_K if args.width == 640 else mvo.synth.KITTI_K
    return [mvo.synth.ba_problem(args.ba_poses, args.ba_points, seed=7 + 1000 * shard_id + k, width=args.width,
                                 height=args.height, K=K) for k in range(n)]


class Shard:
    """One sequence: its frames resident in HBM, its own ctx/stream(s), its pool of BA windows, its native loop."""

    def __init__(self, mvo, torch, device, shard_id, args, ba_mode, pipeline, frames=None, pool=None, ba_cut=None,
                 host_frames=False, chain=False, fix_points=None, ctxs=None):
        self.mvo = mvo
        self.torch = torch
        self.from_host, self.chain = host_frames, chain
        self.fix_points = (args.ba == "pose_only") if fix_points is None else fix_points
        self.ba_cut = ba_cut or (args.ba_cut if args.ba_cut != "auto" else ("throughput" if args.streams > 8 else "latency"))
        self.id = shard_id
        self.args = args
        if ctxs is not None:
            self.ctx, self.ctx_ba = ctxs
        else:
            self.ctx = mvo.Context(device, max_keypoints=args.max_kp)
            # the BA half of the sequence: a sibling context on the same stream (no hardware queue of its own)
            # (MVO_BENCH_SEPARATE_CTX=1: a full second context with a stream of its own -- the round-2 arrangement, kept for the A/B)
            separate = os.environ.get("MVO_BENCH_SEPARATE_CTX") == "1"
            self.ctx_ba = (mvo.Context(device, max_keypoints=args.max_kp) if separate else self.ctx.sibling()) if pipeline else self.ctx
        if frames is None:
            frames = render_sequences(args, [shard_id])[shard_id]
        if isinstance(frames, np.ndarray):
            # one rendered block [frames, H, W, 3]: ONE upload, the frames are views of it (host and device)
            block = np.array(frames, dtype=np.uint8, order="C")   # (out of the unlinked /dev/shm mapping: its pages go with it)
            dev = torch.from_numpy(block).to("cuda:%d" % device)
            frames = ([block[i] for i in range(len(block))], [dev[i] for i in range(len(block))])
            self._host_block = block
        self.host_frames, self.dev_frames = frames
        self.K = mvo.synth.FR1_K if args.width == 640 else mvo.synth.KITTI_K
        self.pool = pool if pool is not None else window_pool(mvo, args, shard_id, max(1, args.windows))
        self.track = None
        if args.track:            # a resident map + the 3D-2D pairs PnP sees
            tp = mvo.synth.tracking_problem(n_map=3000, seed=11 + shard_id, width=args.width, height=args.height, K=self.K)
            self.track = tp
            self.map = self.ctx.map_create()
            self.ctx.map_upload(self.map, tp["map_pos"], tp["map_desc"])
            self.kf = mvo.synth.keyframe_problem(n=1000, seed=21 + shard_id, width=args.width, height=args.height, K=self.K)
        self.traj = []
        self._setup_native(ba_mode, pipeline)

    def _setup_native(self, ba_mode, pipeline):
        """Everything the native frame loop needs, as plain pointers (kept alive on self)."""
        a = self.args
        c = FrameLoopCfg()
        c.ctx, c.ctx_ba = self.ctx.h, self.ctx_ba.h
        self._fr = (C.c_void_p * len(self.dev_frames))(*[t.data_ptr() for t in self.dev_frames])
        c.d_frames = C.cast(self._fr, C.POINTER(C.c_void_p))
        c.n_frames, c.width, c.height, c.stride, c.channels, c.max_kp = len(self.dev_frames), a.width, a.height, a.width * 3, 3, a.max_kp
        c.ba_mode = {"none": 0, "rebuild": 1, "resident": 2}[ba_mode]
        c.pipeline = 1 if pipeline else 0
        c.fix_points = 1 if self.fix_points else 0
        c.chain = 1 if self.chain else 0
        if self.from_host:   # the same frames in pinned host memory: handed over as host images (H2D inside the loop)
            # (one pinned block per shard, the frames are views of it)
            self._pinned_block = self.torch.from_numpy(np.ascontiguousarray(np.stack(self.host_frames))).pin_memory()
            self._pinned = [self._pinned_block[i] for i in range(len(self.host_frames))]
            self._hf = (C.c_void_p * len(self._pinned))(*[t.data_ptr() for t in self._pinned])
            c.h_frames = C.cast(self._hf, C.POINTER(C.c_void_p))
        # (tracking rows / the PnP chain in the loop: the frames wait for those stages, not for the solver -> SHARED mode)
        c.ba_throughput = (2 if (self.track is not None or self.chain) else 1) if self.ba_cut == "throughput" else 0
        c.track = 1 if self.track is not None else 0
        c.keyframe_every = a.keyframe_every
        for i, k in enumerate(("fx", "fy", "cx", "cy")):
            c.K4[i] = self.K[k]
        if self.track is not None:
            tp, kf = self.track, self.kf
            self._keep = [np.ascontiguousarray(tp["T_w_c"], np.float64), np.ascontiguousarray(tp["pts3d"], np.float32),
                          np.ascontiguousarray(tp["pts2d"], np.float32), np.ascontiguousarray(kf["kp_ref"], np.float32),
                          np.ascontiguousarray(kf["kp_cur"], np.float32), np.ascontiguousarray(kf["T_curr_to_prev"], np.float64),
                          np.ascontiguousarray(kf["T_w_cur"], np.float64), np.ascontiguousarray(kf["T_w_ref"], np.float64)]
            ptr = [x.ctypes.data_as(C.c_void_p) for x in self._keep]
            c.map, c.n_map = self.map, len(tp["map_pos"])
            c.T_w_c, c.pts3d, c.pts2d, c.n_pairs = ptr[0], ptr[1], ptr[2], len(tp["pts3d"])
            c.kf_ref, c.kf_cur, c.kf_n = ptr[3], ptr[4], len(kf["kp_ref"])
            c.kf_T_curr_to_prev, c.kf_T_w_cur, c.kf_T_w_ref = ptr[5], ptr[6], ptr[7]
        self.cfg = c
        lib = frame_loop_lib()
        self.loop = lib.frame_loop_create(C.byref(c))
        if ba_mode != "none":
            for pb in (self.pool if ba_mode == "rebuild" else self.pool[:1]):
                arrs = [np.ascontiguousarray(pb["poses0"], np.float64), np.ascontiguousarray(pb["points0"], np.float64),
                        np.ascontiguousarray(pb["edge_pose"], np.int32), np.ascontiguousarray(pb["edge_point"], np.int32),
                        np.ascontiguousarray(pb["edge_uv"], np.float64)]
                r = lib.frame_loop_add_window(self.loop, len(arrs[0]), len(arrs[1]), len(arrs[2]), *[x.ctypes.data_as(C.c_void_p) for x in arrs])
                if r != 0:
                    raise RuntimeError("frame_loop_add_window failed (%d): %s" % (r, self.ctx_ba.last_error()))

    def state(self):
        st = FrameLoopState()
        frame_loop_lib().frame_loop_get_state(self.loop, C.byref(st))
        return st

    def run(self, n):
        """n frames: one call into the native loop (the GIL is released for its whole duration)."""
        out = np.zeros((n, 12))
        r = frame_loop_lib().frame_loop_run(self.loop, n, out.ctypes.data_as(C.c_void_p))
        if r != 0:
            raise RuntimeError("frame loop failed (%d): %s | %s" % (r, self.ctx.last_error(), self.ctx_ba.last_error()))
        self.traj.extend(list(out))

    def export_window(self, k):
        """Window k of the pool as the loop hands it to the solver (restored, marshalled from the Frame / MapPoint objects,
        flattened: host/driver/frame_loop.cpp:frame_loop_export_window) -> (poses, points, edge_pose, edge_point, edge_uv) and
        (poses_now, points_now): the state the window's objects were in BEFORE this call restored them (for the window of the
        last frame of a run: what its solve wrote back), in the same flat order."""
        pb = self.pool[k]
        F, L, cap = len(pb["poses0"]), len(pb["points0"]), len(pb["edge_pose"])
        poses, pts, poses_now, pts_now = np.zeros((F, 4, 4)), np.zeros((L, 3)), np.zeros((F, 4, 4)), np.zeros((L, 3))
        ep, el, uv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 2))
        fle = np.zeros(3, np.int32)
        r = frame_loop_lib().frame_loop_export_window(self.loop, k, cap, *[x.ctypes.data_as(C.c_void_p)
                                                                          for x in (poses, pts, ep, el, uv, fle, poses_now, pts_now)])
        if r != 0:
            raise RuntimeError("frame_loop_export_window failed (%d)" % r)
        return (poses[:fle[0]], pts[:fle[1]], ep[:fle[2]], el[:fle[2]], uv[:fle[2]]), (poses_now[:fle[0]], pts_now[:fle[1]])

    def captured_frame(self):
        """Keypoints, descriptors and matches of the last frame extracted while frame_loop_capture was on."""
        lib = frame_loop_lib()
        fn, n, nm = C.c_int32(-1), C.c_int32(0), C.c_int32(0)
        lib.frame_loop_get_capture(self.loop, C.byref(fn), C.byref(n), C.byref(nm), None, None, None)
        kps = np.zeros(max(n.value, 1), self.mvo.KEYPOINT_DTYPE)
        desc = np.zeros((max(n.value, 1), 32), np.uint8)
        m = np.zeros(max(nm.value, 1), self.mvo.DMATCH_DTYPE)
        lib.frame_loop_get_capture(self.loop, C.byref(fn), C.byref(n), C.byref(nm), *[x.ctypes.data_as(C.c_void_p) for x in (kps, desc, m)])
        return fn.value, kps[:n.value], desc[:n.value], m[:nm.value]

    def close(self):
        frame_loop_lib().frame_loop_destroy(self.loop)
        self.loop = None
        if self.ctx_ba is not self.ctx:
            self.ctx_ba.close()
        self.ctx.close()


def shard_ids(rank, streams):
    """Sequence shards owned by a rank: disjoint, seeds 1234 + id (no data-path collective needed)."""
    return [rank * streams + s for s in range(streams)]


def max_over_ranks(dist, elapsed, device):
    """Contract: the timed region is the MAX over ranks."""
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_trajectories(dist, traj, device):
    """The one collective of the path: all_gather of the per-shard trajectories [streams, steps, 12] f64
    (row format of vo_io.cpp:58-75) -> [world, streams, steps, 12]."""
    if dist is None:
        return traj[None]
    import torch
    tl = torch.from_numpy(np.ascontiguousarray(traj)).to(device)
    out = [torch.empty_like(tl) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tl)
    return torch.stack(out).cpu().numpy()


def run_steps(shards, n):
    """Every shard advances n frames on its own thread: one call into the native frame loop each."""
    errs = []

    def work(s):
        try:
            s.run(n)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(s,)) for s in shards]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


def timed_run(shards, steps, sync):
    sync()
    t0 = time.perf_counter()
    run_steps(shards, steps)
    for s in shards:
        s.ctx.synchronize()
    sync()
    return time.perf_counter() - t0


def ba_trial_flops(E, L, F, fix_points):
    """Algorithmic flops of one FULL LM trial (SURVEY.md 8d's F_trial): linearize + pose blocks, landmark blocks + Schur, reduced
    solve, back-substitution, chi2."""
    n = E / max(L, 1)
    return 330 * E + (0 if fix_points else L * (40 + 144 * n + 216 * n * (n + 1) / 2) + 200 * L) + (6 * F) ** 3 / 3 + 60 * E


def ba_solve_flops(E, L, F, fix_points, iterations, trials, applied):
    """Algorithmic flops of LM solves, counted by what each stage actually runs (the terms are SURVEY.md 8d's F_trial, split):
    once per ITERATION the linearisation + pose blocks (330 E; g2o does not re-linearise for a retry either); once per TRIAL the
    landmark blocks' inverses + the Schur complement (L (40 + 144 n + 216 n (n + 1) / 2)) and the factorisation ((6 F)^3 / 3);
    only for the trials whose step is APPLIED -- all but the ones that end at a failed factorisation with nothing to apply --
    the landmark back-substitution (200 L) and the robust chi2 (60 E)."""
    n = E / max(L, 1)
    per_trial = (0 if fix_points else L * (40 + 144 * n + 216 * n * (n + 1) / 2)) + (6 * F) ** 3 / 3
    per_applied = (0 if fix_points else 200 * L) + 60 * E
    return iterations * 330 * E + trials * per_trial + applied * per_applied


def algorithmic_work(args):
    """Per-launch algorithmic bytes / lane-ops of the extraction and matching kernels."""
    w, h, K = args.width, args.height, args.max_kp
    lv = [(w, h)]
    for l in range(1, 4):
        s = np.float32(1.2) ** l
        lv.append((int(np.rint(w / s)), int(np.rint(h / s))))
    P = sum(a * b for a, b in lv)
    return {
        "k_pyramid": ("hbm", w * h * 3 + P),                           # BGR in, every level out (frames excluded)
        "k_fast_harris": ("hbm", P + (81 + 749) * 8000 + 16 * 8000),   # levels in, Harris/IC windows in, records out
        "k_brief": ("hbm", (45 * 56 + 32 + 16) * K),                   # raw window in, descriptor out
        "k_blur": ("hbm", 2 * P),                                      # every level in, its blurred copy out (throughput mode)
        "k_brief_sample": ("hbm", (512 + 32 + 16) * K),                # 512 taps of the blurred level in, descriptor out
        "k_knn2": ("valu", 16.0 * K * K),                                # the vector-ALU form (xor + bcnt): 16 lane-ops per pair
        "k_knn2_mfma": ("mfma_i8", 2.0 * 256 * K * K),                   # the form that runs: i8 Gram of the +-1 bit vectors, 2 x 256 ops per pair
    }


def pmc_traffic(kernel, pattern="r[0-9]*_pmc_fetch_write_size_per_kernel.csv"):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary matching `pattern`
    (profiles/r*_pmc_fetch_write_size_per_kernel.csv and its siblings, written by tools/pmc_summary.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes; columns located by the header line).  None if absent."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            cols = None
            for line in open(path):
                if line.startswith("#"):
                    continue
                f = [x.strip().strip('"') for x in line.split(",")]
                if cols is None:
                    cols = {name: i for i, name in enumerate(f)}
                    continue
                if f[0].startswith(kernel):
                    kb = float(f[cols["FETCH_SIZE_KB_avg"]]) + float(f[cols["WRITE_SIZE_KB_avg"]])
                    best = (kb * 1024.0, os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError, IndexError):
            pass
    return best


def kernel_resources(kernel):
    """Registers / spills / scratch of `kernel` as the compiler reported them for the shipped build
    (profiles/r*_kernel_resources.csv, written by tools/kernel_resources.py from hipcc -Rpass-analysis=kernel-resource-usage
    over csrc/).  None if absent."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_resources.csv"))):
        try:
            cols = None
            for line in open(path):
                if line.startswith("#"):
                    continue
                f = [x.strip() for x in line.rstrip("\n").split(";")]
                if cols is None:
                    cols = f
                    continue
                if f[0] == kernel:
                    best = dict(zip(cols[1:], [int(x) if x.lstrip("-").isdigit() else x for x in f[1:]]), source=os.path.relpath(path, ROOT))
        except (OSError, ValueError, IndexError):
            pass
    return best


def pmc_mfma_busy(kernel, pattern="r[0-9]*_pmc_mfma_busy.txt"):
    """SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs of the grid) of `kernel` from the newest committed PMC pass
    (profiles/r*_pmc_mfma_busy.txt; the summary's header names the command).  (fraction, source) or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            cols = None
            for line in open(path):
                if line.startswith("#"):
                    continue
                f = [x.strip() for x in line.split(",")]
                if cols is None:
                    cols = {name: i for i, name in enumerate(f)}
                    continue
                if f[0] == kernel:
                    best = (float(f[cols["SQ_VALU_MFMA_BUSY_CYCLES"]]), float(f[cols["GRBM_GUI_ACTIVE"]]), os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError, IndexError):
            pass
    return best


def pmc_grid_busy(pattern="r[0-9]*_pmc_grid_mfma_busy.txt"):
    """The resident solver grid's matrix-core counters from the newest committed PMC pass ON THE GRID (profiles/r*_pmc_grid_mfma_busy.txt,
    tools/pmc_summary.py grid: SQ_VALU_MFMA_BUSY_CYCLES summed over the grid's dispatches, the windows it solved in that pass, the
    shader cycles they were resident and their workgroups).  dict or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            cols = None
            for line in open(path):
                if line.startswith("#"):
                    continue
                f = [x.strip() for x in line.split(",")]
                if cols is None:
                    cols = f
                    continue
                if f[0] == "k_ba_service":
                    best = dict(zip(cols[1:], [float(x) for x in f[1:]]), source=os.path.relpath(path, ROOT))
        except (OSError, ValueError, IndexError):
            pass
    return best


def rank_cpu_slice(cpus, local_rank, local_world, gpu_numa=None, cpu_numa=None):
    """CPUs the threads of local rank `local_rank` are confined to when `local_world` ranks share one host (one
    rank per GPU, each with ~26 threads -- 24 sequence shards, the solver's scheduler, its completion thread).  Without it the
    8 x 26 threads of a node wander over all cores: a shard thread is woken on the far socket, the pinned mailboxes the
    resident solver grid polls end up on whatever NUMA node the allocating thread happened to run on.
    `cpus`: the CPUs this process may use (sorted).  With NUMA information -- `gpu_numa[r]` = node of rank r's GPU,
    `cpu_numa[c]` = node of CPU c -- a rank gets an equal share of ITS GPU's node (split among the ranks of that node);
    otherwise the CPU list is cut into `local_world` contiguous slices.  Pure function (tested on CPU)."""
    cpus = sorted(cpus)
    if local_world <= 1 or not cpus:
        return cpus
    if gpu_numa is not None and cpu_numa is not None and len(gpu_numa) >= local_world:
        node = gpu_numa[local_rank]
        mine = [c for c in cpus if cpu_numa.get(c) == node]
        peers = [r for r in range(local_world) if gpu_numa[r] == node]
        if mine and len(mine) >= len(peers):
            k = peers.index(local_rank)
            per = len(mine) // len(peers)
            return mine[k * per:(k + 1) * per] if k < len(peers) - 1 else mine[k * per:]
    per = max(1, len(cpus) // local_world)
    lo = min(local_rank * per, len(cpus) - 1)
    return cpus[lo:lo + per] if local_rank < local_world - 1 else cpus[lo:]


def host_numa_maps(local_world):
    """(gpu_numa, cpu_numa) from sysfs, or (None, None): node of every local GPU (by the PCI address torch reports) and of
    every CPU."""
    try:
        import torch
        cpu_numa = {}
        for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
            node = int(os.path.basename(d)[4:])
            for part in open(os.path.join(d, "cpulist")).read().strip().split(","):
                a, _, b = part.partition("-")
                for c in range(int(a), int(b or a) + 1):
                    cpu_numa[c] = node
        gpu_numa = []
        for r in range(local_world):
            pr = torch.cuda.get_device_properties(r)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
            gpu_numa.append(max(node, 0))
        return (gpu_numa, cpu_numa) if cpu_numa else (None, None)
    except Exception:  # noqa: BLE001  (no sysfs, no such attribute: fall back to contiguous slices)
        return None, None


def confine_rank_to_its_cpus(local_rank, local_world, numa=(None, None)):
    """Applied before any context, stream, thread or pinned buffer of the rank exists (all of them inherit it)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = rank_cpu_slice(os.sched_getaffinity(0), local_rank, local_world, *numa)
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus


class GpuEnv:
    """What main() needs from the machine: the process group, the device, barriers and shards.  tests/ substitute a CPU
    stand-in (gloo, stub shards) to execute the N > 1 control flow without a GPU."""
    backend = "nccl"

    def __init__(self, args):
        import torch
        self.torch = torch
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        # N > 1 ranks on one host: every rank keeps to its own CPUs (those of its GPU's NUMA node when sysfs tells), and blocks
        # on an interrupt instead of spinning when its threads outnumber its CPUs.  Measured for ONE rank confined like a rank of
        # an 8-GPU node (32 of 256 CPUs; a round-4 one-off run under `taskset`, MVO_BENCH_WAIT_POLICY=<p> -- its script went when tools/gpu_ab.sh replaced the one-offs): block 5029, yield 4955, spin 4904 frames/s vs 5054 unconfined; the
        # N > 1 run itself is the driver's; the control flow is exercised on CPU by tests/test_distributed_gloo.py
        # (no LOCAL_WORLD_SIZE -- a launcher that does not say how many ranks share this host: no confinement rather than a
        # slice of 1/WORLD_SIZE of the CPUs on a multi-node job)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        self.cpus = confine_rank_to_its_cpus(self.local, local_world, host_numa_maps(local_world) if local_world > 1 else (None, None))
        torch.cuda.set_device(self.local)
        self.device = "cuda"
        self.mvo = graft.load_package()
        self.wait_policy = "auto"
        want = os.environ.get("MVO_BENCH_WAIT_POLICY") or ("block" if self.cpus is not None and len(self.cpus) < args.streams + 4 else "auto")
        if want != "auto":
            try:
                self.mvo.set_wait_policy(self.local, want)
                self.wait_policy = want
            except Exception as e:  # noqa: BLE001  (a runtime that refuses the flag on an active device: keep its default)
                print("[bench] wait policy left at the runtime's default: %s" % e, file=sys.stderr)

    def init_process_group(self, dist):
        dist.init_process_group(self.backend, device_id=self.torch.device("cuda", self.local))

    def sync(self):
        self.torch.cuda.synchronize()

    def render(self, args, sids):
        return render_sequences(args, sids)

    def make_shard(self, shard_id, args, ba_mode, pipeline, **kw):
        return Shard(self.mvo, self.torch, self.local, shard_id, args, ba_mode, pipeline, **kw)


def run_benchmark(args, env):
    """Warm-up, the timed region (barrier + device sync on both sides, MAX over ranks), the one collective.  Returns what
    the report needs; identical control flow for 1 and N ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        env.init_process_group(dist)
    pipeline = args.pipeline == 1
    kw = {"chain": True} if getattr(args, "chain", False) else {}
    sids = shard_ids(rank, args.streams)
    rendered = env.render(args, sids) if hasattr(env, "render") else {}
    shards = [env.make_shard(sid, args, args.ba_mode, pipeline, **(dict(kw, frames=rendered[sid]) if sid in rendered else kw)) for sid in sids]
    rendered = None
    env.sync()

    def barrier():
        env.sync()
        if dist is not None:
            dist.barrier()

    fps_ = max(1, args.frames_per_step)
    nframes = args.steps * fps_          # frames per shard inside the timed region
    run_steps(shards, args.warmup * fps_)
    st_before = [s.state() for s in shards]
    shards[0].ctx.ba_launch_stats(reset=True)
    elapsed_local = timed_run(shards, nframes, barrier)
    launch = shards[0].ctx.ba_launch_stats()
    launch["service_ms"] = shards[0].ctx.ba_service_times()
    launch["elapsed_ms"] = elapsed_local * 1e3
    st_after = [s.state() for s in shards]
    elapsed = max_over_ranks(dist, elapsed_local, env.device)
    frames_total = world * args.streams * nframes
    # ---- the one collective: gather the trajectories (frames x 12 f64 per shard)
    traj = np.stack([np.stack(s.traj[-nframes:]) for s in shards])           # [streams, frames, 12]
    traj_all = gather_trajectories(dist, traj, env.device)
    assert traj_all.shape == (world, args.streams, nframes, 12) and np.isfinite(traj_all).all()
    return dict(world=world, rank=rank, dist=dist, shards=shards, pipeline=pipeline, elapsed=elapsed, value=frames_total / elapsed,
                nframes=nframes,
                launch=launch, st_before=st_before, st_after=st_after, traj_all=traj_all)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU, torch.distributed.run on this node,
    --standalone: the launcher itself picks the rendezvous port on 127.0.0.1 -- no bind-close-reuse race), pass the command line
    through, relay the ranks' other output line by line as it comes (a hung rank is visible, and the whole run is bounded:
    MVO_BENCH_SPAWN_TIMEOUT seconds, default 3600), and return the LAST line that parses as the contract's JSON object.
    MVO_BENCH_RANK_SCRIPT names the script the ranks run (default: this file; tests/ substitute a CPU stand-in that calls
    bench.main with a gloo environment)."""
    import subprocess
    import threading
    script = os.environ.get("MVO_BENCH_RANK_SCRIPT", os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), script] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    limit = float(os.environ.get("MVO_BENCH_SPAWN_TIMEOUT", "3600"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)  # (a process group of its own)
    found = []

    def relay():
        for line in proc.stdout:
            line = line.rstrip("\n")
            parsed = None
            if line.startswith("{"):
                try:
                    parsed = json.loads(line)
                except ValueError:
                    parsed = None
            if isinstance(parsed, dict) and parsed.get("metric") == METRIC and "value" in parsed:
                found.append((parsed, line))
            elif line.strip():
                print(line, file=sys.stderr, flush=True)

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)   # (the launcher AND the ranks it started: the group this call created, nothing else)
        except (ProcessLookupError, PermissionError):
            proc.kill()
        proc.wait()
        raise SystemExit("bench.py: the %d-rank run did not finish within %.0f s" % (args.gpus, limit))
    th.join(timeout=10)
    if rc != 0 or not found:
        raise SystemExit("bench.py: the %d-rank run failed (exit code %d)" % (args.gpus, rc))
    result, line = found[-1]
    print(line)
    return result


def contract_line(args, R):
    """The fields of the driver's contract that do not need a device to fill in."""
    return {"metric": METRIC, "value": R["value"], "unit": "frames/s", "n_gpus": R["world"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": R["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (extract/match) + f64 (BA)", "data": "synthetic"}


def main(argv=None, env=None):
    args = parse(argv)
    if env is None and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    env = env or GpuEnv(args)
    R = run_benchmark(args, env)
    world, rank, dist, shards, pipeline = R["world"], R["rank"], R["dist"], R["shards"], R["pipeline"]
    elapsed, value, launch, st_before, st_after = R["elapsed"], R["value"], R["launch"], R["st_before"], R["st_after"]
    s0 = shards[0]
    if world != args.gpus and rank == 0:
        print("[bench] --gpus %d but the launcher started %d rank(s): n_gpus reports the ranks that ran" % (args.gpus, world), file=sys.stderr)

    result = None
    if rank == 0 and getattr(env, "minimal_report", False):      # CPU stand-in environments (tests/): the contract fields only
        result = contract_line(args, R)
        print(json.dumps(result))
    elif rank == 0:
        plan_wgs = int(s0.ctx_ba.ba_plan()["wgs"]) if args.ba_mode == "rebuild" else 0   # (workgroups of the last timed window of shard 0)
        trials = sum(a.ba_trials - b.ba_trials for a, b in zip(st_after, st_before))
        iters = sum(a.ba_iterations - b.ba_iterations for a, b in zip(st_after, st_before))
        failed = sum(a.ba_failed_solves - b.ba_failed_solves for a, b in zip(st_after, st_before))
        stale = sum(a.ba_stale_steps - b.ba_stale_steps for a, b in zip(st_after, st_before))
        solves = sum(a.ba_solves - b.ba_solves for a, b in zip(st_after, st_before))
        edges = sum(a.ba_edges - b.ba_edges for a, b in zip(st_after, st_before))
        E_avg = edges / max(solves, 1)
        fix = args.ba == "pose_only"
        # ---- dominant kernel: k_ba_lm.  Duration = HIP events around every launch on the stream it is launched on
        # (the library's launch thread); algorithmic flops = what the LM stages of the windows solved in the timed region ran
        # (ba_solve_flops: a trial that ends at the failed factorisation is NOT credited with a back-substitution or a chi2,
        # the linearisation is credited once per iteration).  Bound: FP64 -- most of these flops are vector-ALU f64 work, the matrix
        # cores run the Gram chains (their share: mfma_busy); on MI355X the FP64 vector and matrix peaks are the same figure.
        per_kernel = {}
        if solves:
            applied = trials - (failed - stale)
            flops = ba_solve_flops(E_avg, args.ba_points, args.ba_poses, fix, iters, trials, applied)
            lm_counts = dict(iterations_per_solve=iters / max(solves, 1), trials_per_solve=trials / max(solves, 1),
                             failed_solves_per_solve=failed / max(solves, 1), stale_steps_per_solve=stale / max(solves, 1),
                             flops_per_solve=flops / max(solves, 1),
                             flops_if_every_trial_were_full=trials * ba_trial_flops(E_avg, args.ba_points, args.ba_poses, fix) / max(solves, 1))
            resident = launch.get("resident_windows", 0) > 0.5 * max(launch["windows"], 1)
            if resident:
                # resident solver service: ONE grid (k_ba_service) stays on the device for the whole timed region and its slots
                # pull windows; the kernel's duration is the region, its work the windows solved in it.  Per window: the
                # device-clock duration of its solve (launch["ms"] sums them, launches == windows).
                busy_ms = launch.get("elapsed_ms", elapsed * 1e3)
                roof = dict(bound="mfma", bound_note=FP64_NOTE, achieved=flops / (busy_ms * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s")
                roof["frac"] = roof["achieved"] / roof["peak"]
                avg_ms = launch["ms"] / max(launch["windows"], 1)
                roof.update(kernel="k_ba_service (resident k_ba_lm body)", avg_launch_ms=busy_ms, launches=max(1, launch.get("resident_grid_starts", 1)),
                            windows_per_launch=launch["windows"], avg_window_ms=avg_ms,
                            windows_in_flight=launch["ms"] / max(busy_ms, 1e-9),
                            shader_clock_ghz_under_load=round(launch.get("resident_cycles", 0.0) / max(launch["ms"] * 1e6, 1e-9), 3),
                            avg_window_kcycles=round(launch.get("resident_cycles", 0.0) / max(launch["windows"], 1) / 1e3, 1),
                            algorithmic_per_launch=flops, trials_per_solve=trials / max(solves, 1),
                            resident_windows=launch.get("resident_windows", 0), resident_cycles=launch.get("resident_cycles", 0.0),
                            workgroups_per_window=plan_wgs,
                            launch_thread_ms=launch.get("service_ms"), timed_region_ms=round(launch.get("elapsed_ms", 0.0), 2),
                            note="the resident grid is launched once and spans the timed region: duration = the region, work = the "
                                 "windows its 16 slots solved in it; avg_window_ms = mean solve time of a window on the device clock")
            else:
                avg_ms = launch["ms"] / max(launch["launches"], 1)
                roof = dict(bound="mfma", bound_note=FP64_NOTE, achieved=flops / (launch["ms"] * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s")
                roof["frac"] = roof["achieved"] / roof["peak"]
                roof.update(kernel="k_ba_lm", avg_launch_ms=avg_ms, launches=launch["launches"],
                            windows_per_launch=launch["windows"] / max(launch["launches"], 1),
                            algorithmic_per_launch=flops / max(launch["launches"], 1),
                            trials_per_solve=trials / max(solves, 1), launch_thread_ms=launch.get("service_ms"),
                            timed_region_ms=round(launch.get("elapsed_ms", 0.0), 2))
            # HBM-side bytes per launch from the rocprofv3 PMC passes of the same kind of launch (tools/collect_evidence.sh); the
            # hand-offs are 8-byte accesses, a width the guide's 2x FETCH_SIZE correction is not calibrated for -> reported
            # uncorrected
            config4 = args.ba_poses == 10 and args.width == 1242 and args.ba_points == 4000
            wpl = launch["windows"] / max(launch["launches"], 1)
            if resident:
                # the resident grid: bytes per WINDOW of the profiled run (its dispatches span the run) x the windows of this one.
                # Round 5: the WRITE_SIZE pass next to the never-ending grid hung (three attempts, killed by its bound) -- the newest
                # round's figure is then the launch-path form of the SAME cut and kernel flavour (tools/pmc_grid_traffic.sh):
                # whichever of the two files belongs to the later round is used, and the source says which
                def round_of(t):
                    m = re.search(r"r(\d+)_", os.path.basename(t[1])) if t else None
                    return int(m.group(1)) if m else -1
                trg = pmc_traffic("k_ba_service_per_window")
                trl = pmc_traffic("k_ba_lm_per_window", "r[0-9]*_pmc_launch_path_fetch_write_size.csv")
                trw = trl if round_of(trl) > round_of(trg) else trg
                if trw is trl and trl:
                    trw = (trl[0], trl[1] + " (k_ba_lm<false,32,2> per window: the throughput cut on the launch path under the headline load)")
                tr = (trw[0] * launch["windows"], trw[1]) if trw else None
            elif config4:
                # BASELINE configs[3]: the profiled launches of the same command hold ~3.9 BA10 windows like the ones timed here
                tr = pmc_traffic("k_ba_lm", "r[0-9]*_config4_pmc_fetch_write_size.csv")
            elif args.ba_poses == 5 and not fix:
                # launch path, 5-keyframe windows: bytes of the profiled single-window launch x the windows of a launch here
                trw = pmc_traffic("k_ba_lm", "r[0-9]*_pmc_streams1_fetch_write_size.csv")
                tr = (trw[0] * wpl, trw[1] + " (single-window launch x %.2f windows per launch)" % wpl) if trw else None
            else:
                tr = None  # (no PMC pass of this window shape has been committed)
            roof["traffic"], roof["traffic_source"] = (tr[0], tr[1]) if tr else (None, None)
            # `traffic` and `mfma_busy` are NOT measured by this run: they are read from the committed PMC passes under profiles/
            # (counters cannot be collected from inside the process).  traffic_is_proxy: the bytes are those of ANOTHER kernel flavour
            # than the one this block names (the launch-path form of the same cut standing in for the resident grid)
            roof["traffic_is_proxy"] = bool(resident and tr and "k_ba_lm<" in tr[1])
            roof["counters_from"] = "committed rocprofv3 --pmc passes under profiles/ (see traffic_source / mfma_busy_source), not this run"
            roof["lm"] = lm_counts
            # what the compiler gave the dominant kernel (build remarks of the shipped sources) and how busy its matrix cores
            # were in the PMC pass of that kernel (SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle of the CUs its windows occupy)
            cls = 32 if args.ba_poses <= 5 else (64 if args.ba_poses <= 10 else 0)
            roof["resources"] = kernel_resources("k_ba_service<32,2>" if resident else ("k_ba_lm<false,64,2>" if config4 else "k_ba_lm<false,%d,1>" % cls))
            mb = pmc_mfma_busy("k_ba_lm", "r[0-9]*_config4_pmc_mfma_busy.txt") if config4 else (pmc_mfma_busy("k_ba_lm") if cls == 32 and not fix else None)
            gb = pmc_grid_busy() if resident else None
            if gb and gb.get("resident_cycles", 0) > 0:
                # the kernel this block names: matrix-core busy cycles of the resident grid per SIMD-cycle of the CUs its windows occupied
                # in the PMC pass (windows x their shader cycles x workgroups x 4 SIMDs)
                simd_cycles = gb["resident_cycles"] * max(gb.get("workgroups_per_window", 0), 1) * 4
                roof["mfma_busy"] = round(gb.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4)
                roof["mfma_busy_source"] = gb["source"] + (" (k_ba_service: SQ_VALU_MFMA_BUSY_CYCLES / (resident cycles of its %d windows x %d workgroups x 4 SIMDs))"
                                                          % (int(gb.get("windows", 0)), int(gb.get("workgroups_per_window", 0))))
            elif mb and not resident:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs (the launch's duration in cycles = an eighth of it); the profiled
                # launches: one window of 28 workgroups (latency cut), or ~3.9 BA10 windows of 56
                cus = 3.93 * 56 if config4 else 28
                roof["mfma_busy"] = round(mb[0] / max(mb[1] / 8.0 * cus * 4, 1.0), 4)
                roof["mfma_busy_source"] = mb[2] + " (%s: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x %d SIMDs))" % (
                    "launches of 3.93 BA10 windows on 56 CUs each" if config4 else "single-window launch, 28 CUs", int(cus * 4))
            per_kernel["k_ba_lm"] = (launch.get("elapsed_ms", 0.0) if resident else launch["ms"]) / max(R["nframes"] * args.streams, 1)
        else:
            roof = None
        # ---- per-kernel durations of the other kernels: HIP events on the ctx stream of ONE shard running the serial loop
        # with NOTHING else on the GPU: no solver window is submitted (--ba-mode none for this shard) and the resident solver
        # grid of the headline run has left (mvo_synchronize parks it; next to a resident grid a kernel's blocks wait for its
        # CUs and the events show the wait, not the kernel -- the round-3 line carried 3x the rocprofv3 durations that way)
        for s_ in shards:
            s_.ctx.synchronize()
        time.sleep(0.02)
        env.sync()
        # The shard runs in the MODE of the headline shards (THROUGHPUT with > 8 sequences: candidates interleaved by the host,
        # descriptors from k_blur + k_brief_sample), so `kernels` lists the kernels the headline number launched.
        kshard = env.make_shard(shard_ids(rank, args.streams)[0], args, "none", False,
                                frames=(s0.host_frames, s0.dev_frames), pool=s0.pool, ba_cut=s0.ba_cut)
        kshard.run(3)
        kshard.ctx.profile_enable(True)
        kshard.ctx.profile_reset()
        nprof = 20
        kshard.run(nprof)
        prof = kshard.ctx.profile_get()
        kshard.ctx.profile_enable(False)
        kshard.close()
        work = algorithmic_work(args)
        kern = {}
        for k, (n_, ms) in prof.items():
            if k == "k_ba_lm":
                continue
            per_kernel[k] = ms / nprof
            kind, amount = work.get(k, ("hbm", 0.0))
            avg = ms / max(n_, 1)
            kern[k] = dict(avg_launch_us=round(avg * 1e3, 2), launches_per_frame=round(n_ / nprof, 2),
                           frac=round((amount / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if kind == "hbm" else
                                      (amount / (avg * 1e-3) / 1e12 / (I8_MFMA_PEAK_TOPS if kind == "mfma_i8" else VALU_PEAK_TOPS)), 5), bound=kind)
        if roof is None and per_kernel:
            dom = max(per_kernel, key=per_kernel.get)
            roof = dict(kernel=dom, **kern.get(dom, {}))

        # ---- secondary numbers: one sequence alone (extraction of frame i+1 overlapped with the BA of frame i), and the
        # round-1 style "resident window" rate (kernel-side upper bound: nothing is marshalled or uploaded per frame)
        secondary = {}
        # (N > 1: the other ranks wait in the final barrier while rank 0 reports -- the single-GPU side measurements and the CPU
        # baseline belong to the N = 1 run, as the bench contract says)
        if not args.no_secondary and args.ba_mode == "rebuild" and world == 1:
            frames0 = (s0.host_frames, s0.dev_frames)
            one = env.make_shard(shard_ids(rank, args.streams)[0], args, "rebuild", True, frames=frames0, pool=s0.pool, ba_cut="latency")
            one.run(max(5, args.warmup))
            nsingle = max(100, min(R["nframes"], 300))
            dt = timed_run([one], nsingle, env.sync)
            secondary["single_sequence_fps"] = nsingle / dt
            secondary["single_sequence_note"] = ("1 shard, latency cut of the window (mvo_ba_set_mode LATENCY: 28 workgroups), extraction+matching "
                                                 "of frame i+1 overlapped with the BA of frame i (2 ctx)")
            st1 = one.state()
            nf = max(1, st1.frame_no)
            secondary["single_sequence_host_us_per_frame"] = {
                k: round(getattr(st1, "ns_" + k) / nf / 1e3, 1) for k in ("extract", "restore", "build", "begin", "end")}
            secondary["single_sequence_host_note"] = ("wall-clock of the loop's stages: extract = extract+match call (runs "
                "between begin and end, i.e. during the solve), restore = bench scaffolding that puts the window's Frame/MapPoint "
                "objects back to their initial state, build = buildBundleAdjustmentWindow marshalling, begin = flatten + plan + "
                "upload + submit, end = wait for the solve + scatter into the objects")
            one_serial = env.make_shard(shard_ids(rank, args.streams)[0], args, "rebuild", False, frames=frames0, pool=s0.pool, ba_cut="latency")
            one_serial.run(max(5, args.warmup))
            dt = timed_run([one_serial], nsingle, env.sync)
            secondary["single_sequence_serial_fps"] = nsingle / dt
            one.close()
            one_serial.close()
            res = [env.make_shard(s.id, args, "resident", False, frames=(s.host_frames, s.dev_frames), pool=s.pool[:1])
                   for s in shards[:12]]
            run_steps(res, max(5, args.warmup))
            nres = max(20, R["nframes"] // 4)
            dt = timed_run(res, nres, env.sync)
            secondary["resident_window_fps"] = len(res) * nres / dt
            secondary["resident_window_note"] = ("round-1 mode, %d shards, serial frame loop: ONE pre-uploaded window re-solved "
                                                 "per frame (no marshalling / upload)" % len(res))
            for r_ in res:
                r_.close()

            def variant(**kw):
                """The headline configuration (same shards, frames and window pools) with one thing changed."""
                vs = [env.make_shard(s.id, args, "rebuild", pipeline, frames=(s.host_frames, s.dev_frames), pool=s.pool, **kw)
                      for s in shards]
                run_steps(vs, max(5, args.warmup))
                nv = max(40, R["nframes"] // 4)
                dt_ = timed_run(vs, nv, env.sync)
                inl = vs[0].state().n_inliers
                for v_ in vs:
                    v_.close()
                return len(vs) * nv / dt_, inl

            secondary["h2d_inclusive_fps"], _ = variant(host_frames=True)
            secondary["h2d_inclusive_note"] = ("every frame handed over as a HOST image (pinned, %d B) like run_vo.cpp:114 -> "
                                               "mvo_calc_keypoints uploads it inside the loop" % (args.width * args.height * 3))
            if args.ba == "full":
                secondary["pose_only_ba_fps"], _ = variant(fix_points=True)
                secondary["pose_only_ba_note"] = "the shipped default is_ba_fix_map_points: true (config.yaml:123, vo.cpp:395-421)"
            fps_c, inl_c = variant(chain=True)
            secondary["chained_fps"] = fps_c
            secondary["chained_note"] = ("every frame: solvePnPRansac on the new frame's own 3D-2D pairs (a quarter of them wrong), pose "
                                         "from it, ONLY its inliers (%d) become the frame's map-point connections (vo.cpp:304-357), "
                                         "then the window is marshalled and solved (vo.cpp:408-449)" % inl_c)

        # host wall-clock per frame of the loop's stages in the HEADLINE run (mean over the shards): shows which stage the
        # shards wait in (extract = extraction + matching call, end = wait for the solve + scatter)
        nfr = max(1, sum(a.frame_no - b.frame_no for a, b in zip(st_after, st_before)))
        secondary["headline_host_us_per_frame"] = {
            k: round(sum(getattr(a, "ns_" + k) - getattr(b, "ns_" + k) for a, b in zip(st_after, st_before)) / nfr / 1e3, 1)
            for k in ("extract", "restore", "build", "begin", "end")}
        per_shard = [sum(getattr(a, "ns_" + k) - getattr(b, "ns_" + k) for k in ("extract", "restore", "build", "begin", "end")) / 1e6
                     for a, b in zip(st_after, st_before)]
        secondary["headline_shard_busy_ms"] = {"min": round(min(per_shard), 1), "mean": round(sum(per_shard) / len(per_shard), 1),
                                               "max": round(max(per_shard), 1), "timed_region_ms": round(elapsed * 1e3, 1)}
        skip_cpu = args.no_cpu_baseline or world > 1
        cpu = None if skip_cpu else cpu_baseline(args, shards[0])
        cpu_mt = None
        nthr = (os.cpu_count() or 1) if args.cpu_threads < 0 else args.cpu_threads
        if not skip_cpu and nthr > 1:
            cpu_mt = cpu_baseline_threads(args, shards[0], nthr)
        # ---- parity of what the timed loop produced (oracle = the checker; outside the timing).  LAST of the side measurements: the
        # check restores a window of every measured shard and advances shard 0 by two frames -- nothing that reads the shards'
        # state runs after it
        parity = None
        if not args.no_parity:
            try:
                parity = parity_check(args, shards, R["nframes"])
            except Exception as e:  # noqa: BLE001  (the checker must never cost the measurement its line)
                parity = {"error": "parity check did not complete: %r" % (e,)}
        st = st_after[0]
        result = {
            "metric": METRIC, "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (extract/match) + f64 (BA)", "data": "synthetic",
            "config": {"workload": "S%d: %dx%d BGR frames resident in HBM (SURVEY 8d config 2: every shard its own sequence -- seed 1234 + shard --"
                                   " of %d frames rendered from a %dx%d texture, cycled), <=%d kp "
                                   "(ORB 8000 -> grid), 2-NN Hamming + "
                                   "Lowe 0.8 + de-dup vs previous frame, BA%d %s with the BA window rebuilt per frame "
                                   "(%d distinct SYNTHETIC windows per shard rotated -- SURVEY 8d config 3 generator, independent of "
                                   "the extracted frames; secondary.chained_fps is the loop whose windows come from PnP inliers --; marshalled from Frame/MapPoint objects like "
                                   "vo.cpp:408-449, flattened, uploaded, solved, written back: %d poses / %d landmarks / "
                                   "~%d edges, 50 LM iterations)"
                                   % (args.width, args.width, args.height, args.frames, args.tex_size, args.tex_size, args.max_kp + 1, args.ba_poses, args.ba,
                                      len(s0.pool), args.ba_poses, args.ba_points, int(E_avg))
                       if args.ba_mode == "rebuild" else "S%d extract+match, BA mode %s" % (args.width, args.ba_mode),
                       "streams_per_gpu": args.streams, "frames_per_step": args.streams * world * max(1, args.frames_per_step),
                       "step": "%d consecutive frames of each of the %d sequence shards (timed region = %d frames)"
                               % (max(1, args.frames_per_step), args.streams * world, args.streams * world * R["nframes"]),
                       "frame_loop": "native (host/driver/frame_loop.cpp)" + (", extraction of frame i+1 overlapped with BA of frame i (a ctx + its sibling per sequence, THROUGHPUT mode when > 8 sequences)" if pipeline else ""),
                       "keypoints": st.n_kp, "matches": st.n_match,
                       # (`value` = frames resident in HBM when the timed region starts, as the measurement contract defines it; the same
                       # loop with every frame handed over as a pinned HOST image like run_vo.cpp:114 -- H2D inside the loop:)
                       "pcie_inclusive_frames_per_s": secondary.get("h2d_inclusive_fps"),
                       # (copy of the top-level `parity` object's verdicts: what the timed loop produced, held to the oracle)
                       "parity": parity and {k: parity.get(k) for k in ("ba", "ba_windows_checked", "ba_workgroups_per_window", "orb", "match", "error")
                                             if k in parity},
                       "ba_trials_per_solve": trials / max(solves, 1),
                       "tracking_rows": ("map in view (3000 pts) + match vs map + solvePnPRansac (%d pairs, %d inliers) every "
                                         "frame; keyframe row (findEssentialMat filter on 1000 matches + triangulation + "
                                         "culling -> %d points) every %d frames"
                                         % (len(s0.track["pts3d"]), st.n_inliers, st.n_tri, args.keyframe_every))
                       if args.track else "off"},
            "roofline": roof,
            "parity": parity,
            "cpu_baseline": cpu,
            "cpu_baseline_all_threads": cpu_mt,
            "secondary": secondary,
            "kernels": kern,
            "kernel_ms_per_frame": {k: round(v, 5) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])},
        }
        print(json.dumps(result))
    for s in shards:
        s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


def traj_row(T):
    """4 x 4 cam->world pose -> the 12 numbers of a trajectory row (x y z, then R column by column: vo_io.cpp:58-75)."""
    return np.concatenate([T[:3, 3], T[:3, :3].T.reshape(-1)])


def parity_check(args, shards, nframes, max_windows=32):
    """Holds the outputs of the TIMED loop to the oracle (test infrastructure, outside the timing, like cpu_baseline):
    * BA: for every shard, the window of its last timed frame -- exported as the loop marshalled it -- is solved by the blocked
      oracle with the summation plan that solve ran with (mvo_debug_get_ba_plan of the shard's BA context: the cut and kernel
      flavour the headline number used); the trajectory row the timed loop wrote, all poses and the landmarks it scattered back
      into the MapPoint objects (f32) must equal the oracle's bit for bit (g2o_ba.cpp:193-289, 298-316);
    * ORB + matching: shard 0 advances two more frames with capture on; the keypoints, descriptors and matches its loop held
      are compared with the oracle's on the same two images (feature_match.cpp:11-49, 126-260)."""
    O = graft.load_oracle()
    out = {"checked_by": "oracle/ (blocked BA oracle with the device's summation plan; ORB / matcher oracle)"}
    if args.ba_mode == "rebuild" and not args.track and not getattr(args, "chain", False):
        bad, cuts, n_checked = [], set(), 0
        fix = args.ba == "pose_only"
        for s in shards[:max_windows]:
            k = (s.state().frame_no - 1) % len(s.pool)
            plan = s.ctx_ba.ba_plan()
            (P0, X0, ep, el, uv), (P_now, X_now) = s.export_window(k)
            pb = s.pool[k]
            Po, Xo, sto, _ = O.bundle_adjustment_blocked(P0, X0, ep, el, uv, pb["focal"], pb["cx"], pb["cy"], plan=plan, fix_points=fix)
            row = np.asarray(s.traj[-1])
            ok = (np.array_equal(row, traj_row(Po[0])) and np.array_equal(P_now, Po)
                  and (fix or np.array_equal(X_now.astype(np.float32), Xo.astype(np.float32))))
            n_checked += 1
            cuts.add(int(plan["wgs"]))
            if not ok:
                bad.append(dict(shard=s.id, window=k, max_pose_diff=float(np.abs(P_now - Po).max()), trials_oracle=sto["trials"]))
        out["ba"] = "bit-exact" if not bad else "MISMATCH"
        out["ba_windows_checked"] = n_checked
        out["ba_workgroups_per_window"] = sorted(cuts)
        out["ba_note"] = ("last timed frame of every shard: trajectory row, all %d poses and the landmarks written back (f32) vs "
                          "oracle/ba_blocked_oracle.cpp with the plan of that solve" % args.ba_poses)
        if bad:
            out["ba_mismatches"] = bad[:4]
    s0 = shards[0]
    lib = frame_loop_lib()
    lib.frame_loop_capture(s0.loop, 1)
    s0.run(2)
    lib.frame_loop_capture(s0.loop, 0)
    fn, kps, desc, m = s0.captured_frame()
    p = O.default_params(max_keypoints=args.max_kp)
    feats = []
    for f in (fn - 1, fn):
        img = s0.host_frames[f % len(s0.host_frames)]
        ko = O.calc_keypoints(img, p)
        feats.append(O.calc_descriptors(img, ko, p))
    mo = O.match_features(feats[0][1], feats[1][1], 2, 2.0, 0.8)
    ko, do = feats[1]
    out["orb"] = "bit-exact" if (len(kps) == len(ko) and kps.tobytes() == ko.tobytes() and np.array_equal(desc, do)) else "MISMATCH"
    out["match"] = "bit-exact" if m.tobytes() == mo.tobytes() else "MISMATCH"
    out["orb_match_note"] = ("frame %d of shard %d as its loop extracted and matched it (%d keypoints, %d matches) vs oracle/orb_oracle.cpp, "
                             "oracle/match_oracle.cpp on the same images" % (fn, s0.id, len(kps), len(m)))
    return out


def cpu_baseline(args, shard):
    """The oracle ("port": our scalar restatement; the reference itself cannot be built here) on the host cores,
    one thread (the reference is single-threaded), on a bounded sample of the same workload."""
    O = graft.load_oracle()
    p = O.default_params(max_keypoints=args.max_kp)
    kw = dict(fix_points=args.ba == "pose_only")
    prev = None
    n = 0
    t0 = time.perf_counter()
    budget = 25.0
    while n < args.cpu_frames and time.perf_counter() - t0 < budget:
        img = shard.host_frames[n % len(shard.host_frames)]
        k = O.calc_keypoints(img, p)
        k, d = O.calc_descriptors(img, k, p)
        if prev is not None:
            O.match_features(prev, d, 2, 2.0, 0.8)
        prev = d
        if args.ba_mode != "none":
            pb = shard.pool[n % len(shard.pool)]
            O.bundle_adjustment(pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"],
                                pb["cx"], pb["cy"], **kw)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same S%d + BA%d workload (windows rotated), oracle -O3 x86-64-v3, 1 thread, %.1f s on a "
                      "%d-core host" % (n, args.width, args.ba_poses, dt, os.cpu_count() or 0)}


def cpu_baseline_threads(args, shard, nthreads):
    """Secondary CPU number: the same oracle, one independent sequence per host thread
    (the GPU side also fills the chip with independent sequences); ctypes releases the GIL inside the oracle."""
    O = graft.load_oracle()
    p = O.default_params(max_keypoints=args.max_kp)
    kw = dict(fix_points=args.ba == "pose_only")
    per_thread = 4 if nthreads > 32 else max(4, min(16, args.cpu_frames // 8))  # (bounded: a few tens of seconds of wall clock)
    done = []

    def work(tid):
        prev = None
        for n in range(per_thread):
            img = shard.host_frames[(n + tid) % len(shard.host_frames)]
            k = O.calc_keypoints(img, p)
            k, d = O.calc_descriptors(img, k, p)
            if prev is not None:
                O.match_features(prev, d, 2, 2.0, 0.8)
            prev = d
            pb = shard.pool[(n + tid) % len(shard.pool)]
            O.bundle_adjustment(pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"],
                                pb["cx"], pb["cy"], **kw)
        done.append(per_thread)

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "frames/s", "cores": nthreads, "kind": "port",
            "sample": "%d threads x %d frames, %.1f s on a %d-core host" % (nthreads, per_thread, dt, os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
